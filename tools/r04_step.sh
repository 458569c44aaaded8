mkdir -p gpurun_out/r04
for v in "a SD_PF_LANES=3" "b SD_ALIGN_LANES=3" "c SD_PF_LANES=3,SD_ALIGN_LANES=3" "d SD_PF_LANES=4,SD_ALIGN_LANES=3" "e SD_PF_LANES=3,SD_ALIGN_LANES=3,SD_PF_BATCH=4096"; do
set -- $v
env $(echo $2 | tr ',' ' ') python bench.py --no-cpu --no-p1000 --no-index-check --steps 12 --chunk 7500 > gpurun_out/r04/b_l$1.json 2> gpurun_out/r04/b_l$1.err
done
python bench.py --no-cpu --no-p1000 --no-index-check --steps 12 --chunk 7500 > gpurun_out/r04/b_lf.json 2> gpurun_out/r04/b_lf.err
python - <<'P'
import json
for g in 'abcdef':
    f='b_l%s'%g
    try:
        d=json.load(open('gpurun_out/r04/%s.json'%f))
        print(f, round(d['value'],1), round(d['ms_per_step'],1), {k:round(v) for k,v in d['roofline']['stage_kernel_ms'].items()}, d['host_cpu_s_per_step'], round(d['device_memory']['resident_GB'],1), d['results']['clusters'])
    except Exception as e:
        print(f, 'ERR', e); print(open('gpurun_out/r04/%s.err'%f).read()[-500:])
P
