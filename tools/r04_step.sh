mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_prefilter.py tests/test_gpu_sw.py tests/test_gpu_pipeline.py tests/test_gpu_edge_cases.py -m gpu -x -q 2>&1 | tail -4
python bench.py --no-cpu --no-p1000 --no-index-check --steps 12 --chunk 7500 > gpurun_out/r04/b_pool.json 2> gpurun_out/r04/b_pool.err
python bench.py --no-cpu --no-p1000 --no-index-check --steps 12 --chunk 7500 > gpurun_out/r04/b_pool2.json 2> gpurun_out/r04/b_pool2.err
python - <<'P'
import json
for f in ('b_pool','b_pool2'):
    try:
        d=json.load(open('gpurun_out/r04/%s.json'%f))
        print(f, round(d['value'],1), round(d['ms_per_step'],1), {k:round(v) for k,v in d['roofline']['stage_kernel_ms'].items()}, d['host_cpu_s_per_step'], round(d['device_memory']['resident_GB'],1), d['results']['clusters'])
    except Exception as e:
        print(f, 'ERR', e); print(open('gpurun_out/r04/%s.err'%f).read()[-500:])
P
