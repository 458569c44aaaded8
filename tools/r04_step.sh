mkdir -p gpurun_out/r04
for g in 0 1 2 3 4; do
SD_PF_HF=$g SD_DEBUG_TIMING=1 python bench.py --no-cpu --no-p1000 --steps 8 > gpurun_out/r04/b_hf$g.json 2> gpurun_out/r04/b_hf$g.err
done
python - <<'P'
import json
for g in range(5):
    f='b_hf%d'%g
    try:
        d=json.load(open('gpurun_out/r04/%s.json'%f))
        ks=d['kernels']
        print(f, round(d['value'],1), round(d['ms_per_step'],1), {k:round(v) for k,v in d['roofline']['stage_kernel_ms'].items()}, 'inpipe hf/seg/part/bm', [round(ks.get(k,{'ms':0})['ms']) for k in ('prefilter_hot_filter','prefilter_segment_match','prefilter_partition_hits','prefilter_bucket_match')])
        iso=d['roofline'].get('isolated',{})
        print(' iso', round(iso.get('kernel_ms'),1), {k[10:]:v for k,v in iso.get('kernel_ms_by_name').items() if k[10:] in ('hot_filter','segment_match','partition_hits','bucket_match')})
    except Exception as e:
        print(f, 'ERR', e)
P
for g in 0 1 2 3 4; do grep "hot filter" gpurun_out/r04/b_hf$g.err | tail -1; done
