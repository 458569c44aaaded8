mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_index_build.py tests/test_gpu_sw.py -m gpu -x -q 2>&1 | tail -4
python bench.py --no-cpu --no-p1000 --no-index-check --steps 12 --chunk 7500 > gpurun_out/r04/b_chk.json 2> gpurun_out/r04/b_chk.err
python -c "
import json
d=json.load(open('gpurun_out/r04/b_chk.json')); print(round(d['value'],1), d['setup_s'])"
