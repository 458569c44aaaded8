#!/bin/bash
# one GPU-box call of the round-4 (second half) loop: SW / edge tests, a short default-size bench, isolated kernel times
TAG=${1:-s1}
O=gpurun_out/r04b/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sw.py tests/test_gpu_edge_cases.py -m gpu -x -q 2>&1 | tail -4 > $O/pytest.txt
cat $O/pytest.txt
python bench.py --no-cpu --no-p1000 --no-p10000 --no-iter3 --no-index-check --steps 12 > $O/bench.json 2> $O/bench.err
python -c "
import json
d=json.load(open('$O/bench.json')); print('value', round(d['value'],1), 'gcups', round(d['sw_gcups']), d['stage_wall_s'])
ks=d['kernels']; print({k: round(v['ms']) for k, v in sorted(ks.items(), key=lambda kv: -kv[1]['ms'])[:14]})"
bash tools/isolated_times.sh $O > $O/iso_head.txt 2>&1
python tools/iso_sum.py $O/isolated_kernel_times.txt 2>&1 | tail -12
