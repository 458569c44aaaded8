#!/bin/bash
# round 5, host-side step: the tests that go through the aggregation / records / TSV code on the device path, then the 1 000-proteome
# bench with a rank's CPU share (SD_CPUS=2) beside the box's quota, interleaved
mkdir -p gpurun_out/r05j
( time timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli.py tests/test_gpu_distributed.py -m gpu -x -q ) > gpurun_out/r05j/pytest_subset.txt 2>&1
tail -5 gpurun_out/r05j/pytest_subset.txt
BENCH_ARGS="--steps 12 --warmup 3" bash tools/bench_env.sh r05j "SD_CPUS=2" "-" "SD_CPUS=2" 2>&1 | tee gpurun_out/r05j/bench_env.txt
python - <<'PY'
import json
for i in (1, 2, 3):
    try:
        m = json.load(open('gpurun_out/r05j/d%d.json' % i))['main']
        print(i, 'host_cpu_s_per_step', m['host_cpu_s_per_step'], m['host_cpu_by_stage'], [(t['name'], t['cpu_s_per_step']) for t in m['host_cpu_threads'][:4]])
    except Exception as ex:
        print(i, 'no detail', ex)
PY
