#!/bin/bash
# round 5: the queue of aggregation jobs (SD_AGG_DEPTH, default 3) -- pipeline / CLI / distributed GPU tests, then 1 000 proteomes under a
# rank's CPU share with the queue and with the old hand-over (depth 1), and the box's quota
O=gpurun_out/r05r; mkdir -p $O
( time timeout 300 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli.py tests/test_gpu_distributed.py -m gpu -x -q ) 2>&1 | grep -E "passed|failed|error|real" | tee $O/pytest_subset.txt
BENCH_ARGS="--steps 12 --warmup 3" bash tools/bench_env.sh r05r "SD_CPUS=2" "SD_CPUS=2 SD_AGG_DEPTH=1" "-" 2>&1 | grep -v "pipeline ms\|isolated" | cut -c1-260 | tee $O/bench_env.txt
