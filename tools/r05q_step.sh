#!/bin/bash
# round 5, last call: (1) two ranks of bench.py --gpus 2 on one GPU (SD_BENCH_REHEARSAL=1: every rank on cuda:0, the records over gloo) -- the N > 1
# code path of the bench with the records built inside the stream and handed over in one buffer; (2) the 1 000-proteome bench under a rank's CPU
# share beside the box's quota on the final tree; (3) the distributed GPU tests
O=gpurun_out/r05q; mkdir -p $O
( time SD_BENCH_REHEARSAL=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 6 --warmup 1 --proteomes 100 --no-children --no-cpu --detail-out $O/rehearsal_detail.json > $O/bench_rehearsal_n2.json 2> $O/bench_rehearsal_n2.err ) 2> $O/rehearsal.time
tail -c 600 $O/bench_rehearsal_n2.json; tail -3 $O/bench_rehearsal_n2.err; cat $O/rehearsal.time
BENCH_ARGS="--steps 12 --warmup 3" bash tools/bench_env.sh r05q "SD_CPUS=2" "-" "SD_CPUS=2" "-" 2>&1 | grep -v "pipeline ms\|isolated" | cut -c1-260
( time timeout 600 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q ) 2>&1 | grep -E "passed|failed|error|real"
