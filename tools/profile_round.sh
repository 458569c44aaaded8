#!/bin/bash
# The round's profile evidence in one GPU-box call (writes under gpurun_out/$TAG/; copy what is to be judged to profiles/):
#   1. VALU issue micro-benchmark                                   -> valu_calibration.json
#   2. rocprofv3 --kernel-trace --stats of the bench command        -> kernel_trace_stats.txt + bench_under_trace.json
#   3. rocprofv3 --pmc passes (one counter set per run, with --kernel-trace only) of `bench.py --steps 1 --warmup 0` at the
#      bench's default --chunk (the launches the bench line times):
#      FETCH_SIZE, WRITE_SIZE -> pmc_traffic.{txt,json};  SQ_INSTS_VALU -> instr_per_cell in valu_calibration.json
# usage: tools/profile_round.sh TAG      (run from the repo root on the GPU box)
set -u
TAG=${1:-r02}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/valu_peak.py issue $OUT/valu_calibration.json > $OUT/valu_issue.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py --steps 4 --no-cpu --no-p1000 > $OUT/bench_under_trace.json 2> $OUT/bench_under_trace.err
python $R/tools/rocprof_summary.py $(find $OUT/trace -name '*results.db' | head -1) > $OUT/kernel_trace_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-p1000 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
done
F=$(find $OUT/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)
W=$(find $OUT/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
V=$(find $OUT/pmc_SQ_INSTS_VALU -name '*counter_collection.csv' | head -1)
python $R/tools/pmc_summary.py $F $W $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
python $R/tools/valu_peak.py pmc $V $OUT/pmc_SQ_INSTS_VALU.json $OUT/valu_calibration.json > $OUT/valu_pmc.log 2>&1
# the raw traces are large: keep the summaries only
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_INSTS_VALU
ls -la $OUT
