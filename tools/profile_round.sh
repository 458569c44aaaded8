#!/bin/bash
# The round's profile evidence in one GPU-box call (writes under gpurun_out/$TAG/; copy what is to be judged to profiles/):
#   1. VALU issue micro-benchmark                                   -> valu_calibration.json
#   2. rocprofv3 --kernel-trace --stats of the bench command        -> kernel_trace_stats.txt + bench_under_trace.json
#   3. rocprofv3 --pmc passes (one counter set per run, with --kernel-trace only) of `bench.py --steps 1 --warmup 0` at the
#      bench's default --chunk (the launches the bench line times; the headline workload, 1 000 proteomes):
#      FETCH_SIZE, WRITE_SIZE, the per-size read request counters -> pmc_traffic.{txt,json};  SQ_INSTS_VALU -> instr_per_cell in
#      valu_calibration.json
#   4. FETCH_SIZE / WRITE_SIZE calibration on known byte counts     -> fetch_calib/fetch_calibration.{txt,json}
# usage: tools/profile_round.sh TAG [bench args...]      (run from the repo root on the GPU box)
set -u
TAG=${1:-r05}; shift
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/valu_peak.py issue $OUT/valu_calibration.json > $OUT/valu_issue.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py --steps 6 --warmup 1 --no-cpu --no-children --detail-out $OUT/bench_under_trace_detail.json "$@" > $OUT/bench_under_trace.json 2> $OUT/bench_under_trace.err
python $R/tools/rocprof_summary.py $(find $OUT/trace -name '*results.db' | head -1) > $OUT/kernel_trace_stats.txt 2>&1
python $R/tools/stream_gaps.py $(find $OUT/trace -name '*results.db' | head -1) > $OUT/stream_gaps.txt 2>&1
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
    i=$((i + 1))
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$i -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-children --detail-out $OUT/pmc_${i}_detail.json "$@" > $OUT/pmc_$i.json 2> $OUT/pmc_$i.err
done
F=$(find $OUT/pmc_1 -name '*counter_collection.csv' | head -1)
W=$(find $OUT/pmc_2 -name '*counter_collection.csv' | head -1)
V=$(find $OUT/pmc_3 -name '*counter_collection.csv' | head -1)
X=$(find $OUT/pmc_4 -name '*counter_collection.csv' | head -1)
python $R/tools/pmc_summary.py $F $W $OUT/pmc_traffic.json $X > $OUT/pmc_traffic.txt 2>&1
python $R/tools/valu_peak.py pmc $V $OUT/pmc_3.json $OUT/valu_calibration.json > $OUT/valu_pmc.log 2>&1
cd $R && tools/fetch_calib.sh $OUT/fetch_calib > $OUT/fetch_calib.log 2>&1
# the raw traces are large: keep the summaries only
rm -rf $OUT/trace $OUT/pmc_1 $OUT/pmc_2 $OUT/pmc_3 $OUT/pmc_4 $OUT/fetch_calib/pass*
ls -la $OUT
