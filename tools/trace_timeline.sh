#!/bin/bash
# a short kernel trace of the bench and, for every stream with many kernels, a timeline of N dispatches from the middle of the run
# usage: tools/trace_timeline.sh OUTDIR [N] -- bench args
OUT=$1; N=${2:-260}; shift; shift; shift
R=$(pwd)
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python $R/bench.py --no-cpu --no-children --no-index-check "$@" > $OUT/bench.json 2> $OUT/bench.err
DB=$(find $OUT/trace -name '*results.db' | head -1)
python3 $R/tools/stream_gaps.py $DB > $OUT/stream_gaps.txt 2>&1
for s in $(grep '^stream' $OUT/stream_gaps.txt | awk '{print $2}' | tr -d ':' | head -6); do
    python3 $R/tools/stream_gaps.py $DB --timeline $s -1 $N > $OUT/timeline_$s.txt 2>&1
done
rm -rf $OUT/trace
tail -2 $OUT/stream_gaps.txt
