#!/bin/bash
# per-kernel times of one command (rocprofv3 --kernel-trace), summary to stdout.   usage: tools/ktrace.sh OUTDIR -- command...
OUT=$1; shift; shift
R=$(pwd)
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -o trace -- "$@" > $OUT/trace_run.log 2>&1
python3 $R/tools/rocprof_summary.py $(find $OUT/trace -name '*results.db' | head -1)
rm -rf $OUT/trace
