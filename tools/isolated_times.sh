#!/bin/bash
# Isolated (serialised) per-kernel times of the bench step: a --pmc pass makes rocprofv3 run the dispatches one at a time.
# usage: tools/isolated_times.sh OUTDIR [bench args...]
OUT=$1; shift
R=$(pwd)
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES --kernel-trace -d $OUT/iso -o iso -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-children --chunk 30000 "$@" > $OUT/iso_bench.json 2> $OUT/iso_bench.err
python3 $R/tools/rocprof_summary.py $(find $OUT/iso -name '*results.db' | head -1) > $OUT/isolated_kernel_times.txt 2>&1
if [ -n "$ISO_DISPATCHES" ]; then python3 $R/tools/rocprof_summary.py $(find $OUT/iso -name '*results.db' | head -1) --dispatches "$ISO_DISPATCHES" > $OUT/dispatches.txt 2>&1; fi
rm -rf $OUT/iso
head -45 $OUT/isolated_kernel_times.txt | cut -c1-75,93-140
tail -1 $OUT/isolated_kernel_times.txt
