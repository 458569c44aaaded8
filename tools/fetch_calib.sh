#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/csrc/fetch_calib.hip) -- run from the repo root on the GPU box.
# One counter set per rocprofv3 run (PMC slot budget), --kernel-trace only.  usage: tools/fetch_calib.sh OUTDIR
set -u
R=$(pwd)
OUT=${1:-$R/gpurun_out/fetch_calib}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_sum TCC_BUBBLE_sum" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    i=$((i + 1))
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pass$i -o cal -- $R/tools/fetch_calib > $OUT/requested.json 2> $OUT/pass$i.err
done
python $R/tools/fetch_calib.py $OUT $OUT/fetch_calibration.json > $OUT/fetch_calibration.txt 2>&1
cat $OUT/fetch_calibration.txt
