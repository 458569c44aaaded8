#!/bin/bash
# the round's closing evidence in one GPU-box call when the GPU budget is short: GPU tests, smoke, the kernel trace of the bench command,
# two PMC passes (FETCH_SIZE, WRITE_SIZE; --pmc with --kernel-trace only), the driver's bench command
TAG=${1:-r05p}
R=$(pwd); OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
( time python -m pytest tests -m gpu -q ) > $OUT/pytest_full.txt 2>&1
grep -E "passed|failed|error" $OUT/pytest_full.txt > gpurun_out/${TAG}_pytest_gpu.txt; grep real $OUT/pytest_full.txt >> gpurun_out/${TAG}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.txt 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out gpurun_out/${TAG}_bench_detail.json > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err ) 2> gpurun_out/${TAG}_bench_default.time
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py --steps 6 --warmup 1 --no-cpu --no-children --detail-out $OUT/bench_under_trace_detail.json > $OUT/bench_under_trace.json 2> $OUT/bench_under_trace.err
python $R/tools/rocprof_summary.py $(find $OUT/trace -name '*results.db' | head -1) > $OUT/kernel_trace_stats.txt 2>&1
python $R/tools/stream_gaps.py $(find $OUT/trace -name '*results.db' | head -1) > $OUT/stream_gaps.txt 2>&1
rm -rf $OUT/trace
i=0
for c in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$i -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-children --detail-out $OUT/pmc_${i}_detail.json > $OUT/pmc_$i.json 2> $OUT/pmc_$i.err
done
F=$(find $OUT/pmc_1 -name '*counter_collection.csv' | head -1)
W=$(find $OUT/pmc_2 -name '*counter_collection.csv' | head -1)
python $R/tools/pmc_summary.py $F $W $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
rm -rf $OUT/pmc_1 $OUT/pmc_2
cd $R
cat gpurun_out/${TAG}_pytest_gpu.txt gpurun_out/${TAG}_smoke.txt gpurun_out/${TAG}_bench_default.time
tail -c 1200 gpurun_out/${TAG}_bench_default.json
head -12 $OUT/kernel_trace_stats.txt | cut -c1-150
