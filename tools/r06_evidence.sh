#!/bin/bash
# round 6: the driver's bench command, then two PMC passes (FETCH_SIZE, WRITE_SIZE; --pmc with --kernel-trace only) of a one-step run whose
# bench line says how many queries its process ran through the prefilter: profiles/<TAG>_pmc_traffic.json holds bytes PER QUERY.
# usage: tools/r06_evidence.sh TAG [bench|pmc|trace|tests ...]   (default: bench pmc)
TAG=${1:-r06a}; shift
WHAT=${*:-bench pmc}
R=$(pwd); OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for w in $WHAT; do
case $w in
tests)
    ( time python -m pytest tests -m gpu -q -x ) > $OUT/pytest_full.txt 2>&1
    grep -E "passed|failed|error" $OUT/pytest_full.txt | tail -3 > gpurun_out/${TAG}_pytest_gpu.txt; grep real $OUT/pytest_full.txt >> gpurun_out/${TAG}_pytest_gpu.txt
    python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/${TAG}_pytest_gpu.txt 2>&1
    cat gpurun_out/${TAG}_pytest_gpu.txt ;;
bench)
    ( time python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out gpurun_out/${TAG}_bench_detail.json > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err ) 2> gpurun_out/${TAG}_bench_default.time
    cat gpurun_out/${TAG}_bench_default.time; tail -c 1500 gpurun_out/${TAG}_bench_default.json; tail -5 gpurun_out/${TAG}_bench_default.err ;;
trace)
    ( cd /tmp && export TMPDIR=/tmp
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py --steps 6 --warmup 1 --no-cpu --no-children --detail-out $OUT/bench_under_trace_detail.json > $OUT/bench_under_trace.json 2> $OUT/bench_under_trace.err
    python $R/tools/rocprof_summary.py $(find $OUT/trace -name '*results.db' | head -1) > $R/gpurun_out/${TAG}_kernel_trace_stats.txt 2>&1
    python $R/tools/stream_gaps.py $(find $OUT/trace -name '*results.db' | head -1) > $R/gpurun_out/${TAG}_stream_gaps.txt 2>&1
    rm -rf $OUT/trace )
    head -14 gpurun_out/${TAG}_kernel_trace_stats.txt | cut -c1-150 ;;
pmc)
    ( cd /tmp && export TMPDIR=/tmp
    i=0
    for c in "FETCH_SIZE" "WRITE_SIZE"; do
        i=$((i + 1))
        timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$i -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-children --detail-out $OUT/pmc_${i}_detail.json > $OUT/pmc_$i.json 2> $OUT/pmc_$i.err
    done
    F=$(find $OUT/pmc_1 -name '*counter_collection.csv' | head -1)
    W=$(find $OUT/pmc_2 -name '*counter_collection.csv' | head -1)
    python $R/tools/pmc_summary.py $F $W $R/gpurun_out/${TAG}_pmc_traffic.json - $OUT/pmc_1.json > $R/gpurun_out/${TAG}_pmc_traffic.txt 2>&1
    rm -rf $OUT/pmc_1 $OUT/pmc_2 )
    tail -20 gpurun_out/${TAG}_pmc_traffic.txt ;;
esac
done
