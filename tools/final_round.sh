#!/bin/bash
# everything the round's numbers come from, in one GPU-box call: GPU tests, smoke, profile evidence, the default bench line
TAG=${1:-r02}
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > gpurun_out/${TAG}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.txt 2>&1
bash tools/profile_round.sh ${TAG} > gpurun_out/${TAG}_profile_round.txt 2>&1
cp gpurun_out/${TAG}/valu_calibration.json profiles/${TAG}_valu_calibration.json
cp gpurun_out/${TAG}/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
( time python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err ) 2> gpurun_out/${TAG}_bench_default.time
cat gpurun_out/${TAG}_pytest_gpu.txt gpurun_out/${TAG}_smoke.txt gpurun_out/${TAG}_bench_default.time
tail -c 1500 gpurun_out/${TAG}_bench_default.json
