mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_sw.py tests/test_gpu_pipeline.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -15 > gpurun_out/r04/t2.txt
cat gpurun_out/r04/t2.txt
bash tools/isolated_times.sh gpurun_out/r04/iso_a > /dev/null 2>&1
SD_SW_LW16=0 bash tools/isolated_times.sh gpurun_out/r04/iso_b > /dev/null 2>&1
SD_SW_LW16=8 bash tools/isolated_times.sh gpurun_out/r04/iso_c > /dev/null 2>&1
for x in a b c; do python tools/iso_sum.py gpurun_out/r04/iso_$x/isolated_kernel_times.txt; python -c "
import json; d=json.load(open('gpurun_out/r04/iso_$x/iso_bench.json')); print(d['results'])"; done
grep "sw_score" gpurun_out/r04/iso_a/isolated_kernel_times.txt | head -24 | cut -c1-75,93-150
