mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_gpu_profile.py -m gpu -x -q 2>&1 | tail -4
SD_ITER3_VERBOSE=1 SD_DEBUG_TIMING=1 python tools/iter3_scale.py 1000 2 8 > gpurun_out/r04/iter3t.txt 2> gpurun_out/r04/iter3t.err
grep -v "hot filter" gpurun_out/r04/iter3t.txt | grep "^\[r2p\|clustersearch --num\|parity\|time for" | cut -c1-300
