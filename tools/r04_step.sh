mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_profile.py -x -q -k result2profile 2>&1 | tail -3
SD_ITER3_VERBOSE=1 python tools/iter3_scale.py 1000 2 8 > gpurun_out/r04/iter3v.txt 2> gpurun_out/r04/iter3v.err
grep -E "result2profile:|prefilter|align" gpurun_out/r04/iter3v.txt | head -30
tail -1 gpurun_out/r04/iter3v.txt | cut -c1-600
