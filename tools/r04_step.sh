mkdir -p gpurun_out/r04
SD_DEBUG_WS=1 python bench.py --no-cpu --no-p1000 --no-index-check --steps 12 --chunk 7500 > gpurun_out/r04/b_ws.json 2> gpurun_out/r04/b_ws.err
grep -c "^\[ws\]" gpurun_out/r04/b_ws.err
grep "^\[ws\]\|^\[bench\]" gpurun_out/r04/b_ws.err | grep "hipFree\|hipHostFree\|bench" | tail -70 | cut -c1-160
