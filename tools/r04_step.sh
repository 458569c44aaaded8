mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_gpu_sw.py tests/test_gpu_pipeline.py tests/test_gpu_edge_cases.py tests/test_gpu_profile.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -8
