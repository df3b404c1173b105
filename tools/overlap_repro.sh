cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_imitator.py tests/test_gpu_bench_config.py -x -q 2>&1 | tail -3
timeout 600 python tools/depth_bench.py 5 96 2>&1 | tail -8
