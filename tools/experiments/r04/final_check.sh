cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/final/gpu_tests.log 2>&1; tail -6 gpurun_out/final/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
