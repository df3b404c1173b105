#!/bin/bash
# round 3, GPU visit 6: 8-wave 8x32 halo tiles (skippers at batch 8, trunk at batch 16)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03f
mkdir -p $O
cd $R
LWG_HALO_TALL=0 timeout 300 tools/_build/igemm_bench 20 > $O/igemm_short.log 2>&1; echo "--- 4x32 tiles"; cat $O/igemm_short.log
timeout 300 tools/_build/igemm_bench 20 > $O/igemm_tall.log 2>&1; echo "--- 8x32 tiles where eligible"; cat $O/igemm_tall.log
timeout 1200 python -m pytest tests/test_gpu_generator.py tests/test_gpu_bench_config.py tests/test_gpu_imitator.py tests/test_gpu_sizes.py -m gpu -x -q > $O/pytest_halo.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_halo.log
tail -12 $O/pytest_halo.log
B="--no-cpu-baseline --no-fp32-mode --no-secondary"
for rep in 1 2; do
  LWG_HALO_TALL=0 timeout 300 python bench.py $B > $O/bench_short_$rep.json 2> $O/bench_short_$rep.err
  timeout 300 python bench.py $B > $O/bench_tall_$rep.json 2> $O/bench_tall_$rep.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], "fps", d["value"], "ms", d["ms_per_step"], r["kernel"][:34], r["achieved"], "pipe", r["frac_pipe"], "all", r["all_conv_kernels"]["frac_pipe"])
        for k, v in r["all_conv_kernels"]["by_kernel"].items():
            print("     ", k[:40], v)
    except Exception as e:
        print(f, "failed", e, open(f.replace(".json", ".err")).read()[-800:])
PY
