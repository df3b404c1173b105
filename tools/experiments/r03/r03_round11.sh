#!/bin/bash
# round 3, GPU visit 11: 256-row ring tiles (eight waves) for the stride-2 encoders
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03k
mkdir -p $O
cd $R
LWG_RING_TALL=0 timeout 300 tools/_build/igemm_bench 20 > $O/igemm_short.log 2>&1; echo "--- 128-row ring tiles"; grep "enc" $O/igemm_short.log
timeout 300 tools/_build/igemm_bench 20 > $O/igemm_tall.log 2>&1; echo "--- 256-row ring tiles where eligible"; grep "enc" $O/igemm_tall.log
timeout 900 python -m pytest tests/test_gpu_generator.py tests/test_gpu_bench_config.py tests/test_gpu_sizes.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
B="--no-cpu-baseline --no-fp32-mode --no-secondary"
for rep in 1 2 3; do
  LWG_RING_TALL=0 timeout 300 python bench.py $B > $O/bench_short_$rep.json 2> $O/bench_short_$rep.err
  timeout 300 python bench.py $B > $O/bench_tall_$rep.json 2> $O/bench_tall_$rep.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], "fps", d["value"], "ms", d["ms_per_step"], "all", r["all_conv_kernels"]["frac_pipe"], {k[:22]: (v["avg_launch_ms"], v["frac_pipe"]) for k, v in r["all_conv_kernels"]["by_kernel"].items()})
    except Exception as e:
        print(f, "failed", e, open(f.replace(".json", ".err")).read()[-800:])
PY
