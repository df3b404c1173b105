#!/bin/bash
# round 3, GPU visit 8: transposed convs on the halo kernel; round depth / overlap with fused pairs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03h
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_generator.py tests/test_gpu_bench_config.py tests/test_gpu_ops.py tests/test_gpu_sizes.py tests/test_gpu_generator_trainer.py tests/test_gpu_tasks.py -m gpu -x -q > $O/pytest_ct.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_ct.log
tail -12 $O/pytest_ct.log
B="--no-cpu-baseline --no-fp32-mode --no-secondary"
run() { # name env...
  n=$1; shift
  env "$@" timeout 300 python bench.py $B > $O/bench_$n.json 2> $O/bench_$n.err
}
for rep in 1 2; do
  run f2_$rep LWG_FUSE=2
  run f2_d8_$rep LWG_FUSE=2 LWG_ROUND_DEPTH=8
  run f2_d8_ov_$rep LWG_FUSE=2 LWG_ROUND_DEPTH=8 LWG_OVERLAP_GEOMETRY=1
  run f2_ov_$rep LWG_FUSE=2 LWG_OVERLAP_GEOMETRY=1
  run f1_$rep LWG_FUSE=1
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], "fps", d["value"], "ms", d["ms_per_step"], r["kernel"][:34], r["achieved"], "pipe", r["frac_pipe"], "all", r["all_conv_kernels"]["frac_pipe"])
        if "f2_1" in f or "f1_1" in f:
            for k, v in r["all_conv_kernels"]["by_kernel"].items():
                print("     ", k[:40], v)
    except Exception as e:
        print(f, "failed", e, open(f.replace(".json", ".err")).read()[-800:])
PY
