#!/bin/bash
# round 3, GPU visit 7: fused pairs of batches (trunk on 8x32 tiles), whole GPU suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03g
mkdir -p $O
cd $R
B="--no-cpu-baseline --no-fp32-mode --no-secondary"
for rep in 1 2 3; do
  timeout 300 python bench.py $B > $O/bench_fuse1_$rep.json 2> $O/bench_fuse1_$rep.err
  LWG_FUSE=2 timeout 300 python bench.py $B > $O/bench_fuse2_$rep.json 2> $O/bench_fuse2_$rep.err
done
LWG_FUSE=2 timeout 300 python bench.py $B --lanes 1 > $O/bench_fuse2_lanes1.json 2> $O/bench_fuse2_lanes1.err
LWG_FUSE=2 timeout 300 python bench.py $B --lanes 3 > $O/bench_fuse2_lanes3.json 2> $O/bench_fuse2_lanes3.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], "fps", d["value"], "ms", d["ms_per_step"], r["kernel"][:34], r["achieved"], "pipe", r["frac_pipe"], "all", r["all_conv_kernels"]["frac_pipe"])
    except Exception as e:
        print(f, "failed", e, open(f.replace(".json", ".err")).read()[-800:])
PY
LWG_FUSE=2 timeout 900 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_imitator.py -m gpu -x -q > $O/pytest_fuse.log 2>&1; echo "pytest(fuse) rc=$?" | tee -a $O/pytest_fuse.log
tail -5 $O/pytest_fuse.log
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_all.log
tail -8 $O/pytest_all.log
