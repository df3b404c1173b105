#!/bin/bash
# round 3, GPU visit 5: halo kernel with 2-D tiles (bn 64 at two workgroups per CU), whole GPU suite, A/B, victim modes 6/7
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03e
mkdir -p $O
cd $R
timeout 300 tools/_build/igemm_bench 20 > $O/igemm.log 2>&1; cat $O/igemm.log
B="--no-cpu-baseline --no-fp32-mode --no-secondary"
for rep in 1 2; do
  LWG_HALO=0 timeout 300 python bench.py $B > $O/bench_ring_$rep.json 2> $O/bench_ring_$rep.err
  timeout 300 python bench.py $B > $O/bench_halo_$rep.json 2> $O/bench_halo_$rep.err
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
X=tools/_build/coresidency_repro_real
( for v in 1 6 7; do timeout 120 $X 200 $v 240 0; done ) > $O/coresidency_modes.log 2>&1
grep -v "^reference" $O/coresidency_modes.log
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_all.log
tail -15 $O/pytest_all.log
