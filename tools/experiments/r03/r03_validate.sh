#!/bin/bash
# round 3: full validation of the current build -- GPU suite, smoke, the complete bench line (CPU baseline, parity, secondary)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03v}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 2700 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_all.log
tail -6 $O/pytest_all.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log; tail -2 $O/smoke.log
/usr/bin/time -v timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
grep -E "Elapsed|Maximum resident" $O/bench.err
python - <<PY
import json
try:
    d = json.load(open("$O/bench.json"))
    print("fps", d["value"], "ms", d["ms_per_step"], "fp32", d.get("exact_fp32_mode", {}).get("value"))
    print("parity", json.dumps(d.get("parity")))
    r = d.get("roofline", {})
    print("roofline", r.get("kernel"), r.get("achieved"), r.get("frac_pipe"), r.get("all_conv_kernels", {}).get("frac_pipe"), "traffic", r.get("traffic"), r.get("traffic_note"))
    for k, v in r.get("all_conv_kernels", {}).get("by_kernel", {}).items():
        print("  ", k, v)
    print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
    print("secondary", json.dumps(d.get("secondary"))[:1500])
except Exception as e:
    print("bench parse failed", e); print(open("$O/bench.err").read()[-3000:])
PY
