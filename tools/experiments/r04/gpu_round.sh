#!/bin/bash
# One GPU visit (through gpurun): parity tests, lane stress with the geometry overlapped, bench line, kernel stats.
# usage: tools/gpu_round.sh TAG [stress_passes]
set -u
TAG=${1:-q}
PASSES=${2:-300}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python tools/lane_stress.py $PASSES 2,3 8 0 > $O/stress.log 2>&1; echo "stress rc=$?" | tee -a $O/stress.log
tail -3 $O/stress.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-600 $O/bench.json
python - <<PY
import json
try:
    d = json.load(open("$O/bench.json"))
    print("fps", d["value"], "ms", d["ms_per_step"], "fp32", d.get("exact_fp32_mode", {}).get("value"))
    print("parity", json.dumps(d.get("parity")))
    r = d.get("roofline", {})
    print("roofline", r.get("kernel"), r.get("achieved"), r.get("frac_pipe"), r.get("all_conv_kernels", {}).get("frac_pipe"))
    for k, v in r.get("all_conv_kernels", {}).get("by_kernel", {}).items():
        print("  ", k, v)
    print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
except Exception as e:
    print("bench parse failed", e); print(open("$O/bench.err").read()[-2000:])
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- \
    python $R/bench.py --lanes 1 --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-mode > $O/stats.log 2>&1
cd $R
python tools/summarize_profile.py stats $(find $O/stats -name k_kernel_stats.csv | head -1) $O/${TAG}_kernel_stats.md $O/bench.json
sed -n 12,40p $O/${TAG}_kernel_stats.md
