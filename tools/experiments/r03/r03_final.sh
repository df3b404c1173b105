#!/bin/bash
# round 3: validation of the final build -- GPU suite, smoke, the complete bench line, the traffic counters and kernel statistics
# that back it, and a bit-for-bit stress of the two-lane pipeline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03final}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_all.log
tail -4 $O/pytest_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log; tail -2 $O/smoke.log
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o p -- \
      python $R/bench.py --lanes 1 --steps 2 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-roofline --no-fp32-mode --no-secondary > $O/pmc_$C.log 2>&1
done
python $R/tools/summarize_profile.py traffic $(find $O/pmc_FETCH_SIZE -name p_counter_collection.csv) \
    $(find $O/pmc_WRITE_SIZE -name p_counter_collection.csv) $O/${TAG}_traffic.json $O/${TAG}_traffic.md
cp $O/${TAG}_traffic.json $R/profiles/r03_traffic.json   # stamped with this build's source digest: the bench line below attaches it
T0=$(date +%s)
timeout 900 python $R/bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- \
    python $R/bench.py --lanes 1 --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-mode --no-secondary > $O/stats.log 2>&1
cd $R
python tools/summarize_profile.py stats $(find $O/stats -name k_kernel_stats.csv) $O/${TAG}_kernel_stats.md $O/bench.json
(echo "# tools/lane_stress.py 60 2 8 1 on the final build (final build; geometry UNDER the generators: overlap=1, round_depth 1)"; \
 timeout 400 python tools/lane_stress.py 60 2 8 1 2>&1 | tail -3) > $O/lane_stress.log
cat $O/lane_stress.log
python - <<PY
import json
d = json.load(open("$O/bench.json")); r = d.get("roofline") or {}
print("fps", d["value"], "ms", d["ms_per_step"], "fp32", d.get("exact_fp32_mode", {}).get("value"), r.get("kernel"), r.get("achieved"), r.get("frac_pipe"), (r.get("all_conv_kernels") or {}).get("frac_pipe"), "traffic", r.get("traffic"))
for k, v in (r.get("all_conv_kernels") or {}).get("by_kernel", {}).items():
    print("  ", k, v)
print("parity", json.dumps(d.get("parity")))
print("cpu", json.dumps(d.get("cpu_baseline"))[:300])
print("secondary", json.dumps(d.get("secondary"))[:1800])
PY
head -34 $O/${TAG}_kernel_stats.md | tail -24 | cut -c1-200; head -22 $O/${TAG}_traffic.md | tail -16 | cut -c1-200
