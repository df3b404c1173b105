#!/bin/bash
# round 4: the bench line + the profiles that back it.  Every summary records the command line that was traced; the roofline
# tables are derived from the kernel traces alone (tools/roofline_from_profiles.py).
set -u
TAG=${1:-r04p}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T0=$(date +%s)
timeout 900 python $R/bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s"
python $R/bench.py --lanes 1 --no-cpu-baseline --no-roofline --no-fp32-mode --no-secondary > $O/bench_lanes1.json 2>/dev/null

BF="python $R/bench.py --lanes 1 --steps 16 --warmup 4 --settle-ms 0 --no-cpu-baseline --no-fp32-mode --no-secondary"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- $BF > $O/stats.log 2>&1
FP="python $R/bench.py --lanes 1 --steps 8 --warmup 2 --settle-ms 0 --precision fp32 --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp32_stats -o k -- $FP > $O/fp32_stats.log 2>&1
PM="python $R/bench.py --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-mode --no-secondary"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
    -d $O/pmc -o p -- $PM > $O/pmc.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o p -- $PM --settle-ms 0 > $O/pmc_$C.log 2>&1
done
PE="python $R/tools/personalize_once.py"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/personalize_stats -o k -- $PE > $O/personalize_stats.log 2>&1
cd $R
S="rocprofv3 --kernel-trace --stats --"
python tools/summarize_profile.py stats $(find $O/stats -name k_kernel_stats.csv) $O/${TAG}_kernel_stats.md $O/bench.json --cmd "$S ${BF//$R\//}"
python tools/summarize_profile.py stats $(find $O/fp32_stats -name k_kernel_stats.csv) $O/${TAG}_fp32_kernel_stats.md --cmd "$S ${FP//$R\//}"
python tools/summarize_profile.py stats $(find $O/personalize_stats -name k_kernel_stats.csv) $O/${TAG}_personalize_kernel_stats.md --cmd "$S ${PE//$R\//}"
python tools/roofline_from_profiles.py $(find $O/stats -name k_kernel_trace.csv) --cmd "$S ${BF//$R\//}" --out $O/${TAG}_roofline.md > /dev/null
python tools/roofline_from_profiles.py $(find $O/fp32_stats -name k_kernel_trace.csv) --cmd "$S ${FP//$R\//}" --out $O/${TAG}_fp32_roofline.md > /dev/null
python tools/summarize_profile.py traffic $(find $O/pmc_FETCH_SIZE -name p_counter_collection.csv) \
    $(find $O/pmc_WRITE_SIZE -name p_counter_collection.csv) $O/${TAG}_traffic.json $O/${TAG}_traffic.md
python tools/summarize_profile.py pmc $(find $O/pmc -name p_counter_collection.csv) $(find $O/pmc -name p_kernel_trace.csv) $O/${TAG}_pmc_mfma.md
python - <<PY
import json
for n in ("bench", "bench_lanes1"):
    try:
        d = json.load(open("$O/%s.json" % n)); r = d.get("roofline") or {}
        print(n, "fps", d["value"], "ms", d["ms_per_step"], "fp32", d.get("exact_fp32_mode", {}).get("value"), r.get("kernel"), r.get("achieved"), r.get("frac_pipe"), (r.get("all_conv_kernels") or {}).get("frac_pipe"))
    except Exception as e:
        print(n, "failed", e)
d = json.load(open("$O/bench.json"))
print("parity", json.dumps(d.get("parity"))[:900])
print("cpu", json.dumps(d.get("cpu_baseline"))[:300])
print("secondary", json.dumps(d.get("secondary"))[:1500])
PY
sed -n 1,40p $O/${TAG}_roofline.md | cut -c1-220; grep -A12 "By kernel" $O/${TAG}_fp32_roofline.md | cut -c1-200; tail -8 $O/${TAG}_pmc_mfma.md; echo "ATen kernels in the personalize trace: $(grep -c "at::native" $O/${TAG}_personalize_kernel_stats.md)"
