#!/bin/bash
# round 3: the bench line + the profiles that back it (rocprofv3 kernel stats, PMC passes, cycle accounting, config-4 stats)
set -u
TAG=${1:-r03p}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T0=$(date +%s)
timeout 900 python $R/bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s"
python $R/bench.py --lanes 1 --no-cpu-baseline --no-roofline --no-fp32-mode --no-secondary > $O/bench_lanes1.json 2>/dev/null
python $R/bench.py --lanes 3 --no-cpu-baseline --no-roofline --no-fp32-mode --no-secondary > $O/bench_lanes3.json 2>/dev/null
LWG_FUSE=1 python $R/bench.py --no-cpu-baseline --no-fp32-mode --no-secondary > $O/bench_fuse1.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- \
    python $R/bench.py --lanes 1 --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-mode --no-secondary > $O/stats.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
    -d $O/pmc -o p -- python $R/bench.py --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-mode --no-secondary > $O/pmc.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o p -- \
      python $R/bench.py --lanes 1 --steps 2 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-roofline --no-fp32-mode --no-secondary > $O/pmc_$C.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/swap_stats -o k -- python $R/tools/bench_swap.py 30 > $O/swap_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp32_stats -o k -- \
    python $R/bench.py --steps 10 --warmup 3 --settle-ms 0 --precision fp32 --no-cpu-baseline --no-roofline --no-secondary > $O/fp32_stats.log 2>&1
cd $R
timeout 600 python tools/conv_trace.py 2 $O/${TAG}_conv_trace.md > $O/conv_trace.log 2>&1
python tools/summarize_profile.py traffic $(find $O/pmc_FETCH_SIZE -name p_counter_collection.csv) \
    $(find $O/pmc_WRITE_SIZE -name p_counter_collection.csv) $O/${TAG}_traffic.json $O/${TAG}_traffic.md
python tools/summarize_profile.py stats $(find $O/stats -name k_kernel_stats.csv) $O/${TAG}_kernel_stats.md $O/bench.json
python tools/summarize_profile.py stats $(find $O/swap_stats -name k_kernel_stats.csv) $O/${TAG}_swap_kernel_stats.md
python tools/summarize_profile.py stats $(find $O/fp32_stats -name k_kernel_stats.csv) $O/${TAG}_fp32_kernel_stats.md
python tools/summarize_profile.py pmc $(find $O/pmc -name p_counter_collection.csv) $(find $O/pmc -name p_kernel_trace.csv) $O/${TAG}_pmc_mfma.md
python - <<PY
import json
for n in ("bench", "bench_lanes1", "bench_lanes3", "bench_fuse1"):
    try:
        d = json.load(open("$O/%s.json" % n)); r = d.get("roofline") or {}
        print(n, "fps", d["value"], "ms", d["ms_per_step"], "fp32", d.get("exact_fp32_mode", {}).get("value"), r.get("kernel"), r.get("achieved"), r.get("frac_pipe"), (r.get("all_conv_kernels") or {}).get("frac_pipe"))
    except Exception as e:
        print(n, "failed", e)
d = json.load(open("$O/bench.json"))
print("parity", json.dumps(d.get("parity")))
print("cpu", json.dumps(d.get("cpu_baseline"))[:300])
print("secondary", json.dumps(d.get("secondary"))[:2500])
PY
head -30 $O/${TAG}_kernel_stats.md | tail -22; tail -12 $O/${TAG}_pmc_mfma.md; head -16 $O/${TAG}_traffic.md | tail -10; tail -14 $O/conv_trace.log | cut -c1-250
