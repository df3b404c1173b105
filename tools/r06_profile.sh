#!/bin/bash
# ONE command that reproduces every number of the round-6 bench line and the profiles/ that back it (through gpurun on an MI355X):
#
#     gpurun --timeout 2400 -- 'bash tools/r06_profile.sh [TAG]'        # outputs under gpurun_out/TAG/, summaries named r06_*.md
#
#   1. bench.py (the driver's default command)                                -> bench.json  (fps, roofline, cpu_baseline, parity, secondary)
#   2. rocprofv3 --kernel-trace --stats of the bench on ONE lane, bf16x3      -> r06_kernel_stats.md, r06_roofline.md (per layer / per kernel /
#      and exact fp32                                                            apply launches by position; tools/roofline_from_profiles.py)
#   3. rocprofv3 --pmc MFMA-busy / clocks (own pass), FETCH_SIZE, WRITE_SIZE  -> r06_pmc_mfma.md, r06_traffic.{md,json}
#      (own passes: they do not share a run with each other or with a trace domain other than the kernel trace)
#   4. rocprofv3 --kernel-trace --stats of one Imitator.personalize           -> r06_personalize_kernel_stats.md
#   5. rocprofv3 --kernel-trace --stats of the training iteration             -> r06_train_kernel_stats.md
# Every summary records the command line that was traced.  Copy the r06_* files into profiles/ to track them.
set -u
TAG=${1:-r06p}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T0=$(date +%s)
timeout 900 python $R/bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s"

BF="python $R/bench.py --lanes 1 --steps 16 --warmup 4 --repeats 1 --settle-ms 0 --precision bf16x3 --no-cpu-baseline --no-fp32-mode --no-secondary --no-strict"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- $BF > $O/stats.log 2>&1
FP="python $R/bench.py --lanes 1 --steps 8 --warmup 2 --repeats 1 --settle-ms 0 --precision fp32 --no-cpu-baseline --no-secondary --no-strict"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp32_stats -o k -- $FP > $O/fp32_stats.log 2>&1
PM="python $R/bench.py --lanes 1 --steps 2 --warmup 1 --repeats 1 --precision bf16x3 --no-cpu-baseline --no-roofline --no-fp32-mode --no-secondary --no-strict"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
    -d $O/pmc -o p -- $PM > $O/pmc.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o p -- $PM --settle-ms 0 > $O/pmc_$C.log 2>&1
done
PE="python $R/tools/personalize_once.py"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/personalize_stats -o k -- $PE > $O/personalize_stats.log 2>&1
TR="python $R/tools/bench_train.py --precision bf16x3 --batch 4 --image-size 256 --steps 3"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -o k -- $TR > $O/train_stats.log 2>&1
cd $R
S="rocprofv3 --kernel-trace --stats --"
python tools/summarize_profile.py stats $(find $O/stats -name k_kernel_stats.csv) $O/r06_kernel_stats.md $O/bench.json --cmd "$S ${BF//$R\//}"
python tools/summarize_profile.py stats $(find $O/fp32_stats -name k_kernel_stats.csv) $O/r06_fp32_kernel_stats.md --cmd "$S ${FP//$R\//}"
python tools/summarize_profile.py stats $(find $O/personalize_stats -name k_kernel_stats.csv) $O/r06_personalize_kernel_stats.md --cmd "$S ${PE//$R\//}"
python tools/summarize_profile.py stats $(find $O/train_stats -name k_kernel_stats.csv) $O/r06_train_kernel_stats.md --cmd "$S ${TR//$R\//}"
python tools/roofline_from_profiles.py $(find $O/stats -name k_kernel_trace.csv) --cmd "$S ${BF//$R\//}" --out $O/r06_roofline.md > /dev/null
python tools/roofline_from_profiles.py $(find $O/fp32_stats -name k_kernel_trace.csv) --cmd "$S ${FP//$R\//}" --out $O/r06_fp32_roofline.md > /dev/null
python tools/summarize_profile.py traffic $(find $O/pmc_FETCH_SIZE -name p_counter_collection.csv) \
    $(find $O/pmc_WRITE_SIZE -name p_counter_collection.csv) $O/r06_traffic.json $O/r06_traffic.md
python tools/summarize_profile.py pmc $(find $O/pmc -name p_counter_collection.csv) $(find $O/pmc -name p_kernel_trace.csv) $O/r06_pmc_mfma.md
python - <<PY
import json
try:
    d = json.load(open("$O/bench.json")); r = d.get("roofline") or {}
    print("fps", d["value"], "ms", d["ms_per_step"], d.get("ms_per_step_windows"), "fp32", r.get("exact_fp32_fps"), r.get("kernel"), r.get("achieved"),
          "frac", r.get("frac"), "pipe", r.get("frac_pipe"), "all-conv pipe", r.get("all_conv_frac_pipe"), "traffic", r.get("traffic"))
    print("config", json.dumps(d.get("config")))
    print("clocks", d.get("gpu_clocks"))
    print("parity", json.dumps(d.get("parity")))
    print("cpu", json.dumps(d.get("cpu_baseline")))
    for k, v in (d.get("secondary") or {}).items():
        print(k, json.dumps(v))
    print("line bytes", len(open("$O/bench.json").read()))
except Exception as e:
    print("bench parse failed", e); print(open("$O/bench.err").read()[-1500:])
PY
sed -n 1,40p $O/r06_roofline.md | cut -c1-200; sed -n '/HBM-side/,$p' $O/r06_roofline.md | cut -c1-200 | head -30
grep -A12 "By kernel" $O/r06_fp32_roofline.md | cut -c1-200; tail -8 $O/r06_pmc_mfma.md
sed -n 8,30p $O/r06_train_kernel_stats.md | cut -c1-150
echo "ATen kernels in the personalize trace: $(grep -c "at::native" $O/r06_personalize_kernel_stats.md)"
