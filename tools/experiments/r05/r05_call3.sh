#!/bin/bash
# round 5, GPU visit 3: MFMA attention, apply8, tiled weight layouts, AVG buckets; A/B timings
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
run() { local name=$1; shift; local t0=$(date +%s)
  timeout 900 python -m pytest -x -q -m gpu -p no:cacheprovider -s "$@" > $O/$name.log 2>&1
  echo "$name rc=$? wall=$(( $(date +%s) - t0 )) s: $(grep -E 'passed|failed|error' $O/$name.log | tail -1)"; }
run inpaint tests/test_gpu_inpaintor.py
grep -n "attention:\|Error\|assert " $O/inpaint.log | cut -c1-300 | head
run apply8 tests/test_gpu_generator.py -k "apply8 or bit_identical"
grep -n "Error\|assert " $O/apply8.log | cut -c1-300 | head -5
run ops tests/test_gpu_ops.py
grep -n "Error\|assert " $O/ops.log | cut -c1-300 | head -5
run gtrainer tests/test_gpu_generator_trainer.py tests/test_gpu_train_graph.py
grep -n "Error\|assert " $O/gtrainer.log | cut -c1-300 | head -5
run rccl tests/test_gpu_rccl.py
grep -n "rccl:\|Error\|assert " $O/rccl.log | cut -c1-1800 | head -6
B="python bench.py --no-cpu-baseline --no-fp32-mode --no-secondary --no-roofline --repeats 5"
for cfg in "base" "LWG_APPLY8=0" "LWG_FUSE=4" "base"; do
  if [ "$cfg" = base ]; then $B > $O/b.json 2>/dev/null; else env $cfg $B > $O/b.json 2>/dev/null; fi
  python -c "import json; d=json.load(open('$O/b.json')); print('$cfg', d['value'], d['ms_per_step'], d['ms_per_step_windows'], d['gpu_clocks'].get('gfx_clock_mhz'))"
done
python - <<'PY'
import sys, time, json, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import bench, bench_train
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
print("personalize", json.dumps(bench.secondary_personalize(dev)))
for n, s in ((4, 256), (4, 512)):
    r = bench_train.measure(n, s, steps=4, warmup=2, precision="bf16x3", graph=True)
    print("train", n, s, r["ms_per_iteration"], r["launch"])
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pstats -o k -- python $R/tools/personalize_once.py > $O/pstats.log 2>&1
cd $R; python tools/summarize_profile.py stats $(find $O/pstats -name k_kernel_stats.csv) $O/r05_personalize_kernel_stats.md --cmd "rocprofv3 --kernel-trace --stats -- python tools/personalize_once.py" > /dev/null 2>&1
sed -n 8,30p $O/r05_personalize_kernel_stats.md | cut -c1-150
