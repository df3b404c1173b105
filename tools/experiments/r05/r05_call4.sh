#!/bin/bash
# round 5, GPU visit 4: inpaintor on bf16x3, fuse = 4, training A/B (tiled layouts, planned grid_sample gradient), gradient-error diagnostics
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05d
mkdir -p $O
cd $R
run() { local name=$1; shift; local t0=$(date +%s)
  timeout 900 python -m pytest -x -q -m gpu -p no:cacheprovider -s "$@" > $O/$name.log 2>&1
  echo "$name rc=$? wall=$(( $(date +%s) - t0 )) s: $(grep -E 'passed|failed|error' $O/$name.log | tail -1)"; }
run inpaint tests/test_gpu_inpaintor.py
grep -n "attention:\|inpaintor vs\|Error\|assert " $O/inpaint.log | cut -c1-300 | head
run benchcfg tests/test_gpu_bench_config.py tests/test_gpu_imitator.py
grep -n "Error\|assert \|L-inf over" $O/benchcfg.log | cut -c1-300 | head -8
run gdiag tests/test_gpu_generator_trainer.py -k "where_the or bf16x3_conv"
grep -n "ReLU mask flips\|per-tensor\|Error\|assert " $O/gdiag.log | cut -c1-1200 | head -8
for cfg in "A=1" "LWG_LAYOUT_TILED=0" "LWG_GS_ATOMIC=1" "A=1"; do
  env $cfg python tools/bench_train.py --batch 4 --image-size 256 --precision bf16x3 --graph --steps 6 2>/dev/null | python -c "import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train 256 b4 $cfg', d['ms_per_iteration'], d['launch'])"
done
B="python bench.py --no-cpu-baseline --no-fp32-mode --no-secondary --no-roofline --repeats 5"
for cfg in "A=1" "LWG_FUSE=2" "LWG_FUSE=8" "LWG_FUSE=4 LWG_ROUND_DEPTH=8" "A=1"; do
  env $cfg $B > $O/b.json 2>/dev/null
  python -c "import json; d=json.load(open('$O/b.json')); print('$cfg', d['value'], d['ms_per_step'], d['ms_per_step_windows'], d['gpu_clocks'].get('gfx_clock_mhz'))"
done
python - <<'PY'
import sys, json, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
print("personalize", json.dumps(bench.secondary_personalize(dev)))
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pstats -o k -- python $R/tools/personalize_once.py > $O/pstats.log 2>&1
cd $R; python tools/summarize_profile.py stats $(find $O/pstats -name k_kernel_stats.csv) $O/r05_personalize_kernel_stats.md --cmd "rocprofv3 --kernel-trace --stats -- python tools/personalize_once.py" > /dev/null 2>&1
sed -n 8,24p $O/r05_personalize_kernel_stats.md | cut -c1-150
