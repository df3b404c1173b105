#!/bin/bash
# round 3, GPU visit 1: new tests, K-order / ring A/B in situ, microbench, stress log.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
nproc > $O/host.txt; rocm-smi --showproductname 2>/dev/null | head -8 >> $O/host.txt
# ATT probe: is the thread-trace decoder in the image?
( cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --att --kernel-trace -d $O/att -o a -- $R/tools/_build/igemm_bench 1 > $O/att_probe.log 2>&1; echo "att rc=$?" >> $O/att_probe.log )
tail -3 $O/att_probe.log
timeout 1500 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_tasks.py tests/test_gpu_ops.py tests/test_gpu_generator.py tests/test_gpu_bench_config.py tests/test_gpu_imitator.py -m gpu -x -q > $O/pytest_new.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_new.log
tail -15 $O/pytest_new.log
for K in tap channel; do
  LWG_K_ORDER=$K timeout 300 tools/_build/igemm_bench 20 > $O/igemm_$K.log 2>&1
  echo "--- igemm $K"; cat $O/igemm_$K.log
done
B="--no-cpu-baseline --no-fp32-mode --no-secondary"
for rep in 1 2; do
  LWG_K_ORDER=channel timeout 300 python bench.py $B > $O/bench_channel_$rep.json 2> $O/bench_channel_$rep.err
  timeout 300 python bench.py $B > $O/bench_tap_$rep.json 2> $O/bench_tap_$rep.err
  LWG_RING=4 timeout 300 python bench.py $B > $O/bench_tap_ring4_$rep.json 2> $O/bench_tap_ring4_$rep.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], "fps", d["value"], "ms", d["ms_per_step"], r["kernel"][:28], r["achieved"], "pipe", r["frac_pipe"], "all", r["all_conv_kernels"]["frac_pipe"])
        for k, v in r["all_conv_kernels"]["by_kernel"].items():
            print("     ", k[:40], v)
    except Exception as e:
        print(f, "failed", e)
PY
timeout 900 python tools/lane_stress.py 500 2,3 8 1 > $O/lane_stress_overlap.log 2>&1; echo "stress rc=$?" | tee -a $O/lane_stress_overlap.log
tail -3 $O/lane_stress_overlap.log
