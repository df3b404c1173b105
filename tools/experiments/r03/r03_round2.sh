#!/bin/bash
# round 3, GPU visit 2: remaining tests, co-residency reproducer matrix, in-kernel cycle accounting, ring depth A/B.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_multirank.py::test_bench_train_under_torchrun tests/test_gpu_tasks.py tests/test_gpu_ops.py tests/test_gpu_generator.py tests/test_gpu_bench_config.py tests/test_gpu_imitator.py tests/test_gpu_generator_trainer.py -m gpu -q > $O/pytest_new.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_new.log
tail -25 $O/pytest_new.log
# co-residency reproducer: victim shapes x neighbours x CU masks
( for v in 1 2 3 4 5; do for n in 0 1 2; do timeout 120 tools/_build/coresidency_repro 300 $v $n 0; done; done
  for v in 1 2; do timeout 120 tools/_build/coresidency_repro_real 300 $v 3 0; done
  for m in 1 2; do timeout 120 tools/_build/coresidency_repro 300 1 2 $m; timeout 120 tools/_build/coresidency_repro_real 300 1 3 $m; done ) > $O/coresidency.log 2>&1
cat $O/coresidency.log
timeout 600 python tools/conv_trace.py 3 $O/conv_trace.md > $O/conv_trace.log 2>&1; tail -30 $O/conv_trace.log
B="--no-cpu-baseline --no-fp32-mode --no-secondary"
for rep in 1 2; do for ring in 3 4 5; do
  LWG_RING=$ring timeout 300 python bench.py $B > $O/bench_ring${ring}_$rep.json 2> $O/bench_ring${ring}_$rep.err
done; done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], "fps", d["value"], "ms", d["ms_per_step"], r["kernel"][:34], r["achieved"], "pipe", r["frac_pipe"], "all", r["all_conv_kernels"]["frac_pipe"])
    except Exception as e:
        print(f, "failed", e, open(f.replace(".json", ".err")).read()[-600:])
PY
timeout 300 tools/_build/igemm_bench 20 > $O/igemm.log 2>&1; cat $O/igemm.log
