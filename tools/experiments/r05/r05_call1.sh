#!/bin/bash
# round 5, GPU visit 1: the host-side changes (self-launching bench, frame graph, deterministic grid_sample gradient, gradient buckets,
# graph capture with collectives) + a full bench line.  One pytest process per file: a GPU fault aborts the interpreter.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
run() { # name, args...
  local name=$1; shift
  local t0=$(date +%s)
  timeout 900 python -m pytest -x -q -m gpu -p no:cacheprovider "$@" > $O/$name.log 2>&1
  echo "$name rc=$? wall=$(( $(date +%s) - t0 )) s: $(grep -E 'passed|failed|error' $O/$name.log | tail -1)"
}
run graph tests/test_gpu_train_graph.py
if grep -q "Fatal Python error" $O/graph.log; then
  echo "--- graph test aborted; A/B with the atomic scatter:"
  LWG_GS_ATOMIC=1 run graph_atomic tests/test_gpu_train_graph.py
  grep -n "Error\|assert\|Fatal" $O/graph_atomic.log | head -10
fi
grep -n "Error\|assert \|Fatal" $O/graph.log | head -10
run ops tests/test_gpu_ops.py
run rccl tests/test_gpu_rccl.py
grep -n "Error\|assert \|rccl:" $O/rccl.log | cut -c1-1500 | head -12
run imitator tests/test_gpu_imitator.py
grep -n "Error\|assert \|device records" $O/imitator.log | cut -c1-600 | head -12
run multirank tests/test_gpu_multirank.py
grep -n "Error\|assert " $O/multirank.log | cut -c1-600 | head -12
run gtrainer tests/test_gpu_generator_trainer.py
grep -n "Error\|assert " $O/gtrainer.log | cut -c1-400 | head -12
T1=$(date +%s)
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - T1 )) s"
tail -5 $O/bench.err
python - <<PY
import json
try:
    d = json.load(open("$O/bench.json"))
    for k in ("value", "ms_per_step", "ms_per_step_min", "ms_per_step_max", "ms_per_step_windows", "gpu_clocks", "host_enqueue_ms_per_step"):
        print(k, d.get(k))
    print("fp32", {k: d.get("exact_fp32_mode", {}).get(k) for k in ("value", "ms_per_step", "ms_per_step_min", "ms_per_step_max")})
    r = d.get("roofline", {})
    print("roofline", r.get("kernel"), r.get("achieved"), r.get("frac"), r.get("frac_pipe"), r.get("all_conv_kernels", {}).get("frac_pipe"))
    print("parity", json.dumps(d.get("parity"))[:400])
    print("cpu", d.get("cpu_baseline", {}).get("value"))
    s = d.get("secondary", {})
    print("latency", json.dumps(s.get("latency")))
    print("personalize", json.dumps(s.get("personalize")))
    print("swap", json.dumps(s.get("swap"))[:600])
    print("train", json.dumps(s.get("train"))[:1500])
except Exception as e:
    print("bench parse failed", e)
PY
