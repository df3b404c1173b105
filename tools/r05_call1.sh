#!/bin/bash
# round 5, GPU visit 1: the host-side changes (self-launching bench, frame graph, deterministic grid_sample gradient, gradient buckets,
# graph capture with collectives) + a full bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
T0=$(date +%s)
timeout 1200 python -m pytest -x -q -m gpu tests/test_gpu_ops.py tests/test_gpu_train_graph.py tests/test_gpu_rccl.py \
    "tests/test_gpu_imitator.py" tests/test_gpu_multirank.py tests/test_gpu_generator_trainer.py -p no:cacheprovider \
    > $O/pytest.log 2>&1
echo "pytest rc=$? wall=$(( $(date +%s) - T0 )) s"; tail -25 $O/pytest.log
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
