#!/bin/bash
# A/B of the dense-K stem (K = 336) against the previous build kept as _C/liblwg_base.so (K = 448), one box, alternating.
#   gpurun --timeout 1500 -- 'bash tools/experiments/r06/stem_ab.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06stem
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_generator.py tests/test_gpu_bench_config.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
B="python bench.py --steps 40 --warmup 8 --repeats 3 --no-cpu-baseline --no-fp32-mode --no-secondary --no-strict"
for i in 1 2; do
  LWG_LIB=base LWG_ALLOW_STALE_LIB=1 timeout 300 $B > $O/base_$i.json 2> $O/base_$i.err
  timeout 300 $B > $O/new_$i.json 2> $O/new_$i.err
done
python - <<PY
import json
for n in ("base_1", "new_1", "base_2", "new_2"):
    try:
        d = json.load(open("$O/%s.json" % n)); r = d["roofline"]
        print(n, d["value"], d["ms_per_step"], "stem", r["by_kernel"].get("stem_bf16x3_kernel"), "parity", r.get("parity_linf"), r.get("parity_ok"))
    except Exception as e:
        print(n, "failed", e, open("$O/%s.err" % n).read()[-800:])
PY
