#!/bin/bash
# where the stem's tile time goes: knock-outs of the measurement build (LWG_STEM_DBG, WRONG RESULTS), one lane, kernel time from bench.py's events
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06stemdbg
mkdir -p $O
cd $R
B="python bench.py --lanes 1 --steps 16 --warmup 4 --repeats 1 --precision bf16x3 --no-cpu-baseline --no-fp32-mode --no-secondary --no-strict"
for D in 0 1 2 3 4 8 7 15 0; do
  LWG_LIB=exp LWG_STEM_DBG=$D timeout 300 $B > $O/d$D.json 2> $O/d$D.err
  python - <<PY
import json
try:
    d = json.load(open("$O/d$D.json")); print("dbg $D stem", d["roofline"]["by_kernel"].get("stem_bf16x3_kernel"), "fps", d["value"])
except Exception as e:
    print("dbg $D failed", e, open("$O/d$D.err").read()[-600:])
PY
done
