#!/bin/bash
# round 3, GPU visit 3: co-residency bisect (neighbour variants, micro-victim), cycle accounting v2, DMA-count experiment
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
X=tools/_build/coresidency_repro_real
( echo "== which part of the neighbour (victim 1 = fused shape)"
  for n in 200 240 201 204 213 232 456 812 821 100; do timeout 120 $X 200 1 $n 0; done
  echo "== victim variants beside the real kernel (240 = the 4-slot kernel the product runs now)"
  for v in 1 2 3 4 5; do timeout 120 $X 200 $v 240 0; done
  echo "== what is wrong"
  REPRO_DUMP=1 timeout 120 $X 40 1 240 0
  echo "== micro-victim: 96-bit store, K wait states, VALU overwrite of its data registers"
  for v in 10 11 12 13 14; do timeout 120 $X 200 $v 240 0; timeout 120 $X 200 $v 0 0; done
  for v in 11 12; do timeout 120 $X 200 $v 2 0; timeout 120 $X 200 $v 240 1;  timeout 120 $X 200 $v 240 2; done
) > $O/coresidency_bisect.log 2>&1
cat $O/coresidency_bisect.log
timeout 600 python tools/conv_trace.py 3 $O/conv_trace.md > $O/conv_trace.log 2>&1; tail -32 $O/conv_trace.log
timeout 300 tools/_build/igemm_bench 20 > $O/igemm.log 2>&1; cat $O/igemm.log
B="--no-cpu-baseline --no-fp32-mode --no-secondary"
for ring in 4 x; do
  LWG_RING=$ring timeout 300 python bench.py $B > $O/bench_ring${ring}.json 2> $O/bench_ring${ring}.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], "fps", d["value"], "ms", d["ms_per_step"], r["kernel"][:34], r["achieved"], "pipe", r["frac_pipe"], "all", r["all_conv_kernels"]["frac_pipe"])
    except Exception as e:
        print(f, "failed", e, open(f.replace(".json", ".err")).read()[-600:])
PY
