#!/bin/bash
# upper bound of hiding the halo kernels' prologue (measurement build, WRONG RESULTS): LWG_EXP_PAIR=1 -> every halo workgroup runs two
# tiles, the second without prologue loads / waits
O=gpurun_out/r06pair; mkdir -p $O
B="python bench.py --steps 40 --warmup 8 --repeats 3 --no-cpu-baseline --no-fp32-mode --no-secondary --no-strict --precision bf16x3"
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export LWG_EXP_PAIR=1; else unset LWG_EXP_PAIR; fi
  LWG_LIB=exp timeout 300 $B > $O/pair_$v.json 2> $O/pair_$v.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/pair_$v.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("PAIR=$v: %.1f fps %.4f ms/step; all-conv pipe %.3f; " % (d["value"], d["ms_per_step"], r["all_conv_frac_pipe"]) + "; ".join("%s %d x %.1f us" % (k[:44], v[0], v[1]) for k, v in r["by_kernel"].items()))
except Exception as e:
    print("PAIR=$v failed", e, open("$O/pair_$v.err").read()[-600:])
PY
done
