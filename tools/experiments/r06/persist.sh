#!/bin/bash
# persistent halo kernel: correctness (A/B hash test + generator suite) and the frames/s A/B
O=gpurun_out/r06persist; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_generator.py -x -q -k "persistent or matches or fused" 2>&1 | tail -8
B="python bench.py --steps 40 --warmup 8 --repeats 3 --no-cpu-baseline --no-fp32-mode --no-secondary --no-strict --precision bf16x3"
for v in 1 0 1 0; do
  LWG_HALO_PERSIST=$v timeout 300 $B > $O/p_$v.json 2> $O/p_$v.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/p_$v.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("PERSIST=$v: %.1f fps %.4f ms/step; all-conv pipe %.3f; " % (d["value"], d["ms_per_step"], r["all_conv_frac_pipe"]) + "; ".join("%s %d x %.1f us" % (k[:44], v[0], v[1]) for k, v in r["by_kernel"].items()))
except Exception as e:
    print("PERSIST=$v failed", e, open("$O/p_$v.err").read()[-600:])
PY
done
