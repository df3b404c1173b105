#!/bin/bash
# round 3, GPU visit 4: halo kernel (microbench, parity tests, in-situ A/B), micro-victim 2, co-residency tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03d
mkdir -p $O
cd $R
timeout 300 tools/_build/igemm_bench 20 > $O/igemm.log 2>&1; cat $O/igemm.log
timeout 1200 python -m pytest tests/test_gpu_generator.py tests/test_gpu_bench_config.py tests/test_gpu_imitator.py tests/test_gpu_tasks.py -m gpu -x -q > $O/pytest_halo.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_halo.log
tail -12 $O/pytest_halo.log
B="--no-cpu-baseline --no-fp32-mode --no-secondary"
for rep in 1 2; do
  LWG_HALO=0 timeout 300 python bench.py $B > $O/bench_ring_$rep.json 2> $O/bench_ring_$rep.err
  timeout 300 python bench.py $B > $O/bench_halo_$rep.json 2> $O/bench_halo_$rep.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], "fps", d["value"], "ms", d["ms_per_step"], r["kernel"][:34], r["achieved"], "pipe", r["frac_pipe"], "all", r["all_conv_kernels"]["frac_pipe"])
        for k, v in r["all_conv_kernels"]["by_kernel"].items():
            print("     ", k[:40], v)
    except Exception as e:
        print(f, "failed", e, open(f.replace(".json", ".err")).read()[-800:])
PY
X=tools/_build/coresidency_repro_real
( echo "== micro-victim 2: S back-to-back stores of W dwords, s_nop 1, VALU overwrite"
  for v in 41 42 43 51 52 53 61 62 63; do timeout 120 $X 200 $v 240 0; done
  for v in 43 53 63; do timeout 120 $X 200 $v 0 0; timeout 120 $X 200 $v 2 0; done ) > $O/coresidency_micro2.log 2>&1
grep -v "^reference" $O/coresidency_micro2.log
timeout 900 python -m pytest tests/test_gpu_coresidency.py -m gpu -q > $O/pytest_cores.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_cores.log
tail -15 $O/pytest_cores.log
