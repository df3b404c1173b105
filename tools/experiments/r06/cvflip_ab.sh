#!/bin/bash
# RAW conversion writes balanced across the two bank halves (pixels 4..7 write lo first) against the previous build (LWG_LIB=base):
# bit-identity tests, LDS counters, kernel times and frames/s, one box, alternating
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06flip
mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_generator.py tests/test_gpu_bench_config.py -q -x 2>&1 | tail -2
B="python bench.py --steps 40 --warmup 8 --repeats 3 --no-cpu-baseline --no-fp32-mode --no-secondary --no-strict"
for i in 1 2; do
  for L in base ""; do
    LWG_LIB=$L LWG_ALLOW_STALE_LIB=1 timeout 300 $B > $O/b.json 2> $O/b.err
    python -c "
import json; d=json.load(open('$O/b.json')); r=d['roofline']; print('lib=[$L] fps', d['value'], 'all conv ms/step', r['all_conv_ms_per_step'], r['by_kernel'])"
  done
done
bash tools/experiments/r06/lds_pmc.sh 2>&1 | grep -a "0, true>\|1, true>\|^| kernel"
