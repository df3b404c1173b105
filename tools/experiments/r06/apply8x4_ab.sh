#!/bin/bash
# apply8x4_kernel (four pixels per thread, default) against apply8_kernel (LWG_APPLY8=1): bit-identity test, one-lane kernel times
# (rocprofv3), two-lane frames/s, alternating
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06a84
mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_generator.py -q -x -k "apply8 or fused or inference" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
BF="python $R/bench.py --lanes 1 --steps 16 --warmup 4 --repeats 1 --settle-ms 0 --precision bf16x3 --no-cpu-baseline --no-fp32-mode --no-secondary --no-strict --no-roofline"
for V in 4 1; do
  LWG_APPLY8=$V timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v$V -o k -- $BF > $O/v$V.log 2>&1
  grep -h "apply8" $(find $O/v$V -name k_kernel_stats.csv) | cut -d, -f1-4
done
cd $R
B="python bench.py --steps 40 --warmup 8 --repeats 3 --no-cpu-baseline --no-fp32-mode --no-secondary --no-strict --no-roofline"
for i in 1 2; do
  for V in 1 4; do
    LWG_APPLY8=$V timeout 300 $B > $O/b$V.json 2> $O/b$V.err
    python -c "
import json; d=json.load(open('$O/b$V.json')); print('LWG_APPLY8=$V fps', d['value'], d['ms_per_step'])"
  done
done
