#!/bin/bash
# the 8 x 32 tile of the 128-channel 3x3 layers (trunk, skipper.0/1) on FOUR waves of 64 x 128 (LWG_HALO_W4=1) against eight of 64 x 64:
# parity tests under the switch, kernel times (events, one lane) and frames/s (two lanes), alternating
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06w4
mkdir -p $O; cd $R
LWG_HALO_W4=1 timeout 900 python -m pytest tests/test_gpu_bench_config.py -q -x 2>&1 | tail -2
B="python bench.py --steps 40 --warmup 8 --repeats 3 --no-cpu-baseline --no-fp32-mode --no-secondary --no-strict"
for i in 1 2; do
  for W in 0 1; do
    LWG_HALO_W4=$W timeout 300 $B > $O/b.json 2> $O/b.err
    python -c "
import json; d=json.load(open('$O/b.json')); r=d['roofline']; print('LWG_HALO_W4=$W fps', d['value'], 'all conv ms/step', r['all_conv_ms_per_step'], r['by_kernel'])"
  done
done
