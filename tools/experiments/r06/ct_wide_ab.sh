#!/bin/bash
# the 128-channel transposed-conv launches (convT.0 / convT.1: four accumulator sets, one workgroup per CU) against the 64-channel tile
# (two workgroups per CU), one lane kernel times + two-lane frames/s, alternating; LWG_CT_WIDE=0 selects the 64-channel tile
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06ct
mkdir -p $O; cd $R
timeout 600 env LWG_CT_WIDE=0 python -m pytest tests/test_gpu_generator.py -q -x -k "inference or checkpoints or fused" 2>&1 | tail -2
B="python bench.py --steps 40 --warmup 8 --repeats 3 --no-cpu-baseline --no-fp32-mode --no-secondary --no-strict"
for i in 1 2; do
  for W in 1 0; do
    LWG_CT_WIDE=$W timeout 300 $B > $O/w$W.json 2> $O/w$W.err
    python -c "
import json; d=json.load(open('$O/w$W.json')); r=d['roofline']; print('LWG_CT_WIDE=$W fps', d['value'], 'all conv ms/step', r['all_conv_ms_per_step'], {k:v for k,v in r['by_kernel'].items()})"
  done
done
