#!/bin/bash
# training iteration with the InstanceNorm reductions finished by the slab kernels' last arriver (default) against LWG_IN_FUSED=0
# (a separate finalising launch per reduction: round 5), one box, alternating
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "instance_norm" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_train_graph.py tests/test_gpu_generator_trainer.py -q -x 2>&1 | tail -2
for i in 1 2; do
  for F in 0 1; do
    echo "LWG_IN_FUSED=$F: $(LWG_IN_FUSED=$F timeout 600 python tools/bench_train.py --precision bf16x3 --batch 4 --image-size 256 --graph --steps 8 2>&1 | tail -1 | cut -c1-300)"
  done
done
