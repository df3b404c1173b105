#!/bin/bash
# training iteration: prepared weight matrices (one launch per iteration) vs a re-layout inside every conv call
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_generator_trainer.py tests/test_gpu_train_graph.py -x -q 2>&1 | tail -30
for v in 1 0 1 0; do
  for cfg in "4 256" "4 512"; do set -- $cfg
    LWG_PREPARED_WEIGHTS=$v timeout 300 python tools/bench_train.py --batch $1 --image-size $2 --precision bf16x3 --graph --steps 8 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PREPARED=$v batch $1 size $2:', d['ms_per_iteration'], 'ms', d['launch'])"
  done
done
