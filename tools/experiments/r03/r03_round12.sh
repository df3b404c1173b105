#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03l
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_generator_trainer.py tests/test_gpu_multirank.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for cfg in "4 256" "1 512" "4 512"; do set -- $cfg
  timeout 300 python tools/bench_train.py --batch $1 --image-size $2 --steps 5 --precision bf16x3 > $O/train_$1_$2.json 2> $O/train_$1_$2.err; python -c "
import json; d=json.load(open('$O/train_$1_$2.json')); print('train', d['batch'], d['image_size'], d['ms_per_iteration'], 'ms', d['images_per_s'], 'img/s')"
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -o k -- python $R/tools/bench_train.py --batch 4 --image-size 256 --steps 3 --precision bf16x3 > $O/train_stats.log 2>&1
cd $R
python tools/summarize_profile.py stats $(find $O/train_stats -name k_kernel_stats.csv) $O/r03_train_kernel_stats.md
sed -n 8,34p $O/r03_train_kernel_stats.md
