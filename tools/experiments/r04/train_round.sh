#!/bin/bash
# GPU box: op/trainer parity tests, one training iteration in both conv precisions, kernel statistics of the bf16x3 one.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/train; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_generator_trainer.py tests/test_gpu_discriminator.py -x -q > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
python tools/bench_train.py --precision bf16x3 --script-loss --steps 5 > $OUT/train_script_loss.json 2>$OUT/err_script.log
python -c "import json;d=json.loads(open('$OUT/train_script_loss.json').read().strip().splitlines()[-1]);print('script-loss bf16x3',d['ms_per_iteration'],d['images_per_s'],d['losses'])"
python tools/bench_discriminator.py > $OUT/dstep.json 2>/dev/null; tail -c 300 $OUT/dstep.json; echo
for p in fp32 bf16x3; do
  python tools/bench_train.py --precision $p --steps 5 > $OUT/train_$p.json 2>$OUT/err_$p.log
  python -c "import json;d=json.loads(open('$OUT/train_$p.json').read().strip().splitlines()[-1]);print('$p',d['ms_per_iteration'],d['images_per_s'],d['losses'])"
done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o train -- python tools/bench_train.py --precision bf16x3 --steps 3 > $OUT/prof.log 2>&1
python tools/summarize_profile.py stats $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/train_kernel_stats.md 2>>$OUT/prof.log
head -30 $OUT/train_kernel_stats.md
