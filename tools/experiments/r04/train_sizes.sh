cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/train512
for cfg in "512 1" "512 2" "512 4" "256 4" "256 8"; do set -- $cfg
 for p in fp32 bf16x3; do
  python tools/bench_train.py --image-size $1 --batch $2 --precision $p --steps 4 2>/dev/null | tail -1 > gpurun_out/train512/t_$1_$2_$p.json
  python -c "import json;d=json.loads(open('gpurun_out/train512/t_$1_$2_$p.json').read());print($1,$2,'$p',d['ms_per_iteration'],d['images_per_s'])"
 done
done
