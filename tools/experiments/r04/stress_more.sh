cd $GRAFT_REPO_ROOT
for b in 1 4 16; do echo "batch $b:"; timeout 600 python tools/lane_stress.py 150 2 $b 1 2>&1 | tail -1; done
bash tools/overlap_repro.sh 2>&1 | tail -9
