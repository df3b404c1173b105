# rocprofv3 kernel statistics of the training bench (3 timed iterations + 1 warm-up): bash tools/train_profile.sh <tag> [extra bench_train args]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/bench_train.py --precision bf16x3 --steps 3 $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- $CMD > $O/stats.log 2>&1
cd $R; python tools/summarize_profile.py stats $(find $O/stats -name k_kernel_stats.csv) $O/kernel_stats.md --cmd "rocprofv3 --kernel-trace --stats -- $CMD"; sed -n 7,40p $O/kernel_stats.md | cut -c1-160
