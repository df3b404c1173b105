# one rocprofv3 kernel trace of the bench steps on one lane + the per-layer table (tools/roofline_from_profiles.py): bash tools/trace_only.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
BF="python $R/bench.py --lanes 1 --steps 16 --warmup 4 --settle-ms 0 --no-cpu-baseline --no-fp32-mode --no-secondary"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- $BF > $O/stats.log 2>&1
cd $R; python tools/roofline_from_profiles.py $(find $O/stats -name k_kernel_trace.csv) --out $O/roofline.md | sed -n 8,50p | cut -c1-200
