#!/bin/bash
# do the two lanes' kernels actually overlap on the device?  rocprofv3 kernel trace of the default (two-lane) bench, then the share of
# in_finalize / apply8 / heads time during which a kernel of the OTHER queue is in flight (tools/experiments/r06/overlap_trace.py)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06ovl
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 16 --warmup 4 --repeats 1 --settle-ms 0 --precision bf16x3 --no-cpu-baseline --no-fp32-mode --no-secondary --no-strict --no-roofline"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o k -- $B > $O/trace.log 2>&1
cd $R
python tools/experiments/r06/overlap_trace.py $(find $O/trace -name k_kernel_trace.csv) | tee $O/overlap.md
