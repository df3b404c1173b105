#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03i
mkdir -p $O
cd $R
LWG_FUSE=1 timeout 600 python tools/conv_trace.py 2 $O/conv_trace_halo.md > $O/conv_trace.log 2>&1; tail -40 $O/conv_trace.log | cut -c1-260
