#!/bin/bash
# GPU box: the checks of DESIGN.md section 5.1 in one go (about a minute):
#   the rasteriser on fixed inputs beside bf16x3 convolutions / generators / library GEMMs, and the pipeline with every
#   round's geometry under the previous round's generators.  All counts must be 0.
cd ${GRAFT_REPO_ROOT:-.}
REPRO_NOZERO=1 REPRO_MODES=conv128,bf16x3,gemm_bf16 timeout 600 python tools/overlap_repro.py 150 2>&1 | grep -A1 "neighbours" | cut -c1-300
timeout 300 python tools/overlap_detail.py 2>&1 | grep "wrong launches"
timeout 300 python tools/overlap_detail2.py 16 2>&1 | tail -1
timeout 900 python tools/lane_stress.py 100 2,3 8 1 2>&1 | tail -1
