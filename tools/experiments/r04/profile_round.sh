#!/bin/bash
# Runs on the GPU box (through gpurun): bench line + rocprofv3 kernel stats + a PMC pass, summaries into gpurun_out/.
# usage: tools/profile_round.sh r01
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $R/gpurun_out/$TAG/bench.json 2> $R/gpurun_out/$TAG/bench.err
python $R/bench.py --lanes 3 --no-cpu-baseline --no-roofline --no-fp32-mode > $R/gpurun_out/$TAG/bench_lanes3.json 2>/dev/null
python $R/bench.py --lanes 1 --no-cpu-baseline --no-roofline --no-fp32-mode > $R/gpurun_out/$TAG/bench_lanes1.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats -o k -- \
    python $R/bench.py --lanes 1 --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-mode > $R/gpurun_out/$TAG/stats.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
    -d $R/gpurun_out/$TAG/pmc -o p -- python $R/bench.py --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-mode > $R/gpurun_out/$TAG/pmc.log 2>&1
# the exact-fp32 mode and the discriminator update, kernel stats only (QUICK=1 skips them: their code did not change)
if [ -z "${QUICK:-}" ]; then
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/fp32_stats -o k -- \
    python $R/bench.py --steps 10 --warmup 3 --settle-ms 0 --precision fp32 --no-cpu-baseline --no-roofline > $R/gpurun_out/$TAG/fp32_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/dstep_stats -o k -- \
    python $R/tools/bench_discriminator.py --steps 10 > $R/gpurun_out/$TAG/dstep_stats.log 2>&1
python $R/tools/bench_discriminator.py > $R/gpurun_out/$TAG/dstep_bench.json 2>/dev/null
fi
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/pmc_$C -o p -- \
      python $R/bench.py --lanes 1 --steps 2 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-roofline --no-fp32-mode > $R/gpurun_out/$TAG/pmc_$C.log 2>&1
done
cd $R
python tools/summarize_profile.py traffic $(find gpurun_out/$TAG/pmc_FETCH_SIZE -name p_counter_collection.csv) \
    $(find gpurun_out/$TAG/pmc_WRITE_SIZE -name p_counter_collection.csv) gpurun_out/$TAG/${TAG}_traffic.json gpurun_out/$TAG/${TAG}_traffic.md
if [ -z "${QUICK:-}" ]; then
python tools/summarize_profile.py stats $(find gpurun_out/$TAG/fp32_stats -name k_kernel_stats.csv) gpurun_out/$TAG/${TAG}_fp32_kernel_stats.md
python tools/summarize_profile.py stats $(find gpurun_out/$TAG/dstep_stats -name k_kernel_stats.csv) gpurun_out/$TAG/${TAG}_dstep_kernel_stats.md
fi
python tools/summarize_profile.py stats gpurun_out/$TAG/stats/k_kernel_stats.csv gpurun_out/$TAG/${TAG}_kernel_stats.md gpurun_out/$TAG/bench.json
python tools/summarize_profile.py pmc gpurun_out/$TAG/pmc/p_counter_collection.csv gpurun_out/$TAG/pmc/p_kernel_trace.csv gpurun_out/$TAG/${TAG}_pmc_mfma.md
cat gpurun_out/$TAG/bench.json | cut -c1-400; head -14 gpurun_out/$TAG/${TAG}_kernel_stats.md | tail -8; tail -8 gpurun_out/$TAG/${TAG}_pmc_mfma.md
