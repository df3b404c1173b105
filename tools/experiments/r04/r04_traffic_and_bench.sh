#!/bin/bash
# HBM traffic of the dominant kernel by the two PMC passes the guide prescribes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel
# trace only), summarised into profiles/r04_traffic.{json,md}, then the bench line that attaches it: bash tools/r04_traffic_and_bench.sh <tag>
set -u
TAG=${1:-r04tb}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PM="python $R/bench.py --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-mode --no-secondary --settle-ms 0"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o p -- $PM > $O/pmc_$C.log 2>&1
done
cd $R
python tools/summarize_profile.py traffic $(find $O/pmc_FETCH_SIZE -name p_counter_collection.csv) \
    $(find $O/pmc_WRITE_SIZE -name p_counter_collection.csv) $O/r04_traffic.json $O/r04_traffic.md
cp $O/r04_traffic.json $O/r04_traffic.md profiles/
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("fps", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "pipe", r["frac_pipe"], "traffic", r.get("traffic"), "parity", d["parity"]["ok"])
print(json.dumps(d["secondary"]["train"])[:700])
PY
