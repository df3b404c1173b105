#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03j
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
LWG_FUSE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o k -- python $R/bench.py --lanes 1 --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-secondary --no-roofline > $O/kt.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
f = glob.glob("$O/kt/**/k_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    if "conv" not in name and "stem" not in name: continue
    key = (name[:70], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"))
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = v[len(v)//3:]   # drop the warm-up third
    print("%8.1f us x %4d  %s" % (sum(v) / len(v), len(v), k))
PY
