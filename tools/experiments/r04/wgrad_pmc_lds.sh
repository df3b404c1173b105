#!/bin/bash
# LDS bank-conflict counters of the weight-gradient kernels (tools/wgrad_bench.py under one rocprofv3 --pmc pass): bash tools/wgrad_pmc_lds.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04wl}; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/lds -o p -- python $R/tools/wgrad_bench.py > $O/lds.log 2>&1
cd $R
python - $O <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/lds/**/p_counter_collection.csv", recursive=True)
agg = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"].split("::")[-1][:60]
    if "wgrad" not in n and "reduce" not in n: continue
    a = agg.setdefault((n, r["Grid_Size"]), collections.defaultdict(float))
    a[r["Counter_Name"]] += float(r["Counter_Value"]); a["_n"] += 1.0 / 3
lines = ["# rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES -- python tools/wgrad_bench.py  (batch 4, 256 x 256 shapes)\n",
         "| kernel | grid | launches | bank-conflict cycles | LDS-active cycles | conflict / active | LDS-active / CU-busy |", "|---|---|---|---|---|---|---|"]
for (n, g), a in agg.items():
    act, bc, busy = a["SQ_LDS_IDX_ACTIVE"], a["SQ_LDS_BANK_CONFLICT"], a["SQ_BUSY_CU_CYCLES"]
    lines.append("| `%s` | %s | %d | %.3g | %.3g | %.3f | %.3f |" % (n, g, round(a["_n"]), bc, act, bc / act if act else 0, act / busy if busy else 0))
open(sys.argv[1] + "/wgrad_pmc_lds.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
