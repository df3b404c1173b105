#!/bin/bash
# LDS counters of the bench's kernels on one lane (own PMC pass): bank-conflict cycles, active cycles, unaligned stalls -- the stem's
# dword-aligned fragment reads (two ds_read2_b32 at a 12-byte lane pitch) among them
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06lds
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PM="python $R/bench.py --lanes 1 --steps 2 --warmup 1 --repeats 1 --settle-ms 0 --precision bf16x3 --no-cpu-baseline --no-roofline --no-fp32-mode --no-secondary --no-strict"
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d $O/pmc -o p -- $PM > $O/pmc.log 2>&1
cd $R
python - <<PY
import csv, glob, re
from collections import defaultdict
f = glob.glob("$O/pmc/**/p_counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("lwg::(anonymous namespace)::", ""))[:64]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE": n[k] += 1
print("| kernel | dispatches | SQ_LDS_BANK_CONFLICT | SQ_LDS_IDX_ACTIVE | SQ_LDS_UNALIGNED_STALL | conflict / active |")
print("|---|---|---|---|---|---|")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0)):
    a = v.get("SQ_LDS_IDX_ACTIVE", 0)
    if a <= 0: continue
    print("| \`%s\` | %d | %.3g | %.3g | %.3g | %.4f |" % (k, n[k], v.get("SQ_LDS_BANK_CONFLICT", 0), a, v.get("SQ_LDS_UNALIGNED_STALL", 0), v.get("SQ_LDS_BANK_CONFLICT", 0) / a))
PY
