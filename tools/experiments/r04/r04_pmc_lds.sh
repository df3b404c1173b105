#!/bin/bash
# LDS and wait counters of the conv kernels in the bench pipeline (own PMC passes, one lane)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04y
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--lanes 1 --steps 2 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-roofline --no-fp32-mode --no-secondary"
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d $O/lds -o p -- python $R/bench.py $A > $O/lds.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/wait -o p -- python $R/bench.py $A > $O/wait.log 2>&1
cd $R
python - <<PY
import csv, glob, collections, re
def short(n):
    m = re.search(r"namespace\)::(\w+(<[^>]*>)?)", n); return m.group(1) if m else n[:60]
out = ["# LDS and wave-state counters of the conv kernels inside the bench pipeline (rocprofv3 --pmc, own passes, one lane, fused pairs)\n"]
for d, cols in (("lds", ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_UNALIGNED_STALL")), ("wait", ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY"))):
    f = glob.glob("$O/%s/**/p_counter_collection.csv" % d, recursive=True)
    if not f:
        out.append("(%s pass produced no counters)\n" % d); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        n = short(r["Kernel_Name"])
        if "conv" not in n and "stem" not in n and "heads" not in n and "apply" not in n: continue
        a = agg.setdefault((n, r["Grid_Size"]), collections.defaultdict(float))
        a[r["Counter_Name"]] += float(r["Counter_Value"]); a["_n"] += 1.0 / len(cols)
    out.append("| kernel | grid | launches | " + " | ".join(cols) + " | ratio |\n|---|---|---|" + "---|" * (len(cols) + 1))
    for (n, g), a in agg.items():
        if d == "lds":
            ratio = "conflict / active = %.4f" % (a[cols[0]] / a[cols[1]] if a[cols[1]] else 0)
        else:
            wc = a["SQ_WAVE_CYCLES"] or 1
            ratio = "parked %.2f, issue-stalled %.2f, issuing %.2f of wave-cycles" % (a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc, a["SQ_ACTIVE_INST_ANY"] / wc)
        out.append("| \`%s\` | %s | %d | " % (n, g, round(a["_n"])) + " | ".join("%.3g" % a[c] for c in cols) + " | " + ratio + " |")
    out.append("")
open("$O/r04_pmc_lds_wait.md", "w").write("\n".join(out) + "\n")
print("\n".join(out)[:6000])
PY
