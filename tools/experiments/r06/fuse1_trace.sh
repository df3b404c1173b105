#!/bin/bash
# per-frame kernel times at 8 frames per launch (LWG_FUSE=1) against 32 (default), one lane: do the 256 x 256-resolution layers gain
# from a working set (134 MB per tensor at 8 frames) that fits the 256 MB Infinity Cache?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06fuse1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --lanes 1 --steps 16 --warmup 4 --repeats 1 --settle-ms 0 --precision bf16x3 --no-cpu-baseline --no-fp32-mode --no-secondary --no-strict --no-roofline"
for F in 1 4; do
  LWG_FUSE=$F timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/f$F -o k -- $B > $O/f$F.log 2>&1
done
cd $R
python - <<PY
import csv, glob, re
from collections import defaultdict
def load(f):
    rows = list(csv.DictReader(open(glob.glob("$O/f%d/**/k_kernel_trace.csv" % f, recursive=True)[0])))
    acc = defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"].replace("void ", "").replace("lwg::(anonymous namespace)::", "")
        n = re.sub(r"\(.*", "", n)
        acc[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return acc
a1, a4 = load(1), load(4)
print("| kernel | 8 frames per launch: launches, avg us, us per frame | 32 frames per launch: launches, avg us, us per frame |")
print("|---|---|---|")
for k in sorted(a4, key=lambda k: -sum(a4[k])):
    if k not in a1 or sum(a4[k]) < 300: continue
    # the second half of the launches (timed steps)
    v1, v4 = a1[k][len(a1[k]) // 2:], a4[k][len(a4[k]) // 2:]
    m1, m4 = sum(v1) / len(v1), sum(v4) / len(v4)
    print("| %s | %d, %.1f, %.2f | %d, %.1f, %.2f |" % (k[:70], len(v1), m1, m1 / 8, len(v4), m4, m4 / 32))
PY
