"""Reads a rocprofv3 kernel trace of the two-lane bench: per queue, busy time; time with >= 2 kernels in flight; and, for the short
memory-side kernels, how much of their duration had a kernel of another queue in flight."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
ev.sort()
# keep the last 60 % of the trace (the timed window)
t0 = ev[int(len(ev) * 0.4)][0]
ev = [e for e in ev if e[0] >= t0]
queues = defaultdict(float)
for s, e, n, q, st in ev:
    queues[q] += (e - s) / 1e3
span = (max(e[1] for e in ev) - ev[0][0]) / 1e3
print("# two-lane bench, last 60 %% of the trace: %d dispatches over %.1f us" % (len(ev), span))
print("busy time per queue (us):", {q: round(v, 1) for q, v in queues.items()}, " sum / span = %.2f" % (sum(queues.values()) / span))
# sweep: time with k kernels in flight
pts = []
for s, e, n, q, st in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
depth, last, hist = 0, pts[0][0], defaultdict(float)
for t, d in pts:
    hist[depth] += (t - last) / 1e3
    last = t
    depth += d
print("time with k kernels in flight (us):", {k: round(v, 1) for k, v in sorted(hist.items())})
# per kernel class: share of its duration overlapped by a kernel of another queue
byq = defaultdict(list)
for s, e, n, q, st in ev:
    byq[q].append((s, e))
def overlapped(s, e, q):
    tot = 0
    for q2, iv in byq.items():
        if q2 == q:
            continue
        for s2, e2 in iv:
            if e2 <= s:
                continue
            if s2 >= e:
                break
            tot += min(e, e2) - max(s, s2)
    return tot
acc = defaultdict(lambda: [0, 0.0, 0.0])
for s, e, n, q, st in ev:
    key = n.split("(")[0].split("<")[0].replace("void lwg::(anonymous namespace)::", "")
    a = acc[key]
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] += overlapped(s, e, q) / 1e3
print("| kernel | launches | total us | of which beside another queue's kernel |")
print("|---|---|---|---|")
for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print("| `%s` | %d | %.1f | %.0f %% |" % (k, a[0], a[1], 100 * a[2] / max(a[1], 1e-9)))
