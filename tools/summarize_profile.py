"""Turns rocprofv3 CSV output into the small tracked summaries under profiles/ (development aid).

    python tools/summarize_profile.py stats  <kernel_stats.csv> <out.md> [bench.json] [--cmd "<the traced command line>"]
    python tools/summarize_profile.py pmc    <counter_collection.csv> <kernel_trace.csv> <out.md>
"""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"namespace\)::(\w+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:70]


def stats(path, out, bench=None, cmd=None):
    rows = list(csv.DictReader(open(path)))
    with open(out, "w") as f:
        # the header carries the command line that was actually traced (round 3's fp32 summary carried the bf16 run's)
        f.write("# %s\n" % (cmd or "rocprofv3 --kernel-trace --stats -- (command line not recorded)"))
        f.write("# (--lanes 1, where given: one generator at a time, so a kernel's duration is its own -- the default two lanes overlap\n"
                "#  the kernels of two batches, which is what the bench line gains from; bench.py's roofline pass times its kernels\n"
                "#  on one lane for the same reason.  __amd_rocclr_copyBuffer rows are the one-time weight uploads of the setup,\n"
                "#  outside the timed steps)\n\n")
        if bench:
            f.write("bench.py line of the same code (un-profiled run):\n\n```json\n%s\n```\n\n" % open(bench).read().strip())
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows[:24]:
            f.write("| `%s` | %s | %.3f | %.1f | %.2f |\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                            float(r["AverageNs"]) / 1e3, float(r["Percentage"])))


def pmc(counters, trace, out):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(counters)):
        d = disp.setdefault(r["Dispatch_Id"], {"name": short(r["Kernel_Name"]), "grid": r["Grid_Size"]})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(trace))}
    agg = collections.OrderedDict()
    for k, d in disp.items():
        if "igemm" not in d["name"] and "stem_bf16x3" not in d["name"] and "halo" not in d["name"]:
            continue
        a = agg.setdefault((d["name"], d["grid"]), collections.defaultdict(float))
        a["n"] += 1
        a["ns"] += dur.get(k, 0)
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CU_CYCLES"):
            a[c] += d.get(c, 0.0)
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE (own pass) -- bench.py --steps 2\n\n"
                "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles); kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs\n"
                "(GRBM_GUI_ACTIVE is summed over the 8 XCDs; effective clock = that / 8 / duration).  Counter passes serialise\n"
                "kernels and run slower than the un-profiled bench.\n\n"
                "| kernel | grid (threads) | launches | avg us | eff. clock GHz | MFMA busy / SIMD-cycles |\n|---|---|---|---|---|---|\n")
        for (name, grid), a in agg.items():
            cyc = a["GRBM_GUI_ACTIVE"] / 8.0
            util = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc) if cyc else 0.0
            f.write("| `%s` | %s | %d | %.1f | %.2f | %.3f |\n" % (name, grid, a["n"], a["ns"] / a["n"] / 1e3,
                                                                cyc / a["ns"] if a["ns"] else 0, util))


def traffic(fetch_csv, write_csv, out_json, out_md):
    """Per-kernel fabric-side bytes per launch from two PMC passes (FETCH_SIZE, WRITE_SIZE; each its own pass: they do not
    fit one).  Units KB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies the 128-byte requests of
    16-byte-per-lane loads at 64 B, so it is doubled; WRITE_SIZE is taken as reported.  Infinity-Cache hits are counted."""
    res = {}
    for path, cname, key, mult in ((fetch_csv, "FETCH_SIZE", "fetch_bytes_per_launch", 2.0),
                                   (write_csv, "WRITE_SIZE", "write_bytes_per_launch", 1.0)):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != cname:
                continue
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"]) * 1024.0 * mult
        for k, (n, v) in agg.items():
            res.setdefault(k, {})["launches"] = n
            res[k][key] = v / n
    # stamp: bench.py attaches these bytes only while the kernel sources are the ones they were measured on
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench   # one definition of the digest: bench.py's
    digest = bench.csrc_digest()
    commit = os.environ.get("LWG_COMMIT", "")
    if not commit:
        try:
            commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
        except Exception:
            commit = "unknown (no .git on the GPU box; see the commit that added this file)"
    res["_stamp"] = {"csrc_sha256": digest, "commit": commit}
    json.dump(res, open(out_json, "w"), indent=1, sort_keys=True)
    with open(out_md, "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two own passes) -- bench.py --steps 2\n\n"
                "Fabric-side bytes per launch (L2 misses, Infinity-Cache hits included); FETCH_SIZE x2 (gfx950 tallies the 128-B\n"
                "requests of 16-B/lane loads at 64 B), WRITE_SIZE as reported.\n\n| kernel | launches | read MB | written MB |\n|---|---|---|---|\n")
        rows = [(k, v) for k, v in res.items() if k != "_stamp"]
        for k, v in sorted(rows, key=lambda kv: -kv[1].get("fetch_bytes_per_launch", 0) * kv[1].get("launches", 0))[:16]:
            f.write("| `%s` | %d | %.1f | %.1f |\n" % (k, v.get("launches", 0), v.get("fetch_bytes_per_launch", 0) / 1e6,
                                                   v.get("write_bytes_per_launch", 0) / 1e6))


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
    elif sys.argv[1] == "stats":
        argv, cmd = list(sys.argv), None
        if "--cmd" in argv:
            i = argv.index("--cmd")
            cmd = argv[i + 1]
            del argv[i:i + 2]
        stats(argv[2], argv[3], argv[4] if len(argv) > 4 else None, cmd)
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
