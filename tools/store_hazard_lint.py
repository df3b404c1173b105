"""Static census of the code shape behind DESIGN.md section 5.1 in every kernel of liblwg (development aid, CPU only).

    python tools/store_hazard_lint.py [out.md]

The torch-free reproducer (tools/coresidency_repro.hip, profiles/r03_coresidency.md) pins the co-residency miscompute to
this shape: a multi-dword global store, then -- within the couple of wait states hipcc's hazard recogniser leaves -- a VALU
instruction that rewrites one of the store's DATA registers; beside conv_igemm_bf16x3 on the same CU the last 16-lane
pass of such a write can be lost.  24 wait states between store and rewrite, or single-dword stores, were clean.

This tool compiles every csrc/*.hip to gfx950 assembly and lists, per kernel, the sites where a VALU instruction writes
a data register of a global/flat/buffer/scratch store of >= 2 dwords within WINDOW wait states of it, along every path the
code can take from the store (branch targets and fall-throughs are followed; an s_nop N counts N + 1, every other
instruction 1).  It finds the shape; it cannot tell whether a site ever runs
beside the conv kernels, nor whether the shape alone suffices (the inline-asm micro-victims with exactly this shape
stayed clean): a census of exposure, not a verdict."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from impersonator_amd import build as lwg_build  # noqa: E402

WINDOW = 24
STORE = re.compile(r"^\s+(global|flat|buffer|scratch)_store_dwordx([234])\s+(.*)$")
VREG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def vregs(tok):
    m = VREG.fullmatch(tok.strip())
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def lint(asm_path):
    """{kernel: {"stores", "sites", "min"}}.  From every multi-dword store the walk follows the fall-through path, s_branch
    targets and BOTH sides of conditional branches until WINDOW wait states have passed."""
    out = {}
    kernels, cur = {}, None
    for line in open(asm_path):
        if re.match(r"^_Z[\w$.]+:", line) or (re.match(r"^[A-Za-z_][\w$.]*:", line) and not line.startswith(".L")):
            cur = kernels.setdefault(line.split(":")[0], {"ins": [], "labels": {}})
            continue
        if cur is None:
            continue
        if line.startswith(".L"):
            cur["labels"][line.split(":")[0]] = len(cur["ins"])
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*(;.*)?$", line)
        if m and not m.group(1).startswith("."):
            cur["ins"].append((m.group(1), m.group(2)))
    for kernel, k in kernels.items():
        ins, labels = k["ins"], k["labels"]
        for i, (mn, ops) in enumerate(ins):
            m = STORE.match("\t%s %s" % (mn, ops))
            if not m:
                continue
            parts = [p.strip() for p in ops.split(",")]
            # global/flat: vaddr, vdata, saddr|off ; buffer: vdata, vaddr, srsrc ... ; scratch: vaddr|off, vdata, ...
            data = vregs(parts[0] if mn.startswith("buffer") else parts[1]) if len(parts) > 1 else set()
            if not data:
                continue
            rec = out.setdefault(kernel, {"sites": 0, "min": 1 << 30, "stores": 0})
            rec["stores"] += 1
            # `dead`: the path is only taken with EXEC = 0 (fall-through of s_cbranch_execnz, target of s_cbranch_execz):
            # VALU instructions write nothing there until something sets EXEC again
            best, seen, todo = None, set(), [(i + 1, 0, False)]
            while todo:
                pc, dist, dead = todo.pop()
                while pc < len(ins) and (pc, dist, dead) not in seen:
                    seen.add((pc, dist, dead))
                    mn2, ops2 = ins[pc]
                    dist += int(ops2.strip() or 0) + 1 if mn2 == "s_nop" else 1
                    if dist > WINDOW or mn2 == "s_endpgm":
                        break
                    dst = ops2.split(",")[0].strip()
                    if mn2.startswith("s_") and (dst == "exec" or "saveexec" in mn2):
                        dead = False
                    if not dead and mn2.startswith("v_") and not mn2.startswith(("v_cmp", "v_cmpx", "v_readfirstlane", "v_readlane")):
                        if vregs(dst) & data:
                            best = dist if best is None else min(best, dist)
                            break
                    if mn2 == "s_branch" or mn2.startswith("s_cbranch"):
                        tgt = labels.get(ops2.strip())
                        if tgt is not None:
                            todo.append((tgt, dist, dead or mn2 == "s_cbranch_execz"))
                        if mn2 == "s_branch":
                            break
                        if mn2 == "s_cbranch_execnz":
                            dead = True
                    pc += 1
            if best is not None:
                rec["sites"] += 1
                rec["min"] = min(rec["min"], best)
    return out


def demangle(names):
    try:
        p = subprocess.run(["c++filt"], input="\n".join(names) + "\n", stdout=subprocess.PIPE, text=True)
        outs = p.stdout.splitlines()
        return dict(zip(names, outs)) if len(outs) == len(names) else {n: n for n in names}
    except Exception:
        return {n: n for n in names}


def main():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for src, extra in lwg_build.SOURCES:
            s = os.path.join(tmp, src.replace(".hip", ".s"))
            cmd = [lwg_build._hipcc()] + lwg_build.COMMON + extra + ["--cuda-device-only", "-S", os.path.join(lwg_build.CSRC, src), "-o", s]
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            res = lint(s)
            names = demangle(list(res))
            for k, v in res.items():
                short = re.sub(r"lwg::\(anonymous namespace\)::", "", names[k]).replace("void ", "").split("(")[0]
                rows.append((src, short, v["stores"], v["sites"], v["min"] if v["sites"] else None))
    rows.sort(key=lambda r: (-r[3], r[0], r[1]))
    lines = ["# Census of \"multi-dword store, then a VALU rewrite of its data registers within %d wait states\" in liblwg (tools/store_hazard_lint.py)\n" % WINDOW,
             "The shape the torch-free reproducer pins the co-residency miscompute of DESIGN.md section 5.1 to.  A site is exposure, not a failure: "
             "the shape has to run on a CU shared with `conv_igemm_bf16x3` / `conv3x3_halo_bf16x3`, and hand-written sequences of exactly this shape "
             "did not fail.  Kernels that share CUs with the conv kernels in the two-lane pipeline are compared bit for bit with the sequential "
             "order (`tests/test_gpu_imitator.py`, `profiles/r03_lane_stress.log`), `personalize` and the training iteration with their "
             "idle-device results (`tests/test_gpu_coresidency.py`).\n",
             "| source | kernel | multi-dword stores | sites | closest rewrite (wait states) |", "|---|---|---|---|---|"]
    for src, k, st, sites, mn in rows:
        if st:
            lines.append("| %s | `%s` | %d | %d | %s |" % (src, k[:90], st, sites, mn if mn is not None else "–"))
    text = "\n".join(lines) + "\n"
    print(text)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)


if __name__ == "__main__":
    main()
