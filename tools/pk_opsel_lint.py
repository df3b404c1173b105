"""Static check for the instruction form behind DESIGN.md section 5.1 in every kernel of liblwg (development aid, CPU only).

    python tools/pk_opsel_lint.py [out.md]

What the reproducer established (tools/coresidency_repro.hip victims 60-73, tools/coresidency_asm_variants.py;
profiles/r03_coresidency.md): on a CU shared with the LDS-read + bf16-MFMA loop of the conv kernels, a PACKED-FP32 VALU
instruction -- v_pk_mul_f32, v_pk_add_f32, v_pk_fma_f32 -- whose op_sel bit for the SECOND source is set (its low result
reads the high half of src1) returns wrong values; 300 of 300 launches, ~2 million wrong words per launch in the minimal
victim.  op_sel on src0 or src2, every op_sel_hi form (the broadcasts hipcc emits everywhere), v_pk_mov_b32 with op_sel, and
the same instruction on an idle device, on disjoint CUs, or beside the exact-fp32 conv kernel or an MFMA-only loop: clean.

hipcc forms these instructions itself (SLP vectoriser + operand folding), so source code cannot promise their absence.
This tool compiles every csrc/*.hip to gfx950 assembly and lists, per kernel, the packed-fp32 instructions and those with
op_sel[1] = 1 (the form that fails) -- tests/test_pk_opsel_lint.py fails on any.  v_pk_*_f16/bf16/int forms are listed too
when they carry the bit (not observed to fail, not tested either)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from impersonator_amd import build as lwg_build  # noqa: E402

PK = re.compile(r"^\s+(v_pk_\w+)\s+(.*)$")
OPSEL = re.compile(r"op_sel:\[([01])(?:,([01]))?(?:,([01]))?\]")


def lint(asm_path):
    """{kernel: {"pk": packed-fp32 instructions, "bad": [instructions with op_sel set for src1], "other": same bit on non-fp32 packed ops}}"""
    out, kernel = {}, None
    for line in open(asm_path):
        m = re.match(r"^(_Z[\w$.]+|[A-Za-z_][\w$.]*):", line)
        if m and not line.startswith(".L"):
            kernel = m.group(1)
            continue
        m = PK.match(line)
        if not m or kernel is None:
            continue
        mn, ops = m.group(1), m.group(2).split(";")[0].strip()
        if mn.startswith("v_pk_mov"):
            continue                      # op_sel picks the halves of a move: tested clean (victims 71, 72)
        rec = out.setdefault(kernel, {"pk": 0, "bad": [], "other": []})
        fp32 = mn.endswith("_f32")
        rec["pk"] += 1 if fp32 else 0
        sel = OPSEL.search(ops)
        if sel and sel.group(2) == "1":
            rec["bad" if fp32 else "other"].append("%s %s" % (mn, ops))
    return out


def compile_asm(src, out_path):
    extra = dict(lwg_build.SOURCES)[src]
    subprocess.run([lwg_build._hipcc()] + lwg_build.COMMON + extra + ["--cuda-device-only", "-S", os.path.join(lwg_build.CSRC, src), "-o", out_path],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


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
        for src, _ in lwg_build.SOURCES:
            s = os.path.join(tmp, src.replace(".hip", ".s"))
            compile_asm(src, s)
            res = lint(s)
            names = demangle(list(res))
            for k, v in res.items():
                short = re.sub(r"lwg::\(anonymous namespace\)::", "", names[k]).replace("void ", "").split("(")[0]
                rows.append((src, short, v["pk"], len(v["bad"]), len(v["other"]), (v["bad"] or [""])[0]))
    rows.sort(key=lambda r: (-r[3], -r[2], r[0], r[1]))
    lines = ["# Packed-fp32 instructions per kernel of liblwg, and those with op_sel set for src1 (tools/pk_opsel_lint.py)\n",
             "The second column of numbers is the form that miscomputes on a CU shared with the bf16x3 conv kernels (DESIGN.md section 5.1, "
             "`profiles/r03_coresidency.md`): it has to read 0 everywhere (`tests/test_pk_opsel_lint.py`).  raster.hip, smpl.hip and warp.hip "
             "are built with `-fno-slp-vectorize`: no packed-fp32 instruction at all in the kernels that may run underneath the generators.\n",
             "| source | kernel | packed-fp32 instructions | with op_sel[src1] = 1 | other packed ops with that bit |", "|---|---|---|---|---|"]
    for src, k, pk, bad, other, ex in rows:
        lines.append("| %s | `%s` | %d | %d%s | %d |" % (src, k[:90], pk, bad, (" (`%s`)" % ex[:80]) if ex else "", other))
    text = "\n".join(lines) + "\n"
    print(text)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)


if __name__ == "__main__":
    main()
