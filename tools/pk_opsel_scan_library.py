"""Scan a shared library's embedded gfx950 code objects for the instruction form of DESIGN.md section 5.1 (development aid, CPU).

    python tools/pk_opsel_scan_library.py [library=.../torch/lib/libtorch_hip.so] [out.md]

The library's device code sits in compressed clang offload bundles ("CCOB" blobs); each is cut out, listed and unbundled with
clang-offload-bundler, disassembled with llvm-objdump, and every packed-fp32 instruction is checked for op_sel[src1] = 1
(tools/pk_opsel_lint.py describes the form and why it matters).  Nothing of this runs in the product or its tests: it
answers whether kernels of OTHER libraries that might share CUs with the bf16x3 conv kernels carry the form."""
import collections
import mmap
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def scan(lib):
    """-> dict(bundles, symbols, pk, bad, kernels={kernel: [instructions]}) for every gfx950 code object embedded in `lib`."""
    fh = open(lib, "rb")
    mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
    pos = nblob = npk = nbad = nkern = 0
    bad = collections.OrderedDict()
    with tempfile.TemporaryDirectory() as tmp:
        while True:
            i = mm.find(b"CCOB", pos)
            if i < 0:
                break
            ver, method = struct.unpack_from("<HH", mm, i + 4)
            if ver != 2 or method not in (0, 1):
                pos = i + 4
                continue
            total = struct.unpack_from("<I", mm, i + 8)[0]
            blob, co = os.path.join(tmp, "b.bundle"), os.path.join(tmp, "b.co")
            open(blob, "wb").write(mm[i:i + total])
            ids = subprocess.run([LLVM + "clang-offload-bundler", "--list", "--type=o", "--input=" + blob], stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, text=True).stdout.split()
            tgt = [x for x in ids if x.endswith("gfx950")]
            if tgt:
                subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + blob, "--targets=" + tgt[0], "--output=" + co],
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                if os.path.exists(co) and os.path.getsize(co):
                    dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--mcpu=gfx950", co], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
                    kern = None
                    for line in dis.split("\n"):
                        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                        if m:
                            kern = m.group(1)
                            nkern += 1
                        elif re.search(r"\bv_pk_\w+_f32\b", line):
                            npk += 1
                            m = re.search(r"op_sel:\[([01]),([01])", line)
                            if m and m.group(2) == "1":
                                nbad += 1
                                bad.setdefault(kern, []).append(re.sub(r"\s*//.*", "", line).strip())
                    os.remove(co)
            nblob += 1
            pos = i + total
    return {"bundles": nblob, "symbols": nkern, "pk": npk, "bad": nbad, "kernels": bad}


def file_sha256(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] not in ("", "-") else None
    if lib is None:
        import torch
        lib = os.path.join(os.path.dirname(torch.__file__), "lib", "libtorch_hip.so")
    out = sys.argv[2] if len(sys.argv) > 2 else None
    r = scan(lib)
    nblob, nkern, npk, nbad, bad = r["bundles"], r["symbols"], r["pk"], r["bad"], r["kernels"]
    names = list(bad)
    dem = subprocess.run(["c++filt"], input="\n".join(names) + "\n", stdout=subprocess.PIPE, text=True).stdout.splitlines() if names else []
    fam = collections.Counter(re.sub(r"^void ", "", re.sub(r"[<(].*", "", d))[:80] for d in dem)
    lines = ["# Packed-fp32 instructions with op_sel[src1] = 1 in %s (tools/pk_opsel_scan_library.py)\n" % os.path.basename(lib),
             "`%s`, sha256 `%s`.\n" % (lib, file_sha256(lib)),
             "%d compressed offload bundles, %d gfx950 symbols disassembled, %d packed-fp32 instructions, **%d with op_sel set for src1 in %d kernels**. "
             "That is the form that returns wrong values on a CU shared with liblwg's bf16x3 conv kernels (DESIGN.md section 5.1, "
             "`profiles/r03_coresidency.md`).  It is harmless as long as these kernels do not run on another stream beside those conv "
             "kernels -- which is why `Imitator.overlap_geometry` (torch glue kernels underneath the generators) stays opt-in.\n" % (
                 nblob, nkern, npk, nbad, len(bad)),
             "| kernel family | kernels with the form |", "|---|---|"]
    for k, v in fam.most_common(30):
        lines.append("| `%s` | %d |" % (k, v))
    lines.append("\nFirst instances: " + "; ".join("`%s`" % v[0] for v in list(bad.values())[:4]) + "\n")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text)


if __name__ == "__main__":
    main()
