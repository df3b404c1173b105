"""Hand-edited variants of the compiler's assembly for the failing victim of tools/coresidency_repro.hip (development aid).

    python tools/coresidency_asm_variants.py            # -> tools/_build/co/victim9_<variant>.co  (+ the .s files)
    REPRO_CO=tools/_build/co/victim9_B.co REPRO_KERNEL=<name printed by this tool> tools/_build/coresidency_repro_real 300 9 300 0

The reproducer's compiled victims differ in more than the thing each was written to vary (the compiler picks other
instructions when the source changes), so the decisive experiments edit ONE thing in the assembly of a failing victim and
re-assemble it (clang -x assembler, ld.lld), leaving every other instruction and register as it was:

  A  unmodified (victim 9: two 16-byte stores + 24 wait states, then the back-face products)
  B  the two swizzled packed adds  v_pk_add_f32 ... op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]  (and its mirror)
     replaced by the four v_sub_f32 that compute the same values into the same registers
  C  the stores (and their wait states) ahead of the back-face decision deleted, packed adds untouched
  G  the packed adds keep their op_sel swizzle, lose the neg modifiers
  H  the packed adds keep the neg modifiers, lose the op_sel swizzle
  N  A with `s_nop 7` x2 on both sides of each packed add
  I  the swap (low result <- high half, high result <- low half) replaced by high-half-to-both;  J  by low-half-to-both
  K  the same swap on v_pk_mul_f32
(the harness compares every launch with the same kernel's result on an idle device, so variants that compute something
else are still checked bit for bit)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_build", "co")
LLVM = "/opt/rocm/lib/llvm/bin"
MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 9

PK = re.compile(r"^(\s+)v_pk_add_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[(\d),(\d)\] op_sel_hi:\[(\d),(\d)\] neg_lo:\[0,1\] neg_hi:\[0,1\]\s*$")


def edit(lines, variant):
    out, seen_decision, n_pk, n_save = [], False, 0, 0
    for l in lines:
        if "s_and_saveexec" in l and "vcc" in l:
            n_save += 1
            seen_decision = n_save >= 2      # the first one is the kernel's bounds check
        m = PK.match(l)
        if m and not seen_decision:
            n_pk += 1
            ind, d0, d1, a0, a1, b0, b1, s0, s1, h0, h1 = m.groups()
            d, a, b = (int(d0), int(d1)), (int(a0), int(a1)), (int(b0), int(b1))
            if variant == "B":
                out.append("%sv_sub_f32_e32 v%d, v%d, v%d" % (ind, d[0], a[int(s0)], b[int(s1)]))
                out.append("%sv_sub_f32_e32 v%d, v%d, v%d" % (ind, d[1], a[int(h0)], b[int(h1)]))
                continue
            if variant == "G":
                l = re.sub(r" neg_lo:\[0,1\] neg_hi:\[0,1\]", "", l)
            if variant == "H":
                l = re.sub(r" op_sel:\[\d,\d\] op_sel_hi:\[\d,\d\]", "", l)
            if variant == "I":   # one source's HIGH half to both results (op_sel 1, op_sel_hi 1) instead of the swap
                l = l.replace("op_sel:[1,0] op_sel_hi:[0,1]", "op_sel:[1,0] op_sel_hi:[1,1]").replace("op_sel:[0,1] op_sel_hi:[1,0]", "op_sel:[0,1] op_sel_hi:[1,1]")
            if variant == "J":   # one source's LOW half to both results (the broadcast the compiler uses everywhere)
                l = l.replace("op_sel:[1,0] op_sel_hi:[0,1]", "op_sel:[0,0] op_sel_hi:[0,1]").replace("op_sel:[0,1] op_sel_hi:[1,0]", "op_sel:[0,0] op_sel_hi:[1,0]")
            if variant == "K":   # the same swap on a packed multiply
                l = l.replace("v_pk_add_f32", "v_pk_mul_f32")
            if variant == "N":
                out += [ind + "s_nop 7", ind + "s_nop 7", l, ind + "s_nop 7", ind + "s_nop 7"]
                continue
        if variant == "C" and not seen_decision and (re.match(r"^\s+global_store_", l) or re.match(r"^\s+s_nop (15|7)\s*$", l)):
            continue
        out.append(l)
    assert n_pk == 2, "expected two swizzled packed adds ahead of the back-face decision, found %d" % n_pk
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    dev = os.path.join(OUT, "repro_dev.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
                    os.path.join(ROOT, "tools", "coresidency_repro.hip"), "-o", dev], check=True, stderr=subprocess.DEVNULL)
    lines = open(dev).read().split("\n")
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z.*victim_fusedILi%dE.*:" % MODE, l)][0]
    name = lines[start].split(":")[0]
    end = [i for i in range(start, len(lines)) if "s_endpgm" in lines[i]][0]
    for variant in "ABCGHNIJK":
        body = edit(lines[start:end], variant) if variant != "A" else lines[start:end]
        s = os.path.join(OUT, "victim%d_%s.s" % (MODE, variant))
        open(s, "w").write("\n".join(lines[:start] + body + lines[end:]))
        obj = s[:-2] + ".o"
        subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", obj], check=True)
        subprocess.run([LLVM + "/ld.lld", "-shared", obj, "-o", s[:-2] + ".co"], check=True)
        os.remove(obj)
    print(name)


if __name__ == "__main__":
    main()
