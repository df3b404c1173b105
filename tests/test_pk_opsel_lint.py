"""CPU: no kernel of liblwg contains the instruction form that miscomputes on a CU shared with the bf16x3 conv kernels
(DESIGN.md section 5.1): a packed-fp32 VALU instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) with op_sel set for
its second source.  hipcc forms such instructions on its own, so the guard is on the compiled assembly of every source
(tools/pk_opsel_lint.py; hipcc cross-compiles here), and the geometry sources -- the kernels that may run underneath the
generators -- are built without the SLP vectoriser and must not contain a packed-fp32 instruction at all."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
from impersonator_amd import build as lwg_build  # noqa: E402

GEOMETRY = ("raster.hip", "smpl.hip", "warp.hip")


@pytest.mark.parametrize("src", [s for s, _ in lwg_build.SOURCES])
def test_no_packed_fp32_instruction_with_op_sel_on_src1(src, tmp_path):
    import pk_opsel_lint as lint
    asm = str(tmp_path / (src + ".s"))
    lint.compile_asm(src, asm)
    res = lint.lint(asm)
    assert res or src not in ("conv.hip", "inpaint.hip"), "no packed instruction found at all: did the parser break?"
    bad = {k[:70]: v["bad"][:2] for k, v in res.items() if v["bad"]}
    assert not bad, "packed-fp32 instructions with op_sel[src1] = 1: %s" % bad
    if src in GEOMETRY:
        assert not any(v["pk"] for v in res.values()), "packed-fp32 instructions in %s (built with -fno-slp-vectorize?)" % src


def test_the_lint_sees_the_form(tmp_path):
    import pk_opsel_lint as lint
    asm = tmp_path / "k.s"
    asm.write_text("""
_Zbad:
\tv_pk_mul_f32 v[2:3], v[20:21], v[20:21] op_sel:[0,1] op_sel_hi:[1,0]
\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[0,1,0]
\ts_endpgm
_Zfine:
\tv_pk_mul_f32 v[2:3], v[4:5], v[6:7]
\tv_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]
\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[1,0,1] op_sel_hi:[0,1,1]
\tv_pk_fma_f32 v[2:3], s[4:5], v[6:7], v[8:9] op_sel_hi:[0,1,1]
\tv_pk_mov_b32 v[2:3], v[4:5], v[6:7] op_sel:[0,1]
\ts_endpgm
""")
    res = lint.lint(str(asm))
    assert len(res["_Zbad"]["bad"]) == 2 and res["_Zbad"]["pk"] == 2
    assert res["_Zfine"]["bad"] == [] and res["_Zfine"]["pk"] == 4


def _rccl_library():
    import torch
    return os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")


def _check_scanned(lib):
    """The library must be on the recorded allow-list (tests/golden/pk_opsel_scanned_libraries.json, by sha256) with no instruction
    of the form; a library that is not listed is scanned here and now (minutes) and must be clean."""
    import json
    import pk_opsel_scan_library as scanlib
    listed = json.load(open(os.path.join(ROOT, "tests", "golden", "pk_opsel_scanned_libraries.json")))
    digest = scanlib.file_sha256(lib)
    rec = listed.get(digest)
    if rec is None:
        rec = scanlib.scan(lib)
        assert rec["symbols"] > 0, "no gfx950 code object found in %s: did the scanner break?" % lib
    assert rec["bad"] == 0, "%s (sha256 %s): %d packed-fp32 instructions with op_sel[src1] = 1" % (lib, digest, rec["bad"])
    return digest, rec


def test_rccl_kernels_do_not_carry_the_form():
    """sharding.GradientBuckets runs RCCL all-reduces on a side stream UNDER the bf16x3 backward convs: the one third-party library
    whose kernels are co-resident with them by design.  Its gfx950 kernels must not contain the instruction form either."""
    lib = _rccl_library()
    if not os.path.exists(lib):
        pytest.skip("torch ships no librccl.so here")
    _check_scanned(lib)


@pytest.mark.gpu
def test_rccl_library_of_this_box_is_the_scanned_one():
    """The same check against the library the GPU box's torch loads (another image would bring another RCCL)."""
    lib = _rccl_library()
    assert os.path.exists(lib)
    digest, rec = _check_scanned(lib)
    print("librccl.so sha256 %s: %d gfx950 symbols, %d packed-fp32 instructions, %d with op_sel[src1]" % (digest, rec["symbols"], rec["pk"], rec["bad"]))
