"""CPU: the geometry kernels (rasteriser, SMPL, warps -- the ones Imitator.predict_batches may run underneath the generators
with overlap_geometry=True) contain no instance of the code shape the co-residency miscompute of DESIGN.md section 5.1 needs:
a VALU rewrite of a multi-dword global store's data registers within 24 wait states of the store (tools/store_hazard_lint.py
on the gfx950 assembly; hipcc cross-compiles here).  A guard for future edits of those kernels, not a proof of absence of
other triggers."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("src", ["raster.hip", "smpl.hip", "warp.hip"])
def test_geometry_kernels_have_no_store_then_rewrite_site(src, tmp_path):
    import store_hazard_lint as lint
    from impersonator_amd import build as lwg_build
    extra = dict(lwg_build.SOURCES)[src]
    asm = str(tmp_path / (src + ".s"))
    subprocess.run([lwg_build._hipcc()] + lwg_build.COMMON + extra + ["--cuda-device-only", "-S", os.path.join(lwg_build.CSRC, src), "-o", asm],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    res = lint.lint(asm)
    assert res or src != "raster.hip", "no kernel with a multi-dword store found: did the parser break?"
    bad = {k: v for k, v in res.items() if v["sites"]}
    assert not bad, "store-then-rewrite sites (kernel: count, closest rewrite in wait states): %s" % {
        k[:60]: (v["sites"], v["min"]) for k, v in bad.items()}
