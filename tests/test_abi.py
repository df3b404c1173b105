"""CPU checks of the drop-in boundary: liblwg.so builds, loads, and exports exactly what include/lwg.h declares.
No compute entry point is called here (no GPU in the build container)."""
import ctypes
import subprocess

import pytest

from impersonator_amd import _lib, build


def test_library_builds_for_gfx950():
    path = build.build()
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", path], capture_output=True, text=True).stdout
    assert ".hip_fatbin" in out or "hip_fatbin" in out


def test_every_header_symbol_is_exported_and_bound():
    lib = _lib.load()
    names = _lib.header_symbols()
    assert len(names) >= 25
    assert names == sorted(_lib._PROTOS), "ctypes prototypes out of sync with include/lwg.h"
    for n in names:
        assert hasattr(lib, n), n


def test_version_and_error_text():
    lib = _lib.load()
    assert lib.lwg_version() >= 100
    assert isinstance(lib.lwg_last_error(), bytes)


def test_argument_validation_without_a_device():
    # validation happens before any HIP call, so these paths are exercised on CPU
    lib = _lib.load()
    assert lib.lwg_rasterize_workspace_bytes(8, 13776, 256) > 8 * 256 * 256 * 8
    assert lib.lwg_rasterize_workspace_bytes(0, 1, 1) == 0
    assert lib.lwg_rasterize_fim_wim(None, 1, 1, 8, 0.1, 100.0, None, None, None, None, 0, None) == -1
    assert b"NULL" in lib.lwg_last_error()
    assert lib.lwg_grid_sample(None, 1, 1, 1, 1, None, 1, 1, 1, 0, None, None) == -1
    assert lib.lwg_generator_missing_weights(None) == -1


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    rc = _lib.load().lwg_generator_create(ctypes.byref(h), 6, 6, 64, 6, 256, 8)
    assert rc == -4 and h.value is None
    from impersonator_amd.networks.generator import ImpersonatorGenerator
    G = ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6)
    with pytest.raises(RuntimeError):
        G.encode_src(torch.zeros(1, 6, 256, 256))


def test_product_package_never_imports_the_oracle():
    import os
    root = os.path.dirname(os.path.abspath(build.__file__))
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "raster_ref" not in text, f
