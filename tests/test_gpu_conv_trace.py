"""The measurement hook of include/lwg.h (lwg_conv_trace): with a record buffer set, conv launches append per-wave clock records
(tools/conv_trace.py reads them); it must not change a result and must switch off cleanly."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_conv_trace_records_and_leaves_results_alone():
    from impersonator_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(2, 32, 32, 64, generator=g) * 2 - 1).cuda()
    w = (torch.randn(128, 64, 3, 3, generator=g) * 0.05).cuda()
    ref = ops.conv2d_forward(x, w, None, 1, 1, False, "bf16x3")
    buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
    _lib.check(lib.lwg_conv_trace(_lib.ptr(buf), buf.numel() * 8))
    try:
        got = ops.conv2d_forward(x, w, None, 1, 1, False, "bf16x3")
        torch.cuda.synchronize()
        info = (ctypes.c_longlong * 10)()
        assert lib.lwg_conv_trace_launch(0, info) == 0
        off, gx, gy, gz, waves, stages, cin, cout, hm, n = list(info)
        assert (cin, cout, hm, n) == (64, 128, 32, 2) and abs(stages) == 9 * 64 // 32 and gx * gy * gz >= 1
        assert lib.lwg_conv_trace_launch(1, info) != 0          # one traced launch so far
        rec = buf.cpu().numpy().view(np.uint64)[off // 8: off // 8 + gx * gy * gz * waves * 8].reshape(-1, 8).astype(np.int64)
        assert (rec[:, 3] > rec[:, 0]).all() and (rec[:, 2] >= rec[:, 1]).all() and (rec[:, 1] >= rec[:, 0]).all()
    finally:
        _lib.check(lib.lwg_conv_trace(None, 0))
    assert torch.equal(ref, got)
    again = ops.conv2d_forward(x, w, None, 1, 1, False, "bf16x3")
    assert torch.equal(ref, again) and lib.lwg_conv_trace_launch(0, info) != 0
