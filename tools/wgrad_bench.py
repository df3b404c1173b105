#!/usr/bin/env python
"""Times lwg_conv2d_backward_weight (precision bf16x3) on the generator's layer shapes at training batch size (development aid).

    python tools/wgrad_bench.py [batch] [image_size]      env: LWG_WGRAD_ROW3=0 (per-tap kernel), LWG_WGRAD_ROW3_TA, LWG_WGRAD_ROW3_SLICES
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from impersonator_amd import ops  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    shapes = [("trunk 512->512", 512, 512, 3, 1, 1, size // 8), ("skipper.0 512->256", 512, 256, 3, 1, 1, size // 4),
              ("skipper.1 256->128", 256, 128, 3, 1, 1, size // 2), ("skipper.2 128->64", 128, 64, 3, 1, 1, size),
              ("encoder.3 256->512 s2", 256, 512, 3, 2, 1, size // 4), ("stem 8->64 7x7", 8, 64, 7, 1, 3, size)]
    g = torch.Generator().manual_seed(0)
    for name, cin, cout, k, stride, pad, h in shapes:
        x = torch.randn(n, h, h, cin, generator=g).cuda()
        ho = (h + 2 * pad - k) // stride + 1
        dy = torch.randn(n, ho, ho, cout, generator=g).cuda()
        for _ in range(3):
            ops.conv2d_backward_weight(x, dy, (cout, cin, k, k), stride, pad, precision="bf16x3")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            ops.conv2d_backward_weight(x, dy, (cout, cin, k, k), stride, pad, precision="bf16x3")
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        flop = 2.0 * n * ho * ho * cout * cin * k * k
        print("%-24s %8.1f us  %7.1f TFLOP/s algorithmic  (pipe %.3f)" % (name, us, flop / us / 1e6, 3 * flop / us / 1e6 / 2500))


if __name__ == "__main__":
    main()
