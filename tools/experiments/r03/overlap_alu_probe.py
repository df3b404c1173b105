"""Development aid: do plain arithmetic kernels (torch's: int32 multiply, integer division, float division,
a gather + 12-byte-strided store) compute wrong values beside conv_igemm_bf16x3?  python tools/overlap_alu_probe.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impersonator_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator(device="cuda").manual_seed(1)
n = 1 << 22
a = torch.randint(1, 1 << 20, (n,), device="cuda", dtype=torch.int32, generator=g)
b = torch.randint(1, 1 << 10, (n,), device="cuda", dtype=torch.int32, generator=g)
fa = torch.rand(n, device="cuda", generator=g) + 0.5
fb = torch.rand(n, device="cuda", generator=g) + 0.5
idx = torch.randint(0, n // 3, (n // 3,), device="cuda", generator=g)
v3 = torch.rand(n // 3, 3, device="cuda", generator=g)
ref = dict(mul=a * b, div=torch.div(a, b, rounding_mode="floor"), fdiv=fa / fb, gather=v3[idx] * 2.0 + 1.0)
xx, ww = torch.randn(8, 32, 32, 512, device="cuda"), torch.randn(512, 512, 3, 3, device="cuda") * 0.02
lanes, side = [torch.cuda.Stream(), torch.cuda.Stream()], torch.cuda.Stream()
bad = torch.zeros(4, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for it in range(iters):
    for st in lanes:
        with torch.cuda.stream(st):
            for _ in range(12):
                ops.conv2d_forward(xx, ww, None, 1, 1, precision="bf16x3")
    with torch.cuda.stream(side):
        for _ in range(3):
            bad[0] += ((a * b) != ref["mul"]).sum()
            bad[1] += (torch.div(a, b, rounding_mode="floor") != ref["div"]).sum()
            bad[2] += ((fa / fb) != ref["fdiv"]).sum()
            bad[3] += ((v3[idx] * 2.0 + 1.0) != ref["gather"]).sum()
torch.cuda.synchronize()
print("beside conv_igemm_bf16x3, %d launches each: wrong elements  int32 mul %d, int32 floor-div %d, float div %d, gather+fma %d"
      % (3 * iters, *[int(v) for v in bad]))
