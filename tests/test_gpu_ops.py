"""GPU parity of the op-level convolution and its gradients (train.hip, building blocks of the generator-side training
step) against torch autograd on the CPU, on the layer shapes of the ResUnet generator (networks/generator.py:80-133)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (name, Cin, Cout, k, stride, pad, transposed, H)
CASES = [
    ("res 3x3 s1", 64, 64, 3, 1, 1, False, 32),
    ("skipper 3x3 s1 (cat)", 256, 128, 3, 1, 1, False, 32),
    ("encoder 3x3 s2", 64, 128, 3, 2, 1, False, 32),
    ("decoder convT 3x3 s2", 128, 64, 3, 2, 1, True, 16),
    ("stem 7x7 (Cin 6 in 8)", 8, 64, 7, 1, 3, False, 32),
    ("1x1", 64, 64, 1, 1, 0, False, 24),
    # row segments that do not start at column 0 (W = 64: two 32-pixel steps per row) on both tiles of the kernel-row weight gradient
    ("3x3 s1 128->64 @64", 128, 64, 3, 1, 1, False, 64),
    ("3x3 s1 64->128 @64", 64, 128, 3, 1, 1, False, 64),
]


def _rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)


@pytest.mark.parametrize("case", [c for c in CASES if c[1] % 32 == 0], ids=[c[0] for c in CASES if c[1] % 32 == 0])
def test_conv_bf16x3_route(case):
    """lwg_conv2d_desc.precision = 1: forward and data gradient on the inference path's split-bf16 kernel (operands
    carried to 16 significand bits, fp32 accumulation)."""
    from impersonator_amd import ops
    _, cin, cout, k, stride, pad, transposed, H = case
    if H * H % 128:
        H = 32        # the bf16x3 kernel wants whole 128-pixel tiles per image (else the call takes the fp32 kernel)
    g = torch.Generator().manual_seed(9)
    N = 3
    x = torch.randn(N, cin, H, H, generator=g)
    w = torch.randn((cin, cout, k, k) if transposed else (cout, cin, k, k), generator=g) * 0.05
    xr = x.clone().double().requires_grad_(True)
    y = (F.conv_transpose2d(xr, w.double(), stride=2, padding=1, output_padding=1) if transposed
         else F.conv2d(xr, w.double(), None, stride=stride, padding=pad))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    xg, dyg, wg = nhwc(x), nhwc(dy), w.cuda().contiguous()
    y32 = ops.conv2d_forward(xg, wg, None, stride, pad, transposed)
    y16 = ops.conv2d_forward(xg, wg, None, stride, pad, transposed, precision="bf16x3")
    assert not torch.equal(y16, y32), "the bf16x3 route did not run"
    assert _rel(y16.cpu().permute(0, 3, 1, 2).double(), y.detach()) < 3e-5
    dx32 = ops.conv2d_backward_data(dyg, wg, tuple(xg.shape), stride, pad, transposed)
    dx16 = ops.conv2d_backward_data(dyg, wg, tuple(xg.shape), stride, pad, transposed, precision="bf16x3")
    assert not torch.equal(dx16, dx32), "the bf16x3 route did not run"
    assert _rel(dx16.cpu().permute(0, 3, 1, 2).double(), xr.grad) < 3e-5
    # a layer the kernel does not fit (bias) takes the fp32 kernel: same bits as precision 0
    if not transposed:
        b = torch.randn(cout, generator=g).cuda()
        assert torch.equal(ops.conv2d_forward(xg, wg, b, stride, pad, transposed, precision="bf16x3"),
                           ops.conv2d_forward(xg, wg, b, stride, pad, transposed))


@pytest.mark.parametrize("k,stride", [(1, 1), (3, 2)])
def test_ring_kernel_with_an_odd_number_of_tiles_per_image(k, stride):
    """conv_igemm_bf16x3's 256-row tiles (eight waves) are only taken when an image's output grid is whole 256-row tiles: a
    12 x 32 output map is three 128-pixel tiles, and with an even batch big enough to give every CU a tall tile (44 images x 3
    tiles x 4 channel tiles) a 256-row tile would straddle two images -- zero-padded taps and InstanceNorm statistics of the
    wrong image.  Against float64 torch."""
    from impersonator_amd import ops
    g = torch.Generator().manual_seed(77)
    N, cin, cout, Ho, Wo = 44, 32, 512, 12, 32
    H, W = (Ho, Wo) if stride == 1 else (2 * Ho, 2 * Wo)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * 0.05
    y = F.conv2d(x.double(), w.double(), None, stride=stride, padding=k // 2)
    assert tuple(y.shape[2:]) == (Ho, Wo)
    xg, wg = x.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda().contiguous()
    y16 = ops.conv2d_forward(xg, wg, None, stride, k // 2, False, precision="bf16x3")
    assert not torch.equal(y16, ops.conv2d_forward(xg, wg, None, stride, k // 2, False)), "the bf16x3 route did not run"
    per_image = (y16.cpu().permute(0, 3, 1, 2).double() - y).flatten(1).norm(dim=1) / y.flatten(1).norm(dim=1)
    assert float(per_image.max()) < 3e-5, "image %d: %g" % (int(per_image.argmax()), float(per_image.max()))


@pytest.mark.parametrize("H,W", [(4, 32), (12, 96), (8, 64), (36, 160)])
@pytest.mark.parametrize("cin,cout,N", [(32, 64, 2), (64, 128, 5), (128, 64, 1), (64, 256, 9)])
@pytest.mark.parametrize("transposed", [False, True])
def test_halo_kernel_shapes(H, W, cin, cout, N, transposed):
    """conv3x3_halo_bf16x3 away from the generator's shapes: image widths that are not powers of two (three or five 32-column
    tiles per row), a single 4 x 32 tile, one channel slice (power-of-two channel counts: the op-level entry point asks for them), 64- and 128-channel tiles, tile counts that are not multiples of
    eight (no XCD re-deal), the 8 x 32-tile variant (enough tiles at N = 9) -- 3x3 stride-1 and the one-launch transposed conv,
    against float64 torch."""
    from impersonator_amd import ops
    g = torch.Generator().manual_seed(H * 1000 + W + cin + cout + N)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn((cin, cout, 3, 3) if transposed else (cout, cin, 3, 3), generator=g) * 0.05
    y = (F.conv_transpose2d(x.double(), w.double(), stride=2, padding=1, output_padding=1) if transposed
         else F.conv2d(x.double(), w.double(), None, stride=1, padding=1))
    xg, wg = x.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda().contiguous()
    stride = 2 if transposed else 1
    y16 = ops.conv2d_forward(xg, wg, None, stride, 1, transposed, precision="bf16x3")
    y32 = ops.conv2d_forward(xg, wg, None, stride, 1, transposed)
    assert not torch.equal(y16, y32), "the bf16x3 route did not run"
    assert _rel(y16.cpu().permute(0, 3, 1, 2).double(), y) < 3e-5


@pytest.mark.parametrize("N", [3, 1])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_weight_gradient_bf16x3(case, N):
    """lwg_conv2d_backward_weight with precision 1: its own bf16x3 kernel (pixels transposed in registers on the way to
    LDS), every layer shape incl. the 8-channel stem, stride 2 and the transposed conv."""
    from impersonator_amd import ops
    _, cin, cout, k, stride, pad, transposed, H = case
    g = torch.Generator().manual_seed(21)
    x = torch.randn(N, cin, H, H, generator=g)
    w = torch.randn((cin, cout, k, k) if transposed else (cout, cin, k, k), generator=g) * 0.05
    wr = w.clone().double().requires_grad_(True)
    y = (F.conv_transpose2d(x.double(), wr, stride=2, padding=1, output_padding=1) if transposed
         else F.conv2d(x.double(), wr, None, stride=stride, padding=pad))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    xg, dyg = nhwc(x), nhwc(dy)
    dw32 = ops.conv2d_backward_weight(xg, dyg, tuple(w.shape), stride, pad, transposed)
    dw16 = ops.conv2d_backward_weight(xg, dyg, tuple(w.shape), stride, pad, transposed, precision="bf16x3")
    assert not torch.equal(dw16, dw32), "the bf16x3 kernel did not run"
    assert _rel(dw16.cpu().double(), wr.grad) < 3e-5


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_forward_and_gradients(case):
    from impersonator_amd import ops
    _, cin, cout, k, stride, pad, transposed, H = case
    g = torch.Generator().manual_seed(7)
    N = 3
    x = torch.randn(N, cin, H, H, generator=g)
    if cin == 8:
        x[:, 6:] = 0
    w = torch.randn((cin, cout, k, k) if transposed else (cout, cin, k, k), generator=g) * 0.05
    bias = None if transposed else torch.randn(cout, generator=g) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = None if bias is None else bias.clone().requires_grad_(True)
    y = (F.conv_transpose2d(xr, wr, stride=2, padding=1, output_padding=1) if transposed
         else F.conv2d(xr, wr, br, stride=stride, padding=pad))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)

    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    xg, dyg, wg = nhwc(x), nhwc(dy), w.cuda().contiguous()
    yg = ops.conv2d_forward(xg, wg, None if bias is None else bias.cuda(), stride, pad, transposed)
    assert _rel(yg.cpu().permute(0, 3, 1, 2), y.detach()) < 1e-5
    if transposed:
        dw = ops.conv2d_backward_weight(xg, dyg, tuple(w.shape), stride, pad, transposed)
    else:
        dw, db = ops.conv2d_backward_weight(xg, dyg, tuple(w.shape), stride, pad, transposed, with_bias=True)
        assert _rel(db.cpu(), br.grad) < 1e-5
    assert _rel(dw.cpu(), wr.grad) < 2e-5
    if cin >= 64:   # the gradient wrt an 8-channel image input is never needed
        dx = ops.conv2d_backward_data(dyg, wg, tuple(xg.shape), stride, pad, transposed)
        assert _rel(dx.cpu().permute(0, 3, 1, 2), xr.grad) < 1e-5


@pytest.mark.parametrize("H,W", [(64, 64), (24, 40), (9, 33)])
def test_heads_ops(H, W):
    """The regression heads' own kernels (lwg_heads_forward / lwg_heads_backward_weight, data gradient through
    lwg_conv2d_backward_data with Cout = 8) against autograd of conv2d + tanh / sigmoid."""
    from impersonator_amd import ops
    g = torch.Generator().manual_seed(13)
    N = 3
    x = torch.rand(N, 64, H, W, generator=g)                      # post-ReLU activations: non-negative
    w = torch.zeros(8, 64, 7, 7)
    w[:4] = torch.randn(4, 64, 7, 7, generator=g) * 0.03
    xr, wr = x.clone().double().requires_grad_(True), w[:4].clone().double().requires_grad_(True)
    pre = F.conv2d(xr, wr, None, stride=1, padding=3)
    img, mask = torch.tanh(pre[:, 0:3]), torch.sigmoid(pre[:, 3:4])
    d_img, d_mask = torch.randn(img.shape, generator=g).double(), torch.randn(mask.shape, generator=g).double()
    (img * d_img).sum().add((mask * d_mask).sum()).backward()
    d_pre = torch.cat([d_img * (1 - img * img), d_mask * mask * (1 - mask)], dim=1).detach()

    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().float().cuda()
    xg, wg = nhwc(x), w.cuda()
    color, m = ops.heads_forward(xg, wg)
    assert float((color.cpu().double() - img.detach()).abs().max()) < 1e-5     # fp32 sums of 3136 products
    assert float((m.cpu().double() - mask.detach()).abs().max()) < 1e-5
    d8 = torch.zeros(N, H, W, 8, device="cuda")
    d8[..., :4] = nhwc(d_pre)
    dw = ops.heads_backward_weight(xg, d8)
    assert _rel(dw[:4].cpu().double(), wr.grad) < 2e-5
    assert float(dw[4:].abs().max()) == 0.0
    dx = ops.conv2d_backward_data(d8, wg, tuple(xg.shape), 1, 3)
    assert _rel(dx.cpu().permute(0, 3, 1, 2).double(), xr.grad) < 1e-5


def test_heads_forward_clamps_negative_inputs():
    """The documented precondition of lwg_heads_forward (include/lwg.h): x is post-ReLU; a signed x is read through
    max(x, 0) -- stated, and pinned here so that a caller relying on anything else finds out."""
    from impersonator_amd import ops
    g = torch.Generator().manual_seed(14)
    x = torch.randn(2, 40, 48, 64, generator=g).cuda()
    w = torch.zeros(8, 64, 7, 7)
    w[:4] = torch.randn(4, 64, 7, 7, generator=g) * 0.03
    c0, m0 = ops.heads_forward(x, w.cuda())
    c1, m1 = ops.heads_forward(x.clamp(min=0), w.cuda())
    assert torch.equal(c0, c1) and torch.equal(m0, m1)
    pre = F.conv2d(x.clamp(min=0).permute(0, 3, 1, 2).cpu().double(), w[:4].double(), None, stride=1, padding=3)
    assert float((c0.cpu().double() - torch.tanh(pre[:, 0:3])).abs().max()) < 1e-5


def test_unsupported_shapes_fail_loudly():
    from impersonator_amd import _lib, ops
    x = torch.zeros(1, 8, 8, 48, device="cuda")
    with pytest.raises(_lib.LwgError):
        ops.conv2d_forward(x, torch.zeros(64, 48, 3, 3, device="cuda"), None, 1, 1)      # 48 channels: not a power of two
    with pytest.raises(RuntimeError):
        ops.conv2d_forward(x.cpu(), torch.zeros(64, 48, 3, 3), None, 1, 1)               # no CPU fallback


@pytest.mark.parametrize("relu,H", [(False, 24), (True, 24), (True, 96)])   # 96x96: the two-stage (slab) reductions
def test_instance_norm_forward_backward(relu, H):
    from impersonator_amd import ops
    g = torch.Generator().manual_seed(11)
    N, C = 3, 64
    x = torch.randn(N, C, H, H, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    dy = torch.randn(N, C, H, H, generator=g)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = F.instance_norm(xr, weight=gr, bias=br, eps=1e-5)
    if relu:
        y = F.relu(y)
    y.backward(dy)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    yg, stats = ops.instance_norm_forward(nhwc(x), gamma.cuda(), beta.cuda(), relu)
    assert _rel(yg.cpu().permute(0, 3, 1, 2), y.detach()) < 1e-5
    dx, dg, db = ops.instance_norm_backward(nhwc(x), yg if relu else None, nhwc(dy), stats, gamma.cuda())
    assert _rel(dx.cpu().permute(0, 3, 1, 2), xr.grad) < 2e-5
    assert _rel(dg.cpu(), gr.grad) < 2e-5 and _rel(db.cpu(), br.grad) < 2e-5


@pytest.mark.parametrize("shared", [False, True])
def test_grid_sample_backward(shared):
    from impersonator_amd import ops
    g = torch.Generator().manual_seed(12)
    n, C, H = 3, 64, 20
    x = torch.randn(1 if shared else n, C, H, H, generator=g)
    grid = torch.rand(n, H, H, 2, generator=g) * 2.6 - 1.3
    grid[0, 5:9, 3:8] = -2
    dy = torch.randn(n, C, H, H, generator=g)
    xr = x.clone().requires_grad_(True)
    F.grid_sample(xr.expand(n, -1, -1, -1) if shared else xr, grid, align_corners=False).backward(dy)
    dx = ops.grid_sample_backward(dy.permute(0, 2, 3, 1).contiguous().cuda(), grid.cuda(), (x.shape[0], H, H, C), False)
    assert _rel(dx.cpu().permute(0, 3, 1, 2), xr.grad) < 1e-5


@pytest.mark.parametrize("shared", [False, True])
def test_grid_sample_backward_is_bit_reproducible_also_on_crowded_texels(shared):
    """The deterministic gradient (csrc/scatter.hip: per-texel contribution lists in pixel order, gathered): identical bits run
    after run where the atomic scatter is not, also when a whole image samples ONE location (lists of 48 x 48 entries: the
    dense-scan path) and when a 3x magnification packs ~9 pixels per texel (the per-thread sort path); one plan serves several
    tensors; values against float64 autograd."""
    from impersonator_amd import ops
    g = torch.Generator().manual_seed(21)
    n, C, H, Ho = 3, 32, 16, 48
    x = torch.randn(1 if shared else n, C, H, H, generator=g)
    grid = (torch.rand(n, Ho, Ho, 2, generator=g) * 2 - 1) * 0.33           # every output pixel lands in the central third
    grid[1] = torch.tensor([0.113, -0.271])                                   # image 1: one location for all 2304 pixels
    grid[2, :, :24] = -2                                                      # sentinel half: samples nothing
    dy = torch.randn(n, C, Ho, Ho, generator=g)
    xr = x.double().clone().requires_grad_(True)
    F.grid_sample(xr.expand(n, -1, -1, -1) if shared else xr, grid.double(), align_corners=False).backward(dy.double())
    dyd, gd = dy.permute(0, 2, 3, 1).contiguous().cuda(), grid.cuda()
    shape = (x.shape[0], H, H, C)
    plan = ops.GridSamplePlan(gd, shape, False)
    runs = [ops.grid_sample_backward(dyd, gd, shape, False, plan=plan) for _ in range(3)]
    runs.append(ops.grid_sample_backward(dyd, gd, shape, False))              # a plan of its own
    torch.cuda.synchronize()
    assert all(torch.equal(r, runs[0]) for r in runs[1:])
    assert _rel(runs[0].cpu().permute(0, 3, 1, 2).double(), xr.grad) < 2e-6
    atomic = ops.grid_sample_backward(dyd, gd, shape, False, deterministic=False)
    assert _rel(atomic.cpu().permute(0, 3, 1, 2).double(), xr.grad) < 1e-5
    # the same plan, another channel count
    dy2 = torch.randn(n, Ho, Ho, 8, generator=g)
    a = ops.grid_sample_backward(dy2.cuda(), gd, (x.shape[0], H, H, 8), False, plan=plan)
    b = ops.grid_sample_backward(dy2.cuda(), gd, (x.shape[0], H, H, 8), False, deterministic=False)
    assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max())
    with pytest.raises(ValueError):
        ops.grid_sample_backward(dyd, gd, (x.shape[0], H + 1, H, C), False, plan=plan)


def test_grid_sample_backward_minifying_flow_takes_the_sorted_lists_and_stays_fast():
    """A strongly minifying flow (256 x 256 output pixels sampling a 16 x 16-texel region: hundreds of contributions per texel,
    thousands of queued texels) sorts each list in LDS instead of rescanning the image per texel (the rescan was
    O(#heavy x pixels)); one image sends all its 9216 pixels to one location (> 4096: the dense-scan path).  Values against float64
    autograd, bit-reproducible, and the plan of the training step's size is built in milliseconds."""
    import time
    from impersonator_amd import ops
    g = torch.Generator().manual_seed(5)
    n, C, H, Ho = 3, 8, 64, 96
    x = torch.randn(n, C, H, H, generator=g)
    grid = (torch.rand(n, Ho, Ho, 2, generator=g) * 2 - 1) * 0.12            # ~8 x 8 texels collect 96 x 96 pixels: ~500 per texel
    grid[1] = torch.tensor([-0.313, 0.207])                                   # 9216 contributions per tap texel: dense scan
    dy = torch.randn(n, C, Ho, Ho, generator=g)
    xr = x.double().clone().requires_grad_(True)
    F.grid_sample(xr, grid.double(), align_corners=False).backward(dy.double())
    dyd, gd = dy.permute(0, 2, 3, 1).contiguous().cuda(), grid.cuda()
    shape = (n, H, H, C)
    runs = [ops.grid_sample_backward(dyd, gd, shape, False) for _ in range(3)]
    assert all(torch.equal(r, runs[0]) for r in runs[1:])
    # fp32 sums of 500 - 9216 terms in a fixed order: 3e-6 of the tensor's scale (the atomic kernel, any order: the same)
    assert _rel(runs[0].cpu().permute(0, 3, 1, 2).double(), xr.grad) < 1e-5
    # the training step's largest level: 4 images of 256 x 256 pixels sampling a 32 x 32-texel window of a 256 x 256 source
    big = ((torch.rand(4, 256, 256, 2, generator=g) * 2 - 1) * 0.125).cuda()
    ops.GridSamplePlan(big, (4, 256, 256, 8), False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ops.GridSamplePlan(big, (4, 256, 256, 8), False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print("plan of a 64x-minifying flow, 4 x 256 x 256 pixels: %.2f ms" % ms)
    assert ms < 60.0, ms   # (round 5, rescanning the image per queued texel: see profiles/r06_gs_plan.md)


def test_adam_update():
    from impersonator_amd import ops
    g = torch.Generator().manual_seed(13)
    p = torch.randn(1000, generator=g)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=2e-4, betas=(0.5, 0.999))
    pg, m, v = p.cuda(), torch.zeros(1000, device="cuda"), torch.zeros(1000, device="cuda")
    for step in range(1, 4):
        grad = torch.randn(1000, generator=g)
        ref.grad = grad.clone()
        opt.step()
        ops.adam_update(pg, grad.cuda(), m, v, step, 2e-4, (0.5, 0.999))
    assert float((pg.cpu() - ref.detach()).abs().max()) < 1e-6


def test_grid_sample_nhwc_forward():
    from impersonator_amd import ops
    g = torch.Generator().manual_seed(14)
    x = torch.randn(3, 64, 20, 20, generator=g)
    grid = torch.rand(3, 12, 12, 2, generator=g) * 2.6 - 1.3
    ref = F.grid_sample(x, grid, align_corners=False)
    y = ops.grid_sample_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), grid.cuda())
    assert _rel(y.cpu().permute(0, 3, 1, 2), ref) < 1e-5
    y1 = ops.grid_sample_nhwc(x[:1].permute(0, 2, 3, 1).contiguous().cuda(), grid.cuda())
    assert _rel(y1.cpu().permute(0, 3, 1, 2), F.grid_sample(x[:1].expand(3, -1, -1, -1), grid, align_corners=False)) < 1e-5


_WGRAD_PROBE = r"""
import sys, torch
import torch.nn.functional as F
from impersonator_amd import ops
cin, cout, H, W, N = (int(v) for v in sys.argv[1:6])
g = torch.Generator().manual_seed(5)
x = torch.randn(N, cin, H, W, generator=g)
w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
wr = w.clone().double().requires_grad_(True)
y = F.conv2d(x.double(), wr, None, stride=1, padding=1)
dy = torch.randn(y.shape, generator=g)
y.backward(dy.double())
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
dw = ops.conv2d_backward_weight(nhwc(x), nhwc(dy), tuple(w.shape), 1, 1, False, precision="bf16x3")
print("REL %.3e" % (float((dw.cpu().double() - wr.grad).abs().max()) / float(wr.grad.abs().max())))
"""


@pytest.mark.parametrize("slices", ["1", "2", "16"])
@pytest.mark.parametrize("shape", [(128, 256, 8, 96, 2), (128, 64, 16, 32, 1), (64, 128, 4, 64, 3), (64, 64, 12, 32, 1)],
                         ids=["128x128 tile W=96", "64x128 tile", "128x64 tile", "64x64 tile"])
def test_kernel_row_weight_gradient_paths(shape, slices):
    """wgrad_row3_bf16x3_kernel away from what its launcher would pick: ONE slice (the gradient written directly in PyTorch's
    layout, no partials), two, and sixteen (two- or three-step slices: the short pipeline without the steady-state loop); a width
    that is not a power of two (three 32-pixel steps per row), rectangular maps, all four tile shapes.  LWG_WGRAD_ROW3_SLICES is
    read once per process, hence the subprocess.  Against float64 autograd, the bound of test_weight_gradient_bf16x3."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cin, cout, H, W, N = shape
    env = dict(os.environ, LWG_WGRAD_ROW3_SLICES=slices, PYTHONPATH=root)
    p = subprocess.run([sys.executable, "-c", _WGRAD_PROBE] + [str(v) for v in (cin, cout, H, W, N)], env=env, cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    rel = float(p.stdout.strip().split("REL")[-1])
    assert rel < 3e-5, (shape, slices, rel)
