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
]


def _rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)


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
