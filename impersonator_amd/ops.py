"""Op-level convolution and its gradients on MI355X (liblwg, train.hip): thin tensor wrappers over
lwg_conv2d_{forward, backward_data, backward_weight}.  Activations are NHWC fp32 CUDA tensors, weights keep PyTorch's
layouts ((Cout,Cin,k,k); (Cin,Cout,3,3) for the transposed conv).  Building blocks of the generator-side training step
(SURVEY.md 8f row 4); the inference path does not go through here."""
import ctypes

import torch

from . import _lib


class _Desc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("N", "H", "W", "Cin", "Cout", "k", "stride", "pad", "transposed", "precision")]


PRECISIONS = {"fp32": 0, "bf16x3": 1}


def _desc(x_shape, cin, cout, k, stride, pad, transposed, precision="fp32"):
    n, h, w, _ = x_shape
    return _Desc(n, h, w, cin, cout, k, stride, pad, int(transposed), PRECISIONS[precision])


def _ws(d, dev):
    nbytes = _lib.load().lwg_conv2d_workspace_bytes(ctypes.byref(d))
    if not nbytes:
        raise _lib.LwgError(-2, _lib.load().lwg_last_error().decode(errors="replace"))
    return torch.empty(nbytes // 4 + 1, device=dev, dtype=torch.float32), nbytes


def _out_hw(h, w, k, stride, pad, transposed):
    return (2 * h, 2 * w) if transposed else ((h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1)


def _chk(*ts):
    for t in ts:
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("lwg ops take contiguous float32 CUDA tensors (no CPU fallback)")


@torch.no_grad()
def conv2d_forward(x, w, bias=None, stride=1, pad=0, transposed=False, precision="fp32"):
    """x (N,H,W,Cin) -> (N,Ho,Wo,Cout); F.conv2d / F.conv_transpose2d(stride 2, padding 1, output_padding 1).
    precision 'bf16x3': the inference path's split-bf16 kernel where the layer fits it (include/lwg.h)."""
    _chk(x, w, bias)
    cin, cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    d = _desc(x.shape, cin, cout, w.shape[2], stride, pad, transposed, precision)
    ho, wo = _out_hw(x.shape[1], x.shape[2], w.shape[2], stride, pad, transposed)
    y = torch.empty((x.shape[0], ho, wo, cout), device=x.device, dtype=torch.float32)
    ws, nb = _ws(d, x.device)
    _lib.check(_lib.load().lwg_conv2d_forward(ctypes.byref(d), _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(y),
                                              _lib.ptr(ws), nb, _lib.stream_ptr()))
    return y


@torch.no_grad()
def conv2d_backward_data(dy, w, x_shape, stride=1, pad=0, transposed=False, precision="fp32"):
    """Gradient wrt the input: dy (N,Ho,Wo,Cout) -> dx of shape x_shape (N,H,W,Cin)."""
    _chk(dy, w)
    cin, cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    d = _desc(x_shape, cin, cout, w.shape[2], stride, pad, transposed, precision)
    dx = torch.empty(tuple(x_shape), device=dy.device, dtype=torch.float32)
    ws, nb = _ws(d, dy.device)
    _lib.check(_lib.load().lwg_conv2d_backward_data(ctypes.byref(d), _lib.ptr(dy), _lib.ptr(w), _lib.ptr(dx), _lib.ptr(ws), nb,
                                                    _lib.stream_ptr()))
    return dx


@torch.no_grad()
def _out(out, shape, device):
    """`out` when it is a usable destination (contiguous fp32 tensor of `shape` on `device`), else a fresh tensor."""
    if out is None:
        return torch.empty(tuple(shape), device=device, dtype=torch.float32)
    if tuple(out.shape) != tuple(shape) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != device:
        raise ValueError("out= must be a contiguous fp32 tensor of shape %s on %s" % (tuple(shape), device))
    return out


@torch.no_grad()
def conv2d_backward_weight(x, dy, w_shape, stride=1, pad=0, transposed=False, with_bias=False, precision="fp32", out=None):
    """Gradient wrt the weight (PyTorch layout `w_shape`) and, optionally, the bias.  `out`: write dw there (e.g. the
    parameter's slice of a flat gradient buffer) instead of into a fresh tensor."""
    _chk(x, dy)
    cin, cout = (w_shape[0], w_shape[1]) if transposed else (w_shape[1], w_shape[0])
    d = _desc(x.shape, cin, cout, w_shape[2], stride, pad, transposed, precision)
    dw = _out(out, w_shape, x.device)
    db = torch.empty(cout, device=x.device, dtype=torch.float32) if with_bias else None
    ws, nb = _ws(d, x.device)
    _lib.check(_lib.load().lwg_conv2d_backward_weight(ctypes.byref(d), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(db),
                                                      _lib.ptr(ws), nb, _lib.stream_ptr()))
    return (dw, db) if with_bias else dw


def _heads_ws(n, h, w, dev):
    nbytes = _lib.load().lwg_heads_workspace_bytes(n, h, w)
    return torch.empty(nbytes // 4 + 1, device=dev, dtype=torch.float32), nbytes


@torch.no_grad()
def heads_forward(x, w):
    """Regression heads: x (N,H,W,64), w (>=4,64,7,7) -> (tanh(conv rows 0-2) (N,3,H,W), sigmoid(conv row 3) (N,1,H,W)).
    PRECONDITION x >= 0 (post-ReLU activations, what the generator feeds its heads): the kernel is the inference path's,
    which folds the preceding ReLU into its operand load -- negative entries are read as 0 (include/lwg.h)."""
    _chk(x, w)
    n, h, wd, c = x.shape
    if c != 64 or tuple(w.shape[1:]) != (64, 7, 7) or w.shape[0] < 4:
        raise RuntimeError("heads_forward: x (N,H,W,64) and w (>=4,64,7,7) expected")
    color = torch.empty((n, 3, h, wd), device=x.device, dtype=torch.float32)
    mask = torch.empty((n, 1, h, wd), device=x.device, dtype=torch.float32)
    ws, nb = _heads_ws(n, h, wd, x.device)
    _lib.check(_lib.load().lwg_heads_forward(_lib.ptr(x), n, h, wd, _lib.ptr(w), int(w.shape[0]), _lib.ptr(color), _lib.ptr(mask),
                                             _lib.ptr(ws), nb, _lib.stream_ptr()))
    return color, mask


@torch.no_grad()
def heads_backward_weight(x, dy8, out=None):
    """x (N,H,W,64), dy8 (N,H,W,8) (gradient wrt the pre-activation head outputs, channels 4-7 zero) -> dw (8,64,7,7)."""
    _chk(x, dy8)
    n, h, wd, c = x.shape
    if c != 64 or tuple(dy8.shape) != (n, h, wd, 8):
        raise RuntimeError("heads_backward_weight: x (N,H,W,64) and dy8 (N,H,W,8) expected")
    dw = _out(out, (8, 64, 7, 7), x.device)
    ws, nb = _heads_ws(n, h, wd, x.device)
    _lib.check(_lib.load().lwg_heads_backward_weight(_lib.ptr(x), _lib.ptr(dy8), n, h, wd, _lib.ptr(dw), _lib.ptr(ws), nb,
                                                     _lib.stream_ptr()))
    return dw


@torch.no_grad()
def instance_norm_forward(x, gamma, beta, relu=False):
    """x (N,H,W,C) -> (y, stats): F.instance_norm(weight, bias, eps=1e-5) [+ ReLU]; stats (N,C,2) = (mean, rstd)."""
    _chk(x, gamma, beta)
    n, h, w, c = x.shape
    y = torch.empty_like(x)
    stats = torch.empty((n, c, 2), device=x.device, dtype=torch.float32)
    scratch = torch.empty(_lib.load().lwg_instance_norm_scratch_bytes(n, h * w, c) // 8 + 1, device=x.device, dtype=torch.float64)
    _lib.check(_lib.load().lwg_instance_norm_forward(_lib.ptr(x), n, h * w, c, _lib.ptr(gamma), _lib.ptr(beta), int(relu),
                                                     _lib.ptr(y), _lib.ptr(stats), _lib.ptr(scratch), _lib.stream_ptr()))
    return y, stats


@torch.no_grad()
def instance_norm_backward(x, y, dy, stats, gamma, out_dgamma=None, out_dbeta=None):
    """-> (dx, dgamma, dbeta); pass y (the forward output) when the forward applied the ReLU, else None."""
    _chk(x, y, dy, stats, gamma)
    n, h, w, c = x.shape
    dx = torch.empty_like(x)
    dgamma = _out(out_dgamma, (c,), x.device)
    dbeta = _out(out_dbeta, (c,), x.device)
    scratch = torch.empty(_lib.load().lwg_instance_norm_scratch_bytes(n, h * w, c) // 8 + 1, device=x.device, dtype=torch.float64)
    _lib.check(_lib.load().lwg_instance_norm_backward(_lib.ptr(x), _lib.ptr(y), _lib.ptr(dy), _lib.ptr(stats), _lib.ptr(gamma), n,
                                                      h * w, c, _lib.ptr(dx), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(scratch),
                                                      _lib.stream_ptr()))
    return dx, dgamma, dbeta


class GridSamplePlan(object):
    """The scatter of one flow field turned into per-texel contribution lists in a fixed order (lwg_grid_sample_plan): built once
    per (grid, source shape), applied to any number of gradient tensors of that shape (any channel count)."""

    def __init__(self, grid, x_shape, align_corners=False):
        _chk(grid)
        lib = _lib.load()
        self.xn, self.h, self.w = int(x_shape[0]), int(x_shape[1]), int(x_shape[2])
        self.n, self.ho, self.wo = int(grid.shape[0]), int(grid.shape[1]), int(grid.shape[2])
        nbytes = lib.lwg_grid_sample_plan_bytes(self.xn, self.h, self.w, self.n, self.ho, self.wo)
        if not nbytes:
            raise ValueError("grid_sample plan: bad dimensions %s for a grid of %s" % (tuple(x_shape), tuple(grid.shape)))
        self.buf = torch.empty(nbytes // 4, device=grid.device, dtype=torch.int32)
        self.nbytes = nbytes
        _lib.check(lib.lwg_grid_sample_plan(_lib.ptr(grid), self.xn, self.h, self.w, self.n, self.ho, self.wo, int(align_corners),
                                            _lib.ptr(self.buf), nbytes, _lib.stream_ptr()))

    def matches(self, x_shape, dy_shape):
        return (int(x_shape[0]), int(x_shape[1]), int(x_shape[2])) == (self.xn, self.h, self.w) and \
            tuple(int(v) for v in dy_shape[:3]) == (self.n, self.ho, self.wo)


@torch.no_grad()
def grid_sample_backward(dy, grid, x_shape, align_corners=False, plan=None, deterministic=True):
    """Gradient of F.grid_sample(x, grid) (bilinear, zeros) wrt x: dy (n,Ho,Wo,C) NHWC, grid (n,Ho,Wo,2) -> dx of NHWC
    shape x_shape (xn,H,W,C), xn in {1, n} (1: one source shared by the batch, gradients summed).  Deterministic (default): a
    gather over a GridSamplePlan (pass `plan` to share one between the warps of a pyramid level), contributions added in a fixed
    order -- bit-reproducible; deterministic=False: one scatter kernel with float atomics (what torch does)."""
    _chk(dy, grid)
    xn, h, w, c = x_shape
    n, ho, wo, _ = dy.shape
    dx = torch.zeros(tuple(x_shape), device=dy.device, dtype=torch.float32)
    if not deterministic:
        _lib.check(_lib.load().lwg_grid_sample_backward(_lib.ptr(dy), _lib.ptr(grid), xn, c, h, w, n, ho, wo, int(align_corners),
                                                        _lib.ptr(dx), _lib.stream_ptr()))
        return dx
    if plan is None:
        plan = GridSamplePlan(grid, x_shape, align_corners)
    elif not plan.matches(x_shape, dy.shape):
        raise ValueError("grid_sample_backward: the plan was built for other dimensions")
    _lib.check(_lib.load().lwg_grid_sample_backward_planned(_lib.ptr(dy), c, xn, h, w, n, ho, wo, _lib.ptr(plan.buf), plan.nbytes,
                                                            _lib.ptr(dx), _lib.stream_ptr()))
    return dx


@torch.no_grad()
def adam_update(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8):
    """In-place torch.optim.Adam step on flat (contiguous) tensors."""
    _chk(param, grad, exp_avg, exp_avg_sq)
    _lib.check(_lib.load().lwg_adam_update(_lib.ptr(param), _lib.ptr(grad), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq), param.numel(),
                                           int(step), float(lr), float(betas[0]), float(betas[1]), float(eps), _lib.stream_ptr()))


def adam_step_state(step=0, betas=(0.9, 0.999), device="cuda"):
    """The device-resident step state of adam_update_device_step after `step` steps: int64 tensor of three words
    [t, bits of float64(beta1^t), bits of float64(beta2^t)] (include/lwg.h)."""
    import numpy as np
    pows = np.array([float(betas[0]) ** int(step), float(betas[1]) ** int(step)], dtype=np.float64).view(np.int64)
    return torch.tensor([int(step), int(pows[0]), int(pows[1])], dtype=torch.int64, device=device)


def adam_update_device_step(param, grad, exp_avg, exp_avg_sq, step_state, lr, betas=(0.9, 0.999), eps=1e-8):
    """adam_update with the step count in `step_state` (adam_step_state(); advanced by the call on the stream): the form that
    can be captured in a graph and replayed (include/lwg.h, lwg_adam_update_device_step).  `int(step_state[0])` is the count."""
    _chk(param, grad, exp_avg, exp_avg_sq)
    if not step_state.is_cuda or step_state.dtype != torch.int64 or step_state.numel() != 3 or not step_state.is_contiguous():
        raise RuntimeError("adam_update_device_step: step_state must be adam_step_state(): three int64 words on the device")
    _lib.check(_lib.load().lwg_adam_update_device_step(_lib.ptr(param), _lib.ptr(grad), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq),
                                                       param.numel(), _lib.ptr(step_state), float(lr), float(betas[0]), float(betas[1]),
                                                       float(eps), _lib.stream_ptr()))


@torch.no_grad()
def grid_sample_nhwc(x, grid, align_corners=False):
    """F.grid_sample (bilinear, zeros) on NHWC tensors: x (xn,H,W,C) with xn in {1, n}, grid (n,Ho,Wo,2) -> (n,Ho,Wo,C)."""
    _chk(x, grid)
    xn, h, w, c = x.shape
    n, ho, wo, _ = grid.shape
    y = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().lwg_grid_sample_nhwc(_lib.ptr(x), xn, c, h, w, _lib.ptr(grid), n, ho, wo, int(align_corners), _lib.ptr(y),
                                                _lib.stream_ptr()))
    return y
