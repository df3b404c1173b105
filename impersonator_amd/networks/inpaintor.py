"""InpaintSANet (background inpaintor) on MI355X: the reference's module surface, liblwg underneath.

Mirrors networks/inpaintor.py:110-202 of the reference: same constructor, the same 322 `state_dict` entries (so the
reference's `deepfillv2` checkpoints load), `forward(imgs, masks, only_out=False, only_x=False)`.  The modules below
only hold parameters and BatchNorm running statistics; the arithmetic runs in inpaint.hip (eval mode only: this is an
inference component, run once per source image at models/imitator.py:124-125)."""
import ctypes

import torch
import torch.nn as nn

from .. import _lib


class _Conv(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.kaiming_normal_(self.weight)     # inpaintor.py:30-32


class GatedConv2dWithActivation(nn.Module):
    """inpaintor.py:12-48 (parameters only)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, activation=True):
        super().__init__()
        self.conv2d = _Conv(in_channels, out_channels, kernel_size)
        self.mask_conv2d = _Conv(in_channels, out_channels, kernel_size)
        self.batch_norm2d = nn.BatchNorm2d(out_channels)
        self.spec = (in_channels, out_channels, kernel_size, stride, dilation, activation)


class GatedDeConv2dWithActivation(nn.Module):
    """inpaintor.py:51-68: nearest x2 up-sampling followed by a gated conv (parameters only)."""

    def __init__(self, scale_factor, in_channels, out_channels, kernel_size, stride=1, dilation=1, activation=True):
        super().__init__()
        self.conv2d = GatedConv2dWithActivation(in_channels, out_channels, kernel_size, stride, dilation, activation)
        self.scale_factor = scale_factor


class SelfAttention(nn.Module):
    """inpaintor.py:71-107 (parameters only)."""

    def __init__(self, in_dim):
        super().__init__()
        self.query_conv = _Conv(in_dim, in_dim // 8, 1)
        self.key_conv = _Conv(in_dim, in_dim // 8, 1)
        self.value_conv = _Conv(in_dim, in_dim, 1)
        self.gamma = nn.Parameter(torch.zeros(1))


def _gated_stack(specs):
    mods = []
    for cin, cout, k, s, d, up, act in specs:
        mods.append(GatedDeConv2dWithActivation(2, cin, cout, k, s, d, bool(act)) if up
                    else GatedConv2dWithActivation(cin, cout, k, s, d, bool(act)))
    return nn.Sequential(*mods)


class InpaintSANet(nn.Module):
    def __init__(self, c_dim=5, image_size=256):
        super().__init__()
        c = 32
        self.c_dim, self.image_size = c_dim, image_size
        d4 = [(4 * c, 4 * c, 3, 1, d, 0, 1) for d in (2, 4, 8, 16)]
        self.coarse_net = _gated_stack(
            [(c_dim, c, 5, 1, 1, 0, 1), (c, 2 * c, 4, 2, 1, 0, 1), (2 * c, 2 * c, 3, 1, 1, 0, 1),
             (2 * c, 4 * c, 4, 2, 1, 0, 1), (4 * c, 4 * c, 3, 1, 1, 0, 1), (4 * c, 4 * c, 3, 1, 1, 0, 1)] + d4 +
            [(4 * c, 4 * c, 3, 1, 1, 0, 1), (4 * c, 4 * c, 3, 1, 1, 0, 1), (4 * c, 2 * c, 3, 1, 1, 1, 1),
             (2 * c, 2 * c, 3, 1, 1, 0, 1), (2 * c, c, 3, 1, 1, 1, 1), (c, c // 2, 3, 1, 1, 0, 1),
             (c // 2, 3, 3, 1, 1, 0, 0)])
        self.refine_conv_net = _gated_stack(
            [(c_dim, c, 5, 1, 1, 0, 1), (c, c, 4, 2, 1, 0, 1), (c, 2 * c, 3, 1, 1, 0, 1), (2 * c, 2 * c, 4, 2, 1, 0, 1),
             (2 * c, 4 * c, 3, 1, 1, 0, 1), (4 * c, 4 * c, 3, 1, 1, 0, 1), (4 * c, 4 * c, 3, 1, 1, 0, 1)] + d4)
        self.refine_attn = SelfAttention(4 * c)
        self.refine_upsample_net = _gated_stack(
            [(4 * c, 4 * c, 3, 1, 1, 0, 1), (4 * c, 4 * c, 3, 1, 1, 0, 1), (4 * c, 2 * c, 3, 1, 1, 1, 1),
             (2 * c, 2 * c, 3, 1, 1, 0, 1), (2 * c, c, 3, 1, 1, 1, 1), (c, c // 2, 3, 1, 1, 0, 1),
             (c // 2, 3, 3, 1, 1, 0, 0)])
        self._handle = None
        self._uploaded = None
        # conv arithmetic (extension, as ImpersonatorGenerator.precision): "bf16x3" = the split-operand MFMA kernels for the gated
        # convs with >= 32 input channels, "fp32" = exact fp32 MFMA everywhere.  Env LWG_INPAINT_PRECISION overrides.
        import os
        self.precision = os.environ.get("LWG_INPAINT_PRECISION", "bf16x3")

    def _version(self):
        return tuple(t._version for t in list(self.parameters()) + list(self.buffers()))

    def _ensure_handle(self):
        lib = _lib.load()
        if self._handle is None:
            h = ctypes.c_void_p()
            _lib.check(lib.lwg_inpaint_create(ctypes.byref(h), self.c_dim, self.image_size))
            self._handle = h
            self._uploaded = None
        ver = self._version()
        if self._uploaded != ver:
            for key, val in self.state_dict().items():
                if key.endswith("num_batches_tracked"):
                    continue
                arr = val.detach().to("cpu", torch.float32).contiguous()
                shape = (ctypes.c_int64 * max(arr.dim(), 1))(*arr.shape)
                _lib.check(lib.lwg_inpaint_load_weight(self._handle, key.encode(), ctypes.c_void_p(arr.data_ptr()),
                                                       shape, arr.dim()))
            missing = lib.lwg_inpaint_missing_weights(self._handle)
            if missing:
                raise _lib.LwgError(-5, "%d inpaintor weights missing after upload" % missing)
            self._uploaded = ver
        if self.precision not in ("fp32", "bf16x3"):
            raise ValueError("InpaintSANet.precision must be 'fp32' or 'bf16x3'")
        _lib.check(lib.lwg_inpaint_set_precision(self._handle, 1 if self.precision == "bf16x3" else 0))
        return self._handle

    def release(self):
        if self._handle is not None:
            _lib.load().lwg_inpaint_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    @torch.no_grad()
    def forward(self, imgs, masks, only_out=False, only_x=False):
        """inpaintor.py:178-202 -> (coarse_x, x, comp_imgs) | comp_imgs (only_out) | x (only_x)."""
        if self.training:
            raise RuntimeError("InpaintSANet runs in eval mode only (folded BatchNorm statistics); call .eval()")
        if not imgs.is_cuda:
            raise RuntimeError("impersonator_amd runs on the GPU only (got a %s tensor); there is no CPU fallback" % imgs.device)
        if imgs.shape[0] != 1:
            raise ValueError("the inpaintor runs once per source image (batch 1), as models/imitator.py:125 does")
        h = self._ensure_handle()
        imgs = imgs.float().contiguous()
        masks = masks.float().contiguous()
        coarse, x, comp = torch.empty_like(imgs), torch.empty_like(imgs), torch.empty_like(imgs)
        _lib.check(_lib.load().lwg_inpaint_forward(h, _lib.ptr(imgs), _lib.ptr(masks), _lib.ptr(coarse), _lib.ptr(x),
                                                   _lib.ptr(comp), _lib.stream_ptr()))
        if only_out:
            return comp
        if only_x:
            return x
        return coarse, x, comp
