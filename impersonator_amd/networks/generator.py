"""ImpersonatorGenerator on MI355X: the reference's module surface, liblwg underneath.

Mirrors `networks/generator.py` of the reference (ResidualBlock :8-20, ResNetGenerator :23-65,
ResUnetGenerator :68-184, ImpersonatorGenerator :187-320):

  * same constructor arguments, same `state_dict` keys and shapes (including the `attetion_reg`
    spelling, generator.py:134), so reference checkpoints load with `load_state_dict`;
  * same method names / argument order / return structure for the inference path:
    `encode_src`, `inference`, `swap`, `transform`, `stn`, `resize_trans`;
  * NO torch compute: the nn.Modules below only hold parameters.  Every method packs pointers and
    calls the C ABI (include/lwg.h); a missing library or a CPU tensor raises.

Feature maps returned by `encode_src` are NCHW-shaped tensors in `torch.channels_last` memory format:
numerically what the reference returns, physically the NHWC layout the HIP kernels consume.
"""
import ctypes

import torch
import torch.nn as nn

from .. import _lib
from .networks import NetworkBase

N_DOWN = 3


class _ConvParams(nn.Module):
    """Parameter holder named like nn.Conv2d / nn.ConvTranspose2d (weight only: bias=False everywhere)."""

    def __init__(self, cin, cout, k, transposed=False):
        super().__init__()
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        self.weight = nn.Parameter(torch.empty(shape))
        self.transposed = transposed
        nn.init.normal_(self.weight, 0.0, 0.02)  # NetworkBase.init_weights (networks/networks.py:54-65)

    def forward(self, *_):
        raise RuntimeError("parameter holder: computation runs in liblwg (impersonator_amd/csrc)")


class _InstanceNormParams(nn.Module):
    """Parameter holder named like nn.InstanceNorm2d(affine=True): weight (gamma) and bias (beta)."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))

    def forward(self, *_):
        raise RuntimeError("parameter holder: computation runs in liblwg (impersonator_amd/csrc)")


def _seq(*mods):
    return nn.Sequential(*mods)


def _conv_norm(cin, cout, k, transposed=False):
    # index 0 = conv, 1 = InstanceNorm, 2 = ReLU placeholder (keeps the reference's Sequential indices)
    return _seq(_ConvParams(cin, cout, k, transposed), _InstanceNormParams(cout), nn.Identity())


class ResidualBlock(nn.Module):
    """generator.py:8-20: main = [conv3, IN, ReLU, conv3, IN]."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.main = _seq(_ConvParams(dim_in, dim_out, 3), _InstanceNormParams(dim_out), nn.Identity(),
                         _ConvParams(dim_out, dim_out, 3), _InstanceNormParams(dim_out))


class ResNetGenerator(NetworkBase):
    """BGNet (generator.py:23-65): holds the parameters; `forward` runs in liblwg through the ImpersonatorGenerator that
    owns it (Imitator.personalize uses it when --bg_model ORIGINAL, models/imitator.py:30-34,127-132)."""

    def __init__(self, conv_dim=64, c_dim=5, repeat_num=9, k_size=4, n_down=2):
        super().__init__()
        self._name = 'resnet_generator'
        layers = [_ConvParams(c_dim, conv_dim, 7), _InstanceNormParams(conv_dim), nn.Identity()]
        cur = conv_dim
        for _ in range(n_down):
            layers += [_ConvParams(cur, cur * 2, k_size), _InstanceNormParams(cur * 2), nn.Identity()]
            cur *= 2
        for _ in range(repeat_num):
            layers.append(ResidualBlock(cur, cur))
        for _ in range(n_down):
            layers += [_ConvParams(cur, cur // 2, k_size, transposed=True), _InstanceNormParams(cur // 2), nn.Identity()]
            cur //= 2
        layers += [_ConvParams(cur, 3, 7), nn.Identity()]
        self.model = _seq(*layers)
        self._owner = None

    def forward(self, x, c=None):
        """generator.py:60-65 (the domain-label argument `c` is never used on this path)."""
        if c is not None:
            raise NotImplementedError("ResNetGenerator.forward with a domain label")
        owner = self._owner() if self._owner is not None else None
        if owner is None:
            raise RuntimeError("this BGNet is not attached to an ImpersonatorGenerator")
        return owner.infer_bg(x)


class ResUnetGenerator(NetworkBase):
    """generator.py:68-134: encoders / resnets / decoders / skippers / img_reg / attetion_reg."""

    def __init__(self, conv_dim=64, c_dim=5, repeat_num=6, k_size=4, n_down=2):
        super().__init__()
        self._name = 'resunet_generator'
        self.repeat_num = repeat_num
        self.n_down = n_down
        enc = [_conv_norm(c_dim, conv_dim, 7)]
        cur = conv_dim
        for _ in range(n_down):
            enc.append(_conv_norm(cur, cur * 2, k_size))
            cur *= 2
        self.encoders = _seq(*enc)
        self.resnets = _seq(*[ResidualBlock(cur, cur) for _ in range(repeat_num)])
        dec, skp = [], []
        for _ in range(n_down):
            dec.append(_conv_norm(cur, cur // 2, k_size, transposed=True))
            skp.append(_conv_norm(cur, cur // 2, k_size))
            cur //= 2
        self.decoders = _seq(*dec)
        self.skippers = _seq(*skp)
        self.img_reg = _seq(_ConvParams(cur, 3, 7), nn.Identity())
        self.attetion_reg = _seq(_ConvParams(cur, 1, 7), nn.Identity())


class ImpersonatorGenerator(NetworkBase):
    """generator.py:187-320.  `max_batch` (extension) sizes the device scratch; frames of one source
    are independent, so any batch up to it runs in one launch sequence."""

    PRECISIONS = {"fp32": 0, "bf16x3": 1}

    def __init__(self, bg_dim, src_dim, tsf_dim, conv_dim=64, repeat_num=6, image_size=256, max_batch=8,
                 align_corners=False, precision=None):
        super().__init__()
        self._name = 'impersonator_generator'
        self.n_down = N_DOWN
        self.repeat_num = repeat_num
        self.conv_dim = conv_dim
        self.src_dim, self.tsf_dim = src_dim, tsf_dim
        self.image_size = image_size
        self.max_batch = max_batch
        # hazard H1: the reference calls F.grid_sample without align_corners (generator.py:313); torch 1.2
        # meant True, torch >= 1.3 means False.  False is the parity target; True serves 2019 checkpoints.
        self.align_corners = bool(align_corners)
        # conv arithmetic (extension): "bf16x3" = fp32 operands split into two bf16 terms, three MFMA products, fp32
        # accumulate (8e-5 L-inf on the final image, ~3x faster); "fp32" = exact fp32 MFMA, the reference's own arithmetic;
        # "auto" (default) = bf16x3 unless a probe pass in both arithmetics says the weights need fp32 (see `auto_probe`).
        # Env LWG_PRECISION overrides the default.
        import os
        self._policy, self._precision, self.auto_report = "auto", "bf16x3", None
        self.precision = precision or os.environ.get("LWG_PRECISION", "auto")
        self.bg_dim = bg_dim
        self.bg_model = ResNetGenerator(conv_dim=conv_dim, c_dim=bg_dim, repeat_num=repeat_num, k_size=3, n_down=N_DOWN)
        import weakref
        self.bg_model._owner = weakref.ref(self)
        self.src_model = ResUnetGenerator(conv_dim=conv_dim, c_dim=src_dim, repeat_num=repeat_num, k_size=3, n_down=N_DOWN)
        self.tsf_model = ResUnetGenerator(conv_dim=conv_dim, c_dim=tsf_dim, repeat_num=repeat_num, k_size=3, n_down=N_DOWN)
        self._handle = None
        self._uploaded_version = None

    # ------------------------------------------------------------------ conv arithmetic
    AUTO_BOUND = 2.5e-4   # bf16x3 serves a weight set if its probe image is within this of the exact-fp32 one (bound: 1e-3)

    @property
    def precision(self):
        """The arithmetic the per-frame stream runs in: "bf16x3" or "fp32" (under the "auto" policy: what the probe chose, bf16x3
        before any probe)."""
        return self._precision

    @precision.setter
    def precision(self, value):
        if value == "auto":
            self._policy, self.auto_report = "auto", None
            self._precision = "bf16x3"
            return
        if value not in self.PRECISIONS:
            raise ValueError("precision must be one of %s" % (sorted(self.PRECISIONS) + ["auto"]))
        self._policy = self._precision = value

    @property
    def precision_policy(self):
        return self._policy

    def auto_pending(self):
        """True while the "auto" policy has not probed the current weights (a probe is due at the next personalize)."""
        if self._policy != "auto":
            return False
        ver = self._weights_version() + (bool(self.align_corners), int(self.image_size))
        return self.auto_report is None or self.auto_report.get("weights") != ver

    @torch.no_grad()
    def auto_probe(self, src_encoder_outs, src_resnet_outs, tsf_inputs, T, bg_img):
        """Policy "auto": decides the arithmetic for the CURRENT weights, once per weight set.  The reference computes
        networks/generator.py:80-133 in fp32; bf16x3 differs from it by ~2^-17 per product, which random-init and trained weights
        turn into 1e-5..1e-4 on the image but extreme weight sets (conv weights 5x wider than the initialisation, InstanceNorm
        gains spread over two decades: tests/test_gpu_weight_sets.py) into 5e-4..8e-4.  So: one probe frame through BOTH
        arithmetics; bf16x3 is kept when image, colour and mask agree with the fp32 pass within AUTO_BOUND, else this weight set
        is served in fp32.  Costs one bf16x3 + one fp32 frame and a 4-byte read-back; nothing once the weights are known.
        Returns the report (also in self.auto_report)."""
        if not self.auto_pending():
            return self.auto_report
        ver = self._weights_version() + (bool(self.align_corners), int(self.image_size))
        lib, outs = _lib.load(), {}
        for mode in ("bf16x3", "fp32"):
            self._precision = mode
            outs[mode] = self.inference(src_encoder_outs, src_resnet_outs, tsf_inputs, T, bg_img=bg_img)
        worst = torch.empty(3, device=tsf_inputs.device, dtype=torch.float32)   # (lwg_max_abs_diff zeroes its word itself)
        for k, (a, b) in enumerate(zip(outs["bf16x3"], outs["fp32"])):
            _lib.check(lib.lwg_max_abs_diff(_lib.ptr(a), _lib.ptr(b), a.numel(), ctypes.c_void_p(worst.data_ptr() + 4 * k),
                                            _lib.stream_ptr()))
        linf = float(max(worst.tolist()))
        self._precision = "bf16x3" if linf <= self.AUTO_BOUND else "fp32"
        self.auto_report = {"weights": ver, "linf_bf16x3_vs_fp32": linf, "bound": self.AUTO_BOUND, "chosen": self._precision}
        return self.auto_report

    # ------------------------------------------------------------------ handle / weights
    def _weights_version(self):
        return tuple(p._version for p in self.parameters()) + (id(self),)

    def _ensure_handle(self, bs=1):
        lib = _lib.load()
        if bs > self.max_batch:
            self.release()
            self.max_batch = bs
        if self._handle is None:
            h = ctypes.c_void_p()
            _lib.check(lib.lwg_generator_create(ctypes.byref(h), self.src_dim, self.tsf_dim, self.conv_dim,
                                                self.repeat_num, self.image_size, self.max_batch))
            self._handle = h
            self._uploaded_version = None
            _lib.check(lib.lwg_generator_enable_bg(self._handle, self.bg_dim))
        _lib.check(lib.lwg_generator_set_precision(self._handle, self.PRECISIONS[self.precision]))
        ver = self._weights_version()
        if self._uploaded_version != ver:
            if self._uploaded_version is not None and torch.cuda.is_available():
                # lwg_generator_load_weight copies on the null stream, which does not order against PyTorch's
                # non-blocking lane / side streams: kernels of earlier batches may still be reading the old weights
                torch.cuda.synchronize()
            for key, val in self.state_dict().items():
                arr = val.detach().to("cpu", torch.float32).contiguous()
                shape = (ctypes.c_int64 * arr.dim())(*arr.shape)
                _lib.check(lib.lwg_generator_load_weight(self._handle, key.encode(), ctypes.c_void_p(arr.data_ptr()),
                                                         shape, arr.dim()))
            missing = lib.lwg_generator_missing_weights(self._handle)
            if missing:
                raise _lib.LwgError(-5, "%d generator weights missing after upload" % missing)
            self._uploaded_version = ver
        return self._handle

    def reserve(self, bs):
        """Sizes the device scratch for launch sequences of up to `bs` frames ahead of time (extension): growing inside a
        running pipeline would destroy and re-create the handle -- a device synchronisation and a weight upload -- in the
        middle of it (Imitator.predict_batches fuses consecutive batches into one launch sequence)."""
        if bs > self.max_batch:
            self.release()
            self.max_batch = int(bs)

    def replica(self):
        """A second engine over the *same* parameters (extension): own device handle, so own scratch and own copy of
        the re-laid-out weights.  Imitator.predict_batches runs consecutive batches on two of them, each on its own HIP
        stream, so that the launch gaps and workgroup tails of one batch's dependent kernel chain are filled by the
        other's.  Precision and align_corners follow the original at every use (see Imitator._lanes)."""
        r = ImpersonatorGenerator(self.bg_dim, self.src_dim, self.tsf_dim, conv_dim=self.conv_dim,
                                  repeat_num=self.repeat_num, image_size=self.image_size, max_batch=self.max_batch,
                                  align_corners=self.align_corners, precision=self.precision)   # (the resolved arithmetic)
        r.bg_model, r.src_model, r.tsf_model = self.bg_model, self.src_model, self.tsf_model   # shared Parameters
        return r

    def release(self):
        if self._handle is not None:
            if torch.cuda.is_available():
                torch.cuda.synchronize()   # kernels enqueued on any stream may still use the handle's scratch / weights
            _lib.load().lwg_generator_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    @staticmethod
    def _need_cuda(*tensors):
        for t in tensors:
            if t is not None and not t.is_cuda:
                raise RuntimeError("impersonator_amd runs on the GPU only (got a %s tensor); there is no CPU fallback"
                                   % t.device)

    def _feature_shapes(self):
        shapes = [(self.conv_dim << l, self.image_size >> l) for l in range(N_DOWN + 1)]
        shapes += [shapes[-1]] * self.repeat_num
        return shapes

    @staticmethod
    def _nhwc(t):
        """(N,C,H,W)-shaped tensor whose storage is NHWC; converts only if the caller handed plain NCHW."""
        t = t.float()
        if t.shape[1] > 1 and t.is_contiguous(memory_format=torch.channels_last):
            return t
        return t.contiguous(memory_format=torch.channels_last)

    def _input_layout(self, x):
        """1 when `x` is the NHWC8-backed view produced by SMPLRenderer.transfer, else 0 (plain NCHW)."""
        n, c, h, w = x.shape
        if x.stride() == (h * w * 8, 1, w * 8, 8) and x.storage_offset() % (h * w * 8) == 0 and c <= 8:
            return x, 1
        return x.float().contiguous(), 0

    # ------------------------------------------------------------------ reference API
    @torch.no_grad()
    def infer_bg(self, bg_inputs):
        """self.bg_model(bg_inputs) (generator.py:216-218, models/imitator.py:127-132): (bs, bg_dim, H, W) -> (bs, 3, H, W)."""
        self._need_cuda(bg_inputs)
        x = bg_inputs.float().contiguous()
        bs = x.shape[0]
        if x.shape[1:] != (self.bg_dim, self.image_size, self.image_size):
            raise ValueError("bg_inputs must be (bs, %d, %d, %d)" % (self.bg_dim, self.image_size, self.image_size))
        h = self._ensure_handle(bs)
        out = torch.empty((bs, 3, self.image_size, self.image_size), device=x.device, dtype=torch.float32)
        _lib.check(_lib.load().lwg_generator_bg_forward(h, _lib.ptr(x), bs, _lib.ptr(out), _lib.stream_ptr()))
        return out

    @torch.no_grad()
    def encode_src(self, src_inputs):
        """generator.py:213-214 -> (encoder_outs[4], resnet_outs[repeat_num])."""
        self._need_cuda(src_inputs)
        n = src_inputs.shape[0]
        h = self._ensure_handle(n)
        x = src_inputs.float().contiguous()
        # allocated channels_last outright (an NCHW empty tensor converted afterwards would cost a copy kernel each)
        feats = [torch.empty((n, c, s, s), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
                 for c, s in self._feature_shapes()]
        _lib.check(_lib.load().lwg_generator_encode_src_n(h, _lib.ptr(x), n, _lib.ptr_array(feats), _lib.stream_ptr()))
        return feats[:N_DOWN + 1], feats[N_DOWN + 1:]

    @torch.no_grad()
    def inference(self, src_encoder_outs, src_resnet_outs, tsf_inputs, T, bg_img=None):
        """generator.py:277-301 -> (tsf_img, tsf_mask).  With `bg_img` (extension) the blend of
        models/imitator.py:331 is fused into the last kernel and (pred, tsf_img, tsf_mask) is returned."""
        self._need_cuda(tsf_inputs, T)
        bs = tsf_inputs.shape[0]
        h = self._ensure_handle(bs)
        feats = [self._nhwc(f) for f in list(src_encoder_outs) + list(src_resnet_outs)]
        x, layout = self._input_layout(tsf_inputs)
        T = T.float().contiguous()
        dev = x.device
        s = self.image_size
        color = torch.empty((bs, 3, s, s), device=dev, dtype=torch.float32)
        mask = torch.empty((bs, 1, s, s), device=dev, dtype=torch.float32)
        pred = bg = None
        if bg_img is not None:
            bg = bg_img.float().contiguous()
            pred = torch.empty_like(color)
        feats_bs = feats[0].shape[0]     # 1: one source for the whole batch (Imitator); bs: a source per sample (infer_front)
        if feats_bs not in (1, bs) or any(f.shape[0] != feats_bs for f in feats):
            raise ValueError("source features must all have batch 1 or %d" % bs)
        _lib.check(_lib.load().lwg_generator_inference_n(
            h, _lib.ptr(x), layout, _lib.ptr(T), bs, _lib.ptr_array(feats), feats_bs, int(self.align_corners),
            _lib.ptr(color), _lib.ptr(mask), _lib.ptr(bg), 0 if bg is None else bg.shape[0], _lib.ptr(pred),
            _lib.stream_ptr()))
        return (color, mask) if bg_img is None else (pred, color, mask)

    @torch.no_grad()
    def swap(self, tsf_inputs, src_encoder_outs12, src_encoder_outs21, src_resnet_outs12, src_resnet_outs21, T12, T21,
             bg_img=None):
        """generator.py:245-275 -> (tsf_img, tsf_mask); with `bg_img` (extension) the blend of models/swapper.py:269
        is fused and (pred, tsf_img, tsf_mask) is returned."""
        self._need_cuda(tsf_inputs, T12, T21)
        bs = tsf_inputs.shape[0]
        h = self._ensure_handle(bs)
        f12 = [self._nhwc(f) for f in list(src_encoder_outs12) + list(src_resnet_outs12)]
        f21 = [self._nhwc(f) for f in list(src_encoder_outs21) + list(src_resnet_outs21)]
        x, layout = self._input_layout(tsf_inputs)
        T12, T21 = T12.float().contiguous(), T21.float().contiguous()
        s = self.image_size
        color = torch.empty((bs, 3, s, s), device=x.device, dtype=torch.float32)
        mask = torch.empty((bs, 1, s, s), device=x.device, dtype=torch.float32)
        pred = bg = None
        if bg_img is not None:
            bg = bg_img.float().contiguous()
            pred = torch.empty_like(color)
        _lib.check(_lib.load().lwg_generator_swap(h, _lib.ptr(x), layout, _lib.ptr(T12), _lib.ptr(T21), bs,
                                                  _lib.ptr_array(f12), _lib.ptr_array(f21), int(self.align_corners),
                                                  _lib.ptr(color), _lib.ptr(mask), _lib.ptr(bg),
                                                  0 if bg is None else bg.shape[0], _lib.ptr(pred), _lib.stream_ptr()))
        return (color, mask) if bg_img is None else (pred, color, mask)

    @torch.no_grad()
    def resize_trans(self, x, T):
        """generator.py:303-310: bilinear (align_corners=True) resize of the flow to x's resolution."""
        self._need_cuda(T)
        _, _, h, w = x.shape
        T = T.float().contiguous()
        bs, H, W, _ = T.shape
        out = torch.empty((bs, h, w, 2), device=T.device, dtype=torch.float32)
        _lib.check(_lib.load().lwg_resize_flow(_lib.ptr(T), bs, H, W, h, w, _lib.ptr(out), _lib.stream_ptr()))
        return out

    @torch.no_grad()
    def stn(self, x, T):
        """generator.py:312-315: F.grid_sample(x, T)."""
        self._need_cuda(x, T)
        x = x.float().contiguous()
        T = T.float().contiguous()
        n, ho, wo, _ = T.shape
        xn, c, h, w = x.shape
        out = torch.empty((n, c, ho, wo), device=x.device, dtype=torch.float32)
        _lib.check(_lib.load().lwg_grid_sample(_lib.ptr(x), xn, c, h, w, _lib.ptr(T), n, ho, wo,
                                               int(self.align_corners), _lib.ptr(out), _lib.stream_ptr()))
        return out

    def transform(self, x, T):
        """generator.py:317-320."""
        return self.stn(x, self.resize_trans(x, T))

    @torch.no_grad()
    def infer_front(self, src_inputs, tsf_inputs, T):
        """generator.py:216-243 -> (src_img, src_mask, tsf_img, tsf_mask): every sample has its own source.  The reference
        interleaves the two streams level by level; the same arithmetic runs here as source stream (features kept), tsf
        stream with per-sample Liquid-Warping-Block sources, then the source stream's own decoder and heads.
        Inference only (no autograd graph)."""
        self._need_cuda(src_inputs, tsf_inputs, T)
        if src_inputs.shape[0] != tsf_inputs.shape[0]:
            raise ValueError("src_inputs and tsf_inputs must have the same batch")
        enc, res = self.encode_src(src_inputs)
        tsf_img, tsf_mask = self.inference(enc, res, tsf_inputs, T)
        n, s = src_inputs.shape[0], self.image_size
        h = self._ensure_handle(n)
        src_img = torch.empty((n, 3, s, s), device=tsf_img.device, dtype=torch.float32)
        src_mask = torch.empty((n, 1, s, s), device=tsf_img.device, dtype=torch.float32)
        _lib.check(_lib.load().lwg_generator_decode_src(h, _lib.ptr_array(enc + res), n, _lib.ptr(src_img), _lib.ptr(src_mask),
                                                        _lib.stream_ptr()))
        return src_img, src_mask, tsf_img, tsf_mask

    @torch.no_grad()
    def forward(self, bg_inputs, src_inputs, tsf_inputs, T):
        """generator.py:204-211 -> (img_bg, src_img, src_mask, tsf_img, tsf_mask), inference only (the trainer's
        generator pass, models/impersonator_trainer.py:331-333, without an autograd graph)."""
        img_bg = self.bg_model(bg_inputs)
        src_img, src_mask, tsf_img, tsf_mask = self.infer_front(src_inputs, tsf_inputs, T)
        return img_bg, src_img, src_mask, tsf_img, tsf_mask

    def peek(self, which, shape):
        """Test hook (lwg_generator_peek): copy of an internal NHWC scratch buffer, shaped `shape`."""
        out = torch.empty(shape, device="cuda", dtype=torch.float32)
        _lib.check(_lib.load().lwg_generator_peek(self._ensure_handle(1), which, _lib.ptr(out), out.numel(),
                                                  _lib.stream_ptr()))
        return out

    # profiling hooks used by bench.py (roofline of the implicit-GEMM kernel)
    def profile(self, enable=True):
        _lib.check(_lib.load().lwg_generator_profile(self._ensure_handle(1), int(enable)))

    def profile_read(self, variant=-1):
        """(launches, total_ms, algorithmic_flops) of the bracketed conv launches: all kernels (variant=-1) or one."""
        n, ms, fl = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        _lib.check(_lib.load().lwg_generator_profile_read(self._ensure_handle(1), variant, ctypes.byref(n), ctypes.byref(ms),
                                                          ctypes.byref(fl)))
        return n.value, ms.value, fl.value

    def profile_table(self):
        """{kernel name as rocprofv3 prints it: (launches, total_ms, flops)} for every variant that ran."""
        lib = _lib.load()
        out = {}
        for v in range(lib.lwg_generator_profile_variants()):
            n, ms, fl = self.profile_read(v)
            if n:
                out[lib.lwg_generator_profile_variant_name(v).decode()] = (n, ms, fl)
        return out
