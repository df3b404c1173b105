"""Generator update of the training step on MI355X (second slice).

Replaces what torch autograd + cuDNN + torch.optim.Adam do for the generator in the reference's training iteration
(models/impersonator_trainer.py): `forward` (:329-348, bg_both=False), `_optimize_G` (:368-394), `loss_G.backward()`
and `self._optimizer_G.step()` (:355-357), for the loss terms that need no pretrained network -- adversarial (LSGAN,
target 0, through the HIP discriminator's input gradient), L1 reconstruction of the source, L1 on the transferred image
(the documented default without --use_vgg), mask MSE and mask total variation.

This is a hand-written backward pass: every layer of the three streams (BGNet, source ResUnet, transfer ResUnet with
the Liquid Warping Block) keeps what its gradient needs and runs the op-level HIP kernels of liblwg
(`impersonator_amd/ops.py`: conv / transposed-conv forward, data and weight gradients on the fp32 MFMA kernels,
InstanceNorm-affine forward/backward, grid_sample and its input gradient, Adam).  torch only carries tensors between
them (views, cat/split, a handful of elementwise expressions of the loss on 3- and 1-channel images).
Parameters, gradients and Adam moments live in flat buffers; the gradient buffer is what a data-parallel job averages
(`sharding.average_gradients`, RCCL) before `step()`."""
import torch

from .. import ops, sharding

N_DOWN = 3


def _nhwc(x, cpad=None):
    """NCHW -> contiguous NHWC, channels zero-padded to `cpad`."""
    y = x.permute(0, 2, 3, 1)
    if cpad is not None and cpad > x.shape[1]:
        y = torch.nn.functional.pad(y, (0, cpad - x.shape[1]))
    return y.contiguous()


class _ConvIN(object):
    """conv (no bias) -> InstanceNorm2d(affine) [-> ReLU]: the building block of generator.py:8-20, 80-133."""

    def __init__(self, tr, wkey, nkey, stride, pad, transposed=False, relu=True):
        self.tr, self.wkey, self.gkey, self.bkey = tr, wkey, nkey + ".weight", nkey + ".bias"
        self.stride, self.pad, self.transposed, self.relu = stride, pad, transposed, relu

    def forward(self, x):
        P = self.tr.P
        self.x = x
        self.raw = ops.conv2d_forward(x, P[self.wkey], None, self.stride, self.pad, self.transposed, self.tr.conv_precision)
        self.y, self.stats = ops.instance_norm_forward(self.raw, P[self.gkey], P[self.bkey], self.relu)
        return self.y

    def backward(self, dy, need_dx=True):
        P, G, tr = self.tr.P, self.tr.G, self.tr
        # a parameter's FIRST gradient contribution of an iteration is written straight into its slice of the flat gradient buffer
        # (GeneratorTrainer.first_write); a further one (none today: every layer runs once per iteration) would be added
        og, ob, ow = tr.first_write(self.gkey), tr.first_write(self.bkey), tr.first_write(self.wkey)
        draw, dg, db = ops.instance_norm_backward(self.raw, self.y if self.relu else None, dy.contiguous(), self.stats, P[self.gkey],
                                                  out_dgamma=og, out_dbeta=ob)
        if og is None:
            G[self.gkey].add_(dg)
        if ob is None:
            G[self.bkey].add_(db)
        dw = ops.conv2d_backward_weight(self.x, draw, tuple(P[self.wkey].shape), self.stride, self.pad, self.transposed,
                                        precision=self.tr.conv_precision, out=ow)
        if ow is None:
            G[self.wkey].add_(dw)
        tr.grads_checkpoint()     # the kernels writing this layer's three gradients are enqueued: complete buckets may go on the wire
        if not need_dx:
            return None
        return ops.conv2d_backward_data(draw, P[self.wkey], tuple(self.x.shape), self.stride, self.pad, self.transposed,
                                        self.tr.conv_precision)


class _Res(object):
    """ResidualBlock (generator.py:8-20): x + IN(conv(ReLU(IN(conv x))))."""

    def __init__(self, tr, prefix):
        self.a = _ConvIN(tr, prefix + ".main.0.weight", prefix + ".main.1", 1, 1, relu=True)
        self.b = _ConvIN(tr, prefix + ".main.3.weight", prefix + ".main.4", 1, 1, relu=False)

    def forward(self, x):
        return x + self.b.forward(self.a.forward(x))

    def backward(self, dy):
        return dy + self.a.backward(self.b.backward(dy))


HEAD_ROWS = 8   # device rows of a head's weight tensor: 3 colour + 1 mask (+ zero rows up to the kernels' granularity)


class _Head(object):
    """The 7x7 regression heads (generator.py:142-152) on their own kernels: tanh(conv) / sigmoid(conv) fused in the
    forward; the gradients take d(pre-activation) as an 8-channel NHWC tensor."""

    def __init__(self, tr, key):
        self.tr, self.key = tr, key

    def forward(self, x):
        """x (N,H,W,64) -> (img (N,H,W,3) = tanh, mask (N,H,W,1) = sigmoid), NHWC like every tensor of the trainer."""
        self.x = x
        color, mask = ops.heads_forward(x, self.tr.P[self.key])
        self.img = color.permute(0, 2, 3, 1).contiguous()
        self.mask = mask.permute(0, 2, 3, 1).contiguous()
        return self.img, self.mask

    def backward(self, d_img, d_mask=None):
        """gradients wrt the activated outputs -> dW accumulated, returns d x"""
        P, G = self.tr.P, self.tr.G
        d8 = torch.zeros(self.img.shape[:-1] + (HEAD_ROWS,), device=self.img.device, dtype=torch.float32)
        d8[..., 0:3] = d_img * (1 - self.img * self.img)
        if d_mask is not None:
            d8[..., 3:4] = d_mask * self.mask * (1 - self.mask)
        ow = self.tr.first_write(self.key)
        dw = ops.heads_backward_weight(self.x, d8, out=ow)
        if ow is None:
            G[self.key].add_(dw)
        self.tr.grads_checkpoint()
        return ops.conv2d_backward_data(d8, P[self.key], tuple(self.x.shape), 1, 3)


class _ResUnet(object):
    """ResUnetGenerator (generator.py:68-184) of stream `p` ('src_model' / 'tsf_model')."""

    def __init__(self, tr, p, repeat):
        self.tr, self.p = tr, p
        self.enc = [_ConvIN(tr, "%s.encoders.0.0.weight" % p, "%s.encoders.0.1" % p, 1, 3)]
        self.enc += [_ConvIN(tr, "%s.encoders.%d.0.weight" % (p, i), "%s.encoders.%d.1" % (p, i), 2, 1) for i in range(1, N_DOWN + 1)]
        self.res = [_Res(tr, "%s.resnets.%d" % (p, i)) for i in range(repeat)]
        self.dec = [_ConvIN(tr, "%s.decoders.%d.0.weight" % (p, i), "%s.decoders.%d.1" % (p, i), 2, 1, transposed=True)
                    for i in range(N_DOWN)]
        self.skip = [_ConvIN(tr, "%s.skippers.%d.0.weight" % (p, i), "%s.skippers.%d.1" % (p, i), 1, 1) for i in range(N_DOWN)]
        self.head = _Head(tr, "heads:" + p)

    def decode_regress(self, x, enc_outs):
        d = x
        self.cat_c = []
        for i in range(N_DOWN):
            d = self.dec[i].forward(d)
            skip = enc_outs[N_DOWN - 1 - i]
            self.cat_c.append(skip.shape[-1])
            d = self.skip[i].forward(torch.cat([skip, d], dim=-1))
        self.img, self.mask = self.head.forward(d)
        return self.img, self.mask

    def decode_regress_backward(self, d_img, d_mask):
        """-> (d trunk output, [d enc_outs[0..N_DOWN-1]])"""
        d = self.head.backward(d_img, d_mask)
        d_skips = [None] * N_DOWN
        for i in reversed(range(N_DOWN)):
            d_cat = self.skip[i].backward(d)
            c = self.cat_c[i]
            d_skips[N_DOWN - 1 - i] = d_cat[..., :c]
            d = self.dec[i].backward(d_cat[..., c:].contiguous())
        return d, d_skips


class GeneratorTrainer(object):
    """One Adam optimiser over the ImpersonatorGenerator's 194 parameter tensors with a hand-written backward pass."""

    def __init__(self, generator, discriminator, lambda_D_prob=1.0, lambda_rec=10.0, lambda_tsf=10.0, lambda_mask=0.1,
                 lambda_mask_smooth=1e-5, lr=0.0002, betas=(0.5, 0.999), eps=1e-8, conv_precision="fp32", mask_bce=False,
                 bg_both=False, vgg=None, face=None, lambda_face=1.0, use_vgg=None, use_style=False,
                 lambda_style=5.0):
        """conv_precision 'bf16x3': the convolutions of the three streams (forward, data gradient, weight gradient) run on
        split-bf16 operands (include/lwg.h, lwg_conv2d_desc.precision); norms, heads, losses, Adam: fp32."""
        if conv_precision not in ops.PRECISIONS:
            raise ValueError("conv_precision must be one of %s" % sorted(ops.PRECISIONS))
        self.conv_precision = conv_precision
        # impersonator_trainer.py:251-254 (--mask_bce: BCELoss on the masks), :333-339 (--bg_both: BGNet on the source's
        # and the target's background, 2N inputs), :256-260 + :376-377 (--use_vgg: `vgg` = networks.vgg.Vgg19Perceptual)
        # :268-273 + :383-385 (--use_face: `face` = networks.facenet.SphereFaceLoss; the batch then carries 'head_bbox')
        # `vgg` also serves :262-267 + :379-381 (--use_style: Gram-matrix term, use_style=True); use_vgg=False keeps the L1
        # transfer term while the network is only there for the style term
        self.mask_bce, self.bg_both, self.vgg, self.face = bool(mask_bce), bool(bg_both), vgg, face
        self.use_vgg = (vgg is not None) if use_vgg is None else bool(use_vgg)
        self.use_style, self.lambda_style, self.lambda_face = bool(use_style), lambda_style, lambda_face
        if (self.use_vgg or self.use_style) and vgg is None:
            raise ValueError("use_vgg / use_style need `vgg` (networks.vgg.Vgg19Perceptual)")
        self.generator, self.D = generator, discriminator
        self.lam = dict(adv=lambda_D_prob, rec=lambda_rec, tsf=lambda_tsf, mask=lambda_mask, smooth=lambda_mask_smooth)
        self.lr, self.betas, self.eps, self.t = lr, betas, eps, 0
        self.t_dev = None   # use_device_step(): Adam's step count on the device (graph replay)
        self.repeat = generator.repeat_num
        self.align = bool(generator.align_corners)
        dev = torch.device("cuda", torch.cuda.current_device())
        sd = {k: v.detach().float() for k, v in generator.state_dict().items()}
        # device layout of every trainable tensor: state_dict layout, except 7x7 stems (input channels padded to 8)
        # and the few-channel 7x7 heads (colour + mask combined, padded to HEAD_ROWS output rows)
        self.spec = []   # (device key, shape, [(state_dict key, row slice, in-channel count)])
        for k, v in sd.items():
            if k.endswith("img_reg.0.weight"):
                p = k[:-len(".img_reg.0.weight")]
                self.spec.append(("heads:" + p, (HEAD_ROWS, 64, 7, 7), [(k, slice(0, 3), 64), (p + ".attetion_reg.0.weight", slice(3, 4), 64)]))
            elif k.endswith("attetion_reg.0.weight"):
                continue
            elif k == "bg_model.model.%d.weight" % (3 + 3 * N_DOWN + self.repeat + 3 * N_DOWN):
                self.spec.append(("heads:bg", (HEAD_ROWS, 64, 7, 7), [(k, slice(0, 3), 64)]))
            elif v.dim() == 4 and v.shape[2] == 7:
                self.spec.append((k, (v.shape[0], 8, 7, 7), [(k, slice(0, v.shape[0]), v.shape[1])]))
            else:
                self.spec.append((k, tuple(v.shape), [(k, None, None)]))
        n = sum(int(torch.Size(s).numel()) for _, s, _ in self.spec)
        self.flat_p = torch.zeros(n, device=dev)
        self.flat_g = torch.zeros(n, device=dev)
        self.flat_m = torch.zeros(n, device=dev)
        self.flat_v = torch.zeros(n, device=dev)
        self._untouched = set()   # gradient slices nothing has been written to in the current backward pass (see first_write)
        self._ranges = []         # (key, lo, hi) element ranges of the flat buffers, in layout order (sharding.GradientBuckets)
        self._buckets = None
        self.P, self.G, off = {}, {}, 0
        for key, shape, parts in self.spec:
            cnt = int(torch.Size(shape).numel())
            self._ranges.append((key, off, off + cnt))
            self.P[key] = self.flat_p[off:off + cnt].view(shape)
            self.G[key] = self.flat_g[off:off + cnt].view(shape)
            off += cnt
            for sk, rows, cin in parts:
                src = sd[sk].to(dev)
                if rows is None:
                    self.P[key].copy_(src)
                else:
                    self.P[key][rows, :cin].copy_(src)
        self.bg_enc = [_ConvIN(self, "bg_model.model.0.weight", "bg_model.model.1", 1, 3)]
        self.bg_enc += [_ConvIN(self, "bg_model.model.%d.weight" % (3 + 3 * i), "bg_model.model.%d" % (4 + 3 * i), 2, 1)
                        for i in range(N_DOWN)]
        r0 = 3 + 3 * N_DOWN
        self.bg_res = [_Res(self, "bg_model.model.%d" % (r0 + i)) for i in range(self.repeat)]
        u0 = r0 + self.repeat
        self.bg_dec = [_ConvIN(self, "bg_model.model.%d.weight" % (u0 + 3 * i), "bg_model.model.%d" % (u0 + 3 * i + 1), 2, 1,
                               transposed=True) for i in range(N_DOWN)]
        self.bg_head = _Head(self, "heads:bg")
        self.src = _ResUnet(self, "src_model", self.repeat)
        self.tsf = _ResUnet(self, "tsf_model", self.repeat)

    def first_write(self, key):
        """The gradient slice of `key` if nothing has been written to it in this backward pass yet (then the kernel writes there
        directly: no temporary, no add launch), else None (the caller accumulates)."""
        if key in self._untouched:
            self._untouched.discard(key)
            if self._buckets is not None:
                self._buckets.written(key)
            return self.G[key]
        return None

    def grads_checkpoint(self):
        """Called by a layer's backward once the kernels that write its gradients are enqueued (see sharding.GradientBuckets)."""
        if self._buckets is not None:
            self._buckets.checkpoint()

    def _begin_buckets(self, all_reduce=True):
        """Data-parallel job: the gradient all-reduce runs bucket by bucket on a side stream while the backward pass goes on
        (env LWG_GRAD_BUCKETS=0: one blocking all-reduce of the whole buffer after it; LWG_BUCKET_MB: bucket size, default 64).
        `all_reduce=False` (a caller that wants this rank's raw gradients): no bucket is ever launched in this pass."""
        import os
        self._join_buckets()      # a previous pass whose step() was never called: its reductions are over before flat_g is rewritten
        if all_reduce and sharding.collectives_active() and os.environ.get("LWG_GRAD_BUCKETS", "1") != "0":
            mb = os.environ.get("LWG_BUCKET_MB", "64")
            if self._buckets is None or self._bucket_mb != mb:
                self._bucket_mb = mb
                self._buckets = sharding.GradientBuckets(self.flat_g, self._ranges,
                                                         int(float(os.environ.get("LWG_BUCKET_MB", "64")) * (1 << 20)))
            self._buckets.begin()
        else:
            self._buckets = None

    def _join_buckets(self):
        """Every bucket of the current pass averaged, and the compute stream behind the side stream: whoever reads `flat_g` next
        (Adam, gradients(), a new backward pass) sees either all of it averaged or -- without buckets -- none of it."""
        if self._buckets is not None and not self._buckets.joined:
            self._buckets.finish()

    # ------------------------------------------------------------------ parameters
    def state_dict(self):
        """Current parameters in the generator's state_dict layout (CPU tensors)."""
        out = {}
        for key, _, parts in self.spec:
            for sk, rows, cin in parts:
                out[sk] = (self.P[key] if rows is None else self.P[key][rows, :cin]).detach().cpu().clone()
        return out

    def gradients(self):
        self._join_buckets()
        out = {}
        for key, _, parts in self.spec:
            for sk, rows, cin in parts:
                out[sk] = (self.G[key] if rows is None else self.G[key][rows, :cin]).detach().cpu().clone()
        return out

    # ------------------------------------------------------------------ forward (impersonator_trainer.py:329-348)
    def _T_levels(self, T):
        return [None] + [self.generator.resize_trans(torch.empty(1, 1, T.shape[1] >> l, T.shape[2] >> l), T) for l in range(1, N_DOWN + 1)]

    @torch.no_grad()
    def forward(self, batch):
        b = {k: (v.cuda().float() if k != "head_bbox" else v) for k, v in batch.items()}
        self.b = b
        # --- BGNet
        x = _nhwc(b["input_G_bg"], 8)
        for m in self.bg_enc + self.bg_res + self.bg_dec:
            x = m.forward(x)
        self.bg_img = self.bg_head.forward(x)[0]
        # --- source stream (features kept for the Liquid Warping Block)
        x = _nhwc(b["input_G_src"], 8)
        self.src_enc = []
        for m in self.src.enc:
            x = m.forward(x)
            self.src_enc.append(x)
        self.src_res = []
        for m in self.src.res:
            x = m.forward(x)
            self.src_res.append(x)
        src_img, src_mask = self.src.decode_regress(x, self.src_enc)
        # --- transfer stream (generator.py:216-243)
        Ts = self._T_levels(b["T"].contiguous())
        self.Ts = Ts
        x = self.tsf.enc[0].forward(_nhwc(b["input_G_tsf"], 8))
        tsf_enc = [x]
        for i in range(1, N_DOWN + 1):
            x = self.tsf.enc[i].forward(x) + ops.grid_sample_nhwc(self.src_enc[i], Ts[i], self.align)
            tsf_enc.append(x)
        for i in range(self.repeat):
            x = self.tsf.res[i].forward(x) + ops.grid_sample_nhwc(self.src_res[i], Ts[N_DOWN], self.align)
        tsf_img, tsf_mask = self.tsf.decode_regress(x, tsf_enc)
        # --- blends (bg_both=False: one background, from the source's inputs)
        n = src_img.shape[0]
        if self.bg_img.shape[0] != (2 * n if self.bg_both else n):
            raise ValueError("input_G_bg carries %d images for a batch of %d (bg_both=%s)" % (self.bg_img.shape[0], n, self.bg_both))
        bg_s, bg_t = self.bg_img[:n], self.bg_img[n:] if self.bg_both else self.bg_img[:n]
        self.fake_src = src_mask * bg_s + (1 - src_mask) * src_img
        self.fake_tsf = tsf_mask * bg_t + (1 - tsf_mask) * tsf_img
        nchw = lambda t: t.permute(0, 3, 1, 2).contiguous()
        return nchw(self.bg_img), nchw(self.fake_src), nchw(self.fake_tsf), torch.cat([nchw(src_mask), nchw(tsf_mask)], dim=0)

    # ------------------------------------------------------------------ losses + backward (:368-394, :355-356)
    @torch.no_grad()
    def backward(self, all_reduce=True):
        """`all_reduce` is decided HERE, before the first gradient is written: in a data-parallel job the buckets go on the wire
        underneath this pass (sharding.GradientBuckets).  backward(all_reduce=False) leaves this rank's own gradients in flat_g."""
        b, lam = self.b, self.lam
        self._begin_buckets(all_reduce)
        self.flat_g.zero_()
        self._untouched = set(self.G)
        to_nhwc = lambda t: t.permute(0, 2, 3, 1)
        src_img, src_mask, tsf_img, tsf_mask = self.src.img, self.src.mask, self.tsf.img, self.tsf.mask
        n = src_img.shape[0]
        bg, bg_t = self.bg_img[:n], self.bg_img[n:] if self.bg_both else self.bg_img[:n]
        # adversarial term through the discriminator (its parameters are not touched)
        fake_in = torch.cat([self.fake_tsf.permute(0, 3, 1, 2), b["input_G_tsf"][:, 3:]], dim=1).contiguous()
        adv, d_in = self.D.input_grad(fake_in, 0.0)
        terms = dict(g_adv=adv * lam["adv"])
        d_ft = to_nhwc(d_in[:, 0:3]) * lam["adv"]
        # L1 terms
        diff_s = self.fake_src - to_nhwc(b["real_src"])
        diff_t = self.fake_tsf - to_nhwc(b["real_tsf"])
        terms["g_rec"] = diff_s.abs().mean() * lam["rec"]
        d_fs = torch.sign(diff_s) * (lam["rec"] / diff_s.numel())
        if self.use_vgg:
            v_loss, v_grad = self.vgg.loss_and_grad(self.fake_tsf.contiguous(), to_nhwc(b["real_tsf"]).contiguous())
            terms["g_tsf"] = v_loss * lam["tsf"]
            d_ft = d_ft + v_grad * lam["tsf"]
        else:
            terms["g_tsf"] = diff_t.abs().mean() * lam["tsf"]
            d_ft = d_ft + torch.sign(diff_t) * (lam["tsf"] / diff_t.numel())
        if self.use_style:
            s_loss, s_grad = self.vgg.style_loss_and_grad(self.fake_tsf.contiguous(), to_nhwc(b["real_tsf"]).contiguous())
            terms["g_style"] = s_loss * self.lambda_style
            d_ft = d_ft + s_grad * self.lambda_style
        if self.face is not None:
            f_loss, f_grad = self.face.loss_and_grad(self.fake_tsf.contiguous(), to_nhwc(b["real_tsf"]).contiguous(), b["head_bbox"])
            terms["g_face"] = f_loss * self.lambda_face
            d_ft = d_ft + f_grad * self.lambda_face
        # mask terms on cat([src_mask, tsf_mask])
        masks = torch.cat([src_mask, tsf_mask], dim=0)
        target = to_nhwc(b["bg_mask"])
        if self.mask_bce:   # torch.nn.BCELoss: logs clamped at -100
            lm, l1m = torch.log(masks).clamp_(min=-100.0), torch.log1p(-masks).clamp_(min=-100.0)
            terms["g_mask"] = -(target * lm + (1 - target) * l1m).mean() * lam["mask"]
            d_masks = ((1 - target) / (1 - masks).clamp_(min=1e-12) - target / masks.clamp(min=1e-12)) * (lam["mask"] / masks.numel())
        else:
            dm = masks - target
            terms["g_mask"] = (dm * dm).mean() * lam["mask"]
            d_masks = dm * (2.0 * lam["mask"] / dm.numel())
        dx = masks[:, :, :-1] - masks[:, :, 1:]
        dy = masks[:, :-1] - masks[:, 1:]
        terms["g_mask_smooth"] = (dx.abs().mean() + dy.abs().mean()) * lam["smooth"]
        gx = torch.sign(dx) * (lam["smooth"] / dx.numel())
        gy = torch.sign(dy) * (lam["smooth"] / dy.numel())
        d_masks[:, :, :-1] += gx
        d_masks[:, :, 1:] -= gx
        d_masks[:, :-1] += gy
        d_masks[:, 1:] -= gy
        # blends: fake = m * bg + (1 - m) * c
        d_src_mask = d_masks[:n] + (d_fs * (bg - src_img)).sum(-1, keepdim=True)
        d_tsf_mask = d_masks[n:] + (d_ft * (bg_t - tsf_img)).sum(-1, keepdim=True)
        d_src_img = d_fs * (1 - src_mask)
        d_tsf_img = d_ft * (1 - tsf_mask)
        d_bg = torch.cat([d_fs * src_mask, d_ft * tsf_mask], dim=0) if self.bg_both else d_fs * src_mask + d_ft * tsf_mask
        # --- transfer stream
        d, d_skips = self.tsf.decode_regress_backward(d_tsf_img, d_tsf_mask)
        g_res = [None] * self.repeat
        # one deterministic scatter plan per pyramid level (the trunk's seven warps share the 1/8-resolution flow)
        # (env LWG_GS_ATOMIC=1: the float-atomic scatter instead, an A/B switch)
        import os
        atomic = os.environ.get("LWG_GS_ATOMIC", "0") == "1"
        plans = {l: None if atomic else ops.GridSamplePlan(self.Ts[l], tuple(self.src_enc[l].shape), self.align)
                 for l in range(1, N_DOWN + 1)}
        for i in reversed(range(self.repeat)):
            g_res[i] = ops.grid_sample_backward(d.contiguous(), self.Ts[N_DOWN], tuple(self.src_res[i].shape), self.align,
                                                plan=plans[N_DOWN], deterministic=not atomic)
            d = self.tsf.res[i].backward(d)
        g_enc = [None] * (N_DOWN + 1)
        for i in reversed(range(1, N_DOWN + 1)):
            if i < N_DOWN:
                d = d + d_skips[i]
            g_enc[i] = ops.grid_sample_backward(d.contiguous(), self.Ts[i], tuple(self.src_enc[i].shape), self.align, plan=plans[i],
                                                deterministic=not atomic)
            d = self.tsf.enc[i].backward(d)
        self.tsf.enc[0].backward(d + d_skips[0], need_dx=False)
        # --- source stream: its own decoder/heads plus the Liquid Warping Block gradients
        d, d_skips = self.src.decode_regress_backward(d_src_img, d_src_mask)
        for i in reversed(range(self.repeat)):
            d = self.src.res[i].backward(d + g_res[i])
        for i in reversed(range(1, N_DOWN + 1)):
            d = d + g_enc[i]
            if i < N_DOWN:
                d = d + d_skips[i]
            d = self.src.enc[i].backward(d)
        self.src.enc[0].backward(d + d_skips[0], need_dx=False)
        # --- BGNet
        d = self.bg_head.backward(d_bg)
        mods = self.bg_enc + self.bg_res + self.bg_dec
        for j in reversed(range(len(mods))):
            d = mods[j].backward(d) if j > 0 else mods[j].backward(d, need_dx=False)
        self.terms = terms
        return terms

    @torch.no_grad()
    def step(self, all_reduce=True):
        """Adam on the flat buffers.  With buckets (backward(all_reduce=True) in a data-parallel job) the averaging already ran
        underneath the backward pass and is joined here whatever `all_reduce` says -- half-averaged gradients are never stepped
        on; without them `all_reduce` selects one blocking all-reduce of the whole buffer."""
        if self._buckets is not None:
            if not all_reduce and not self._buckets.joined:
                raise RuntimeError("step(all_reduce=False) after backward(all_reduce=True): the buckets of this pass are already "
                                   "on the wire; decide in backward(all_reduce=False)")
            self._join_buckets()            # most buckets were averaged underneath the backward pass; wait for the rest
        elif all_reduce:
            sharding.average_gradients(self.flat_g)
        self.t += 1
        if self.t_dev is not None:
            ops.adam_update_device_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.t_dev, self.lr, self.betas, self.eps)
        else:
            ops.adam_update(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.t, self.lr, self.betas, self.eps)

    def use_device_step(self, on=True):
        """Keep Adam's step count in device memory (ops.adam_update_device_step) so that an iteration captured in a graph
        replays with the right bias corrections; off: back to the host count (which replays did not advance)."""
        if on and self.t_dev is None:
            self.t_dev = ops.adam_step_state(self.t, self.betas, self.flat_p.device)
        elif not on and self.t_dev is not None:
            self.t = int(self.t_dev[0].item())
            self.t_dev = None

    def optimize_G(self, batch):
        """forward + losses + backward + Adam: the generator half of optimize_parameters (impersonator_trainer.py:350-357)."""
        fake = self.forward(batch)
        terms = self.backward(all_reduce=True)
        self.step(all_reduce=True)
        return terms, fake
