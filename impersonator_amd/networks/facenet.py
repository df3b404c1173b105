"""Face identity term of the generator update on the device: FaceLoss (networks/networks.py:211-312) over Sphere20a
(networks/facenet.py:200-290), used by impersonator_trainer.py:268-273, 383-385 under --use_face -- the loss value and its
gradient wrt the generated image.

The reference loads `--face_model` (assets/pretrains/sphere20a_20171020.pth, a download); here the same file, or any
state_dict in Sphere20a's naming, is handed in.  The network is frozen: only data gradients are needed, computed with the
op-level convolution kernels of impersonator_amd.ops; the 112x96 head crops are resampled by torch's bilinear
interpolate (glue on a tensor of a few hundred KB), fc5 is one small library GEMM.
"""
import torch
import torch.nn.functional as F

from .. import ops

STAGES = (("1", 64, 1), ("2", 128, 2), ("3", 256, 4), ("4", 512, 1))   # stage, channels, residual units
HW = (112, 96)


def _prelu(a, slope):
    return torch.where(a > 0, a, a * slope)


class SphereFaceLoss(object):
    def __init__(self, state_dict, device=None):
        dev = device or torch.device("cuda", torch.cuda.current_device())
        sd = {k: v.detach().float() for k, v in state_dict.items() if not k.startswith("fc6")}   # networks.py:391-394
        need = ["conv%s_%d.weight" % (st, j) for st, _, u in STAGES for j in range(1, 2 + 2 * u)] + ["fc5.weight", "fc5.bias"]
        missing = [k for k in need if k not in sd]
        if missing:
            raise KeyError("Sphere20a state_dict lacks %s" % ", ".join(missing[:4]))
        self.p = {k: v.contiguous().to(dev) for k, v in sd.items()}
        w0 = self.p["conv1_1.weight"]                                       # (64, 3, 3, 3)
        self.p["conv1_1.weight"] = F.pad(w0, (0, 0, 0, 0, 0, 5)).contiguous()   # images travel as 8-channel NHWC tensors
        # data gradient of conv1_1 (stride 2) = ConvTranspose2d forward with the same tensor read as (in 64, out 3):
        # padded to 64 output channels (the conv kernels produce multiples of 64)
        self.w0_t = F.pad(w0, (0, 0, 0, 0, 0, 61)).contiguous()

    # ---- the network, keeping what the backward pass needs
    def _conv(self, x, name, stride, trace):
        a = ops.conv2d_forward(x, self.p["conv%s.weight" % name], self.p["conv%s.bias" % name], stride, 1)
        trace.append((name, stride, tuple(x.shape), a))
        return _prelu(a, self.p["relu%s.weight" % name])

    def _forward(self, z):
        feats, plan = [], []
        for st, _, units in STAGES:
            t = []
            z = self._conv(z, "%s_1" % st, 2, t)
            us = []
            for u in range(units):
                tu = []
                z = z + self._conv(self._conv(z, "%s_%d" % (st, 2 + 2 * u), 1, tu), "%s_%d" % (st, 3 + 2 * u), 1, tu)
                us.append(tu)
            plan.append((t[0], us))
            feats.append(z)
        flat = z.permute(0, 3, 1, 2).reshape(z.shape[0], -1)               # NCHW order, as x.view(x.size(0), -1)
        feats.append(flat @ self.p["fc5.weight"].t() + self.p["fc5.bias"])
        return feats, plan

    def _back(self, d, rec):
        """through PReLU and the conv of trace record `rec` = (name, stride, input shape, pre-activation)"""
        name, stride, xshape, a = rec
        n = d.shape[0]
        d = (d * torch.where(a[:n] > 0, torch.ones_like(d), self.p["relu%s.weight" % name].expand_as(d))).contiguous()
        if name == "1_1":
            return ops.conv2d_forward(d, self.w0_t, None, 2, 1, transposed=True)[..., :3]
        return ops.conv2d_backward_data(d, self.p["conv%s.weight" % name], (n,) + xshape[1:], stride, 1)

    @torch.no_grad()
    def loss_and_grad(self, x, y, bbox):
        """x (generated), y (target): (N,H,W,3) NHWC on the device; bbox (N,4) = (min_x, max_x, min_y, max_y) pixel indices
        (BodyRecoveryFlow.cal_head_bbox).  -> (sum of the five L1 feature distances, its gradient wrt x (N,H,W,3))."""
        n = x.shape[0]
        boxes = [[int(v) for v in row] for row in (bbox.tolist() if torch.is_tensor(bbox) else bbox)]
        crops, heads = [], []
        with torch.enable_grad():
            for img, grad in ((x, True), (y, False)):
                for i in range(n):
                    x0, x1, y0, y1 = boxes[i]
                    c = img[i:i + 1, y0:y1, x0:x1, :].permute(0, 3, 1, 2).contiguous().requires_grad_(grad)
                    heads.append(F.interpolate(c, size=HW, mode="bilinear", align_corners=True))     # networks.py:306
                    if grad:
                        crops.append(c)
        z = F.pad(torch.cat([h.detach() for h in heads], dim=0).permute(0, 2, 3, 1), (0, 5)).contiguous()
        feats, plan = self._forward(z)
        loss = torch.zeros((), device=x.device)
        grads = []
        for f in feats:
            diff = f[:n] - f[n:]
            loss += diff.abs().mean()
            grads.append(torch.sign(diff) / diff.numel())
        d = (grads[4] @ self.p["fc5.weight"]).reshape(n, 512, 7, 6).permute(0, 2, 3, 1)
        for s in reversed(range(len(STAGES))):
            d = d + grads[s]
            opening, units = plan[s]
            for tu in reversed(units):
                d = d + self._back(self._back(d, tu[1]), tu[0])
            d = self._back(d, opening)
        d_heads = d.permute(0, 3, 1, 2).contiguous()
        dx = torch.zeros_like(x)
        for i in range(n):
            x0, x1, y0, y1 = boxes[i]
            (dc,) = torch.autograd.grad(heads[i], crops[i], d_heads[i:i + 1])
            dx[i, y0:y1, x0:x1, :] = dc[0].permute(1, 2, 0)
        return loss, dx
