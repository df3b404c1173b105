"""VGG19 perceptual loss of the generator update on the device: VGGLoss + Vgg19 (networks/networks.py:83-186, used by
impersonator_trainer.py:256-260, 376-377 under --use_vgg) -- the loss value and its gradient wrt the generated image.

The reference builds it from torchvision.models.vgg19(pretrained=True); that download does not exist here, so the
weights come in as a state_dict in torchvision's naming (features.N.weight / features.N.bias, N = 0 .. 28).  The network
is frozen: only data gradients are needed, the op-level kernels of impersonator_amd.ops compute them.
"""
import torch

from .. import ops

# torchvision's vgg19().features up to relu5_1: (index, out channels), 'M' = MaxPool2d(2, 2)
CFG = [(0, 64), (2, 64), "M", (5, 128), (7, 128), "M", (10, 256), (12, 256), (14, 256), (16, 256), "M",
       (19, 512), (21, 512), (23, 512), (25, 512), "M", (28, 512)]
TAPS = (0, 5, 10, 19, 28)                               # slice outputs relu1_1 .. relu5_1 (networks.py:137-155)
WEIGHTS = (1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)   # networks.py:179


def _pool(x):
    n, h, w, c = x.shape
    return x.view(n, h // 2, 2, w // 2, 2, c).amax(dim=(2, 4))


def _pool_backward(x, y, dy):
    """Gradient of the 2x2 max: to the position(s) holding the maximum.  A window of equal values (after the ReLU: zeros)
    hands the gradient to all four instead of torch's first one; those positions have a zero ReLU gradient right after."""
    n, h, w, c = x.shape
    sel = x.view(n, h // 2, 2, w // 2, 2, c) == y.view(n, h // 2, 1, w // 2, 1, c)
    return (sel * dy.view(n, h // 2, 1, w // 2, 1, c)).reshape(n, h, w, c)


class Vgg19Perceptual(object):
    def __init__(self, state_dict, precision="fp32", device=None):
        if precision not in ops.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(ops.PRECISIONS))
        dev = device or torch.device("cuda", torch.cuda.current_device())
        self.precision = precision
        self.w, self.b = {}, {}
        for item in CFG:
            if item == "M":
                continue
            i = item[0]
            try:
                w = state_dict["features.%d.weight" % i].detach().float()
                b = state_dict["features.%d.bias" % i].detach().float()
            except KeyError:
                raise KeyError("VGG19 state_dict lacks features.%d.weight / .bias (torchvision vgg19 naming expected)" % i)
            if i == 0:
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 5))      # the image travels as an 8-channel NHWC tensor
            self.w[i], self.b[i] = w.contiguous().to(dev), b.contiguous().to(dev)
        # data gradient of the first conv (64 -> 3 image channels) as a forward conv with the transposed, flipped filter
        # padded to 64 output rows (the conv kernels produce multiples of 64 channels)
        w0 = self.w[0][:, :3]
        wt = torch.zeros(64, 64, 3, 3, device=dev)
        wt[:3] = w0.flip(2, 3).permute(1, 0, 2, 3)
        self.w0_t = wt.contiguous()

    def _conv(self, x, i):
        if self.precision == "bf16x3" and i != 0:
            y = ops.conv2d_forward(x, self.w[i], None, 1, 1, precision="bf16x3")
            return y.add_(self.b[i]).clamp_(min=0)
        return ops.conv2d_forward(x, self.w[i], self.b[i], 1, 1).clamp_(min=0)

    def _features(self, z):
        """the conv stack on an 8-channel NHWC batch; returns the trace the backward pass walks"""
        trace = []          # ('conv', index, output) / ('pool', input, output)
        for item in CFG:
            if item == "M":
                p = _pool(z)
                trace.append(("pool", z, p))
                z = p
            else:
                z = self._conv(z, item[0])
                trace.append(("conv", item[0], z))
        return trace

    def _backward(self, trace, n, tap_grad):
        """gradient wrt the first n images of the batch, given tap_grad(tap position, slice output) -> gradient wrt that
        slice output's first n images"""
        d = None
        for kind, a, out in reversed(trace):
            if kind == "pool":
                d = _pool_backward(a[:n], out[:n], d)
                continue
            if a in TAPS:
                g = tap_grad(TAPS.index(a), out)
                d = g if d is None else d + g
            d = (d * (out[:n] > 0)).contiguous()
            if a == 0:
                d = ops.conv2d_forward(d, self.w0_t, None, 1, 1, precision=self.precision)[..., :3]
            else:
                cin = self.w[a].shape[1]
                d = ops.conv2d_backward_data(d, self.w[a], (n, d.shape[1], d.shape[2], cin), 1, 1, precision=self.precision)
        return d.contiguous()

    @torch.no_grad()
    def loss_and_grad(self, x, y):
        """VGGLoss.  x (generated), y (target): (N,H,W,3) NHWC on the device, H and W multiples of 16.
        -> (sum_i w_i * mean|f_i(x) - f_i(y)|, its gradient wrt x (N,H,W,3))."""
        n = x.shape[0]
        trace = self._features(torch.nn.functional.pad(torch.cat([x, y], dim=0), (0, 5)).contiguous())
        loss = torch.zeros((), device=x.device)

        def tap_grad(i, out):
            nonlocal loss
            diff = out[:n] - out[n:]
            loss += WEIGHTS[i] * diff.abs().mean()
            return torch.sign(diff) * (WEIGHTS[i] / diff.numel())
        return loss, self._backward(trace, n, tap_grad)

    @torch.no_grad()
    def style_loss_and_grad(self, x, y, size=224):
        """StyleLoss.forward (networks/networks.py:414-423, weight 1): both images resized to 224x224 (nearest), then
        sum_i mean|gram(f_i(x)) - gram(f_i(y))| / (H_i W_i) with gram(f) = f f^T per image.  -> (loss, gradient wrt x)."""
        n, h, w, _ = x.shape
        iy = (torch.arange(size, device=x.device) * h) // size        # F.interpolate(mode='nearest'): floor(i * in / out)
        ix = (torch.arange(size, device=x.device) * w) // size
        z = torch.cat([x, y], dim=0)[:, iy][:, :, ix]
        trace = self._features(torch.nn.functional.pad(z, (0, 5)).contiguous())
        loss = torch.zeros((), device=x.device)

        def tap_grad(i, out):
            nonlocal loss
            b, hh, ww, c = out.shape
            f = out.reshape(b, hh * ww, c)
            gram = torch.bmm(f.transpose(1, 2), f)                       # (2n, c, c): a small library GEMM per level
            diff = gram[:n] - gram[n:]
            loss += diff.abs().mean() / (hh * ww)
            s = torch.sign(diff) / (diff.numel() * hh * ww)
            return torch.bmm(f[:n], s + s.transpose(1, 2)).reshape(n, hh, ww, c)
        d224 = self._backward(trace, n, tap_grad)
        dx = torch.zeros_like(x)
        flat = (iy[:, None] * w + ix[None, :]).reshape(-1)               # nearest resize backward: scatter-add
        dx.view(n, h * w, 3).index_add_(1, flat, d224.reshape(n, size * size, 3))
        return loss, dx
