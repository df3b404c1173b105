"""GPU parity: InpaintSANet (background inpaintor, once per source) through the C ABI vs the CPU oracle."""
import numpy as np
import pytest
import torch

from impersonator_amd.utils import synthetic
from oracle import torch_ref

pytestmark = pytest.mark.gpu


def _net_and_sd(seed=0):
    from impersonator_amd.networks.inpaintor import InpaintSANet
    net = InpaintSANet(c_dim=4).eval()
    shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    sd = {k: torch.from_numpy(v) for k, v in synthetic.random_inpaintor_state_dict(shapes, seed).items()}
    net.load_state_dict(sd)
    return net.cuda(), sd


def test_inpaintor_matches_oracle():
    net, sd = _net_and_sd(0)
    img = torch.from_numpy(synthetic.smooth_image(5))
    yy, xx = np.mgrid[0:256, 0:256]
    mask = torch.from_numpy((((yy - 120) / 90.0) ** 2 + ((xx - 128) / 50.0) ** 2 < 1).astype(np.float32))[None, None]
    coarse, x, comp = net(img.cuda(), mask.cuda())
    with torch.no_grad():
        oc, ox, ocomp = torch_ref.inpaint_forward(sd, img, mask)
    for name, a, b in (("coarse", coarse, oc), ("x", x, ox), ("comp", comp, ocomp)):
        err = float((a.cpu() - b).abs().max())
        assert err <= 1e-3, (name, err)
    # the three return conventions of the reference (inpaintor.py:198-202)
    assert torch.equal(net(img.cuda(), mask.cuda(), only_x=True), x)
    assert torch.equal(net(img.cuda(), mask.cuda(), only_out=True), comp)
    assert float(x.abs().max()) <= 1.0


def test_imitator_personalize_with_inpaintor():
    """models/imitator.py:116-131: bg = bgnet(img, masks=body_mask, only_x=True) when no bg_img is supplied."""
    from impersonator_amd import demo
    net, sd = _net_and_sd(1)
    imitator, src_smpl, src_img, _ = demo.build_synthetic_imitator(batch_size=1, seed=0)
    imitator.bgnet = net
    imitator.personalize(src_img, src_smpl=src_smpl)
    si = imitator.src_info
    with torch.no_grad():
        bg_mask = torch_ref.morph(si["cond"][:, -1:].cpu(), imitator._opt.bg_ks, "erode")
        _, ox, _ = torch_ref.inpaint_forward(sd, torch.from_numpy(src_img)[None], 1 - bg_mask)
    assert float((si["bg"].cpu() - ox).abs().max()) <= 1e-3
