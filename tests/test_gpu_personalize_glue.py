"""GPU: the once-per-source glue of Imitator.personalize as liblwg kernels (personalize.hip: lwg_morph, lwg_mask_compose,
lwg_source_p2verts, lwg_vis_f2pts) against the reference's expressions (utils/util.py:73-89, models/imitator.py:105-107,127-135,
utils/nmr.py:506-546), exactly; and `personalize` itself under the profiler: no framework (ATen) kernel is launched."""
import numpy as np
import pytest
import torch

from impersonator_amd import demo
from impersonator_amd.utils import util
from impersonator_amd.utils.nmr import SMPLRenderer
from oracle import torch_ref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(1, 256, 256), (2, 100, 75), (3, 16, 64), (1, 17, 200)])
def test_morph_kernel_equals_the_reference_expression(shape):
    n, h, w = shape
    g = torch.Generator().manual_seed(h * 7 + w)
    m = (torch.rand(n, 1, h, w, generator=g) > 0.3).float()
    m[:, :, h // 4:h // 2 + 3, w // 5:w // 2 + 9] = 1
    m[:, :, : h // 6, :] = 1        # a block touching the border: the pad value decides
    for ks in (3, 13, 5, 31):
        for mode in ("erode", "dilate"):
            want = torch_ref.morph(m, ks, mode)
            assert torch.equal(util.morph(m.cuda(), ks, mode).cpu(), want), (ks, mode)
            assert torch.equal(util.morph(m.cuda(), ks, mode, complement=True).cpu(), 1 - want), (ks, mode)
    # a channel slice of an NCHW tensor, as personalize passes it (cond[:, -1:]): image planes a batch stride apart
    cond = torch.rand(n, 3, h, w, generator=g)
    cond[:, -1:] = m
    assert torch.equal(util.morph(cond.cuda()[:, -1:], 13, "erode").cpu(), torch_ref.morph(m, 13, "erode"))


def test_mask_compose_and_source_p2verts():
    g = torch.Generator().manual_seed(3)
    r = SMPLRenderer(image_size=64, faces=np.zeros((4, 3), np.int32), map_fn=np.zeros((5, 3), np.float32))
    img = torch.rand(2, 3, 40, 56, generator=g) * 2 - 1
    mask = (torch.rand(2, 1, 40, 56, generator=g) > 0.5).float()
    tail = torch.rand(2, 3, 40, 56, generator=g)
    assert torch.equal(r.mask_compose(img.cuda(), mask.cuda(), tail.cuda()).cpu(), torch.cat([img * mask, tail], 1))
    assert torch.equal(r.mask_compose(img.cuda(), mask.cuda(), tail.cuda(), invert=True).cpu(), torch.cat([img * (1 - mask), tail], 1))
    assert torch.equal(r.mask_compose(img.cuda(), mask.cuda(), mask.cuda()).cpu(), torch.cat([img * mask, mask], 1))
    f2v = torch.randn(2, 777, 3, 3, generator=g)
    ref = f2v.clone()
    p = ref[:, :, :, 0:2]            # models/imitator.py:105-107: the view ...
    p[:, :, :, 1] *= -1              # ... mutates f2verts
    dev = f2v.cuda()
    got = r.source_p2verts(dev)
    assert torch.equal(dev.cpu(), ref) and torch.equal(got.cpu(), p.contiguous()) and got.is_contiguous()


def test_get_vis_f2pts_kernel_reproduces_unique_slicing():
    """utils/nmr.py:506-546 incl. hazard H10: `fim.unique()[1:]` drops the smallest value PRESENT -- the background's -1 when a
    background pixel exists, the lowest visible face id when the body covers the whole image."""
    g = torch.Generator().manual_seed(5)
    nf = 500
    pts = torch.randn(3, nf, 3, 2, generator=g)
    fim = torch.randint(0, nf // 3, (3, 32, 32), generator=g, dtype=torch.int32)
    fim[0, :5] = -1                   # image 0: background present
    fim[1] = fim[1].clamp(min=7)      # image 1: no background pixel, lowest visible id 7 (dropped by the reference's slicing)
    fim[2, 3, 3] = -1
    want = SMPLRenderer.get_vis_f2pts(pts, fim)                  # CPU tensors: the reference's own indexing
    got = SMPLRenderer.get_vis_f2pts(pts.cuda(), fim.cuda()).cpu()
    assert torch.equal(got, want)
    assert bool((want[1, 7] == -2).all()) and int((fim[1] == 7).sum()) > 0       # the H10 case is really exercised
    assert torch.equal(SMPLRenderer.get_vis_f2pts(pts[0].cuda(), fim[0].cuda()).cpu(), want[0])   # un-batched form


@pytest.mark.parametrize("variant", ["given_bg", "bgnet", "inpaintor_only_vis"])
def test_personalize_launches_no_framework_kernel(variant):
    """Every kernel `Imitator.personalize` puts on the device is liblwg's: the torch
    profiler's device-side records of the call must not contain an ATen kernel (`at::native::...`)."""
    from torch.profiler import ProfilerActivity, profile
    opt = demo.default_opt(batch_size=2, image_size=256, only_vis=(variant == "inpaintor_only_vis"))
    imitator, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=2, seed=0, image_size=256, opt=opt)
    if variant == "inpaintor_only_vis":
        from impersonator_amd.networks.inpaintor import InpaintSANet
        from tests import helpers
        net = InpaintSANet(c_dim=4).eval()
        net.load_state_dict({k: torch.from_numpy(v) for k, v in helpers.inpaintor_state_dict(seed=1).items()})
        imitator.bgnet = net.cuda()
    kw = dict(bg_img=bg_img) if variant == "given_bg" else {}
    imitator.personalize(src_img, src_smpl=src_smpl, **kw)      # first call: handles are created, weights uploaded
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        imitator.personalize(src_img, src_smpl=src_smpl, **kw)
        torch.cuda.synchronize()
    from torch.autograd import DeviceType
    kernels = [e.name for e in prof.events() if e.device_type == DeviceType.CUDA]
    ours = [k for k in kernels if "lwg" in k]
    foreign = sorted({k for k in kernels if "at::native" in k or "at::cuda" in k or "elementwise" in k})
    print("%s: %d device records, %d liblwg kernels; framework kernels: %s" % (variant, len(kernels), len(ours), foreign))
    assert len(ours) >= 20, "the profiler saw no liblwg kernels: %s" % sorted(set(kernels))[:10]
    assert not foreign, foreign
