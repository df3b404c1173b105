"""GPU parity at other image sizes / batch shapes than the headline 256x256 x 8 (the kernels take sizes at run time)."""
import numpy as np
import pytest
import torch

from impersonator_amd.utils import synthetic
from oracle import raster as oracle_raster
from oracle import torch_ref
from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("image_size,bs", [(128, 3), (512, 1)])
def test_generator_other_image_sizes(image_size, bs):
    from impersonator_amd.networks.generator import ImpersonatorGenerator
    sd = torch_ref.state_dict_from_numpy(helpers.generator_state_dict(seed=2, affine="random"))
    G = ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6, image_size=image_size, max_batch=bs)
    G.load_state_dict(sd)
    G = G.cuda()
    gen = torch.Generator().manual_seed(image_size)
    src = torch.rand(1, 6, image_size, image_size, generator=gen) * 2 - 1
    tsf = torch.rand(bs, 6, image_size, image_size, generator=gen) * 2 - 1
    T = torch.rand(bs, image_size, image_size, 2, generator=gen) * 2.2 - 1.1
    T[:, : image_size // 3] = -2.0
    bg = torch.rand(1, 3, image_size, image_size, generator=gen) * 2 - 1
    enc, res = G.encode_src(src.cuda())
    pred, color, mask = G.inference(enc, res, tsf.cuda(), T.cuda(), bg_img=bg.cuda())
    with torch.no_grad():
        o_enc, o_res = torch_ref.encode_src(sd, src)
        o_pred, o_color, o_mask = torch_ref.imitator_forward(sd, o_enc, o_res, bg, tsf, T)
    for name, a, b in (("color", color, o_color), ("mask", mask, o_mask), ("pred", pred, o_pred)):
        d, where = helpers.maxdiff(a, b)
        assert d <= 1e-3, (name, d, where)
    G.release()


@pytest.mark.parametrize("image_size", [32, 100, 512])
def test_rasteriser_other_image_sizes(image_size):
    """non-power-of-two and large sizes: the (2.*i + 1 - is) / is centre formula stays bit-identical"""
    from impersonator_amd.utils.nmr import SMPLRenderer
    rest, faces = synthetic.body_mesh()
    r = SMPLRenderer(image_size=image_size, faces=faces, map_fn=synthetic.uv_seg_map_fn(rest, faces)).cuda()
    verts = torch.from_numpy(np.stack([synthetic.motion_verts(rest, t) for t in (1, 77)]))
    cam = torch.from_numpy(synthetic.cams(2, seed=9))
    f2v, fim, wim = r.render_fim_wim(cam.cuda(), verts.cuda())
    ofim, owim, _ = oracle_raster.rasterize_fim_wim(f2v.cpu().numpy(), image_size, 0.1, 100.0)
    assert np.array_equal(fim.cpu().numpy(), ofim)
    assert np.array_equal(wim.cpu().numpy().view(np.uint32), owim.view(np.uint32))


def test_empty_and_single_face_inputs():
    """ragged / degenerate inputs: a mesh entirely behind the far plane, one face, faces outside the frame"""
    from impersonator_amd.utils.nmr import SMPLRenderer
    rest, faces = synthetic.body_mesh()
    r = SMPLRenderer(image_size=64, faces=faces, map_fn=synthetic.uv_seg_map_fn(rest, faces)).cuda()
    tri = np.array([[[[-0.5, -0.5, 1.0], [0.5, -0.5, 1.0], [0.0, 0.6, 1.0]]]], np.float32)
    far = tri.copy()
    far[..., 2] = 500.0
    off = tri.copy()
    off[..., 0] += 5.0
    for f, expect_cov in ((tri, True), (far, False), (off, False)):
        fim, wim = r.rasterize(torch.from_numpy(f).cuda())
        ofim, owim, _ = oracle_raster.rasterize_fim_wim(f, 64)
        assert np.array_equal(fim.cpu().numpy(), ofim) and np.array_equal(wim.cpu().numpy(), owim)
        assert bool((fim >= 0).any()) == expect_cov
    # everything background -> cond is the background row, T is the -2 sentinel, the warped source is black
    out_T = r.cal_bc_transform(torch.zeros(1, 1, 3, 2).cuda(), torch.full((1, 64, 64), -1, dtype=torch.int32).cuda(),
                               torch.zeros(1, 64, 64, 3).cuda())
    assert bool((out_T == -2).all())
