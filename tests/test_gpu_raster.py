"""GPU parity: the HIP rasteriser / flow path (through the C ABI) against the CPU oracle and the
reference's known-answer fixtures.  Integer/index results must be identical, barycentrics bit-equal."""
import os

import numpy as np
import pytest
import torch

from oracle import raster as oracle_raster
from oracle import torch_ref
from tests import helpers

pytestmark = pytest.mark.gpu


def _renderer(image_size=256, faces=None, map_fn=None):
    from impersonator_amd.utils.nmr import SMPLRenderer
    if faces is None:
        s = helpers.scene()
        faces, map_fn = s["faces"], s["map_fn"]
    return SMPLRenderer(image_size=image_size, faces=faces, map_fn=map_fn, has_front=False).cuda()


def _check_raster(faces_np, image_size, near=0.1, far=100.0):
    r = _renderer(image_size)
    fim, wim, depth = r.rasterize(torch.from_numpy(faces_np).cuda(), near=near, far=far, return_depth=True)
    ofim, owim, odepth = oracle_raster.rasterize_fim_wim(faces_np, image_size, near, far)
    fim, wim, depth = fim.cpu().numpy(), wim.cpu().numpy(), depth.cpu().numpy()
    nbad = int((fim != ofim).sum())
    assert nbad == 0, "face index map differs at %d pixels, first %s" % (nbad, np.argwhere(fim != ofim)[:5])
    assert np.array_equal(wim.view(np.uint32), owim.view(np.uint32)), \
        "weight map not bit-equal, max |d| = %g" % np.abs(wim - owim).max()
    assert np.array_equal(depth.view(np.uint32), odepth.view(np.uint32))
    return fim, wim


def test_teapot_known_answer():
    # thirdparty/neural_renderer/tests/test_rasterize_silhouettes.py:16-35, through the HIP rasteriser
    z = helpers.golden("teapot_kat.npz")
    sil = np.unpackbits(z["silhouette"])[:256 * 256].reshape(256, 256).astype(bool)
    fim, _ = _check_raster(z["faces"], 256)
    assert np.array_equal(fim[2] >= 0, sil)
    for b in (0, 1, 3):  # the all-zero (fully degenerate) samples of the to_minibatch fixture
        assert (fim[b] == -1).all()


def test_synthetic_body_batch8_exact():
    s = helpers.scene()
    from impersonator_amd.utils import synthetic
    verts = np.stack([synthetic.motion_verts(s["rest"], t) for t in range(0, 1024, 128)])
    cam = synthetic.cams(8, seed=3)
    f2v = torch_ref.vertices_to_faces(torch_ref.project_vertices(torch.from_numpy(verts), torch.from_numpy(cam)),
                                      torch.from_numpy(s["faces"])).numpy()
    fim, _ = _check_raster(f2v, 256)
    assert (fim >= 0).sum() > 8 * 5000


@pytest.mark.parametrize("image_size,seed", [(64, 0), (64, 1), (256, 2), (128, 3)])
def test_triangle_soup_exact(image_size, seed):
    # random soup: large, tiny, sliver, degenerate, partly off-screen, near/far-clipped triangles
    rng = np.random.default_rng(seed)
    nf = 600
    c = rng.uniform(-1.2, 1.2, (nf, 1, 2))
    size = np.exp(rng.uniform(np.log(0.003), np.log(1.5), (nf, 1, 1)))
    xy = c + rng.normal(0, 1, (nf, 3, 2)) * size
    zc = rng.uniform(0.05, 3.0, (nf, 1))
    zz = np.clip(zc + rng.normal(0, 0.2, (nf, 3)), 0.02, None)
    zz[::50] = 150.0   # beyond far
    f = np.concatenate([xy, zz[:, :, None]], -1).astype(np.float32)
    f[5] = f[5][[0, 0, 0]]                    # point
    f[6, 2] = f[6, 1]                         # segment
    f[7, 2, :2] = 0.5 * (f[7, 0, :2] + f[7, 1, :2])   # collinear
    f[8, 2, :2] = f[8, 0, :2] + (f[8, 1, :2] - f[8, 0, :2]) * 0.3 + np.float32(1e-7)  # sliver
    f[9] = np.array([[-3, -3, 1], [3, -3, 1], [0, 4, 1]], np.float32)  # covers the whole image
    f[10] = f[9]                              # exact depth tie: lowest index must win
    f[11, :, 0] = 2e6                         # absurd coordinates
    faces = np.stack([f, f[::-1].copy()])     # second sample: reversed order (ties resolve differently)
    _check_raster(faces, image_size)


@pytest.mark.parametrize("image_size,nf", [(64, 14001), (40, 12003)])
def test_crowded_tile_flushes_its_face_list(image_size, nf):
    """more faces in one 32x8 tile than its LDS list holds (4096): the tile flushes mid-scan and keeps going; nf is
    not a multiple of 4 (the padded box slots must stay empty); image size not a multiple of the tile"""
    rng = np.random.default_rng(nf)
    c = np.array([0.1, -0.2]) + rng.uniform(-0.12, 0.12, (nf, 1, 2))
    xy = c + rng.normal(0, 0.03, (nf, 3, 2))
    zz = np.round(rng.uniform(0.5, 2.5, (nf, 1)), 2) + np.zeros((nf, 3))   # flat faces on a coarse depth grid: many exact ties
    f = np.concatenate([xy, zz[:, :, None]], -1).astype(np.float32)
    f[1::97] = f[0]                               # exact duplicates of face 0 far apart in the list
    faces = np.stack([f, f[::-1].copy()])
    _check_raster(faces, image_size)


def test_thousands_of_whole_image_faces():
    """slivers / degenerate faces get the whole image as their box: every tile lists every one of them (> 4096) and
    sweeps them with all its lanes"""
    rng = np.random.default_rng(5)
    nf = 10000          # about half survive the back-face cull
    a = rng.uniform(-0.9, 0.9, (nf, 1, 2))
    d = rng.normal(0, 1, (nf, 1, 2))
    t = np.array([0.0, 0.5, 1.0]).reshape(1, 3, 1)
    xy = a + d * t * 0.8
    xy[:, 2] += rng.normal(0, 2e-6, (nf, 2))       # almost collinear: |2*area| << 1e-4 * extent^2
    zz = rng.uniform(0.5, 2.0, (nf, 3))
    f = np.concatenate([xy, zz[:, :, None]], -1).astype(np.float32)
    f[7] = np.array([[-3, -3, 1.5], [3, -3, 1.5], [0, 4, 1.5]], np.float32)   # one honest full-screen face among them
    _check_raster(f[None], 64)


def test_rasteriser_is_unaffected_by_other_streams():
    """the entry point keeps no state in global memory between launches except its per-face records: results on a
    busy device (other streams hammering the fabric and the L2s) equal the quiet result bit for bit"""
    s = helpers.scene()
    from impersonator_amd.utils import synthetic
    r = _renderer()
    verts = torch.from_numpy(np.stack([synthetic.motion_verts(s["rest"], t) for t in range(0, 1024, 128)])).cuda()
    cam = torch.from_numpy(synthetic.cams(8, seed=3)).cuda()
    quiet = [r.render_fim_wim(cam.roll(k, 0), verts.roll(k, 0)) for k in range(4)]
    quiet = [(f.clone(), w.clone()) for _, f, w in quiet]
    torch.cuda.synchronize()
    noise_streams = [torch.cuda.Stream() for _ in range(3)]
    a = torch.randn(4096, 4096, device="cuda")
    big = torch.empty(64 << 20, device="cuda")
    side = torch.cuda.Stream()
    bad = 0
    for it in range(40):
        for k, ns in enumerate(noise_streams):
            with torch.cuda.stream(ns):
                if k == 0:
                    (a @ a).sum()
                else:
                    big.add_(1.0)
        with torch.cuda.stream(side):
            got = [r.render_fim_wim(cam.roll(k, 0), verts.roll(k, 0)) for k in range(4)]
        torch.cuda.synchronize()
        for (qf, qw), (_, f, w) in zip(quiet, got):
            bad += int(not (torch.equal(qf, f) and torch.equal(qw, w)))
    assert bad == 0, "%d of 160 batches differ under load" % bad


def test_negative_near_plane():
    tri = np.array([[[-0.5, -0.5, -1.0], [0.5, -0.5, -1.0], [0.0, 0.6, -1.0]],
                    [[-0.5, -0.5, 1.0], [0.5, -0.5, 1.0], [0.0, 0.6, 1.0]]], np.float32)[None]
    fim, _ = _check_raster(tri, 64, near=-5.0, far=100.0)
    assert set(np.unique(fim)) == {-1, 0}     # the negative-depth triangle is nearer


def test_render_fim_wim_encode_and_flow_match_oracle():
    s = helpers.scene()
    r = _renderer()
    cam, verts = helpers.t(s["tgt_cam"]), helpers.t(s["tgt_verts"])
    f2v, fim, wim = r.render_fim_wim(cam.cuda(), verts.cuda())
    of2v, ofim, owim = torch_ref.render_fim_wim(cam, verts, helpers.t(s["faces"]))
    assert torch.equal(f2v.cpu(), of2v)
    assert torch.equal(fim.cpu(), ofim)
    assert torch.equal(wim.cpu(), owim)

    cond, _ = r.encode_fim(cam.cuda(), verts.cuda(), fim=fim, transpose=True)
    assert torch.equal(cond.cpu(), torch_ref.encode_fim(ofim, helpers.t(s["map_fn"])))
    cond_nt, _ = r.encode_fim(cam.cuda(), verts.cuda(), fim=fim, transpose=False)
    assert torch.equal(cond_nt.cpu(), torch_ref.encode_fim(ofim, helpers.t(s["map_fn"]), transpose=False))

    sf2v, _, _ = torch_ref.render_fim_wim(helpers.t(s["src_cam"]), helpers.t(s["src_verts"]), helpers.t(s["faces"]))
    p2v = torch_ref.source_p2verts(sf2v)
    T = r.cal_bc_transform(p2v.cuda(), fim, wim)
    oT = torch_ref.cal_bc_transform(p2v, ofim, owim)
    d, where = helpers.maxdiff(T, oT)
    assert d <= 1e-6, (d, where)


def test_golden_fim_and_flow_from_reference():
    # outputs of the reference's own SMPLRenderer code (tests/golden/make_golden.py)
    g = helpers.golden("frame_golden.npz")
    s = helpers.scene()
    r = _renderer()
    _, sfim, _ = r.render_fim_wim(helpers.t(s["src_cam"]).cuda(), helpers.t(s["src_verts"]).cuda())
    assert np.array_equal(sfim.cpu().numpy(), g["src_fim"])
    f2v, fim, wim = r.render_fim_wim(helpers.t(s["tgt_cam"]).cuda(), helpers.t(s["tgt_verts"]).cuda())
    assert np.array_equal(fim.cpu().numpy(), g["fim"])
    assert np.array_equal(wim.cpu().numpy()[g["fim"] >= 0], g["wim_covered"])
    sf2v, _, _ = r.render_fim_wim(helpers.t(s["src_cam"]).cuda(), helpers.t(s["src_verts"]).cuda())
    p2v = sf2v[:, :, :, 0:2].clone()
    p2v[:, :, :, 1] *= -1
    T = r.cal_bc_transform(p2v, fim, wim)
    d, where = helpers.maxdiff(T, g["T"])
    assert d <= 1e-6, (d, where)


@pytest.mark.parametrize("align", [False, True])
def test_grid_sample_and_resize(align):
    from impersonator_amd.networks.generator import ImpersonatorGenerator
    G = ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6, align_corners=align)
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 24, 40, generator=gen)
    grid = torch.rand(2, 17, 33, 2, generator=gen) * 2.6 - 1.3
    grid[0, :3] = -2.0
    grid[1, 0, 0] = torch.tensor([1.0, -1.0])
    out = G.stn(x.cuda(), grid.cuda())
    ref = torch.nn.functional.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=align)
    d, where = helpers.maxdiff(out, ref)
    assert d <= 1e-5, (d, where)   # |x| reaches ~4: a few ulp of the 4-tap sum
    out1 = G.stn(x[:1].cuda(), grid.cuda())   # shared source broadcast
    ref1 = torch.nn.functional.grid_sample(x[:1].expand(2, -1, -1, -1), grid, align_corners=align)
    assert helpers.maxdiff(out1, ref1)[0] <= 1e-5

    T = torch.rand(2, 64, 64, 2, generator=gen) * 2 - 1
    T[0, 10:20, 10:30] = -2.0
    for hw in (32, 16, 8, 64):
        feat = torch.zeros(2, 3, hw, hw)
        o = G.resize_trans(feat, T.cuda())
        rf = torch_ref.resize_trans(T, hw, hw)
        d, where = helpers.maxdiff(o, rf)
        assert d <= 2e-6, (hw, d, where)


def test_transfer_fused_equals_granular_and_oracle():
    s = helpers.scene()
    r = _renderer()
    cam, verts = helpers.t(s["tgt_cam"]), helpers.t(s["tgt_verts"])
    sf2v, _, _ = r.render_fim_wim(helpers.t(s["src_cam"]).cuda(), helpers.t(s["src_verts"]).cuda())
    p2v = sf2v[:, :, :, 0:2].clone()
    p2v[:, :, :, 1] *= -1
    src_img = helpers.t(s["src_img"]).cuda()
    out = r.transfer(cam.cuda(), verts.cuda(), p2v, src_img)

    f2v, fim, wim = r.render_fim_wim(cam.cuda(), verts.cuda())
    cond, _ = r.encode_fim(cam.cuda(), verts.cuda(), fim=fim)
    T = r.cal_bc_transform(p2v, fim, wim)
    assert torch.equal(out["f2verts"], f2v) and torch.equal(out["fim"], fim) and torch.equal(out["wim"], wim)
    assert torch.equal(out["cond"], cond) and torch.equal(out["T"], T)

    o = torch_ref.transfer_frame(helpers.t(s["src_img"]), p2v.cpu(), cam, verts, helpers.t(s["faces"]),
                                 helpers.t(s["map_fn"]))
    assert helpers.maxdiff(out["tsf_img"], o["tsf_img"])[0] <= 2e-6
    d, where = helpers.maxdiff(out["tsf_inputs"], o["tsf_inputs"])
    assert d <= 2e-6, (d, where)
    g = helpers.golden("frame_golden.npz")
    assert abs(float(out["tsf_img"].double().mean()) - g["tsf_img_stat"][0]) < 1e-6


def test_pack_unpack_roundtrip():
    import ctypes
    from impersonator_amd import _lib
    lib = _lib.load()
    x = torch.randn(3, 6, 40, 24).cuda()
    nhwc = torch.empty(3, 40, 24, 8).cuda()
    _lib.check(lib.lwg_pack_nhwc(_lib.ptr(x), 3, 6, 40, 24, 8, _lib.ptr(nhwc), _lib.stream_ptr()))
    assert torch.equal(nhwc[..., :6], x.permute(0, 2, 3, 1)) and (nhwc[..., 6:] == 0).all()
    back = torch.empty_like(x)
    _lib.check(lib.lwg_unpack_nchw(_lib.ptr(nhwc), 3, 6, 40, 24, 8, _lib.ptr(back), _lib.stream_ptr()))
    assert torch.equal(back, x)


def test_argument_errors_are_reported_not_swallowed():
    from impersonator_amd import _lib
    lib = _lib.load()
    rc = lib.lwg_rasterize_fim_wim(None, 1, 1, 8, 0.1, 100.0, None, None, None, None, 0, None)
    assert rc == -1 and b"NULL" in lib.lwg_last_error()
    x = torch.zeros(1, 1, 3, 3).cuda()
    fim = torch.zeros(1, 8, 8, dtype=torch.int32).cuda()
    wim = torch.zeros(1, 8, 8, 3).cuda()
    rc = lib.lwg_rasterize_fim_wim(_lib.ptr(x), 1, 1, 8, 0.1, 100.0, _lib.ptr(fim), _lib.ptr(wim), None, None, 0, None)
    assert rc == -3
    r = _renderer()
    with pytest.raises(RuntimeError):
        r.render_fim_wim(torch.zeros(1, 3), torch.zeros(1, 6890, 3))   # CPU tensors: no fallback


def test_rasteriser_beside_bf16x3_convolutions():
    """The regression test of DESIGN.md section 5.1: SMPLRenderer.transfer and .rasterize on fixed inputs while
    conv_igemm_bf16x3 runs on two other streams.  Until the end of round 2 this gave wrong pixels in 25 % (rasterize:
    sliver faces painted by the wave-per-face sweep) to 99 % (transfer: groups of 16 faces missing, projection fused into
    the setup kernel) of the launches."""
    from impersonator_amd import demo, ops
    im, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=8, seed=0, affine="random")
    im.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
    smpls = torch.from_numpy(demo.synthetic_smpls(48, seed=3)).cuda()
    im.first_cam = smpls[0:1, 0:3].clone()
    im.transfer_params_by_smpl(smpls[32:48], "smooth", t=32)
    cam, verts, si = im.tsf_info["cam"].clone(), im.tsf_info["verts"].clone(), im.src_info
    ref = {k: v.clone() for k, v in im.render.transfer(cam, verts, si["p2verts"], si["img"]).items()}
    f2v = ref["f2verts"].clone()
    fim_ref = im.render.rasterize(f2v)[0].clone()
    xx, ww = torch.randn(8, 32, 32, 512, device="cuda"), torch.randn(512, 512, 3, 3, device="cuda") * 0.02
    lanes, side = [torch.cuda.Stream(), torch.cuda.Stream()], torch.cuda.Stream()
    torch.cuda.synchronize()
    bad = torch.zeros(2, dtype=torch.int64, device="cuda")
    for _ in range(60):
        for st in lanes:
            with torch.cuda.stream(st):
                for _ in range(12):
                    ops.conv2d_forward(xx, ww, None, 1, 1, precision="bf16x3")
        with torch.cuda.stream(side):
            for _ in range(2):
                out = im.render.transfer(cam, verts, si["p2verts"], si["img"])
                bad[0] += sum((out[k] != ref[k]).sum() for k in ("f2verts", "fim", "wim", "T", "tsf_img"))
                bad[1] += (im.render.rasterize(f2v)[0] != fim_ref).sum()
    torch.cuda.synchronize()
    assert bad.tolist() == [0, 0]
