"""Pins oracle/torch_ref.py (the CPU restatement that travels to the GPU box) against the REAL reference
imported from /root/reference.  Skipped where the reference tree is absent (the GPU box)."""
import types

import numpy as np
import pytest
import torch

from oracle import reference_loader, torch_ref
from tests import helpers

pytestmark = pytest.mark.skipif(not reference_loader.available(), reason="/root/reference not present")


@pytest.fixture(scope="module")
def ref():
    return reference_loader.load()


def test_state_dict_keys_match_reference(ref):
    from impersonator_amd.networks.generator import ImpersonatorGenerator
    mine = [(k, tuple(v.shape)) for k, v in ImpersonatorGenerator(4, 6, 6).state_dict().items()]
    theirs = [(k, tuple(v.shape)) for k, v in ref.generator.ImpersonatorGenerator(4, 6, 6).state_dict().items()]
    assert mine == theirs


def test_geometry_restatement_is_bit_identical(ref):
    s = helpers.scene()
    R = ref.nmr.SMPLRenderer
    rs = types.SimpleNamespace(faces=helpers.t(s["faces"]), image_size=256, map_fn=helpers.t(s["map_fn"]),
                               proj_func=ref.nmr.orthographic_proj_withz_idrot,
                               eye=[0, 0, -(1. / np.tan(np.radians(30)) + 1)])
    cam, verts = helpers.t(s["tgt_cam"]), helpers.t(s["tgt_verts"])
    f2v, fim, wim = R.render_fim_wim(rs, cam, verts)
    of2v, ofim, owim = torch_ref.render_fim_wim(cam, verts, helpers.t(s["faces"]))
    assert torch.equal(f2v, of2v) and torch.equal(fim, ofim) and torch.equal(wim, owim)
    cond, _ = R.encode_fim(rs, cam, verts, fim=fim, transpose=True)
    assert torch.equal(cond, torch_ref.encode_fim(ofim, helpers.t(s["map_fn"])))
    p2v = torch_ref.source_p2verts(f2v[:1])
    T = torch.cat([R.cal_bc_transform(rs, p2v, fim[i:i + 1], wim[i:i + 1]) for i in range(2)])
    assert torch.equal(T, torch_ref.cal_bc_transform(p2v, ofim, owim))
    assert torch.equal(ref.util.morph(cond[:, -1:], ks=3, mode="erode"), torch_ref.morph(cond[:, -1:], 3, "erode"))
    assert torch.equal(ref.util.morph(cond[:, -1:], ks=13, mode="dilate"), torch_ref.morph(cond[:, -1:], 13, "dilate"))


def test_generator_restatement_matches_reference_small(ref):
    # a 64x64 problem keeps this fast; the full-size pass is pinned by tests/golden/frame_golden.npz
    G = ref.generator.ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6, repeat_num=6).eval()
    sd = torch_ref.state_dict_from_numpy(helpers.generator_state_dict(seed=1))
    G.load_state_dict(sd)
    gen = torch.Generator().manual_seed(0)
    src = torch.rand(1, 6, 64, 64, generator=gen) * 2 - 1
    tsf = torch.rand(2, 6, 64, 64, generator=gen) * 2 - 1
    T = torch.rand(2, 64, 64, 2, generator=gen) * 2.4 - 1.2
    T[0, 20:40, 10:30] = -2
    with torch.no_grad():
        enc, res = G.encode_src(src)
        o_enc, o_res = torch_ref.encode_src(sd, src)
        for a, b in zip(enc + res, o_enc + o_res):
            assert torch.allclose(a, b, atol=1e-5, rtol=1e-5)
        for i in range(2):
            c, m = G.inference(enc, res, tsf[i:i + 1], T[i:i + 1])
            oc, om = torch_ref.generator_inference(sd, o_enc, o_res, tsf[i:i + 1], T[i:i + 1])
            assert torch.allclose(c, oc, atol=1e-5) and torch.allclose(m, om, atol=1e-5)
        c, m = G.swap(tsf[:1], enc, enc, res, res, T[:1], T[1:])
        oc, om = torch_ref.generator_swap(sd, tsf[:1], o_enc, o_enc, o_res, o_res, T[:1], T[1:])
        assert torch.allclose(c, oc, atol=1e-5) and torch.allclose(m, om, atol=1e-5)
        # batching target frames over one source (this build's extension) equals the reference's batch-1 loop
        oc2, om2 = torch_ref.generator_inference(sd, o_enc, o_res, tsf, T)
        c0, m0 = G.inference(enc, res, tsf[:1], T[:1])
        assert torch.allclose(oc2[:1], c0, atol=1e-5) and torch.allclose(om2[:1], m0, atol=1e-5)


def test_golden_frame_is_reproduced_by_the_oracle():
    """The committed golden (made by the real reference) equals what the travelling oracle computes."""
    g = helpers.golden("frame_golden.npz")
    s = helpers.scene()
    faces_t = helpers.t(s["faces"])
    sf2v, sfim, _ = torch_ref.render_fim_wim(helpers.t(s["src_cam"]), helpers.t(s["src_verts"]), faces_t)
    assert np.array_equal(sfim.numpy(), g["src_fim"])
    fr = torch_ref.transfer_frame(helpers.t(s["src_img"]), torch_ref.source_p2verts(sf2v), helpers.t(s["tgt_cam"]),
                                  helpers.t(s["tgt_verts"]), faces_t, helpers.t(s["map_fn"]))
    assert np.array_equal(fr["fim"].numpy(), g["fim"])
    assert np.array_equal(fr["wim"].numpy()[g["fim"] >= 0], g["wim_covered"])
    assert np.abs(fr["T"].numpy() - g["T"]).max() <= 1e-6


def test_inpaintor_restatement_matches_reference(ref):
    from impersonator_amd.networks.inpaintor import InpaintSANet
    from impersonator_amd.utils import synthetic
    net = ref.inpaintor.InpaintSANet(c_dim=4).eval()
    shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert shapes == [(k, tuple(v.shape)) for k, v in InpaintSANet(c_dim=4).state_dict().items()]
    sd = {k: torch.from_numpy(v) for k, v in synthetic.random_inpaintor_state_dict(shapes, 0).items()}
    net.load_state_dict(sd)
    gen = torch.Generator().manual_seed(0)
    img = torch.rand(1, 3, 256, 256, generator=gen) * 2 - 1
    mask = (torch.rand(1, 1, 256, 256, generator=gen) > 0.6).float()
    with torch.no_grad():
        a = net(img, mask)
        b = torch_ref.inpaint_forward(sd, img, mask)
    for x, y in zip(a, b):
        assert torch.allclose(x, y, atol=1e-6)


def test_discriminator_restatement_matches_reference(ref):
    """oracle discriminator forward / LSGAN loss / gradients == the reference's PatchDiscriminator under torch autograd,
    and the MI355X mirror exposes the same state_dict keys."""
    from impersonator_amd.networks.discriminator import PatchDiscriminator
    D = ref.discriminator.PatchDiscriminator(input_nc=6, ndf=64, n_layers=4, norm_type='instance', use_sigmoid=False)
    mine = [(k, tuple(v.shape)) for k, v in PatchDiscriminator(6, 64, 4, 'instance', False, image_size=64).state_dict().items()]
    assert mine == [(k, tuple(v.shape)) for k, v in D.state_dict().items()]
    sd = helpers.discriminator_state_dict(seed=3)
    D.load_state_dict(sd)
    gen = torch.Generator().manual_seed(1)
    real = torch.rand(2, 6, 64, 64, generator=gen) * 2 - 1
    fake = torch.rand(2, 6, 64, 64, generator=gen) * 2 - 1
    out = D(real)
    assert torch.allclose(out, torch_ref.discriminator_forward(sd, real), atol=1e-6, rtol=1e-5)
    loss = torch.mean((D(real) - 1) ** 2) + torch.mean((D(fake) + 1) ** 2)   # impersonator_trainer.py:404-414
    D.zero_grad()
    loss.backward()
    losses, grads, _ = torch_ref.discriminator_train_steps(sd, [(real, fake)])
    assert abs(losses[0] - float(loss)) < 1e-6
    for k, p in D.named_parameters():
        assert torch.allclose(p.grad, grads[k], atol=1e-7, rtol=1e-4), k


def test_bgnet_restatement_matches_reference(ref):
    G = ref.generator.ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6, repeat_num=6).eval()
    sd = torch_ref.state_dict_from_numpy(helpers.generator_state_dict(seed=2, affine="random"))
    G.load_state_dict(sd)
    x = torch.rand(1, 4, 64, 64, generator=torch.Generator().manual_seed(4)) * 2 - 1
    with torch.no_grad():
        assert torch.allclose(G.bg_model(x), torch_ref.bgnet_forward(sd, x), atol=1e-5, rtol=1e-5)


def test_generator_forward_restatement_matches_reference(ref):
    """The trainer's generator pass (ImpersonatorGenerator.forward / infer_front, generator.py:204-243)."""
    G = ref.generator.ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6, repeat_num=6).eval()
    sd = torch_ref.state_dict_from_numpy(helpers.generator_state_dict(seed=2, affine="random"))
    G.load_state_dict(sd)
    gen = torch.Generator().manual_seed(6)
    bg = torch.rand(2, 4, 64, 64, generator=gen) * 2 - 1
    src = torch.rand(2, 6, 64, 64, generator=gen) * 2 - 1
    tsf = torch.rand(2, 6, 64, 64, generator=gen) * 2 - 1
    T = torch.rand(2, 64, 64, 2, generator=gen) * 2.4 - 1.2
    with torch.no_grad():
        theirs = G(bg, src, tsf, T)
        mine = torch_ref.generator_forward(sd, bg, src, tsf, T)
    for a, b in zip(theirs, mine):
        assert torch.allclose(a, b, atol=1e-5, rtol=1e-5)


def test_generator_train_loss_matches_reference_trainer(ref):
    """oracle generator_train_loss / gradients == the reference's ImpersonatorTrainer.forward + _optimize_G run unbound
    on a stub `self` (its __init__ needs datasets and downloads).  `_crt_tsf` is torch.nn.L1Loss: without --use_vgg the
    reference never defines it (impersonator_trainer.py:256-260 vs :376-380), with it it needs the VGG19 download."""
    T = ref.trainer.Impersonator
    G = ref.generator.ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6, repeat_num=6)
    D = ref.discriminator.PatchDiscriminator(input_nc=6, ndf=64, n_layers=4, norm_type='instance', use_sigmoid=False)
    gsd = torch_ref.state_dict_from_numpy(helpers.generator_state_dict(seed=2, affine="random"))
    dsd = helpers.discriminator_state_dict(seed=3)
    G.load_state_dict(gsd)
    D.load_state_dict(dsd)
    b = helpers.train_batch(seed=5, n=2, size=64)
    opt = types.SimpleNamespace(bg_both=False, use_vgg=False, use_style=False, use_face=False, lambda_D_prob=1, lambda_rec=10,
                                lambda_tsf=10, lambda_mask=0.1, lambda_mask_smooth=1e-5)
    me = types.SimpleNamespace(_G=G, _D=D, _opt=opt, _input_G_bg=b["input_G_bg"], _input_G_src=b["input_G_src"],
                               _input_G_tsf=b["input_G_tsf"], _T=b["T"], _real_src=b["real_src"], _real_tsf=b["real_tsf"],
                               _bg_mask=b["bg_mask"], _crt_l1=torch.nn.L1Loss(), _crt_tsf=torch.nn.L1Loss(),
                               _crt_mask=torch.nn.MSELoss(), _loss_g_style=torch.zeros(1), _loss_g_face=torch.zeros(1),
                               _loss_g_mask_smooth=torch.zeros(1))
    me._compute_loss_D = types.MethodType(T._compute_loss_D, me)
    me._compute_loss_smooth = types.MethodType(T._compute_loss_smooth, me)
    fake = T.forward(me)
    loss = T._optimize_G(me, *fake)
    G.zero_grad()
    loss.backward()
    total, terms, mine_fake = torch_ref.generator_train_loss({k: v.clone().requires_grad_(True) for k, v in gsd.items()}, dsd, b)
    assert abs(float(loss) - float(total)) < 1e-5 * max(1.0, float(total))
    for a, c in zip(fake, mine_fake):
        assert torch.allclose(a, c, atol=1e-5, rtol=1e-5)
    _, grads, _ = torch_ref.generator_train_steps(gsd, dsd, [b])
    for k, p in G.named_parameters():
        assert torch.allclose(p.grad, grads[k], atol=1e-6 + 1e-4 * float(p.grad.abs().max()), rtol=0), k


def _torchvision_vgg19(vsd):
    """What torchvision.models.vgg19().features is, with the given weights (the reference slices it by index)."""
    layers, cin = [], 3
    for v in [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]:
        if v == "M":
            layers.append(torch.nn.MaxPool2d(2, 2))
        else:
            conv = torch.nn.Conv2d(cin, v, 3, padding=1)
            k = "features.%d" % len(layers)
            if k + ".weight" in vsd:
                conv.weight.data.copy_(vsd[k + ".weight"])
                conv.bias.data.copy_(vsd[k + ".bias"])
            layers += [conv, torch.nn.ReLU(inplace=True)]
            cin = v
    return types.SimpleNamespace(features=torch.nn.Sequential(*layers))


def test_generator_train_loss_with_the_training_scripts_flags(ref):
    """scripts/train_iPER.sh trains with --mask_bce --use_vgg (and --bg_both in train_iPER_Place2.sh): the reference's
    forward + _optimize_G with BCELoss, its own VGGLoss / Vgg19 classes (torchvision.models.vgg19 replaced by the same
    layer stack with seeded weights: the real ones are a download) and two backgrounds == the oracle with those options."""
    T = ref.trainer.Impersonator
    G = ref.generator.ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6, repeat_num=6)
    D = ref.discriminator.PatchDiscriminator(input_nc=6, ndf=64, n_layers=4, norm_type='instance', use_sigmoid=False)
    gsd = torch_ref.state_dict_from_numpy(helpers.generator_state_dict(seed=2, affine="random"))
    dsd = helpers.discriminator_state_dict(seed=3)
    vsd = helpers.vgg19_state_dict(seed=4)
    G.load_state_dict(gsd)
    D.load_state_dict(dsd)
    b = helpers.train_batch(seed=5, n=2, size=64, bg_both=True)
    ref.networks.models.vgg19 = lambda pretrained=True: _torchvision_vgg19(vsd)
    crt_tsf = ref.networks.VGGLoss(vgg=ref.networks.Vgg19())
    opt = types.SimpleNamespace(bg_both=True, use_vgg=True, use_style=False, use_face=False, lambda_D_prob=1, lambda_rec=10,
                                lambda_tsf=10, lambda_mask=1.0, lambda_mask_smooth=1.0)
    me = types.SimpleNamespace(_G=G, _D=D, _opt=opt, _input_G_bg=b["input_G_bg"], _input_G_src=b["input_G_src"],
                               _input_G_tsf=b["input_G_tsf"], _T=b["T"], _real_src=b["real_src"], _real_tsf=b["real_tsf"],
                               _bg_mask=b["bg_mask"], _crt_l1=torch.nn.L1Loss(), _crt_tsf=crt_tsf,
                               _crt_mask=torch.nn.BCELoss(), _loss_g_style=torch.zeros(1), _loss_g_face=torch.zeros(1),
                               _loss_g_mask_smooth=torch.zeros(1))
    me._compute_loss_D = types.MethodType(T._compute_loss_D, me)
    me._compute_loss_smooth = types.MethodType(T._compute_loss_smooth, me)
    fake = T.forward(me)
    loss = T._optimize_G(me, *fake)
    G.zero_grad()
    loss.backward()
    o = dict(bg_both=True, mask_bce=True, vgg=vsd, lambda_mask=1.0, lambda_mask_smooth=1.0)
    total, terms, mine_fake = torch_ref.generator_train_loss({k: v.clone().requires_grad_(True) for k, v in gsd.items()}, dsd, b, o)
    assert abs(float(loss) - float(total)) < 1e-5 * max(1.0, float(total))
    assert abs(float(me._loss_g_tsf) - float(terms["g_tsf"])) < 1e-5 * max(1.0, float(terms["g_tsf"]))
    assert abs(float(me._loss_g_mask) - float(terms["g_mask"])) < 1e-5 * max(1.0, float(terms["g_mask"]))
    for a, c in zip(fake, mine_fake):
        assert torch.allclose(a, c, atol=1e-5, rtol=1e-5)
    _, grads, _ = torch_ref.generator_train_steps(gsd, dsd, [b], o)
    for k, p in G.named_parameters():
        assert torch.allclose(p.grad, grads[k], atol=1e-6 + 1e-4 * float(p.grad.abs().max()), rtol=0), k


def test_style_loss_matches_reference(ref):
    """oracle style_loss == the reference's StyleLoss over its own Vgg19 class (seeded weights), value and gradient."""
    vsd = helpers.vgg19_state_dict(seed=4)
    ref.networks.models.vgg19 = lambda pretrained=True: _torchvision_vgg19(vsd)
    crit = ref.networks.StyleLoss(feat_extractors=ref.networks.Vgg19())
    g = torch.Generator().manual_seed(12)
    x = (torch.rand(2, 3, 96, 96, generator=g) * 2 - 1).requires_grad_(True)
    y = torch.rand(2, 3, 96, 96, generator=g) * 2 - 1
    theirs = crit(x, y)
    theirs.backward()
    x2 = x.detach().clone().requires_grad_(True)
    mine = torch_ref.style_loss(vsd, x2, y)
    mine.backward()
    assert abs(float(theirs) - float(mine)) < 1e-6 * max(1.0, float(mine))
    assert torch.allclose(x.grad, x2.grad, atol=1e-9 + 1e-5 * float(x.grad.abs().max()), rtol=0)


def test_face_loss_matches_reference(ref, tmp_path):
    """oracle face_loss == the reference's FaceLoss (networks/networks.py:211-312) + Sphere20a (networks/facenet.py), loaded
    the reference's way from a .pth with seeded weights (the real file is a download), value and gradient."""
    fsd = helpers.sphere20a_state_dict(seed=6)
    path = str(tmp_path / "sphere20a_seeded.pth")
    torch.save(dict(fsd, **{"fc6.weight": torch.zeros(4, 512)}), path)     # load_sphere_model drops fc6.*
    crit = ref.networks.FaceLoss(pretrained_path=path)
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(2, 3, 128, 128, generator=g) * 2 - 1).requires_grad_(True)
    y = torch.rand(2, 3, 128, 128, generator=g) * 2 - 1
    bbox = torch.tensor([[30, 90, 10, 70], [44, 101, 3, 58]])
    theirs = crit(x, y, bbox1=bbox, bbox2=bbox)
    theirs.backward()
    x2 = x.detach().clone().requires_grad_(True)
    mine = torch_ref.face_loss(fsd, x2, y, bbox)
    mine.backward()
    assert abs(float(theirs) - float(mine)) < 1e-6 * max(1.0, float(mine))
    assert torch.allclose(x.grad, x2.grad, atol=1e-9 + 1e-5 * float(x.grad.abs().max()), rtol=0)


def test_body_recovery_flow_restatement_matches_reference(ref):
    """BodyRecoveryFlow.forward (models/impersonator_trainer.py:44-87), the trainer's input preparation, run unbound on a
    stub that carries the reference's own SMPLRenderer methods and the CPU SMPL (batch of 2 source / target pairs)."""
    from impersonator_amd.networks.batch_smpl import HumanModelRecovery, synthetic_smpl_params
    from impersonator_amd import demo
    s = helpers.scene()
    R, B = ref.nmr.SMPLRenderer, ref.trainer.BodyRecoveryFlow
    rs = types.SimpleNamespace(faces=helpers.t(s["faces"]), image_size=256, map_fn=helpers.t(s["map_fn"]),
                               proj_func=ref.nmr.orthographic_proj_withz_idrot,
                               eye=[0, 0, -(1. / np.tan(np.radians(30)) + 1)])
    render = types.SimpleNamespace(
        render_fim_wim=lambda cam, verts: R.render_fim_wim(rs, cam, verts),
        encode_fim=lambda cam, verts, fim=None, transpose=True: R.encode_fim(rs, cam, verts, fim=fim, transpose=transpose),
        cal_bc_transform=lambda a, b, c: R.cal_bc_transform(rs, a, b, c))
    hmr = HumanModelRecovery(smpl_params=synthetic_smpl_params(0))
    stub = types.SimpleNamespace(_hmr=hmr, _render=render, _opt=types.SimpleNamespace(bg_both=False, image_size=256))
    stub.cal_head_bbox = types.MethodType(B.cal_head_bbox, stub)
    stub.cal_body_bbox = types.MethodType(B.cal_body_bbox, stub)
    smpls = torch.from_numpy(demo.synthetic_smpls(64, seed=2))
    src_smpl, ref_smpl = smpls[[3, 20]], smpls[[40, 55]]
    gen = torch.Generator().manual_seed(1)
    src_img, ref_img = torch.rand(2, 3, 256, 256, generator=gen) * 2 - 1, torch.rand(2, 3, 256, 256, generator=gen) * 2 - 1
    with torch.no_grad():
        theirs = B.forward(stub, src_img, ref_img, src_smpl, ref_smpl)
        mine = torch_ref.body_recovery_flow(hmr.get_details, helpers.t(s["faces"]), helpers.t(s["map_fn"]), src_img, ref_img,
                                            src_smpl, ref_smpl)
    assert len(theirs) == len(mine) == 9
    for i, (a, b) in enumerate(zip(theirs, mine)):
        if a is None:
            assert b is None
            continue
        assert a.shape == b.shape and a.dtype == b.dtype, i
        assert torch.equal(a, b), "output %d differs by %g" % (i, float((a.float() - b.float()).abs().max()))


def test_smpl_restatement_matches_reference_in_fp32_and_fp64(ref):
    """oracle/torch_ref.py::smpl_forward / get_details == the reference's SMPL.forward (networks/batch_smpl.py:285-375) and
    HumanModelRecovery.get_details (networks/hmr.py:302-330) BIT FOR BIT, in fp32 (the reference's arithmetic) and with the
    reference's own code running on float64 tensors (the `compensated` mode's oracle)."""
    import types
    from impersonator_amd import demo
    from impersonator_amd.networks.batch_smpl import SMPL, synthetic_smpl_params
    m = SMPL(params=synthetic_smpl_params(0))
    th = torch.from_numpy(demo.synthetic_smpls(6, 0))
    th[0, 3:75] = 0
    for dt in (torch.float32, torch.float64):
        stub = types.SimpleNamespace(shapedirs=m.shapedirs.to(dt), v_template=m.v_template.to(dt), size=m.size,
                                     J_regressor=m.J_regressor.to(dt), posedirs=m.posedirs.to(dt), parents=m.parents,
                                     weights=m.weights.to(dt), joint_regressor=m.joint_regressor.to(dt), rotate=False)
        rv, rj, rR = ref.batch_smpl.SMPL.forward(stub, th[:, 75:].contiguous().to(dt), th[:, 3:75].contiguous().to(dt), get_skin=True)
        sm = torch_ref.smpl_tensors(m, dt)
        ov, oj, oR = torch_ref.smpl_forward(sm, th[:, 75:], th[:, 3:75])
        assert rv.dtype == dt and torch.equal(rv, ov) and torch.equal(rj, oj) and torch.equal(rR, oR), dt
    hmr = types.SimpleNamespace(smpl=lambda beta, theta, get_skin=False: ref.batch_smpl.SMPL.forward(stub32, beta, theta, get_skin=get_skin))
    stub32 = types.SimpleNamespace(shapedirs=m.shapedirs, v_template=m.v_template, size=m.size, J_regressor=m.J_regressor,
                                   posedirs=m.posedirs, parents=m.parents, weights=m.weights, joint_regressor=m.joint_regressor, rotate=False)
    want = ref.networks.HumanModelRecovery.get_details(hmr, th)
    got = torch_ref.get_details(torch_ref.smpl_tensors(m), th)
    for k in ("theta", "cam", "pose", "shape", "verts", "j2d", "j3d"):
        assert torch.equal(want[k], got[k]), k
