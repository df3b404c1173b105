"""GPU end-to-end: the Imitator task model (reference API) on the synthetic configuration vs the CPU oracle."""
import numpy as np
import pytest
import torch

from impersonator_amd import demo
from oracle import torch_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def imi():
    imitator, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=4, seed=0, affine="random")
    imitator.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
    return imitator, src_smpl, src_img, bg_img


def _oracle_frames(imitator, src_img, bg_img, verts, cam):
    sd = {k: v.detach().cpu() for k, v in imitator.generator.state_dict().items()}
    faces_t, map_fn, si = imitator.render.faces.cpu(), imitator.render.map_fn.cpu(), imitator.src_info
    with torch.no_grad():
        sf2v, sfim, _ = torch_ref.render_fim_wim(si["cam"].cpu(), si["verts"].cpu(), faces_t)
        p2v = torch_ref.source_p2verts(sf2v)
        scond = torch_ref.encode_fim(sfim, map_fn)
        ft = 1 - torch_ref.morph(scond[:, -1:], imitator._opt.ft_ks, "erode")
        src = torch.from_numpy(src_img)[None]
        enc, res = torch_ref.encode_src(sd, torch.cat([src * ft, scond], 1))
        fr = torch_ref.transfer_frame(src, p2v, cam, verts, faces_t, map_fn)
        pred = torch_ref.imitator_forward(sd, enc, res, torch.from_numpy(bg_img)[None], fr["tsf_inputs"], fr["T"])[0]
    return fr, pred


def test_personalize_fills_src_info(imi):
    imitator, _, src_img, _ = imi
    si = imitator.src_info
    for k in ("theta", "cam", "pose", "shape", "verts", "j2d", "j3d", "fim", "wim", "cond", "f2verts", "p2verts",
              "img", "bg", "feats"):
        assert k in si, k
    assert si["fim"].shape == (1, 256, 256) and si["cond"].shape == (1, 3, 256, 256)
    # hazard H9: p2verts aliases f2verts, so the y flip is visible through both
    assert si["p2verts"].data_ptr() == si["f2verts"].data_ptr()
    enc, res = si["feats"]
    assert [tuple(t.shape) for t in enc] == [(1, 64, 256, 256), (1, 128, 128, 128), (1, 256, 64, 64), (1, 512, 32, 32)]
    assert len(res) == 6 and all(tuple(t.shape) == (1, 512, 32, 32) for t in res)


def test_inference_by_smpls_matches_oracle_and_is_batch_invariant(imi):
    imitator, _, src_img, bg_img = imi
    smpls = demo.synthetic_smpls(64, seed=0)[[0, 9, 17, 30, 45, 63]]
    outs = imitator.inference_by_smpls(smpls, cam_strategy="smooth")           # batches of 4 + 2
    assert len(outs) == 6 and outs[0].shape == (256, 256, 3) and outs[0].dtype == np.float32
    # the reference processes one frame at a time: same numbers
    imitator._opt.batch_size = 1
    outs1 = imitator.inference_by_smpls(smpls, cam_strategy="smooth")
    imitator._opt.batch_size = 4
    # every stage (SMPL skinning, rasteriser, generator) is batch-invariant bit for bit
    for a, b in zip(outs, outs1):
        assert np.array_equal(a, b)
    # oracle on the vertices of the last launch (tsf_info belongs to the batch-1 pass: frame 63 only)
    info = imitator.tsf_info
    n = info["verts"].shape[0]
    fr, pred = _oracle_frames(imitator, src_img, bg_img, info["verts"].cpu(), info["cam"].cpu())
    assert torch.equal(fr["fim"], info["fim"].cpu())
    got = np.stack(outs[-n:]).transpose(0, 3, 1, 2)
    assert np.abs(got - pred.numpy()).max() <= 1e-3
    for k in ("theta", "cam", "pose", "shape", "verts", "j2d", "j3d", "fim", "wim", "cond", "tsf_img", "T"):
        assert k in info, k


def test_empty_sequence_off_screen_and_full_frame_bodies(imi):
    """Edge cases of the sequence drivers: no target frames at all; a target whose body is entirely outside the image (every
    face-index pixel -1: cond is the background row, T the -2 sentinel, the warped source zero -- the generator still runs);
    a body that covers most of the frame (camera scale 4: faces dozens of pixels wide); a ragged last batch.
    All against the CPU oracle on the same vertices, 'copy' strategy so that the camera is the target's own."""
    imitator, _, src_img, bg_img = imi
    assert imitator.inference_by_smpls(np.zeros((0, 85), np.float32), cam_strategy="copy") == []
    smpls = demo.synthetic_smpls(64, seed=0)[[3, 20, 41, 50, 60]].copy()      # batches of 4 + 1
    smpls[1, 0:3] = (1.0, 5.0, 0.0)       # translated five image half-widths away: nothing on screen
    smpls[3, 0:3] = (4.0, 0.0, 0.0)       # scaled up: the torso covers most of the frame
    outs = imitator.inference_by_smpls(smpls, cam_strategy="copy")
    assert len(outs) == 5
    imitator.transfer_params_by_smpl(torch.from_numpy(smpls).cuda(), cam_strategy="copy", t=1)   # the geometry of all five at once
    info = imitator.tsf_info
    fim = info["fim"].cpu()
    assert bool((fim[1] == -1).all()) and bool((info["T"][1] == -2).all()) and bool((info["tsf_img"][1] == 0).all())
    assert int((fim[3] == -1).sum()) < 0.3 * fim[3].numel(), "the scaled-up body should cover most of the frame"
    fr, pred = _oracle_frames(imitator, src_img, bg_img, info["verts"].cpu(), info["cam"].cpu())
    assert torch.equal(fr["fim"], fim) and float((fr["T"] - info["T"].cpu()).abs().max()) <= 1e-6
    got = np.stack(outs).transpose(0, 3, 1, 2)
    err = np.abs(got - pred.numpy()).reshape(5, -1).max(1)
    assert err.max() <= 1e-3, err


def test_reference_call_sequence_and_camera_strategies(imi):
    imitator, _, _, _ = imi
    smpls = demo.synthetic_smpls(64, seed=0)
    for strat in ("smooth", "source", "copy"):
        tsf_inputs = imitator.transfer_params_by_smpl(smpls[5], cam_strategy=strat, t=0)   # one (85,) vector
        assert tsf_inputs.shape == (1, 6, 256, 256)
        preds = imitator.forward(tsf_inputs, imitator.tsf_info["T"])
        assert preds.shape == (1, 3, 256, 256) and bool(torch.isfinite(preds).all())
        assert float(preds.abs().max()) <= 1.0 + 1e-5
    cam = imitator.tsf_info["cam"]
    assert torch.allclose(cam.cpu(), torch.from_numpy(smpls[5:6, :3]))               # 'copy' keeps the target camera


def test_swap_smpl_and_get_details_as_kernels_equal_the_tensor_expressions(imi):
    """lwg_smpl_swap + lwg_smpl_project_joints (the per-frame prelude's glue as two launches) against Imitator.swap_smpl +
    HumanModelRecovery.get_details (models/imitator.py:216-234, networks/hmr.py:302-330), bit for bit, every strategy."""
    imitator, _, _, _ = imi
    smpls = torch.from_numpy(demo.synthetic_smpls(64, seed=0)).cuda()
    imitator.first_cam = smpls[0:1, 0:3].clone()
    si = imitator.src_info
    from impersonator_amd.networks.batch_smpl import batch_orth_proj_idrot
    for strat in ("smooth", "source", "copy"):
        theta = imitator.swap_smpl(si["cam"], si["shape"], smpls[8:24], cam_strategy=strat)       # tensor expressions
        verts, j3d, _ = imitator.hmr.smpl.forward_theta(theta)
        cam = theta[:, 0:3].contiguous()
        ref = dict(theta=theta, cam=cam, pose=theta[:, 3:75].contiguous(), shape=theta[:, 75:].contiguous(), verts=verts, j3d=j3d,
                   j2d=batch_orth_proj_idrot(j3d, cam))                                             # hmr.py:302-330
        got = imitator.hmr.get_details_swapped(smpls[8:24], si["cam"], si["shape"], imitator.first_cam, strat)
        assert set(ref) == set(got)
        for k in ref:
            assert torch.equal(ref[k], got[k]), (strat, k)
        # get_details on a CUDA theta takes the same launches with the vector as it is
        same = imitator.hmr.get_details(theta)
        for k in ref:
            assert torch.equal(ref[k], same[k]), (strat, k)


def test_front_warp(imi):
    imitator, _, _, _ = imi
    smpls = demo.synthetic_smpls(64, seed=0)
    x = imitator.transfer_params_by_smpl(smpls[3:5], t=3)
    base = imitator.forward(x, imitator.tsf_info["T"])
    imitator._opt.front_warp = True
    try:
        warped = imitator.forward(x, imitator.tsf_info["T"])
    finally:
        imitator._opt.front_warp = False
    fm = imitator.render.encode_front_fim(imitator.tsf_info["fim"], transpose=True, front_fn=True)
    assert fm.shape == (2, 1, 256, 256)
    same = (fm == 0).expand_as(base)
    assert torch.equal(warped[same], base[same])


@pytest.mark.parametrize("precision", ["fp32", "compensated"])
def test_smpl_device_kernels_match_tensor_op_formulation(precision):
    """smpl.hip (fused LBS) vs the reference's formulation evaluated on the CPU (oracle/torch_ref.py::smpl_forward == the
    reference's SMPL.forward in fp32 AND in fp64, tests/test_oracle_vs_reference.py).  `fp32`: the reference's arithmetic,
    another summation order: 1e-5.  `compensated`: every intermediate in fp64, one rounding -- must equal the fp64
    evaluation rounded to fp32 BIT FOR BIT (up to a rounding-boundary case in a million values)."""
    from impersonator_amd.networks.batch_smpl import SMPL, synthetic_smpl_params
    from oracle import torch_ref
    m = SMPL(params=synthetic_smpl_params(0))
    m.precision = precision
    g = torch.Generator().manual_seed(0)
    beta = torch.randn(5, 10, generator=g)
    theta = torch.randn(5, 72, generator=g) * 0.4
    theta[0] = 0                                     # rest pose: the 1e-8 guard of batch_rodrigues matters here
    v, j, Rs = torch_ref.smpl_forward(torch_ref.smpl_tensors(m), beta, theta)
    assert torch.equal(v, m.forward_ops(beta, theta, get_skin=True)[0])     # the module's CPU form is the same expression
    v64, j64, Rs64 = (x.float() for x in torch_ref.smpl_forward(torch_ref.smpl_tensors(m, torch.float64), beta, theta))
    md = m.cuda()
    dv, dj, dRs = md(beta.cuda(), theta.cuda(), get_skin=True)
    if precision == "fp32":
        # fp32 sums over 207 pose-blend terms and 6890-vertex regressions, different association order
        assert float((dv.cpu() - v).abs().max()) <= 1e-5
        assert float((dj.cpu() - j).abs().max()) <= 2e-5
        assert float((dRs.cpu() - Rs).abs().max()) <= 1e-6
    else:
        ndiff = int((dv.cpu() != v64).sum())
        print("compensated SMPL vs fp64-rounded oracle: %d of %d vertex coordinates differ, max %.3g; fp32 oracle is %.3g away"
              % (ndiff, v64.numel(), float((dv.cpu() - v64).abs().max()), float((v - v64).abs().max())))
        assert ndiff <= 2 and float((dv.cpu() - v64).abs().max()) <= 1.2e-7
        assert float((dRs.cpu() - Rs64).abs().max()) <= 1.2e-7
        assert float((dj.cpu() - j64).abs().max()) <= 2e-7       # fp64 sums of the ROUNDED vertices here, of the fp64 ones there
    # batch-size / batch-position invariance of the device kernels, bit for bit (the vertex kernel handles four frames
    # per lane: every position of a group, and a second group, must give the numbers of a batch of one)
    for i in range(5):
        dv1, dj1, _ = md(beta[i:i + 1].cuda(), theta[i:i + 1].cuda(), get_skin=True)
        assert torch.equal(dv1, dv[i:i + 1]) and torch.equal(dj1, dj[i:i + 1]), "frame %d depends on its batch" % i


@pytest.mark.parametrize("lanes", [1, 2])
def test_stream_pipeline_equals_the_sequential_path(imi, lanes):
    """Imitator.predict_batches enqueues the geometry of batch i+1 on a side stream and deals the generators of
    consecutive batches to `lanes` engines on their own streams; every batch must come out, in order, bit-identical to
    transfer_params_by_smpl + forward run one after the other."""
    imitator = imi[0]
    smpls = torch.from_numpy(demo.synthetic_smpls(24, seed=3)).cuda()
    imitator.first_cam = smpls[0:1, 0:3].clone()
    chunks = [(smpls[s:s + 4], s) for s in range(0, 24, 4)]
    seq = []
    for chunk, t in chunks:
        x = imitator.transfer_params_by_smpl(chunk, "smooth", t=t)
        seq.append(imitator.forward(x, imitator.tsf_info["T"]).clone())
    for rep in range(3):   # repeated: a race would not show every time
        got = []
        for t, p in imitator.predict_batches(iter(chunks), "smooth", lanes=lanes):
            assert imitator.tsf_info["T"].shape[0] == p.shape[0]   # tsf_info is the yielded batch's
            got.append((t, p.clone()))
        assert [t for t, _ in got] == [t for _, t in chunks]
        for (_, p), q in zip(got, seq):
            assert torch.equal(p, q)
    assert list(imitator.predict_batches(iter([]), "smooth", lanes=lanes)) == []
    assert len(list(imitator.predict_batches(iter(chunks[:1]), "smooth", lanes=lanes))) == 1


@pytest.mark.parametrize("depth", [1, 4])
def test_lane_pipeline_stress(imi, depth):
    """Thirty passes of the two-lane pipeline with a consumer that never synchronises (tools/lane_stress.py in small), at
    rounds of one batch per lane and at the default depth: every batch bit-identical to the sequential path."""
    imitator = imi[0]
    smpls = torch.from_numpy(demo.synthetic_smpls(24, seed=5)).cuda()
    imitator.first_cam = smpls[0:1, 0:3].clone()
    chunks = [(smpls[s:s + 4], s) for s in range(0, 24, 4)]
    seq = []
    for chunk, t in chunks:
        x = imitator.transfer_params_by_smpl(chunk, "smooth", t=t)
        seq.append(imitator.forward(x, imitator.tsf_info["T"]).clone())
    keep = imitator.round_depth
    imitator.round_depth = depth
    try:
        for _ in range(30):
            got = [p.clone() for _, p in imitator.predict_batches(iter(chunks), "smooth", lanes=2)]
            torch.cuda.synchronize()
            for p, q in zip(got, seq):
                assert torch.equal(p, q)
    finally:
        imitator.round_depth = keep


def test_predict_batches_launches_no_framework_kernel():
    """The timed step of bench.py (Imitator.predict_batches over adjacent row blocks of one SMPL tensor) under the torch profiler:
    every device record is a liblwg kernel -- no ATen kernel (`torch.cat` of the round's chunks used to be one), no runtime
    fill / copy kernel (zero-fills belong to the first writer, not to a hipMemsetAsync)."""
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    imitator, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=8, seed=0, image_size=256)
    imitator.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
    smpls = torch.from_numpy(demo.synthetic_smpls(256, seed=0)).cuda()
    imitator.first_cam = smpls[0:1, 0:3].clone()
    chunks = lambda: ((smpls[s:s + 8], s) for s in range(8, 8 + 16 * 8, 8))     # 16 batches = two rounds of two lanes x depth 4
    for _ in imitator.predict_batches(chunks()):          # first pass: lanes, replicas, scratch, lazy kernel attributes
        pass
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        n = 0
        for _, preds in imitator.predict_batches(chunks()):
            n += preds.shape[0]
        torch.cuda.synchronize()
    assert n == 128
    records = [e.name for e in prof.events() if e.device_type == DeviceType.CUDA]
    ours = [k for k in records if "lwg" in k]
    foreign = sorted({k for k in records if "lwg" not in k})
    print("%d device records, %d liblwg kernels; others: %s" % (len(records), len(ours), foreign))
    # 16 batches = four generator launch sequences of 32 frames (Imitator.fuse = 4, ~55 launches each) + two rounds of geometry
    assert len(ours) >= 4 * 50, "the profiler saw too few liblwg kernels: %s" % sorted(set(records))[:10]
    assert not [k for k in foreign if "at::" in k or "elementwise" in k or "Cat" in k], foreign
    assert not [k for k in foreign if "fill" in k.lower() or "memset" in k.lower() or "copy" in k.lower()], foreign


@pytest.mark.parametrize("batch", [1, 2])
def test_frame_graph_replay_equals_the_eager_frame_loop(batch):
    """Imitator.frame_graph (one HIP-graph launch per call, the latency form of the reference's per-frame loop,
    models/imitator.py:166-171) returns what transfer_params_by_smpl + forward return, bit for bit, frame after frame,
    including the camera reference taken at t == 0."""
    imitator, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=batch, seed=0, image_size=256)
    imitator.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
    smpls = torch.from_numpy(demo.synthetic_smpls(64, seed=0)).cuda()
    eager = []
    for t in range(0, 6 * batch, batch):
        x = imitator.transfer_params_by_smpl(smpls[t:t + batch], "smooth", t=t)
        eager.append((imitator.forward(x, imitator.tsf_info["T"]).clone(), imitator.tsf_info["fim"].clone(),
                      imitator.tsf_info["T"].clone()))
    imitator.first_cam = None
    run = imitator.frame_graph(batch=batch)
    for i, t in enumerate(range(0, 6 * batch, batch)):
        p = run(smpls[t:t + batch], t=t)
        torch.cuda.synchronize()
        assert torch.equal(p, eager[i][0]), "frame %d" % t
        assert torch.equal(imitator.tsf_info["fim"], eager[i][1]) and torch.equal(imitator.tsf_info["T"], eager[i][2])
