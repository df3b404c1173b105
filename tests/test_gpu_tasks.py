"""GPU end-to-end: Swapper (appearance transfer, BASELINE config 4) and Viewer (novel view) against outputs of the
reference's own methods (tests/golden/tasks_golden.npz); BGNet and the three-stream generator pass vs the CPU oracle."""
import numpy as np
import pytest
import torch

from impersonator_amd import demo
from tests import helpers

pytestmark = pytest.mark.gpu


def _subject_hmr(sc, names):
    """Fixed posed vertices in call order (tests/helpers.FixedHMR): the golden file pins everything after the SMPL stage."""
    t = lambda a: torch.from_numpy(a).cuda()
    return helpers.FixedHMR([(t(sc["cam_" + n]), t(sc["verts_" + n])) for n in names])


def test_swapper_matches_the_reference_golden():
    """Product Swapper.swap_setup + swap (BASELINE config 4) against tests/golden/tasks_golden.npz = the reference's own
    Swapper.personalize / swap / calculate_trans / forward run unbound (make_golden.py::make_tasks; the CPU oracle
    reproduces the same file in tests/test_oracle_tasks_golden.py)."""
    g, sc = helpers.golden("tasks_golden.npz"), helpers.task_scene()
    sw, _, _, _ = demo.build_synthetic_imitator(batch_size=1, seed=0, affine="random", model="swapper")
    sw.hmr = _subject_hmr(sc, "ab")
    smpl = np.zeros(85, np.float32)
    sw.swap_setup(sc["img_a"][0], sc["img_b"][0], src_smpl=smpl, tgt_smpl=smpl)     # --bg_model ORIGINAL: BGNet inpaints
    A, B = sw.src_info, sw.tsf_info
    assert A["part"].shape == (1, 11, 256, 256)
    assert np.array_equal(A["fim"].cpu().numpy(), g["fim_a"]) and np.array_equal(B["fim"].cpu().numpy(), g["fim_b"])
    assert np.array_equal(A["part"].argmax(1).cpu().numpy(), g["part_a"])
    assert np.abs(A["bg"].cpu().numpy()[:, :, ::4, ::4] - g["bg_a_sub"]).max() < 1e-3
    assert np.abs(B["bg"].cpu().numpy()[:, :, ::4, ::4] - g["bg_b_sub"]).max() < 1e-3
    seen = {}
    forward = sw.forward
    sw.forward = lambda x, *a: (seen.setdefault("x", x.clone()), forward(x, *a))[1]
    preds = sw.swap(A, B, target_part="body")
    assert preds.shape == (1, 3, 256, 256)
    assert np.array_equal(sw.T12.cpu().numpy(), g["T11"])
    assert np.abs(sw.T21.cpu().numpy() - g["T21"]).max() <= 1e-6
    assert np.abs(seen["x"].cpu().numpy()[:, :, ::2, ::2] - g["tsf_inputs_sub"]).max() <= 1e-5
    err = float(np.abs(preds.cpu().numpy() - g["swap_preds"]).max())
    assert err <= 1e-3, err


def test_viewer_matches_the_reference_golden():
    """Product Viewer.view against the reference's own Viewer.view / rotate_trans / forward (same golden file).  The
    rotated mesh is compared on its own (a matrix product: 1e-6), then view() runs from the golden mesh so that the
    face-index maps are those of the same vertices."""
    g, sc = helpers.golden("tasks_golden.npz"), helpers.task_scene()
    vw, _, _, _ = demo.build_synthetic_imitator(batch_size=1, seed=0, affine="random", model="viewer")
    vw.hmr = _subject_hmr(sc, "a")
    vw.personalize(sc["img_a"][0], src_smpl=np.zeros(85, np.float32))
    rotate = vw.rotate_trans
    for i, (rt, tr, replace) in enumerate(sc["views"]):
        mesh = torch.from_numpy(g["view%d_mesh" % i]).cuda()
        assert float((rotate(rt, tr, vw.src_info["verts"]) - mesh).abs().max()) <= 1e-6
        vw.rotate_trans = lambda rt, t, X, m=mesh: m
        vw._opt.bg_replace = replace
        preds = vw.view(rt, tr)
        err = float(np.abs(preds.cpu().numpy() - g["view%d_preds" % i]).max())
        assert err <= 1e-3, (i, err)


def test_bgnet_original_background_model():
    """--bg_model ORIGINAL: Imitator.personalize takes the background from the generator's own BGNet
    (models/imitator.py:30-34,126-132; networks/generator.py:23-65)."""
    import torch
    from impersonator_amd import demo
    from oracle import torch_ref
    imitator, src_smpl, src_img, _ = demo.build_synthetic_imitator(batch_size=2, seed=0, affine="random", image_size=128)
    assert imitator.bgnet is imitator.generator.bg_model
    sd = {k: v.detach().cpu() for k, v in imitator.generator.state_dict().items()}
    x = torch.rand(2, 4, 128, 128, generator=torch.Generator().manual_seed(4)) * 2 - 1
    out = imitator.generator.bg_model(x.cuda()).cpu()
    with torch.no_grad():
        ref = torch_ref.bgnet_forward(sd, x)
    assert out.shape == ref.shape and float((out - ref).abs().max()) < 1e-3
    imitator.personalize(src_img, src_smpl=src_smpl)          # no bg_img: BGNet inpaints
    bg = imitator.src_info['bg'].cpu()
    si = imitator.src_info
    bg_mask = torch_ref.morph(si['cond'][:, -1:].cpu(), imitator._opt.bg_ks, "erode")
    img = torch.from_numpy(src_img)[None]
    with torch.no_grad():
        ref_bg = torch_ref.bgnet_forward(sd, torch.cat([img * bg_mask, bg_mask], 1))
    assert float((bg - ref_bg).abs().max()) < 1e-3


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_generator_forward_and_infer_front(precision):
    """ImpersonatorGenerator.forward / infer_front (generator.py:204-243): per-sample sources, both streams decoded,
    BGNet -- the trainer's generator pass (impersonator_trainer.py:331-333), inference only."""
    import torch
    from impersonator_amd.networks.generator import ImpersonatorGenerator
    from oracle import torch_ref
    from tests import helpers
    sd = torch_ref.state_dict_from_numpy(helpers.generator_state_dict(seed=2, affine="random"))
    G = ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6, repeat_num=6, image_size=128, max_batch=2, precision=precision)
    G.load_state_dict(sd)
    G = G.cuda()
    gen = torch.Generator().manual_seed(6)
    bg = torch.rand(2, 4, 128, 128, generator=gen) * 2 - 1
    src = torch.rand(2, 6, 128, 128, generator=gen) * 2 - 1
    tsf = torch.rand(2, 6, 128, 128, generator=gen) * 2 - 1
    T = torch.rand(2, 128, 128, 2, generator=gen) * 2.4 - 1.2
    T[0, 40:80, 20:60] = -2
    outs = G(bg.cuda(), src.cuda(), tsf.cuda(), T.cuda())
    with torch.no_grad():
        ref = torch_ref.generator_forward(sd, bg, src, tsf, T)
    for name, a, b in zip(("img_bg", "src_img", "src_mask", "tsf_img", "tsf_mask"), outs, ref):
        assert a.shape == b.shape and float((a.cpu() - b).abs().max()) < 1e-3, name
    # the two samples have different sources: swapping the sources must change the tsf output
    outs2 = G.infer_front(src.flip(0).cuda(), tsf.cuda(), T.cuda())
    assert float((outs2[2] - outs[3]).abs().max()) > 1e-3
    G.release()


def test_swap_launches_no_framework_kernel_and_replays_as_one_graph():
    """Swapper.swap (models/swapper.py:198-239) with its mask bookkeeping as liblwg kernels (lwg_swap_masks, lwg_mask_faces,
    lwg_swap_compose, lwg_clamp): under the torch profiler every device record of a swap is liblwg's (no ATen elementwise / index /
    cat kernel, no copy of an index list, no read-back); the same call captured once (`swap_graph`) replays to the same bits; and
    `calculate_trans` with the reference's own signature (bool mask, face-id list) returns the fields `swap` used."""
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    from impersonator_amd.utils import synthetic
    sw, smpl_a, img_a, bg_a = demo.build_synthetic_imitator(batch_size=1, seed=0, image_size=256, model="swapper")
    smpl_b = demo.synthetic_smpls(8, seed=3)[5]
    img_b = synthetic.smooth_image(77, (1, 3, 256, 256))[0]
    sw.swap_setup(img_a, img_b, src_smpl=smpl_a, tgt_smpl=smpl_b, src_bg=bg_a, tgt_bg=synthetic.smooth_image(78, (1, 3, 256, 256))[0])
    A, B = sw.src_info, sw.tsf_info
    eager = sw.swap(A, B, target_part="body").clone()
    T11, T21 = sw.T12.clone(), sw.T21.clone()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        again = sw.swap(A, B, target_part="body")
        torch.cuda.synchronize()
    assert torch.equal(again, eager)
    records = [e.name for e in prof.events() if e.device_type == DeviceType.CUDA]
    foreign = sorted({k for k in records if "lwg" not in k})
    print("swap: %d device records, others: %s" % (len(records), foreign))
    assert len(records) >= 40 and not foreign, foreign
    # the reference-signature entry point gives the same two fields
    left_ids = [i for i in sw.PART_IDS['all'] if i not in sw.PART_IDS['body']]
    left_mask = torch.sum(A['part'][:, left_ids], dim=1).bool()
    left_faces = sorted(set(f for i in left_ids for f in sw.part_faces[i]))
    t11, t21 = sw.calculate_trans(left_mask, left_faces)
    assert torch.equal(t11, T11) and torch.equal(t21, T21)
    ref11 = sw.grid.clone()
    ref11[~left_mask[0]] = -2
    assert torch.equal(T11[0], ref11)
    # one HIP graph launch per swap
    run = sw.swap_graph(A, B, target_part="body")
    for _ in range(3):
        out = run()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)
