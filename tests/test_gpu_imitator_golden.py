"""GPU: the product's Imitator against tests/golden/imitator_golden.npz = the reference's OWN `Imitator.personalize`,
`inference_by_smpls`, `transfer_params_by_smpl`, `swap_smpl`, `forward`, `warp_front` (models/imitator.py:82-155, 191-268,
326-342) run unbound (tests/golden/make_golden.py::make_imitator; the CPU oracle reproduces the same file in
tests/test_oracle_imitator_golden.py), for every variant of tests/helpers.py::IMITATOR_VARIANTS: --only_vis on/off,
--bg_model ORIGINAL / InpaintSANet, --front_warp on/off, cam_strategy smooth / source / copy, four frames from t = 0.

Two passes per variant:
  * `pinned SMPL`: the `hmr` is a stand-in that checks every SMPL vector the Imitator hands it against the golden's BIT FOR BIT
    (camera policy, `first_cam` at t == 0, source shape, batching) and answers with the golden's vertices -- everything
    downstream must then match exactly where it is integer (face-index maps, visible-face sets) and within the stated bounds
    where it is float (T 1e-6, images 1e-3);
  * `device SMPL`: the real path (lwg_smpl_swap + the SMPL kernels): theta bit for bit, vertices within 1e-5 of the golden's
    (the reference's fp32 SMPL.forward on 8 CPU threads, one frame per call).  The images are compared on the frames whose
    face-index maps agree, with the bounds the reference sets itself: its own SMPL run with another thread count or batch size
    moves T by 5e-4 and the image by 2.7e-3 and flips up to 2 face-index pixels (profiles/r04_theta_chain_reference_self.md);
    the bounds here are twice what the DEVICE shows against the golden (<= 1 flipped pixel, T 4.9e-4, image 6.4e-4), and the all-pixel
    figure with nothing excluded is printed beside them.  (Against the correctly rounded SMPL the default `compensated` device mode has no such
    slack: tests/test_gpu_bench_config.py::test_theta_to_image_chain_with_the_oracles_own_smpl.)"""
import numpy as np
import pytest
import torch

from impersonator_amd import demo
from tests import helpers

pytestmark = pytest.mark.gpu


class CheckedHMR(object):
    """`hmr` stand-in: get_details(theta) asserts theta == the golden's next rows exactly and returns the golden's cam / vertices."""

    def __init__(self, g, k):
        self.g, self.k, self.i, self.src_done = g, k, 0, False

    def cuda(self):
        return self

    def get_details(self, theta):
        g, k = self.g, self.k
        th = theta.detach().cpu().numpy()
        if not self.src_done:
            self.src_done = True
            assert np.array_equal(th, g[k + "src_theta"])
            cam, verts, j2d = g[k + "src_cam"], g[k + "src_verts"], np.zeros((1, 19, 2), np.float32)
        else:
            n = th.shape[0]
            assert np.array_equal(th, g[k + "theta"][self.i:self.i + n]), "frames %d..%d: swapped SMPL vectors differ" % (self.i, self.i + n)
            cam, verts, j2d = (g[k + key][self.i:self.i + n] for key in ("cam", "verts", "j2d"))
            self.i += n
        c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        return dict(theta=theta, cam=c(cam), pose=theta[:, 3:75].contiguous(), shape=theta[:, 75:].contiguous(), verts=c(verts),
                    j2d=c(j2d), j3d=torch.zeros(th.shape[0], 19, 3, device="cuda"))


def _build(v):
    size = v["size"]
    opt = demo.default_opt(batch_size=4, image_size=size, only_vis=v["only_vis"], front_warp=v["front_warp"])
    imitator, src_smpl, src_img, _ = demo.build_synthetic_imitator(batch_size=4, seed=0, image_size=size, affine="random", opt=opt)
    if v["bg_model"] != "ORIGINAL":
        from impersonator_amd.networks.inpaintor import InpaintSANet
        net = InpaintSANet(c_dim=4, image_size=size).eval()
        net.load_state_dict({kk: torch.from_numpy(x) for kk, x in helpers.inpaintor_state_dict(seed=1).items()})
        imitator.bgnet = net.cuda()
        imitator._opt.bg_model = v["bg_model"]
    return imitator, src_smpl, src_img


def _run(imitator, tgt_smpls, strategy, batch):
    """inference_by_smpls with the per-call tsf_info recorded; returns (frames (n,3,H,W), per-frame dict of tensors)."""
    imitator._opt.batch_size = batch
    imitator.first_cam = None
    calls, inner = [], imitator.transfer_params_by_smpl

    def recording(tgt_smpl, cam_strategy='smooth', t=0):
        x = inner(tgt_smpl, cam_strategy, t)
        calls.append({kk: vv.clone() for kk, vv in imitator.tsf_info.items() if torch.is_tensor(vv)})
        return x

    imitator.transfer_params_by_smpl = recording
    try:
        outs = imitator.inference_by_smpls(tgt_smpls, cam_strategy=strategy)
    finally:
        del imitator.transfer_params_by_smpl
    info = {kk: torch.cat([c[kk] for c in calls]).cpu() for kk in ("theta", "cam", "verts", "j2d", "fim", "T")}
    return np.stack(outs).transpose(0, 3, 1, 2), info


def _compare_images(v, g, k, T, preds, frames):
    """T / preds (of the frames listed) against the golden's full or sub-sampled entries -> (T error, image error)."""
    eT = eP = 0.0
    for i in frames:
        if v["size"] > 128 and i in (1, 2):
            gT, gP, mT, mP = g[k + "T_sub"][i - 1], g[k + "preds_sub"][i - 1], T[i][::2, ::2], preds[i][:, ::2, ::2]
        else:
            j = {0: 0, 3: 1}[i] if v["size"] > 128 else i
            gT, gP, mT, mP = g[k + "T_full"][j], g[k + "preds_full"][j], T[i], preds[i]
        eT, eP = max(eT, float(np.abs(mT - gT).max())), max(eP, float(np.abs(mP - gP).max()))
    return eT, eP


@pytest.mark.parametrize("name", list(helpers.IMITATOR_VARIANTS))
def test_imitator_methods_match_the_reference_golden(name):
    v, g, k = helpers.IMITATOR_VARIANTS[name], helpers.golden("imitator_golden.npz"), name + "/"
    sc = helpers.imitator_scene(v["size"])
    imitator, src_smpl, src_img = _build(v)
    assert np.array_equal(src_smpl, sc["src_smpl"]) and np.array_equal(src_img, sc["src_img"][0])
    device_hmr = imitator.hmr

    # ---- pinned SMPL
    imitator.hmr = CheckedHMR(g, k)
    imitator.personalize(src_img, src_smpl=src_smpl)
    si = imitator.src_info
    assert np.array_equal(si["fim"].cpu().numpy(), g[k + "src_fim"])
    assert np.array_equal(si["p2verts"].cpu().numpy(), g[k + "src_p2verts"])        # H9 (y flip through the view) and H10 (--only_vis)
    assert np.allclose(helpers.tensor_stat(si["f2verts"].cpu()), g[k + "src_f2verts_stat"], rtol=1e-9, atol=0)   # ... which mutated f2verts
    assert np.allclose(helpers.tensor_stat(si["cond"].cpu()), g[k + "src_cond_stat"], rtol=1e-9, atol=0)
    assert np.abs(si["bg"].cpu().numpy()[:, :, ::4, ::4] - g[k + "src_bg_sub"]).max() <= 1e-3
    for key, feats in (("src_enc_stat", si["feats"][0]), ("src_res_stat", si["feats"][1])):
        got = np.stack([helpers.tensor_stat(x.cpu()) for x in feats])
        assert np.allclose(got, g[k + key], rtol=2e-3, atol=1e-5), key
    preds, info = _run(imitator, sc["tgt_smpls"], v["cam_strategy"], batch=4)
    assert imitator.hmr.i == 4
    assert np.array_equal(info["fim"].numpy(), g[k + "fim"])
    fc = g[k + "first_cam"][-1]
    assert (imitator.first_cam is None and np.isnan(fc).all()) or np.array_equal(imitator.first_cam.cpu().numpy()[0], fc)
    eT, eP = _compare_images(v, g, k, info["T"].numpy(), preds, range(4))
    print("%s pinned SMPL: T %.2g, image %.3g" % (name, eT, eP))
    assert eT <= 1e-6 and eP <= 1e-3, (eT, eP)
    # the same sequence in batches of 3 + 1 and of 1: `first_cam` must still be frame 0's, results bit-identical
    for batch in (3, 1):
        imitator.hmr.i = 0
        p2, i2 = _run(imitator, sc["tgt_smpls"], v["cam_strategy"], batch=batch)
        assert np.array_equal(p2, preds) and torch.equal(i2["fim"], info["fim"]) and torch.equal(i2["T"], info["T"]), batch

    # ---- device SMPL
    imitator.hmr = device_hmr
    imitator.personalize(src_img, src_smpl=src_smpl)
    assert np.array_equal(imitator.src_info["theta"].cpu().numpy(), g[k + "src_theta"])
    assert float((imitator.src_info["verts"].cpu() - torch.from_numpy(g[k + "src_verts"])).abs().max()) <= 1e-5
    preds, info = _run(imitator, sc["tgt_smpls"], v["cam_strategy"], batch=4)
    assert np.array_equal(info["theta"].numpy(), g[k + "theta"])                    # lwg_smpl_swap: bit for bit
    assert np.array_equal(info["cam"].numpy(), g[k + "cam"])
    assert float((info["verts"] - torch.from_numpy(g[k + "verts"])).abs().max()) <= 1e-5
    assert float((info["j2d"] - torch.from_numpy(g[k + "j2d"])).abs().max()) <= 1e-5
    diff = (info["fim"].numpy() != g[k + "fim"]).reshape(4, -1).sum(1)
    src_diff = int((imitator.src_info["fim"].cpu().numpy() != g[k + "src_fim"]).sum())
    same = [i for i in range(4) if diff[i] == 0] if src_diff == 0 else []
    eT, eP = _compare_images(v, g, k, info["T"].numpy(), preds, same)
    # ... and over ALL pixels of ALL frames, excluding nothing (reported: a flipped face-index pixel is another face's flow at that
    # pixel, so where one flips the all-pixel figure is whatever the two faces' colours differ by -- there is no bound to assert)
    eT_all, eP_all = _compare_images(v, g, k, info["T"].numpy(), preds, range(4))
    print("%s device SMPL: face-index pixels differing per frame %s (source %d); on the %d identical frames T %.2g, image %.3g; "
          "all pixels of all 4 frames, none excluded: T %.3g, image %.3g"
          % (name, diff.tolist(), src_diff, len(same), eT, eP, eT_all, eP_all))
    # observed on MI355X (profiles/r04_theta_chain_gpu_tests.log, unchanged since): at most ONE flipped pixel in one frame, T <= 4.9e-4,
    # image <= 6.4e-4 -- asserted at twice that (the reference against itself, 1 thread vs 8: 5e-4 / 2.7e-3 / 2 pixels)
    assert diff.max() <= 1 and src_diff == 0, (diff.tolist(), src_diff)
    assert eT <= 1e-3 and eP <= 1.5e-3, (eT, eP)
    if diff.max() == 0:
        assert eT_all == eT and eP_all == eP
    imitator.generator.release()
