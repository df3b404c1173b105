"""GPU: the 1e-3 image bound on weights that are NOT the initialisation.

Every other parity test draws the generator from `init_weights()` (conv N(0, 0.02), networks/networks.py:54-65) with a mild random
InstanceNorm affine.  The default `bf16x3` arithmetic (two bf16 terms per operand, three MFMA products, fp32 accumulate) is narrower
than the reference's fp32 (networks/generator.py:80-133 computes in fp32 throughout); its 8e-5 on those weights leaves a 12x margin
to north_star's 1e-3 -- here the margin is measured where it could be smaller:

  * `trained_a`, `trained_b`: generators TRAINED by this repo's own trainer (impersonator_amd/models/impersonator_trainer.py =
    models/impersonator_trainer.py:350-418: G + D updates, Adam) for ITERS iterations on synthetic smooth images from two seeds --
    Adam-shaped weights, InstanceNorm affines that moved, a discriminator in the loop;
  * `wide_convs`: conv weights N(0, 0.1) (25x the initialisation's variance; activations before every InstanceNorm 5x larger);
  * `outlier_gamma`: InstanceNorm gamma log-uniform in [0.1, 10] with every 37th channel at 30 and beta N(0, 0.5): a few channels
    dominate every reduction, the rest sit 2-3 decimal orders below them.

Each set runs the product's Imitator (personalize + a batch of frames) at 256x256 and 512x512 in both arithmetics -- and under the
default `precision="auto"` policy, whose probe must send the sets bf16x3 is too narrow for to fp32 -- against the CPU oracle on the same posed vertices: face-index maps identical, image L-inf <= 1e-3, the margin printed
(the WEIGHTSET lines of `pytest -s` are kept in profiles/r06_weight_sets.md)."""
import numpy as np
import pytest
import torch

from impersonator_amd import demo
from impersonator_amd.utils import synthetic
from oracle import torch_ref
from tests import helpers

pytestmark = pytest.mark.gpu

ITERS = 240
BOUND = 1e-3
SETS = ("trained_a", "trained_b", "wide_convs", "outlier_gamma")
_TRAINED = {}


def _trained_state_dict(seed):
    """ITERS training iterations (256x256, batch 4, bf16x3 convolutions, a new synthetic batch every 8 iterations) from the seeded
    initialisation -> the generator's state_dict (numpy)."""
    if seed in _TRAINED:
        return _TRAINED[seed]
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_train
    model = bench_train.build(4, 256, precision="bf16x3", seed=seed)
    g = torch.Generator().manual_seed(1000 + seed)
    n, s = 4, 256

    def batch(k):
        img = lambda j: torch.from_numpy(synthetic.smooth_image(seed * 100000 + k * 16 + j, (n, 3, s, s))).cuda()
        cond = lambda j: torch.from_numpy(synthetic.smooth_image(seed * 100000 + k * 16 + 8 + j, (n, 3, s, s))).cuda()
        T = (torch.rand(n, s, s, 2, generator=g) * 2.4 - 1.2).cuda()
        mask = (torch.from_numpy(synthetic.smooth_image(seed * 100000 + k * 16 + 15, (2 * n, 1, s, s))) > 0).float().cuda()
        model.set_input(torch.cat([img(0), cond(0)], 1), img(1), input_G_bg=torch.cat([img(2), mask[:n]], 1),
                        input_G_src=torch.cat([img(3), cond(1)], 1), T=T, real_src=img(3), bg_mask=mask)

    losses = None
    for it in range(ITERS):
        if it % 8 == 0:
            batch(it // 8)
        losses = model.optimize_parameters()
    assert all(np.isfinite(v) for v in losses.values()), losses
    sd = {k: v.detach().cpu().numpy().copy() for k, v in model._generator_trainer().state_dict().items()}
    model._D.release()
    model._G.release()
    _TRAINED[seed] = sd
    return sd


def _weights(name):
    if name == "trained_a":
        return _trained_state_dict(0)
    if name == "trained_b":
        return _trained_state_dict(1)
    sd = helpers.generator_state_dict(seed=5, affine="random")
    rng = np.random.default_rng(77)
    for k, v in sd.items():
        if name == "wide_convs" and v.ndim == 4:
            sd[k] = (v * np.float32(5.0)).astype(np.float32)                     # N(0, 0.02) -> N(0, 0.1)
        elif name == "outlier_gamma" and v.ndim == 1 and k.endswith(".weight"):
            gmm = np.exp(rng.uniform(np.log(0.1), np.log(10.0), v.shape)).astype(np.float32)
            gmm[::37] = 30.0
            sd[k] = gmm
        elif name == "outlier_gamma" and v.ndim == 1 and k.endswith(".bias"):
            sd[k] = (rng.standard_normal(v.shape) * 0.5).astype(np.float32)
    return sd


def _distance_from_init(sd):
    """how far a trained set moved: relative L2 change of the conv weights against the same-seed initialisation is not available
    here (the trainer draws its own); report the spread of the conv weights and of gamma instead"""
    conv = np.concatenate([v.ravel() for v in sd.values() if v.ndim == 4])
    gam = np.concatenate([v.ravel() for k, v in sd.items() if v.ndim == 1 and k.endswith(".weight")])
    return float(conv.std()), float(np.abs(conv).max()), float(gam.min()), float(gam.max())


@pytest.mark.parametrize("size,frames", [(256, 8), (512, 2)])
@pytest.mark.parametrize("name", SETS)
def test_image_bound_holds_on_other_weight_sets(name, size, frames):
    sd = _weights(name)
    imitator, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=frames, seed=0, image_size=size)
    imitator.generator.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    smpls = torch.from_numpy(demo.synthetic_smpls(64, seed=0))[8:8 + frames].cuda()
    imitator.first_cam = torch.from_numpy(demo.synthetic_smpls(64, seed=0))[0:1, 0:3].cuda()
    errs, ref = {}, None
    for precision in ("bf16x3", "fp32"):
        imitator.generator.precision = precision
        imitator.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
        x = imitator.transfer_params_by_smpl(smpls, "smooth", t=8)
        info = {k: v.clone() for k, v in imitator.tsf_info.items() if torch.is_tensor(v)}
        pred = imitator.forward(x, info["T"]).cpu()
        if ref is None:
            gsd = {k: v.detach().cpu() for k, v in imitator.generator.state_dict().items()}
            faces_t, map_fn, si = imitator.render.faces.cpu(), imitator.render.map_fn.cpu(), imitator.src_info
            src_t, bg_t = torch.from_numpy(src_img)[None], torch.from_numpy(bg_img)[None]
            with torch.no_grad():
                src = torch_ref.personalize(gsd, src_t, si["cam"].cpu(), si["verts"].cpu(), faces_t, map_fn, ft_ks=imitator._opt.ft_ks,
                                            image_size=size)
                fr, ref = torch_ref.imitator_frames(gsd, src, src_t, bg_t, info["cam"].cpu(), info["verts"].cpu(), faces_t, map_fn,
                                                    image_size=size, chunk=2)
            assert torch.equal(src["fim"], si["fim"].cpu()) and torch.equal(fr["fim"], info["fim"].cpu())
        errs[precision] = float((pred - ref).abs().max())
    # the default policy: one probe frame through both arithmetics at personalize decides (ImpersonatorGenerator.auto_probe)
    imitator.generator.precision = "auto"
    imitator.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
    rep = imitator.generator.auto_report
    chosen = imitator.generator.precision
    assert rep is not None and rep["chosen"] == chosen and imitator.generator.precision_policy == "auto"
    cs, cm, g0, g1 = _distance_from_init(sd)
    print("WEIGHTSET %s %dx%d frames=%d: bf16x3 L-inf %.3g (margin %.1fx), fp32 L-inf %.3g | auto: probe |bf16x3 - fp32| %.3g -> %s, "
          "L-inf %.3g (margin %.1fx) | conv std %.3g max|w| %.3g gamma [%.3g, %.3g] | image range [%.2f, %.2f]"
          % (name, size, size, frames, errs["bf16x3"], BOUND / max(errs["bf16x3"], 1e-12), errs["fp32"], rep["linf_bf16x3_vs_fp32"],
             chosen, errs[chosen], BOUND / max(errs[chosen], 1e-12), cs, cm, g0, g1, float(ref.min()), float(ref.max())))
    imitator.generator.release()
    assert errs["fp32"] <= BOUND, (name, size, errs)
    assert errs["bf16x3"] <= BOUND, (name, size, errs)
    # under the default policy every set stays at least 2x inside the bound, and the narrow arithmetic is only kept where the
    # probe saw it agree with fp32
    assert errs[chosen] <= BOUND / 2, (name, size, chosen, errs)
    assert (chosen == "bf16x3") == (rep["linf_bf16x3_vs_fp32"] <= imitator.generator.AUTO_BOUND)
    if name.startswith("trained"):
        assert chosen == "bf16x3", rep
