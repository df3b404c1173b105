"""GPU parity of the training path's first slice: PatchGAN discriminator forward, LSGAN loss,
every parameter gradient and two Adam updates through the C ABI against the CPU oracle (torch autograd,
oracle/torch_ref.py, pinned to the reference's PatchDiscriminator in tests/test_oracle_vs_reference.py).
fp32 MFMA end to end, and the bf16x3 mode of the convolutions; tolerances are relative to each tensor's scale."""
import pytest
import torch

from oracle import torch_ref
from tests import helpers

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)


# InstanceNorm cancels the bias of the conv in front of it: those gradients are pure round-off in both implementations
NORMED_BIAS = {"model.%d.bias" % k for k in torch_ref.discriminator_conv_keys(4)[1:-1]}


def _make_ctx(precision):
    from impersonator_amd.networks.discriminator import PatchDiscriminator
    sd = helpers.discriminator_state_dict(seed=3)
    D = PatchDiscriminator(6, 64, 4, 'instance', False, image_size=64, max_batch=2, conv_precision=precision)
    D.load_state_dict(sd)
    D = D.cuda()
    gen = torch.Generator().manual_seed(1)
    batches = [(torch.rand(2, 6, 64, 64, generator=gen) * 2 - 1, torch.rand(2, 6, 64, 64, generator=gen) * 2 - 1)
               for _ in range(2)]
    return dict(sd=sd, D=D, batches=batches)


@pytest.fixture(scope="module")
def ctx():
    c = _make_ctx("fp32")
    yield c
    c["D"].release()


@pytest.fixture(scope="module", params=["fp32", "bf16x3"])
def ctx_p(request):
    """Both arithmetic modes of the convolutions (lwg_discriminator_set_precision) against the same fp32 oracle and the same
    bounds: the bf16x3 products carry 16 mantissa bits per operand, sums are fp32 in both."""
    c = _make_ctx(request.param)
    yield c
    c["D"].release()


def test_forward_matches_oracle(ctx_p):
    ctx = ctx_p
    real = ctx["batches"][0][0]
    out = ctx["D"](real.cuda()).cpu()
    ref = torch_ref.discriminator_forward(ctx["sd"], real)
    assert out.shape == ref.shape == (2, 1, 2, 2)
    assert _rel(out, ref) < 1e-4


def test_loss_gradients_and_adam_steps(ctx_p):
    ctx = ctx_p
    D, batches = ctx["D"], ctx["batches"]
    losses, grads, final = torch_ref.discriminator_train_steps(ctx["sd"], batches)
    real, fake = batches[0]
    loss0 = D.optimize_D(real.cuda(), fake.cuda(), all_reduce=False)
    mine = D.gradients()
    assert abs(float(loss0) - losses[0]) < 1e-5 * max(1.0, abs(losses[0]))
    for k, g in grads.items():
        assert mine[k].shape == g.shape
        if k in NORMED_BIAS:
            assert float(g.abs().max()) < 1e-5 and float(mine[k].abs().max()) < 1e-5, k
            continue
        assert _rel(mine[k], g) < 2e-3, (k, _rel(mine[k], g))
    real, fake = batches[1]
    loss1 = D.optimize_D(real.cuda(), fake.cuda(), all_reduce=False)
    assert abs(float(loss1) - losses[1]) < 1e-3 * max(1.0, abs(losses[1]))
    D.pull_parameters()
    if D.conv_precision != "fp32":
        # 16-bit operands move pre-activations by ~1e-5: LeakyReLU masks flip where float32 keeps them, single gradient entries
        # move by 1e-3 of the tensor's scale (test_gradients_against_float64_autograd_128) and Adam's second step divides by |g|
        return
    for k, v in final.items():
        if k in NORMED_BIAS:
            continue   # Adam normalises round-off gradients to +-lr: not comparable
        # Adam turns a gradient into +-lr whatever its size, so entries whose gradient is round-off level may legitimately
        # differ by 2*lr per step; the entries with a significant gradient must move together
        big = grads[k].abs() > 1e-2 * grads[k].abs().max()
        diff = (D.state_dict()[k].cpu() - v).abs()
        assert float(diff[big].max()) < 0.2 * 0.0002, k
        assert float(diff.max()) < 2.1 * 2 * 0.0002, k


def test_matches_reference_golden(ctx):
    """Against numbers produced by the real reference code (tests/golden/make_golden.py::make_discriminator)."""
    from impersonator_amd.networks.discriminator import PatchDiscriminator
    import numpy as np
    g = helpers.golden("discriminator_golden.npz")
    D = PatchDiscriminator(6, 64, 4, 'instance', False, image_size=64, max_batch=2)
    D.load_state_dict(ctx["sd"])
    D = D.cuda()
    real, fake = ctx["batches"][0]
    assert np.allclose(D(real.cuda()).cpu().numpy(), g["d_real"], atol=1e-5, rtol=1e-4)
    loss = D.optimize_D(real.cuda(), fake.cuda(), all_reduce=False)
    assert abs(float(loss) - float(g["loss"][0])) < 1e-5 * float(g["loss"][0])
    grads = D.gradients()
    for k, v in grads.items():
        if k in NORMED_BIAS:
            continue
        ref = g["gnorm/" + k]
        assert abs(v.double().abs().sum().item() - ref[0]) <= 2e-3 * ref[0], k
        s = g["gsample/" + k]
        assert float(np.abs(v.flatten()[::97].numpy() - s).max()) <= 2e-3 * float(np.abs(s).max()), k
    D.release()


def test_deterministic_and_flat_buffers(ctx):
    from impersonator_amd.networks.discriminator import PatchDiscriminator
    outs = []
    for _ in range(2):
        D = PatchDiscriminator(6, 64, 4, 'instance', False, image_size=64, max_batch=2)
        D.load_state_dict(ctx["sd"])
        D = D.cuda()
        real, fake = ctx["batches"][0]
        D.optimize_D(real.cuda(), fake.cuda(), all_reduce=False)
        p, g = D.flat_buffers()
        outs.append((p.clone(), g.clone()))
        assert p.numel() == g.numel() and p.is_cuda
        D.release()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_reference_size_forward(ctx):
    """The trainer's configuration (impersonator_trainer.py:219-222) at 256x256: 14x14 patch map."""
    from impersonator_amd.networks.discriminator import PatchDiscriminator
    D = PatchDiscriminator(6, 64, 4, 'instance', False, image_size=256, max_batch=1)
    D.load_state_dict(ctx["sd"])
    D = D.cuda()
    x = torch.rand(1, 6, 256, 256, generator=torch.Generator().manual_seed(5)) * 2 - 1
    out = D(x.cuda()).cpu()
    ref = torch_ref.discriminator_forward(ctx["sd"], x)
    assert out.shape == ref.shape == (1, 1, 14, 14) and _rel(out, ref) < 1e-4
    D.release()


@pytest.mark.parametrize("precision,bound", [("fp32", 2e-5), ("bf16x3", 1e-2)])
def test_gradients_against_float64_autograd_128(precision, bound):
    """At 128x128 every gradient is within a few 1e-6 (relative to the tensor's max) of float64 autograd -- the same
    distance torch's own float32 autograd keeps.  (At larger sizes float32 round-off flips LeakyReLU masks and torch-f32
    itself drifts 1e-3..1e-2 from float64; there the HIP path tracks torch-f32.)  The bf16x3 mode of the convolutions
    (16 mantissa bits per operand, forward error ~1e-5) flips LeakyReLU masks the way float32 does at larger sizes: 3.6e-3
    measured on the first layer's weights, bound 1e-2."""
    from impersonator_amd.networks.discriminator import PatchDiscriminator
    sd = helpers.discriminator_state_dict(seed=3)
    D = PatchDiscriminator(6, 64, 4, 'instance', False, image_size=128, max_batch=1, conv_precision=precision)
    D.load_state_dict(sd)
    D = D.cuda()
    gen = torch.Generator().manual_seed(2)
    real = torch.rand(1, 6, 128, 128, generator=gen) * 2 - 1
    fake = torch.rand(1, 6, 128, 128, generator=gen) * 2 - 1
    _, gd, _ = torch_ref.discriminator_train_steps({k: v.double() for k, v in sd.items()}, [(real.double(), fake.double())])
    D.optimize_D(real.cuda(), fake.cuda(), all_reduce=False)
    mine = D.gradients()
    for k, g in gd.items():
        if k in NORMED_BIAS:
            continue
        err = float((mine[k].double() - g).abs().max()) / float(g.abs().max())
        print("%s %s: %.2e" % (precision, k, err))
        assert err <= bound, k
    D.release()


def test_trainer_mirror_forward_and_d_phase():
    """models/impersonator_trainer.py: generator pass (impersonator_trainer.py:329-348) + discriminator update (:396-411)
    chained as optimize_parameters does for D; checked against the oracle composed the same way."""
    import types
    from impersonator_amd.models.impersonator_trainer import Impersonator
    opt = types.SimpleNamespace(image_size=128, batch_size=2, map_name='uv_seg', norm_type='instance', repeat_num=6,
                                lr_D=0.0002, D_adam_b1=0.5, D_adam_b2=0.999, lambda_D_prob=1, is_train=True)
    model = Impersonator(opt)
    gsd = torch_ref.state_dict_from_numpy(helpers.generator_state_dict(seed=2, affine="random"))
    dsd = helpers.discriminator_state_dict(seed=3)
    model._G.load_state_dict(gsd)
    model._D.load_state_dict(dsd)
    gen = torch.Generator().manual_seed(8)
    bg = torch.rand(2, 4, 128, 128, generator=gen) * 2 - 1
    src = torch.rand(2, 6, 128, 128, generator=gen) * 2 - 1
    tsf = torch.rand(2, 6, 128, 128, generator=gen) * 2 - 1
    T = torch.rand(2, 128, 128, 2, generator=gen) * 2.4 - 1.2
    real = torch.rand(2, 3, 128, 128, generator=gen) * 2 - 1
    model.set_input(tsf.cuda(), real.cuda(), input_G_bg=bg.cuda(), input_G_src=src.cuda(), T=T.cuda())
    fake_bg, fake_src, fake_tsf, masks = model.forward()
    with torch.no_grad():
        o_bg, o_sc, o_sm, o_tc, o_tm = torch_ref.generator_forward(gsd, bg, src, tsf, T)
        o_tsf = o_tm * o_bg + (1 - o_tm) * o_tc
    assert masks.shape == (4, 1, 128, 128) and float((fake_tsf.cpu() - o_tsf).abs().max()) < 1e-3
    assert float((fake_src.cpu() - (o_sm * o_bg + (1 - o_sm) * o_sc)).abs().max()) < 1e-3
    loss = model.optimize_D_phase()
    with torch.no_grad():
        ref_loss = torch_ref.discriminator_loss(dsd, torch.cat([real, tsf[:, 3:]], 1), torch.cat([o_tsf, tsf[:, 3:]], 1))
    assert abs(float(loss) - float(ref_loss)) < 2e-3 * max(1.0, float(ref_loss))
    model._D.pull_parameters()
    assert float((model._D.state_dict()["model.0.weight"].cpu() - dsd["model.0.weight"]).abs().max()) > 1e-5


def test_input_gradient_for_the_generator_adversarial_term(ctx_p):
    """loss_g_adv = mean(D(fake)^2) (impersonator_trainer.py:369-371) and its gradient wrt the fake image."""
    ctx = ctx_p
    from impersonator_amd.networks.discriminator import PatchDiscriminator
    D = PatchDiscriminator(6, 64, 4, 'instance', False, image_size=128, max_batch=2, conv_precision=ctx["D"].conv_precision)
    D.load_state_dict(ctx["sd"])
    D = D.cuda()
    x = torch.rand(2, 6, 128, 128, generator=torch.Generator().manual_seed(9)) * 2 - 1
    loss, dx = D.input_grad(x.cuda(), 0.0)
    xr = x.clone().requires_grad_(True)
    ref = torch.mean(torch_ref.discriminator_forward(ctx["sd"], xr) ** 2)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * max(1.0, float(ref))
    assert _rel(dx.cpu(), xr.grad) < 1e-3
    D.release()
