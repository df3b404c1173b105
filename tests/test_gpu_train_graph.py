"""GPU: a training iteration captured in a HIP graph (Impersonator.optimize_parameters_graphed) IS the eager iteration: identical
loss terms, gradients, Adam moments and parameters iteration by iteration (Adam's step count lives on the device:
lwg_adam_update_device_step, lwg_discriminator_use_device_step), a new batch reaches the replay through set_input without the
caller's tensors being written, a changed learning rate or batch shape re-captures after a fresh warm-up."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _inputs(model):
    return dict(input_G_tsf=model._input_G_tsf, real_tsf=model._real_tsf, input_G_bg=model._input_G_bg, input_G_src=model._input_G_src,
                T=model._T, real_src=model._real_src, bg_mask=model._bg_mask)


def test_adam_with_the_step_count_on_the_device():
    """lwg_adam_update_device_step == lwg_adam_update step for step (the count incremented on the stream, the bias corrections
    evaluated from it in the kernel)."""
    from impersonator_amd import ops
    g = torch.Generator().manual_seed(5)
    n = 10007
    p0 = torch.randn(n, generator=g).cuda()
    pa, pb = p0.clone(), p0.clone()
    ma, va, mb, vb = (torch.zeros(n, device="cuda") for _ in range(4))
    t_dev = ops.adam_step_state(0, (0.5, 0.999))
    for t in range(1, 7):
        grad = (torch.randn(n, generator=g) * 10.0 ** float(torch.randint(-4, 2, (1,), generator=g))).cuda()
        ops.adam_update(pa, grad, ma, va, t, 2e-4, (0.5, 0.999), 1e-8)
        ops.adam_update_device_step(pb, grad, mb, vb, t_dev, 2e-4, (0.5, 0.999), 1e-8)
        assert int(t_dev[0]) == t
        assert torch.equal(ma, mb) and torch.equal(va, vb)
        # beta^t as a running product in double on the device against powf on the host: a few float32 ulps of 1 - 0.999^t
        assert float((pa - pb).abs().max()) <= 1e-4 * 2e-4, t
    assert float((pa - p0).abs().max()) > 5e-4


def _state(model):
    tr = model._generator_trainer()
    d_par, d_grad = model._D.flat_buffers()
    return dict(g_par=tr.flat_p, g_grad=tr.flat_g, g_m=tr.flat_m, g_v=tr.flat_v, d_par=d_par, d_grad=d_grad)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_graph_replay_is_the_eager_iteration_bit_for_bit(precision):
    """Every kernel of a training iteration is deterministic (the grid_sample gradient is a gather in a fixed order,
    csrc/scatter.hip), so a replayed graph and the eager calls are THE SAME computation: two models built from one seed, one
    stepped eagerly, one through optimize_parameters_graphed, must hold identical loss terms, gradients, Adam moments and
    parameters of G and D after every iteration -- a stale gradient slice, a skipped zero-fill, a frozen bias correction or a
    batch that did not reach the replay all break equality.  (The eager model switches to the device-resident Adam step count
    when the other one captures: host powf and the device's running product differ in the last bits of 1 - beta^t.)"""
    import bench_train
    eager, graphed = bench_train.build(2, 64, precision, seed=3), bench_train.build(2, 64, precision, seed=3)
    other = _inputs(bench_train.build(2, 64, precision, seed=11))          # a second batch, same shapes
    mine = {k: v.clone() for k, v in other.items()}                          # what the caller keeps of it
    for it in range(7):
        if it == 2:
            eager._device_steps(True)                                        # the graphed model captures in this call
        if it == 5:                                                          # after the capture (iteration 2) and two replays
            eager.set_input(**{k: v.clone() for k, v in other.items()})
            graphed.set_input(**other)
        ref, got = eager.optimize_parameters(), graphed.optimize_parameters_graphed()
        assert (graphed._graph is not None) == (it >= 2)
        assert ref.keys() == got.keys()
        for k in ref:
            assert ref[k] == got[k], (it, k, ref[k], got[k])
        a, b = _state(eager), _state(graphed)
        for k in a:
            assert torch.equal(a[k], b[k]), (it, k, float((a[k] - b[k]).abs().max()))
    # the step counts advanced on the device, once per iteration
    assert int(graphed._generator_trainer().t_dev[0]) == 7 == int(eager._generator_trainer().t_dev[0])
    # the second batch reached the replay (and is a different one) ...
    assert abs(got["g_rec"] - ref["g_rec"]) == 0.0
    # ... through the graph's PRIVATE tensors: what the caller passed to set_input is untouched, and not what the graph reads
    for k, v in other.items():
        assert torch.equal(v, mine[k]) and getattr(graphed, "_" + k).data_ptr() != v.data_ptr()
    # a learning-rate change is noticed: the graph is dropped, two eager iterations warm up, the next call captures again
    graphed._current_lr_D = eager._current_lr_D = 1e-4
    for it in range(4):
        if it == 2:
            eager._device_steps(True)
        if it == 0:
            eager._device_steps(False)                                       # the graphed model goes back to host counts, too
        ref, got = eager.optimize_parameters(), graphed.optimize_parameters_graphed()
        assert (graphed._graph is not None) == (it >= 2), it
        for k in ref:
            assert ref[k] == got[k], ("after the lr change", it, k)
    assert graphed._graph_lrs[0] == 1e-4
    a, b = _state(eager), _state(graphed)
    for k in a:
        assert torch.equal(a[k], b[k]), ("after the lr change", k)
    graphed.drop_graph()
    assert graphed._generator_trainer().t == 11 and graphed._graph_warm == 0


def test_new_batch_shape_warms_up_again_before_it_captures():
    """set_input with another batch size drops the graph AND the warm-up count: the next calls run eagerly (new kernel variants,
    scratch sizes) before anything is captured at the new shape."""
    import bench_train
    m = bench_train.build(2, 64, "fp32", seed=3)
    for _ in range(4):
        m.optimize_parameters_graphed()
    assert m._graph is not None
    m.set_input(**_inputs(bench_train.build(1, 64, "fp32", seed=5)))
    assert m._graph is None and m._graph_warm == 0
    m.optimize_parameters_graphed()
    assert m._graph is None and m._graph_warm == 1
    m.optimize_parameters_graphed()
    losses = m.optimize_parameters_graphed()
    assert m._graph is not None and all(v == v for v in losses.values())
