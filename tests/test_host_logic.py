"""CPU checks of the Python host code that wraps the C ABI (no device work)."""
import types

import numpy as np
import pytest
import torch

from impersonator_amd import demo
from impersonator_amd.utils import synthetic, util
from oracle import torch_ref


def test_synthetic_mesh_has_smpl_counts_and_is_deterministic():
    v, f = synthetic.body_mesh()
    assert v.shape == (6890, 3) and f.shape == (13776, 3) and f.min() == 0 and f.max() == 6889
    v2, f2 = synthetic.body_mesh()
    assert np.array_equal(v, v2) and np.array_equal(f, f2)
    # closed genus-0 surface: every edge shared by exactly two faces
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), 1)
    _, counts = np.unique(e, axis=0, return_counts=True)
    assert (counts == 2).all()
    m = synthetic.uv_seg_map_fn(v, f)
    assert m.shape == (13777, 3) and tuple(m[-1]) == (0.0, 0.0, 1.0) and (m[:-1, 2] == 0).all()
    a = synthetic.random_state_dict([("x.0.weight", (4, 3, 3, 3)), ("x.1.weight", (4,)), ("x.1.bias", (4,))], 0)
    b = synthetic.random_state_dict([("x.0.weight", (4, 3, 3, 3)), ("x.1.weight", (4,)), ("x.1.bias", (4,))], 0)
    assert all(np.array_equal(a[k], b[k]) for k in a)


def test_morph_matches_reference_restatement():
    g = torch.Generator().manual_seed(0)
    m = (torch.rand(2, 1, 48, 48, generator=g) > 0.3).float()
    m[:, :, 10:40, 10:40] = 1
    for ks in (3, 13):
        for mode in ("erode", "dilate"):
            assert torch.equal(util.morph(m, ks, mode), torch_ref.morph(m, ks, mode))


def test_swap_smpl_camera_policies():
    """Imitator.swap_smpl against the reference's own method (models/imitator.py:216-234): run unbound where /root/reference
    exists, and against the SMPL vectors that method produced inside the reference's inference_by_smpls
    (tests/golden/imitator_golden.npz, make_golden.py::make_imitator) everywhere."""
    from impersonator_amd.models.imitator import Imitator
    from oracle import reference_loader
    from tests import helpers
    tgt = torch.from_numpy(demo.synthetic_smpls(4, 0))
    src_cam = torch.tensor([[0.9, 0.0, 0.05]])
    shape = torch.arange(10).float()[None]
    if reference_loader.available():
        ref = reference_loader.load().imitator.Imitator
        for strategy in ("smooth", "source", "copy"):
            me = types.SimpleNamespace(first_cam=torch.tensor([[1.0, 0.1, 0.2]]))
            # the reference takes one frame per call (its loop, imitator.py:166,196); this build's method takes a batch
            want = torch.cat([ref.swap_smpl(me, src_cam, shape, tgt[i:i + 1], strategy) for i in range(4)])
            assert torch.equal(Imitator.swap_smpl(me, src_cam, shape, tgt, strategy), want), strategy
    g = helpers.golden("imitator_golden.npz")
    for name, v in helpers.IMITATOR_VARIANTS.items():
        k = name + "/"
        sc = helpers.imitator_scene(64)        # the SMPL vectors do not depend on the image size
        src = torch.from_numpy(g[k + "src_theta"])
        fc = g[k + "first_cam"][-1]
        me = types.SimpleNamespace(first_cam=None if np.isnan(fc).all() else torch.from_numpy(fc)[None])
        out = Imitator.swap_smpl(me, src[:, :3], src[:, 75:], torch.from_numpy(sc["tgt_smpls"]), v["cam_strategy"])
        assert np.array_equal(out.numpy(), g[k + "theta"]), name
        if v["cam_strategy"] == "smooth":
            assert np.array_equal(fc, sc["tgt_smpls"][0, :3])      # first_cam = the frame at t == 0 (imitator.py:243-244)


def test_options_keep_reference_flag_names():
    from impersonator_amd.options.test_options import TestOptions
    opt = TestOptions().parse(["--src_path", "a.jpg", "--tgt_path", "b", "--cam_strategy", "copy", "--front_warp"])
    assert opt.image_size == 256 and opt.bg_ks == 13 and opt.ft_ks == 3 and opt.map_name == "uv_seg"
    assert opt.repeat_num == 6 and opt.cam_strategy == "copy" and opt.front_warp and not opt.only_vis


def test_smpl_stage_matches_reference():
    from oracle import reference_loader
    if not reference_loader.available():
        pytest.skip("/root/reference not present")
    ref = reference_loader.load()
    from impersonator_amd.networks.batch_smpl import SMPL, synthetic_smpl_params
    m = SMPL(params=synthetic_smpl_params(0))
    g = torch.Generator().manual_seed(0)
    beta, theta = torch.randn(3, 10, generator=g), torch.randn(3, 72, generator=g) * 0.3
    v, j, _ = m(beta, theta, get_skin=True)
    stub = types.SimpleNamespace(shapedirs=m.shapedirs, v_template=m.v_template, size=m.size,
                                 J_regressor=m.J_regressor, posedirs=m.posedirs, parents=m.parents,
                                 weights=m.weights, joint_regressor=m.joint_regressor, rotate=False)
    rv, rj, _ = ref.batch_smpl.SMPL.forward(stub, beta, theta, get_skin=True)
    assert torch.allclose(v, rv, atol=1e-6) and torch.allclose(j, rj, atol=1e-6)


def test_adam_step_state_words():
    """ops.adam_step_state: the three 8-byte words lwg_adam_update_device_step keeps on the device (include/lwg.h) -- the step count
    and beta1^t, beta2^t as float64 bit patterns; {0, 1.0, 1.0} before the first step."""
    import numpy as np
    from impersonator_amd import ops
    s0 = ops.adam_step_state(0, (0.5, 0.999), device="cpu")
    assert s0.dtype == torch.int64 and s0.shape == (3,) and int(s0[0]) == 0
    assert np.array_equal(s0.numpy()[1:].view(np.float64), np.array([1.0, 1.0]))
    s7 = ops.adam_step_state(7, (0.5, 0.999), device="cpu")
    assert int(s7[0]) == 7
    assert np.array_equal(s7.numpy()[1:].view(np.float64), np.array([0.5 ** 7, 0.999 ** 7]))


def test_adjacent_row_blocks_are_taken_as_one_view():
    """Imitator._adjacent_rows: consecutive row blocks of ONE contiguous tensor become a view of it (no copy kernel in a round's launch
    sequence); anything else -- a gap, another tensor, another width, a strided block -- is refused (the caller concatenates)."""
    from impersonator_amd.models.imitator import Imitator
    base = torch.arange(40 * 85, dtype=torch.float32).reshape(40, 85)
    blocks = [base[8:16], base[16:24], base[24:29]]
    whole = Imitator._adjacent_rows(blocks)
    assert whole is not None and whole.shape == (21, 85) and whole.data_ptr() == base[8:].data_ptr()
    assert torch.equal(whole, base[8:29])
    assert Imitator._adjacent_rows([base[0:8]]).data_ptr() == base.data_ptr()
    assert Imitator._adjacent_rows([base[0:8], base[16:24]]) is None                 # a gap
    assert Imitator._adjacent_rows([base[8:16], base[0:8]]) is None                  # out of order
    assert Imitator._adjacent_rows([base[0:8], base.clone()[8:16]]) is None          # another tensor
    assert Imitator._adjacent_rows([base[0:8], base[8:16, :80]]) is None             # another width / not contiguous
    assert Imitator._adjacent_rows([base[0:8].double()]) is None                     # not fp32


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    """bench.py's launch guard is pure host logic up to its first device call; on a box without a GPU the refusal it prints is the
    'needs an MI355X' one -- the N-rank guards themselves are exercised on the GPU box (tests/test_gpu_multirank.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by tests/test_gpu_multirank.py")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], cwd=root, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300, env=dict(os.environ, PYTHONPATH=root))
    assert p.returncode != 0 and "MI355X" in p.stderr and not p.stdout.strip()


def test_bench_self_launch_builds_the_torchrun_command_or_refuses(monkeypatch, capsys):
    """bench.self_launch (what `python bench.py --gpus N` does without a torch.distributed environment), with the device count and
    the exec mocked: enough devices -> re-execution under `python -m torch.distributed.run --nproc-per-node N` with the original
    arguments; too few -> exit 1 and a message, unless the gloo test hook lets the ranks share a device."""
    import os
    import sys
    import types as _types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    calls = []
    monkeypatch.setattr(bench.os, "execve", lambda exe, cmd, env: calls.append((exe, cmd, env)))
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    args = _types.SimpleNamespace(gpus=8)
    monkeypatch.delenv("LWG_DIST_BACKEND", raising=False)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    bench.self_launch(args)
    exe, cmd, env = calls.pop()
    assert exe == sys.executable and cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-7:] == [os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # one GPU, eight ranks: refused
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.self_launch(args)
    assert e.value.code == 1 and not calls
    err = capsys.readouterr().err
    assert "refusing" in err and "--gpus 8" in err and "1 GPU(s)" in err
    # ... unless the test hook lets the ranks share the visible device
    monkeypatch.setenv("LWG_DIST_BACKEND", "gloo")
    bench.self_launch(args)
    assert calls and calls[0][1][calls[0][1].index("--nproc-per-node") + 1] == "8"
