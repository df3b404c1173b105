"""The multi-rank code paths BASELINE.json's configs 3 and 5 depend on, executed on the ONE visible GPU:
two processes with a gloo rendezvous (RCCL wants a GPU per rank; on a multi-GPU node the same code takes "nccl").

  * `bench.py --gpus 2` -- BARE (it starts its own ranks) and under `python -m torch.distributed.run`, exactly the driver's
    launch line -- prints ONE JSON line with n_gpus 2, weak scaling, a finite value, the roofline block and the per-rank block
    that proves the two ranks; with fewer GPUs than ranks and no test hook it exits 1;
  * a data-parallel training iteration: two ranks run `Impersonator.optimize_parameters` on half a batch each with the
    gradient all-reduce on; the averaged gradients and the updated parameters of G and D equal the single-process step
    on the whole batch (the reference gets the same thing from nn.DataParallel, models/impersonator_trainer.py:196-214,
    350-366: every loss term is a batch mean and InstanceNorm has no cross-sample statistics, so the mean of the two
    half-batch gradients IS the full-batch gradient);
  * tools/bench_train.py under torchrun (per-iteration time + the all-reduce of the G and D gradient buffers)."""
import json
import os
import socket
import subprocess
import sys
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests import helpers

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZE, BATCH = 64, 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(script_args, nproc=2, timeout=900):
    env = dict(os.environ, LWG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    assert p.returncode == 0, "torchrun failed (%d)\n%s\n%s" % (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return lines


def _bare(script_args, env_extra, timeout=900):
    """`python <script> ...` with NO torch.distributed environment (how the driver starts `bench.py --gpus 1`)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT, **env_extra)
    return subprocess.run([sys.executable] + script_args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=timeout, text=True)


def _check_two_rank_line(line):
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 6 and line["warmup"] == 2
    assert np.isfinite(line["value"]) and line["value"] > 0
    assert abs(line["value"] - 2 * 8 * 6 / (line["ms_per_step"] * 6e-3)) <= 1e-2 * line["value"]   # whole-job frames / max-over-ranks time
    assert line["roofline"]["kernel"].startswith("conv") and 0 < line["roofline"]["frac"] < 1
    assert line["config"]["parallelism"] == "frame-sharded replicas x2"
    assert "cpu_baseline" not in line      # rank 0 at N = 1 only
    # the N-rank line proves itself: communicator size, per-rank rates and devices, the single-rank rate of the same run
    assert line["rccl"]["ranks"] == 2 and line["rccl"]["allreduce_of_ones"] == 2.0
    rk = line["ranks"]
    assert len(rk["per_rank"]) == 2 and sorted(r["rank"] for r in rk["per_rank"]) == [0, 1]
    assert all(r["fps"] > 0 and r["device"]["name"] for r in rk["per_rank"])
    assert rk["fps_min"] <= rk["fps_median"] <= rk["fps_max"] and rk["single_rank_fps"] > 0
    assert abs(rk["linear_frac"] - line["value"] / (2 * rk["single_rank_fps"])) <= 1e-3
    assert line["repeats"] == 3 and line["ms_per_step_min"] <= line["ms_per_step"] <= line["ms_per_step_max"]
    assert len(line["ms_per_step_windows"]) == 3


def test_bench_two_ranks_bare_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torchrun (the way the driver starts --gpus 1): bench.py re-executes itself under
    torch.distributed.run with two ranks (gloo hook: they share the one GPU) and the line says n_gpus 2 -- never a silent n_gpus 1."""
    p = _bare(["bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2", "--repeats", "3", "--no-cpu-baseline", "--no-fp32-mode",
               "--no-secondary", "--no-strict"], {"LWG_DIST_BACKEND": "gloo"})
    assert p.returncode == 0, "bare bench.py --gpus 2 failed (%d)\n%s\n%s" % (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines          # rank 0 only
    _check_two_rank_line(json.loads(lines[0]))


def test_bench_refuses_more_ranks_than_gpus():
    """On a box with ONE GPU `python bench.py --gpus 2` (no gloo hook) must exit 1 with a message, not print n_gpus 1."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    p = _bare(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"], {}, timeout=300)
    assert p.returncode == 1, (p.returncode, p.stdout[-500:], p.stderr[-500:])
    assert "refusing" in p.stderr and "--gpus 2" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    # and a WORLD_SIZE that disagrees with --gpus is refused as well
    q = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=ROOT, text=True,
                       env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", PYTHONPATH=ROOT), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert q.returncode != 0 and "WORLD_SIZE" in q.stderr and not [l for l in q.stdout.splitlines() if l.startswith("{")]


def test_bench_two_ranks_under_torchrun():
    lines = _torchrun(["bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2", "--repeats", "3", "--no-cpu-baseline",
                       "--no-fp32-mode", "--no-secondary", "--no-strict"])
    assert len(lines) == 1, lines          # rank 0 only
    _check_two_rank_line(json.loads(lines[0]))


def _opt(batch):
    return types.SimpleNamespace(image_size=SIZE, batch_size=batch, map_name='uv_seg', norm_type='instance', repeat_num=6,
                                 is_train=True, conv_precision="fp32")


def _model(batch):
    from impersonator_amd.models.impersonator_trainer import Impersonator
    from oracle import torch_ref
    m = Impersonator(_opt(batch))
    m._G.load_state_dict(torch_ref.state_dict_from_numpy(helpers.generator_state_dict(seed=2, affine="random")))
    m._D.load_state_dict(helpers.discriminator_state_dict(seed=3))
    return m


def _set(m, b, lo, hi):
    n = b["input_G_src"].shape[0]
    cut = lambda t: t[lo:hi].cuda()
    # bg_mask is cat([source masks, target masks]) (impersonator_trainer.py:304): a rank takes its rows of both halves
    m.set_input(cut(b["input_G_tsf"]), cut(b["real_tsf"]), input_G_bg=cut(b["input_G_bg"]), input_G_src=cut(b["input_G_src"]),
                T=cut(b["T"]), real_src=cut(b["real_src"]),
                bg_mask=torch.cat([b["bg_mask"][lo:hi], b["bg_mask"][n + lo:n + hi]]).cuda())


def _snapshot(m):
    tr = m._generator_trainer()
    d_par, d_grad = m._D.flat_buffers()
    return dict(g_grad=tr.flat_g.cpu().numpy().copy(), g_par=tr.flat_p.cpu().numpy().copy(),
                d_grad=d_grad.cpu().numpy().copy(), d_par=d_par.cpu().numpy().copy())


def _dp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from impersonator_amd import sharding
    sharding.init_process_group(backend="gloo")
    torch.cuda.set_device(0)
    per = BATCH // world
    m = _model(per)
    b = helpers.train_batch(seed=9, n=BATCH, size=SIZE)
    _set(m, b, rank * per, (rank + 1) * per)
    losses = m.optimize_parameters()
    snap = _snapshot(m)
    # every rank must hold the same averaged gradients and the same parameters after the step
    for k, v in snap.items():
        t = torch.from_numpy(v)
        other = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(other, t)
        assert all(torch.equal(o, other[0]) for o in other), "ranks disagree on %s after the step" % k
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp.npz"), **snap, **{"loss_" + k: np.float64(v) for k, v in losses.items()})
    # the loss terms a rank reports are those of its half batch: their mean over the ranks is the full-batch value
    keys = sorted(losses)
    t = torch.tensor([losses[k] for k in keys], dtype=torch.float64)
    dist.all_reduce(t)
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp_losses.npz"), **{k: float(v) / world for k, v in zip(keys, t)})
    sharding.barrier()
    dist.destroy_process_group()


def test_data_parallel_training_step_equals_the_full_batch_step(tmp_path):
    m = _model(BATCH)
    b = helpers.train_batch(seed=9, n=BATCH, size=SIZE)
    _set(m, b, 0, BATCH)
    losses = m.optimize_parameters()
    single = _snapshot(m)
    del m
    torch.cuda.empty_cache()
    mp.spawn(_dp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    dp = np.load(str(tmp_path / "dp.npz"))
    dpl = np.load(str(tmp_path / "dp_losses.npz"))
    for k, v in losses.items():
        assert abs(float(dpl[k]) - v) <= 2e-5 * max(1.0, abs(v)), (k, float(dpl[k]), v)
    lr = 2e-4
    for net in ("g", "d"):
        a, c = dp[net + "_grad"].astype(np.float64), single[net + "_grad"].astype(np.float64)
        rel = np.linalg.norm(a - c) / np.linalg.norm(c)
        # fp32 sums in another order (a weight gradient's pixel slices depend on the batch, the all-reduce adds two
        # partial sums) and the atomic adds of the grid_sample gradient: 1e-6-level relative noise, nothing structural
        assert rel <= 2e-5, (net, rel)
        # Adam's first step moves every entry by lr * g / (|g| + eps): +-lr whatever |g| is, so the parameters only see
        # the gradient's SIGN -- an entry whose gradient is at the noise level may land 2 lr apart.  Bound both.
        pa, pc = dp[net + "_par"].astype(np.float64), single[net + "_par"].astype(np.float64)
        d = np.abs(pa - pc)
        assert d.max() <= 2.0 * lr * 1.001, (net, d.max())
        assert (d > 0.01 * lr).mean() <= 2e-3, (net, float((d > 0.01 * lr).mean()))


def test_bench_train_under_torchrun():
    lines = _torchrun(["tools/bench_train.py", "--batch", "2", "--image-size", "128", "--steps", "2", "--precision", "bf16x3"])
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["world"] == 2 and line["batch_per_rank"] == 2 and line["images_per_s"] > 0
    ar = line["all_reduce"]
    assert ar["G_bytes"] > 3.8e8 and 2.7e7 < ar["D_bytes"] < 3.1e7   # 27.8 MB of parameters + the layout's padding rows and ar["G_ms"] > 0 and ar["D_ms"] > 0
