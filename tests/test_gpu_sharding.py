"""The real product path under two ranks: two processes, one MI355X, gloo rendezvous.  Each rank builds
the synthetic Imitator, personalises, and runs run_imitator.py's sharded loop (sharding.imitate_sharded ->
Imitator.predict_batches on its round-robin blocks -> gather in frame order); the gathered sequence must equal a
single-process run bit for bit.  The same processes average a CUDA gradient tensor the way the training step does."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

FRAMES, BATCH = 40, 8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK="0",
                      WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from impersonator_amd import demo, sharding
    r, _, w = sharding.init_process_group(backend="gloo")   # gloo: both ranks share cuda:0 (RCCL wants one GPU per rank)
    assert (r, w) == (rank, world)
    torch.cuda.set_device(0)
    imitator, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=BATCH, seed=0, affine="random")
    imitator.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
    smpls = demo.synthetic_smpls(FRAMES, seed=7)
    outs = sharding.imitate_sharded(imitator, smpls, BATCH, "smooth", rank, world)
    # the training path's one collective on a CUDA tensor (sharding.average_gradients; RCCL on a real multi-GPU node)
    g = torch.full((1 << 16,), float(rank + 1), device="cuda")
    sharding.average_gradients(g)
    assert bool((g == (1 + world) / 2.0).all())
    if rank == 0:
        assert len(outs) == FRAMES
        # single-process run of the whole sequence in the same process (world 1 semantics: every block is mine)
        single = sharding.imitate_sharded(imitator, smpls, BATCH, "smooth", 0, 1)
        same = all(np.array_equal(a, b) for a, b in zip(outs, single))
        np.save(out_path, np.array([int(same), len(outs)]))
    else:
        assert outs is None
    sharding.barrier()
    dist.destroy_process_group()


def test_two_ranks_product_path_equals_single_process(tmp_path):
    out = str(tmp_path / "result.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    same, n = np.load(out)
    assert n == FRAMES and same == 1, "sharded sequence differs from the single-process run"
