#!/usr/bin/env python
"""RCCL itself, executed from this repository's own multi-rank code, on however many GPUs are visible.

    LWG_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=P python tools/rccl_smoke.py
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/rccl_smoke.py

gpurun offers one GPU and RCCL wants a GPU per rank, so the tests run this at world_size 1 with LWG_FORCE_DIST=1
(`sharding.forced()`): backend "nccl" (= RCCL on ROCm), communicator bound to the rank's device at init
(`init_process_group(device_id=...)`), and then exactly the calls `bench.py`, `run_imitator.py` and the trainer make --
`sharding.barrier`, `max_over_ranks` / `sum_over_ranks` on a device tensor, `gather_in_frame_order` (object gather) and
`average_gradients` on the trainer's real flat gradient buffers (generator 390 MB, discriminator 29.9 MB).  A
single-rank all-reduce is a copy-through inside RCCL, but everything around it -- bootstrap over the rendezvous, the
HIP streams/events ProcessGroupNCCL puts around a collective, dmabuf IPC initialisation (HSA_ENABLE_IPC_MODE_LEGACY=0),
the kernels RCCL launches -- is what a first N-GPU run would otherwise meet for the first time.  Rank 0 prints one
JSON line."""
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from impersonator_amd import sharding  # noqa: E402


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, local_rank, world = sharding.init_process_group(backend=os.environ.get("LWG_DIST_BACKEND", "nccl"))
    import torch.distributed as dist
    assert dist.is_initialized(), "no process group: launch under torchrun or with LWG_FORCE_DIST=1"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    out = {"world": world, "device": torch.cuda.get_device_name(dev)}
    out["rccl"] = sharding.collective_info(dev)
    assert out["rccl"]["ranks"] == world and out["rccl"]["allreduce_of_ones"] == world

    sharding.barrier(dev)
    assert sharding.max_over_ranks(1.5 + rank, dev) == 1.5 + (world - 1)
    assert sharding.sum_over_ranks(2.0, dev) == 2.0 * world

    # frame-ordered gather of per-frame outputs (what run_imitator.py collects on rank 0)
    frames, batch = 8 * world + 3, 4
    mine = [("f%d" % t, rank) for s, e in sharding.shard_blocks(frames, batch, rank, world) for t in range(s, e)]
    got = sharding.gather_in_frame_order(mine, frames, batch, rank, world)
    if rank == 0:
        assert [g[0] for g in got] == ["f%d" % t for t in range(frames)]

    # the training path's one collective, on the trainer's real flat gradient buffers
    from impersonator_amd.models.impersonator_trainer import Impersonator
    size = int(os.environ.get("LWG_RCCL_SMOKE_SIZE", 64))
    m = Impersonator(types.SimpleNamespace(image_size=size, batch_size=1, map_name='uv_seg', norm_type='instance', repeat_num=6,
                                           is_train=True, conv_precision="bf16x3"))
    tr = m._generator_trainer()
    g_grad, d_grad = tr.flat_g, m._D.flat_buffers()[1]
    times = {}
    for name, buf in (("G", g_grad), ("D", d_grad)):
        n = buf.numel()
        pattern = torch.arange(n, device=dev, dtype=torch.float32).remainder_(1021.0).mul_(1.0 / 1021.0)
        buf.copy_(pattern * (rank + 1))
        sharding.average_gradients(buf)      # warm-up call: communicator's first large collective
        expect = pattern * ((world + 1) / 2.0)
        err = float((buf - expect).abs().max())
        assert err <= 1e-6, (name, err)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            sharding.average_gradients(buf)
        torch.cuda.synchronize(dev)
        times[name] = {"bytes": n * 4, "ms": round((time.perf_counter() - t0) / reps * 1e3, 4), "max_abs_err": err}
    out["average_gradients"] = times
    m._D.release()
    m._G.release()
    del m, tr, g_grad, d_grad
    torch.cuda.empty_cache()

    if os.environ.get("LWG_RCCL_SMOKE_TRAIN", "1") != "0":
        # the data-parallel training iteration on this process group: gradient buckets all-reduced on a side stream underneath
        # the backward pass (sharding.GradientBuckets), and the whole iteration -- collectives included -- captured in a HIP graph
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_train
        eager, graphed = bench_train.build(2, 64, "bf16x3", seed=3), bench_train.build(2, 64, "bf16x3", seed=3)
        same_terms = True
        for it in range(5):
            if it == 2:
                eager._device_steps(True)
            a, b = eager.optimize_parameters(), graphed.optimize_parameters_graphed()
            same_terms = same_terms and a == b
        te, tg = eager._generator_trainer(), graphed._generator_trainer()
        log = te._buckets.launched_log if te._buckets is not None else []
        out["train"] = {"bucket_count": len(te._buckets.buckets) if te._buckets is not None else 0,
                        "bucket_mbytes": [round((hi - lo) * 4 / 2 ** 20, 1) for lo, hi, _ in te._buckets.buckets] if te._buckets else [],
                        "launched_while_gradients_were_open": sum(1 for _, left in log if left > 0), "launches": len(log),
                        "graph_captured_with_collectives": graphed._graph is not None, "graph_failed": graphed._graph_failed,
                        "replay_equals_eager_terms": bool(same_terms),
                        "replay_equals_eager_parameters": bool(torch.equal(te.flat_p, tg.flat_p)) and
                        bool(torch.equal(eager._D.flat_buffers()[0], graphed._D.flat_buffers()[0]))}
        for mdl in (eager, graphed):
            mdl._D.release()
            mdl._G.release()
        del eager, graphed, te, tg
        torch.cuda.empty_cache()
        # what the collectives cost an iteration (256x256, batch 4, graph replay): bucketed + overlapped, one blocking all-reduce
        # after the backward pass, and no collective at all
        timing = {}
        active = sharding.collectives_active
        for name, env, off, mb in (("no_collective", "1", True, "64"), ("bucketed_overlapped", "1", False, "64"),
                                   ("blocking_after_backward", "0", False, "64"), ("bucketed_32MB", "1", False, "32"),
                                   ("bucketed_128MB", "1", False, "128"), ("no_collective_again", "1", True, "64")):
            os.environ["LWG_GRAD_BUCKETS"] = env
            os.environ["LWG_BUCKET_MB"] = mb
            sharding.collectives_active = (lambda: False) if off else active
            r = bench_train.measure(4, 256, steps=6, warmup=2, precision="bf16x3", graph=True)
            timing[name] = r["ms_per_iteration"]
            torch.cuda.empty_cache()
        sharding.collectives_active = active
        os.environ.pop("LWG_GRAD_BUCKETS", None)
        os.environ.pop("LWG_BUCKET_MB", None)
        none = 0.5 * (timing["no_collective"] + timing["no_collective_again"])       # first and last measurement: the box's drift
        timing["overlapped_over_none"] = round(timing["bucketed_overlapped"] / none, 4)
        timing["blocking_over_none"] = round(timing["blocking_after_backward"] / none, 4)
        out["train"]["ms_per_iteration_256_b4"] = timing
    sharding.barrier(dev)
    if rank == 0:
        print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
