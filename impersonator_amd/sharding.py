"""Frame sharding across the GPUs of one node.

After `personalize`, frame t depends only on the cached source features, the background, the source face
vertices, tgt_smpl[t] and `first_cam` (= tgt_smpl[0][:3]).  So motion imitation shards embarrassingly:
one process per GPU, each with a full replica of weights + cached source features, round-robin blocks of
`batch` consecutive frames, NO data-path collective.  torch.distributed (RCCL on GPUs, gloo in the CPU
tests) is used only for the timing barrier / max-over-ranks and to collect outputs in frame order.
The reference has no multi-GPU inference at all (it hard-codes `.cuda()` and loops frames, imitator.py:166).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment; (0, 0, 1) standalone."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def forced():
    """LWG_FORCE_DIST=1: initialise the process group and run every collective also at world_size 1 -- how the RCCL
    code path (communicator creation bound to `device_id`, barrier, all-reduce of CUDA tensors, object gather) is
    executed on a box with ONE GPU (tests/test_gpu_rccl.py); RCCL wants a GPU per rank, so two ranks cannot share it."""
    return os.environ.get("LWG_FORCE_DIST", "0") not in ("0", "", "false", "False")


def init_process_group(backend=None):
    """Initialises torch.distributed when launched under torchrun (or at world 1 under LWG_FORCE_DIST=1); returns
    (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # LWG_DIST_BACKEND=gloo: several ranks on ONE GPU (RCCL wants a GPU per rank) -- how the multi-rank code
            # paths are exercised where only one device is visible
            backend = os.environ.get("LWG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kwargs = {}
        if backend == "nccl" and torch.cuda.is_available():
            # RCCL: bind the rank to its GPU before the first collective (barrier() otherwise guesses the device)
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
            kwargs["device_id"] = torch.device("cuda", local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def frame_blocks(num_frames, batch):
    """[(start, end)) blocks of `batch` consecutive frames covering 0..num_frames."""
    return [(s, min(s + batch, num_frames)) for s in range(0, num_frames, batch)]


def shard_blocks(num_frames, batch, rank, world):
    """Blocks owned by `rank`: block b goes to rank b % world (round-robin keeps ranks within one block of each other)."""
    return [blk for i, blk in enumerate(frame_blocks(num_frames, batch)) if i % world == rank]


def barrier(device=None):
    if dist.is_initialized():
        dist.barrier()
    if device is not None and torch.cuda.is_available():
        torch.cuda.synchronize(device)


def max_over_ranks(value, device="cpu"):
    """MAX all-reduce of a python float (the step time every rank must wait for)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_in_frame_order(local_items, num_frames, batch, rank, world):
    """local_items: this rank's per-frame outputs in the order of shard_blocks().  Returns the full
    frame-ordered list on rank 0 (None elsewhere)."""
    if not dist.is_initialized() or (world == 1 and not forced()):
        return list(local_items)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(list(local_items), gathered, dst=0)
    if rank != 0:
        return None
    out = [None] * num_frames
    for r in range(world):
        it = iter(gathered[r])
        for (s, e) in shard_blocks(num_frames, batch, r, world):
            for t in range(s, e):
                out[t] = next(it)
    return out


def imitate_sharded(imitator, tgt_smpls, batch, cam_strategy='smooth', rank=0, world=1):
    """Motion imitation of a whole sequence on `world` ranks (what run_imitator.py runs; the reference loops frames on
    one device, models/imitator.py:157-214): this rank pushes its round-robin blocks through
    Imitator.predict_batches (geometry stream + generator lanes), results go to the host once per block, and rank 0
    gets the (H,W,3) frames of ALL ranks in frame order (None elsewhere).  `first_cam` is frame 0's camera on every
    rank (the reference discovers it at t == 0, imitator.py:243-244).  No data-path collective."""
    import numpy as np
    if len(tgt_smpls) == 0:
        return [] if rank == 0 else None
    smpls = torch.as_tensor(np.asarray(tgt_smpls), dtype=torch.float32).reshape(len(tgt_smpls), -1).cuda()
    n = smpls.shape[0]
    if cam_strategy == 'smooth' and n:
        imitator.first_cam = smpls[0:1, 0:3].clone()
    blocks = shard_blocks(n, batch, rank, world)
    # t = 0 would re-derive first_cam from the chunk, which is only right for the block that starts the sequence
    chunks = ((smpls[s:e], s) for s, e in blocks)
    local = []
    for _, preds in imitator.predict_batches(chunks, cam_strategy):
        local += list(preds.permute(0, 2, 3, 1).cpu().numpy())
    return gather_in_frame_order(local, n, batch, rank, world)


def average_gradients(flat_grads):
    """Data-parallel training: averages a flat gradient tensor over the ranks in place (all-reduce SUM
    then divide) -- RCCL over xGMI for CUDA tensors, gloo on CPU.  No-op without an initialised multi-rank group.  The
    reference does this implicitly through nn.DataParallel (models/impersonator_trainer.py:196-214)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced()):
        dist.all_reduce(flat_grads)
        flat_grads.div_(dist.get_world_size())
    return flat_grads


def collective_info(device=None):
    """What carries the collectives of this process group, for the bench line's `rccl` block: backend name, the number
    of ranks the communicator spans, the RCCL version torch was built against, and a one-element SUM all-reduce on
    `device` that must come back equal to the number of ranks (proof that the communicator really connects them)."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    info = {"backend": dist.get_backend(), "ranks": dist.get_world_size()}
    if info["backend"] == "nccl":
        v = torch.cuda.nccl.version()
        info["version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
        info["library"] = "RCCL (torch.distributed backend 'nccl' on ROCm %s)" % getattr(torch.version, "hip", None)
    dev = device if device is not None else ("cuda" if info["backend"] == "nccl" else "cpu")
    one = torch.ones(1, dtype=torch.float32, device=dev)
    dist.all_reduce(one)
    info["allreduce_of_ones"] = float(one.item())
    return info
