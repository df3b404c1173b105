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


def local_rows(rows, blocks):
    """The rows of `blocks` packed into ONE contiguous tensor, plus each block's [a, b) row range inside it."""
    if not blocks:
        return rows[0:0], []
    if all(blocks[i][1] == blocks[i + 1][0] for i in range(len(blocks) - 1)):
        mine = rows[blocks[0][0]:blocks[-1][1]]          # world 1: already one run, a view
    else:
        mine = torch.cat([rows[s:e] for s, e in blocks], dim=0)
    bounds, a = [], 0
    for s, e in blocks:
        bounds.append((a, a + e - s))
        a += e - s
    return mine.contiguous(), bounds


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
    # this rank's frames packed once (round-robin blocks are not neighbours in `smpls`): consecutive chunks are then adjacent
    # row blocks of one tensor and a round's geometry takes them as a view (Imitator._adjacent_rows) -- no per-round copy kernel
    mine, bounds = local_rows(smpls, blocks)
    # t = 0 would re-derive first_cam from the chunk, which is only right for the block that starts the sequence
    chunks = ((mine[a:b], s) for (a, b), (s, _) in zip(bounds, blocks))
    local = []
    for _, preds in imitator.predict_batches(chunks, cam_strategy):
        local += list(preds.permute(0, 2, 3, 1).cpu().numpy())
    return gather_in_frame_order(local, n, batch, rank, world)


def collectives_active():
    """True when gradient averaging really runs a collective: an initialised group with more than one rank, or
    LWG_FORCE_DIST=1 (the one-GPU test hook).  THE one predicate every data-parallel branch asks (generator and discriminator
    gradient averaging, the bucketed overlap, the graph-replay path)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def average_gradients(flat_grads):
    """Data-parallel training: averages a flat gradient tensor over the ranks in place (all-reduce SUM
    then divide) -- RCCL over xGMI for CUDA tensors, gloo on CPU.  No-op without an initialised multi-rank group.  The
    reference does this implicitly through nn.DataParallel (models/impersonator_trainer.py:196-214)."""
    if collectives_active():
        _all_reduce_mean(flat_grads)
    return flat_grads


def _all_reduce_mean(t):
    """In-place mean over the ranks.  RCCL: one collective with the averaging folded in (ReduceOp.AVG: no second pass over the
    buffer; for a power-of-two number of ranks the same bits as sum-then-divide); gloo has no AVG: sum, then divide."""
    if dist.get_backend() == "nccl":
        dist.all_reduce(t, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(t)
        t.div_(dist.get_world_size())


class GradientBuckets(object):
    """Bucketed gradient averaging overlapped with the backward pass (the reference leaves this to
    nn.DataParallel's reduce, models/impersonator_trainer.py:196-214).

    The flat gradient buffer is cut into contiguous buckets of at least `bucket_bytes` (default 64 MB; RCCL rings over xGMI are
    per-link bound -- few, large messages -- and every bucket is a fork / join of the compute stream, which a captured graph pays for).  The backward pass writes every parameter's gradient exactly once; `written(key)` counts a bucket's
    keys down and `checkpoint()` -- called by the backward pass AFTER it has enqueued the kernels that write them -- hands every
    complete bucket to a side stream: event on the compute stream, all-reduce + 1/world scaling of that slice there, while the
    compute stream goes on with the layers in front.  `finish()` launches what is left and makes the compute stream wait; the
    optimiser step comes after it.  The hand-written backward runs transfer stream -> source stream -> BGNet, and the flat buffer
    is laid out BGNet, source, transfer: it fills back to front, so the last third of the bytes is on the wire while two thirds of
    the backward pass are still to run.  Results equal `average_gradients(flat)` bit for bit (the same all-reduce per element,
    only cut into pieces).  Capturable in a HIP graph: fork and join are events on the capturing stream."""

    def __init__(self, flat, ranges, bucket_bytes=64 << 20):
        """ranges: [(key, lo, hi)] element ranges of `flat`, ascending and contiguous."""
        self.flat = flat
        self.buckets = []          # [lo, hi, set(keys)], ascending
        cur = None
        for key, lo, hi in reversed(list(ranges)):   # cut from the BACK: the buffer fills back to front, full-size buckets first
            if cur is None:
                cur = [lo, hi, {key}]
            else:
                cur[0] = lo
                cur[2].add(key)
            if (cur[1] - cur[0]) * flat.element_size() >= bucket_bytes:
                self.buckets.insert(0, cur)
                cur = None
        if cur is not None:
            if self.buckets:       # a small remainder at the front joins its neighbour (the last bucket to complete anyway)
                self.buckets[0][0] = cur[0]
                self.buckets[0][2] |= cur[2]
            else:
                self.buckets.append(cur)
        self.of_key = {k: i for i, b in enumerate(self.buckets) for k in b[2]}
        self.stream = torch.cuda.Stream(device=flat.device) if flat.is_cuda else None
        self.launched_log = []     # (bucket index, number of keys still unwritten elsewhere) per launch of the last pass: tests read it
        self.begin()

    def begin(self):
        """start of a backward pass"""
        self.left = [len(b[2]) for b in self.buckets]
        self.ready, self.done = [], [False] * len(self.buckets)
        self.launched_log = []
        self.joined = False        # finish() ran for this pass: every bucket averaged, the compute stream waits for the side stream

    def written(self, key):
        i = self.of_key[key]
        self.left[i] -= 1
        if self.left[i] == 0:
            self.ready.append(i)

    def _launch(self, i):
        lo, hi, _ = self.buckets[i]
        view = self.flat[lo:hi]
        self.launched_log.append((i, sum(self.left)))
        if self.stream is None:
            _all_reduce_mean(view)
        else:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                _all_reduce_mean(view)
        self.done[i] = True

    def checkpoint(self):
        """the kernels writing every key reported so far are enqueued: complete buckets go on the wire"""
        ready, self.ready = self.ready, []
        for i in ready:
            self._launch(i)

    def finish(self):
        self.checkpoint()
        for i, d in enumerate(self.done):
            if not d:                      # keys nobody reported (a layer without a gradient this pass): still averaged
                self._launch(i)
        if self.stream is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)
        self.joined = True


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
