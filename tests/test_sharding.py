"""N>1 path on CPU: frame sharding + timing reduction over torch.distributed (gloo, world_size 2)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from impersonator_amd import sharding


def test_blocks_cover_every_frame_once():
    for n, b, w in ((1024, 8, 8), (1000, 8, 3), (7, 8, 2), (16, 4, 1)):
        seen = []
        for r in range(w):
            for s, e in sharding.shard_blocks(n, b, r, w):
                assert 0 < e - s <= b
                seen += list(range(s, e))
        assert sorted(seen) == list(range(n))
    # round-robin: per-rank block counts differ by at most one
    counts = [len(sharding.shard_blocks(1000, 8, r, 3)) for r in range(3)]
    assert max(counts) - min(counts) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_frames, batch, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    r, lr, w = sharding.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    mine = []
    for s, e in sharding.shard_blocks(num_frames, batch, rank, world):
        # stand-in for Imitator.inference on this rank's frames: the "image" encodes its frame index
        mine += [np.full((2, 2, 3), t, np.float32) for t in range(s, e)]
    sharding.barrier()
    slow = sharding.max_over_ranks(1.0 + rank)          # the slowest rank defines the step time
    total = sharding.sum_over_ranks(len(mine))
    full = sharding.gather_in_frame_order(mine, num_frames, batch, rank, world)
    if rank == 0:
        ret["max"] = slow
        ret["total"] = total
        ret["order_ok"] = all(float(full[t][0, 0, 0]) == t for t in range(num_frames))
    else:
        assert full is None
    dist.destroy_process_group()


def test_two_rank_gloo_run():
    world, num_frames, batch = 2, 37, 8
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), num_frames, batch, ret), nprocs=world, join=True)
    assert ret["max"] == 2.0 and ret["total"] == num_frames and ret["order_ok"]


def test_single_process_is_a_noop():
    assert sharding.env_world()[2] >= 1
    assert sharding.max_over_ranks(3.5) == 3.5
    assert sharding.gather_in_frame_order([1, 2, 3], 3, 8, 0, 1) == [1, 2, 3]


def test_local_rows_pack_a_ranks_blocks():
    rows = torch.arange(40 * 3, dtype=torch.float32).reshape(40, 3)
    for world in (1, 2, 3):
        for rank in range(world):
            blocks = sharding.shard_blocks(37, 8, rank, world)
            mine, bounds = sharding.local_rows(rows[:37], blocks)
            assert mine.is_contiguous() and len(bounds) == len(blocks)
            for (a, b), (s, e) in zip(bounds, blocks):
                assert torch.equal(mine[a:b], rows[s:e])
            if world == 1:
                assert mine.data_ptr() == rows.data_ptr()        # one run: a view, no copy


def _bucket_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    sharding.init_process_group(backend="gloo")
    assert sharding.collectives_active()
    g = torch.Generator().manual_seed(100 + rank)
    sizes = [5, 300, 900, 1200, 64, 900, 3]                      # "parameters", in layout order
    ranges, off = [], 0
    for i, n in enumerate(sizes):
        ranges.append(("p%d" % i, off, off + n))
        off += n
    flat = torch.randn(off, generator=g)
    want = sharding.average_gradients(flat.clone())            # one blocking all-reduce of the whole buffer
    b = sharding.GradientBuckets(flat, ranges, bucket_bytes=1000 * 4)
    # buckets: contiguous, cover the buffer, cut from the back, each >= 1000 elements; the small remainder at the front joins bucket 0
    assert b.buckets[0][0] == 0 and b.buckets[-1][1] == off
    assert all(b.buckets[i][1] == b.buckets[i + 1][0] for i in range(len(b.buckets) - 1))
    assert all(hi - lo >= 1000 for lo, hi, _ in b.buckets) and len(b.buckets) == 2
    for _pass in range(2):                                      # a second backward pass re-arms the counters
        flat.copy_(torch.randn(off, generator=torch.Generator().manual_seed(100 + rank)))
        b.begin()
        # the backward pass writes back to front; the LAST bucket is complete (and goes on the wire) while keys of the first are open
        for key in ("p6", "p5", "p4"):
            b.written(key)
        b.checkpoint()
        assert b.launched_log == []                             # p3 is still open
        b.written("p3")
        b.checkpoint()
        assert b.launched_log == [(1, 3)], b.launched_log       # bucket 1 launched with 3 keys still unwritten
        for key in ("p2", "p1"):
            b.written(key)
        b.checkpoint()
        assert len(b.launched_log) == 1                         # p0 is still missing: bucket 0 waits
        b.finish()                                              # nobody reported p0: finish() averages the bucket anyway
        assert [i for i, _ in b.launched_log] == [1, 0]
        assert torch.equal(flat, want), float((flat - want).abs().max())
    if rank == 0:
        ret["ok"] = True
    dist.destroy_process_group()


def test_gradient_buckets_equal_the_blocking_all_reduce():
    """sharding.GradientBuckets (the overlapped, bucketed gradient averaging of the data-parallel trainer) on two gloo ranks:
    bucket layout, early launch of complete buckets, and the result == average_gradients of the whole buffer, bit for bit."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bucket_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret["ok"]
