"""Secondary measurement (NOT the bench.py contract): one PatchGAN discriminator update on an MI355X.

    python tools/bench_discriminator.py [--batch 8] [--image-size 256] [--steps 20]

Prints one JSON line: ms per update (forward of 2N images + backward + Adam), algorithmic conv TFLOP/s
(forward + data gradient + weight gradient of the six 4x4 convs), under torch.distributed.run also with the gradient
all-reduce (RCCL) between backward and Adam."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impersonator_amd import sharding  # noqa: E402
from impersonator_amd.networks.discriminator import PatchDiscriminator  # noqa: E402


def conv_flops(image_size, n_layers=4, input_nc=6, ndf=64):
    """fwd + dgrad (all but the first layer) + wgrad FLOPs per image."""
    H, cin, total = image_size, input_nc, 0.0
    chans = [ndf] + [ndf * min(2 ** n, 8) for n in range(1, n_layers)] + [ndf * min(2 ** n_layers, 8), 1]
    for l, cout in enumerate(chans):
        Ho = H // 2 if l < n_layers else H - 1
        f = 2.0 * Ho * Ho * cout * 16 * cin
        total += f * (2 if l == 0 else 3)
        H, cin = Ho, cout
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    rank, local_rank, world = sharding.init_process_group()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    D = PatchDiscriminator(6, 64, 4, 'instance', False, image_size=args.image_size, max_batch=args.batch)
    D.init_weights()
    D = D.to(dev)
    gen = torch.Generator().manual_seed(rank)
    real = (torch.rand(args.batch, 6, args.image_size, args.image_size, generator=gen) * 2 - 1).to(dev)
    fake = (torch.rand(args.batch, 6, args.image_size, args.image_size, generator=gen) * 2 - 1).to(dev)
    for _ in range(5):
        D.optimize_D(real, fake)
    sharding.barrier(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = D.optimize_D(real, fake)
    sharding.barrier(dev)
    dt = sharding.max_over_ranks(time.perf_counter() - t0, dev if world > 1 else "cpu")
    if rank == 0:
        fl = conv_flops(args.image_size) * 2 * args.batch
        print(json.dumps({"metric": "PatchGAN discriminator update", "ms_per_update": round(dt / args.steps * 1e3, 3),
                          "images_per_s": round(world * args.batch * args.steps / dt, 1), "n_gpus": world,
                          "batch_per_gpu": args.batch, "image_size": args.image_size,
                          "conv_gflop_per_update": round(fl / 1e9, 1),
                          "conv_tflops": round(fl * args.steps / dt / 1e12, 2), "dtype": "f32",
                          "grad_allreduce_mb": round(sum(p.numel() for p in D.parameters()) * 4 / 1e6, 1) if world > 1 else 0,
                          "last_loss": float(loss)}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
