"""Secondary measurement (NOT the bench.py contract): one full training iteration (generator update + discriminator
update, impersonator_trainer.py:350-366) on an MI355X, synthetic inputs.

    python tools/bench_train.py [--batch 4] [--image-size 256] [--steps 5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py ...

Under torchrun every rank trains on its own `--batch` images (data parallel, BASELINE.json config 5) and the two flat
gradient buffers (G 390 MB, D 27.8 MB) are averaged over the ranks between backward and the Adam step
(sharding.average_gradients: RCCL over xGMI; LWG_DIST_BACKEND=gloo lets several ranks share one GPU for a functional
run).  The line then also carries the time of those two all-reduces alone, measured on the same buffers.
`measure()` is what bench.py's `secondary.train` block calls."""
import argparse
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impersonator_amd import sharding  # noqa: E402
from impersonator_amd.models.impersonator_trainer import Impersonator  # noqa: E402


def build(batch, image_size, precision="bf16x3", script_loss=False, seed=0):
    opt = types.SimpleNamespace(image_size=image_size, batch_size=batch, map_name='uv_seg', norm_type='instance',
                                repeat_num=6, is_train=True, conv_precision=precision)
    if script_loss:
        from tests import helpers
        opt.mask_bce, opt.use_vgg, opt.use_face = True, True, True
        opt.vgg_weights, opt.face_model = helpers.vgg19_state_dict(0), helpers.sphere20a_state_dict(0)
        opt.lambda_face, opt.lambda_mask, opt.lambda_mask_smooth = 5.0, 1.0, 1.0
    torch.manual_seed(seed)          # every rank starts from the same parameters
    model = Impersonator(opt)
    model._G.init_weights()
    model._D.init_weights()
    g = torch.Generator().manual_seed(seed + 1 + sharding.env_world()[0])   # ... and trains on its own images
    n, s = batch, image_size
    r = lambda *sh: (torch.rand(*sh, generator=g) * 2 - 1).cuda()
    # head boxes as BodyRecoveryFlow.cal_head_bbox would give them (a 1/5-size box near the top)
    boxes = torch.tensor([[s * 2 // 5, s * 3 // 5, s // 10, s * 3 // 10]] * n) if script_loss else None
    model.set_input(r(n, 6, s, s), r(n, 3, s, s), input_G_bg=r(n, 4, s, s), input_G_src=r(n, 6, s, s),
                    T=(torch.rand(n, s, s, 2, generator=g) * 2.4 - 1.2).cuda(), real_src=r(n, 3, s, s),
                    bg_mask=(torch.rand(2 * n, 1, s, s, generator=g) > 0.5).float().cuda(), head_bbox=boxes)
    return model


def conv_gflop_per_iteration(batch, image_size, repeat=6):
    """Algorithmic work of the convolutions of one training iteration (2 * output pixels * Cout * taps * Cin per conv, real taps and
    channels), in GFLOP.  Forward of one image: the two ResUnets (networks/generator.py:68-184: 7x7 stem, three stride-2 convs, 2 x
    `repeat` trunk convs, three transposed convs, three skipper convs on the concatenated maps, 7x7 heads with 3 + 1 outputs) and
    the BGNet (generator.py:23-65: the same without skippers, 4 input channels, 3 outputs).  A training iteration runs every conv
    three times (forward, data gradient, weight gradient) on `batch` images; the PatchGAN discriminator (networks/discriminator.py:
    8-57: 4x4 convs 6-64-128-256-512 stride 2, 512-512 and 512-1 stride 1) runs forward + both gradients on 2 x batch images for
    its own update and forward + data gradient on `batch` images for the generator's adversarial term.  Loss networks (--script-loss)
    are not counted."""
    s = float(image_size)

    def conv(cin, cout, k, out_edge):
        return 2.0 * out_edge * out_edge * cout * k * k * cin

    def unet(cin, heads_out, skippers):
        f = conv(cin, 64, 7, s)
        for i in range(3):
            f += conv(64 << i, 128 << i, 3, s / (2 << i))
        f += 2 * repeat * conv(512, 512, 3, s / 8)
        for i in range(3):
            c = 512 >> i
            f += 2.0 * (s / (8 >> i)) ** 2 * (c // 2) * 9 * c        # ConvTranspose2d: 9 taps per INPUT pixel over its four phases
            if skippers:
                f += conv(c, c // 2, 3, s / (4 >> i))
        return f + conv(64, heads_out, 7, s)

    g_fwd = 2 * unet(6, 4, True) + unet(4, 3, False)
    e = s / 2
    d_fwd = conv(6, 64, 4, e) + conv(64, 128, 4, e / 2) + conv(128, 256, 4, e / 4) + conv(256, 512, 4, e / 8) + \
        conv(512, 512, 4, e / 8 - 1) + conv(512, 1, 4, e / 8 - 2)
    return (3.0 * batch * g_fwd + 3.0 * 2 * batch * d_fwd + 2.0 * batch * d_fwd) / 1e9


def measure(batch, image_size, steps=5, warmup=1, precision="bf16x3", script_loss=False, graph=False):
    """-> dict(ms_per_iteration, images_per_s, ...) of `steps` optimize_parameters() calls after `warmup`; under an
    initialised multi-rank group: max over ranks, images of all ranks."""
    rank, _, world = sharding.env_world()
    dev = torch.device("cuda", torch.cuda.current_device())
    model = build(batch, image_size, precision, script_loss)
    # graph: Impersonator.optimize_parameters_graphed -- two eager iterations, one that captures, then replays (single process)
    iterate = model.optimize_parameters_graphed if graph else model.optimize_parameters
    for _ in range(max(4 if graph else 1, warmup)):
        losses = iterate()
    sharding.barrier(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = iterate()
    sharding.barrier(dev)
    import torch.distributed as dist
    rdev = dev if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl" else "cpu"   # RCCL reduces device tensors
    dt = sharding.max_over_ranks((time.perf_counter() - t0) / steps, rdev)
    out = {"ms_per_iteration": round(dt * 1e3, 2), "launch": "one HIP graph replay per iteration" if graph and model._graph is not None else "eager", "images_per_s": round(world * batch / dt, 2), "batch_per_rank": batch,
           "world": world, "image_size": image_size,
           "dtype": "f32" if precision == "fp32" else "bf16x3 convs of generator and discriminator (forward, data and weight gradient) + f32",
           "loss": "train_iPER.sh (mask_bce, vgg, face)" if script_loss else "adv + L1 + mask",
           "conv_gflop_per_iteration": round(conv_gflop_per_iteration(batch, image_size), 1),
           "conv_tflops": round(world * conv_gflop_per_iteration(batch, image_size) / dt / 1e3, 1),
           "conv_tflops_note": "algorithmic conv work (forward + data gradient + weight gradient of the three generator streams, the "
                               "discriminator update and the adversarial term) / iteration time; the bf16x3 kernels' ceiling is "
                               "833 TFLOP/s algorithmic (three products per multiply-add), fp32's 157",
           "losses": {k: round(v, 6) for k, v in losses.items()}}
    peak = 2500.0 / 3.0 if precision != "fp32" else 157.3
    out["roofline"] = {"bound": "mfma", "achieved": out["conv_tflops"], "peak": round(peak, 1), "unit": "TFLOP/s",
                       "frac": round(out["conv_tflops"] / peak, 4),
                       "what": "algorithmic conv FLOP of the whole iteration / iteration time, against the matrix pipe's ceiling for "
                               "this arithmetic (bf16x3: 2500 / 3); everything that is not a convolution (norms, losses, Adam, "
                               "re-layouts, launch gaps) counts against it",
                       "per_kernel": "profiles/r05_train_kernel_stats.md (rocprofv3 --kernel-trace --stats of tools/bench_train.py)"}
    if world > 1:
        # the two collectives of an iteration on their own: the same buffers, the same call
        fg, dg = model._generator_trainer().flat_g, model._D.flat_buffers()[1]
        keep_g, keep_d = fg.clone(), dg.clone()
        ms = {}
        for name, buf in (("G", fg), ("D", dg)):
            sharding.average_gradients(buf)
            sharding.barrier(dev)
            t0 = time.perf_counter()
            for _ in range(3):
                sharding.average_gradients(buf)
            torch.cuda.synchronize(dev)
            ms[name] = sharding.max_over_ranks((time.perf_counter() - t0) / 3, rdev) * 1e3
        fg.copy_(keep_g)
        dg.copy_(keep_d)
        import torch.distributed as dist
        out["all_reduce"] = {"backend": dist.get_backend(), "G_bytes": fg.numel() * 4, "D_bytes": dg.numel() * 4,
                             "G_ms": round(ms["G"], 3), "D_ms": round(ms["D"], 3),
                             "note": "ring all-reduce over xGMI moves 2(N-1)/N x bytes per GPU at <= ~153 GB/s per link"}
    model._D.release()
    model._G.release()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4, help="images per rank")
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--script-loss", action="store_true",
                    help="the loss of scripts/train_iPER.sh: --mask_bce --use_vgg --use_face (seeded VGG19 / Sphere20a weights)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"],
                    help="forward / data-gradient convolutions of the generator update")
    ap.add_argument("--graph", action="store_true", help="replay the iteration as a captured HIP graph (single process)")
    a = ap.parse_args()
    rank, local_rank, world = sharding.init_process_group()
    if local_rank >= torch.cuda.device_count() and os.environ.get("LWG_DIST_BACKEND") == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    line = measure(a.batch, a.image_size, a.steps, 1, a.precision, a.script_loss, a.graph)
    if rank == 0:
        line = dict({"metric": "training iteration (G update + D update)",
                     "note": "BASELINE.json config 5 is --image-size 512 (--batch 1..4 per GPU)", "batch": a.batch}, **line)
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
