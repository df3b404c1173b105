#!/usr/bin/env python
"""bench.py -- frames/sec of the Imitator.forward() hot path on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of 8 synthetic 256x256 target frames of ONE personalised
source (BASELINE.json configs[1]): camera policy -> SMPL vertices -> project/rasterise -> cond, flow T, warped
source (one fused launch sequence) -> tsf ResUnet with the Liquid Warping Block adds -> tanh/sigmoid heads and the
background blend.  Inputs (SMPL vectors, source image, cached source features) are resident in HBM before the timed
region; the per-batch device->host copy of the result is outside it.  Every rank runs the same per-GPU work on its
own frames (weak scaling, no data-path collective).

Launching.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under
`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU, RCCL); if fewer than N devices are visible it exits 1 instead
of measuring one GPU and printing `n_gpus: 1`.  A WORLD_SIZE that disagrees with --gpus is an error, too.

Timing.  After the warm-up the `--steps` window is timed `--repeats` times (default 5), every window bracketed by a barrier and a
device synchronisation on both sides and reduced with MAX over the ranks; `ms_per_step` / `value` are the MEDIAN window, the line
also carries `ms_per_step_min/max`, the per-window list and the GPU's clock / power sampled (amdsmi) while the windows ran -- the part
is power-limited and boxes differ by +-5 %, a single 45 ms window is as noisy as a round's gain.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the dominant implicit-GEMM conv kernel, algorithmic FLOP / HIP-event time of its launches, as a
                 fraction of the MFMA peak both ways (algorithmic and executed products); exact_fp32_mode has its own
  cpu_baseline : the CPU oracle (port of the reference's PyTorch path + C rasteriser) timed on this box's host cores
  parity       : the timed pipeline re-run on 16 frames after the timed region and compared with the oracle -- from the device's
                 posed vertices (`linf`, `fim_mismatch`) and from theta with the oracle's own SMPL in fp64 (`theta_chain`: the
                 whole chain on identical SMPL inputs); a failed check marks the line `"invalid"` and the process exits 1
  rccl         : with a process group (N > 1, or LWG_FORCE_DIST=1): backend, communicator size, RCCL version, and an all-reduce
                 of ones that must equal the rank count
  ranks        : (N > 1) per-rank frames/s of the median window measured to each rank's OWN device synchronisation (a straggler
                 shows here, not only in the max), min / median / max, every rank's device name + UUID / PCI id, `single_rank_fps`
                 = rank 0 running the same window alone while the others idle, and `linear_frac` = value / (N x single_rank_fps)
  secondary    : after the timed region -- `swap`: Swapper.swap (BASELINE config 4, appearance transfer with the
                 two-stream Liquid Warping Block) at 256x256, one pair and eight pairs per launch sequence, with the
                 same HIP-event roofline pass; `train`: one G + D training iteration (config 5) at 512x512 batch 1
                 and 256x256 batch 4; `latency`: ONE frame per call as the reference loops it (models/imitator.py:166-171), eager
                 and as a captured HIP graph, both precisions, and one Swapper.swap; `personalize`: Imitator.personalize (once
                 per source) with the generator's BGNet and with the InpaintSANet background model, the CPU oracle beside it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from impersonator_amd import demo, sharding  # noqa: E402

# /opt/skills/guides/MI355X_MICROARCH.md, dense MFMA peaks: v_mfma_f32_32x32x2_f32 157.3 TFLOP/s, v_mfma_f32_32x32x16_bf16
# 2.5 PFLOP/s.  The bf16x3 kernels execute 3 bf16 products per algorithmic multiply-add, so against ALGORITHMIC
# flops (105.58 GFLOP/frame, what `achieved` counts) their ceiling is 2500/3.
FP32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0


def kernel_peak(name):
    return BF16_MFMA_PEAK_TFLOPS / 3.0 if "bf16x3" in name else FP32_MFMA_PEAK_TFLOPS

BATCH = 8
IMAGE_SIZE = 256
TRAFFIC_FILE = "r05_traffic.json"


# sources of the kernels the timed step launches (the training and inpainting kernels are not among them)
STEP_SOURCES = ("common.h", "conv.h", "conv.hip", "direct.hip", "generator.hip", "heads.hip", "raster.hip", "sample.h", "smpl.hip",
                "warp.hip")


def csrc_digest():
    """sha256 over the sources of the timed step's kernels: what profiles/*_traffic.json is stamped with by
    tools/summarize_profile.py (which calls this function)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "impersonator_amd", "csrc")
    for f in STEP_SOURCES:
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--repeats", type=int, default=5,
                   help="timed windows of --steps steps each; the line reports the median window (min / max beside it)")
    p.add_argument("--settle-ms", type=float, default=300.0,
                   help="untimed steps before the warm-up until this much wall time has passed (DVFS: the first ~100 ms "
                        "after idle run ~15%% slower); 0 disables")
    p.add_argument("--frames", type=int, default=1024, help="length of the synthetic reference sequence")
    p.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle timing (profiling runs)")
    p.add_argument("--no-roofline", action="store_true", help="skip the HIP-event pass over the conv kernel")
    p.add_argument("--precision", choices=["bf16x3", "fp32"], default=None,
                   help="conv arithmetic of the per-frame stream (default: the library default, bf16x3)")
    p.add_argument("--no-fp32-mode", action="store_true", help="skip the extra timed pass in exact-fp32 mode")
    p.add_argument("--no-secondary", action="store_true",
                   help="skip the `secondary` block (appearance transfer = BASELINE config 4, training iteration = config 5)")
    p.add_argument("--lanes", type=int, default=None,
                   help="generator engines/streams consecutive batches are dealt to (default: Imitator.lanes = 2)")
    return p.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) started bare, i.e. without torch.distributed.run's environment: re-execute under it, one
    rank per GPU.  With fewer than N visible devices the run is refused (exit 1) -- it would otherwise measure ONE GPU.  Test hook:
    LWG_DIST_BACKEND=gloo lets the N ranks share the visible device(s) (RCCL wants a GPU per rank)."""
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    shared = os.environ.get("LWG_DIST_BACKEND") == "gloo" and ndev >= 1
    if ndev < args.gpus and not shared:
        sys.stderr.write("bench.py: --gpus %d but %d GPU(s) visible on this box: refusing to run (a line with n_gpus < --gpus "
                         "would be a single-GPU measurement).  Start fewer ranks, or for a functional run of the N-rank code path "
                         "on shared devices set LWG_DIST_BACKEND=gloo.\n" % (args.gpus, ndev))
        sys.exit(1)
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without a torch.distributed environment: launching %s\n" % (args.gpus, " ".join(cmd[1:8])))
    sys.stderr.flush()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), LWG_SELF_LAUNCHED="1")
    os.execve(sys.executable, cmd, env)


class GpuSampler:
    """gfx clock (MHz) and socket power (W) of one GPU sampled through amdsmi on a thread while the timed windows run; every
    failure (no amdsmi, no permission, another field layout) degrades to `None` in the line, never to an exception."""

    def __init__(self, dev_index):
        self.samples, self._stop, self._thread, self._h, self.error = [], False, None, None, None
        try:
            import amdsmi
            self._smi = amdsmi
            amdsmi.amdsmi_init()
            handles = amdsmi.amdsmi_get_processor_handles()
            want = None
            try:
                want = torch.cuda.get_device_properties(dev_index).pci_bus_id
            except Exception:
                pass
            for h in handles:
                try:
                    bdf = amdsmi.amdsmi_get_gpu_device_bdf(h)
                    if want is not None and int(bdf.split(":")[1], 16) == int(want):
                        self._h = h
                except Exception:
                    pass
            if self._h is None and handles:
                self._h = handles[min(dev_index, len(handles) - 1)]
        except Exception as e:   # noqa: BLE001
            self.error = "%s: %s" % (type(e).__name__, e)

    def _read(self):
        smi, out = self._smi, {}
        try:
            c = smi.amdsmi_get_clock_info(self._h, smi.AmdSmiClkType.GFX)
            out["mhz"] = float(c.get("clk", c.get("cur_clk")))
        except Exception:
            pass
        try:
            pw = smi.amdsmi_get_power_info(self._h)
            for k in ("current_socket_power", "average_socket_power", "socket_power"):
                v = pw.get(k)
                if isinstance(v, (int, float)) and v > 0:
                    out["watt"] = float(v)
                    break
        except Exception:
            pass
        return out

    def _loop(self):
        while not self._stop:
            r = self._read()
            if r:
                self.samples.append(r)
            time.sleep(0.004)

    def start(self):
        if self._h is not None:
            import threading
            self._stop = False
            self._thread = threading.Thread(target=self._loop, daemon=True)
            self._thread.start()

    def stop(self):
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=1.0)
            self._thread = None

    def summary(self):
        def stat(key):
            v = sorted(s[key] for s in self.samples if key in s)
            return None if not v else {"min": v[0], "median": v[len(v) // 2], "max": v[-1]}
        if not self.samples:
            return {"samples": 0, "note": "amdsmi gave no sample" + (" (%s)" % self.error if self.error else "")}
        return {"samples": len(self.samples), "gfx_clock_mhz": stat("mhz"), "socket_power_w": stat("watt"),
                "note": "amdsmi, ~4 ms period, this rank's GPU, only while the timed windows ran"}


def device_identity(dev):
    """name + a stable identifier of this rank's GPU (what proves that N ranks sit on N different devices)."""
    p = torch.cuda.get_device_properties(dev)
    ident = {"name": p.name, "index": dev.index}
    for k in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
        v = getattr(p, k, None)
        if v is not None:
            ident[k] = str(v)
    return ident


def cpu_baseline(seed=0, batches=2):
    """Times the CPU oracle (kind 'port': oracle/torch_ref.py + oracle/raster_ref.c, see their headers) on the same
    workload: one warm-up + `batches` batches of 8 frames, all host cores."""
    from oracle import torch_ref
    from impersonator_amd.networks.batch_smpl import SMPL, synthetic_smpl_params
    from impersonator_amd.networks.generator import ImpersonatorGenerator
    from impersonator_amd.utils import synthetic

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    rest, faces = synthetic.body_mesh()
    faces_t = torch.from_numpy(faces)
    map_fn = torch.from_numpy(synthetic.uv_seg_map_fn(rest, faces))
    G = ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6)
    shapes = [(k, tuple(v.shape)) for k, v in G.state_dict().items()]
    sd = torch_ref.state_dict_from_numpy(synthetic.random_state_dict(shapes, seed=seed, affine="identity"))
    # the oracle's SMPL (== the reference's SMPL.forward, tests/test_oracle_vs_reference.py) on float64 tensors, rounded to fp32:
    # the correctly rounded vertices, which is also what the device's default `compensated` SMPL mode produces -- so the
    # frames computed here double as the theta -> image parity vectors.  (SMPL is 0.02 of 105.6 GFLOP per frame: its dtype
    # does not show in the timing.)
    sm = torch_ref.smpl_tensors(SMPL(params=synthetic_smpl_params(seed)), torch.float64)
    src_smpl = torch.from_numpy(demo.synthetic_smpls(1, seed + 1))
    src_smpl[:, 3:75] = 0
    src_img = torch.from_numpy(synthetic.smooth_image(seed + 11))
    bg_img = torch.from_numpy(synthetic.smooth_image(seed + 12))
    smpls = torch.from_numpy(demo.synthetic_smpls(1024, seed))
    with torch.no_grad():
        si = torch_ref.get_details(sm, src_smpl)
        sf2v, sfim, _ = torch_ref.render_fim_wim(si["cam"], si["verts"], faces_t)
        p2v = torch_ref.source_p2verts(sf2v)
        scond = torch_ref.encode_fim(sfim, map_fn)
        ft = 1 - torch_ref.morph(scond[:, -1:], 3, "erode")
        enc, res = torch_ref.encode_src(sd, torch.cat([src_img * ft, scond], 1))

        def one_batch(b):
            chunk = smpls[b * BATCH:(b + 1) * BATCH]
            cam = si["cam"].expand(BATCH, -1).clone()
            cam[:, 1:] += chunk[:, 1:3] - smpls[0:1, 1:3]
            info = torch_ref.get_details(sm, torch.cat([cam, chunk[:, 3:75], si["shape"].expand(BATCH, -1)], 1))
            fr = torch_ref.transfer_frame(src_img, p2v, info["cam"], info["verts"], faces_t, map_fn)
            return fr["fim"], torch_ref.imitator_forward(sd, enc, res, bg_img, fr["tsf_inputs"], fr["T"])[0]

        # a 256-thread intra-op pool is slower than 32 threads on these convs: pick the best of a few pool sizes on one
        # batch each (the first call also warms up), then time `batches` batches with it
        best = (None, float("inf"))
        for nt in sorted({min(cores, n) for n in (16, 32, 64)}):
            torch.set_num_threads(nt)
            one_batch(0)
            t0 = time.perf_counter()
            one_batch(0)
            d1 = time.perf_counter() - t0
            if d1 < best[1]:
                best = (nt, d1)
        cores = best[0]
        torch.set_num_threads(cores)
        t0 = time.perf_counter()
        kept = [one_batch(b) for b in range(1, batches + 1)]
        dt = time.perf_counter() - t0
    line = {"value": round(batches * BATCH / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d batches of %d frames (256x256) after warm-up, best of 16/32/64 intra-op threads on a box with %d "
                      "logical cores, torch %s CPU fp32 + OpenMP C rasteriser" % (batches, BATCH, os.cpu_count(), torch.__version__),
            "port_vs_reference": "calibration where /root/reference exists (8-core build container, 5 batches each): the "
                                 "port runs at 0.90x the speed of the reference's own modules (median; per-batch spread "
                                 "0.76-1.06x) with bit-identical outputs -- profiles/r02_port_vs_reference.md"}
    return line, {"fim": torch.cat([k[0] for k in kept]), "pred": torch.cat([k[1] for k in kept]), "first_batch": 1, "src_fim": sfim}


def parity_block(imitator, src_img, bg_img, smpls, lanes, theta_chain):
    """Runs, OUTSIDE the timed region, the two batches the CPU baseline computed (frames 8..23) through the pipeline
    that was timed and checks them against the oracle: (a) `same_vertices` -- the oracle restarts from the posed
    vertices the device produced (the definition every parity test uses: a 1e-6 difference in a vertex can move a
    face edge across a pixel centre, which is a different, legitimate image); (b) `theta_chain` -- the oracle's
    own SMPL from the same theta, reported with the number of face-index pixels that differ."""
    from oracle import torch_ref
    b0 = theta_chain["first_batch"]
    nb = theta_chain["pred"].shape[0] // BATCH
    chunks = [(smpls[b * BATCH:(b + 1) * BATCH], b * BATCH) for b in range(b0, b0 + nb)]
    preds, verts, cams, fims, Ts = [], [], [], [], []
    for _, p in imitator.predict_batches(iter(chunks), "smooth", lanes=lanes):
        info = imitator.tsf_info
        preds.append(p.cpu())
        verts.append(info["verts"].cpu())
        cams.append(info["cam"].cpu())
        fims.append(info["fim"].cpu())
        Ts.append(info["T"].cpu())
    pred, fim = torch.cat(preds), torch.cat(fims)
    sd = {k: v.detach().cpu() for k, v in imitator.generator.state_dict().items()}
    faces_t, map_fn, si = imitator.render.faces.cpu(), imitator.render.map_fn.cpu(), imitator.src_info
    src_t, bg_t = torch.from_numpy(src_img)[None], torch.from_numpy(bg_img)[None]
    with torch.no_grad():
        src = torch_ref.personalize(sd, src_t, si["cam"].cpu(), si["verts"].cpu(), faces_t, map_fn, ft_ks=imitator._opt.ft_ks)
        fr, ref = torch_ref.imitator_frames(sd, src, src_t, bg_t, torch.cat(cams), torch.cat(verts), faces_t, map_fn)
    agree = (fim == theta_chain["fim"])
    d_chain = (pred - theta_chain["pred"]).abs()
    frames_agree = agree.flatten(1).all(1)
    linf = float((pred - ref).abs().max())
    fim_mismatch = int((fim != fr["fim"]).sum()) + int((si["fim"].cpu() != src["fim"]).sum())
    smpl_mode = imitator.hmr.smpl.precision
    chain_fim = int((~agree).sum()) + int((si["fim"].cpu() != theta_chain["src_fim"]).sum())
    chain_ok = (chain_fim == 0 and float(d_chain.max()) <= 1e-3) if smpl_mode == "compensated" else True
    return {"ok": bool(linf <= 1e-3 and fim_mismatch == 0 and chain_ok),
            "frames": int(pred.shape[0]), "linf": round(linf, 7), "fim_mismatch": fim_mismatch,
            "T_linf": round(float((torch.cat(Ts) - fr["T"]).abs().max()), 9),
            "bound": 1e-3, "oracle": "same_vertices: oracle/torch_ref.py + raster_ref.c restarted from the device's posed "
                                     "vertices; pipeline = the timed one (%d lanes)" % lanes,
            "theta_chain": {"ok": bool(chain_ok), "smpl_precision": smpl_mode, "fim_mismatch_pixels": chain_fim,
                            "frames_with_identical_fim": int(frames_agree.sum()),
                            "linf_on_those_frames": (round(float(d_chain[frames_agree].max()), 7)
                                                     if bool(frames_agree.any()) else None),
                            "linf_all": round(float(d_chain.max()), 7),
                            "note": "oracle's own SMPL from the same theta in fp64, rounded (= the reference's SMPL.forward on float64 "
                                    "tensors); device SMPL in its `compensated` mode (fp64 intermediates, one rounding): part of `ok`.  "
                                    "Against ONE fp32 evaluation there is no 1e-3 answer: the reference's own SMPL moves the image by "
                                    "1.2e-3..3e-3 with the thread count / frames per call (profiles/r04_theta_chain_reference_self.md)"}}


def conv_roofline(generator, run, steps=1):
    """HIP-event pass over the conv kernels of `run()` (liblwg records events around every conv launch on the launch
    stream): the dominant kernel and the all-conv figure, both ways (see main()'s roofline_pass for the definitions)."""
    generator.profile(True)
    run()
    n, ms, flops = generator.profile_read()
    table = generator.profile_table()
    generator.profile(False)
    name, (kn, kms, kfl) = max(table.items(), key=lambda kv: kv[1][1])
    achieved = kfl / (kms * 1e-3) / 1e12
    x3 = "bf16x3" in name
    peak = BF16_MFMA_PEAK_TFLOPS if x3 else FP32_MFMA_PEAK_TFLOPS
    ideal_ms = sum(v[2] / (kernel_peak(k) * 1e12) * 1e3 for k, v in table.items())
    return {"bound": "mfma", "kernel": name, "achieved": round(achieved, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "frac_pipe": round(achieved * (3.0 if x3 else 1.0) / peak, 4),
            "launches": kn, "avg_launch_ms": round(kms / max(kn, 1), 5),
            "all_conv_kernels": {"achieved": round(flops / (ms * 1e-3) / 1e12, 3), "frac_pipe": round(ideal_ms / ms, 4),
                                 "launches": n, "ms": round(ms / steps, 4)}}


def secondary_swap(dev, steps=30):
    """BASELINE config 4: appearance transfer, models/swapper.py:198-271 (Swapper.swap: part masks, calculate_trans, two
    image warps, generator.swap = tsf ResUnet with TWO warped source-feature sets per Liquid-Warping-Block level,
    networks/generator.py:245-275, blend) on two synthetic subjects at 256x256.  `one_pair`: Swapper.swap as the
    reference calls it (batch 1), back to back on one stream.  `eight_pairs`: the generator part of eight swaps as one
    launch sequence (batch 8 through lwg_generator_swap, the two subjects' cached features shared) -- the throughput
    form, with the HIP-event roofline of its conv kernels."""
    from impersonator_amd.utils import synthetic
    sw, smpl_a, img_a, bg_a = demo.build_synthetic_imitator(batch_size=BATCH, seed=0, image_size=IMAGE_SIZE, model="swapper")
    smpl_b = demo.synthetic_smpls(8, seed=3)[5]
    img_b = synthetic.smooth_image(77, (1, 3, IMAGE_SIZE, IMAGE_SIZE))[0]
    bg_b = synthetic.smooth_image(78, (1, 3, IMAGE_SIZE, IMAGE_SIZE))[0]
    sw.swap_setup(img_a, img_b, src_smpl=smpl_a, tgt_smpl=smpl_b, src_bg=bg_a, tgt_bg=bg_b)
    src, tgt = sw.src_info, sw.tsf_info

    def timed(fn, n):
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize(dev)
        assert bool(torch.isfinite(out).all())
        return (time.perf_counter() - t0) / n * 1e3

    one_ms = timed(lambda: sw.swap(src, tgt, target_part="body"), steps)
    T11, T21 = sw.T12, sw.T21          # what swap() left behind: (1,is,is,2) each
    sel = sw.PART_IDS["body"]
    part_mask = (src['part'][:, sel].sum(1, keepdim=True) != 0).float()
    left_mask = src['part'][:, [0]].sum(1, keepdim=True).bool().float()
    tsf_img = sw.generator.transform(tgt['img'], T21) * part_mask + sw.generator.transform(src['img'], T11) * left_mask
    x8 = torch.cat([tsf_img, src['cond']], 1).expand(BATCH, -1, -1, -1).contiguous()
    T11_8, T21_8 = T11.expand(BATCH, -1, -1, -1).contiguous(), T21.expand(BATCH, -1, -1, -1).contiguous()
    eight = lambda: sw.forward(x8, tgt['feats'], T21_8, src['feats'], T11_8, src['bg'])[0]
    eight_ms = timed(eight, steps)
    roof = conv_roofline(sw.generator, lambda: [eight() for _ in range(4)], steps=4)
    sw.generator.release()
    return {"workload": "Swapper.swap, two synthetic subjects, 256x256, swap_part='body' (BASELINE config 4)",
            "one_pair": {"ms": round(one_ms, 4), "swaps_per_s": round(1e3 / one_ms, 2),
                         "what": "the whole Swapper.swap call at batch 1 (masks, T11/T21, 2 image warps, generator.swap, blend)"},
            "eight_pairs": {"ms": round(eight_ms, 4), "swaps_per_s": round(BATCH * 1e3 / eight_ms, 2),
                            "what": "Swapper.forward on 8 prepared inputs: one launch sequence of the two-stream generator",
                            "roofline": roof},
            "dtype": "bf16x3" if sw.generator.precision != "fp32" else "f32"}


def _stats(ms):
    v = sorted(ms)
    return {"median": round(v[len(v) // 2], 4), "min": round(v[0], 4), "max": round(v[-1], 4), "calls": len(v)}


def secondary_latency(dev, frames=40):
    """BASELINE config 1's call pattern on the GPU: ONE frame per call, batch 1, `transfer_params_by_smpl` + `forward` with the
    result awaited before the next frame -- the reference's own loop (models/imitator.py:166-171), no batching across frames, no
    lanes.  `eager`: the ~70 liblwg launches of a frame issued from Python; `graph`: the same launches captured once
    (Imitator.frame_graph) and replayed as one launch.  Both precisions; and one Swapper.swap (models/swapper.py:198-239) awaited."""
    from impersonator_amd.utils import synthetic
    out = {"workload": "1 frame per call (batch 1), 256x256, device synchronised after every frame; ms per frame",
           "reference_loop": "models/imitator.py:166-171"}
    for precision in ("bf16x3", "fp32"):
        imitator, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=1, seed=0, image_size=IMAGE_SIZE)
        imitator.generator.precision = precision
        imitator.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
        smpls = torch.from_numpy(demo.synthetic_smpls(frames + 8, seed=0)).to(dev)

        def eager(t):
            x = imitator.transfer_params_by_smpl(smpls[t:t + 1], "smooth", t=t)
            return imitator.forward(x, imitator.tsf_info["T"])

        def timed(fn):
            for t in range(4):
                last = fn(t)
            torch.cuda.synchronize(dev)
            ms = []
            for t in range(4, 4 + frames):
                t0 = time.perf_counter()
                last = fn(t)
                torch.cuda.synchronize(dev)
                ms.append((time.perf_counter() - t0) * 1e3)
            assert bool(torch.isfinite(last).all())
            return _stats(ms), last.clone()

        e_ms, e_last = timed(eager)
        run = imitator.frame_graph(batch=1)
        g_ms, g_last = timed(lambda t: run(smpls[t:t + 1], t=t))
        out["f32" if precision == "fp32" else "bf16x3"] = {
            "eager_ms": e_ms, "graph_ms": g_ms, "frames_per_s_graph": round(1e3 / g_ms["median"], 1),
            "graph_equals_eager": bool(torch.equal(e_last, g_last))}
        del run
        imitator.generator.release()
    sw, smpl_a, img_a, bg_a = demo.build_synthetic_imitator(batch_size=1, seed=0, image_size=IMAGE_SIZE, model="swapper")
    smpl_b = demo.synthetic_smpls(8, seed=3)[5]
    img_b = synthetic.smooth_image(77, (1, 3, IMAGE_SIZE, IMAGE_SIZE))[0]
    sw.swap_setup(img_a, img_b, src_smpl=smpl_a, tgt_smpl=smpl_b, src_bg=bg_a, tgt_bg=synthetic.smooth_image(78, (1, 3, IMAGE_SIZE, IMAGE_SIZE))[0])
    ms = []
    for i in range(4 + 20):
        t0 = time.perf_counter()
        p = sw.swap(sw.src_info, sw.tsf_info, target_part="body")
        torch.cuda.synchronize(dev)
        if i >= 4:
            ms.append((time.perf_counter() - t0) * 1e3)
    out["swap_one_pair_ms"] = dict(_stats(ms), what="Swapper.swap at batch 1 awaited, eager launches (liblwg kernels only)")
    run = sw.swap_graph(sw.src_info, sw.tsf_info, target_part="body")
    gms = []
    for i in range(4 + 20):
        t0 = time.perf_counter()
        q = run()
        torch.cuda.synchronize(dev)
        if i >= 4:
            gms.append((time.perf_counter() - t0) * 1e3)
    out["swap_one_pair_graph_ms"] = dict(_stats(gms), what="the same swap captured once (Swapper.swap_graph) and replayed as one HIP graph",
                                         graph_equals_eager=bool(torch.equal(p, q)))
    del run
    sw.generator.release()
    return out


def secondary_personalize(dev, reps=5):
    """Imitator.personalize (models/imitator.py:82-155), the once-per-source step north_star names: render + masks + background
    model + source-stream encoder.  `ORIGINAL` = the generator's own BGNet (--bg_model ORIGINAL), `deepfillv2` = InpaintSANet
    (networks/inpaintor.py:178-202: 28 gated convs + 4096-token self-attention).  ms per source on the GPU (median of `reps` awaited
    calls) and the CPU oracle (oracle/torch_ref.py imitator_personalize, all host cores, one call after a warm-up) beside it."""
    from oracle import torch_ref
    from impersonator_amd.networks.inpaintor import InpaintSANet
    from impersonator_amd.utils import synthetic
    out = {"workload": "Imitator.personalize of one 256x256 source (synthetic SMPL + random-init networks), awaited"}
    for variant in ("ORIGINAL", "deepfillv2"):
        imitator, src_smpl, src_img, _ = demo.build_synthetic_imitator(batch_size=1, seed=0, image_size=IMAGE_SIZE)
        bg_sd = None
        if variant == "deepfillv2":
            net = InpaintSANet(c_dim=4).eval()
            shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
            net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.random_inpaintor_state_dict(shapes, 1).items()})
            imitator.bgnet = net.cuda()
            bg_sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        ms = []
        for i in range(2 + reps):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            imitator.personalize(src_img, src_smpl=src_smpl)
            torch.cuda.synchronize(dev)
            if i >= 2:
                ms.append((time.perf_counter() - t0) * 1e3)
        si = imitator.src_info
        sd = {k: v.detach().cpu() for k, v in imitator.generator.state_dict().items()}
        faces_t, map_fn = imitator.render.faces.cpu(), imitator.render.map_fn.cpu()
        info = {k: si[k].cpu() for k in ("cam", "verts", "shape")}
        img_t = torch.from_numpy(src_img)[None]
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        with torch.no_grad():
            torch_ref.imitator_personalize(sd, img_t, info, faces_t, map_fn, bg_sd=bg_sd)
            t0 = time.perf_counter()
            ref = torch_ref.imitator_personalize(sd, img_t, info, faces_t, map_fn, bg_sd=bg_sd)
            cpu_ms = (time.perf_counter() - t0) * 1e3
        bg_err = float((ref["bg"] - si["bg"].cpu()).abs().max())
        feat_err = max(float((a - b.cpu()).abs().max()) for a, b in zip(list(ref["enc"]) + list(ref["res"]),
                                                                       list(si["feats"][0]) + list(si["feats"][1])))
        extra = {}
        if variant == "deepfillv2":
            # the background model alone: InpaintSANet.forward (networks/inpaintor.py:178-202), awaited, GPU and CPU oracle
            img_d = si["img"]
            body = 1 - torch_ref.morph(si["cond"][:, -1:].cpu(), imitator._opt.bg_ks, "erode")
            body_d = body.to(dev)
            fw = []
            for i in range(3 + reps):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                imitator.bgnet(img_d, masks=body_d, only_x=True)
                torch.cuda.synchronize(dev)
                if i >= 3:
                    fw.append((time.perf_counter() - t0) * 1e3)
            with torch.no_grad():
                t0 = time.perf_counter()
                torch_ref.inpaint_forward(bg_sd, img_t, body)
                inp_cpu = (time.perf_counter() - t0) * 1e3
            extra = {"inpaintor_forward": {"gpu_ms": _stats(fw), "cpu_oracle_ms": round(inp_cpu, 1), "precision": imitator.bgnet.precision,
                                           "what": "35 gated convs (31 on the bf16x3 kernels), 4096-token self-attention on the fp32 "
                                                   "matrix cores, mask compositing; wall time of the awaited call (31 + 37 launches)"}}
        out[variant] = {"gpu_ms": _stats(ms), "cpu_oracle_ms": round(cpu_ms, 1), "cpu_threads": torch.get_num_threads(), **extra,
                        "speedup": round(cpu_ms / _stats(ms)["median"], 1),
                        "parity": {"fim_equal": bool(torch.equal(ref["fim"], si["fim"].cpu())), "bg_linf": round(bg_err, 7),
                                   "src_feature_linf": round(feat_err, 7)}}
        imitator.generator.release()
        if variant == "deepfillv2":
            imitator.bgnet.release() if hasattr(imitator.bgnet, "release") else None
    out["kernel_table"] = "profiles/r05_personalize_kernel_stats.md (rocprofv3 --kernel-trace --stats -- python tools/personalize_once.py)"
    return out


def secondary_train(steps=3):
    """BASELINE config 5 per GPU: one training iteration (generator fwd/bwd + PatchGAN discriminator update,
    impersonator_trainer.py:350-366) -- tools/bench_train.py's measurement."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_train
    out = {}
    for name, (n, s) in (("512x512_batch1", (1, 512)), ("256x256_batch4", (4, 256)), ("512x512_batch4", (4, 512))):
        out[name] = {}
        # eager: ~1500 launches per iteration from Python; graph: Impersonator.optimize_parameters_graphed, the same iteration
        # captured once and replayed (what a training loop on one GPU would call)
        for mode in ("eager", "graph"):
            r = bench_train.measure(n, s, steps=steps, warmup=2, precision="bf16x3", graph=mode == "graph")
            out[name][mode] = {"ms_per_iteration": r["ms_per_iteration"], "images_per_s": r["images_per_s"], "conv_tflops": r["conv_tflops"],
                               "roofline": r["roofline"]}
            torch.cuda.empty_cache()
        out[name]["conv_gflop_per_iteration"] = r["conv_gflop_per_iteration"]
    out["what"] = ("G update (three streams forward, losses adv + L1 + mask, hand-written backward, Adam) + D update; "
                   "bf16x3 convolutions of generator and discriminator, fp32 elsewhere; 1 GPU, no all-reduce")
    return out


def main():
    args = parse()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)          # does not return: re-executes under torch.distributed.run, or exits 1
    if int(os.environ.get("WORLD_SIZE", 1)) != args.gpus:
        # never a line whose n_gpus differs from what was asked for
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: launch with --nproc-per-node %d (or bare, bench.py starts the "
                         "ranks itself)" % (args.gpus, os.environ.get("WORLD_SIZE"), args.gpus))
    rank, local_rank, world = sharding.init_process_group()
    if local_rank >= torch.cuda.device_count() and os.environ.get("LWG_DIST_BACKEND") == "gloo":
        local_rank %= torch.cuda.device_count()   # test hook: N ranks sharing the visible GPU(s), see sharding.init_process_group
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    imitator, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=BATCH, seed=0, image_size=IMAGE_SIZE)
    if args.precision:
        imitator.generator.precision = args.precision
    precision = imitator.generator.precision
    imitator.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
    smpls = torch.from_numpy(demo.synthetic_smpls(args.frames, seed=0)).to(dev)
    imitator.first_cam = smpls[0:1, 0:3].clone()
    blocks = sharding.shard_blocks(args.frames, BATCH, rank, world)
    # this rank's frames packed once, before anything is timed: consecutive chunks are adjacent rows of one tensor
    my_rows, bounds = sharding.local_rows(smpls, blocks)

    lanes = args.lanes if args.lanes is not None else imitator.lanes

    def run_steps(first, n, lanes=lanes):
        """n steps through Imitator.predict_batches (what Imitator.inference runs): the geometry of step i+1 is
        enqueued on a side stream, the generators of consecutive steps on `lanes` engines with a stream each; every
        step's work is inside the loop."""
        out = None
        idx = [(first + i) % len(blocks) for i in range(n)]
        chunks = ((my_rows[bounds[k][0]:bounds[k][1]], blocks[k][0]) for k in idx)
        for _, out in imitator.predict_batches(chunks, "smooth", lanes=lanes):
            pass
        return out

    # clock settle (untimed, before the warm-up): a cold GPU ramps its clocks over the first ~100 ms of load
    # (whole pipeline rounds, not single steps: the 32-frame launch sequences, their scratch and their kernel variants are in use --
    # and the chip at its steady-state power -- before anything is timed)
    ts = time.perf_counter()
    while (time.perf_counter() - ts) * 1e3 < args.settle_ms:
        run_steps(0, 8)
        torch.cuda.synchronize(dev)
    run_steps(0, args.warmup)
    sharding.barrier(dev)
    # host cost of enqueueing a step, measured on a short burst into empty queues (over hundreds of steps the host
    # simply blocks on the full launch queue, which says nothing)
    th = time.perf_counter()
    run_steps(0, 4)
    host_dt = (time.perf_counter() - th) / 4 * args.steps
    sharding.barrier(dev)
    rdev = dev if torch.distributed.is_initialized() else "cpu"   # RCCL reduces device tensors

    def timed_window(first, everyone=True):
        """EXACTLY args.steps steps between a barrier + device synchronisation on both sides -> (seconds until every rank was done
        = MAX over ranks, seconds until THIS rank's device was done)."""
        if everyone:
            sharding.barrier(dev)
        else:
            torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        o = run_steps(first, args.steps)
        torch.cuda.synchronize(dev)
        mine = time.perf_counter() - t0
        if not everyone:
            return mine, mine, o
        sharding.barrier(dev)
        return sharding.max_over_ranks(time.perf_counter() - t0, rdev), mine, o

    single_rank_fps = None
    if world > 1:
        # rank 0 runs one window ALONE (the others wait at the barrier below): the single-GPU rate of this very run, the
        # denominator of `linear_frac`
        if rank == 0:
            solo, _, _ = timed_window(args.warmup, everyone=False)
            single_rank_fps = BATCH * args.steps / solo
        sharding.barrier(dev)
    sampler = GpuSampler(local_rank)
    sampler.start()
    windows, mine_all, out = [], [], None
    for r in range(max(1, args.repeats)):
        w, mine, out = timed_window(args.warmup + r * args.steps)
        windows.append(w)
        mine_all.append(mine)
    sampler.stop()
    order = sorted(range(len(windows)), key=lambda i: windows[i])
    med = order[len(order) // 2]
    dt = windows[med]                       # the median window: what `value` and `ms_per_step` report
    assert bool(torch.isfinite(out).all())
    ranks_block = None
    if world > 1:
        # every rank's own completion time of the median window, its device, its clocks: a straggler GPU is visible here
        mine_fps = BATCH * args.steps / mine_all[med]
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, {"rank": rank, "fps": round(mine_fps, 2), "device": device_identity(dev),
                                                        "clocks": sampler.summary()})
        if rank == 0:
            fps = sorted(g["fps"] for g in gathered)
            ids = [(g["device"].get("uuid"), g["device"].get("pci_bus_id"), g["device"].get("pci_device_id"), g["device"]["index"])
                   for g in gathered]
            ranks_block = {"per_rank": gathered, "fps_min": fps[0], "fps_median": fps[len(fps) // 2], "fps_max": fps[-1],
                           "distinct_devices": len(set(ids)),
                           "single_rank_fps": round(single_rank_fps, 3),
                           "linear_frac": round(world * BATCH * args.steps / dt / (world * single_rank_fps), 4),
                           "note": "per-rank fps: the median window timed to each rank's own device synchronisation; "
                                   "single_rank_fps: rank 0 running one window alone in this run; linear_frac = value / (N x that)"}

    def roofline_pass():
        """K steps again with HIP events around every launch of the conv kernels (recorded by liblwg on the launch
        stream), on ONE lane: with two lanes the kernels of two batches share the chip and a launch's elapsed time is
        no longer that kernel's own (the timed region above is what gains from the overlap, not the kernel).  Returns
        the block for the conv arithmetic currently selected."""
        imitator.generator.profile(True)
        run_steps(args.warmup, args.steps, lanes=1)
        n, ms, flops = imitator.generator.profile_read()
        table = imitator.generator.profile_table()
        imitator.generator.profile(False)
        # the conv runs as a few instantiations of one implicit-GEMM kernel; `roofline` is the one with the most time
        # (names are the ones rocprofv3 --stats prints, so profiles/ can be checked against this line)
        name, (kn, kms, kfl) = max(table.items(), key=lambda kv: kv[1][1])
        achieved = kfl / (kms * 1e-3) / 1e12
        x3 = "bf16x3" in name
        peak = BF16_MFMA_PEAK_TFLOPS if x3 else FP32_MFMA_PEAK_TFLOPS
        ideal_ms = sum(v[2] / (kernel_peak(k) * 1e12) * 1e3 for k, v in table.items())
        block = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 3), "peak": round(peak, 1),
                 "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                 "frac_algorithmic": round(achieved / peak, 4),
                 "frac_pipe": round(achieved * (3.0 if x3 else 1.0) / peak, 4),
                 "executed_tflops": round(achieved * (3.0 if x3 else 1.0), 2),
                 "traffic": None, "launches": kn, "avg_launch_ms": round(kms / max(kn, 1), 5),
                 "flop_per_launch": kfl / max(kn, 1),
                 "measured": "HIP events around each launch, steps re-run on one lane (kernels not overlapped).  An event pair spans "
                             "the launch's dispatch latency too (~10 us on a dependent chain): rocprofv3's kernel durations "
                             "(profiles/r05_kernel_stats.md) are that much shorter, the fractions here that much lower",
                 "frac_note": ("achieved = ALGORITHMIC flops (2*M*Cout*taps*Cin, real taps and channels) / launch time; "
                               "frac = frac_algorithmic = achieved / peak of the MFMA instruction used (bf16 dense 2500, "
                               "fp32 157.3).  A bf16x3 kernel executes 3 bf16 MFMA products per algorithmic multiply-add: "
                               "frac_pipe = 3 * achieved / 2500 is the matrix-pipe utilisation (what the PMC pass in "
                               "profiles/ measures as MFMA busy)"),
                 "all_conv_kernels": {"achieved": round(flops / (ms * 1e-3) / 1e12, 3),
                                      "frac_pipe": round(ideal_ms / ms, 4),
                                      "launches": n, "ms_per_step": round(ms / args.steps, 4),
                                      "by_kernel": {k: {"launches": v[0], "avg_launch_ms": round(v[1] / v[0], 5),
                                                        "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 2),
                                                        "frac_pipe": round(v[2] / (v[1] * 1e-3) / 1e12 / kernel_peak(k), 4)}
                                                    for k, v in table.items()}}}
        # bytes per launch of the dominant kernel from the committed PMC passes of the same command
        # (tools/r05_profile.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected as MI355X_MICROARCH.md
        # prescribes); bench.py cannot collect counters itself.  The file carries the digest of the kernel sources it
        # was measured on: a stale measurement is dropped, not attached.
        tpath = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            stamp = tj.get("_stamp", {})
            # rocprofv3 prints every template argument; a variant of the library's table may cover several instantiations
            # (tile heights of the halo kernel): launch-weighted mean over the matching rows
            rows = [v for k, v in tj.items() if k == name or k.startswith(name.rstrip('>') + ',')]
            rows = [v for v in rows if "fetch_bytes_per_launch" in v and "write_bytes_per_launch" in v and v.get("launches")]
            t = None
            if rows:
                nl = sum(v["launches"] for v in rows)
                t = {"fetch_bytes_per_launch": sum(v["launches"] * v["fetch_bytes_per_launch"] for v in rows) / nl,
                     "write_bytes_per_launch": sum(v["launches"] * v["write_bytes_per_launch"] for v in rows) / nl}
            if stamp.get("csrc_sha256") != csrc_digest():
                block["traffic_note"] = ("profiles/%s was measured on other kernel sources (stamp %s..., now %s...): "
                                         "not attached" % (TRAFFIC_FILE, str(stamp.get("csrc_sha256"))[:10], csrc_digest()[:10]))
            elif t and "fetch_bytes_per_launch" in t and "write_bytes_per_launch" in t:
                block["traffic"] = round(t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"])
                block["traffic_note"] = ("fabric-side bytes per launch (profiles/%s, commit %s): %.0f MB read + %.0f MB "
                                         "written; Infinity-Cache hits included" % (TRAFFIC_FILE, stamp.get("commit", "?"),
                                                          t["fetch_bytes_per_launch"] / 1e6, t["write_bytes_per_launch"] / 1e6))
        return block

    roofline = roofline_pass() if not args.no_roofline else None
    # what carried the barrier / max-over-ranks above (every rank takes part in its one-element all-reduce)
    rccl = sharding.collective_info(dev) if torch.distributed.is_initialized() else None

    fp32_mode = None
    if precision != "fp32" and not args.no_fp32_mode:
        # the same steps with the convolutions on the exact-fp32 MFMA path, for the record
        imitator.generator.precision = "fp32"
        run_steps(0, args.warmup)
        w32 = sorted(timed_window(args.warmup + r * args.steps)[0] for r in range(max(1, min(3, args.repeats))))
        dt32 = w32[len(w32) // 2]
        fp32_mode = {"value": round(world * BATCH * args.steps / dt32, 3), "unit": "frames/s",
                     "ms_per_step": round(dt32 / args.steps * 1e3, 4), "repeats": len(w32),
                     "ms_per_step_min": round(w32[0] / args.steps * 1e3, 4), "ms_per_step_max": round(w32[-1] / args.steps * 1e3, 4),
                     "dtype": "f32",
                     "note": "same workload, precision='fp32' (v_mfma_f32_32x32x2_f32, bit-exact fmaf chains)"}
        if not args.no_roofline:
            fp32_mode["roofline"] = roofline_pass()
        imitator.generator.precision = precision

    if rank == 0:
        frames = world * BATCH * args.steps
        line = {
            "metric": "frames/sec (256x256 motion-imitation, batch=8)",
            "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "repeats": len(windows), "ms_per_step_min": round(min(windows) / args.steps * 1e3, 4),
            "ms_per_step_max": round(max(windows) / args.steps * 1e3, 4),
            "ms_per_step_windows": [round(w / args.steps * 1e3, 4) for w in windows],
            "timing": "median of %d windows of %d steps, each bracketed by barrier + device synchronisation, MAX over ranks" % (len(windows), args.steps),
            "gpu_clocks": sampler.summary(),
            "host_enqueue_ms_per_step": round(host_dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if precision == "fp32" else "bf16x3", "data": "synthetic",
            "config": {"workload": "Imitator inference 256x256 batch=8, random-init ImpersonatorGenerator (tsf ResUnet, "
                                   "105.58 GFLOP/frame) + synthetic SMPL (6890 verts / 13776 faces), 1 source, "
                                   "%d-frame synthetic reference sequence" % args.frames,
                       "batch_per_gpu": BATCH, "image_size": IMAGE_SIZE, "parallelism": "frame-sharded replicas x%d" % world,
                       "grid_sample_align_corners": False,
                       "streams": "%d generator lane(s) + 1 geometry stream per GPU; %s consecutive batches of %d per generator "
                                  "launch sequence (frames of one source are independent: bit-identical to one batch at a time)"
                                  % (lanes, os.environ.get("LWG_FUSE", str(imitator.fuse)), BATCH),
                       "precision": precision + (" (fp32 operands carried as 2 bf16 terms, 3 MFMA products, fp32 "
                                                 "accumulate; 8e-5 L-inf on the image vs fp32, bound 1e-3)"
                                                 if precision == "bf16x3" else " (exact fp32 MFMA)")},
        }
        if rccl is not None:
            # N > 1 (or LWG_FORCE_DIST=1 at N = 1): the process group behind the timing barrier -- backend 'nccl' IS RCCL on
            # ROCm; `ranks` is the communicator's size as the library reports it, `allreduce_of_ones` must equal it
            line["rccl"] = rccl
            if rccl["allreduce_of_ones"] != world or rccl["ranks"] != args.gpus:
                line["invalid"] = ("the process group spans %r ranks and its all-reduce of ones returned %r; --gpus %d"
                                   % (rccl["ranks"], rccl["allreduce_of_ones"], args.gpus))
        if ranks_block is not None:
            line["ranks"] = ranks_block   # (distinct_devices is reported, not enforced: device identifiers are the runtime's to define)
        if fp32_mode is not None:
            line["exact_fp32_mode"] = fp32_mode
        if roofline is not None:
            line["roofline"] = roofline
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], kept = cpu_baseline()
            # self-check of the timed pipeline against the oracle, after and outside the timed region
            line["parity"] = parity_block(imitator, src_img, bg_img, smpls, lanes, kept)
            if not line["parity"]["ok"]:
                line["invalid"] = "the timed pipeline's output failed the parity check against the oracle (see `parity`)"
        if world == 1 and not args.no_secondary:
            # other workloads of BASELINE.json, measured after (and outside) the timed region
            line["secondary"] = {"swap": secondary_swap(dev), "latency": secondary_latency(dev),
                                 "personalize": secondary_personalize(dev), "train": secondary_train()}
        print(json.dumps(line))
        if line.get("invalid"):
            sys.exit(1)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
