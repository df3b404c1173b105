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
TRAFFIC_FILE = "r06_traffic.json"
# speed of the CPU port (oracle/) relative to the reference's own modules on the same cores, same inputs, bit-identical outputs:
# tools/port_vs_reference.py, measured where /root/reference exists (profiles/r06_port_vs_reference.md)
PORT_VS_REFERENCE = 0.97


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
    p.add_argument("--precision", choices=["auto", "bf16x3", "fp32"], default=None,
                   help="conv arithmetic of the per-frame stream (default: the library default, auto = bf16x3 unless the probe at "
                        "personalize sends the weights to fp32)")
    p.add_argument("--no-fp32-mode", action="store_true", help="skip the extra timed pass in exact-fp32 mode")
    p.add_argument("--no-strict", action="store_true", help="skip the strict one-batch-per-launch-sequence windows")
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
            "sample": "%d batches of %d frames 256x256 after warm-up; best of 16/32/64 threads, %d logical cores" % (batches, BATCH, os.cpu_count()),
            "port_vs_reference": PORT_VS_REFERENCE, "torch": torch.__version__}
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
    # flat numbers only (what every field means: DESIGN.md section 5.2).  `linf` / `fim_mismatch` / `T_linf`: oracle restarted from the
    # device's posed vertices.  `theta_*`: the oracle's own SMPL from the same theta (fp64, rounded); `theta_linf_all` is over ALL
    # pixels of ALL frames (no exclusions), `theta_fim_mismatch_pixels` counts face-index pixels that differ.
    return {"ok": bool(linf <= 1e-3 and fim_mismatch == 0 and chain_ok), "bound": 1e-3, "frames": int(pred.shape[0]), "lanes": lanes,
            "linf": round(linf, 7), "fim_mismatch": fim_mismatch, "T_linf": round(float((torch.cat(Ts) - fr["T"]).abs().max()), 9),
            "theta_ok": bool(chain_ok), "theta_smpl_precision": smpl_mode, "theta_fim_mismatch_pixels": chain_fim,
            "theta_frames_with_identical_fim": int(frames_agree.sum()), "theta_linf_all": round(float(d_chain.max()), 7)}


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
    return {"kernel": name, "achieved": round(achieved, 1), "peak": round(peak, 1), "frac": round(achieved / peak, 4),
            "frac_pipe": round(achieved * (3.0 if x3 else 1.0) / peak, 4), "all_conv_frac_pipe": round(ideal_ms / ms, 4),
            "all_conv_ms": round(ms / steps, 4)}


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
    # one_pair: the whole Swapper.swap call at batch 1; eight_pairs: Swapper.forward on 8 prepared inputs as one launch sequence
    return {"one_pair_ms": round(one_ms, 4), "swaps_per_s": round(1e3 / one_ms, 1), "eight_pairs_ms": round(eight_ms, 4),
            "eight_pairs_swaps_per_s": round(BATCH * 1e3 / eight_ms, 1), "eight_pairs_roofline": roof,
            "dtype": "bf16x3" if sw.generator.precision != "fp32" else "f32"}


def _stats(ms):
    v = sorted(ms)
    return {"median": round(v[len(v) // 2], 3), "min": round(v[0], 3), "max": round(v[-1], 3)}


def secondary_latency(dev, frames=40):
    """BASELINE config 1's call pattern on the GPU: ONE frame per call, batch 1, `transfer_params_by_smpl` + `forward` with the
    result awaited before the next frame -- the reference's own loop (models/imitator.py:166-171), no batching across frames, no
    lanes.  `eager`: the ~70 liblwg launches of a frame issued from Python; `graph`: the same launches captured once
    (Imitator.frame_graph) and replayed as one launch.  Both precisions; and one Swapper.swap (models/swapper.py:198-239) awaited."""
    from impersonator_amd.utils import synthetic
    out = {}
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
            "eager_ms": e_ms["median"], "graph_ms": g_ms["median"], "graph_ms_min": g_ms["min"], "graph_ms_max": g_ms["max"],
            "frames_per_s_graph": round(1e3 / g_ms["median"], 1), "graph_equals_eager": bool(torch.equal(e_last, g_last))}
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
    out["swap_one_pair_ms"] = _stats(ms)["median"]
    run = sw.swap_graph(sw.src_info, sw.tsf_info, target_part="body")
    gms = []
    for i in range(4 + 20):
        t0 = time.perf_counter()
        q = run()
        torch.cuda.synchronize(dev)
        if i >= 4:
            gms.append((time.perf_counter() - t0) * 1e3)
    out["swap_one_pair_graph_ms"] = _stats(gms)["median"]
    out["swap_graph_equals_eager"] = bool(torch.equal(p, q))
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
    out = {}
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
            extra = {"inpaintor_forward_ms": _stats(fw)["median"], "inpaintor_cpu_oracle_ms": round(inp_cpu, 1),
                     "inpaintor_precision": imitator.bgnet.precision}
        out[variant] = {"gpu_ms": _stats(ms)["median"], "gpu_ms_min": _stats(ms)["min"], "cpu_oracle_ms": round(cpu_ms, 1),
                        "cpu_threads": torch.get_num_threads(), **extra,
                        "fim_equal": bool(torch.equal(ref["fim"], si["fim"].cpu())), "bg_linf": round(bg_err, 7),
                        "src_feature_linf": round(feat_err, 7)}
        imitator.generator.release()
        if variant == "deepfillv2":
            imitator.bgnet.release() if hasattr(imitator.bgnet, "release") else None
    return out


def secondary_train(steps=3):
    """BASELINE config 5 per GPU: one training iteration (generator fwd/bwd + PatchGAN discriminator update,
    impersonator_trainer.py:350-366) -- tools/bench_train.py's measurement."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_train
    out = {}
    for name, (n, s) in (("512x512_batch1", (1, 512)), ("256x256_batch4", (4, 256)), ("512x512_batch4", (4, 512))):
        # eager: ~1500 launches per iteration from Python; graph: Impersonator.optimize_parameters_graphed, the same iteration
        # captured once and replayed (what a training loop on one GPU would call).  frac = algorithmic conv FLOP of the WHOLE
        # iteration / its time, against the bf16x3 ceiling 2500 / 3 TFLOP/s
        e = bench_train.measure(n, s, steps=steps, warmup=2, precision="bf16x3", graph=False)
        torch.cuda.empty_cache()
        g = bench_train.measure(n, s, steps=steps, warmup=2, precision="bf16x3", graph=True)
        torch.cuda.empty_cache()
        out[name] = {"eager_ms": e["ms_per_iteration"], "graph_ms": g["ms_per_iteration"], "images_per_s": g["images_per_s"],
                     "conv_tflops": g["conv_tflops"], "frac": g["roofline"]["frac"], "conv_gflop": g["conv_gflop_per_iteration"]}
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
    imitator.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
    policy, auto_report = imitator.generator.precision_policy, imitator.generator.auto_report
    precision = imitator.generator.precision          # what the pass runs in (policy auto: what the probe chose)
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
        # flat numbers (definitions: DESIGN.md section 5.1).  achieved = ALGORITHMIC flops / launch time of the dominant kernel;
        # frac = achieved / dense peak of the MFMA instruction used; frac_pipe = executed products / peak (bf16x3: 3 per multiply-add)
        block = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 3), "peak": round(peak, 1),
                 "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                 "frac_pipe": round(achieved * (3.0 if x3 else 1.0) / peak, 4),
                 "traffic": None, "launches": kn, "avg_launch_ms": round(kms / max(kn, 1), 5),
                 "flop_per_launch": round(kfl / max(kn, 1)), "timer": "HIP events on the launch stream, one lane",
                 "all_conv_tflops": round(flops / (ms * 1e-3) / 1e12, 2), "all_conv_frac_pipe": round(ideal_ms / ms, 4),
                 "all_conv_launches": n, "all_conv_ms_per_step": round(ms / args.steps, 4),
                 "by_kernel": {k: [v[0], round(v[1] / v[0] * 1e3, 1), round(v[2] / (v[1] * 1e-3) / 1e12 / kernel_peak(k), 3)]
                               for k, v in table.items()}}   # kernel -> [launches, avg us, pipe fraction]
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
                block["traffic_note"] = "profiles/%s is stamped with other kernel sources: not attached" % TRAFFIC_FILE
            elif t:
                block["traffic"] = round(t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"])
                block["traffic_read"] = round(t["fetch_bytes_per_launch"])
                block["traffic_written"] = round(t["write_bytes_per_launch"])
                block["traffic_note"] = "fabric-side bytes per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, profiles/%s" % TRAFFIC_FILE
        return block

    roofline = roofline_pass() if not args.no_roofline else None
    # what carried the barrier / max-over-ranks above (every rank takes part in its one-element all-reduce)
    rccl = sharding.collective_info(dev) if torch.distributed.is_initialized() else None

    def median_window(n_windows):
        w = sorted(timed_window(args.warmup + r * args.steps)[0] for r in range(n_windows))
        return w[len(w) // 2], w

    fp32_mode = None
    if precision != "fp32" and not args.no_fp32_mode:
        # the same steps with the convolutions on the exact-fp32 MFMA path: the reference's own arithmetic (like for like)
        imitator.generator.precision = "fp32"
        run_steps(0, args.warmup)
        dt32, w32 = median_window(max(1, min(3, args.repeats)))
        fp32_mode = {"value": round(world * BATCH * args.steps / dt32, 3), "unit": "frames/s",
                     "ms_per_step": round(dt32 / args.steps * 1e3, 4), "repeats": len(w32),
                     "ms_per_step_min": round(w32[0] / args.steps * 1e3, 4), "ms_per_step_max": round(w32[-1] / args.steps * 1e3, 4),
                     "dtype": "f32"}
        if not args.no_roofline:
            fp32_mode["roofline"] = roofline_pass()
        imitator.generator.precision = precision

    # the reference's call pattern strictly: ONE batch of 8 per generator launch sequence (models/imitator.py:166-171 runs one
    # forward per batch) -- with the two lanes, and on a single stream
    strict = {}
    if not args.no_strict:
        keep = os.environ.get("LWG_FUSE")
        os.environ["LWG_FUSE"] = "1"
        try:
            for key, nl in (("strict_batch8_fps", lanes), ("strict_batch8_one_lane_fps", 1)):
                run_steps(0, args.warmup, lanes=nl)
                w = []
                for r in range(max(1, min(3, args.repeats))):
                    sharding.barrier(dev)
                    t0 = time.perf_counter()
                    run_steps(args.warmup + r * args.steps, args.steps, lanes=nl)
                    torch.cuda.synchronize(dev)
                    sharding.barrier(dev)
                    w.append(sharding.max_over_ranks(time.perf_counter() - t0, rdev))
                strict[key] = round(world * BATCH * args.steps / sorted(w)[len(w) // 2], 1)
        finally:
            if keep is None:
                os.environ.pop("LWG_FUSE", None)
            else:
                os.environ["LWG_FUSE"] = keep

    if rank == 0:
        frames = world * BATCH * args.steps
        fuse = int(os.environ.get("LWG_FUSE", imitator.fuse))
        line = {
            "metric": "frames/sec (256x256 motion-imitation, batch=8)",
            "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if precision == "fp32" else "bf16x3", "data": "synthetic",
            "config": {"workload": "Imitator inference 256x256 batch=8, random-init ImpersonatorGenerator + synthetic SMPL, 1 source",
                       "gflop_per_frame": 105.58, "frames": args.frames, "batch_per_gpu": BATCH, "image_size": IMAGE_SIZE,
                       "parallelism": "frame-sharded replicas x%d" % world, "grid_sample_align_corners": False,
                       "lanes": lanes, "batches_per_launch_sequence": fuse, "frames_per_launch": fuse * BATCH,
                       "precision": precision, "precision_policy": policy,
                       "auto_probe_linf_vs_fp32": None if not auto_report else round(auto_report["linf_bf16x3_vs_fp32"], 7), **strict},
            "repeats": len(windows), "ms_per_step_min": round(min(windows) / args.steps * 1e3, 4),
            "ms_per_step_max": round(max(windows) / args.steps * 1e3, 4),
            "ms_per_step_windows": [round(w / args.steps * 1e3, 4) for w in windows],
            "gpu_clocks": sampler.summary(),
            "host_enqueue_ms_per_step": round(host_dt / args.steps * 1e3, 4),
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
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], kept = cpu_baseline()
            # self-check of the timed pipeline against the oracle, after and outside the timed region
            line["parity"] = parity_block(imitator, src_img, bg_img, smpls, lanes, kept)
            if not line["parity"]["ok"]:
                line["invalid"] = "the timed pipeline's output failed the parity check against the oracle (see `parity`)"
        if roofline is not None:
            # the like-for-like (exact fp32) result and the parity numbers as flat fields of `roofline`, next to the bf16x3 figures
            if fp32_mode is not None:
                r32 = fp32_mode.get("roofline") or {}
                roofline.update({"exact_fp32_fps": fp32_mode["value"], "exact_fp32_ms_per_step": fp32_mode["ms_per_step"],
                                 "exact_fp32_frac": r32.get("frac"), "exact_fp32_all_conv_frac": r32.get("all_conv_frac_pipe")})
            if "parity" in line:
                pb = line["parity"]
                roofline.update({"parity_ok": pb["ok"], "parity_linf": pb["linf"], "fim_mismatch": pb["fim_mismatch"],
                                 "theta_chain_linf_all": pb["theta_linf_all"],
                                 "theta_chain_fim_mismatch_pixels": pb["theta_fim_mismatch_pixels"]})
            line["roofline"] = roofline
        if world == 1 and not args.no_secondary:
            # other workloads of BASELINE.json, measured after (and outside) the timed region (field definitions: DESIGN.md 5.3)
            line["secondary"] = {"swap": secondary_swap(dev), "latency": secondary_latency(dev),
                                 "personalize": secondary_personalize(dev), "train": secondary_train()}
        # the numbers a truncated record must still show, once more at the very end of the line
        line["summary"] = {"fps": line["value"], "ms_per_step": line["ms_per_step"], "dtype": line["dtype"],
                           "exact_fp32_fps": fp32_mode["value"] if fp32_mode else None,
                           "frac": roofline["frac"] if roofline else None, "frac_pipe": roofline["frac_pipe"] if roofline else None,
                           "all_conv_frac_pipe": roofline["all_conv_frac_pipe"] if roofline else None,
                           "parity_linf": line.get("parity", {}).get("linf"), "fim_mismatch": line.get("parity", {}).get("fim_mismatch"),
                           "theta_chain_linf_all": line.get("parity", {}).get("theta_linf_all"),
                           "cpu_fps": line.get("cpu_baseline", {}).get("value"), **strict}
        print(json.dumps(line))
        if line.get("invalid"):
            sys.exit(1)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
