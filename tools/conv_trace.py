"""Per-wave cycle accounting of the dominant conv kernel INSIDE the timed pipeline (the stand-in for an instruction-level
trace: rocprofv3 --att has no decoder library in this image, gpurun_out/r03a/att_probe.log).

    python tools/conv_trace.py [steps=3] [out.md]

liblwg's measurement hook (include/lwg.h, lwg_conv_trace) swaps every launch of conv_igemm_bf16x3<128,...> for its
instrumented twin -- the same code plus s_memtime reads (a) on entry, (b) at the start of the main loop, (c) before and
after the stage's `s_waitcnt vmcnt(N) lgkmcnt(0)` (the data wait: DMA pieces of the next stage + own LDS reads), (d) after
the stage barrier, (e) at the end of the loop and of the kernel.  The reads of one stage are consumed behind the next
stage's own lgkmcnt(0), so the twin waits nowhere the production kernel does not.  One step of the bench workload
(batch 8, one lane so that a launch has the chip to itself, plus a two-lane run for comparison) is traced and summarised
per layer shape: where a wave's cycles go and how far the per-stage time is from the 24 MFMAs x 32 cycles a stage must
issue."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impersonator_amd import _lib, demo  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
OUT = sys.argv[2] if len(sys.argv) > 2 else None
BATCH = 8
MFMA_CYCLES_PER_STAGE = 24 * 32      # 24 v_mfma_f32_32x32x16_bf16 per wave and stage, 8 passes x 4 cycles each


def run(lanes, lines):
    lib = _lib.load()
    im, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=BATCH, seed=0)
    im.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
    smpls = torch.from_numpy(demo.synthetic_smpls(1024, seed=0)).cuda()
    im.first_cam = smpls[0:1, 0:3].clone()
    chunks = lambda n, first=0: ((smpls[(first + i) * BATCH:(first + i + 1) * BATCH], (first + i) * BATCH) for i in range(n))
    for _ in im.predict_batches(chunks(24), "smooth", lanes=lanes):     # warm-up: clocks, allocator, lanes
        pass
    torch.cuda.synchronize()
    buf = torch.zeros(64 << 20, dtype=torch.int64, device="cuda")     # 512 MB of records
    _lib.check(lib.lwg_conv_trace(_lib.ptr(buf), buf.numel() * 8))
    for _ in im.predict_batches(chunks(STEPS * lanes, 24), "smooth", lanes=lanes):
        pass
    torch.cuda.synchronize()
    launches, info = [], (ctypes.c_longlong * 10)()
    while lib.lwg_conv_trace_launch(len(launches), info) == 0:
        launches.append(list(info))
    _lib.check(lib.lwg_conv_trace(None, 0))
    host = buf.cpu().numpy().view(np.uint64)
    by_shape = {}
    for off, gx, gy, gz, waves, stages, cin, cout, hm, n in launches:
        rec = host[off // 8: off // 8 + gx * gy * gz * waves * 8].reshape(gz, gy, gx, waves, 8).astype(np.int64)
        if not rec[..., 3].any():
            continue   # launch never ran (the buffer was filled behind it)
        by_shape.setdefault((cin, cout, hm, n, gx, gy, gz, stages), []).append(rec)
    lines.append("### %d lane%s: %d traced launches over %d step(s)\n" % (lanes, "" if lanes == 1 else "s", len(launches), STEPS))
    lines.append("| layer (Cin->Cout @ Hm, grid) | launches | prologue (cycles) | main loop | epilogue | cycles per stage (min %d) | "
                 "data wait | barrier wait | cycles per DMA issue | shader clock (GHz) | wave lifetime (us) |" % MFMA_CYCLES_PER_STAGE)
    lines.append("|---|---|---|---|---|---|---|---|---|---|---|")
    for key in sorted(by_shape, key=lambda k: -len(by_shape[k]) * k[4] * k[5] * k[6]):
        cin, cout, hm, n, gx, gy, gz, stages = key
        recs = by_shape[key]
        pro, loop, epi, per_stage, wait, bar, dma, ghz, us = [], [], [], [], [], [], [], [], []
        for r in recs:
            t0, l0, l1, t1, w, b, packed = (r[..., i] for i in range(7))
            st, dm, rt = packed & 0xffff, (packed >> 16) & 0xfffffff, (packed >> 44) & 0xfffff
            pro.append(np.mean(l0 - t0))
            loop.append(np.mean(l1 - l0))
            epi.append(np.mean(t1 - l1))
            per_stage.append(np.mean((l1 - l0) / np.maximum(st, 1)))
            wait.append(np.mean(w / np.maximum(l1 - l0, 1)))
            bar.append(np.mean(b / np.maximum(l1 - l0, 1)))
            dma.append(np.mean(dm / np.maximum(2 * (st - 1), 1)))       # two timed pieces per stage, from the second stage on
            ghz.append(np.mean((t1 - t0) / np.maximum(rt, 1)) * 0.1)     # shader cycles per 10-ns tick
            us.append(np.mean(rt) * 0.01)
        m = lambda v: float(np.mean(v))
        kind = "convT " if stages < 0 else ""
        waves = recs[0].shape[-2]
        lines.append("| %s%d->%d @%d N=%d (%dx%dx%d x %d waves, %d stages) | %d | %.0f | %.0f | %.0f | %.0f | %.1f %% | %.1f %% | %.0f | %.2f | %.1f |" % (
            kind, cin, cout, hm, n, gx, gy, gz, waves, abs(stages), len(recs), m(pro), m(loop), m(epi), m(per_stage),
            100 * m(wait), 100 * m(bar), m(dma), m(ghz), m(us)))
    # the trunk layer in detail: distribution over workgroups and XCDs
    trunk = [k for k in by_shape if k[0] == 512 and k[1] == 512]
    if trunk:
        r = np.concatenate([x.reshape(-1, x.shape[-2], 8) for x in by_shape[trunk[0]]])      # [launch x workgroup][wave][8]
        loopc = (r[..., 2] - r[..., 1]).astype(np.float64)
        w, b = r[..., 4] / loopc, r[..., 5] / loopc
        xcc = (r[:, 0, 7] >> 32) & 0xf
        lines.append("\nTrunk layer 512->512 @32, per workgroup (all traced launches): data wait %.1f %% (p10 %.1f, p90 %.1f), barrier %.1f %% "
                     "(p10 %.1f, p90 %.1f); by XCD (data wait %%): %s\n" % (
                         100 * w.mean(), 100 * np.percentile(w, 10), 100 * np.percentile(w, 90), 100 * b.mean(),
                         100 * np.percentile(b, 10), 100 * np.percentile(b, 90),
                         ", ".join("%d: %.1f" % (x, 100 * w[xcc == x].mean()) for x in sorted(set(xcc.tolist())))))
    im.generator.release()


def main():
    lines = ["# Cycle accounting of the generator's conv kernels inside the bench pipeline (tools/conv_trace.py; LWG_HALO=%s LWG_FUSE=%s)\n" % (os.environ.get("LWG_HALO", "1"), os.environ.get("LWG_FUSE", "default")),
             "s_memtime counts shader-clock cycles.  `data wait` = cycles between the reads before and after the stage's "
             "`s_waitcnt vmcnt(N) lgkmcnt(0)`; `barrier wait` = cycles in the `s_barrier` that follows (waves of a workgroup "
             "waiting for the slowest one's data); `cycles per DMA issue` = s_memtime before to after one global_load_lds_dwordx4 "
             "(+ its s_mov m0), averaged over the first activation piece and the first weight piece of every stage (a wave issues "
             "8 per stage); the shader clock is the wave's cycle count over its 100 MHz s_memrealtime ticks.  A stage issues 24 "
             "MFMAs of 32 cycles = %d cycles per wave at least; in an eight-wave launch two waves share a SIMD's matrix pipe, so "
             "%d cycles per wave-stage is its floor.  The halo kernels (conv3x3_halo_bf16x3: every row without a wait figure, "
             "`convT` rows = its transposed-conv form) record entry / loop start / loop end / exit only: their wait columns read 0.0 = "
             "not measured.  The ring kernel's traced twin always has four waves and a 4-slot ring.\n" % (MFMA_CYCLES_PER_STAGE, 2 * MFMA_CYCLES_PER_STAGE)]
    for lanes in (1, 2):
        run(lanes, lines)
    text = "\n".join(lines) + "\n"
    print(text)
    if OUT:
        with open(OUT, "w") as fh:
            fh.write(text)


if __name__ == "__main__":
    main()
