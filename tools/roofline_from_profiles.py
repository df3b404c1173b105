#!/usr/bin/env python
"""Per-layer and per-kernel roofline figures from a rocprofv3 kernel trace alone (development aid; numpy-free, runs anywhere).

    python tools/roofline_from_profiles.py <k_kernel_trace.csv> [--out profiles/rNN_roofline.md] [--cmd "<the traced command>"]

Every dispatch of a per-frame generator pass is attributed to its layer by POSITION: a pass is the 22 convolution launches
from one stem launch to the next (networks/generator.py:277-301 in liblwg's launch order: stem, three stride-2 encoders, twelve
trunk convolutions, then transposed conv / skipper conv per decoder level); the frames per launch follow from the first
trunk launch's grid.  Algorithmic work of a launch = 2 * frames * Hout * Wout * Cout * taps * Cin with the real taps and
channels (SURVEY.md section 8a's layer table: 105.58 GFLOP per frame over the 24 convolutions, the two 7x7 heads not among
the 22).  `TFLOP/s` = that / the launch's rocprofv3 duration; `frac` = / the dense peak of the MFMA instruction the kernel
uses (bf16 2500, fp32 157.3 TFLOP/s: /opt/skills/guides/MI355X_MICROARCH.md); `pipe` = executed products / peak (a bf16x3 kernel
executes three bf16 products per algorithmic multiply-add).  The HBM-side kernels are priced with SURVEY.md section 8d's
algorithmic bytes against 8.0 TB/s (spec) and 6.29 TB/s (measured copy).  Passes of the source stream (encode_src: 16 launches,
always exact fp32) and anything that is not a 22-launch pass are listed but not attributed."""
import argparse
import collections
import csv
import re
import sys

BF16_PEAK, FP32_PEAK = 2500.0, 157.3          # TFLOP/s dense
HBM_SPEC, HBM_COPY = 8000.0, 6290.0           # GB/s

# (name, Cin, Cout, taps, output edge at image size 256, kind); a transposed conv's taps are per INPUT pixel (1+2+2+4 over its phases)
LAYERS = [("stem 7x7 6->64", 6, 64, 49, 256, "conv")] + \
         [("encoder.%d 3x3 s2 %d->%d" % (i + 1, 64 << i, 128 << i), 64 << i, 128 << i, 9, 128 >> i, "conv") for i in range(3)] + \
         [("trunk.%d 3x3 512->512" % i, 512, 512, 9, 32, "conv") for i in range(12)]
for lvl, (cin, cout, edge) in enumerate([(512, 256, 32), (256, 128, 64), (128, 64, 128)]):
    LAYERS.append(("convT.%d 3x3 s2 %d->%d" % (lvl, cin, cout), cin, cout, 9, edge, "convT"))        # edge = INPUT edge
    LAYERS.append(("skipper.%d 3x3 %d->%d" % (lvl, 2 * cout, cout), 2 * cout, cout, 9, 2 * edge, "conv"))


def layer_flop(i, frames, scale=1.0):
    _, cin, cout, taps, edge, _ = LAYERS[i]
    return 2.0 * frames * (edge * scale) ** 2 * cout * taps * cin


def short(name):
    m = re.search(r"namespace\)::(\w+(<.*>)?)", name)
    s = m.group(1) if m else name
    return re.sub(r"\(.*$", "", s)[:90]


def is_conv(n):
    return n.startswith(("conv_igemm", "conv3x3_halo", "stem_bf16x3"))


def is_stem(n):
    return n.startswith("stem_bf16x3") or re.match(r"conv_igemm_f32<64, 1, 2, true", n) is not None


def tile_rows(n):
    """output rows (pixels) per workgroup in grid.x"""
    a = [x.strip() for x in re.search(r"<(.*)>", n).group(1).split(",")] if "<" in n else []
    if n.startswith("conv3x3_halo"):
        return int(a[5]) if len(a) > 5 else 128
    if n.startswith("conv_igemm_bf16x3"):
        return int(a[5]) if len(a) > 5 else 128
    return 128


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--out", default=None)
    ap.add_argument("--cmd", default=None, help="the traced command line, recorded in the header")
    ap.add_argument("--image-size", type=int, default=256)
    args = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(args.trace)):
        n = short(r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), n,
                     int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])))
    rows.sort()
    scale = args.image_size / 256.0
    convs = [r for r in rows if is_conv(r[2])]
    # passes: from one stem launch to the next
    passes, cur = [], None
    for r in convs:
        if is_stem(r[2]):
            if cur:
                passes.append(cur)
            cur = [r]
        elif cur is not None:
            cur.append(r)
    if cur:
        passes.append(cur)
    good = [p for p in passes if len(p) == len(LAYERS)]
    lines = []
    w = lines.append
    w("# Roofline figures from a rocprofv3 kernel trace (tools/roofline_from_profiles.py)\n")
    if args.cmd:
        w("Traced command: `%s`\n" % args.cmd)
    w("Trace: `%s` -- %d dispatches, %d convolution launches in %d passes, %d of them per-frame generator passes of %d launches "
      "(the others: %s).\n" % (args.trace.split("gpurun_out/")[-1], len(rows), len(convs), len(passes), len(good), len(LAYERS),
                               dict(collections.Counter(len(p) for p in passes if len(p) != len(LAYERS))) or "none"))
    if not good:
        w("no per-frame generator pass found")
        print("\n".join(lines))
        return
    per_layer = collections.OrderedDict()
    per_kernel = collections.OrderedDict()
    frames_seen = collections.Counter()
    for p in good:
        frames = p[4][3] * tile_rows(p[4][2]) / (32 * scale) ** 2       # first trunk launch: grid.x tiles of its row count
        frames_seen[frames] += 1
        for i, (_, dur, name, gx, gy, gz) in enumerate(p):
            fl = layer_flop(i, frames, scale)
            for key, tab in ((LAYERS[i][0] if not LAYERS[i][0].startswith("trunk") else "trunk.0-11 3x3 512->512 (each)", per_layer),
                             (name, per_kernel)):
                a = tab.setdefault(key, dict(n=0, ns=0, flop=0.0, names=collections.Counter()))
                a["n"] += 1
                a["ns"] += dur
                a["flop"] += fl
                a["names"][name] += 1
    w("Frames per launch (from the trunk grid): %s.\n" % ", ".join("%g x %d passes" % (k, v) for k, v in frames_seen.items()))

    def table(tab, first):
        w("| %s | launches | avg us | GFLOP per launch | TFLOP/s | frac of peak | pipe |" % first)
        w("|---|---|---|---|---|---|---|")
        tot_ns = tot_fl = tot_ideal = 0.0
        for k, a in tab.items():
            kn = a["names"].most_common(1)[0][0]
            x3 = "bf16x3" in kn
            peak = BF16_PEAK if x3 else FP32_PEAK
            tf = a["flop"] / a["ns"] / 1e3
            w("| `%s`%s | %d | %.1f | %.2f | %.1f | %.4f | %.3f |" % (
                k, "" if first == "kernel" else " -- `%s`" % kn, a["n"], a["ns"] / a["n"] / 1e3, a["flop"] / a["n"] / 1e9, tf, tf / peak,
                tf * (3 if x3 else 1) / peak))
            tot_ns += a["ns"]
            tot_fl += a["flop"]
            tot_ideal += a["flop"] / (peak / (3 if x3 else 1))
        w("| **all %d convolution launches of a pass** | %d | %.1f (sum per pass) | %.2f (per pass) | %.1f | | %.3f |\n" % (
            len(LAYERS), sum(a["n"] for a in tab.values()), tot_ns / len(good) / 1e3, tot_fl / len(good) / 1e9, tot_fl / tot_ns / 1e3,
            tot_ideal / tot_ns / 1e3))

    w("## By layer\n")
    table(per_layer, "layer -- kernel")
    w("## By kernel instantiation\n")
    table(per_kernel, "kernel")

    # ---- HBM-side kernels (SURVEY.md section 8d's algorithmic bytes), over the time window of the attributed passes
    t_lo, t_hi = good[0][0][0], good[-1][-1][0] + good[-1][-1][1]
    total_frames = sum(k * v for k, v in frames_seen.items())
    s2 = scale * scale
    # The InstanceNorm-apply passes that still EXIST as launches (the others are folded into the consumer conv's halo,
    # ConvArgs::raw_in): after the stem and the three stride-2 encoders (raw read + activation write; the encoders' also gather the
    # cached source features of their level through the Liquid Warping Block), and after the SECOND conv of each residual block
    # (raw read + residual read + write + source-feature gather).  A launch is identified by its POSITION in the pass (the conv
    # launch it follows) and priced with the bytes of that layer only: frames x per-frame bytes + the shared source features once.
    def apply_bytes(after_conv, frames):
        name, _, cout, _, edge, kind = LAYERS[after_conv]
        act = (edge * scale) ** 2 * cout * 4.0                  # one activation map of that layer, bytes per frame
        if after_conv == 0:
            return frames * 2 * act, 0.0                        # stem: raw in, activation out
        if after_conv <= 3:
            return frames * 2 * act, act                        # encoder: + the level's source features (shared by the batch)
        return frames * 3 * act, act                            # trunk second conv: + residual in, + source features
    apply_rows = collections.OrderedDict()
    for p in good:
        frames = p[4][3] * tile_rows(p[4][2]) / (32 * scale) ** 2
        for i in range(len(p) - 1):     # (the last conv of a pass, skipper.2, feeds the heads: no apply launch follows it)
            lo = p[i][0]
            hi = p[i + 1][0]
            for r in rows:
                if lo < r[0] < hi and r[2].startswith(("apply_kernel", "apply8_kernel")):
                    key = LAYERS[i][0] if not LAYERS[i][0].startswith("trunk") else "trunk second convs (each)"
                    per_frames, shared = apply_bytes(i, frames)
                    a = apply_rows.setdefault(key, dict(n=0, ns=0, bytes=0.0, kn=r[2]))
                    a["n"] += 1
                    a["ns"] += r[1]
                    a["bytes"] += per_frames + shared
                    break
    w("## HBM-side kernels\n")
    w("### The apply passes that remain (InstanceNorm + ReLU [+ residual] [+ Liquid-Warping-Block gather] + split), by position\n")
    w("Algorithmic bytes of THAT launch (raw read + activation write [+ residual read], per frame, + the level's cached source "
      "features once per launch: they are shared by the batch) / its rocprofv3 duration.\n")
    w("| apply after | kernel | launches | avg us | algorithmic MB per launch | GB/s | of 8.0 TB/s spec | of 6.29 TB/s measured copy |")
    w("|---|---|---|---|---|---|---|---|")
    tot_b = tot_ns = 0.0
    for k, a in apply_rows.items():
        gbs = a["bytes"] / a["ns"]
        tot_b += a["bytes"]
        tot_ns += a["ns"]
        w("| %s | `%s` | %d | %.1f | %.1f | %.0f | %.3f | %.3f |" % (k, a["kn"], a["n"], a["ns"] / a["n"] / 1e3, a["bytes"] / a["n"] / 1e6, gbs,
                                                              gbs / HBM_SPEC, gbs / HBM_COPY))
    if tot_ns:
        w("| **all %d apply launches of a pass** | | %d | %.1f (sum per pass) | %.1f (per pass) | %.0f | %.3f | %.3f |\n" % (
            sum(a["n"] for a in apply_rows.values()) // len(good), sum(a["n"] for a in apply_rows.values()),
            tot_ns / len(good) / 1e3, tot_b / len(good) / 1e6, tot_b / tot_ns, tot_b / tot_ns / HBM_SPEC, tot_b / tot_ns / HBM_COPY))
    fin = [r for r in rows if r[2].startswith("in_finalize_kernel") and t_lo <= r[0] <= t_hi]
    if fin:
        w("`in_finalize_kernel`: %d launches in the attributed window, %.1f us each, %.1f us per pass: KBs of data, pure dependency "
          "latency.\n" % (len(fin), sum(r[1] for r in fin) / len(fin) / 1e3, sum(r[1] for r in fin) / len(good) / 1e3))
    # raster (setup + tiles, fused outputs): faces 0.50 MB in; fim 0.26 + wim 0.79 + cond 0.79 + T 0.52 + tsf_img 0.79 + NHWC8 input 2.10 MB
    # out, source face vertices 0.33 + source image 0.79 MB in
    raster_frame = 495936 + 330624 + (0.262144 + 0.786432 + 0.786432 + 0.524288 + 0.786432 + 2.097152 + 0.786432) * 1e6 * s2
    nv3 = 6890 * 3
    smpl_frame = 207 * nv3 * 4 / 4.0 + nv3 * 4      # the pose-blend table once per four frames + the vertices written
    w("### Geometry kernels\n")
    w("Over the attributed passes (%d frames): algorithmic bytes per frame x frames / summed launch durations.\n" % total_frames)
    w("| kernel | launches | avg us | algorithmic MB per frame | GB/s | of 8.0 TB/s spec | of 6.29 TB/s measured copy |")
    w("|---|---|---|---|---|---|---|")
    for kname, per_frame in (("raster_tile_kernel", raster_frame), ("smpl_verts_kernel", smpl_frame)):
        sel = [r for r in rows if r[2].startswith(kname) and t_lo <= r[0] <= t_hi]
        if not sel:
            continue
        ns = sum(r[1] for r in sel)
        gbs = per_frame * total_frames / ns
        w("| `%s` | %d | %.1f | %.2f | %.0f | %.3f | %.3f |" % (kname, len(sel), ns / len(sel) / 1e3, per_frame / 1e6, gbs, gbs / HBM_SPEC,
                                                          gbs / HBM_COPY))
    w("\n(`raster_tile_kernel` and `smpl_verts_kernel` are latency-bound launches of a few dozen microseconds over the frames of a "
      "whole round: their byte rate says how far from a bandwidth problem they are, not how well they run.)")
    text = "\n".join(lines) + "\n"
    if args.out:
        open(args.out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
