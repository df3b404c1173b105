"""Development aid for DESIGN.md section 5.1: does a plain producer -> consumer kernel pair (torch's own fill / compare and
copy / compare kernels on one stream, nothing of this library's geometry code) ever read stale data while this library's
generator passes run on two other streams?   python tools/overlap_repro.py [iterations=300]

The geometry kernels do (tools/lane_stress.py ... 1): records the setup kernel stores are read stale by the tile kernel in
16-lane groups, only beside bf16x3 generators.  This script separates "any kernel pair under that load" from "something
about the geometry kernels"."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impersonator_amd import demo  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
im, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=8, seed=0)
im.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
smpls = torch.from_numpy(demo.synthetic_smpls(64, seed=0)).cuda()
im.first_cam = smpls[0:1, 0:3].clone()
x = im.transfer_params_by_smpl(smpls[:8], "smooth", t=0)
T = im.tsf_info["T"]
torch.cuda.synchronize()

side = torch.cuda.Stream()
n = 1 << 20
probe = torch.zeros(n, dtype=torch.int32, device="cuda")                 # 4 MB: filled, then compared
rec = torch.zeros(13776 * 8, 7, dtype=torch.float32, device="cuda")      # shaped like the per-face records of a batch
srcs = [torch.full_like(rec, float(k)) for k in range(4)]
# this library's own geometry kernels on fixed inputs: the projection alone (one kernel, no hand-over through memory),
# and the rasteriser on given faces (setup kernel -> per-face records in the workspace -> tile kernel)
info = im.tsf_info
verts, cam = info["verts"].clone(), info["cam"].clone()
_, fim_ref, _ = im.render.render_fim_wim(cam, verts)
f2v_ref = im.render.render_fim_wim(cam, verts)[0].clone()
from impersonator_amd import _lib  # noqa: E402
lib = _lib.load()
bs8, nf = f2v_ref.shape[:2]
S = im.render.image_size
ws_bytes = lib.lwg_rasterize_workspace_bytes(bs8, nf, S)


def raster_ws(faces, ws):
    """lwg_rasterize_fim_wim on given faces with a caller-owned, zeroed workspace (unwritten record slots then compare equal)"""
    if not os.environ.get("REPRO_NOZERO"):   # without it every byte the tile kernel reads holds the same value launch after launch
        ws.zero_()
    fim = fim_buf if os.environ.get("REPRO_NOZERO") else torch.empty((bs8, S, S), device="cuda", dtype=torch.int32)
    wim = torch.empty((bs8, S, S, 3), device="cuda", dtype=torch.float32)
    _lib.check(lib.lwg_rasterize_fim_wim(_lib.ptr(faces), bs8, nf, S, im.render.RASTER_NEAR, im.render.RASTER_FAR, _lib.ptr(fim),
                                         _lib.ptr(wim), None, _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
    return fim


fim_buf = torch.empty((bs8, S, S), device="cuda", dtype=torch.int32)
ws_ref = torch.zeros(ws_bytes, dtype=torch.uint8, device="cuda")
ws_run = torch.zeros(ws_bytes, dtype=torch.uint8, device="cuda")
fim_ref2 = raster_ws(f2v_ref, ws_ref)
assert bool((fim_ref2 == fim_ref).all())
torch.cuda.synchronize()
ga = {dt: (torch.randn(8192, 8192, device="cuda", dtype=dt), torch.randn(8192, 8192, device="cuda", dtype=dt))
      for dt in (torch.bfloat16, torch.float32)}
from impersonator_amd import ops  # noqa: E402
cx = {"conv128": (torch.randn(8, 32, 32, 512, device="cuda"), torch.randn(512, 512, 3, 3, device="cuda") * 0.02, "bf16x3"),
      "conv128_fp32": (torch.randn(8, 32, 32, 512, device="cuda"), torch.randn(512, 512, 3, 3, device="cuda") * 0.02, "fp32"),
      "conv64": (torch.randn(8, 256, 256, 64, device="cuda"), torch.randn(64, 64, 3, 3, device="cuda") * 0.05, "bf16x3"),
      "conv1x1": (torch.randn(8, 32, 32, 512, device="cuda"), torch.randn(512, 512, 1, 1, device="cuda") * 0.05, "bf16x3")}
modes = os.environ.get("REPRO_MODES", "bf16x3,fp32").split(",")
for mode in modes:
    # neighbours on the two lane streams: this library's generator passes ("bf16x3" / "fp32"), or plain library GEMMs
    # ("gemm_bf16": dense bf16 MFMA load without any of this library's kernels, "gemm_fp32")
    gemm = {"gemm_bf16": torch.bfloat16, "gemm_fp32": torch.float32}.get(mode)
    if gemm is None and mode not in cx:
        im.generator.precision = mode
    lanes = im._lanes(2)
    bad = torch.zeros(3, dtype=torch.int64, device="cuda")
    npx = torch.zeros(1, dtype=torch.int64, device="cuda")
    cases = torch.zeros(4, dtype=torch.int64, device="cuda")   # [records wrong & fim wrong, records ok & fim wrong, records wrong & fim ok, wrong record bytes]
    c = 0
    for st, _ in lanes:
        st.wait_stream(torch.cuda.current_stream())
    side.wait_stream(torch.cuda.current_stream())
    for it in range(iters):
        for st, gen in lanes:
            with torch.cuda.stream(st):
                if mode in cx:   # the op-level conv alone: split_pack + weight re-layout + conv_igemm_bf16x3 (or the fp32 kernel)
                    xx, ww, prec = cx[mode]
                    for _ in range(12):
                        ops.conv2d_forward(xx, ww, None, 1, (ww.shape[2] - 1) // 2, precision=prec)
                elif gemm is None:
                    im.forward(x, T, generator=gen)
                else:
                    for _ in range(2):
                        torch.matmul(ga[gemm][0], ga[gemm][1])
        with torch.cuda.stream(side):
            for j in range(12):
                c += 1
                probe.fill_(c)
                bad[0] += (probe != c).sum()
                rec.copy_(srcs[c & 3])
                bad[1] += (rec != float(c & 3)).sum()
                if j < 2:   # the rasteriser itself (as SMPLRenderer.render_fim_wim launches it), same inputs every time
                    _, fim, _ = im.render.render_fim_wim(cam, verts)
                    bad[2] += (fim != fim_ref).sum()
                elif j < 4:   # setup + tile on FIXED faces, records compared with an isolated run's
                    fim = raster_ws(f2v_ref, ws_run)
                    a = (fim != fim_ref).any().long()
                    npx[0] += (fim != fim_ref).sum()
                    nb = (ws_run != ws_ref).sum()
                    b = (nb > 0).long()
                    cases[0] += a * b
                    cases[1] += a * (1 - b)
                    cases[2] += (1 - a) * b
                    cases[3] += nb
    torch.cuda.synchronize()
    print("neighbours %-9s: %d producer/consumer pairs of each kind, stale elements: fill/compare %d, copy/compare %d; "
          "%d rasteriser launches on fixed inputs, wrong fim pixels: %d" % (mode, c, int(bad[0]), int(bad[1]), 2 * iters, int(bad[2])))
    print("    setup + tile on fixed faces, %d launches: records wrong & fim wrong %d, records ok & fim wrong %d, records wrong & fim ok "
          "%d (wrong record bytes in total: %d; wrong pixels %d)" % (2 * iters, int(cases[0]), int(cases[1]), int(cases[2]), int(cases[3]), int(npx[0])))
