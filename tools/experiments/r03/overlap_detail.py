"""Development aid: what exactly is wrong in a rasteriser launch that ran beside bf16x3 convolutions?
Prints, for the first few wrong launches, the wrong pixels grouped by 32x8 tile with expected / obtained face ids."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impersonator_amd import demo, ops, _lib  # noqa: E402

im, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=8, seed=0)
im.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
smpls = torch.from_numpy(demo.synthetic_smpls(64, seed=0)).cuda()
im.first_cam = smpls[0:1, 0:3].clone()
im.transfer_params_by_smpl(smpls[:8], "smooth", t=0)
info = im.tsf_info
f2v, fim_ref, _ = im.render.render_fim_wim(info["cam"].clone(), info["verts"].clone())
f2v = f2v.clone()
lib = _lib.load()
bs, nf = f2v.shape[:2]
S = im.render.image_size
ws = torch.zeros(lib.lwg_rasterize_workspace_bytes(bs, nf, S), dtype=torch.uint8, device="cuda")
xx, ww = torch.randn(8, 32, 32, 512, device="cuda"), torch.randn(512, 512, 3, 3, device="cuda") * 0.02
lanes = [torch.cuda.Stream(), torch.cuda.Stream()]
side = torch.cuda.Stream()
torch.cuda.synchronize()
keep = []
for it in range(200):
    for st in lanes:
        with torch.cuda.stream(st):
            for _ in range(12):
                ops.conv2d_forward(xx, ww, None, 1, 1, precision="bf16x3")
    with torch.cuda.stream(side):
        for j in range(2):
            fim = torch.empty((bs, S, S), device="cuda", dtype=torch.int32)
            wim = torch.empty((bs, S, S, 3), device="cuda", dtype=torch.float32)
            _lib.check(lib.lwg_rasterize_fim_wim(_lib.ptr(f2v), bs, nf, S, im.render.RASTER_NEAR, im.render.RASTER_FAR, _lib.ptr(fim),
                                                 _lib.ptr(wim), None, _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
            keep.append(fim)
torch.cuda.synchronize()
shown = 0
for k, fim in enumerate(keep):
    bad = (fim != fim_ref).nonzero()
    if not len(bad):
        continue
    shown += 1
    if shown > 6:
        continue
    print("launch %d: %d wrong pixels" % (k, len(bad)))
    tiles = {}
    for b, y, x in bad.tolist():
        yy = S - 1 - y                      # the maps are flipped on the way out: tile rows are in the pre-flip orientation
        tiles.setdefault((b, yy // 8, x // 32), []).append((yy, x, int(fim_ref[b, y, x]), int(fim[b, y, x])))
    for (b, ty, tx), px in sorted(tiles.items()):
        exp = sorted({e for _, _, e, _ in px})
        got = sorted({g for _, _, _, g in px})
        print("   frame %d tile (%d,%d): %d pixels, rows %d-%d cols %d-%d, expected faces %s, got %s"
              % (b, ty, tx, len(px), min(p[0] for p in px), max(p[0] for p in px), min(p[1] for p in px), max(p[1] for p in px),
                 exp[:8], got[:8]))
print("wrong launches: %d of %d" % (shown, len(keep)))
for b, fn in ((7, 3632), (4, 10590)):
    where = (fim_ref[b] == fn).nonzero()
    v = f2v[b, fn].cpu()
    px = ((v[:, 0] + 1) * S - 1) / 2
    py = ((v[:, 1] + 1) * S - 1) / 2      # pre-flip orientation
    print("frame %d face %d: %d pixels in the reference map%s; vertices (pixel x, y pre-flip, z): %s"
          % (b, fn, len(where), "" if not len(where) else " rows %d-%d cols %d-%d (post-flip rows)" % (
              int(where[:, 0].min()), int(where[:, 0].max()), int(where[:, 1].min()), int(where[:, 1].max())),
             [(round(float(px[k]), 2), round(float(py[k]), 2), round(float(v[k, 2]), 3)) for k in range(3)]))
