"""Development aid: SMPLRenderer.transfer (lwg_transfer_frame: projection + records + tiles + fused epilogue)
on FIXED inputs of `bs` frames beside bf16x3 convolutions: which pixels go wrong.  python tools/overlap_detail2.py [bs=16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impersonator_amd import demo, ops  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
im, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=8, seed=0, affine="random")
im.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
smpls = torch.from_numpy(demo.synthetic_smpls(48, seed=3)).cuda()
im.first_cam = smpls[0:1, 0:3].clone()
im.transfer_params_by_smpl(smpls[32:32 + bs], "smooth", t=32)
info = im.tsf_info
cam, verts = info["cam"].clone(), info["verts"].clone()
si = im.src_info
ref = im.render.transfer(cam, verts, si["p2verts"], si["img"])
ref = {k: v.clone() for k, v in ref.items()}
S = im.render.image_size
xx, ww = torch.randn(8, 32, 32, 512, device="cuda"), torch.randn(512, 512, 3, 3, device="cuda") * 0.02
lanes = [torch.cuda.Stream(), torch.cuda.Stream()]
side = torch.cuda.Stream()
torch.cuda.synchronize()
keep = []
for it in range(150):
    for st in lanes:
        with torch.cuda.stream(st):
            for _ in range(12):
                ops.conv2d_forward(xx, ww, None, 1, 1, precision="bf16x3")
    with torch.cuda.stream(side):
        for j in range(2):
            keep.append(im.render.transfer(cam, verts, si["p2verts"], si["img"]))
torch.cuda.synchronize()
shown = wrong = 0
for k, out in enumerate(keep):
    nb = {key: int((out[key] != ref[key]).sum()) for key in ("f2verts", "fim", "wim", "T")}
    if not any(nb.values()):
        continue
    wrong += 1
    if shown >= 6:
        continue
    shown += 1
    print("launch %d: differing elements %s" % (k, nb))
    bad = (out["fim"] != ref["fim"]).nonzero()
    tiles = {}
    for b, y, x in bad.tolist():
        yy = S - 1 - y
        tiles.setdefault((b, yy // 8, x // 32), []).append((yy, x, int(ref["fim"][b, y, x]), int(out["fim"][b, y, x])))
    for (b, ty, tx), px in sorted(tiles.items())[:8]:
        print("   frame %d tile (%d,%d): %d pixels rows %d-%d cols %d-%d expected %s got %s"
              % (b, ty, tx, len(px), min(p[0] for p in px), max(p[0] for p in px), min(p[1] for p in px), max(p[1] for p in px),
                 sorted({e for _, _, e, _ in px})[:6], sorted({g for _, _, _, g in px})[:6]))
print("bs=%d: wrong launches %d of %d" % (bs, wrong, len(keep)))
