"""Development aid: time of the per-frame geometry launch sequence (SMPLRenderer.transfer: setup + tile kernel) alone.
    python tools/raster_bench.py [iters=200]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impersonator_amd import demo  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
im, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=16, seed=0)
im.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
smpls = torch.from_numpy(demo.synthetic_smpls(1024, seed=0)).cuda()
im.first_cam = smpls[0:1, 0:3].clone()
for bs in (1, 8, 16):
    tsf = im.swap_smpl(im.src_info['cam'], im.src_info['shape'], smpls[8:8 + bs])
    info = im.hmr.get_details(tsf)
    for _ in range(10):
        im.hmr.get_details(tsf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        im.hmr.get_details(tsf)
    e1.record()
    torch.cuda.synchronize()
    print("get_details (SMPL) bs=%2d: %.1f us per call" % (bs, e0.elapsed_time(e1) * 1e3 / iters))
    r = im.render
    for _ in range(10):
        r.transfer(info['cam'], info['verts'], im.src_info['p2verts_c'], im.src_info['img'])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        r.transfer(info['cam'], info['verts'], im.src_info['p2verts_c'], im.src_info['img'])
    e1.record()
    torch.cuda.synchronize()
    print("transfer bs=%2d: %.1f us per call (%.1f us per frame), includes 7 output allocations" %
          (bs, e0.elapsed_time(e1) * 1e3 / iters, e0.elapsed_time(e1) * 1e3 / iters / bs))
