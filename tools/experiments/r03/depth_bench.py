"""Development aid: frames/s of Imitator.predict_batches for several (lanes, round_depth) settings, interleaved in one
process (same clocks, same thermal state): python tools/depth_bench.py [repeats=5] [steps=96]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impersonator_amd import demo  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 96
im, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=8, seed=0)
im.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
smpls = torch.from_numpy(demo.synthetic_smpls(1024, seed=0)).cuda()
im.first_cam = smpls[0:1, 0:3].clone()


def run(n, lanes, depth, overlap=True):
    im.round_depth = depth
    im.overlap_geometry = overlap
    out = None
    chunks = ((smpls[(i % 128) * 8:(i % 128) * 8 + 8], (i % 128) * 8) for i in range(n))
    for _, out in im.predict_batches(chunks, "smooth", lanes=lanes):
        pass
    return out


settings = [(2, 1, True), (2, 2, True), (2, 4, True), (2, 4, False), (2, 1, False), (3, 2, True), (3, 4, True)]
for s in settings:
    run(24, *s)
torch.cuda.synchronize()
res = {s: [] for s in settings}
for r in range(reps):
    for s in settings:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps, *s)
        torch.cuda.synchronize()
        res[s].append(steps * 8 / (time.perf_counter() - t0))
for s in settings:
    v = sorted(res[s])
    print("lanes %d depth %d overlap %d: median %.0f fps (min %.0f max %.0f)" % (s[0], s[1], s[2], v[len(v) // 2], v[0], v[-1]))
