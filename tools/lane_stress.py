"""Stress check of Imitator.predict_batches (development aid): N passes of six batches through the lane pipeline,
every batch compared bit for bit with transfer_params_by_smpl + forward run one after the other.
    python tools/lane_stress.py [passes=40] [lanes=2,3] [batch=8] [overlap=0]
overlap=1 lifts the barrier between a round's generators and the next round's geometry (Imitator.predict_batches keeps
it): the configuration that shows stale geometry records under concurrent bf16x3 convolutions.
This is the run that exposed the stale-depth-key glitch of the round-1 rasteriser (global 64-bit atomics) and now
guards its tile-owned replacement."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impersonator_amd import demo  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lane_counts = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "2,3").split(",")]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
OVERLAP = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
im, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=B, seed=0, affine="random")
im.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
if OVERLAP:
    im.round_depth = 1   # rounds of `lanes` batches: with deeper rounds a six-batch pass is one round and nothing overlaps
smpls = torch.from_numpy(demo.synthetic_smpls(6 * B, seed=3)).cuda()
im.first_cam = smpls[0:1, 0:3].clone()
chunks = [(smpls[s:s + B], s) for s in range(0, 6 * B, B)]
seq = []
for chunk, t in chunks:
    x = im.transfer_params_by_smpl(chunk, "smooth", t=t)
    seq.append(im.forward(x, im.tsf_info["T"]).clone())
torch.cuda.synchronize()
bad = tot = 0
for p in range(passes):
    for nl in lane_counts:
        got = [q.clone() for _, q in im.predict_batches(iter(chunks), "smooth", lanes=nl, overlap_geometry=OVERLAP)]   # no sync per batch
        torch.cuda.synchronize()
        tot += 1
        for k, (a, b) in enumerate(zip(got, seq)):
            if not torch.equal(a, b):
                print("pass", p, "lanes", nl, "batch", k, "differs, max |d| =", float((a - b).abs().max()), flush=True)
                bad += 1
print("differing batches: %d in %d passes of %d batches" % (bad, tot, len(chunks)))
sys.exit(1 if bad else 0)
