"""Development aid: which per-batch tensors differ between the lane pipeline (geometry overlapped with the generators
of other streams) and the sequential path.  python tools/overlap_diag.py [passes=20] [lanes=2] [batch=8] [overlap=1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impersonator_amd import demo  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
OVERLAP = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
KEYS = ("theta", "verts", "cam", "f2verts", "fim", "wim", "cond", "T", "tsf_img")
im, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=B, seed=0, affine="random")
im.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
if OVERLAP:
    im.round_depth = 1   # rounds of `lanes` batches: with deeper rounds a six-batch pass is one round and nothing overlaps
_orig_transfer = im.render.transfer
_orig_tp = im.transfer_params_by_smpl


def _tp(tgt_smpl, cam_strategy='smooth', t=0):
    stash = {}

    def _tr(*a, **k):
        out = _orig_transfer(*a, **k)
        stash['f2verts'] = out['f2verts']
        return out
    im.render.transfer = _tr
    x = _orig_tp(tgt_smpl, cam_strategy, t)
    im.render.transfer = _orig_transfer
    im.tsf_info['f2verts'] = stash['f2verts']
    return x


im.transfer_params_by_smpl = _tp
smpls = torch.from_numpy(demo.synthetic_smpls(6 * B, seed=3)).cuda()
im.first_cam = smpls[0:1, 0:3].clone()
chunks = [(smpls[s:s + B], s) for s in range(0, 6 * B, B)]
seq = []
for chunk, t in chunks:
    x = im.transfer_params_by_smpl(chunk, "smooth", t=t)
    d = {k: im.tsf_info[k].clone() for k in KEYS}
    d["tsf_inputs"] = x.clone()
    d["pred"] = im.forward(x, im.tsf_info["T"]).clone()
    seq.append(d)
torch.cuda.synchronize()
count = {}
shown = 0
for p in range(passes):
    got = []
    for _, q in im.predict_batches(iter(chunks), "smooth", lanes=nl, overlap_geometry=OVERLAP):
        d = {k: im.tsf_info[k].clone() for k in KEYS}
        d["pred"] = q.clone()
        got.append(d)
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(got, seq)):
        for key in a:
            if not torch.equal(a[key], b[key]):
                count[(k, key)] = count.get((k, key), 0) + 1
                if shown < 12:
                    shown += 1
                    ne = (a[key] != b[key])
                    idx = ne.nonzero()
                    print("pass %d batch %d %s: %d of %d elements differ; first %s last %s" %
                          (p, k, key, int(ne.sum()), ne.numel(), idx[0].tolist(), idx[-1].tolist()), flush=True)
                    if key == "fim":
                        prev = seq[k - nl]["fim"] if k >= nl else None
                        for j in idx[:24].tolist():
                            print("    px", j, "seq", int(b[key][tuple(j)]), "got", int(a[key][tuple(j)]),
                                  "prev-round seq", int(prev[tuple(j)]) if prev is not None else None)
                    if key == "f2verts":
                        faces = sorted(set((j[0], j[1]) for j in idx.tolist()))
                        print("    faces (frame, face):", faces[:40], "count", len(faces))
print("overlap=%s lanes=%d passes=%d:" % (OVERLAP, nl, passes))
for (k, key), n in sorted(count.items()):
    print("  batch %d %-8s differs in %d passes" % (k, key, n))
if not count:
    print("  no difference")
