import os, sys, torch
sys.path.insert(0, ".")
from impersonator_amd import demo
n = int(sys.argv[1])
im, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=4, seed=0, affine="random")
im.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
smpls = torch.from_numpy(demo.synthetic_smpls(24, seed=3)).cuda()
im.first_cam = smpls[0:1, 0:3].clone()
chunks = [(smpls[s:s + 4], s) for s in range(0, 24, 4)]
keys = ("verts", "cam", "f2verts", "fim", "wim", "T")
orig_transfer = im.render.transfer
last = {}
def transfer(*a, **k):
    out = orig_transfer(*a, **k)
    last["f2verts"] = out["f2verts"]
    return out
im.render.transfer = transfer
seq = []
for chunk, t in chunks:
    x = im.transfer_params_by_smpl(chunk, "smooth", t=t)
    im.tsf_info["f2verts"] = last["f2verts"]
    seq.append([im.forward(x, im.tsf_info["T"]).clone()] + [im.tsf_info[k].clone() for k in keys])
torch.cuda.synchronize()
orig = im.transfer_params_by_smpl
snaps = []
def patched(chunk, cam_strategy="smooth", t=0):
    x = orig(chunk, cam_strategy, t=t)
    im.tsf_info["f2verts"] = last["f2verts"]
    snaps.append([im.tsf_info[k].clone() for k in keys])
    return x
im.transfer_params_by_smpl = patched
shown = 0
for cold in range(n):
    for nl in (2, 3):
        for r in range(3):
            snaps.clear()
            got = [p.clone() for _, p in im.predict_batches(iter(chunks), "smooth", lanes=nl)]
            torch.cuda.synchronize()
            for k in range(6):
                if not torch.equal(got[k], seq[k][0]) and shown < 4:
                    shown += 1
                    print("cold", cold, "lanes", nl, "rep", r, "batch", k, "pred wrong; geometry snapshot:")
                    for name, s_, q in zip(keys, snaps[k], seq[k][1:]):
                        d = (s_ != q)
                        per = d.reshape(d.shape[0], -1).sum(1).tolist()
                        print("    %-8s differing elems per sample %s" % (name, per))
                    d = (snaps[k][3] != seq[k][4])
                    idx = d.nonzero()
                    print("    fim wrong at (b,y,x):", idx[:20].tolist(), "got", snaps[k][3][d][:20].tolist(), "want", seq[k][4][d][:20].tolist())
                    fd = (snaps[k][2] != seq[k][3]).reshape(4, -1, 9).any(-1).nonzero()
                    print("    f2verts wrong faces (b,f):", fd[:20].tolist())
print("done")
