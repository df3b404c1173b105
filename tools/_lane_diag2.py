import os, sys, torch
sys.path.insert(0, ".")
from impersonator_amd import demo
n = int(sys.argv[1])
im, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=4, seed=0, affine="random")
im.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
smpls = torch.from_numpy(demo.synthetic_smpls(24, seed=3)).cuda()
im.first_cam = smpls[0:1, 0:3].clone()
chunks = [(smpls[s:s + 4], s) for s in range(0, 24, 4)]
seq = []
for chunk, t in chunks:
    x = im.transfer_params_by_smpl(chunk, "smooth", t=t)
    seq.append(im.forward(x, im.tsf_info["T"]).clone())
torch.cuda.synchronize()
bad = tot = 0
for cold in range(n):
    for nl in (2, 3):
        for r in range(3):
            got = [p.clone() for _, p in im.predict_batches(iter(chunks), "smooth", lanes=nl)]
            torch.cuda.synchronize()
            tot += 1
            for k in range(6):
                if not torch.equal(got[k], seq[k]):
                    print("cold", cold, "lanes", nl, "rep", r, "batch", k, "pred wrong", flush=True)
                    bad += 1
print("failures:", bad, "of", tot, "passes")
