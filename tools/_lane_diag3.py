import os, sys, torch
sys.path.insert(0, ".")
from impersonator_amd import demo
n = int(sys.argv[1])
im, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=4, seed=0, affine="random")
im.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
smpls = torch.from_numpy(demo.synthetic_smpls(24, seed=3)).cuda()
im.first_cam = smpls[0:1, 0:3].clone()
chunks = [(smpls[s:s + 4], s) for s in range(0, 24, 4)]
seq, ins = [], []
for chunk, t in chunks:
    x = im.transfer_params_by_smpl(chunk, "smooth", t=t)
    ins.append((x, im.tsf_info["T"]))
    seq.append(im.forward(x, im.tsf_info["T"]).clone())
torch.cuda.synchronize()
main = torch.cuda.current_stream()
bad = tot = 0
for nl in (2, 3):
    lanes = im._lanes(nl)
    for r in range(n):
        outs = []
        for j, (x, T) in enumerate(ins):
            st, gen = lanes[j % nl]
            st = main if st is None else st
            with torch.cuda.stream(st):
                outs.append(im.forward(x, T, generator=gen))
        torch.cuda.synchronize()
        tot += 1
        for k in range(6):
            if not torch.equal(outs[k], seq[k]):
                print("lanes", nl, "rep", r, "batch", k, "pred wrong", float((outs[k] - seq[k]).abs().max()), flush=True)
                bad += 1
print("generator-only concurrency failures:", bad, "of", tot, "passes")
