"""Experiment: two independent Imitator pipelines (own generator handle, own streams) fed from two host threads,
against one pipeline -- does overlapping the batches of two streams fill the launch gaps of one dependent chain?"""
import sys, threading, time
import torch
sys.path.insert(0, ".")
from impersonator_amd import demo

BATCH, STEPS = 8, 100
dev = torch.device("cuda", 0)
smpls = torch.from_numpy(demo.synthetic_smpls(1024, seed=0)).to(dev)


def make():
    im, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=BATCH, seed=0, image_size=256)
    im.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
    im.first_cam = smpls[0:1, 0:3].clone()
    return im


def run(im, first, n, stream):
    with torch.cuda.stream(stream), torch.no_grad():
        chunks = ((smpls[((first + i) * BATCH) % 1024:((first + i) * BATCH) % 1024 + BATCH], (first + i) * BATCH) for i in range(n))
        for _, out in im.predict_batches(chunks, "smooth"):
            pass
    return out


ims = [make(), make()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for im, st in zip(ims, streams):
    run(im, 0, 30, st)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    run(ims[0], 0, STEPS, streams[0])
    torch.cuda.synchronize()
    one = STEPS * BATCH / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(im, k * 64, STEPS // 2, st)) for k, (im, st) in enumerate(zip(ims, streams))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    two = STEPS * BATCH / (time.perf_counter() - t0)
    print("one pipeline %.1f fps   two concurrent pipelines %.1f fps   (%+.1f %%)" % (one, two, (two / one - 1) * 100), flush=True)
