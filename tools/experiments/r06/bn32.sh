#!/bin/bash
# exact-fp32 launches with few tiles on 32-channel tiles: correctness, then personalize / one-frame latency A/B
timeout 900 python -m pytest tests/test_gpu_generator.py -x -q -k "32_channel or matches or encode" 2>&1 | tail -5
for v in 1 0 1 0; do
LWG_F32_BN32=$v timeout 300 python - <<'PY'
import os, time, torch, numpy as np
from impersonator_amd import demo
from impersonator_amd.networks.inpaintor import InpaintSANet
from impersonator_amd.utils import synthetic
out = []
for variant in ("ORIGINAL", "deepfillv2"):
    imitator, src_smpl, src_img, _ = demo.build_synthetic_imitator(batch_size=1, seed=0)
    if variant == "deepfillv2":
        net = InpaintSANet(c_dim=4).eval()
        shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.random_inpaintor_state_dict(shapes, 1).items()})
        imitator.bgnet = net.cuda()
    ms = []
    for i in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        imitator.personalize(src_img, src_smpl=src_smpl)
        torch.cuda.synchronize()
        if i >= 3: ms.append((time.perf_counter() - t0) * 1e3)
    out.append("%s personalize %.3f ms" % (variant, sorted(ms)[len(ms) // 2]))
    if variant == "ORIGINAL":
        imitator.generator.precision = "fp32"
        smpls = torch.from_numpy(demo.synthetic_smpls(64, seed=0)).cuda()
        run = imitator.frame_graph(batch=1)
        ms = []
        for t in range(4, 44):
            t0 = time.perf_counter(); run(smpls[t:t + 1], t=t); torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
        out.append("fp32 one frame per call (graph) %.3f ms" % sorted(ms)[len(ms) // 2])
    imitator.generator.release()
print("LWG_F32_BN32=%s: " % os.environ.get("LWG_F32_BN32") + "; ".join(out))
PY
done
