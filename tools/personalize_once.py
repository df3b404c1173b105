#!/usr/bin/env python
"""One Imitator.personalize of the synthetic configuration with an InpaintSANet background model and --only_vis (the variant with the
most glue), for `rocprofv3 --kernel-trace --stats`: every kernel of the call must be liblwg's (tools/r05_profile.sh; the GPU test
tests/test_gpu_personalize_glue.py checks the same through the torch profiler).  Module construction and the weight uploads of the
first call are in the trace too (copyBuffer rows); ATen kernels (at::native::...) must not be."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impersonator_amd import demo  # noqa: E402
from impersonator_amd.networks.inpaintor import InpaintSANet  # noqa: E402
from impersonator_amd.utils import synthetic  # noqa: E402

opt = demo.default_opt(batch_size=8, image_size=256, only_vis=True)
imitator, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=8, seed=0, image_size=256, opt=opt)
net = InpaintSANet(c_dim=4).eval()
shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.random_inpaintor_state_dict(shapes, 1).items()})
imitator.bgnet = net.cuda()
for _ in range(3):
    imitator.personalize(src_img, src_smpl=src_smpl)
torch.cuda.synchronize()
bg = imitator.src_info["bg"].cpu().numpy()          # (a device-side min / max would put ATen reduce kernels into the trace)
print("personalized; background range %.3f..%.3f" % (bg.min(), bg.max()))
