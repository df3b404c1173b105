#!/usr/bin/env python
"""Writes the frames of a short synthetic imitation run to an .npy file -- the A/B helper of the tests that compare two builds of one
pass selected by an environment switch the library reads once per process (LWG_APPLY8, LWG_FUSED_APPLY, LWG_HALO, ...):

    LWG_APPLY8=0 python tools/dump_preds.py out_a.npy [frames] ; python tools/dump_preds.py out_b.npy [frames]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impersonator_amd import demo  # noqa: E402

out, frames = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24
imitator, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=8, seed=0, image_size=256)
imitator.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
smpls = demo.synthetic_smpls(64, seed=0)[:frames]
preds = np.stack(imitator.inference_by_smpls(smpls, cam_strategy="smooth"))
torch.cuda.synchronize()
np.save(out, preds)
print("wrote %s %s" % (out, preds.shape))
