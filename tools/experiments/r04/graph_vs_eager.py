#!/usr/bin/env python
"""Separates the two things optimize_parameters_graphed changes: Adam's step count on the device, and the replay itself."""
import os, sys
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import bench_train
from impersonator_amd import _lib

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
A, B, C, D = (bench_train.build(2, 64, prec, seed=3) for _ in range(4))
tr = lambda m: m._generator_trainer()
for it in range(6):
    if it == 2:   # C: device step, eager
        tr(C).use_device_step(True)
        _lib.check(_lib.load().lwg_discriminator_use_device_step(C._D._ensure_handle(), 1, 0.5, 0.999))
    a, b, c, d = A.optimize_parameters(), B.optimize_parameters(), C.optimize_parameters(), D.optimize_parameters_graphed()
    pa = tr(A).flat_p
    print("it %d  g_tsf A %.7f  B-A %+.2e  devstep-A %+.2e  graph-A %+.2e | d_loss A %.6f  B-A %+.2e devstep-A %+.2e graph-A %+.2e | |p| diffs B %.2e C %.2e D %.2e"
          % (it, a["g_tsf"], b["g_tsf"] - a["g_tsf"], c["g_tsf"] - a["g_tsf"], d["g_tsf"] - a["g_tsf"], a["d_loss"], b["d_loss"] - a["d_loss"],
             c["d_loss"] - a["d_loss"], d["d_loss"] - a["d_loss"], float((tr(B).flat_p - pa).abs().max()), float((tr(C).flat_p - pa).abs().max()),
             float((tr(D).flat_p - pa).abs().max())))
print("t_dev C", int(tr(C).t_dev[0]), "t_dev D", int(tr(D).t_dev[0]), "host t A", tr(A).t)
