"""Secondary measurement (NOT the bench.py contract): one full training iteration (generator update + discriminator
update, impersonator_trainer.py:350-366) on an MI355X, synthetic inputs.

    python tools/bench_train.py [--batch 4] [--image-size 256] [--steps 5]"""
import argparse
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impersonator_amd.models.impersonator_trainer import Impersonator  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--script-loss", action="store_true",
                    help="the loss of scripts/train_iPER.sh: --mask_bce --use_vgg --use_face (seeded VGG19 / Sphere20a weights)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"],
                    help="forward / data-gradient convolutions of the generator update")
    a = ap.parse_args()
    opt = types.SimpleNamespace(image_size=a.image_size, batch_size=a.batch, map_name='uv_seg', norm_type='instance',
                                repeat_num=6, is_train=True, conv_precision=a.precision)
    if a.script_loss:
        from tests import helpers
        opt.mask_bce, opt.use_vgg, opt.use_face = True, True, True
        opt.vgg_weights, opt.face_model = helpers.vgg19_state_dict(0), helpers.sphere20a_state_dict(0)
        opt.lambda_face, opt.lambda_mask, opt.lambda_mask_smooth = 5.0, 1.0, 1.0
    model = Impersonator(opt)
    model._G.init_weights()
    model._D.init_weights()
    g = torch.Generator().manual_seed(0)
    n, s = a.batch, a.image_size
    r = lambda *sh: (torch.rand(*sh, generator=g) * 2 - 1).cuda()
    model.set_input(r(n, 6, s, s), r(n, 3, s, s), input_G_bg=r(n, 4, s, s), input_G_src=r(n, 6, s, s),
                    T=(torch.rand(n, s, s, 2, generator=g) * 2.4 - 1.2).cuda(), real_src=r(n, 3, s, s),
                    bg_mask=(torch.rand(2 * n, 1, s, s, generator=g) > 0.5).float().cuda())
    if a.script_loss:   # head boxes as BodyRecoveryFlow.cal_head_bbox would give them (a 1/5-size box near the top)
        model._head_bbox = torch.tensor([[s * 2 // 5, s * 3 // 5, s // 10, s * 3 // 10]] * n)
    losses = model.optimize_parameters()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = model.optimize_parameters()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"metric": "training iteration (G update + D update)", "ms_per_iteration": round(dt * 1e3, 2),
                      "note": "BASELINE.json config 5 is --image-size 512 (--batch 1..4 per GPU)",
                      "images_per_s": round(n / dt, 2), "batch": n, "image_size": s, "loss": "train_iPER.sh (mask_bce, vgg, face)" if a.script_loss else "adv + L1 + mask", "dtype": "f32" if a.precision == "fp32" else "bf16x3 generator convs (forward, data and weight gradient) + f32", "losses": losses}))


if __name__ == "__main__":
    main()
