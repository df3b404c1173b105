#!/usr/bin/env python
"""CPU experiment: how far does the final image move when the generator's per-frame convolutions evaluate fewer split products?

The product path evaluates x*w as xh*wh + xh*wl + xl*wh on bf16 halves (3 MFMA products, error ~2^-17).  Candidates with TWO
products on the same matrix pipe (fp16 runs at the bf16 rate on gfx950):
    a16      x rounded to ONE fp16 term, w = fp16 hi + lo     (xh*wh + xh*wl)
    w16      w rounded to ONE fp16 term, x = fp16 hi + lo     (xh*wh + xl*wh)
    a_bf16   x rounded to one bf16 term, w split              (for scale)
Evaluated by rounding the operand in the fp32 torch oracle (products and sums stay fp32 -- the rounding of the dropped term is
the whole effect), on the 256x256 main variant of tests/helpers.py, 4 frames.  Prints L-inf of the final image vs plain fp32."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import numpy as np
import torch
import torch.nn.functional as F
from oracle import torch_ref
from tests import helpers

real_conv, real_convT = F.conv2d, F.conv_transpose2d
MODE = [None]


def q(x, dt):
    return x.to(dt).to(torch.float32)


def conv(x, w, *a, **k):
    m = MODE[0]
    if m == "a16": x = q(x, torch.float16)
    elif m == "w16": w = q(w, torch.float16)
    elif m == "a_bf16": x = q(x, torch.bfloat16)
    elif m == "aw16": x, w = q(x, torch.float16), q(w, torch.float16)
    return real_conv(x, w, *a, **k)


def convT(x, w, *a, **k):
    m = MODE[0]
    if m == "a16": x = q(x, torch.float16)
    elif m == "w16": w = q(w, torch.float16)
    elif m == "a_bf16": x = q(x, torch.bfloat16)
    elif m == "aw16": x, w = q(x, torch.float16), q(w, torch.float16)
    return real_convT(x, w, *a, **k)


def main():
    from impersonator_amd.networks.batch_smpl import HumanModelRecovery
    sc = helpers.imitator_scene(256)
    t = torch.from_numpy
    outs = {}
    for seed in (0, 4):
        sd = torch_ref.state_dict_from_numpy(helpers.generator_state_dict(seed=seed, affine="random"))
        faces, map_fn = t(sc["faces"]), t(sc["map_fn"])
        hmr = HumanModelRecovery(smpl_params=sc["smpl_params"])
        with torch.no_grad():
            si = hmr.get_details(t(sc["src_smpl"])[None])
            src = torch_ref.imitator_personalize(sd, t(sc["src_img"]), si, faces, map_fn, image_size=256)
            for mode in (None, "a16", "w16", "aw16", "a_bf16"):
                MODE[0] = mode
                torch_ref.F.conv2d, torch_ref.F.conv_transpose2d = conv, convT
                try:
                    frames = torch_ref.imitator_inference_by_smpls(sd, src, hmr.get_details, t(sc["tgt_smpls"]), faces, map_fn,
                                                                   cam_strategy="smooth", image_size=256)
                finally:
                    torch_ref.F.conv2d, torch_ref.F.conv_transpose2d = real_conv, real_convT
                outs[mode] = np.stack([f["preds"].numpy() for f in frames])
            for mode in ("a16", "w16", "aw16", "a_bf16"):
                d = np.abs(outs[mode] - outs[None])
                print("seed %d  %-7s  linf %.3e  mean %.3e" % (seed, mode, d.max(), d.mean()))


if __name__ == "__main__":
    main()
