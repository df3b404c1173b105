#!/usr/bin/env python
"""Appearance transfer entry point -- command line of the reference's run_swap.py (run_swap.py:39-69).

    python run_swap.py --synthetic --save_res --output_dir OUT          (seeded synthetic subjects, no assets)
    python run_swap.py --src_path A.jpg --tgt_path B.jpg --load_path G.pth --save_res ...

With real assets the subjects' SMPL vectors / backgrounds are read from `<image>.smpl.npy` / `<image>.bg.npy`."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from impersonator_amd import demo  # noqa: E402
from impersonator_amd.options.test_options import TestOptions  # noqa: E402
from impersonator_amd.utils import cv_utils, synthetic, util  # noqa: E402


def main():
    opt = TestOptions().parse()
    torch.cuda.set_device(0)
    if opt.synthetic:
        sw, smpl_a, img_a, bg_a = demo.build_synthetic_imitator(
            batch_size=1, model="swapper", opt=demo.default_opt(batch_size=1, front_warp=opt.front_warp,
                                                                 align_corners=opt.align_corners))
        smpl_b = demo.synthetic_smpls(8, seed=3)[5]
        img_b = synthetic.smooth_image(77, (1, 3, opt.image_size, opt.image_size))[0]
        sw.swap_setup(img_a, img_b, src_smpl=smpl_a, tgt_smpl=smpl_b, src_bg=bg_a, tgt_bg=bg_a)
        names = ("synthetic_a", "synthetic_b")
    else:
        from impersonator_amd.models.swapper import Swapper
        sw = Swapper(opt)
        load = lambda p, ext: np.load(p + ext) if os.path.exists(p + ext) else None
        sw.swap_setup(opt.src_path, opt.tgt_path, src_smpl=load(opt.src_path, '.smpl.npy'),
                      tgt_smpl=load(opt.tgt_path, '.smpl.npy'), src_bg=load(opt.src_path, '.bg.npy'),
                      tgt_bg=load(opt.tgt_path, '.bg.npy'))
        names = tuple(os.path.split(p)[-1].split('.')[0] for p in (opt.src_path, opt.tgt_path))
    preds = sw.swap(src_info=sw.src_info, tgt_info=sw.tsf_info, target_part=opt.swap_part)
    if opt.save_res:
        out_dir = util.mkdir(os.path.join(opt.output_dir, 'swappers'))
        path = os.path.join(out_dir, '{}->{}.png'.format(*names))
        cv_utils.save_cv2_img(preds[0].permute(1, 2, 0).cpu().numpy(), path, normalize=True)
        print('Saving results to {}'.format(path))


if __name__ == "__main__":
    main()
