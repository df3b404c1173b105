#!/usr/bin/env python
"""Motion imitation entry point -- command line of the reference's run_imitator.py (run_imitator.py:214-241).

    python run_imitator.py --src_path S.jpg --tgt_path DIR --load_path G.pth --output_dir OUT   (real assets)
    python run_imitator.py --synthetic --num_frames 64 --output_dir OUT                          (no assets)
    python -m torch.distributed.run --nproc-per-node 8 ... run_imitator.py --synthetic ...       (8 GPUs, frame-sharded)

With real assets the SMPL parameters of source and targets must be supplied as `<image>.smpl.npy` files next to the
images (the HMR image regressor that the reference uses to estimate them is outside this build's scope).
"""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from impersonator_amd import demo, sharding  # noqa: E402
from impersonator_amd.options.test_options import TestOptions  # noqa: E402
from impersonator_amd.utils import cv_utils, util  # noqa: E402


def scan_tgt_paths(tgt_path, itv=20):
    """run_imitator.py:58-66."""
    if os.path.isdir(tgt_path):
        paths = sorted(glob.glob(os.path.join(tgt_path, '*')))
        paths = [p for p in paths if not p.endswith('.npy')][::itv]
    else:
        paths = [tgt_path]
    return paths


def _smpl_of(path):
    f = path + '.smpl.npy'
    if not os.path.exists(f):
        raise FileNotFoundError("%s: SMPL vector (85,) expected next to the image" % f)
    return np.load(f).astype(np.float32).reshape(85)


def main():
    opt = TestOptions().parse()
    rank, local_rank, world = sharding.init_process_group()
    if local_rank >= torch.cuda.device_count() and os.environ.get("LWG_DIST_BACKEND") == "gloo":
        local_rank %= torch.cuda.device_count()   # test hook: N gloo ranks sharing the visible GPU(s) (RCCL wants one each)
    torch.cuda.set_device(local_rank)
    if opt.synthetic:
        imitator, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(
            batch_size=opt.batch_size, image_size=opt.image_size,
            opt=demo.default_opt(batch_size=opt.batch_size, image_size=opt.image_size, front_warp=opt.front_warp,
                                 only_vis=opt.only_vis, align_corners=opt.align_corners,
                                 map_name=getattr(opt, 'map_name', 'uv_seg')))
        imitator.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
        tgt_smpls = demo.synthetic_smpls(opt.num_frames, seed=0)
        tgt_paths = None
    else:
        from impersonator_amd.models.imitator import Imitator
        imitator = Imitator(opt)
        bg = None
        bg_file = opt.src_path + '.bg.npy'
        if os.path.exists(bg_file):
            bg = np.load(bg_file)
        imitator.personalize(opt.src_path, src_smpl=_smpl_of(opt.src_path), bg_img=bg)
        tgt_paths = scan_tgt_paths(opt.tgt_path, itv=1)
        tgt_smpls = np.stack([_smpl_of(p) for p in tgt_paths])

    # frame sharding: every rank imitates its own blocks of `batch_size` frames (first_cam = frame 0's camera everywhere)
    outs = sharding.imitate_sharded(imitator, tgt_smpls, opt.batch_size, opt.cam_strategy, rank, world)
    if rank == 0 and opt.output_dir:
        out_dir = util.mkdir(opt.output_dir)
        for t, img in enumerate(outs):
            # the reference names its outputs after the target files (run_imitator.py:232, 'pred_%.8d.jpg' in
            # inference_by_smpls); the synthetic sequence has no files and is written losslessly, so that what lands on
            # disk IS the uint8 truncation of utils/cv_utils.py:31-33 (hazard H11) and can be compared as such
            name = os.path.split(tgt_paths[t])[-1] if tgt_paths else '%.8d.png' % t
            cv_utils.save_cv2_img(img, os.path.join(out_dir, 'pred_' + name), normalize=True)
        print('wrote %d frames to %s' % (len(outs), out_dir))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
