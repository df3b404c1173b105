"""Builds the synthetic configuration BASELINE.json names: random-init ImpersonatorGenerator + synthetic SMPL
(no downloaded assets).  Used by bench.py, run_imitator.py --synthetic, __graft_entry__.smoke() and the tests."""
import types

import numpy as np
import torch

from .utils import synthetic


def default_opt(batch_size=8, image_size=256, **over):
    opt = types.SimpleNamespace(
        image_size=image_size, tex_size=3, repeat_num=6, map_name='uv_seg', gen_name='impersonator',
        batch_size=batch_size, bg_model='ORIGINAL', bg_ks=13, ft_ks=3, only_vis=False, has_detector=False,
        front_warp=False, load_path='', load_epoch=-1, smpl_model='', hmr_model='', align_corners=False,
        bg_replace=False, swap_part='body', uv_mapping='', part_info='',
        is_train=False)
    for k, v in over.items():
        setattr(opt, k, v)
    return opt


def synthetic_smpls(num_frames, seed=0):
    """(num_frames, 85) SMPL vectors [cam(3), pose(72), shape(10)]: a smooth pose trajectory, per-frame cameras."""
    rng = np.random.default_rng(seed + 99)
    t = np.linspace(0, 2 * np.pi, num_frames, endpoint=False)[:, None]
    amp = rng.normal(0, 0.2, (1, 72))
    ph = rng.uniform(0, 2 * np.pi, (1, 72))
    pose = amp * np.sin(t * rng.integers(1, 4, (1, 72)) + ph)
    pose[:, :3] = 0.0
    pose[:, 1] = t[:, 0]                     # slow yaw of the root joint
    shape = np.repeat(rng.normal(0, 1.0, (1, 10)), num_frames, 0)
    cam = synthetic.cams(num_frames, seed=seed)
    return np.concatenate([cam, pose, shape], 1).astype(np.float32)


def synthetic_map_fn(map_name, rest, faces):
    """The face -> condition table of `--map_name` (utils/mesh.py:368-421) on the synthetic body mesh, background row included:
    'uv_seg' (3 channels, the default), 'par' (10 part labels + background = 11), 'binary' (the face index in binary, background -1)."""
    if map_name == 'uv_seg':
        return synthetic.uv_seg_map_fn(rest, faces)
    if map_name == 'par':
        return synthetic.part_map_fn(rest, faces)[0]
    if map_name == 'binary':
        from .utils import mesh
        return np.concatenate(mesh.binary_mapping(faces.shape[0]), axis=0)
    raise ValueError('map name error {}'.format(map_name))


def build_synthetic_imitator(batch_size=8, seed=0, image_size=256, affine="identity", opt=None, model="imitator"):
    """Returns (imitator, src_smpl (85,), src_img (3,H,W) in [-1,1], bg_img (3,H,W))."""
    from .models.imitator import Imitator
    from .models.swapper import Swapper
    from .models.viewer import Viewer
    from .networks.batch_smpl import HumanModelRecovery, synthetic_smpl_params
    from .networks.generator import ImpersonatorGenerator
    from .utils.nmr import SMPLRenderer

    opt = opt or default_opt(batch_size=batch_size, image_size=image_size)
    rest, faces = synthetic.body_mesh()
    map_fn = synthetic_map_fn(getattr(opt, 'map_name', 'uv_seg'), rest, faces)
    dim = 3 + map_fn.shape[1]                       # models/imitator.py:68-70: src_dim = tsf_dim = 3 + cond_nc
    render = SMPLRenderer(image_size=image_size, faces=faces, map_fn=map_fn,
                          front_map_fn=synthetic.front_map_fn(rest, faces), has_front=True,
                          align_corners=opt.align_corners)
    gen = ImpersonatorGenerator(bg_dim=4, src_dim=dim, tsf_dim=dim, repeat_num=opt.repeat_num, image_size=image_size,
                                max_batch=batch_size, align_corners=opt.align_corners)
    shapes = [(k, tuple(v.shape)) for k, v in gen.state_dict().items()]
    sd = synthetic.random_state_dict(shapes, seed=seed, affine=affine)
    gen.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    hmr = HumanModelRecovery(smpl_params=synthetic_smpl_params(seed))
    if model == "swapper":
        part_fn, part_faces = synthetic.part_map_fn(rest, faces)
        imitator = Swapper(opt, hmr=hmr, render=render, generator=gen, part_fn=part_fn, part_faces=part_faces)
    elif model == "viewer":
        imitator = Viewer(opt, hmr=hmr, render=render, generator=gen)
    else:
        imitator = Imitator(opt, hmr=hmr, render=render, generator=gen)
    src_smpl = synthetic_smpls(1, seed=seed + 1)[0]
    src_smpl[3:75] = 0.0
    src_img = synthetic.smooth_image(seed + 11, (1, 3, image_size, image_size))[0]
    bg_img = synthetic.smooth_image(seed + 12, (1, 3, image_size, image_size))[0]
    return imitator, src_smpl, src_img, bg_img
