"""Shared fixtures for the parity tests: the seeded synthetic scene of tests/golden/make_golden.py,
rebuilt WITHOUT the reference (it does not exist on the GPU box)."""
import os

import numpy as np
import torch

from impersonator_amd.utils import synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def has_gpu():
    return torch.cuda.is_available()


def scene():
    """Numpy/torch CPU inputs identical to make_golden.make_frame()."""
    rest, faces = synthetic.body_mesh()
    s = dict(rest=rest, faces=faces, map_fn=synthetic.uv_seg_map_fn(rest, faces))
    s["src_cam"] = synthetic.cams(1, seed=100)
    s["src_verts"] = rest[None].copy()
    s["src_img"] = synthetic.smooth_image(11)
    s["bg_img"] = synthetic.smooth_image(12)
    s["tgt_verts"] = np.stack([synthetic.motion_verts(rest, t) for t in (3, 200)])
    s["tgt_cam"] = synthetic.cams(2, seed=5)
    return s


def generator_state_dict(seed=0, affine="random"):
    """Seeded weights keyed like ImpersonatorGenerator.state_dict() (numpy)."""
    from impersonator_amd.networks.generator import ImpersonatorGenerator
    G = ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6, repeat_num=6)
    shapes = [(k, tuple(v.shape)) for k, v in G.state_dict().items()]
    return synthetic.random_state_dict(shapes, seed=seed, affine=affine)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def t(x, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(x)).to(device)


def maxdiff(a, b):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.from_numpy(np.asarray(a)).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.from_numpy(np.asarray(b)).double()
    d = (a - b).abs()
    idx = int(d.argmax())
    return float(d.max()), np.unravel_index(idx, tuple(d.shape))


def discriminator_state_dict(seed=0, input_nc=6, ndf=64, n_layers=4):
    """Seeded PatchDiscriminator parameters (conv weights ~ N(0, 0.02) as networks.py:57-58, biases ~ U(-0.1, 0.1))."""
    import torch
    g = torch.Generator().manual_seed(seed)
    sd, idx, cin, mult = {}, 0, input_nc, 1
    chans = [ndf] + [ndf * min(2 ** n, 8) for n in range(1, n_layers)] + [ndf * min(2 ** n_layers, 8), 1]
    for l, cout in enumerate(chans):
        sd["model.%d.weight" % idx] = torch.randn(cout, cin, 4, 4, generator=g) * 0.02
        sd["model.%d.bias" % idx] = torch.rand(cout, generator=g) * 0.2 - 0.1
        idx += 2 if l == 0 else 3
        cin = cout
    return sd


def vgg19_state_dict(seed=0):
    """Random VGG19 conv stack in torchvision's naming (features.N.weight / .bias) up to relu5_1 -- the real weights are a
    download; He-scaled so that activations keep their magnitude through the 13 layers."""
    import torch
    g = torch.Generator().manual_seed(seed)
    sd, cin = {}, 3
    for idx, cout in [(0, 64), (2, 64), (5, 128), (7, 128), (10, 256), (12, 256), (14, 256), (16, 256), (19, 512), (21, 512),
                      (23, 512), (25, 512), (28, 512)]:
        sd["features.%d.weight" % idx] = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
        sd["features.%d.bias" % idx] = torch.randn(cout, generator=g) * 0.05
        cin = cout
    return sd


def sphere20a_state_dict(seed=0):
    """Random Sphere20a (networks/facenet.py:200-262) in its own state_dict naming -- the real file
    (sphere20a_20171020.pth) is a download; He-scaled convs, PReLU slopes around 0.25."""
    import torch
    g = torch.Generator().manual_seed(seed)
    sd, cin = {}, 3
    for st, c, units in (("1", 64, 1), ("2", 128, 2), ("3", 256, 4), ("4", 512, 1)):
        for j in range(1, 2 + 2 * units):
            name = "%s_%d" % (st, j)
            ci = cin if j == 1 else c
            sd["conv%s.weight" % name] = torch.randn(c, ci, 3, 3, generator=g) * (1.0 / (9 * ci)) ** 0.5
            sd["conv%s.bias" % name] = torch.randn(c, generator=g) * 0.05
            sd["relu%s.weight" % name] = 0.25 + 0.1 * torch.rand(c, generator=g)
        cin = c
    sd["fc5.weight"] = torch.randn(512, 512 * 7 * 6, generator=g) * (1.0 / (512 * 42)) ** 0.5
    sd["fc5.bias"] = torch.randn(512, generator=g) * 0.05
    return sd


def train_batch(seed=0, n=2, size=64, bg_both=False):
    """Seeded stand-in for what the reference's BodyRecoveryFlow hands the trainer (impersonator_trainer.py:300-319);
    bg_both: the background input carries the source's and the target's (2n images, :333-337)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g) * 2 - 1
    T = torch.rand(n, size, size, 2, generator=g) * 2.4 - 1.2
    T[0, size // 4:size // 2, size // 8:size // 3] = -2
    if bg_both:
        return dict(input_G_src=r(n, 6, size, size), input_G_tsf=r(n, 6, size, size), T=T, real_src=r(n, 3, size, size),
                    real_tsf=r(n, 3, size, size), bg_mask=(torch.rand(2 * n, 1, size, size, generator=g) > 0.5).float(),
                    input_G_bg=r(2 * n, 4, size, size))
    return dict(input_G_bg=r(n, 4, size, size), input_G_src=r(n, 6, size, size), input_G_tsf=r(n, 6, size, size), T=T,
                real_src=r(n, 3, size, size), real_tsf=r(n, 3, size, size),
                bg_mask=(torch.rand(2 * n, 1, size, size, generator=g) > 0.5).float())


class FixedHMR(object):
    """`hmr` stand-in: get_details() hands out prepared (cam, verts) in call order (the SMPL regressor / LBS are not
    what these fixtures pin)."""

    def __init__(self, infos):
        self.infos = list(infos)

    def get_details(self, smpl):
        cam, verts = self.infos.pop(0)
        return dict(theta=smpl, cam=cam, pose=smpl[:, 3:75], shape=smpl[:, 75:], verts=verts)

    def cuda(self):
        return self


def task_scene():
    """Inputs of tests/golden/tasks_golden.npz (numpy): two subjects (rest pose / frame 200 of the synthetic motion, own
    cameras and images), the synthetic 10-part partition, two views."""
    rest, faces = synthetic.body_mesh()
    part_fn, part_faces = synthetic.part_map_fn(rest, faces)
    return dict(rest=rest, faces=faces, map_fn=synthetic.uv_seg_map_fn(rest, faces), part_fn=part_fn, part_faces=part_faces,
                cam_a=synthetic.cams(1, seed=100), verts_a=rest[None].copy(), img_a=synthetic.smooth_image(11),
                cam_b=synthetic.cams(1, seed=101), verts_b=synthetic.motion_verts(rest, 200)[None].copy(),
                img_b=synthetic.smooth_image(77),
                views=[((0.0, 0.6, 0.0), (0.0, 0.0, 0.0), False), ((0.2, -1.1, 0.1), (0.02, 0.0, 0.0), True)])


# ---------------------------------------------------------------------------------------------------------------------
# tests/golden/imitator_golden.npz: the reference's own Imitator methods run unbound (make_golden.py::make_imitator)
IMITATOR_VARIANTS = {
    # name: image size, --only_vis, --bg_model (ORIGINAL = the generator's BGNet, else InpaintSANet), --front_warp, --cam_strategy
    "main": dict(size=256, only_vis=False, bg_model="ORIGINAL", front_warp=False, cam_strategy="smooth"),
    "vis_inpaint_front_source": dict(size=128, only_vis=True, bg_model="deepfillv2", front_warp=True, cam_strategy="source"),
    "vis_bgnet_front_copy": dict(size=128, only_vis=True, bg_model="ORIGINAL", front_warp=True, cam_strategy="copy"),
    "inpaint_smooth": dict(size=128, only_vis=False, bg_model="deepfillv2", front_warp=False, cam_strategy="smooth"),
}


def imitator_scene(size):
    """Inputs of imitator_golden.npz (numpy) at one image size: synthetic SMPL model + mesh tables, one source (rest pose,
    own camera and betas), four target SMPL vectors far apart in the synthetic motion, seeded images."""
    from impersonator_amd import demo
    from impersonator_amd.networks.batch_smpl import synthetic_smpl_params
    rest, faces = synthetic.body_mesh()
    src_smpl = demo.synthetic_smpls(1, seed=1)[0]
    src_smpl[3:75] = 0.0
    return dict(rest=rest, faces=faces, map_fn=synthetic.uv_seg_map_fn(rest, faces), front_map_fn=synthetic.front_map_fn(rest, faces),
                smpl_params=synthetic_smpl_params(0), src_smpl=src_smpl, tgt_smpls=demo.synthetic_smpls(64, seed=0)[::16].copy(),
                src_img=synthetic.smooth_image(11, (1, 3, size, size)), size=size)


def inpaintor_state_dict(seed=1):
    """Seeded InpaintSANet(c_dim=4) weights (numpy), keyed like the reference's state_dict."""
    from impersonator_amd.networks.inpaintor import InpaintSANet
    shapes = [(k, tuple(v.shape)) for k, v in InpaintSANet(c_dim=4).state_dict().items()]
    return synthetic.random_inpaintor_state_dict(shapes, seed)


def tensor_stat(x):
    x = torch.as_tensor(x).double()
    return np.array([x.mean().item(), x.abs().mean().item(), (x * x).mean().item()])
