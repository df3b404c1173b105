"""Deterministic synthetic stand-ins for the assets the reference downloads.

The reference needs `smpl_model.pkl`, `smpl_faces.npy`, `mapper.txt` and trained checkpoints
(README.md:48-68); none of them ship with it and there is no network here.  BASELINE.json's configs
are therefore quoted on "random-init ResUnetGenerator + synthetic SMPL".  Everything below is
generated from `numpy.random.default_rng(seed)` (PCG64: bit-reproducible across machines) so that
the committed golden vectors, the parity tests and bench.py all see identical inputs.

Counts match SMPL exactly: 6890 vertices / 13776 faces (a closed genus-0 UV sphere with
84 rings x 82 segments + 2 poles).
"""
import math

import numpy as np

SMPL_NUM_VERTS = 6890
SMPL_NUM_FACES = 13776
_RINGS, _SEGS = 84, 82
BODY_RADII = (0.25, 0.85, 0.18)


def uv_sphere(rings=_RINGS, segs=_SEGS):
    """Unit sphere: (rings*segs+2, 3) float32 vertices, (2*segs*rings, 3) int32 faces.

    Winding is chosen so that, after SMPLRenderer's y-flip (utils/nmr.py:271) and with the camera
    on the -z side (nmr.py:177), the faces whose outward normal points at the camera survive the
    rasteriser's back-face test (rasterize_cuda_kernel.cu:57)."""
    v = [(0.0, 1.0, 0.0)]
    for r in range(rings):
        phi = math.pi * (r + 1) / (rings + 1)
        for s in range(segs):
            th = 2.0 * math.pi * s / segs
            v.append((math.sin(phi) * math.cos(th), math.cos(phi), math.sin(phi) * math.sin(th)))
    v.append((0.0, -1.0, 0.0))
    verts = np.asarray(v, np.float64)

    def vid(r, s):
        return 1 + r * segs + (s % segs)

    south = verts.shape[0] - 1
    f = []
    for s in range(segs):
        f.append((0, vid(0, s + 1), vid(0, s)))
    for r in range(rings - 1):
        for s in range(segs):
            a, b, c, d = vid(r, s), vid(r + 1, s), vid(r + 1, s + 1), vid(r, s + 1)
            f.append((a, c, b))
            f.append((a, d, c))
    for s in range(segs):
        f.append((south, vid(rings - 1, s), vid(rings - 1, s + 1)))
    faces = np.asarray(f, np.int32)
    return verts.astype(np.float32), faces


def body_mesh():
    """Rest-pose stand-in for the SMPL template: an ellipsoid that fills the frame like a person."""
    verts, faces = uv_sphere()
    assert verts.shape[0] == SMPL_NUM_VERTS and faces.shape[0] == SMPL_NUM_FACES
    return verts * np.asarray(BODY_RADII, np.float32)[None, :], faces


def motion_verts(rest, t, num_frames=1024, seed=0):
    """Frame `t` of a smooth synthetic motion: low-frequency surface waves + a slow yaw.
    rest: (nv,3) float32.  Returns (nv,3) float32."""
    rng = np.random.default_rng(seed)
    amp = rng.uniform(0.01, 0.02, size=(4, 3))
    freq = rng.uniform(1.0, 4.0, size=(4, 3))
    phase = rng.uniform(0.0, 2 * math.pi, size=(4,))
    x = rest.astype(np.float64)
    disp = np.zeros_like(x)
    tt = 2.0 * math.pi * t / num_frames
    for k in range(4):
        arg = (x * freq[k][None, :]).sum(1) * 3.0 + phase[k] + (k + 1) * tt
        disp += np.sin(arg)[:, None] * amp[k][None, :]
    y = x + disp
    yaw = 2.0 * math.pi * t / num_frames
    c, s = math.cos(yaw), math.sin(yaw)
    rot = np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])
    return (y @ rot.T).astype(np.float32)


def cams(n, seed=0):
    """(n,3) [scale, tx, ty] weak-perspective cameras as HMR produces (networks/hmr.py:302-330)."""
    rng = np.random.default_rng(seed + 7919)
    out = np.empty((n, 3), np.float32)
    out[:, 0] = rng.uniform(0.8, 1.1, n)
    out[:, 1:] = rng.uniform(-0.1, 0.1, (n, 2))
    return out


def uv_seg_map_fn(rest, faces):
    """(nf+1, 3) face -> condition table shaped like `create_mapping('uv_seg', contain_bg=True)`
    (utils/mesh.py:368-421): one (u, v, 0) row per face and a final background row (0, 0, 1)
    that `fim == -1` reaches through negative indexing (utils/nmr.py:336)."""
    bary = rest[faces.astype(np.int64)].mean(1).astype(np.float64)
    n = bary / np.asarray(BODY_RADII, np.float64)[None, :]
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    u = (np.arctan2(n[:, 2], n[:, 0]) / (2 * math.pi)) % 1.0
    v = np.arccos(np.clip(n[:, 1], -1, 1)) / math.pi
    tab = np.zeros((faces.shape[0] + 1, 3), np.float32)
    tab[:-1, 0] = u
    tab[:-1, 1] = v
    tab[-1, 2] = 1.0
    return tab


def front_map_fn(rest, faces):
    """(nf+1, 1) stand-in for `create_mapping('front', ...)`: 1 for faces of the upper-front cap."""
    bary = rest[faces.astype(np.int64)].mean(1)
    tab = np.zeros((faces.shape[0] + 1, 1), np.float32)
    tab[:-1, 0] = ((bary[:, 1] > 0.55) & (bary[:, 2] < 0)).astype(np.float32)
    return tab


def part_map_fn(rest, faces, num_parts=10):
    """Stand-in for `create_mapping('par', ...)` / `get_part_face_ids('par')` (utils/mesh.py:247-345): `num_parts`
    contiguous ring bands along the body axis, part 0 at the top (the 'head').  Returns the (nf+1, num_parts+1)
    one-hot table with its background row and the list of face-id lists."""
    bary_y = rest[faces.astype(np.int64)].mean(1)[:, 1]
    order = np.argsort(-bary_y, kind="stable")
    nf = faces.shape[0]
    part_of = np.empty(nf, np.int64)
    part_of[order] = (np.arange(nf) * num_parts) // nf
    tab = np.zeros((nf + 1, num_parts + 1), np.float32)
    tab[np.arange(nf), part_of] = 1.0
    tab[-1, -1] = 1.0
    return tab, [np.nonzero(part_of == i)[0].tolist() for i in range(num_parts)]


def image(seed, shape=(1, 3, 256, 256)):
    """U(-1,1) image-like tensor (source image / background stand-in)."""
    rng = np.random.default_rng(seed)
    return rng.uniform(-1.0, 1.0, size=shape).astype(np.float32)


def smooth_image(seed, shape=(1, 3, 256, 256)):
    """Low-frequency image in [-1,1]: a more image-like source than white noise."""
    rng = np.random.default_rng(seed)
    n, c, h, w = shape
    yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing="ij")
    out = np.zeros(shape, np.float64)
    for i in range(n):
        for j in range(c):
            for _ in range(6):
                fx, fy = rng.uniform(0.5, 6.0, 2)
                ph = rng.uniform(0, 2 * math.pi)
                out[i, j] += rng.uniform(0.1, 0.4) * np.sin(2 * math.pi * (fx * xx + fy * yy) + ph)
    return np.clip(out, -1, 1).astype(np.float32)


def random_state_dict(shapes, seed=0, affine="random"):
    """Seeded weights for a list of (key, shape) pairs in state_dict order.

    conv / conv-transpose weights (4-D): N(0, 0.02) -- `NetworkBase.init_weights` (networks/networks.py:54-65).
    InstanceNorm affine (1-D): gamma=1, beta=0 when affine == 'identity' (what init_weights leaves,
    since 'InstanceNorm2d' does not match 'BatchNorm2d'); gamma~U(0.5,1.5), beta~N(0,0.1) when
    affine == 'random' (exercises the affine path in parity tests)."""
    rng = np.random.default_rng(seed)
    out = {}
    for key, shape in shapes:
        shape = tuple(int(s) for s in shape)
        if len(shape) == 4:
            out[key] = (rng.standard_normal(shape, dtype=np.float32) * np.float32(0.02))
        elif key.endswith(".weight"):
            out[key] = (np.ones(shape, np.float32) if affine == "identity"
                        else rng.uniform(0.5, 1.5, shape).astype(np.float32))
        elif key.endswith(".bias"):
            out[key] = (np.zeros(shape, np.float32) if affine == "identity"
                        else (rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1)))
        else:
            raise ValueError("unexpected parameter %s %s" % (key, shape))
    return out


def random_inpaintor_state_dict(shapes, seed=0):
    """Seeded InpaintSANet weights for (key, shape) pairs in state_dict order: kaiming-normal convs
    (networks/inpaintor.py:30-32), small biases, non-trivial BatchNorm statistics and attention gain so that every
    term of the gated layers is exercised."""
    rng = np.random.default_rng(seed + 31337)
    out = {}
    for key, shape in shapes:
        shape = tuple(int(s) for s in shape)
        if key.endswith("num_batches_tracked"):
            out[key] = np.zeros(shape, np.int64)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            out[key] = (rng.standard_normal(shape, dtype=np.float32) * np.float32(math.sqrt(2.0 / fan_in)))
        elif key.endswith("running_var"):
            out[key] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif key.endswith("running_mean"):
            out[key] = (rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1))
        elif key.endswith("gamma"):
            out[key] = np.full(shape, 0.7, np.float32)
        elif key.endswith("batch_norm2d.weight"):
            out[key] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        else:   # conv biases, BatchNorm bias
            out[key] = (rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1))
    return out
