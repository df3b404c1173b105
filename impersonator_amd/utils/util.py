"""Small host-side utilities with the reference's names (utils/util.py)."""
import pickle

import torch
import torch.nn.functional as F


def morph(src_bg_mask, ks, mode='erode', kernel=None, complement=False):
    """utils/util.py:73-89: erode / dilate a {0,1} mask with a ks x ks box (border padded with 1 / 0).

    CUDA tensors: liblwg's box-count kernel (lwg_morph, personalize.hip) -- `complement` (extension) returns 1 - result from the
    same launch.  CPU tensors (the host-side tests): the same counts from an integral image (two cumulative sums).  Counts
    are small integers, exact in fp32, so the result equals the reference's ones-kernel conv2d.  Runs once per source image
    (models/imitator.py:116,132)."""
    if kernel is not None:
        raise NotImplementedError("custom structuring elements are not used on the Imitator path")
    if mode not in ('erode', 'dilate'):
        mode = 'dilate'   # the reference's `else` branch
    if src_bg_mask.is_cuda:
        from .. import _lib
        m = src_bg_mask.float()
        n, c, h, w = m.shape
        if c != 1 or m.stride(3) != 1 or m.stride(2) != w:
            raise ValueError("morph: (n,1,H,W) mask with dense image planes expected, got %s strides %s" % (tuple(m.shape), m.stride()))
        out = torch.empty((n, 1, h, w), device=m.device, dtype=torch.float32)
        _lib.check(_lib.load().lwg_morph(_lib.ptr(m), n, h, w, m.stride(0) if n > 1 else h * w, ks, 0 if mode == 'erode' else 1,
                                         int(bool(complement)), _lib.ptr(out), _lib.stream_ptr()))
        return out
    n_ks = ks ** 2
    pad = ks // 2
    x = F.pad(src_bg_mask, [pad, pad, pad, pad], value=1.0 if mode == 'erode' else 0.0)
    ii = F.pad(x.cumsum(-1).cumsum(-2), [1, 0, 1, 0])
    box = ii[..., ks:, ks:] - ii[..., :-ks, ks:] - ii[..., ks:, :-ks] + ii[..., :-ks, :-ks]
    out = (box == n_ks).float() if mode == 'erode' else (box >= 1).float()
    return 1 - out if complement else out


def load_pickle_file(pkl_path):
    """utils/util.py:235-239."""
    with open(pkl_path, 'rb') as f:
        return pickle.load(f, encoding='latin1')


def mkdir(path):
    import os
    os.makedirs(path, exist_ok=True)
    return path
