"""Small host-side utilities with the reference's names (utils/util.py)."""
import pickle

import torch
import torch.nn.functional as F


def morph(src_bg_mask, ks, mode='erode', kernel=None):
    """utils/util.py:73-89: erode / dilate a {0,1} mask with a ks x ks box (border padded with 1 / 0).

    The reference convolves with a ones kernel and compares the count; the same counts are taken here from
    an integral image (two cumulative sums), which needs no convolution library.  Counts are small integers,
    exact in fp32, so the result is identical.  Runs once per source image (models/imitator.py:116,132)."""
    if kernel is not None:
        raise NotImplementedError("custom structuring elements are not used on the Imitator path")
    n_ks = ks ** 2
    pad = ks // 2
    x = F.pad(src_bg_mask, [pad, pad, pad, pad], value=1.0 if mode == 'erode' else 0.0)
    ii = F.pad(x.cumsum(-1).cumsum(-2), [1, 0, 1, 0])
    box = ii[..., ks:, ks:] - ii[..., :-ks, ks:] - ii[..., ks:, :-ks] + ii[..., :-ks, :-ks]
    if mode == 'erode':
        return (box == n_ks).float()
    return (box >= 1).float()


def load_pickle_file(pkl_path):
    """utils/util.py:235-239."""
    with open(pkl_path, 'rb') as f:
        return pickle.load(f, encoding='latin1')


def mkdir(path):
    import os
    os.makedirs(path, exist_ok=True)
    return path
