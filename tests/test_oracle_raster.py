"""Pins the CPU restatement of the rasteriser (oracle/raster_ref.c) against the reference's own
known-answer fixtures."""
import os

import numpy as np

from oracle import raster, torch_ref


def _teapot(golden_dir):
    z = np.load(os.path.join(golden_dir, "teapot_kat.npz"))
    sil = np.unpackbits(z["silhouette"])[:256 * 256].reshape(256, 256).astype(bool)
    return z["faces"], sil, z["depth_png"]


def test_teapot_silhouette_exact(golden_dir):
    # thirdparty/neural_renderer/tests/test_rasterize_silhouettes.py:16-35 (alpha = fim >= 0)
    faces, sil, _ = _teapot(golden_dir)
    fim, wim, depth = raster.rasterize_fim_wim(faces, 256, 0.1, 100.0)
    assert np.array_equal(fim[2] >= 0, sil)
    # to_minibatch fixture (tests/utils.py:11-27): the three all-zero samples render nothing
    for b in (0, 1, 3):
        assert (fim[b] == -1).all() and (wim[b] == 0).all() and (depth[b] == 100.0).all()


def test_teapot_depth(golden_dir):
    # tests/test_rasterize_depth.py:15-54
    faces, sil, depth_png = _teapot(golden_dir)
    _, _, depth = raster.rasterize_fim_wim(faces, 256, 0.1, 100.0)
    image = depth[2].copy()
    assert np.array_equal(image != image.max(), sil)
    image[image == image.max()] = image.min()
    image = (image - image.min()) / (image.max() - image.min())
    ref = depth_png.astype(np.float32) / 255.
    assert np.allclose(image, ref, atol=1e-2)
    assert np.abs(image - ref).max() < 4e-3  # 8-bit quantisation of the fixture


def test_weights_are_normalised_barycentrics(golden_dir):
    faces, _, _ = _teapot(golden_dir)
    fim, wim, _ = raster.rasterize_fim_wim(faces, 256, 0.1, 100.0)
    cov = fim >= 0
    w = wim[cov]
    assert (w >= 0).all() and (w <= 1).all()
    assert np.abs(w.sum(-1) - 1).max() < 1e-6
    assert (wim[~cov] == 0).all()


def test_depth_tie_lowest_face_wins():
    # H6: duplicate the same triangle; the strict '<' keeps the first (lowest index)
    tri = np.array([[[-0.5, -0.5, 1.0], [0.5, -0.5, 1.0], [0.0, 0.6, 1.0]]], np.float32)
    faces = np.stack([tri[0], tri[0], tri[0]])[None]
    fim, _, _ = raster.rasterize_fim_wim(faces, 64)
    assert set(np.unique(fim)) == {-1, 0}


def test_near_far_reject():
    tri = np.array([[-0.5, -0.5, 1.0], [0.5, -0.5, 1.0], [0.0, 0.6, 1.0]], np.float32)
    for z, visible in ((0.05, False), (0.09, False), (0.11, True), (99.0, True), (101.0, False), (150.0, False)):
        f = tri.copy()
        f[:, 2] = z
        fim, _, _ = raster.rasterize_fim_wim(f[None, None], 32)
        assert (fim >= 0).any() == visible, z


def test_backface_is_culled():
    tri = np.array([[-0.5, -0.5, 1.0], [0.5, -0.5, 1.0], [0.0, 0.6, 1.0]], np.float32)
    fim, _, _ = raster.rasterize_fim_wim(tri[None, None], 32)
    assert (fim >= 0).sum() > 50
    fim_b, _, _ = raster.rasterize_fim_wim(tri[::-1].copy()[None, None], 32)
    assert (fim_b == -1).all()


def test_look_at_kat(golden_dir):
    # thirdparty/neural_renderer/tests/test_look_at.py:9-25
    import torch
    z = np.load(os.path.join(golden_dir, "look_at_kat.npz"))
    v = torch.from_numpy(z["vertex"])[None, None, :]
    for eye, ans in zip(z["eyes"], z["answers"]):
        out = torch_ref.look_at(v, eye).squeeze().numpy()
        assert np.allclose(out, ans)


def test_renderer_eye_gives_identity_rotation():
    # the product path hard-wires look_at for SMPLRenderer's eye (utils/nmr.py:177): check it is a pure z shift
    import torch
    g = torch.Generator().manual_seed(0)
    v = torch.randn(2, 50, 3, generator=g)
    out = torch_ref.look_at(v, [0.0, 0.0, torch_ref.EYE_Z])
    exp = v.clone()
    exp[:, :, 2] -= np.float32(torch_ref.EYE_Z)
    assert torch.equal(out, exp)
