"""CPU check of the rasteriser's conservative pixel boxes (impersonator_amd/csrc/raster.hip, setup kernel): a pixel that
passes the reference's three float edge tests (rasterize_cuda_kernel.cu:132-134) must lie inside the box the tile
kernel scans for that face, otherwise the tiled result could differ from brute force.  The margin is
max(0.02, 0.025 * extent) pixels around the triangle's hull for faces with |2*area| > 1e-4 * extent^2 (everything else
gets the whole image).  Here the same float32 arithmetic is replayed in numpy on random triangles -- tiny, large and
as thin as the sliver rule admits -- and every pixel of a 3-pixel ring around the box is tested."""
import numpy as np
import pytest

F = np.float32


def _px(v, is_):
    return (F(0.5) * (v * F(is_) + F(is_) - F(1))).astype(F)


def _inside(v, xp, yp):
    """v (3,3) float32 (x, y, z per vertex); xp, yp float32 arrays -- .cu:132-134, float arithmetic, no contraction."""
    x0, y0, x1, y1, x2, y2 = v[0, 0], v[0, 1], v[1, 0], v[1, 1], v[2, 0], v[2, 1]
    a = ((yp - y0) * (x1 - x0)).astype(F) < ((xp - x0) * (y1 - y0)).astype(F)
    b = ((yp - y1) * (x2 - x1)).astype(F) < ((xp - x1) * (y2 - y1)).astype(F)
    c = ((yp - y2) * (x0 - x2)).astype(F) < ((xp - x2) * (y0 - y2)).astype(F)
    return ~(a | b | c)


@pytest.mark.parametrize("is_,seed", [(64, 0), (256, 1), (256, 2), (1024, 3)])
def test_pixels_passing_the_edge_tests_lie_in_the_scanned_box(is_, seed):
    rng = np.random.default_rng(seed)
    n = 6000
    c = rng.uniform(-1.0, 1.0, (n, 1, 2))
    size = np.exp(rng.uniform(np.log(1.0 / is_), np.log(0.6), (n, 1, 1)))
    xy = c + rng.normal(0, 1, (n, 3, 2)) * size
    thin = rng.random(n) < 0.5                      # squash half of them towards the sliver limit
    t = np.exp(rng.uniform(np.log(1e-4), np.log(3e-2), n))
    d = xy[:, 1] - xy[:, 0]
    nrm = np.stack([-d[:, 1], d[:, 0]], -1)
    mid = xy[:, 0] + d * rng.uniform(0, 1, (n, 1))
    xy[thin, 2] = (mid + nrm * t[:, None] * np.sign(rng.normal(size=(n, 1))))[thin]
    verts = np.concatenate([xy, np.ones((n, 3, 1))], -1).astype(F)
    checked = escaped = 0
    for v in verts:
        if (v[2, 1] - v[0, 1]) * (v[1, 0] - v[0, 0]) < (v[1, 1] - v[0, 1]) * (v[2, 0] - v[0, 0]):
            v = v[[0, 2, 1]]                        # front-facing orientation (the kernel culls the other one)
        px, py = _px(v[:, 0], is_), _px(v[:, 1], is_)
        det = px[2] * (py[0] - py[1]) + px[0] * (py[1] - py[2]) + px[1] * (py[2] - py[0])
        xmn, xmx, ymn, ymx = px.min(), px.max(), py.min(), py.max()
        ext = max(xmx - xmn, ymx - ymn)
        if not (abs(det) > F(1e-4) * ext * ext):
            continue                                # whole-image box: nothing to check
        m = max(F(0.02), F(ext * F(0.025)))
        x0, x1 = int(np.ceil(F(xmn - m))), int(np.floor(F(xmx + m)))
        y0, y1 = int(np.ceil(F(ymn - m))), int(np.floor(F(ymx + m)))
        ring = 3
        xs = np.arange(max(0, x0 - ring), min(is_ - 1, x1 + ring) + 1)
        ys = np.arange(max(0, y0 - ring), min(is_ - 1, y1 + ring) + 1)
        if not len(xs) or not len(ys):
            continue
        xp = ((2 * xs + 1 - is_).astype(F) / F(is_)).astype(F)[None, :]
        yp = ((2 * ys + 1 - is_).astype(F) / F(is_)).astype(F)[:, None]
        ins = _inside(v, xp, yp)
        outside_box = (xs[None, :] < x0) | (xs[None, :] > x1) | (ys[:, None] < y0) | (ys[:, None] > y1)
        escaped += int((ins & outside_box).sum())
        checked += 1
    assert checked > 2000
    assert escaped == 0, "%d pixels pass the edge tests outside their face's box" % escaped
