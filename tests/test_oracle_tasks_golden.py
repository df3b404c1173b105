"""CPU: the oracle's restatement of the Swapper / Viewer glue (oracle/torch_ref.py: swapper_personalize, swapper_swap,
viewer_view) reproduces tests/golden/tasks_golden.npz -- outputs of the reference's OWN `Swapper.personalize`, `Swapper.swap`,
`calculate_trans`, `forward` (models/swapper.py:99-271) and `Viewer.view`, `rotate_trans`, `forward`
(models/viewer.py:262-311) run unbound (tests/golden/make_golden.py::make_tasks).  The GPU tests (test_gpu_tasks.py) compare
the product with the same file."""
import numpy as np
import torch

from oracle import torch_ref
from tests import helpers


def test_oracle_swapper_and_viewer_reproduce_the_reference():
    g = helpers.golden("tasks_golden.npz")
    sc = helpers.task_scene()
    sd = torch_ref.state_dict_from_numpy(helpers.generator_state_dict(seed=0, affine="random"))
    t = torch.from_numpy
    faces, map_fn, part_fn = t(sc["faces"]), t(sc["map_fn"]), t(sc["part_fn"])
    with torch.no_grad():
        A = torch_ref.swapper_personalize(sd, t(sc["img_a"]), t(sc["cam_a"]), t(sc["verts_a"]), faces, map_fn, part_fn)
        B = torch_ref.swapper_personalize(sd, t(sc["img_b"]), t(sc["cam_b"]), t(sc["verts_b"]), faces, map_fn, part_fn)
        assert np.array_equal(A["fim"].numpy(), g["fim_a"]) and np.array_equal(B["fim"].numpy(), g["fim_b"])
        assert np.array_equal(A["part"].argmax(1).numpy(), g["part_a"])
        assert np.abs(A["bg"].numpy()[:, :, ::4, ::4] - g["bg_a_sub"]).max() < 2e-5
        assert np.abs(B["bg"].numpy()[:, :, ::4, ::4] - g["bg_b_sub"]).max() < 2e-5
        sw = torch_ref.swapper_swap(sd, A, B, sc["part_faces"])
        assert np.array_equal(np.packbits(sw["left_mask"].numpy()), g["left_mask"])
        assert np.array_equal(sw["T11"].numpy(), g["T11"])
        assert np.abs(sw["T21"].numpy() - g["T21"]).max() <= 1e-6
        assert np.abs(sw["tsf_inputs"].numpy()[:, :, ::2, ::2] - g["tsf_inputs_sub"]).max() <= 1e-5
        assert np.abs(sw["preds"].numpy() - g["swap_preds"]).max() < 2e-5
        for i, (rt, tr, replace) in enumerate(sc["views"]):
            _, pv = torch_ref.viewer_view(sd, A, t(g["view%d_mesh" % i]), t(sc["cam_a"]), faces, map_fn, bg_replace=replace)
            assert np.abs(pv.numpy() - g["view%d_preds" % i]).max() < 2e-5, i
