"""CPU: the loaders for the reference's downloadable assets (smpl_model.pkl, mapper.txt, smpl_part_info.json,
front_facial.json, head.json -- README.md:48-68 of the reference; absent here) exercised on synthetic files in the same
formats, and -- where /root/reference exists -- compared with the reference's own loaders on those files."""
import json
import pickle

import numpy as np
import pytest
import scipy.sparse
import torch

from oracle import reference_loader


@pytest.fixture(scope="module")
def assets(tmp_path_factory):
    d = tmp_path_factory.mktemp("assets")
    rng = np.random.RandomState(0)
    nv, nf = 40, 60
    # --- mapper.txt: an .obj with per-corner texture and normal indices ("f v/vt/vn v/vt/vn v/vt/vn")
    nvt = 70
    vts = rng.rand(nvt, 2)
    faces_v = rng.randint(0, nv, (nf, 3))
    faces_vt = rng.randint(0, nvt, (nf, 3))
    with open(d / "mapper.txt", "w") as fp:
        for v in rng.randn(nv, 3):
            fp.write("v %f %f %f\n" % tuple(v))
        for t in vts:
            fp.write("vt %f %f\n" % tuple(t))
        for v in rng.randn(nv, 3):
            fp.write("vn %f %f %f\n" % tuple(v))
        for fv, ft in zip(faces_v, faces_vt):
            fp.write("f " + " ".join("%d/%d/%d" % (a + 1, b + 1, a + 1) for a, b in zip(fv, ft)) + "\n")
    # --- part / front / head json
    order = rng.permutation(nf)
    names = ["%02d_part" % i for i in range(5)]
    parts = {n: {"face": sorted(int(f) for f in order[i::5])} for i, n in enumerate(names)}
    json.dump(parts, open(d / "smpl_part_info.json", "w"))
    json.dump({"face": sorted(int(f) for f in order[:9])}, open(d / "front_facial.json", "w"))
    json.dump({"face": sorted(int(f) for f in order[:20])}, open(d / "head.json", "w"))
    # --- smpl_model.pkl: the dict layout of the SMPL release (sparse regressors)
    from impersonator_amd.networks.batch_smpl import synthetic_smpl_params
    p = synthetic_smpl_params(0)
    dd = dict(p)
    dd["J_regressor"] = scipy.sparse.csc_matrix(np.asarray(p["J_regressor"]))
    dd["cocoplus_regressor"] = scipy.sparse.csc_matrix(np.asarray(p["cocoplus_regressor"]))
    pickle.dump(dd, open(d / "smpl_model.pkl", "wb"), protocol=2)
    return d, p


def test_smpl_pickle_loader(assets):
    from impersonator_amd.networks.batch_smpl import SMPL
    d, p = assets
    a, b = SMPL(pkl_path=str(d / "smpl_model.pkl")), SMPL(params=p)
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    assert torch.equal(a.faces, b.faces) and (a.parents == b.parents).all()
    with pytest.raises(FileNotFoundError):
        SMPL(pkl_path=str(d / "missing.pkl"))


def test_smpl_state_dict_has_the_reference_keys_and_derived_buffers_follow_a_load(assets):
    """The SMPL module's state_dict is exactly the reference's six buffers (networks/batch_smpl.py:251-283): a strict load of a
    reference-keyed checkpoint works, a checkpoint of an earlier round (derived buffers as extra keys) still loads, and the derived
    buffers of the device kernels are rebuilt from what was loaded."""
    from impersonator_amd.networks.batch_smpl import SMPL, synthetic_smpl_params
    a, b = SMPL(params=synthetic_smpl_params(0)), SMPL(params=synthetic_smpl_params(1))
    assert sorted(a.state_dict()) == ["J_regressor", "joint_regressor", "posedirs", "shapedirs", "v_template", "weights"]
    assert not torch.equal(a.J_shapedirs_d, b.J_shapedirs_d)
    res = b.load_state_dict(a.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for name in ("J_template", "J_shapedirs", "J_template_d", "J_shapedirs_d"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert b.J_template_d.dtype == torch.float64 and b.J_template.dtype == torch.float32
    legacy = dict(a.state_dict(), J_template=a.J_template.clone(), J_shapedirs=a.J_shapedirs.clone(), parents_t=a.parents_t.clone(),
                  J_template_d=a.J_template_d.clone(), J_shapedirs_d=a.J_shapedirs_d.clone())
    c = SMPL(params=synthetic_smpl_params(2))
    c.load_state_dict(legacy, strict=True)       # the extra keys of an old checkpoint are tolerated, not loaded
    assert torch.equal(c.J_shapedirs_d, a.J_shapedirs_d)


def test_mapping_tables(assets):
    from impersonator_amd.utils import mesh
    d, _ = assets
    kw = dict(mapping_path=str(d / "mapper.txt"), part_info=str(d / "smpl_part_info.json"),
              front_info=str(d / "front_facial.json"), head_info=str(d / "head.json"))
    uv_seg = mesh.create_mapping("uv_seg", **kw)
    assert uv_seg.shape == (61, 3) and (uv_seg[-1] == [0, 0, 1]).all() and (uv_seg[:-1, 2] == 0).all()
    par = mesh.create_mapping("par", **kw)
    assert par.shape == (61, 6) and (par.sum(1) == 1).all() and par[-1, -1] == 1
    assert mesh.create_mapping("front", **kw).sum() == 9 and mesh.create_mapping("head", **kw).sum() == 20
    assert mesh.create_mapping("back", **kw).sum() == 11
    fb = mesh.create_mapping("par", fill_back=True, **kw)
    assert fb.shape == (121, 6) and (fb[:60] == fb[60:120]).all()
    binary = mesh.create_mapping("binary", **kw)      # 60 faces -> 6 bits, most significant first; background row of -1
    assert binary.shape == (61, 6) and (binary[-1] == -1).all() and (binary[37] == [1, 0, 0, 1, 0, 1]).all()
    ids = mesh.get_part_face_ids("par", mapping_path=kw["mapping_path"], part_info=kw["part_info"])
    assert sorted(sum(ids.values(), [])) == list(range(60))
    with pytest.raises(FileNotFoundError):
        mesh.create_mapping("uv_seg", mapping_path=str(d / "nope.txt"))


@pytest.mark.skipif(not reference_loader.available(), reason="/root/reference not present")
def test_mapping_tables_equal_the_reference_loaders(assets):
    import importlib
    from impersonator_amd.utils import mesh
    reference_loader.load()
    rmesh = importlib.import_module("utils.mesh")
    d, _ = assets
    kw = dict(mapping_path=str(d / "mapper.txt"), part_info=str(d / "smpl_part_info.json"),
              front_info=str(d / "front_facial.json"), head_info=str(d / "head.json"))
    for name in ("uv", "seg", "uv_seg", "par", "front", "head", "back", "binary"):
        for fill_back in (False, True):
            if name == "par" and fill_back:
                continue   # the reference drops fill_back on this branch (utils/mesh.py:404) and trips its own assert
            a = mesh.create_mapping(name, fill_back=fill_back, **kw)
            b = rmesh.create_mapping(name, fill_back=fill_back, **kw)
            assert a.shape == b.shape and np.array_equal(a, np.asarray(b, np.float32)), (name, fill_back)
