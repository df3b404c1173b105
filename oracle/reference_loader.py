"""oracle/reference_loader.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Imports the *real* reference (svip-lab/impersonator, read-only at /root/reference) on CPU so that
(a) tests can validate the CPU restatement in oracle/torch_ref.py against it, and
(b) tests/golden/make_golden.py can generate golden vectors from it.

The reference needs five modules that are not installed here (ipdb, h5py, cv2, torchvision,
neural_renderer); they are replaced by empty stubs.  The stub `neural_renderer` is populated with
the reference's own pure-Python functions loaded by file path, and its CUDA rasteriser entry point
`rasterize_face_index_map_and_weight_map` (rasterize.py:543-571) is routed to the C restatement in
oracle/raster_ref.c, so that the reference's own `SMPLRenderer.render_fim_wim` body
(utils/nmr.py:263-278) runs unmodified.

/root/reference does not exist on the GPU box: nothing that runs there may import this module
without checking `available()` first.
"""
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("LWG_REFERENCE_ROOT", "/root/reference")
_NR_DIR = os.path.join(REFERENCE_ROOT, "thirdparty", "neural_renderer", "neural_renderer")

_loaded = None


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "networks")) and os.path.isdir(_NR_DIR)


def _stub(name):
    mod = types.ModuleType(name)
    mod.__dict__["__path__"] = []
    sys.modules[name] = mod
    return mod


def _load_by_path(mod_name, path):
    spec = importlib.util.spec_from_file_location(mod_name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load():
    """Returns a namespace with the reference modules: generator, nmr, imitator, nr (stub), ..."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)

    import torch
    from . import raster as oracle_raster

    for name in ("ipdb", "h5py", "cv2"):
        if name not in sys.modules:
            _stub(name)
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.models = _stub("torchvision.models")
        tv.transforms = _stub("torchvision.transforms")
        tv.transforms.functional = _stub("torchvision.transforms.functional")
        tv.ops = _stub("torchvision.ops")

    nr = _stub("neural_renderer")
    for fn in ("look_at", "vertices_to_faces", "perspective", "look"):
        mod = _load_by_path("_ref_nr_" + fn, os.path.join(_NR_DIR, fn + ".py"))
        setattr(nr, fn, getattr(mod, fn))

    def rasterize_face_index_map_and_weight_map(faces, image_size=256, anti_aliasing=True,
                                                near=0.1, far=100, eps=1e-4):
        # defaults of rasterize.py:8-13; anti_aliasing=True would super-sample 2x (rasterize.py:309-313)
        if anti_aliasing:
            raise NotImplementedError("the hot path always passes anti_aliasing=False (nmr.py:277)")
        fim, wim, _ = oracle_raster.rasterize_fim_wim(faces.detach().cpu().numpy(), image_size, near, far)
        return torch.from_numpy(fim), torch.from_numpy(wim)

    nr.rasterize_face_index_map_and_weight_map = rasterize_face_index_map_and_weight_map

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    ns = types.SimpleNamespace()
    ns.nr = nr
    ns.generator = importlib.import_module("networks.generator")
    ns.networks = importlib.import_module("networks.networks")
    ns.inpaintor = importlib.import_module("networks.inpaintor")
    ns.discriminator = importlib.import_module("networks.discriminator")
    ns.trainer = importlib.import_module("models.impersonator_trainer")
    ns.nmr = importlib.import_module("utils.nmr")
    ns.util = importlib.import_module("utils.util")
    ns.batch_smpl = importlib.import_module("networks.batch_smpl")
    ns.imitator = importlib.import_module("models.imitator")
    _loaded = ns
    return ns
