"""GPU: the reference-side binding INTEGRATION.md shows a maintainer -- executed.  The python blocks of the document's header
(the ctypes stub), section 1 (the body of `rasterize_face_index_map_and_weight_map`, thirdparty/neural_renderer/
neural_renderer/rasterize.py:543-571) and section 3 (`ImpersonatorGenerator.encode_src` / `inference`, networks/generator.py:
213-214, 277-301) are extracted from the file AS WRITTEN, exec'd against the built liblwg.so, and their results compared with
the CPU oracle.  The only edit is the library's path (the document says "liblwg.so": a maintainer puts it on the loader path)."""
import os
import re

import numpy as np
import pytest
import torch

from impersonator_amd import _lib
from oracle import raster as oracle_raster
from oracle import torch_ref
from tests import helpers

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _blocks():
    """{section title: [python code blocks]} of INTEGRATION.md, in document order."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    out, title = {}, "header"
    for m in re.finditer(r"^(#+ [^\n]*)$|^```python\n(.*?)^```", text, re.M | re.S):
        if m.group(1):
            title = m.group(1).lstrip("# ").strip()
        else:
            out.setdefault(title, []).append(m.group(2))
    return out


@pytest.fixture(scope="module")
def binding():
    blocks = _blocks()
    stub = next(v for k, v in blocks.items() if k.startswith("INTEGRATION"))[0]
    assert 'ctypes.CDLL("liblwg.so")' in stub
    ns = {}
    exec(compile(stub.replace('"liblwg.so"', repr(_lib.LIB_PATH)), "INTEGRATION.md#binding", "exec"), ns)
    return ns, blocks


def test_section_1_rasteriser_binding(binding):
    ns, blocks = binding
    code = next(b for k, v in blocks.items() if k.startswith("1.") for b in v)
    exec(compile(code, "INTEGRATION.md#1", "exec"), ns)
    fn = ns["rasterize_face_index_map_and_weight_map"]
    s = helpers.scene()
    with torch.no_grad():
        f2v, ofim, owim = torch_ref.render_fim_wim(helpers.t(np.concatenate([s["src_cam"], s["tgt_cam"]])),
                                                   helpers.t(np.concatenate([s["src_verts"], s["tgt_verts"]])), helpers.t(s["faces"]))
    fim, wim = fn(f2v.cuda(), image_size=256, anti_aliasing=False)          # the reference's call (utils/nmr.py:277)
    assert fim.dtype == torch.int32 and tuple(fim.shape) == (3, 256, 256) and tuple(wim.shape) == (3, 256, 256, 3)
    assert torch.equal(fim.cpu(), ofim) and torch.equal(wim.cpu(), owim)
    # the reference's own known-answer fixture through the same function (teapot silhouette, exact)
    g = helpers.golden("teapot_kat.npz")
    kfim, _ = fn(torch.from_numpy(g["faces"]).cuda(), image_size=256, anti_aliasing=False)
    sil = np.unpackbits(g["silhouette"]).reshape(256, 256).astype(bool)
    assert np.array_equal(kfim[2].cpu().numpy() >= 0, sil)
    with pytest.raises(AssertionError):
        fn(f2v.cuda(), anti_aliasing=True)
    with pytest.raises(RuntimeError):                                        # AT_CHECK -> RuntimeError (rasterize_cuda.cpp:66-68)
        ns["_ok"](ns["_lwg"].lwg_rasterize_fim_wim(None, 1, 1, 8, 0, 0, None, None, None, None, 0, None))


def test_section_3_generator_binding(binding):
    ns, blocks = binding
    code = next(b for k, v in blocks.items() if k.startswith("3.") for b in v)
    sd = torch_ref.state_dict_from_numpy(helpers.generator_state_dict(seed=0, affine="random"))

    class NetworkBase(torch.nn.Module):          # stand-in for networks/networks.py::NetworkBase: the checkpoint's tensors
        repeat_num = 6

        def state_dict(self):
            return sd

    ns["NetworkBase"] = NetworkBase
    exec(compile(code, "INTEGRATION.md#3", "exec"), ns)
    G = ns["ImpersonatorGenerator"]()
    g = torch.Generator().manual_seed(11)
    src_inputs = torch.rand(1, 6, 256, 256, generator=g) * 2 - 1
    tsf_inputs = torch.rand(2, 6, 256, 256, generator=g) * 2 - 1
    T = torch.rand(2, 256, 256, 2, generator=g) * 2.4 - 1.2
    T[0, 60:140, 30:90] = -2
    enc, res = G.encode_src(src_inputs.cuda())
    color, mask = G.inference(enc, res, tsf_inputs.cuda(), T.cuda())
    with torch.no_grad():
        oenc, ores = torch_ref.encode_src(sd, src_inputs)
        ocolor, omask = torch_ref.generator_inference(sd, oenc, ores, tsf_inputs, T)
    assert len(enc) == 4 and len(res) == 6
    for a, b in zip(list(enc) + list(res), list(oenc) + list(ores)):
        assert tuple(a.shape) == tuple(b.shape)
        assert float((a.cpu() - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max()))
    assert float((color.cpu() - ocolor).abs().max()) <= 1e-3 and float((mask.cpu() - omask).abs().max()) <= 1e-3
    ns["_lwg"].lwg_generator_destroy(G._h)
