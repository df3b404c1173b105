"""Quick on-GPU timing of the hot-path stages (development aid; bench.py is the contract)."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from impersonator_amd.networks.generator import ImpersonatorGenerator
from impersonator_amd.utils import synthetic
from impersonator_amd.utils.nmr import SMPLRenderer
from tests import helpers


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    bs = 8
    s = helpers.scene()
    r = SMPLRenderer(image_size=256, faces=s["faces"], map_fn=s["map_fn"]).cuda()
    G = ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6, max_batch=bs)
    sd = {k: torch.from_numpy(v) for k, v in helpers.generator_state_dict(0, "identity").items()}
    G.load_state_dict(sd)
    G = G.cuda()
    verts = torch.from_numpy(np.stack([synthetic.motion_verts(s["rest"], t) for t in range(bs)])).cuda()
    cam = torch.from_numpy(synthetic.cams(bs, seed=1)).cuda()
    sf2v, sfim, _ = r.render_fim_wim(helpers.t(s["src_cam"]).cuda(), helpers.t(s["src_verts"]).cuda())
    p2v = sf2v[:, :, :, :2].clone()
    p2v[..., 1] *= -1
    src_img = helpers.t(s["src_img"]).cuda()
    bg = helpers.t(s["bg_img"]).cuda()
    scond, _ = r.encode_fim(None, None, fim=sfim)
    src_inputs = torch.cat([src_img, scond], 1)
    res = {}
    res["encode_src_ms"] = timeit(lambda: G.encode_src(src_inputs))
    enc, rs = G.encode_src(src_inputs)
    res["transfer_ms"] = timeit(lambda: r.transfer(cam, verts, p2v, src_img))
    out = r.transfer(cam, verts, p2v, src_img)
    res["inference_ms"] = timeit(lambda: G.inference(enc, rs, out["tsf_inputs"], out["T"], bg_img=bg))
    G.profile(True)
    G.inference(enc, rs, out["tsf_inputs"], out["T"], bg_img=bg)
    n, ms, fl = G.profile_read()
    res['by_kernel'] = {k: (v[0], round(v[1], 3), round(v[2] / v[1] / 1e9, 1)) for k, v in G.profile_table().items()}
    G.profile(False)
    res["igemm_launches"], res["igemm_ms"], res["igemm_tflops"] = n, ms, fl / ms / 1e9
    res["fps"] = bs / ((res["transfer_ms"] + res["inference_ms"]) / 1e3)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
