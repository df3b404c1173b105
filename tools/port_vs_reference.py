"""Calibrates bench.py's `cpu_baseline.kind: "port"`: times the reference's OWN modules (imported from
/root/reference through oracle/reference_loader.py) against the CPU port (oracle/torch_ref.py + oracle/raster_ref.c)
on the same batch of 8 frames, same box, same thread count.  Runs only where the reference tree exists (the build
container); the result is committed under profiles/.
    python tools/port_vs_reference.py [runs=3] > profiles/r02_port_vs_reference.md
What is timed per batch: SMPLRenderer.render_fim_wim + encode_fim + cal_bc_transform (the
reference loops over the batch), F.grid_sample + cat, ImpersonatorGenerator.inference, the blend of Imitator.forward.
The reference's CUDA rasteriser cannot run here; BOTH sides call the same C restatement for that step (its share of
the batch time is printed)."""
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import reference_loader, torch_ref  # noqa: E402
from impersonator_amd import demo  # noqa: E402
from impersonator_amd.networks.batch_smpl import HumanModelRecovery, synthetic_smpl_params  # noqa: E402
from impersonator_amd.networks.generator import ImpersonatorGenerator  # noqa: E402
from impersonator_amd.utils import synthetic  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ref = reference_loader.load()
cores = os.cpu_count()
torch.set_num_threads(cores)
B = 8
rest, faces = synthetic.body_mesh()
faces_t = torch.from_numpy(faces)
map_fn = torch.from_numpy(synthetic.uv_seg_map_fn(rest, faces))
shapes = [(k, tuple(v.shape)) for k, v in ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6).state_dict().items()]
sd = torch_ref.state_dict_from_numpy(synthetic.random_state_dict(shapes, seed=0, affine="identity"))
G = ref.generator.ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6, repeat_num=6).eval()
G.load_state_dict(sd)
hmr = HumanModelRecovery(smpl_params=synthetic_smpl_params(0))
src_smpl = torch.from_numpy(demo.synthetic_smpls(1, 1))
src_smpl[:, 3:75] = 0
src_img = torch.from_numpy(synthetic.smooth_image(11))
bg_img = torch.from_numpy(synthetic.smooth_image(12))
smpls = torch.from_numpy(demo.synthetic_smpls(1024, 0))
R = ref.nmr.SMPLRenderer
rs = types.SimpleNamespace(faces=faces_t, image_size=256, map_fn=map_fn, proj_func=ref.nmr.orthographic_proj_withz_idrot,
                           eye=[0, 0, -(1. / np.tan(np.radians(30)) + 1)])
imi = types.SimpleNamespace(generator=G, src_info=None, _opt=types.SimpleNamespace(front_warp=False))

with torch.no_grad():
    si = hmr.get_details(src_smpl)
    sf2v, sfim, _ = torch_ref.render_fim_wim(si["cam"], si["verts"], faces_t)
    p2v = torch_ref.source_p2verts(sf2v)
    scond = torch_ref.encode_fim(sfim, map_fn)
    ft = 1 - torch_ref.morph(scond[:, -1:], 3, "erode")
    src_inputs = torch.cat([src_img * ft, scond], 1)
    enc, res = torch_ref.encode_src(sd, src_inputs)
    r_enc, r_res = G.encode_src(src_inputs)
    # the reference's grid_sample needs the source features at the batch size of the flow (it runs frame by frame)
    imi.src_info = {"feats": ([f.expand(B, -1, -1, -1) for f in r_enc], [f.expand(B, -1, -1, -1) for f in r_res]),
                    "bg": bg_img}

    def geometry_inputs(b):
        chunk = smpls[b * B:(b + 1) * B]
        cam = si["cam"].expand(B, -1).clone()
        cam[:, 1:] += chunk[:, 1:3] - smpls[0:1, 1:3]
        return hmr.get_details(torch.cat([cam, chunk[:, 3:75], si["shape"].expand(B, -1)], 1))

    def port(info):
        fr = torch_ref.transfer_frame(src_img, p2v, info["cam"], info["verts"], faces_t, map_fn)
        return torch_ref.imitator_forward(sd, enc, res, bg_img, fr["tsf_inputs"], fr["T"])[0]

    def reference(info):
        # models/imitator.py:250-260 + 326-336 with the reference's own classes (methods run unbound on stubs)
        f2v, fim, wim = R.render_fim_wim(rs, info["cam"], info["verts"])
        cond, _ = R.encode_fim(rs, info["cam"], info["verts"], fim=fim, transpose=True)
        T = R.cal_bc_transform(rs, p2v.expand(B, -1, -1, -1), fim, wim)
        tsf_img = torch.nn.functional.grid_sample(src_img.expand(B, -1, -1, -1), T)
        tsf_inputs = torch.cat([tsf_img, cond], dim=1)
        return ref.imitator.Imitator.forward(imi, tsf_inputs, T)

    def raster_only(info):
        torch_ref.render_fim_wim(info["cam"], info["verts"], faces_t)

    infos = [geometry_inputs(b) for b in range(1, 1 + runs)]
    a, b = port(infos[0]), reference(infos[0])          # warm-up + agreement
    agree = float((a - b).abs().max())
    tp, tr, tz = [], [], []
    for info in infos:
        t0 = time.perf_counter(); port(info); tp.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); reference(info); tr.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); raster_only(info); tz.append(time.perf_counter() - t0)

med = lambda v: sorted(v)[len(v) // 2]
print("# CPU port vs the reference's own modules (tools/port_vs_reference.py, %d runs, batch of %d frames 256x256)\n" % (runs, B))
print("box: %d logical cores, torch %s, torch.set_num_threads(%d)\n" % (cores, torch.__version__, cores))
print("| what | seconds per batch (each run) | median | frames/s |\n|---|---|---|---|")
print("| reference modules (`SMPLRenderer.render_fim_wim/encode_fim/cal_bc_transform`, `F.grid_sample`, `Imitator.forward` -> "
      "`ImpersonatorGenerator.inference`) | %s | %.3f | %.2f |" % (", ".join("%.3f" % v for v in tr), med(tr), B / med(tr)))
print("| port (`oracle/torch_ref.py`, what `bench.py: cpu_baseline` times on the GPU box) | %s | %.3f | %.2f |"
      % (", ".join("%.3f" % v for v in tp), med(tp), B / med(tp)))
print("| of which the C rasteriser restatement (both sides call it) | %s | %.3f | |" % (", ".join("%.3f" % v for v in tz), med(tz)))
print("\nport / reference speed ratio: **%.3f** (>1 = the port is faster); max |port - reference| on the final image: %.2e\n"
      % (med(tr) / med(tp), agree))
print("The reference warns that `grid_sample`'s default `align_corners` changed (hazard H1); both sides use torch %s's default."
      % torch.__version__)
