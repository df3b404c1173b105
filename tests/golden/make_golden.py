"""Generates the committed golden vectors under tests/golden/ FROM THE REAL REFERENCE.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py

What comes from where
  * teapot_kat.npz   -- the reference's own known-answer fixtures for the rasteriser
                        (thirdparty/neural_renderer/tests/test_rasterize_silhouettes.py:16-35,
                        tests/test_rasterize_depth.py:15-54): the teapot faces after the reference's
                        load_obj normalisation, look_at and perspective (its own python functions),
                        laid out as its to_minibatch fixture (tests/utils.py:11-27: sample 2 of 4,
                        the rest zero), plus the Blender silhouette and the depth PNG.
  * look_at_kat.npz  -- tests/test_look_at.py:9-25.
  * frame_golden.npz -- one end-to-end pass of the hot path executed by the reference's own code
                        (SMPLRenderer.render_fim_wim / encode_fim / cal_bc_transform run unbound,
                        ImpersonatorGenerator, Imitator.forward run unbound) on the seeded synthetic
                        inputs of impersonator_amd/utils/synthetic.py.  The only non-reference piece
                        in that pass is the C restatement of the CUDA rasteriser (oracle/raster_ref.c),
                        which teapot_kat.npz pins.
  * tasks_golden.npz -- BASELINE config 4 and the Viewer, executed by the reference's own METHODS run unbound on a
                        stand-in `self`: `Swapper.personalize` (models/swapper.py:99-165) for two subjects,
                        `Swapper.swap` + `calculate_trans` + `forward` (:198-271) and `Viewer.view` + `rotate_trans` +
                        `forward` (models/viewer.py:262-311) for two views.  Stand-ins: a fixed-vertices `hmr`, the
                        image reader (no cv2), `.cuda()` as the identity, the C rasteriser.
  * imitator_golden.npz -- the headline model's own METHODS run unbound on a stand-in `self`: `Imitator.personalize`
                        (models/imitator.py:82-155; --only_vis on and off, --bg_model ORIGINAL = the generator's BGNet and
                        an InpaintSANet) followed by `Imitator.inference_by_smpls` (:191-214) over four target frames from
                        t = 0, i.e. `transfer_params_by_smpl` (:236-268, `first_cam` set at t == 0), `swap_smpl` (:216-234)
                        under 'smooth' / 'source' / 'copy', `forward` (:326-336) and `warp_front` (:338-342) with
                        --front_warp on and off; the SMPL stage is the reference's `HumanModelRecovery.get_details`
                        (networks/hmr.py:302-330) over its `SMPL.forward` (networks/batch_smpl.py:285-375) on the synthetic
                        body model.  Variants and inputs: tests/helpers.py::IMITATOR_VARIANTS / imitator_scene.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from impersonator_amd.utils import synthetic  # noqa: E402
from oracle import reference_loader  # noqa: E402

NR_TESTS = os.path.join(reference_loader.REFERENCE_ROOT, "thirdparty", "neural_renderer", "tests", "data")


def _read_png(path):
    from PIL import Image
    return np.asarray(Image.open(path))


def make_teapot(ref):
    # load_obj.py:100-147 (vertex / face parsing and normalisation), on CPU
    verts, faces = [], []
    with open(os.path.join(NR_TESTS, "teapot.obj")) as f:
        lines = f.readlines()
    for line in lines:
        tok = line.split()
        if len(tok) == 0:
            continue
        if tok[0] == "v":
            verts.append([float(v) for v in tok[1:4]])
    for line in lines:
        tok = line.split()
        if len(tok) == 0:
            continue
        if tok[0] == "f":
            vs = tok[1:]
            v0 = int(vs[0].split("/")[0])
            for i in range(len(vs) - 2):
                faces.append((v0, int(vs[i + 1].split("/")[0]), int(vs[i + 2].split("/")[0])))
    vertices = torch.from_numpy(np.vstack(verts).astype(np.float32))
    faces = torch.from_numpy(np.vstack(faces).astype(np.int32)) - 1
    vertices -= vertices.min(0)[0][None, :]
    vertices /= torch.abs(vertices).max()
    vertices *= 2
    vertices -= vertices.max(0)[0][None, :] / 2

    # tests/utils.py:11-27 to_minibatch: batch of 4, the sample at index 2, zeros elsewhere
    vb = torch.zeros(4, *vertices.shape)
    fb = torch.zeros(4, *faces.shape, dtype=torch.int32)
    vb[2], fb[2] = vertices, faces

    # renderer.py:75-96 render_silhouettes with camera_mode='look_at', fill_back=True, perspective, 30 deg
    import math
    eye = [0, 0, -(1. / math.tan(math.radians(30)) + 1)]
    fb = torch.cat((fb, fb[:, :, [2, 1, 0]]), dim=1)
    v = ref.nr.look_at(vb, eye)
    v = ref.nr.perspective(v, angle=30)
    f2v = ref.nr.vertices_to_faces(v, fb)

    sil = _read_png(os.path.join(NR_TESTS, "teapot_blender.png"))
    sil = (sil.min(-1) != 255)
    depth = _read_png(os.path.join(NR_TESTS, "test_depth.png"))
    np.savez_compressed(os.path.join(HERE, "teapot_kat.npz"),
                        faces=f2v.numpy().astype(np.float32),
                        silhouette=np.packbits(sil), depth_png=depth.astype(np.uint8))
    print("teapot_kat: faces", tuple(f2v.shape), "covered", int(sil.sum()))


def make_look_at():
    eyes = np.array([[1, 0, 1], [0, 0, -10], [-1, 1, 0]], np.float32)
    answers = np.array([[-np.sqrt(2) / 2, 0, np.sqrt(2) / 2], [1, 0, 10],
                        [0, np.sqrt(2) / 2, 3. / 2. * np.sqrt(2)]], np.float64)
    np.savez(os.path.join(HERE, "look_at_kat.npz"), vertex=np.array([1, 0, 0], np.float32), eyes=eyes,
             answers=answers)


def make_frame(ref):
    torch.set_num_threads(os.cpu_count())
    rest, faces = synthetic.body_mesh()
    map_fn = torch.from_numpy(synthetic.uv_seg_map_fn(rest, faces))
    faces_t = torch.from_numpy(faces)
    image_size = 256

    # --- a stand-in `self` for the reference's SMPLRenderer (its __init__ needs absent asset files)
    R = ref.nmr.SMPLRenderer
    rs = types.SimpleNamespace(faces=faces_t, image_size=image_size, map_fn=map_fn,
                               proj_func=ref.nmr.orthographic_proj_withz_idrot,
                               eye=[0, 0, -(1. / np.tan(np.radians(30)) + 1)])

    # source: rest pose, frame-0 camera.  models/imitator.py:100-107
    src_cam = torch.from_numpy(synthetic.cams(1, seed=100))
    src_verts = torch.from_numpy(rest)[None]
    src_f2verts, src_fim, src_wim = R.render_fim_wim(rs, src_cam, src_verts)
    src_cond, _ = R.encode_fim(rs, src_cam, src_verts, fim=src_fim, transpose=True)
    src_p2verts = src_f2verts[:, :, :, 0:2]
    src_p2verts[:, :, :, 1] *= -1

    src_img = torch.from_numpy(synthetic.smooth_image(11))
    bg_img = torch.from_numpy(synthetic.smooth_image(12))
    ft_mask = 1 - ref.util.morph(src_cond[:, -1:, :, :], ks=3, mode="erode")
    src_inputs = torch.cat([src_img * ft_mask, src_cond], dim=1)

    # generator with seeded weights (random affine so the InstanceNorm gamma/beta path is exercised)
    G = ref.generator.ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6, repeat_num=6).eval()
    shapes = [(k, tuple(v.shape)) for k, v in G.state_dict().items()]
    sd = synthetic.random_state_dict(shapes, seed=0, affine="random")
    G.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})

    with torch.no_grad():
        src_feats = G.encode_src(src_inputs)

        # two target frames (t = 3 and t = 200 of a 1024-frame motion), each with its own camera
        bs = 2
        tgt_verts = torch.from_numpy(np.stack([synthetic.motion_verts(rest, t) for t in (3, 200)]))
        tgt_cam = torch.from_numpy(synthetic.cams(bs, seed=5))
        _, fim, wim = R.render_fim_wim(rs, tgt_cam, tgt_verts)
        cond, _ = R.encode_fim(rs, tgt_cam, tgt_verts, fim=fim, transpose=True)
        # the reference runs batch 1 (models/imitator.py:166); per-sample calls keep its semantics
        T = torch.cat([R.cal_bc_transform(rs, src_p2verts, fim[i:i + 1], wim[i:i + 1]) for i in range(bs)])
        tsf_img = torch.cat([torch.nn.functional.grid_sample(src_img, T[i:i + 1]) for i in range(bs)])
        tsf_inputs = torch.cat([tsf_img, cond], dim=1)

        stub = types.SimpleNamespace(generator=G, src_info=dict(bg=bg_img, feats=src_feats),
                                     _opt=types.SimpleNamespace(front_warp=False))
        preds = torch.cat([ref.imitator.Imitator.forward(stub, tsf_inputs[i:i + 1], T[i:i + 1]) for i in range(bs)])
        color, mask = G.inference(src_feats[0], src_feats[1], tsf_inputs[:1], T[:1])

    def stat(x):
        x = x.double()
        return np.array([x.mean().item(), x.abs().mean().item(), (x * x).mean().item()])

    out = dict(
        src_fim=src_fim.numpy().astype(np.int32),
        fim=fim.numpy().astype(np.int32),
        wim=wim.numpy().astype(np.float16).astype(np.float32) * 0,  # placeholder, replaced below
        T=T.numpy(), preds=preds.numpy(),
        color0_sub=color.numpy()[:, :, ::4, ::4], mask0_sub=mask.numpy()[:, :, ::4, ::4],
        tsf_img_stat=stat(tsf_img), cond_stat=stat(cond), wim_stat=stat(wim),
        src_enc_stat=np.stack([stat(t) for t in src_feats[0]]),
        src_res_stat=np.stack([stat(t) for t in src_feats[1]]),
        src_inputs_stat=stat(src_inputs), weights_stat=np.array(
            [float(np.sum([np.abs(v.astype(np.float64)).sum() for v in sd.values()]))]),
    )
    # wim: keep exact float32 values only where covered (sparse), to keep the fixture small
    cov = fim.numpy() >= 0
    out["wim_covered"] = wim.numpy()[cov]
    del out["wim"]
    np.savez_compressed(os.path.join(HERE, "frame_golden.npz"), **out)
    print("frame_golden: covered px", int(cov.sum()), "preds range", float(preds.min()), float(preds.max()))


def make_tasks(ref):
    import importlib
    ref_swapper = importlib.import_module("models.swapper")
    ref_viewer = importlib.import_module("models.viewer")
    S, V, R = ref_swapper.Swapper, ref_viewer.Viewer, ref.nmr.SMPLRenderer
    torch.set_num_threads(os.cpu_count())
    from tests import helpers
    sc = helpers.task_scene()
    FixedHMR = helpers.FixedHMR
    image_size = 256
    rs = types.SimpleNamespace(faces=torch.from_numpy(sc["faces"]), image_size=image_size, map_fn=torch.from_numpy(sc["map_fn"]),
                               proj_func=ref.nmr.orthographic_proj_withz_idrot, eye=[0, 0, -(1. / np.tan(np.radians(30)) + 1)])
    for name in ("render_fim_wim", "encode_fim", "cal_bc_transform"):
        setattr(rs, name, types.MethodType(getattr(R, name), rs))
    G = ref.generator.ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6, repeat_num=6).eval()
    shapes = [(k, tuple(v.shape)) for k, v in G.state_dict().items()]
    sd = synthetic.random_state_dict(shapes, seed=0, affine="random")
    G.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})

    images = {"A": sc["img_a"][0], "B": sc["img_b"][0]}
    # the reference reads a file and maps [0,255] -> [-1,1]; the stand-ins hand the float image through exactly
    # (float64 in between: (x + 1) / 2 * 2 - 1 == x there)
    for mod in (ref_swapper.cv_utils,):
        mod.read_cv2_img = lambda path: images[path]
        mod.transform_img = lambda img, size, transpose=True: (img.astype(np.float64) + 1.0) / 2.0
    cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        t = torch.from_numpy
        stub = types.SimpleNamespace(
            _opt=types.SimpleNamespace(image_size=image_size, only_vis=False, bg_model='ORIGINAL', bg_ks=13, ft_ks=3, front_warp=False),
            hmr=FixedHMR([(t(sc["cam_a"]), t(sc["verts_a"])), (t(sc["cam_b"]), t(sc["verts_b"]))]), render=rs, detector=None,
            bgnet=G.bg_model, generator=G, part_fn=t(sc["part_fn"]), part_faces=sc["part_faces"], PART_IDS=S.PART_IDS,
            grid=R.create_meshgrid(image_size), src_info=None, tsf_info=None)
        captured = {}

        def calculate_trans(self, mask, faces):
            captured["left_mask"], captured["left_faces"] = mask.clone(), list(faces)
            captured["T11"], captured["T21"] = S.calculate_trans(self, mask, faces)
            return captured["T11"], captured["T21"]

        def forward(self, tsf_inputs, *args):
            captured["tsf_inputs"] = tsf_inputs.clone()
            return S.forward(self, tsf_inputs, *args)

        stub.calculate_trans = types.MethodType(calculate_trans, stub)
        stub.forward = types.MethodType(forward, stub)
        smpl = np.zeros(85, np.float32)
        with torch.no_grad():
            stub.src_info = S.personalize(stub, "A", smpl)
            stub.tsf_info = S.personalize(stub, "B", smpl)
            preds = S.swap(stub, stub.src_info, stub.tsf_info, target_part='body')
        A, B = stub.src_info, stub.tsf_info
        out = dict(
            fim_a=A["fim"].numpy().astype(np.int16), fim_b=B["fim"].numpy().astype(np.int16),
            part_a=A["part"].argmax(1).numpy().astype(np.uint8),
            bg_a_sub=A["bg"].numpy()[:, :, ::4, ::4], bg_b_sub=B["bg"].numpy()[:, :, ::4, ::4],
            left_mask=np.packbits(captured["left_mask"].numpy()), left_faces_n=np.array([len(captured["left_faces"])]),
            T11=captured["T11"].numpy(), T21=captured["T21"].numpy(), tsf_inputs_sub=captured["tsf_inputs"].numpy()[:, :, ::2, ::2],
            swap_preds=preds.numpy())

        # --- Viewer.view on subject A (the reference's view / rotate_trans / forward)
        vstub = types.SimpleNamespace(src_info=A, render=rs, generator=G, _opt=types.SimpleNamespace(bg_replace=False, front_warp=False))
        meshes = []

        def rotate_trans(self, rt, tr, X):
            meshes.append(V.rotate_trans(self, rt, tr, X))
            return meshes[-1]

        vstub.rotate_trans = types.MethodType(rotate_trans, vstub)
        vstub.forward = types.MethodType(V.forward, vstub)
        for i, (rt, tr, replace) in enumerate(sc["views"]):
            vstub._opt.bg_replace = replace
            with torch.no_grad():
                vp = V.view(vstub, rt, tr)
            out["view%d_mesh" % i] = meshes[-1].numpy()
            out["view%d_preds" % i] = vp.numpy()
    finally:
        torch.Tensor.cuda = cuda
    np.savez_compressed(os.path.join(HERE, "tasks_golden.npz"), **out)
    print("tasks_golden: swap preds range", float(preds.min()), float(preds.max()), "kept px", int(captured["left_mask"].sum()),
          "left faces", len(captured["left_faces"]))


def make_imitator(ref):
    from impersonator_amd.networks.batch_smpl import SMPL as ProductSMPL
    from tests import helpers
    I, R = ref.imitator.Imitator, ref.nmr.SMPLRenderer
    torch.set_num_threads(os.cpu_count())
    G = ref.generator.ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6, repeat_num=6).eval()
    G.load_state_dict({k: torch.from_numpy(v) for k, v in helpers.generator_state_dict(seed=0, affine="random").items()})
    inpaint = ref.inpaintor.InpaintSANet(c_dim=4).eval()
    inpaint.load_state_dict({k: torch.from_numpy(v) for k, v in helpers.inpaintor_state_dict(seed=1).items()})

    images = {}
    ref.imitator.cv_utils.read_cv2_img = lambda path: images[path]
    # the reference maps [0,255] -> [-1,1]; the stand-in hands the float image through exactly (float64 in between)
    ref.imitator.cv_utils.transform_img = lambda img, size, transpose=True: (img.astype(np.float64) + 1.0) / 2.0
    cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = {}
    try:
        for name, v in helpers.IMITATOR_VARIANTS.items():
            sc = helpers.imitator_scene(v["size"])
            t = torch.from_numpy
            # the reference's SMPL.forward on the synthetic body model's tensors (its __init__ reads the absent pickle)
            pm = ProductSMPL(params=sc["smpl_params"])
            smpl_self = types.SimpleNamespace(shapedirs=pm.shapedirs, v_template=pm.v_template, size=pm.size, J_regressor=pm.J_regressor,
                                              posedirs=pm.posedirs, parents=pm.parents, weights=pm.weights,
                                              joint_regressor=pm.joint_regressor, rotate=False)
            hmr = types.SimpleNamespace(smpl=lambda beta, theta, get_skin=False: ref.batch_smpl.SMPL.forward(smpl_self, beta, theta,
                                                                                                            get_skin=get_skin))
            hmr.get_details = types.MethodType(ref.networks.HumanModelRecovery.get_details, hmr)
            rs = types.SimpleNamespace(faces=t(sc["faces"]), image_size=v["size"], map_fn=t(sc["map_fn"]), front_map_fn=t(sc["front_map_fn"]),
                                       proj_func=ref.nmr.orthographic_proj_withz_idrot, eye=[0, 0, -(1. / np.tan(np.radians(30)) + 1)],
                                       get_vis_f2pts=R.get_vis_f2pts)
            for m in ("render_fim_wim", "encode_fim", "encode_front_fim", "cal_bc_transform"):
                setattr(rs, m, types.MethodType(getattr(R, m), rs))
            stub = types.SimpleNamespace(
                _opt=types.SimpleNamespace(image_size=v["size"], only_vis=v["only_vis"], bg_model=v["bg_model"], bg_ks=13, ft_ks=3,
                                           front_warp=v["front_warp"]),
                hmr=hmr, render=rs, detector=None, generator=G, bgnet=G.bg_model if v["bg_model"] == "ORIGINAL" else inpaint,
                src_info=None, tsf_info=None, first_cam=None)
            frames = []

            def transfer_params_by_smpl(self, tgt_smpl, cam_strategy='smooth', t=0):
                x = I.transfer_params_by_smpl(self, tgt_smpl, cam_strategy, t)
                frames.append(dict(self.tsf_info, tsf_inputs=x, first_cam=None if self.first_cam is None else self.first_cam.clone()))
                return x

            stub.transfer_params_by_smpl = types.MethodType(transfer_params_by_smpl, stub)
            for m in ("swap_smpl", "forward", "warp_front"):
                setattr(stub, m, types.MethodType(getattr(I, m), stub))
            images["SRC"] = sc["src_img"][0]
            with torch.no_grad():
                I.personalize(stub, "SRC", sc["src_smpl"])
                outs = I.inference_by_smpls(stub, sc["tgt_smpls"], cam_strategy=v["cam_strategy"], output_dir='')
            si = stub.src_info
            preds = np.stack(outs).transpose(0, 3, 1, 2)                 # the method returns (H,W,3) arrays
            k = name + "/"
            out[k + "src_theta"], out[k + "src_cam"], out[k + "src_verts"] = si["theta"].numpy(), si["cam"].numpy(), si["verts"].numpy()
            out[k + "src_fim"] = si["fim"].numpy().astype(np.int16)
            out[k + "src_p2verts"] = si["p2verts"].numpy()
            out[k + "src_f2verts_stat"] = helpers.tensor_stat(si["f2verts"])
            out[k + "src_cond_stat"] = helpers.tensor_stat(si["cond"])
            out[k + "src_bg_sub"] = si["bg"].numpy()[:, :, ::4, ::4]
            out[k + "src_enc_stat"] = np.stack([helpers.tensor_stat(x) for x in si["feats"][0]])
            out[k + "src_res_stat"] = np.stack([helpers.tensor_stat(x) for x in si["feats"][1]])
            cat = lambda key: torch.cat([f[key] for f in frames]).numpy()
            out[k + "theta"], out[k + "cam"], out[k + "verts"], out[k + "j2d"] = cat("theta"), cat("cam"), cat("verts"), cat("j2d")
            out[k + "fim"] = cat("fim").astype(np.int16)
            out[k + "first_cam"] = np.stack([np.full(3, np.nan, np.float32) if f["first_cam"] is None else f["first_cam"][0].numpy()
                                             for f in frames])
            out[k + "tsf_img_stat"] = np.stack([helpers.tensor_stat(f["tsf_img"]) for f in frames])
            out[k + "cond_stat"] = np.stack([helpers.tensor_stat(f["cond"]) for f in frames])
            T = cat("T")
            if v["size"] > 128:      # the 256x256 pass: first and last frame in full, the others on every second pixel
                out[k + "T_full"], out[k + "preds_full"] = T[[0, -1]], preds[[0, -1]]
                out[k + "T_sub"], out[k + "preds_sub"] = T[1:-1, ::2, ::2], preds[1:-1, :, ::2, ::2]
            else:
                out[k + "T_full"], out[k + "preds_full"] = T, preds
            print("imitator_golden[%s]: %d frames, covered px %s, visible faces %d, preds range %.3f..%.3f" % (
                name, len(frames), [int((f["fim"] >= 0).sum()) for f in frames], int((si["p2verts"][0, :, 0, 0] != -2).sum()),
                preds.min(), preds.max()))
    finally:
        torch.Tensor.cuda = cuda
    np.savez_compressed(os.path.join(HERE, "imitator_golden.npz"), **out)


def make_discriminator(ref):
    """One discriminator update of the REAL reference code: PatchDiscriminator (networks/discriminator.py) as the
    trainer builds it (impersonator_trainer.py:219-222), the LSGAN loss of _optimize_D/_compute_loss_D (:396-414),
    torch autograd and torch.optim.Adam (:231-232), on seeded weights/inputs (tests/helpers.discriminator_state_dict)."""
    from tests import helpers
    D = ref.discriminator.PatchDiscriminator(input_nc=6, ndf=64, n_layers=4, norm_type='instance', use_sigmoid=False)
    sd = helpers.discriminator_state_dict(seed=3)
    D.load_state_dict(sd)
    gen = torch.Generator().manual_seed(1)
    real = torch.rand(2, 6, 64, 64, generator=gen) * 2 - 1
    fake = torch.rand(2, 6, 64, 64, generator=gen) * 2 - 1
    opt = torch.optim.Adam(D.parameters(), lr=0.0002, betas=(0.5, 0.999))
    with torch.no_grad():
        d_real = D(real).numpy().copy()
    opt.zero_grad()
    loss = torch.mean((D(real) - 1) ** 2) + torch.mean((D(fake) + 1) ** 2)
    loss.backward()
    out = dict(d_real=d_real, loss=np.array([float(loss.detach())]))
    for k, p in D.named_parameters():   # first weight gradient in full, the rest as (L1, L2) norms + a strided sample
        g = p.grad.detach().double()
        out["gnorm/" + k] = np.array([g.abs().sum().item(), (g * g).sum().sqrt().item()])
        out["gsample/" + k] = p.grad.detach().flatten()[::97].numpy().copy()
    opt.step()
    for k, p in D.named_parameters():
        out["psample/" + k] = p.detach().flatten()[::97].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "discriminator_golden.npz"), **out)
    print("discriminator_golden: loss", float(loss.detach()))


if __name__ == "__main__":
    ref = reference_loader.load()
    if len(sys.argv) > 1 and sys.argv[1] == "discriminator":   # add this fixture without regenerating the others
        make_discriminator(ref)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tasks":
        make_tasks(ref)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "imitator":
        make_imitator(ref)
        sys.exit(0)
    make_teapot(ref)
    make_look_at()
    make_frame(ref)
    make_discriminator(ref)
    make_tasks(ref)
    make_imitator(ref)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
