"""The command-line entry points, executed (reference: run_imitator.py:214-241, run_swap.py:39-69): `run_imitator.py
--synthetic` as TWO ranks under `python -m torch.distributed.run` (gloo rendezvous on the one GPU: frame-sharded blocks,
gather in frame order, rank 0 writes) and `run_swap.py --synthetic --save_res`.  What lands on disk is compared
  * bit for bit with the truncating uint8 conversion (utils/cv_utils.py:31-33, hazard H11) of the float frames the same
    model produces in this process, and
  * with the same conversion of the CPU oracle's frames: the float images agree within 1e-3, i.e. 0.13 of a grey level, so
    a truncated value may land one level apart where the float value sits that close to an integer -- never more."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from impersonator_amd import demo
from oracle import torch_ref

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRAMES, BATCH = 20, 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, timeout=900):
    env = dict(os.environ, LWG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    assert p.returncode == 0, "%s failed (%d)\n%s\n%s" % (cmd, p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    return p.stdout


def _u8(x):
    """utils/cv_utils.py:31-33: (x + 1) / 2 * 255 then astype(uint8) -- truncation, not rounding."""
    return ((np.asarray(x) + 1) / 2.0 * 255).astype(np.uint8)


def _read(path):
    from PIL import Image
    return np.asarray(Image.open(path))


def test_run_imitator_two_ranks_writes_the_frames(tmp_path):
    out_dir = str(tmp_path / "imitate")
    stdout = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                   "--master-port", str(_free_port()), "run_imitator.py", "--synthetic", "--num_frames", str(FRAMES), "--batch_size",
                   str(BATCH), "--output_dir", out_dir])
    assert "wrote %d frames" % FRAMES in stdout
    files = sorted(os.listdir(out_dir))
    assert files == ["pred_%.8d.png" % t for t in range(FRAMES)]
    disk = np.stack([_read(os.path.join(out_dir, f)) for f in files])
    assert disk.shape == (FRAMES, 256, 256, 3) and disk.dtype == np.uint8

    # the same model in this process, one rank: float frames + the posed vertices the oracle restarts from
    imitator, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=BATCH, image_size=256,
                                                                        opt=demo.default_opt(batch_size=BATCH, image_size=256))
    imitator.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
    smpls = torch.from_numpy(demo.synthetic_smpls(FRAMES, seed=0)).cuda()
    imitator.first_cam = smpls[0:1, 0:3].clone()
    preds, verts, cams = [], [], []
    for _, p in imitator.predict_batches(((smpls[s:s + BATCH], s) for s in range(0, FRAMES, BATCH)), "smooth"):
        preds.append(p.permute(0, 2, 3, 1).cpu().numpy())
        verts.append(imitator.tsf_info["verts"].cpu())
        cams.append(imitator.tsf_info["cam"].cpu())
    mine = np.concatenate(preds)
    assert np.array_equal(disk, _u8(mine)), "the files are not the truncated uint8 form of the frames this model computes"

    sd = {k: v.detach().cpu() for k, v in imitator.generator.state_dict().items()}
    faces_t, map_fn, si = imitator.render.faces.cpu(), imitator.render.map_fn.cpu(), imitator.src_info
    src_t, bg_t = torch.from_numpy(src_img)[None], torch.from_numpy(bg_img)[None]
    with torch.no_grad():
        src = torch_ref.personalize(sd, src_t, si["cam"].cpu(), si["verts"].cpu(), faces_t, map_fn, ft_ks=imitator._opt.ft_ks)
        _, ref = torch_ref.imitator_frames(sd, src, src_t, bg_t, torch.cat(cams), torch.cat(verts), faces_t, map_fn)
    ref = ref.permute(0, 2, 3, 1).numpy()
    assert float(np.abs(ref - mine).max()) <= 1e-3
    d = np.abs(disk.astype(np.int16) - _u8(ref).astype(np.int16))
    print("uint8 frames vs oracle: %.2f %% of values one level apart, max %d" % (100.0 * (d > 0).mean(), d.max()))
    assert d.max() <= 1 and (d > 0).mean() < 0.25


def test_run_swap_writes_the_swap(tmp_path):
    out_dir = str(tmp_path / "swap")
    stdout = _run([sys.executable, "run_swap.py", "--synthetic", "--save_res", "--output_dir", out_dir])
    path = os.path.join(out_dir, "swappers", "synthetic_a->synthetic_b.png")
    assert "Saving results to" in stdout and os.path.exists(path)
    disk = _read(path)

    from impersonator_amd.utils import synthetic
    sw, smpl_a, img_a, bg_a = demo.build_synthetic_imitator(batch_size=1, model="swapper", opt=demo.default_opt(batch_size=1))
    smpl_b = demo.synthetic_smpls(8, seed=3)[5]
    img_b = synthetic.smooth_image(77, (1, 3, 256, 256))[0]
    sw.swap_setup(img_a, img_b, src_smpl=smpl_a, tgt_smpl=smpl_b, src_bg=bg_a, tgt_bg=bg_a)
    mine = sw.swap(src_info=sw.src_info, tgt_info=sw.tsf_info, target_part="body")[0].permute(1, 2, 0).cpu().numpy()
    assert np.array_equal(disk, _u8(mine))

    sd = {k: v.detach().cpu() for k, v in sw.generator.state_dict().items()}
    faces_t, map_fn, part_fn = sw.render.faces.cpu(), sw.render.map_fn.cpu(), sw.part_fn.cpu()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))[None]
    with torch.no_grad():
        A = torch_ref.swapper_personalize(sd, t(img_a), sw.src_info["cam"].cpu(), sw.src_info["verts"].cpu(), faces_t, map_fn, part_fn)
        B = torch_ref.swapper_personalize(sd, t(img_b), sw.tsf_info["cam"].cpu(), sw.tsf_info["verts"].cpu(), faces_t, map_fn, part_fn)
        A["bg"] = B["bg"] = t(bg_a)                    # the CLI hands the backgrounds in (no inpainting)
        ref = torch_ref.swapper_swap(sd, A, B, sw.part_faces)["preds"][0].permute(1, 2, 0).numpy()
    assert float(np.abs(ref - mine).max()) <= 1e-3
    d = np.abs(disk.astype(np.int16) - _u8(ref).astype(np.int16))
    assert d.max() <= 1 and (d > 0).mean() < 0.25


def test_run_imitator_with_asset_files(tmp_path):
    """The non-synthetic command line (run_imitator.py:214-241 of the reference): every input comes from FILES in the
    reference's formats and default locations -- `assets/pretrains/smpl_model.pkl` (pickle with sparse regressors),
    `smpl_faces.npy`, `mapper.txt` (an .obj with per-corner texture indices), a generator checkpoint written by `torch.save`
    (with DataParallel's `module.` prefix on half of the keys, which `_load_params` strips), a source image and a directory of
    target images as PNG files with their SMPL vectors beside them.  `Imitator(opt)` builds everything from `opt` (no injected
    component), BGNet inpaints the background (`--bg_model ORIGINAL`), and the written frames must be the uint8 truncation of
    what an Imitator assembled in this process from the same arrays computes."""
    import pickle
    import scipy.sparse
    from PIL import Image
    from impersonator_amd.networks.batch_smpl import HumanModelRecovery, synthetic_smpl_params
    from impersonator_amd.networks.generator import ImpersonatorGenerator
    from impersonator_amd.models.imitator import Imitator
    from impersonator_amd.utils import cv_utils, mesh, synthetic
    from impersonator_amd.utils.nmr import SMPLRenderer

    root = tmp_path
    pre = root / "assets" / "pretrains"
    pre.mkdir(parents=True)
    rest, faces = synthetic.body_mesh()
    np.save(pre / "smpl_faces.npy", faces.astype(np.int32))
    # mapper.txt: one texture vertex per face corner, placed so that the face's UV barycentre is its (u, v) of the synthetic table
    uv = synthetic.uv_seg_map_fn(rest, faces)[:-1, :2]
    with open(pre / "mapper.txt", "w") as fp:
        for (u, v) in uv:
            for du, dv in ((0.0, 0.0), (1e-3, 0.0), (0.0, 1e-3)):
                fp.write("vt %.7f %.7f\n" % (u + du, 1.0 - (v + dv)))
        for f, (a, b, c) in enumerate(faces):
            fp.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a + 1, 3 * f + 1, a + 1, b + 1, 3 * f + 2, b + 1, c + 1, 3 * f + 3, c + 1))
    p = synthetic_smpl_params(0)
    dd = dict(p)
    dd["J_regressor"] = scipy.sparse.csc_matrix(np.asarray(p["J_regressor"]))
    dd["cocoplus_regressor"] = scipy.sparse.csc_matrix(np.asarray(p["cocoplus_regressor"]))
    pickle.dump(dd, open(pre / "smpl_model.pkl", "wb"), protocol=2)
    gen = ImpersonatorGenerator(bg_dim=4, src_dim=6, tsf_dim=6, repeat_num=6, image_size=256, max_batch=4)
    sd = {k: torch.from_numpy(v) for k, v in synthetic.random_state_dict(
        [(k, tuple(v.shape)) for k, v in gen.state_dict().items()], seed=4, affine="random").items()}
    torch.save({("module." + k if i % 2 else k): v for i, (k, v) in enumerate(sd.items())}, root / "G.pth")

    def write_png(path, seed):
        img = ((synthetic.smooth_image(seed)[0].transpose(1, 2, 0) + 1) / 2 * 255).astype(np.uint8)
        Image.fromarray(img).save(path)
        return img

    src_u8 = write_png(root / "src.png", 31)
    src_smpl = demo.synthetic_smpls(1, seed=1)[0]
    src_smpl[3:75] = 0
    np.save(str(root / "src.png") + ".smpl.npy", src_smpl)
    tgt_dir = root / "targets"
    tgt_dir.mkdir()
    tgt_smpls = demo.synthetic_smpls(64, seed=2)[::11][:6]
    for i, th in enumerate(tgt_smpls):
        write_png(tgt_dir / ("frame_%03d.png" % i), 100 + i)          # the target images themselves are not read on this path
        np.save(str(tgt_dir / ("frame_%03d.png" % i)) + ".smpl.npy", th)

    out_dir = root / "out"
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, os.path.join(ROOT, "run_imitator.py"), "--src_path", str(root / "src.png"), "--tgt_path", str(tgt_dir),
           "--load_path", str(root / "G.pth"), "--bg_model", "ORIGINAL", "--batch_size", "4", "--output_dir", str(out_dir)]
    pr = subprocess.run(cmd, cwd=str(root), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, text=True)
    assert pr.returncode == 0, pr.stderr[-3000:]
    files = sorted(os.listdir(out_dir))
    assert files == ["pred_frame_%03d.png" % i for i in range(6)], files
    disk = np.stack([_read(os.path.join(out_dir, f)) for f in files])

    # the same model from the same arrays, assembled here
    opt = demo.default_opt(batch_size=4, image_size=256)
    render = SMPLRenderer(image_size=256, faces=faces, map_fn=mesh.create_mapping("uv_seg", str(pre / "mapper.txt")))
    tab = np.asarray(render.map_fn.cpu())
    assert tab.shape == (faces.shape[0] + 1, 3) and np.abs(tab[:-1, :2] - uv).max() < 1e-3 and (tab[-1] == [0, 0, 1]).all()
    gen.load_state_dict(sd)
    imitator = Imitator(opt, hmr=HumanModelRecovery(smpl_params=p), render=render, generator=gen)
    imitator.personalize(src_u8, src_smpl=src_smpl)          # HxWx3 uint8: the same conversion as a file read
    mine = np.stack(imitator.inference_by_smpls(tgt_smpls, cam_strategy="smooth"))
    assert np.array_equal(disk, _u8(mine))
    assert disk.std() > 10      # not a constant image
    imitator.generator.release()
