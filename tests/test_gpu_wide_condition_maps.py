"""The condition maps wider than 'uv_seg' (utils/mesh.py:446-473: 'par' = 10 part labels + background, 'binary' = the face index in
binary): 3 + cond_nc = 14 / 17 input channels for both streams of ImpersonatorGenerator (models/models.py:85-94,
models/imitator.py:68-70).  Such inputs arrive NCHW (SMPLRenderer.transfer packs NHWC8 only for three condition channels) and the 7x7
stems run on the 16- / 32-channel padded kernels; everything behind the stem is the default path.  Parity against the CPU oracle on
the whole chain: renderer -> cond -> T -> generator -> blend, both arithmetic modes, 1e-3 on the image (north-star bound)."""
import numpy as np
import pytest
import torch

from impersonator_amd.utils import mesh, synthetic
from oracle import torch_ref
from tests import helpers

pytestmark = pytest.mark.gpu
TOL_IMAGE = 1e-3


def _binary_table(nf):
    """create_mapping('binary') reads the reference's mapper.txt for the face count only: the same table from nf."""
    return np.concatenate(mesh.binary_mapping(nf), axis=0)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("map_name", ["par", "binary"])
def test_wide_condition_map_chain_matches_oracle(map_name, precision):
    from impersonator_amd.networks.generator import ImpersonatorGenerator
    from impersonator_amd.utils.nmr import SMPLRenderer
    s = helpers.scene()
    rest, faces = s["rest"], s["faces"]
    map_fn = synthetic.part_map_fn(rest, faces)[0] if map_name == "par" else _binary_table(faces.shape[0])
    nc = map_fn.shape[1]
    assert nc == {"par": 11, "binary": len(np.binary_repr(faces.shape[0]))}[map_name] and 3 + nc > 8
    G = ImpersonatorGenerator(bg_dim=4, src_dim=3 + nc, tsf_dim=3 + nc, repeat_num=6, max_batch=2, precision=precision)
    shapes = [(k, tuple(v.shape)) for k, v in G.state_dict().items()]
    sd = torch_ref.state_dict_from_numpy(synthetic.random_state_dict(shapes, seed=3, affine="random"))
    G.load_state_dict(sd)
    G = G.cuda()
    r = SMPLRenderer(image_size=256, faces=faces, map_fn=map_fn).cuda()
    try:
        faces_t, map_t = helpers.t(faces), helpers.t(map_fn)
        src_img, bg_img = helpers.t(s["src_img"]), helpers.t(s["bg_img"])
        # oracle: source side (models/imitator.py:95-143) and two target frames (:250-260)
        sf2v, sfim, _ = torch_ref.render_fim_wim(helpers.t(s["src_cam"]), helpers.t(s["src_verts"]), faces_t)
        scond = torch_ref.encode_fim(sfim, map_t)
        p2v = torch_ref.source_p2verts(sf2v)
        src_inputs = torch.cat([src_img, scond], dim=1)
        fr = torch_ref.transfer_frame(src_img, p2v, helpers.t(s["tgt_cam"]), helpers.t(s["tgt_verts"]), faces_t, map_t)
        with torch.no_grad():
            o_enc, o_res = torch_ref.encode_src(sd, src_inputs)
            o_pred, o_color, o_mask = torch_ref.imitator_forward(sd, o_enc, o_res, bg_img, fr["tsf_inputs"], fr["T"])
        # device: the renderer's transfer (cond has nc channels: no NHWC8 buffer, tsf_inputs is a plain NCHW cat)
        out = r.transfer(helpers.t(s["tgt_cam"]).cuda(), helpers.t(s["tgt_verts"]).cuda(), p2v.cuda(), src_img.cuda())
        assert tuple(out["tsf_inputs"].shape) == (2, 3 + nc, 256, 256) and out["tsf_inputs"].is_contiguous()
        assert torch.equal(out["fim"].cpu(), fr["fim"].to(torch.int32)) and torch.equal(out["cond"].cpu(), fr["cond"])
        assert helpers.maxdiff(out["T"], fr["T"])[0] <= 2e-6
        enc, res = G.encode_src(src_inputs.cuda())
        tol_f = {"fp32": 2e-5, "bf16x3": 4e-5}[precision]   # as tests/test_gpu_generator.py (the source stream is exact fp32 in both modes)
        for i, (a, b) in enumerate(zip(enc + res, o_enc + o_res)):
            d, where = helpers.maxdiff(a, b)
            assert d <= tol_f * max(1.0, float(b.abs().max())), ("source feature %d" % i, d, where)
        pred, color, mask = G.inference(enc, res, out["tsf_inputs"], out["T"], bg_img=bg_img.cuda())
        for name, a, b in (("color", color, o_color), ("mask", mask, o_mask), ("pred", pred, o_pred)):
            d, where = helpers.maxdiff(a, b)
            print("%s %s %s: L-inf %.3g" % (map_name, precision, name, d))
            assert d <= TOL_IMAGE, (name, d, where)
        # the NHWC8 fast path refuses a stream that is wider than eight channels instead of misreading it
        from impersonator_amd import _lib
        x8 = torch.zeros((2, 256, 256, 8), device="cuda").permute(0, 3, 1, 2)
        with pytest.raises((ValueError, _lib.LwgError)):
            G.inference(enc, res, x8, out["T"])
    finally:
        G.release()


def test_imitator_with_the_par_condition_map_matches_the_oracle():
    """The reference's own API on `--map_name par` (models/models.py:85-94: src_dim = tsf_dim = 3 + 11): Imitator.personalize ->
    transfer_params_by_smpl -> forward for four frames against the oracle's chain on the same posed vertices, and the two-lane
    pipeline against the sequential calls, bit for bit."""
    from impersonator_amd import demo
    B = 4
    opt = demo.default_opt(batch_size=B, image_size=256, map_name="par")
    imitator, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=B, seed=0, image_size=256, affine="random", opt=opt)
    assert imitator.generator.src_dim == 14 and imitator.render.map_fn.shape[1] == 11
    imitator.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
    smpls = torch.from_numpy(demo.synthetic_smpls(64, seed=0))[8:8 + 2 * B].cuda()
    imitator.first_cam = smpls[0:1, 0:3].clone()
    chunks = [(smpls[s:s + B], 8 + s) for s in range(0, 2 * B, B)]
    seq, infos = [], []
    for chunk, t in chunks:
        x = imitator.transfer_params_by_smpl(chunk, "smooth", t=t)
        info = imitator.tsf_info
        assert tuple(x.shape) == (B, 14, 256, 256)
        seq.append(imitator.forward(x, info["T"]).clone())
        infos.append({k: info[k].clone() for k in ("verts", "cam", "fim", "T", "cond")})
    piped = [p.clone() for _, p in imitator.predict_batches(iter(chunks), "smooth", lanes=2)]
    torch.cuda.synchronize()
    for a, b in zip(seq, piped):
        assert torch.equal(a, b)
    sd = {k: v.detach().cpu() for k, v in imitator.generator.state_dict().items()}
    faces_t, map_fn, si = imitator.render.faces.cpu(), imitator.render.map_fn.cpu(), imitator.src_info
    src_t, bg_t = torch.from_numpy(src_img)[None], torch.from_numpy(bg_img)[None]
    verts, cam = torch.cat([i["verts"] for i in infos]).cpu(), torch.cat([i["cam"] for i in infos]).cpu()
    with torch.no_grad():
        src = torch_ref.personalize(sd, src_t, si["cam"].cpu(), si["verts"].cpu(), faces_t, map_fn, ft_ks=imitator._opt.ft_ks)
        fr, pred = torch_ref.imitator_frames(sd, src, src_t, bg_t, cam, verts, faces_t, map_fn)
    assert torch.equal(src["fim"], imitator.src_info["fim"].cpu())
    assert torch.equal(torch.cat([i["fim"] for i in infos]).cpu(), fr["fim"])
    assert torch.equal(torch.cat([i["cond"] for i in infos]).cpu(), fr["cond"])
    err = float((torch.cat(seq).cpu() - pred).abs().max())
    print("Imitator, map_name par: L-inf over %d frames = %.3g" % (2 * B, err))
    assert err <= TOL_IMAGE
