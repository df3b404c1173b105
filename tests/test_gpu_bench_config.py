"""GPU parity of the BENCHMARKED configuration (BASELINE.json configs[1]: 256x256, batch 8, one source) -- the exact
imitator bench.py builds, in both conv arithmetics.  At batch 8 the 512->512 trunk runs the 128-wide bf16x3 tile on a
256-workgroup grid with the XCD band re-deal, a (kernel, shape, grid) combination the batch-2/4 tests never reach."""
import numpy as np
import pytest
import torch

from impersonator_amd import demo
from oracle import torch_ref

pytestmark = pytest.mark.gpu

BATCH, NBATCH = 8, 4
_ORACLE = {}


@pytest.fixture(scope="module")
def bench_imitator():
    imitator, src_smpl, src_img, bg_img = demo.build_synthetic_imitator(batch_size=BATCH, seed=0, image_size=256)
    imitator.personalize(src_img, src_smpl=src_smpl, bg_img=bg_img)
    smpls = torch.from_numpy(demo.synthetic_smpls(1024, seed=0))[8:8 + BATCH * NBATCH].cuda()   # bench.py's frames 8..39
    imitator.first_cam = torch.from_numpy(demo.synthetic_smpls(1024, seed=0))[0:1, 0:3].cuda()
    return imitator, src_img, bg_img, smpls


def _oracle(imitator, src_img, bg_img, verts, cam):
    """CPU oracle on the device-produced vertices (computed once: it does not depend on the conv arithmetic)."""
    if not _ORACLE:
        sd = {k: v.detach().cpu() for k, v in imitator.generator.state_dict().items()}
        faces_t, map_fn, si = imitator.render.faces.cpu(), imitator.render.map_fn.cpu(), imitator.src_info
        src_t, bg_t = torch.from_numpy(src_img)[None], torch.from_numpy(bg_img)[None]
        with torch.no_grad():
            src = torch_ref.personalize(sd, src_t, si["cam"].cpu(), si["verts"].cpu(), faces_t, map_fn,
                                        ft_ks=imitator._opt.ft_ks)
            fr, pred = torch_ref.imitator_frames(sd, src, src_t, bg_t, cam, verts, faces_t, map_fn)
        _ORACLE.update(src=src, fr=fr, pred=pred)
    return _ORACLE


@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
def test_batch8_bench_workload_matches_oracle(bench_imitator, precision):
    imitator, src_img, bg_img, smpls = bench_imitator
    imitator.generator.precision = precision
    chunks = [(smpls[s:s + BATCH], 8 + s) for s in range(0, BATCH * NBATCH, BATCH)]

    # (1) strictly sequential, batch 8
    seq, infos = [], []
    for chunk, t in chunks:
        x = imitator.transfer_params_by_smpl(chunk, "smooth", t=t)
        info = imitator.tsf_info
        seq.append(imitator.forward(x, info["T"]).clone())
        infos.append({k: info[k].clone() for k in ("verts", "cam", "fim", "T", "wim", "cond")})
    # (2) what bench.py times: two generator lanes + geometry stream
    piped = [p.clone() for _, p in imitator.predict_batches(iter(chunks), "smooth", lanes=2)]
    torch.cuda.synchronize()
    for a, b in zip(seq, piped):
        assert torch.equal(a, b)
    # (3) batch 1 == batch 8, bit for bit (every kernel variant adds an output's products in the same order)
    for k in (0, 5, 13, 31):
        x1 = imitator.transfer_params_by_smpl(smpls[k:k + 1], "smooth", t=8 + k)
        p1 = imitator.forward(x1, imitator.tsf_info["T"])
        assert torch.equal(imitator.tsf_info["fim"], infos[k // BATCH]["fim"][k % BATCH:k % BATCH + 1])
        assert torch.equal(p1, seq[k // BATCH][k % BATCH:k % BATCH + 1]), "frame %d: batch 1 != batch 8" % k

    verts = torch.cat([i["verts"] for i in infos]).cpu()
    cam = torch.cat([i["cam"] for i in infos]).cpu()
    o = _oracle(imitator, src_img, bg_img, verts, cam)
    assert torch.equal(o["src"]["fim"], imitator.src_info["fim"].cpu())
    fim = torch.cat([i["fim"] for i in infos]).cpu()
    assert torch.equal(fim, o["fr"]["fim"]), "%d face-index pixels differ" % int((fim != o["fr"]["fim"]).sum())
    assert torch.equal(torch.cat([i["cond"] for i in infos]).cpu(), o["fr"]["cond"])
    T = torch.cat([i["T"] for i in infos]).cpu()
    assert float((T - o["fr"]["T"]).abs().max()) <= 1e-6
    pred = torch.cat(seq).cpu()
    err = (pred - o["pred"]).abs().flatten(1).max(1).values      # per frame, all 32
    worst = int(err.argmax())
    assert float(err.max()) <= 1e-3, "frame %d: L-inf %g (%s)" % (worst, float(err.max()), precision)
    print("bench workload, %s: L-inf over 32 frames = %.3g" % (precision, float(err.max())))


def test_theta_to_image_chain_with_the_oracles_own_smpl(bench_imitator):
    """The other parity tests restart the oracle from the device's posed vertices.  Here the oracle runs its OWN SMPL
    (the reference's tensor-op formulation on the CPU, pinned in tests/test_host_logic.py) from the same theta, so the
    whole chain theta -> image is compared.  The two SMPL evaluations agree to ~1e-6; the reference's barycentric
    formula w = face_inv * (xi, yi, 1) (rasterize_cuda_kernel.cu:139-141) amplifies that by |face_inv| ~ 1e3..1e4, so
    the flow and the image legitimately move by more than the 1e-3 same-input bound, and a pixel centre within 1e-6 of
    a face edge may change owner.  Stated bounds: at most 2 face-index pixels per frame differ; on the frames whose
    face-index maps agree entirely the image stays within 1e-2."""
    from impersonator_amd.networks.batch_smpl import HumanModelRecovery, synthetic_smpl_params
    imitator, src_img, bg_img, smpls = bench_imitator
    imitator.generator.precision = "bf16x3"
    n = 16
    chunks = [(smpls[s:s + BATCH], 8 + s) for s in range(0, n, BATCH)]
    got_pred, got_fim = [], []
    for chunk, t in chunks:
        x = imitator.transfer_params_by_smpl(chunk, "smooth", t=t)
        got_fim.append(imitator.tsf_info["fim"].cpu())
        got_pred.append(imitator.forward(x, imitator.tsf_info["T"]).cpu())
    got_pred, got_fim = torch.cat(got_pred), torch.cat(got_fim)

    hmr = HumanModelRecovery(smpl_params=synthetic_smpl_params(0))          # CPU tensors -> the tensor-op formulation
    sd = {k: v.detach().cpu() for k, v in imitator.generator.state_dict().items()}
    faces_t, map_fn = imitator.render.faces.cpu(), imitator.render.map_fn.cpu()
    src_t, bg_t = torch.from_numpy(src_img)[None], torch.from_numpy(bg_img)[None]
    all_smpls = torch.from_numpy(demo.synthetic_smpls(1024, seed=0))
    src_smpl = torch.from_numpy(demo.synthetic_smpls(1, seed=1))
    src_smpl[:, 3:75] = 0
    with torch.no_grad():
        si = hmr.get_details(src_smpl)
        src = torch_ref.personalize(sd, src_t, si["cam"], si["verts"], faces_t, map_fn, ft_ks=imitator._opt.ft_ks)
        chunk = all_smpls[8:8 + n]
        cam = si["cam"].expand(n, -1).clone()
        cam[:, 1:] += chunk[:, 1:3] - all_smpls[0:1, 1:3]
        info = hmr.get_details(torch.cat([cam, chunk[:, 3:75], si["shape"].expand(n, -1)], 1))
        fr, ref = torch_ref.imitator_frames(sd, src, src_t, bg_t, info["cam"], info["verts"], faces_t, map_fn)
    mism = (got_fim != fr["fim"]).flatten(1).sum(1)
    assert int(mism.max()) <= 2 and int(mism.sum()) <= n, "face-index pixels differing per frame: %s" % mism.tolist()
    agree = mism == 0
    assert int(agree.sum()) >= n // 2
    err = (got_pred - ref).abs().flatten(1).max(1).values
    assert float(err[agree].max()) <= 1e-2, "theta chain, frames with identical face-index maps: L-inf %g" % float(err[agree].max())
    print("theta chain: %d face-index pixels differ over %d frames; L-inf on the %d agreeing frames %.3g, all frames %.3g"
          % (int(mism.sum()), n, int(agree.sum()), float(err[agree].max()), float(err.max())))
