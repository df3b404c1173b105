"""oracle/torch_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU (fp32, plain torch ops) restatement of the reference's per-frame hot path, function by function.
It exists because /root/reference cannot travel to the GPU box; tests/test_oracle_vs_reference.py
pins every function here against the real reference imported through oracle/reference_loader.py
(in the build container), and tests/golden/*.npz hold outputs of the real reference.

The conv / instance-norm / grid_sample / interpolate arithmetic itself is PyTorch's (the reference
calls torch for it: networks/generator.py:13-17,80-133,307,313; pinned de facto to torch 2.10 CPU,
SURVEY.md section 8c).  grid_sample therefore follows torch>=1.3's default align_corners=False (hazard H1).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import raster as _raster

EYE_Z = -(1.0 / math.tan(math.radians(30.0)) + 1.0)  # utils/nmr.py:177 (viewing_angle=30)


# --------------------------------------------------------------------------- geometry (a3-a9)
def look_at(vertices, eye, at=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)):
    """neural_renderer/look_at.py:6-62 for the general case (used only by the KAT test)."""
    bs = vertices.shape[0]
    eye = torch.as_tensor(eye, dtype=torch.float32).reshape(1, 3).repeat(bs, 1)
    at = torch.as_tensor(at, dtype=torch.float32).reshape(1, 3).repeat(bs, 1)
    up = torch.as_tensor(up, dtype=torch.float32).reshape(1, 3).repeat(bs, 1)
    z = F.normalize(at - eye, eps=1e-5)
    x = F.normalize(torch.cross(up, z, dim=1), eps=1e-5)
    y = F.normalize(torch.cross(z, x, dim=1), eps=1e-5)
    r = torch.stack((x, y, z), dim=1)
    return torch.matmul(vertices - eye[:, None, :], r.transpose(1, 2))


def project_vertices(verts, cam):
    """utils/nmr.py:10-28 (orthographic_proj_withz_idrot) + nmr.py:271 (y flip) + look_at with the
    renderer's fixed eye (nmr.py:177,273): rotation is the identity, z is shifted by -eye_z."""
    scale = cam[:, 0].reshape(-1, 1, 1)
    trans = cam[:, 1:3].reshape(cam.shape[0], 1, 2)
    xy = scale * (verts[:, :, :2] + trans)
    out = torch.cat((xy, verts[:, :, 2:3]), 2)
    out[:, :, 1] *= -1
    eye = torch.tensor([0.0, 0.0, EYE_Z], dtype=torch.float32)
    return out - eye[None, None, :]


def vertices_to_faces(vertices, faces_idx):
    """neural_renderer/vertices_to_faces.py:4-22.  faces_idx: (nf,3) or (bs,nf,3) int."""
    bs = vertices.shape[0]
    if faces_idx.dim() == 2:
        faces_idx = faces_idx[None].expand(bs, -1, -1)
    idx = faces_idx.long()
    return torch.stack([vertices[b][idx[b]] for b in range(bs)], 0)


def render_fim_wim(cam, verts, faces_idx, image_size=256):
    """utils/nmr.py:263-278.  Returns (f2verts, fim, wim); near/far are the rasteriser defaults
    0.1/100 (rasterize.py:8-13) because nmr.py:277 passes only three arguments."""
    f2verts = vertices_to_faces(project_vertices(verts, cam), faces_idx)
    fim, wim, _ = _raster.rasterize_fim_wim(f2verts.numpy(), image_size, 0.1, 100.0)
    return f2verts, torch.from_numpy(fim), torch.from_numpy(wim)


def encode_fim(fim, map_fn, transpose=True):
    """utils/nmr.py:328-341: table lookup; fim == -1 wraps to the last (background) row."""
    enc = map_fn[fim.long()]
    return enc.permute(0, 3, 1, 2) if transpose else enc


def cal_bc_transform(src_f2pts, dst_fims, dst_wims):
    """utils/nmr.py:617-659.  src_f2pts (bs|1, nf, 3, 2); a single source is shared by every frame
    of a batch (the reference runs batch 1; batching target frames over one source is this
    build's only extension)."""
    bs, h, w = dst_fims.shape
    T = -2 * torch.ones((bs, h * w, 2), dtype=torch.float32)
    for i in range(bs):
        pts = src_f2pts[i if src_f2pts.shape[0] > 1 else 0]
        fi = dst_fims[i].long().reshape(-1)
        wi = dst_wims[i].reshape(-1, 3)
        m = fi != -1
        T[i, m] = (pts[fi[m]] * wi[m][:, :, None]).sum(dim=1)
    return T.view(bs, h, w, 2)


def grid_sample(x, grid, align_corners=False):
    """F.grid_sample as called at models/imitator.py:259 and networks/generator.py:313."""
    return F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=align_corners)


def resize_trans(T, h, w):
    """networks/generator.py:303-310."""
    t = F.interpolate(T.permute(0, 3, 1, 2), size=(h, w), mode="bilinear", align_corners=True)
    return t.permute(0, 2, 3, 1)


def morph(mask, ks, mode="erode"):
    """utils/util.py:73-89."""
    pad = ks // 2
    kernel = torch.ones(1, 1, ks, ks)
    if mode == "erode":
        out = F.conv2d(F.pad(mask, [pad] * 4, value=1.0), kernel)
        return (out == ks * ks).float()
    out = F.conv2d(F.pad(mask, [pad] * 4, value=0.0), kernel)
    return (out >= 1).float()


# --------------------------------------------------------------------------- generator (a12-a15)
def _in_relu(x, sd, prefix, relu=True):
    x = F.instance_norm(x, weight=sd[prefix + ".weight"], bias=sd[prefix + ".bias"], eps=1e-5)
    return F.relu(x) if relu else x


def _encoder(x, sd, p, i):
    """ResUnetGenerator.encoders[i]: conv -> IN -> ReLU (networks/generator.py:79-92)."""
    if i == 0:
        x = F.conv2d(x, sd["%s.encoders.0.0.weight" % p], padding=3)
    else:
        x = F.conv2d(x, sd["%s.encoders.%d.0.weight" % (p, i)], stride=2, padding=1)
    return _in_relu(x, sd, "%s.encoders.%d.1" % (p, i))


def _resblock(x, sd, p, i):
    """ResidualBlock (networks/generator.py:8-20)."""
    q = "%s.resnets.%d.main" % (p, i)
    y = F.conv2d(x, sd[q + ".0.weight"], padding=1)
    y = _in_relu(y, sd, q + ".1")
    y = F.conv2d(y, sd[q + ".3.weight"], padding=1)
    y = _in_relu(y, sd, q + ".4", relu=False)
    return x + y


def _decode(x, enc_outs, sd, p, n_down=3):
    """ResUnetGenerator.decode (networks/generator.py:173-181)."""
    d = x
    for i in range(n_down):
        d = F.conv_transpose2d(d, sd["%s.decoders.%d.0.weight" % (p, i)], stride=2, padding=1, output_padding=1)
        d = _in_relu(d, sd, "%s.decoders.%d.1" % (p, i))
        d = torch.cat([enc_outs[n_down - 1 - i], d], dim=1)
        d = F.conv2d(d, sd["%s.skippers.%d.0.weight" % (p, i)], padding=1)
        d = _in_relu(d, sd, "%s.skippers.%d.1" % (p, i))
    return d


def _regress(x, sd, p):
    """ResUnetGenerator.regress (networks/generator.py:183-184); 'attetion_reg' is the reference's spelling."""
    img = torch.tanh(F.conv2d(x, sd["%s.img_reg.0.weight" % p], padding=3))
    mask = torch.sigmoid(F.conv2d(x, sd["%s.attetion_reg.0.weight" % p], padding=3))
    return img, mask


def encode_src(sd, src_inputs, n_down=3, repeat_num=6):
    """ImpersonatorGenerator.encode_src -> ResUnetGenerator.inference (generator.py:213-214,136-147)."""
    x = src_inputs
    enc = []
    for i in range(n_down + 1):
        x = _encoder(x, sd, "src_model", i)
        enc.append(x)
    res = []
    for i in range(repeat_num):
        x = _resblock(x, sd, "src_model", i)
        res.append(x)
    return enc, res


def generator_inference(sd, src_enc, src_res, tsf_inputs, T, n_down=3, repeat_num=6, align_corners=False):
    """ImpersonatorGenerator.inference (networks/generator.py:277-301) with the Liquid Warping Block
    transform/stn/resize_trans (generator.py:303-320)."""
    def lwb(feat, Tfull):
        h, w = feat.shape[2:]
        if feat.shape[0] != Tfull.shape[0]:
            feat = feat.expand(Tfull.shape[0], -1, -1, -1)
        return grid_sample(feat, resize_trans(Tfull, h, w), align_corners)

    x = _encoder(tsf_inputs, sd, "tsf_model", 0)
    enc = [x]
    for i in range(1, n_down + 1):
        x = _encoder(x, sd, "tsf_model", i) + lwb(src_enc[i], T)
        enc.append(x)
    for i in range(repeat_num):
        x = _resblock(x, sd, "tsf_model", i) + lwb(src_res[i], T)
    return _regress(_decode(x, enc, sd, "tsf_model", n_down), sd, "tsf_model")


def generator_swap(sd, tsf_inputs, enc12, enc21, res12, res21, T12, T21, n_down=3, repeat_num=6,
                   align_corners=False):
    """ImpersonatorGenerator.swap (networks/generator.py:245-275): two warps per level."""
    def lwb(feat, Tfull):
        h, w = feat.shape[2:]
        if feat.shape[0] != Tfull.shape[0]:
            feat = feat.expand(Tfull.shape[0], -1, -1, -1)
        return grid_sample(feat, resize_trans(Tfull, h, w), align_corners)

    x = _encoder(tsf_inputs, sd, "tsf_model", 0)
    enc = [x]
    for i in range(1, n_down + 1):
        x = _encoder(x, sd, "tsf_model", i) + lwb(enc12[i], T12) + lwb(enc21[i], T21)
        enc.append(x)
    for i in range(repeat_num):
        x = _resblock(x, sd, "tsf_model", i) + lwb(res12[i], T12) + lwb(res21[i], T21)
    return _regress(_decode(x, enc, sd, "tsf_model", n_down), sd, "tsf_model")


def imitator_forward(sd, src_enc, src_res, bg_img, tsf_inputs, T, align_corners=False):
    """Imitator.forward (models/imitator.py:326-336) without front_warp."""
    color, mask = generator_inference(sd, src_enc, src_res, tsf_inputs, T, align_corners=align_corners)
    return mask * bg_img + (1 - mask) * color, color, mask


def transfer_frame(src_img, src_p2verts, cam, verts, faces_idx, map_fn, image_size=256, align_corners=False):
    """models/imitator.py:250-260 given posed vertices: render -> cond -> T -> warped source -> tsf_inputs."""
    f2verts, fim, wim = render_fim_wim(cam, verts, faces_idx, image_size)
    cond = encode_fim(fim, map_fn)
    T = cal_bc_transform(src_p2verts, fim, wim)
    img = src_img if src_img.shape[0] == T.shape[0] else src_img.expand(T.shape[0], -1, -1, -1)
    tsf_img = grid_sample(img, T, align_corners)
    return dict(f2verts=f2verts, fim=fim, wim=wim, cond=cond, T=T, tsf_img=tsf_img,
                tsf_inputs=torch.cat([tsf_img, cond], dim=1))


def source_p2verts(f2verts):
    """models/imitator.py:105-107 (hazard H9): xy of the source face vertices with y negated."""
    p = f2verts[:, :, :, 0:2].clone()
    p[:, :, :, 1] *= -1
    return p


def personalize(sd, src_img, src_cam, src_verts, faces_idx, map_fn, ft_ks=3, image_size=256):
    """Imitator.personalize (models/imitator.py:95-143) given the posed source vertices and an explicit background:
    source face-index map, p2verts (H9), cond, the eroded front mask and the cached source features."""
    sf2v, sfim, _ = render_fim_wim(src_cam, src_verts, faces_idx, image_size)
    p2v = source_p2verts(sf2v)
    scond = encode_fim(sfim, map_fn)
    ft = 1 - morph(scond[:, -1:], ft_ks, "erode")
    enc, res = encode_src(sd, torch.cat([src_img * ft, scond], 1))
    return dict(fim=sfim, p2verts=p2v, cond=scond, enc=enc, res=res)


def imitator_frames(sd, src, src_img, bg_img, cam, verts, faces_idx, map_fn, image_size=256, align_corners=False,
                    chunk=8):
    """Imitator.transfer_params_by_smpl after the SMPL stage + Imitator.forward (models/imitator.py:250-260, 326-336) for
    a sequence of posed meshes; `src` = personalize(...).  Returns (per-frame dict of fim / T / tsf_inputs, preds)."""
    outs, preds = [], []
    with torch.no_grad():
        for s in range(0, verts.shape[0], chunk):
            fr = transfer_frame(src_img, src["p2verts"], cam[s:s + chunk], verts[s:s + chunk], faces_idx, map_fn,
                                image_size, align_corners)
            outs.append(fr)
            preds.append(imitator_forward(sd, src["enc"], src["res"], bg_img, fr["tsf_inputs"], fr["T"], align_corners)[0])
    keys = ("fim", "T", "tsf_inputs", "cond", "wim")
    return {k: torch.cat([o[k] for o in outs], 0) for k in keys}, torch.cat(preds, 0)


def cal_head_bbox(kps, image_size):
    """models/impersonator_trainer.py:89-130 (NECK_IDS = 12; kps (N,19,2) in [-1,1]) -> (N,4) long [min_x,max_x,min_y,max_y]."""
    kps = (kps + 1) / 2.0
    zeros, ones = torch.zeros_like(kps[:, 12, 0]), torch.ones_like(kps[:, 12, 0])
    min_x = torch.max(torch.min(kps[:, 12:, 0] - 0.05, dim=1)[0], zeros)
    max_x = torch.min(torch.max(kps[:, 12:, 0] + 0.05, dim=1)[0], ones)
    min_y = torch.max(torch.min(kps[:, 12:, 1] - 0.05, dim=1)[0], zeros)
    max_y = torch.min(torch.max(kps[:, 12:, 1], dim=1)[0], ones)
    return torch.stack([(v * image_size).long() for v in (min_x, max_x, min_y, max_y)], dim=1)


def cal_body_bbox(kps, image_size, factor=1.2):
    """models/impersonator_trainer.py:132-170."""
    kps = (kps + 1) / 2.0
    zeros, ones = torch.zeros(kps.shape[0]), torch.ones(kps.shape[0])
    out = []
    for c in (0, 1):
        lo, hi = kps[:, :, c].min(dim=1)[0], kps[:, :, c].max(dim=1)[0]
        mid, ext = (lo + hi) / 2, (hi - lo) * factor
        out += [torch.max(zeros, mid - ext / 2), torch.min(ones, mid + ext / 2)]
    return torch.stack([(v * image_size).long() for v in out], dim=1)


def body_recovery_flow(get_details, faces_idx, map_fn, src_img, ref_img, src_smpl, ref_smpl, image_size=256, bg_both=False,
                       align_corners=False):
    """BodyRecoveryFlow.forward (models/impersonator_trainer.py:44-87): what the trainer's set_input derives from a pair
    of images and SMPL vectors.  `get_details` = HumanModelRecovery.get_details on CPU tensors."""
    src_info, ref_info = get_details(src_smpl), get_details(ref_smpl)
    src_f2verts, src_fim, _ = render_fim_wim(src_info['cam'], src_info['verts'], faces_idx, image_size)
    src_f2pts = source_p2verts(src_f2verts)
    src_cond = encode_fim(src_fim, map_fn)
    src_crop_mask = morph(src_cond[:, -1:], 3, 'erode')
    _, ref_fim, ref_wim = render_fim_wim(ref_info['cam'], ref_info['verts'], faces_idx, image_size)
    ref_cond = encode_fim(ref_fim, map_fn)
    T = cal_bc_transform(src_f2pts, ref_fim, ref_wim)
    syn_img = grid_sample(src_img, T, align_corners)
    input_G_src = torch.cat([src_img * (1 - src_crop_mask), src_cond], dim=1)
    input_G_tsf = torch.cat([syn_img, ref_cond], dim=1)
    src_bg_mask = morph(src_cond[:, -1:], 15, 'erode')
    input_G_src_bg = torch.cat([src_img * src_bg_mask, src_bg_mask], dim=1)
    input_G_tsf_bg = None
    if bg_both:
        ref_bg_mask = morph(ref_cond[:, -1:], 15, 'erode')
        input_G_tsf_bg = torch.cat([ref_img * ref_bg_mask, ref_bg_mask], dim=1)
    tsf_crop_mask = morph(ref_cond[:, -1:], 3, 'erode')
    return (input_G_src_bg, input_G_tsf_bg, input_G_src, input_G_tsf, T, src_crop_mask, tsf_crop_mask,
            cal_head_bbox(ref_info['j2d'], image_size), cal_body_bbox(ref_info['j2d'], image_size))


def state_dict_from_numpy(sd_np):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}


# --------------------------------------------------------------------------- background inpaintor (a16)
# (in_ch, out_ch, kernel, stride, dilation, up2x, activation) per gated layer of InpaintSANet(c_dim)
# (networks/inpaintor.py:110-176); padding = get_pad(...) reduces to dilation*(k-1)//2 for stride 1 and 1 for k4/s2.
def inpaint_layers(c_dim=4, cnum=32):
    c = cnum
    coarse = [(c_dim, c, 5, 1, 1, 0, 1), (c, 2 * c, 4, 2, 1, 0, 1), (2 * c, 2 * c, 3, 1, 1, 0, 1),
              (2 * c, 4 * c, 4, 2, 1, 0, 1), (4 * c, 4 * c, 3, 1, 1, 0, 1), (4 * c, 4 * c, 3, 1, 1, 0, 1),
              (4 * c, 4 * c, 3, 1, 2, 0, 1), (4 * c, 4 * c, 3, 1, 4, 0, 1), (4 * c, 4 * c, 3, 1, 8, 0, 1),
              (4 * c, 4 * c, 3, 1, 16, 0, 1), (4 * c, 4 * c, 3, 1, 1, 0, 1), (4 * c, 4 * c, 3, 1, 1, 0, 1),
              (4 * c, 2 * c, 3, 1, 1, 1, 1), (2 * c, 2 * c, 3, 1, 1, 0, 1), (2 * c, c, 3, 1, 1, 1, 1),
              (c, c // 2, 3, 1, 1, 0, 1), (c // 2, 3, 3, 1, 1, 0, 0)]
    refine_conv = [(c_dim, c, 5, 1, 1, 0, 1), (c, c, 4, 2, 1, 0, 1), (c, 2 * c, 3, 1, 1, 0, 1),
                   (2 * c, 2 * c, 4, 2, 1, 0, 1), (2 * c, 4 * c, 3, 1, 1, 0, 1), (4 * c, 4 * c, 3, 1, 1, 0, 1),
                   (4 * c, 4 * c, 3, 1, 1, 0, 1), (4 * c, 4 * c, 3, 1, 2, 0, 1), (4 * c, 4 * c, 3, 1, 4, 0, 1),
                   (4 * c, 4 * c, 3, 1, 8, 0, 1), (4 * c, 4 * c, 3, 1, 16, 0, 1)]
    refine_up = [(4 * c, 4 * c, 3, 1, 1, 0, 1), (4 * c, 4 * c, 3, 1, 1, 0, 1), (4 * c, 2 * c, 3, 1, 1, 1, 1),
                 (2 * c, 2 * c, 3, 1, 1, 0, 1), (2 * c, c, 3, 1, 1, 1, 1), (c, c // 2, 3, 1, 1, 0, 1),
                 (c // 2, 3, 3, 1, 1, 0, 0)]
    return coarse, refine_conv, refine_up


def _gated(x, sd, prefix, spec):
    """GatedConv2dWithActivation / GatedDeConv2dWithActivation in eval mode (networks/inpaintor.py:12-68)."""
    cin, cout, k, stride, dil, up, act = spec
    if up:
        x = F.interpolate(x, scale_factor=2)
        prefix = prefix + ".conv2d"
    pad = 1 if (k == 4 and stride == 2) else dil * (k - 1) // 2
    f = F.conv2d(x, sd[prefix + ".conv2d.weight"], sd[prefix + ".conv2d.bias"], stride, pad, dil)
    g = F.conv2d(x, sd[prefix + ".mask_conv2d.weight"], sd[prefix + ".mask_conv2d.bias"], stride, pad, dil)
    y = (F.leaky_relu(f, 0.2) if act else f) * torch.sigmoid(g)
    return F.batch_norm(y, sd[prefix + ".batch_norm2d.running_mean"], sd[prefix + ".batch_norm2d.running_var"],
                        sd[prefix + ".batch_norm2d.weight"], sd[prefix + ".batch_norm2d.bias"], False, 0.0, 1e-5)


def _self_attention(x, sd, p):
    """SelfAttention (networks/inpaintor.py:71-107)."""
    b, c, w, h = x.shape
    q = F.conv2d(x, sd[p + ".query_conv.weight"], sd[p + ".query_conv.bias"]).view(b, -1, w * h).permute(0, 2, 1)
    k = F.conv2d(x, sd[p + ".key_conv.weight"], sd[p + ".key_conv.bias"]).view(b, -1, w * h)
    attn = torch.softmax(torch.bmm(q, k), dim=-1)
    v = F.conv2d(x, sd[p + ".value_conv.weight"], sd[p + ".value_conv.bias"]).view(b, -1, w * h)
    out = torch.bmm(v, attn.permute(0, 2, 1)).view(b, c, w, h)
    return sd[p + ".gamma"] * out + x


def inpaint_forward(sd, imgs, masks, c_dim=4):
    """InpaintSANet.forward (networks/inpaintor.py:178-202) -> (coarse_x, x, comp_imgs)."""
    coarse, refine_conv, refine_up = inpaint_layers(c_dim)
    x = torch.cat([imgs * (1 - masks) + masks, masks], dim=1)
    for i, spec in enumerate(coarse):
        x = _gated(x, sd, "coarse_net.%d" % i, spec)
    coarse_x = torch.clamp(x, -1., 1.)
    x = torch.cat([imgs * (1 - masks) + coarse_x * masks, masks], dim=1)
    for i, spec in enumerate(refine_conv):
        x = _gated(x, sd, "refine_conv_net.%d" % i, spec)
    x = _self_attention(x, sd, "refine_attn")
    for i, spec in enumerate(refine_up):
        x = _gated(x, sd, "refine_upsample_net.%d" % i, spec)
    x = torch.clamp(x, -1., 1.)
    return coarse_x, x, x * masks + imgs * (1 - masks)


# ------------------------------------------------------------------------------------------------
# Training, first slice: PatchGAN discriminator update.  CPU restatement with torch autograd.

def discriminator_conv_keys(n_layers):
    """nn.Sequential indices of the convs of PatchDiscriminator (networks/discriminator.py:29-49)."""
    idx, keys = 0, []
    for l in range(n_layers + 2):
        keys.append(idx)
        idx += 2 if l == 0 else 3
    return keys


def discriminator_forward(sd, x, n_layers=4):
    """PatchDiscriminator.forward (networks/discriminator.py:29-57) with norm_type='instance' (affine=False, eps 1e-5),
    use_sigmoid=False: conv4x4 s2 + LeakyReLU(0.2); (n_layers-1) x [conv4x4 s2, IN, LeakyReLU]; [conv4x4 s1, IN, LeakyReLU];
    conv4x4 s1 -> 1 channel.  Differentiable wrt the tensors in `sd`."""
    keys = discriminator_conv_keys(n_layers)
    for l, k in enumerate(keys):
        stride = 2 if l < n_layers else 1
        x = F.conv2d(x, sd["model.%d.weight" % k], sd["model.%d.bias" % k], stride=stride, padding=1)
        if l == len(keys) - 1:
            break
        if l > 0:
            x = F.instance_norm(x, eps=1e-5)
        x = F.leaky_relu(x, 0.2)
    return x


def discriminator_loss(sd, real, fake, n_layers=4):
    """ImpersonatorTrainer._optimize_D / _compute_loss_D (models/impersonator_trainer.py:396-414), lambda_D_prob = 1."""
    d_real = discriminator_forward(sd, real, n_layers)
    d_fake = discriminator_forward(sd, fake, n_layers)
    return torch.mean((d_real - 1) ** 2) + torch.mean((d_fake + 1) ** 2)


def discriminator_train_steps(sd, batches, n_layers=4, lr=0.0002, betas=(0.5, 0.999), eps=1e-8):
    """`len(batches)` updates with torch.optim.Adam (impersonator_trainer.py:231-232, train_options.py:36-38).
    Returns (losses, gradients of the FIRST step, final parameters)."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(params.values()), lr=lr, betas=betas, eps=eps)
    losses, first = [], None
    for real, fake in batches:
        opt.zero_grad()
        loss = discriminator_loss(params, real, fake, n_layers)
        loss.backward()
        if first is None:
            first = {k: v.grad.detach().clone() for k, v in params.items()}
        opt.step()
        losses.append(float(loss.detach()))
    return losses, first, {k: v.detach().clone() for k, v in params.items()}


def create_meshgrid(image_size):
    """utils/nmr.py:490-504: the identity sampling grid (is, is, 2), x along the last image axis."""
    factor = (torch.arange(0, image_size, dtype=torch.float32) / (image_size - 1) - 0.5) * 2
    xv, yv = torch.meshgrid([factor, factor], indexing="ij")
    return torch.stack([yv, xv], dim=-1)


def swapper_personalize(sd, img, cam, verts, faces_idx, map_fn, part_fn, bg_ks=13, ft_ks=3, image_size=256):
    """Swapper.personalize (models/swapper.py:99-165) given the posed vertices, --bg_model ORIGINAL (the generator's own
    BGNet inpaints the background)."""
    f2v, fim, wim = render_fim_wim(cam, verts, faces_idx, image_size)
    cond = encode_fim(fim, map_fn)
    part = encode_fim(fim, part_fn)
    bg_mask = morph(cond[:, -1:], bg_ks, "erode")
    bg = bgnet_forward(sd, torch.cat([img * bg_mask, bg_mask], dim=1))
    ft = 1 - morph(cond[:, -1:], ft_ks, "erode")
    enc, res = encode_src(sd, torch.cat([img * ft, cond], 1))
    return dict(fim=fim, wim=wim, cond=cond, part=part, p2verts=source_p2verts(f2v), img=img, bg=bg, enc=enc, res=res)


def swapper_swap(sd, src, tgt, part_faces, selected_ids=(1, 2, 3, 4, 5, 6, 7, 8, 9), all_ids=tuple(range(10))):
    """Swapper.swap + calculate_trans + forward (models/swapper.py:198-271), front_warp off; src / tgt as
    swapper_personalize returns them.  -> dict(T11, T21, tsf_inputs, preds, mask)."""
    left_ids = [i for i in all_ids if i not in selected_ids]
    part_mask = (torch.sum(src["part"][:, list(selected_ids)], dim=1) != 0)
    left_mask = torch.sum(src["part"][:, left_ids], dim=1).bool()
    left_faces = sorted(set(f for i in left_ids for f in part_faces[i]))
    T11 = create_meshgrid(src["img"].shape[-1]).clone()
    T11[~left_mask[0]] = -2
    T11 = T11[None]
    f2p = tgt["p2verts"].clone()
    f2p[0, left_faces] = -2
    T21 = cal_bc_transform(f2p, src["fim"], src["wim"]).clamp(-2, 2)
    tsf_img = grid_sample(tgt["img"], T21) * part_mask[:, None].float() + grid_sample(src["img"], T11) * left_mask[:, None].float()
    x = torch.cat([tsf_img, src["cond"]], 1)
    color, mask = generator_swap(sd, x, tgt["enc"], src["enc"], tgt["res"], src["res"], T21, T11)
    return dict(T11=T11, T21=T21, left_mask=left_mask, tsf_inputs=x, preds=mask * src["bg"] + (1 - mask) * color, mask=mask)


def viewer_view(sd, src, tsf_mesh, cam, faces_idx, map_fn, bg_replace=False, image_size=256):
    """Viewer.view + forward (models/viewer.py:273-311) after rotate_trans, front_warp off."""
    fr = transfer_frame(src["img"], src["p2verts"], cam, tsf_mesh, faces_idx, map_fn, image_size)
    bg = src["bg"] if bg_replace else torch.zeros_like(src["bg"])
    return fr, imitator_forward(sd, src["enc"], src["res"], bg, fr["tsf_inputs"], fr["T"])[0]


def bgnet_forward(sd, x, repeat=6, n_down=3):
    """ResNetGenerator.forward (networks/generator.py:23-65) as ImpersonatorGenerator builds it (k_size=3, n_down=3):
    conv7-IN-ReLU, 3 x [conv3 s2-IN-ReLU], `repeat` residual blocks, 3 x [convT3 s2-IN-ReLU], conv7, tanh.
    `sd` holds the 'bg_model.model.N...' entries of the generator's state_dict."""
    P = "bg_model.model."

    def cin(x, i, stride, pad, transposed=False):
        w = sd[P + "%d.weight" % i]
        x = (F.conv_transpose2d(x, w, stride=stride, padding=pad, output_padding=1) if transposed
             else F.conv2d(x, w, stride=stride, padding=pad))
        return F.relu(F.instance_norm(x, weight=sd[P + "%d.weight" % (i + 1)], bias=sd[P + "%d.bias" % (i + 1)], eps=1e-5))

    x = cin(x, 0, 1, 3)
    for i in range(n_down):
        x = cin(x, 3 + 3 * i, 2, 1)
    r0 = 3 + 3 * n_down
    for i in range(repeat):
        q = P + "%d.main." % (r0 + i)
        y = F.conv2d(x, sd[q + "0.weight"], padding=1)
        y = F.relu(F.instance_norm(y, weight=sd[q + "1.weight"], bias=sd[q + "1.bias"], eps=1e-5))
        y = F.conv2d(y, sd[q + "3.weight"], padding=1)
        x = x + F.instance_norm(y, weight=sd[q + "4.weight"], bias=sd[q + "4.bias"], eps=1e-5)
    u0 = r0 + repeat
    for i in range(n_down):
        x = cin(x, u0 + 3 * i, 2, 1, transposed=True)
    return torch.tanh(F.conv2d(x, sd[P + "%d.weight" % (u0 + 3 * n_down)], padding=3))


def generator_infer_front(sd, src_inputs, tsf_inputs, T, align_corners=False):
    """ImpersonatorGenerator.infer_front (networks/generator.py:216-243): one source per sample, both streams decoded."""
    enc, res = encode_src(sd, src_inputs)
    tsf_img, tsf_mask = generator_inference(sd, enc, res, tsf_inputs, T, align_corners=align_corners)
    src_img, src_mask = _regress(_decode(res[-1], enc, sd, "src_model"), sd, "src_model")
    return src_img, src_mask, tsf_img, tsf_mask


def generator_forward(sd, bg_inputs, src_inputs, tsf_inputs, T, align_corners=False):
    """ImpersonatorGenerator.forward (networks/generator.py:204-211)."""
    return (bgnet_forward(sd, bg_inputs),) + generator_infer_front(sd, src_inputs, tsf_inputs, T, align_corners)


# ------------------------------------------------------------------------------------------------
# Training, generator side: ImpersonatorTrainer.forward + _optimize_G + Adam, CPU autograd.

G_TRAIN_DEFAULTS = dict(lambda_D_prob=1.0, lambda_rec=10.0, lambda_tsf=10.0, lambda_mask=0.1, lambda_mask_smooth=1e-5,
                        lr_G=0.0002, G_adam_b1=0.5, G_adam_b2=0.999)   # options/train_options.py:33-45


# torchvision's vgg19().features: (index, out channels) of the convs up to relu5_1, 'M' = MaxPool2d(2, 2)
VGG19_CFG = [(0, 64), (2, 64), "M", (5, 128), (7, 128), "M", (10, 256), (12, 256), (14, 256), (16, 256), "M",
             (19, 512), (21, 512), (23, 512), (25, 512), "M", (28, 512)]
VGG19_TAPS = (0, 5, 10, 19, 28)   # convs whose ReLU output is a slice output (networks/networks.py:137-155: slice_ids 2,7,12,21,30)
VGG_LOSS_WEIGHTS = (1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)   # networks/networks.py:179


def vgg19_features(vsd, x):
    """Vgg19.forward (networks/networks.py:162-169) on a state_dict in torchvision's naming (features.N.weight / .bias):
    the five slice outputs relu1_1, relu2_1, relu3_1, relu4_1, relu5_1."""
    outs = []
    for item in VGG19_CFG:
        if item == "M":
            x = F.max_pool2d(x, 2, 2)
            continue
        i = item[0]
        x = F.relu(F.conv2d(x, vsd["features.%d.weight" % i], vsd["features.%d.bias" % i], padding=1))
        if i in VGG19_TAPS:
            outs.append(x)
    return outs


def vgg_loss(vsd, x, y):
    """VGGLoss.forward (networks/networks.py:181-186)."""
    fx, fy = vgg19_features(vsd, x), vgg19_features(vsd, y)
    return sum(w * F.l1_loss(a, b.detach()) for w, a, b in zip(VGG_LOSS_WEIGHTS, fx, fy))


# Sphere20a (networks/facenet.py:200-262): stage name, channels, residual units; every stage opens with a stride-2 conv
SPHERE20A_STAGES = (("1", 64, 1), ("2", 128, 2), ("3", 256, 4), ("4", 512, 1))
FACE_HW = (112, 96)   # networks/networks.py:221


def sphere20a_features(fsd, x):
    """Sphere20a.forward (networks/facenet.py:264-290): the four stage outputs and the fc5 embedding."""
    def unit(x, name):
        return F.prelu(F.conv2d(x, fsd["conv%s.weight" % name], fsd["conv%s.bias" % name], stride=1, padding=1),
                       fsd["relu%s.weight" % name])
    feats = []
    for st, _, units in SPHERE20A_STAGES:
        x = F.prelu(F.conv2d(x, fsd["conv%s_1.weight" % st], fsd["conv%s_1.bias" % st], stride=2, padding=1),
                    fsd["relu%s_1.weight" % st])
        for u in range(units):
            x = x + unit(unit(x, "%s_%d" % (st, 2 + 2 * u)), "%s_%d" % (st, 3 + 2 * u))
        feats.append(x)
    feats.append(F.linear(x.reshape(x.shape[0], -1), fsd["fc5.weight"], fsd["fc5.bias"]))
    return feats


def crop_head_bbox(imgs, bboxs):
    """FaceLoss.crop_head_bbox (networks/networks.py:290-312): bbox rows are (min_x, max_x, min_y, max_y) pixel indices."""
    heads = []
    for i in range(imgs.shape[0]):
        min_x, max_x, min_y, max_y = [int(v) for v in bboxs[i]]
        heads.append(F.interpolate(imgs[i:i + 1, :, min_y:max_y, min_x:max_x], size=FACE_HW, mode="bilinear", align_corners=True))
    return torch.cat(heads, dim=0)


def face_loss(fsd, x, y, bbox):
    """FaceLoss.forward with bbox1 = bbox2 (impersonator_trainer.py:383-385) + compute_loss (networks.py:272-285): the
    unweighted sum of the five L1 feature distances."""
    f1, f2 = sphere20a_features(fsd, crop_head_bbox(x, bbox)), sphere20a_features(fsd, crop_head_bbox(y, bbox))
    return sum(F.l1_loss(a, b.detach()) for a, b in zip(f1, f2))


def style_loss(vsd, imgs, recon_imgs):
    """StyleLoss.forward (networks/networks.py:414-423) with weight 1 and Vgg19 as the feature extractor
    (impersonator_trainer.py:262-264)."""
    def gram(x):
        g = x.view(x.size(0), x.size(1), x.size(2) * x.size(3))
        return torch.bmm(g, torch.transpose(g, 1, 2))
    feats = vgg19_features(vsd, F.interpolate(imgs, (224, 224)))
    recon = vgg19_features(vsd, F.interpolate(recon_imgs, (224, 224)))
    return sum(torch.mean(torch.abs(gram(a) - gram(b))) / (a.size(2) * a.size(3)) for a, b in zip(feats, recon))


def generator_train_loss(gsd, dsd, batch, opt=None, align_corners=False):
    """models/impersonator_trainer.py: forward (:329-348) + _optimize_G (:368-394): adversarial (LSGAN, target 0), L1
    reconstruction of the source, the transfer term -- L1 (what the `--use_vgg` help text calls the default; the
    reference's own code path for it only works with `self._crt_tsf` set, so the pin test injects torch.nn.L1Loss there)
    or, with opt['vgg'] = a VGG19 state_dict, VGGLoss (--use_vgg) --, the mask term (MSE, or BCE with opt['mask_bce']),
    mask total variation.  opt['bg_both']: BGNet ran on 2N inputs and the transferred image blends with the second half
    (:336-339).  opt['face'] = a Sphere20a state_dict: the face term on the head crops batch['head_bbox'] (--use_face,
    :383-385).  opt['style'] (needs opt['vgg']): the Gram-matrix style term (--use_style, :379-381).
    batch: input_G_bg (N or 2N,4,H,W), input_G_src, input_G_tsf (N,6,H,W), T (N,H,W,2), real_src, real_tsf (N,3,H,W),
    bg_mask (2N,1,H,W).  Returns (total, dict of terms, (fake_bg, fake_src_imgs, fake_tsf_imgs, fake_masks))."""
    o = dict(G_TRAIN_DEFAULTS)
    o.update(opt or {})
    fake_bg, src_img, src_mask, tsf_img, tsf_mask = generator_forward(gsd, batch["input_G_bg"], batch["input_G_src"],
                                                                      batch["input_G_tsf"], batch["T"], align_corners)
    bs = src_img.shape[0]
    fake_src_bg = fake_bg[0:bs]
    fake_tsf_bg = fake_bg[bs:] if o.get("bg_both") else fake_src_bg
    fake_src_imgs = src_mask * fake_src_bg + (1 - src_mask) * src_img
    fake_tsf_imgs = tsf_mask * fake_tsf_bg + (1 - tsf_mask) * tsf_img
    fake_masks = torch.cat([src_mask, tsf_mask], dim=0)
    d_fake = discriminator_forward(dsd, torch.cat([fake_tsf_imgs, batch["input_G_tsf"][:, 3:]], dim=1))
    tsf_term = vgg_loss(o["vgg"], fake_tsf_imgs, batch["real_tsf"]) if o.get("vgg") else F.l1_loss(fake_tsf_imgs, batch["real_tsf"])
    mask_term = (F.binary_cross_entropy if o.get("mask_bce") else F.mse_loss)(fake_masks, batch["bg_mask"])
    terms = dict(
        g_adv=torch.mean(d_fake ** 2) * o["lambda_D_prob"],
        g_rec=F.l1_loss(fake_src_imgs, batch["real_src"]) * o["lambda_rec"],
        g_tsf=tsf_term * o["lambda_tsf"],
        g_mask=mask_term * o["lambda_mask"],
        **({"g_face": face_loss(o["face"], fake_tsf_imgs, batch["real_tsf"], batch["head_bbox"]) * o.get("lambda_face", 1.0)}
           if o.get("face") else {}),
        **({"g_style": style_loss(o["vgg"], fake_tsf_imgs, batch["real_tsf"]) * o.get("lambda_style", 5.0)}
           if o.get("style") else {}),
        g_mask_smooth=(torch.mean(torch.abs(fake_masks[:, :, :, :-1] - fake_masks[:, :, :, 1:])) +
                       torch.mean(torch.abs(fake_masks[:, :, :-1, :] - fake_masks[:, :, 1:, :]))) * o["lambda_mask_smooth"])
    return sum(terms.values()), terms, (fake_bg, fake_src_imgs, fake_tsf_imgs, fake_masks)


def generator_train_steps(gsd, dsd, batches, opt=None, align_corners=False):
    """`len(batches)` generator updates with torch.optim.Adam (impersonator_trainer.py:229-230, 355-357), the
    discriminator frozen.  Returns (per-step loss terms, gradients of the first step, final parameters)."""
    o = dict(G_TRAIN_DEFAULTS)
    o.update(opt or {})
    params = {k: v.clone().requires_grad_(True) for k, v in gsd.items()}
    optim = torch.optim.Adam(list(params.values()), lr=o["lr_G"], betas=(o["G_adam_b1"], o["G_adam_b2"]))
    hist, first = [], None
    for batch in batches:
        optim.zero_grad()
        total, terms, _ = generator_train_loss(params, dsd, batch, o, align_corners)
        total.backward()
        if first is None:
            first = {k: (v.grad.detach().clone() if v.grad is not None else torch.zeros_like(v)) for k, v in params.items()}
        optim.step()
        hist.append({k: float(v.detach()) for k, v in terms.items()})
    return hist, first, {k: v.detach().clone() for k, v in params.items()}


# --------------------------------------------------------------------------- Imitator host methods (a1, a11, H9, H10)
def swap_smpl(src_cam, src_shape, tgt_smpl, first_cam, cam_strategy="smooth"):
    """Imitator.swap_smpl (models/imitator.py:216-234) for one frame (1,85): the camera policy."""
    tgt_cam, pose = tgt_smpl[:, 0:3].contiguous(), tgt_smpl[:, 3:75].contiguous()
    if cam_strategy == "smooth":
        cam = src_cam.clone()
        cam[:, 1:] += tgt_cam[:, 1:] - first_cam[:, 1:]
    elif cam_strategy == "source":
        cam = src_cam
    else:
        cam = tgt_cam
    return torch.cat([cam, pose, src_shape], dim=1)


def get_vis_f2pts(f2pts, fims):
    """SMPLRenderer.get_vis_f2pts (utils/nmr.py:506-546, --only_vis, hazard H10): faces that are not among
    `fim.unique()[1:]` become -2 (the first unique value is dropped unconditionally: it is -1 whenever a
    background pixel exists)."""
    out = torch.zeros_like(f2pts) - 2.0
    for i in range(f2pts.shape[0]):
        ids = fims[i].unique()[1:].long()
        out[i, ids] = f2pts[i, ids]
    return out


def imitator_personalize(sd, img, src_info, faces_idx, map_fn, only_vis=False, bg_sd=None, bg_ks=13, ft_ks=3, image_size=256):
    """Imitator.personalize (models/imitator.py:82-155) after `hmr.get_details` (src_info: cam, verts, shape, ...):
    bg_sd None = --bg_model ORIGINAL (the generator's BGNet on the eroded-mask image, :126-132), else the InpaintSANet
    state dict (:124-125).  Returns the reference's src_info entries (feats as enc/res)."""
    f2v, fim, wim = render_fim_wim(src_info["cam"], src_info["verts"], faces_idx, image_size)
    cond = encode_fim(fim, map_fn)
    p2v = source_p2verts(f2v)            # H9: the reference negates y through a view of f2verts
    f2v = f2v.clone()
    f2v[:, :, :, 1] *= -1
    if only_vis:
        p2v = get_vis_f2pts(p2v, fim)
    bg_mask = morph(cond[:, -1:], bg_ks, "erode")
    if bg_sd is not None:
        bg = inpaint_forward(bg_sd, img, 1 - bg_mask)[1]
    else:
        bg = bgnet_forward(sd, torch.cat([img * bg_mask, bg_mask], dim=1))
    ft = 1 - morph(cond[:, -1:], ft_ks, "erode")
    enc, res = encode_src(sd, torch.cat([img * ft, cond], 1))
    out = dict(src_info)
    out.update(fim=fim, wim=wim, cond=cond, f2verts=f2v, p2verts=p2v, img=img, bg=bg, enc=enc, res=res)
    return out


def imitator_inference_by_smpls(sd, src, get_details, tgt_smpls, faces_idx, map_fn, cam_strategy="smooth", front_map_fn=None,
                                image_size=256, align_corners=False):
    """Imitator.inference_by_smpls (models/imitator.py:191-214): per frame t, transfer_params_by_smpl (:236-268; first_cam
    is set from the frame at t == 0 under 'smooth') then forward (:326-336) with the optional warp_front (:338-342,
    front_map_fn given = --front_warp).  `src` = imitator_personalize(...).  Returns a list of per-frame dicts."""
    first_cam, frames = None, []
    for t in range(tgt_smpls.shape[0]):
        tgt = tgt_smpls[t:t + 1]
        if t == 0 and cam_strategy == "smooth":
            first_cam = tgt[:, 0:3].clone()
        info = get_details(swap_smpl(src["cam"], src["shape"], tgt, first_cam, cam_strategy))
        fr = transfer_frame(src["img"], src["p2verts"], info["cam"], info["verts"], faces_idx, map_fn, image_size, align_corners)
        pred, _, mask = imitator_forward(sd, src["enc"], src["res"], src["bg"], fr["tsf_inputs"], fr["T"], align_corners)
        if front_map_fn is not None:
            front = encode_fim(fr["fim"], front_map_fn)
            pred = (1 - front) * pred + fr["tsf_img"] * front * (1 - mask)
        info = dict(info)
        info.update(fim=fr["fim"], wim=fr["wim"], cond=fr["cond"], tsf_img=fr["tsf_img"], T=fr["T"], preds=pred,
                    first_cam=None if first_cam is None else first_cam.clone())
        frames.append(info)
    return frames


# --------------------------------------------------------------------------- SMPL (a2), any float dtype
def smpl_tensors(model, dtype=torch.float32):
    """The fp32 tensors an SMPL module holds (the reference's buffer names, networks/batch_smpl.py:242-283), cast to `dtype`."""
    names = ("v_template", "shapedirs", "J_regressor", "posedirs", "weights", "joint_regressor")
    out = {n: getattr(model, n).detach().cpu().to(dtype) for n in names}
    out["parents"] = np.asarray(model.parents)
    return out


def smpl_forward(sm, beta, theta):
    """SMPL.forward (networks/batch_smpl.py:285-375; batch_rodrigues :64-101, batch_global_rigid_transformation :129-218
    with rotate_base=False) in the dtype of `sm`'s tensors.  float32 = the reference's arithmetic; float64 on the same fp32
    model values, rounded to fp32 afterwards, = the correctly rounded value of the function that code defines (what the
    device's "compensated" SMPL mode computes).  -> (verts, joints, Rs)"""
    dt = sm["v_template"].dtype
    beta, theta = beta.to(dt), theta.to(dt)
    n, nv = beta.shape[0], sm["v_template"].shape[0]
    v_shaped = torch.matmul(beta, sm["shapedirs"]).view(-1, nv, 3) + sm["v_template"]
    J = torch.stack([torch.matmul(v_shaped[:, :, k], sm["J_regressor"]) for k in range(3)], dim=2)
    r3 = theta.reshape(-1, 3)
    angle = torch.norm(r3 + 1e-8, p=2, dim=1, keepdim=True)
    r = torch.div(r3, angle).unsqueeze(-1)
    angle = angle.unsqueeze(-1)
    cos, sin = torch.cos(angle), torch.sin(angle)
    outer = torch.matmul(r, r.permute(0, 2, 1))
    eyes = torch.eye(3, dtype=dt).unsqueeze(0).repeat(r3.shape[0], 1, 1)
    rx, ry, rz = r[:, 0, 0], r[:, 1, 0], r[:, 2, 0]
    zero = torch.zeros_like(rx)
    skew = torch.stack([zero, -rz, ry, rz, zero, -rx, -ry, rx, zero], dim=1).view(-1, 3, 3)   # batch_skew, :19-60
    Rs = (cos * eyes + (1 - cos) * outer + sin * skew).view(-1, 24, 3, 3)
    pose_feature = (Rs[:, 1:] - torch.eye(3, dtype=dt)).view(-1, 207)
    v_posed = torch.matmul(pose_feature, sm["posedirs"]).view(-1, nv, 3) + v_shaped

    Js = J.unsqueeze(-1)

    def make_A(R, t):
        return torch.cat([F.pad(R, [0, 0, 0, 1, 0, 0]), torch.cat([t, torch.ones(n, 1, 1, dtype=dt)], dim=1)], dim=2)

    results = [make_A(Rs[:, 0], Js[:, 0])]
    for i in range(1, 24):
        p = int(sm["parents"][i])
        results.append(torch.matmul(results[p], make_A(Rs[:, i], Js[:, i] - Js[:, p])))
    results = torch.stack(results, dim=1)
    Js_w0 = torch.cat([Js, torch.zeros(n, 24, 1, 1, dtype=dt)], dim=2)
    A = results - F.pad(torch.matmul(results, Js_w0), [3, 0, 0, 0, 0, 0, 0, 0])

    W = sm["weights"].repeat(n, 1).view(n, -1, 24)
    T = torch.matmul(W, A.view(n, 24, 16)).view(n, -1, 4, 4)
    v_homo = torch.matmul(T, torch.cat([v_posed, torch.ones(n, nv, 1, dtype=dt)], dim=2).unsqueeze(-1))
    verts = v_homo[:, :, :3, 0]
    joints = torch.stack([torch.matmul(verts[:, :, k], sm["joint_regressor"]) for k in range(3)], dim=2)
    return verts, joints, Rs


def get_details(sm, theta):
    """HumanModelRecovery.get_details (networks/hmr.py:302-330): fp32 dictionary whatever dtype `sm` computes in (the SMPL
    stage's results are rounded to fp32 once, then the keypoint projection of batch_smpl.py:221-234 runs in fp32)."""
    theta = theta.float()
    cam, pose, shape = theta[:, 0:3].contiguous(), theta[:, 3:75].contiguous(), theta[:, 75:].contiguous()
    verts, j3d, _ = smpl_forward(sm, shape, pose)
    verts, j3d = verts.float(), j3d.float()
    return dict(theta=theta, cam=cam, pose=pose, shape=shape, verts=verts, j3d=j3d, j2d=cam[:, None, 0:1] * (j3d[:, :, :2] + cam[:, None, 1:]))
