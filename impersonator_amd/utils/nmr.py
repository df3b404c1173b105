"""SMPLRenderer on MI355X: the reference's geometry glue (utils/nmr.py:103-662) over liblwg.

Keeps the method names, argument order and return types of the methods on the Imitator.forward()
path -- `render_fim_wim`, `encode_fim`, `encode_front_fim`, `cal_bc_transform`, `get_vis_f2pts` --
and adds `transfer()`, the fused per-frame sequence of models/imitator.py:250-260.
The textured-rendering half of the reference class (forward/render/extract_tex, lighting) is not used
by Liquid-Warping-Block inference and is not provided.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _lib


def orthographic_proj_withz_idrot(X, cam, offset_z=0.):
    """utils/nmr.py:10-28 (kept for API compatibility; the device path fuses it into the face gather)."""
    scale = cam[:, 0].contiguous().view(-1, 1, 1)
    trans = cam[:, 1:3].contiguous().view(cam.size(0), 1, -1)
    proj_xy = scale * (X[:, :, :2] + trans)
    proj_z = X[:, :, 2, None] + offset_z
    return torch.cat((proj_xy, proj_z), 2)


class SMPLRenderer(nn.Module):
    # rasteriser defaults of neural_renderer (rasterize.py:8-13): render_fim_wim does not forward the
    # renderer's own near/far (nmr.py:277), so these are the values the reference effectively uses.
    RASTER_NEAR = 0.1
    RASTER_FAR = 100.0

    def __init__(self, face_path='assets/pretrains/smpl_faces.npy', uv_map_path='assets/pretrains/mapper.txt',
                 map_name='uv_seg', tex_size=3, image_size=256, anti_aliasing=True, fill_back=False,
                 background_color=(0, 0, 0), viewing_angle=30, near=0.1, far=25.0, has_front=False,
                 faces=None, map_fn=None, front_map_fn=None, back_map_fn=None, align_corners=False):
        """Same signature as utils/nmr.py:104-107.  The trailing keyword arguments (extension) supply the
        face list and the face->condition tables directly; without them they are read from `face_path` and
        from `<uv_map_path stem>_<map_name>.npy` tables (see utils/mesh.py of this package)."""
        super().__init__()
        self.background_color = background_color
        self.anti_aliasing = anti_aliasing
        self.image_size = image_size
        self.fill_back = fill_back
        self.map_name = map_name
        self.tex_size = tex_size
        self.align_corners = bool(align_corners)

        if faces is None:
            if not os.path.exists(face_path):
                raise FileNotFoundError("SMPL face list %s not found (README.md:48-68 of the reference: a download); "
                                        "pass faces=/map_fn= explicitly, e.g. from impersonator_amd.utils.synthetic" % face_path)
            faces = np.load(face_path)
        faces = np.asarray(faces)
        self.base_nf = faces.shape[0]
        if self.fill_back:
            faces = np.concatenate((faces, faces[:, ::-1]), axis=0)
        self.nf = faces.shape[0]
        self.register_buffer('faces', torch.tensor(np.ascontiguousarray(faces.astype(np.int32))).int().contiguous())

        if map_fn is None:
            from . import mesh
            map_fn = mesh.create_mapping(map_name, uv_map_path, contain_bg=True, fill_back=fill_back)
            if has_front and front_map_fn is None:
                front_map_fn = mesh.create_mapping('front', uv_map_path, contain_bg=True, fill_back=fill_back)
        self.register_buffer('map_fn', torch.as_tensor(np.ascontiguousarray(map_fn)).float().contiguous())
        if back_map_fn is not None:
            self.register_buffer('back_map_fn', torch.as_tensor(np.ascontiguousarray(back_map_fn)).float().contiguous())
        else:
            self.back_map_fn = None
        if front_map_fn is not None:
            self.register_buffer('front_map_fn', torch.as_tensor(np.ascontiguousarray(front_map_fn)).float().contiguous())
        else:
            self.front_map_fn = None

        self.near = near
        self.far = far
        self.proj_func = orthographic_proj_withz_idrot
        self.viewing_angle = viewing_angle
        self.eye = [0, 0, -(1. / np.tan(np.radians(self.viewing_angle)) + 1)]  # nmr.py:177
        self._eye_z = float(np.float32(self.eye[2]))
        self._ws = None

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _cuda(t, dtype=torch.float32):
        if not t.is_cuda:
            raise RuntimeError("impersonator_amd runs on the GPU only (got a %s tensor); there is no CPU fallback" % t.device)
        return t.to(dtype).contiguous()

    def _workspace(self, bs, nf, device):
        need = _lib.load().lwg_rasterize_workspace_bytes(bs, nf, self.image_size)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    def _faces_for(self, faces):
        if faces is None:
            return self.faces
        if faces.dim() == 3:
            # the reference accepts a per-sample face list (nmr.py:264-266); the SMPL topology is shared
            if not bool((faces == faces[0:1]).all()):
                raise NotImplementedError("per-sample face topologies are not supported")
            faces = faces[0]
        return faces.int().contiguous()

    # ------------------------------------------------------------------ reference API
    @torch.no_grad()
    def render_fim_wim(self, cam, vertices, faces=None):
        """utils/nmr.py:263-278 -> (f2verts (bs,nf,3,3), fim int32 (bs,is,is), wim (bs,is,is,3))."""
        lib = _lib.load()
        cam, vertices = self._cuda(cam), self._cuda(vertices)
        fidx = self._faces_for(faces).to(vertices.device)
        bs, nv = vertices.shape[:2]
        nf, s, dev = fidx.shape[0], self.image_size, vertices.device
        f2verts = torch.empty((bs, nf, 3, 3), device=dev, dtype=torch.float32)
        fim = torch.empty((bs, s, s), device=dev, dtype=torch.int32)
        wim = torch.empty((bs, s, s, 3), device=dev, dtype=torch.float32)
        st = _lib.stream_ptr()
        _lib.check(lib.lwg_project_faces(_lib.ptr(vertices), _lib.ptr(cam), _lib.ptr(fidx), bs, nv, nf, self._eye_z,
                                         _lib.ptr(f2verts), st))
        ws = self._workspace(bs, nf, dev)
        _lib.check(lib.lwg_rasterize_fim_wim(_lib.ptr(f2verts), bs, nf, s, self.RASTER_NEAR, self.RASTER_FAR,
                                             _lib.ptr(fim), _lib.ptr(wim), None, _lib.ptr(ws), ws.numel(), st))
        return f2verts, fim, wim

    @torch.no_grad()
    def rasterize(self, faces, near=None, far=None, return_depth=False):
        """nr.rasterize_face_index_map_and_weight_map (rasterize.py:543-571) on given (bs,nf,3,3) faces."""
        lib = _lib.load()
        faces = self._cuda(faces)
        bs, nf = faces.shape[:2]
        s, dev = self.image_size, faces.device
        fim = torch.empty((bs, s, s), device=dev, dtype=torch.int32)
        wim = torch.empty((bs, s, s, 3), device=dev, dtype=torch.float32)
        depth = torch.empty((bs, s, s), device=dev, dtype=torch.float32) if return_depth else None
        ws = self._workspace(bs, nf, dev)
        _lib.check(lib.lwg_rasterize_fim_wim(_lib.ptr(faces), bs, nf, s,
                                             self.RASTER_NEAR if near is None else near,
                                             self.RASTER_FAR if far is None else far,
                                             _lib.ptr(fim), _lib.ptr(wim), _lib.ptr(depth), _lib.ptr(ws), ws.numel(),
                                             _lib.stream_ptr()))
        return (fim, wim, depth) if return_depth else (fim, wim)

    @torch.no_grad()
    def encode_fim(self, cam, vertices, fim=None, transpose=True, map_fn=None):
        """utils/nmr.py:328-341 -> (fim_enc, fim)."""
        if fim is None:
            _, fim, _ = self.render_fim_wim(cam, vertices)
        table = self.map_fn if map_fn is None else map_fn
        return self._lookup(fim, table, transpose), fim

    @torch.no_grad()
    def encode_front_fim(self, fim, transpose=True, front_fn=True):
        """utils/nmr.py:343-352."""
        table = self.front_map_fn if front_fn else self.back_map_fn
        if table is None:
            raise RuntimeError("renderer was built without the %s map" % ('front' if front_fn else 'back'))
        return self._lookup(fim, table, transpose)

    def _lookup(self, fim, table, transpose):
        fim = self._cuda(fim, torch.int32)
        table = self._cuda(table).to(fim.device)
        bs, h, w = fim.shape
        nrows, nc = table.shape
        out = torch.empty((bs, nc, h, w) if transpose else (bs, h, w, nc), device=fim.device, dtype=torch.float32)
        _lib.check(_lib.load().lwg_encode_fim(_lib.ptr(fim), _lib.ptr(table), bs, h * w, nrows, nc, int(bool(transpose)),
                                              _lib.ptr(out), _lib.stream_ptr()))
        return out

    @torch.no_grad()
    def cal_bc_transform(self, src_f2pts, dst_fims, dst_wims):
        """utils/nmr.py:617-659 -> T (bs, is, is, 2).  src_f2pts (1|bs, nf, 3, 2)."""
        src = self._cuda(src_f2pts)
        fim = self._cuda(dst_fims, torch.int32)
        wim = self._cuda(dst_wims)
        bs = fim.shape[0]
        T = torch.empty((bs, self.image_size, self.image_size, 2), device=fim.device, dtype=torch.float32)
        _lib.check(_lib.load().lwg_cal_bc_transform(_lib.ptr(src), src.shape[0], _lib.ptr(fim), _lib.ptr(wim), bs,
                                                    src.shape[1], self.image_size, _lib.ptr(T), _lib.stream_ptr()))
        return T

    @torch.no_grad()
    def grid_sample(self, x, T):
        """F.grid_sample(x, T) as the reference calls it (models/imitator.py:259, impersonator_trainer.py:62): bilinear,
        zeros padding, the renderer's align_corners; x (1|n, C, H, W), T (n, Ho, Wo, 2)."""
        x, T = self._cuda(x), self._cuda(T)
        n, ho, wo, _ = T.shape
        xn, c, h, w = x.shape
        out = torch.empty((n, c, ho, wo), device=x.device, dtype=torch.float32)
        _lib.check(_lib.load().lwg_grid_sample(_lib.ptr(x), xn, c, h, w, _lib.ptr(T), n, ho, wo, int(self.align_corners),
                                               _lib.ptr(out), _lib.stream_ptr()))
        return out

    @staticmethod
    @torch.no_grad()
    def get_vis_f2pts(f2pts, fims):
        """utils/nmr.py:506-546 (--only_vis, hazard H10): faces not in fim.unique()[1:] become -2 -- i.e. the sorted unique
        values minus the SMALLEST one present (the background's -1 whenever a background pixel exists).  CUDA tensors: two
        liblwg launches (lwg_vis_f2pts: flag the values present, select); CPU tensors: the reference's indexing."""
        if f2pts.is_cuda:
            batched = f2pts.dim() == 4
            pts = (f2pts if batched else f2pts[None]).float().contiguous()
            fim = (fims if batched else fims[None]).to(torch.int32).contiguous()
            bs, nf = pts.shape[:2]
            per_face = pts[0, 0].numel()
            lib = _lib.load()
            ws = torch.empty(lib.lwg_vis_f2pts_workspace_bytes(bs, nf), dtype=torch.uint8, device=pts.device)
            out = torch.empty_like(pts)
            _lib.check(lib.lwg_vis_f2pts(_lib.ptr(pts), bs, nf, per_face, _lib.ptr(fim), fim.shape[1], fim.shape[2], _lib.ptr(out),
                                         _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
            return out if batched else out[0]

        def vis(orig, fim):
            out = torch.zeros_like(orig) - 2.0
            ids = fim.unique()[1:].long()
            out[ids] = orig[ids]
            return out
        if f2pts.dim() == 4:
            return torch.stack([vis(f2pts[i], fims[i]) for i in range(f2pts.shape[0])], dim=0)
        return vis(f2pts, fims)

    @torch.no_grad()
    def source_p2verts(self, f2verts):
        """models/imitator.py:105-107 (hazard H9) as one launch: negates y of `f2verts` (bs,nf,3,3) IN PLACE -- the reference
        does it through the `p2verts = f2verts[..., 0:2]` view -- and returns the contiguous (bs,nf,3,2) copy of that view
        which the per-frame flow kernel reads."""
        f2verts = self._cuda(f2verts)
        bs, nf = f2verts.shape[:2]
        p2 = torch.empty((bs, nf, 3, 2), device=f2verts.device, dtype=torch.float32)
        _lib.check(_lib.load().lwg_source_p2verts(_lib.ptr(f2verts), bs, nf, _lib.ptr(p2), _lib.stream_ptr()))
        return p2

    @torch.no_grad()
    def mask_compose(self, img, mask, tail, invert=False):
        """torch.cat([img * (1 - mask if invert else mask), tail], dim=1) in one launch (models/imitator.py:127-128,135):
        img (n,3,H,W), mask (n,1,H,W), tail (n,ct,H,W)."""
        img, mask, tail = self._cuda(img), self._cuda(mask), self._cuda(tail)
        n, _, h, w = img.shape
        ct = tail.shape[1]
        out = torch.empty((n, 3 + ct, h, w), device=img.device, dtype=torch.float32)
        _lib.check(_lib.load().lwg_mask_compose(_lib.ptr(img), _lib.ptr(mask), int(bool(invert)), _lib.ptr(tail), ct, n, h, w,
                                                _lib.ptr(out), _lib.stream_ptr()))
        return out

    # ------------------------------------------------------------------ fused per-frame path
    @torch.no_grad()
    def transfer(self, cam, vertices, src_p2verts, src_img):
        """models/imitator.py:250-260 in one launch sequence.  cam (bs,3), vertices (bs,nv,3), shared source
        src_p2verts (1,nf,3,2), src_img (1,3,is,is).  Returns a dict with the reference's tsf_info entries
        (f2verts, fim, wim, cond, T, tsf_img) and `tsf_inputs` = cat(tsf_img, cond) as an NCHW-shaped view of
        the NHWC8 buffer the generator reads directly."""
        lib = _lib.load()
        cam, vertices = self._cuda(cam), self._cuda(vertices)
        p2v, img = self._cuda(src_p2verts), self._cuda(src_img)
        bs, nv = vertices.shape[:2]
        nf, s, dev = self.nf, self.image_size, vertices.device
        nc = self.map_fn.shape[1]
        if p2v.shape[0] != 1 or img.shape[0] != 1:
            raise ValueError("transfer() warps ONE source onto a batch of target poses")
        out = dict(
            f2verts=torch.empty((bs, nf, 3, 3), device=dev), fim=torch.empty((bs, s, s), device=dev, dtype=torch.int32),
            wim=torch.empty((bs, s, s, 3), device=dev), cond=torch.empty((bs, nc, s, s), device=dev),
            T=torch.empty((bs, s, s, 2), device=dev), tsf_img=torch.empty((bs, 3, s, s), device=dev))
        x0 = torch.empty((bs, s, s, 8), device=dev) if nc == 3 else None
        ws = self._workspace(bs, nf, dev)
        _lib.check(lib.lwg_transfer_frame(
            _lib.ptr(vertices), _lib.ptr(cam), _lib.ptr(self.faces), bs, nv, nf, s, self._eye_z, self.RASTER_NEAR,
            self.RASTER_FAR, _lib.ptr(self.map_fn), nc, _lib.ptr(p2v), _lib.ptr(img), int(self.align_corners),
            _lib.ptr(out['f2verts']), _lib.ptr(out['fim']), _lib.ptr(out['wim']), _lib.ptr(out['cond']),
            _lib.ptr(out['T']), _lib.ptr(out['tsf_img']), _lib.ptr(x0), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
        if x0 is not None:
            out['tsf_inputs'] = x0.permute(0, 3, 1, 2)[:, :3 + nc]
        else:
            out['tsf_inputs'] = torch.cat([out['tsf_img'], out['cond']], dim=1)
        return out


def eye_z(viewing_angle=30):
    return -(1. / math.tan(math.radians(viewing_angle)) + 1)
