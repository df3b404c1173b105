"""Viewer (novel view synthesis) on MI355X -- reference surface: models/viewer.py:240-314.

Same kernels as the Imitator: rotate the personalised source mesh, render -> cond -> T -> warped source ->
generator.inference -> blend (viewer.py:273-314)."""
import torch

from ..utils import cv_utils
from .imitator import Imitator


class Viewer(Imitator):
    def __init__(self, opt, **kwargs):
        super().__init__(opt, **kwargs)
        self._name = 'Viewer'
        self.T = None

    def rotate_trans(self, rt, t, X):
        """viewer.py:240-247: X @ R + t with R = euler2matrix(rt) -- one liblwg launch (lwg_rotate_translate): the rotation and
        the translation are twelve host floats handed to the kernel by value."""
        import numpy as np
        from .. import _lib
        if not X.is_cuda:
            raise RuntimeError("Viewer.rotate_trans: the mesh must be a CUDA tensor (no CPU path)")
        R = np.ascontiguousarray(np.asarray(cv_utils.euler2matrix(rt), dtype=np.float32).reshape(3, 3))
        tv = np.ascontiguousarray(np.asarray(t, dtype=np.float32).reshape(3))
        x = X.float().contiguous()
        out = torch.empty_like(x)
        _lib.check(_lib.load().lwg_rotate_translate(_lib.ptr(x), x.numel() // 3, R.ctypes.data, tv.ctypes.data, _lib.ptr(out),
                                                    _lib.stream_ptr()))
        return out

    @torch.no_grad()
    def view(self, rt, t, visualizer=None, name='1'):
        """viewer.py:273-303 -> preds (1,3,is,is)."""
        src_info = self.src_info
        tsf_mesh = self.rotate_trans(rt, t, src_info['verts'])
        out = self.render.transfer(src_info['cam'], tsf_mesh, src_info['p2verts_c'], src_info['img'])
        self.T = out['T']
        self.tsf_info = dict(verts=tsf_mesh, cam=src_info['cam'], fim=out['fim'], wim=out['wim'], cond=out['cond'],
                             tsf_img=out['tsf_img'], T=out['T'])
        bg = src_info['bg'] if getattr(self._opt, 'bg_replace', False) else torch.zeros_like(src_info['bg'])
        enc, res = src_info['feats']
        preds, _, tsf_mask = self.generator.inference(enc, res, out['tsf_inputs'], out['T'], bg_img=bg)
        if self._opt.front_warp:
            preds = self.warp_front(preds, tsf_mask)
        if visualizer is not None:
            visualizer.vis_named_img('src_img', src_info['img'])
            visualizer.vis_named_img('pred_' + name, preds)
            visualizer.vis_named_img('cond_' + name, out['cond'])
        return preds
