"""Swapper (appearance transfer) on MI355X -- reference surface: models/swapper.py:15-271 (BASELINE config 4).

`swap_setup` personalises two subjects (each exactly as Imitator.personalize, plus a 'par' part map,
swapper.py:128-129); `swap` builds the two flow fields (T11 = identity grid with the non-kept pixels sent to -2,
T21 = barycentric flow from the target's visible faces, clamped; swapper.py:242-253), warps the two images and
runs the two-stream Liquid Warping Block generator (generator.py:245-275) with the blend fused.
All per-pixel device work is liblwg; the mask bookkeeping around it (a handful of elementwise ops, once per swap)
stays in torch.
"""
import numpy as np
import torch

from ..utils.nmr import SMPLRenderer
from .imitator import Imitator


class Swapper(Imitator):
    PART_IDS = {
        'body': [1, 2, 3, 4, 5, 6, 7, 8, 9],
        'all': [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]
    }

    def __init__(self, opt, hmr=None, render=None, generator=None, bgnet=None, part_fn=None, part_faces=None):
        """`part_fn` (nf+1, 11) / `part_faces` (10 face-id lists) are read from the reference's asset files
        (utils/mesh.py of this package) unless injected (synthetic configuration)."""
        super().__init__(opt, hmr=hmr, render=render, generator=generator, bgnet=bgnet)
        self._name = 'Swapper'
        self.T = self.T12 = self.T21 = None
        self.grid = self.create_meshgrid(opt.image_size).cuda()
        if part_fn is None or part_faces is None:
            from ..utils import mesh
            part_fn = mesh.create_mapping('par', opt.uv_mapping, part_info=opt.part_info, contain_bg=True, fill_back=False)
            part_faces = list(mesh.get_part_face_ids('par', opt.uv_mapping, opt.part_info, fill_back=False).values())
        self.part_fn = torch.as_tensor(np.ascontiguousarray(part_fn)).float().contiguous().cuda()
        self.part_faces = [list(p) for p in part_faces]

    @staticmethod
    def create_meshgrid(image_size):
        """utils/nmr.py:490-504: identity sampling grid (is, is, 2), x fastest."""
        factor = (torch.arange(0, image_size, dtype=torch.float32) / (image_size - 1) - 0.5) * 2
        yv, xv = torch.meshgrid(factor, factor, indexing='ij')
        return torch.stack([xv, yv], dim=-1).contiguous()

    @torch.no_grad()
    def personalize(self, src_path, src_smpl=None, output_path='', visualizer=None, bg_img=None):
        """swapper.py:99-165: Imitator.personalize + the part map; returns the info dict (it does not set src_info)."""
        keep = self.src_info
        super().personalize(src_path, src_smpl=src_smpl, output_path=output_path, visualizer=visualizer, bg_img=bg_img)
        info = self.src_info
        self.src_info = keep
        info['part'], _ = self.render.encode_fim(info['cam'], info['verts'], fim=info['fim'], transpose=True,
                                                 map_fn=self.part_fn)
        return info

    @torch.no_grad()
    def swap_setup(self, src_path, tgt_path, src_smpl=None, tgt_smpl=None, output_dir='', src_bg=None, tgt_bg=None):
        """swapper.py:194-196."""
        self.src_info = self.personalize(src_path, src_smpl, bg_img=src_bg)
        self.tsf_info = self.personalize(tgt_path, tgt_smpl, bg_img=tgt_bg)

    @torch.no_grad()
    def swap(self, src_info, tgt_info, target_part='body', visualizer=None):
        """swapper.py:198-239."""
        assert target_part in self.PART_IDS.keys()
        selected_ids = self.PART_IDS[target_part]
        left_ids = [i for i in self.PART_IDS['all'] if i not in selected_ids]
        src_part_mask = (torch.sum(src_info['part'][:, selected_ids, ...], dim=1) != 0).bool()
        src_left_mask = torch.sum(src_info['part'][:, left_ids, ...], dim=1).bool()
        left_faces = sorted(set(f for i in left_ids for f in self.part_faces[i]))

        T11, T21 = self.calculate_trans(src_left_mask, left_faces)
        tsf21 = self.generator.transform(tgt_info['img'], T21)
        tsf11 = self.generator.transform(src_info['img'], T11)
        src_part_mask = src_part_mask[:, None, :, :].float()
        src_left_mask = src_left_mask[:, None, :, :].float()
        tsf_img = tsf21 * src_part_mask + tsf11 * src_left_mask
        tsf_inputs = torch.cat([tsf_img, src_info['cond']], dim=1)

        preds, tsf_mask = self.forward(tsf_inputs, tgt_info['feats'], T21, src_info['feats'], T11, src_info['bg'])
        if self._opt.front_warp:
            preds = self.warp(preds, src_info['img'], src_info['fim'], tsf_mask)
        if visualizer is not None:
            self.visualize(visualizer, src_img=src_info['img'], tgt_img=tgt_info['img'], preds=preds)
        self.T12, self.T21 = T11, T21
        return preds

    def calculate_trans(self, src_left_mask, left_faces):
        """swapper.py:242-253."""
        T11 = self.grid.clone()
        T11[~src_left_mask[0]] = -2
        T11 = T11.unsqueeze(0)
        tsf_f2p = self.tsf_info['p2verts'].clone()
        tsf_f2p[0, left_faces] = -2
        T21 = self.render.cal_bc_transform(tsf_f2p, self.src_info['fim'], self.src_info['wim'])
        T21.clamp_(-2, 2)
        return T11, T21

    def warp(self, preds, tsf, fim, fake_tsf_mask):
        """swapper.py:255-259."""
        front_mask = self.render.encode_front_fim(fim, transpose=True)
        return (1 - front_mask) * preds + tsf * front_mask * (1 - fake_tsf_mask)

    @torch.no_grad()
    def forward(self, tsf_inputs, feats21, T21, feats11, T11, bg):
        """swapper.py:261-271 -> (pred_imgs, tsf_mask)."""
        enc21, res21 = feats21
        enc11, res11 = feats11
        pred, _, mask = self.generator.swap(tsf_inputs, enc21, enc11, res21, res11, T21, T11, bg_img=bg)
        return pred, mask
