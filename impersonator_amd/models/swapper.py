"""Swapper (appearance transfer) on MI355X -- reference surface: models/swapper.py:15-271 (BASELINE config 4).

`swap_setup` personalises two subjects (each exactly as Imitator.personalize, plus a 'par' part map,
swapper.py:128-129); `swap` builds the two flow fields (T11 = identity grid with the non-kept pixels sent to -2,
T21 = barycentric flow from the target's visible faces, clamped; swapper.py:242-253), warps the two images and
runs the two-stream Liquid Warping Block generator (generator.py:245-275) with the blend fused.
All device work is liblwg, the mask bookkeeping included (personalize.hip: lwg_swap_masks, lwg_mask_faces, lwg_swap_compose,
lwg_clamp): a swap launches no framework kernel and can be replayed as one HIP graph (`swap_graph`).
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

    def _part_sets(self, target_part):
        """(selected bits, kept bits, device byte-per-face flags of the kept parts' faces) of `target_part`, built once."""
        cache = getattr(self, '_part_cache', None)
        if cache is None:
            cache = self._part_cache = {}
        if target_part not in cache:
            selected_ids = self.PART_IDS[target_part]
            left_ids = [i for i in self.PART_IDS['all'] if i not in selected_ids]
            drop = np.zeros(self.render.nf, np.uint8)
            for i in left_ids:                                   # left_faces of swapper.py:208, as one byte per face
                drop[np.asarray(self.part_faces[i], np.int64)] = 1
            bits = lambda ids: int(sum(1 << i for i in ids))
            cache[target_part] = (bits(selected_ids), bits(left_ids), torch.from_numpy(drop).cuda())
        return cache[target_part]

    @torch.no_grad()
    def swap(self, src_info, tgt_info, target_part='body', visualizer=None):
        """swapper.py:198-239.  Every device step is a liblwg launch (lwg_swap_masks, lwg_mask_faces, lwg_cal_bc_transform, lwg_clamp,
        the two image warps, lwg_swap_compose, the two-stream generator with the blend fused): no framework kernel, no index list
        copied to the device, no read-back -- the call can be captured in a HIP graph (`swap_graph`)."""
        from .. import _lib
        assert target_part in self.PART_IDS.keys()
        lib = _lib.load()
        sel_bits, left_bits, drop = self._part_sets(target_part)
        part = src_info['part'].float().contiguous()
        if part.shape[0] != 1:
            raise ValueError("Swapper.swap transfers ONE source / target pair per call, as the reference does")
        _, nparts, h, w = part.shape
        dev = part.device
        part_mask = torch.empty((1, 1, h, w), device=dev)
        left_mask = torch.empty((1, 1, h, w), device=dev)
        T11 = torch.empty((1, h, w, 2), device=dev)
        _lib.check(lib.lwg_swap_masks(_lib.ptr(part), nparts, h, w, sel_bits, left_bits, _lib.ptr(self.grid), _lib.ptr(part_mask),
                                      _lib.ptr(left_mask), _lib.ptr(T11), _lib.stream_ptr()))
        T21 = self._flow_through_kept_faces(drop)
        tsf21 = self.generator.transform(tgt_info['img'], T21)
        tsf11 = self.generator.transform(src_info['img'], T11)
        cond = src_info['cond'].float().contiguous()
        tsf_inputs = torch.empty((1, 3 + cond.shape[1], h, w), device=dev)
        _lib.check(lib.lwg_swap_compose(_lib.ptr(tsf21), _lib.ptr(tsf11), _lib.ptr(part_mask), _lib.ptr(left_mask), _lib.ptr(cond),
                                        cond.shape[1], h, w, _lib.ptr(tsf_inputs), _lib.stream_ptr()))

        preds, tsf_mask = self.forward(tsf_inputs, tgt_info['feats'], T21, src_info['feats'], T11, src_info['bg'])
        if self._opt.front_warp:
            preds = self.warp(preds, src_info['img'], src_info['fim'], tsf_mask)
        if visualizer is not None:
            self.visualize(visualizer, src_img=src_info['img'], tgt_img=tgt_info['img'], preds=preds)
        self.T12, self.T21 = T11, T21
        return preds

    def _flow_through_kept_faces(self, drop_faces):
        """swapper.py:246-251, the T21 half of calculate_trans: the target's face vertices with the kept parts' faces sent to -2
        (`drop_faces`: one byte per face on the device, `_part_sets`), the barycentric flow of the source's visible faces through
        them, clamped to [-2, 2]."""
        from .. import _lib
        lib = _lib.load()
        p2v = self.tsf_info.get('p2verts_c')
        if p2v is None:
            p2v = self.tsf_info['p2verts'].float().contiguous()
        tsf_f2p = torch.empty_like(p2v)
        _lib.check(lib.lwg_mask_faces(_lib.ptr(p2v), _lib.ptr(drop_faces), p2v.shape[1], 6, _lib.ptr(tsf_f2p), _lib.stream_ptr()))
        T21 = self.render.cal_bc_transform(tsf_f2p, self.src_info['fim'], self.src_info['wim'])
        _lib.check(lib.lwg_clamp(_lib.ptr(T21), T21.numel(), -2.0, 2.0, _lib.stream_ptr()))
        return T21

    @torch.no_grad()
    def calculate_trans(self, src_left_mask, left_faces):
        """swapper.py:242-253 with the reference's signature: src_left_mask (1, H, W) bool, left_faces a list of face ids
        -> (T11, T21).  `swap` itself takes the same two kernels with the face set cached on the device (`_part_sets`)."""
        from .. import _lib
        lib = _lib.load()
        m = src_left_mask.reshape(1, *src_left_mask.shape[-2:]).float().contiguous().cuda()
        _, h, w = m.shape
        scratch = torch.empty((2, h, w), device=m.device)
        T11 = torch.empty((1, h, w, 2), device=m.device)
        _lib.check(lib.lwg_swap_masks(_lib.ptr(m), 1, h, w, 0, 1, _lib.ptr(self.grid), _lib.ptr(scratch[0]), _lib.ptr(scratch[1]),
                                      _lib.ptr(T11), _lib.stream_ptr()))
        drop = np.zeros(self.render.nf, np.uint8)
        drop[np.asarray(sorted(set(left_faces)), np.int64)] = 1
        return T11, self._flow_through_kept_faces(torch.from_numpy(drop).cuda())

    @torch.no_grad()
    def swap_graph(self, src_info, tgt_info, target_part='body'):
        """(extension) `swap(src_info, tgt_info, target_part)` captured once as a HIP graph: returns `run() -> preds` that replays it
        (the two personalised subjects are the graph's inputs: personalise again -> capture again).  Same values as `swap`."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self.swap(src_info, tgt_info, target_part)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            preds = self.swap(src_info, tgt_info, target_part)

        def run():
            graph.replay()
            return preds

        run.graph, run.preds = graph, preds
        return run

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
