"""Imitator (motion imitation) on MI355X -- the reference's task model surface (models/imitator.py:14-342).

Same public methods and side effects:
    Imitator(opt), personalize(src_path, src_smpl=None, output_path='', visualizer=None),
    inference(tgt_paths, tgt_smpls=None, cam_strategy='smooth', output_dir='', visualizer=None, verbose=True),
    inference_by_smpls(tgt_smpls, cam_strategy='smooth', output_dir='', visualizer=None),
    swap_smpl, transfer_params_by_smpl, transfer_params, forward(tsf_inputs, T), warp_front
with `self.src_info` / `self.tsf_info` carrying the reference's keys.

What changed underneath: the reference walks frames one by one in Python (imitator.py:166), syncing the device
every frame; here frames of one source are independent given the cached source features, so `inference*`
groups them into batches of `opt.batch_size` and each batch is ONE launch sequence in liblwg
(SMPLRenderer.transfer + ImpersonatorGenerator.inference with the blend fused), with one device->host copy
per batch.  `first_cam` (imitator.py:243-244) is taken from frame 0 up front instead of being discovered at t == 0.
"""
import os

import numpy as np
import torch

from ..networks.networks import NetworksFactory
from ..utils import cv_utils, util
from ..utils.nmr import SMPLRenderer
from .models import BaseModel


class Imitator(BaseModel):
    def __init__(self, opt, hmr=None, render=None, generator=None, bgnet=None):
        """`opt` as produced by options.test_options.TestOptions.  The keyword arguments (extension) inject
        pre-built components -- used for the synthetic configuration where the reference's downloaded assets
        (SMPL pickle, UV mapper, checkpoints) do not exist."""
        super().__init__(opt)
        self._name = 'Imitator'
        self._create_networks(hmr, render, generator, bgnet)
        self.src_info = None
        self.tsf_info = None
        self.first_cam = None

    # ------------------------------------------------------------------ construction (imitator.py:26-74)
    def _create_networks(self, hmr, render, generator, bgnet):
        opt = self._opt
        self.generator = (generator if generator is not None else self._create_generator()).cuda()
        if bgnet is not None:
            self.bgnet = bgnet
        elif getattr(opt, 'bg_model', 'ORIGINAL') != 'ORIGINAL':
            self.bgnet = self._create_bgnet()
        else:
            self.bgnet = self.generator.bg_model   # imitator.py:33-34: the generator's own BGNet
        self.hmr = (hmr if hmr is not None else self._create_hmr()).cuda()
        if render is None:
            render = SMPLRenderer(image_size=opt.image_size, tex_size=opt.tex_size, has_front=opt.front_warp,
                                  fill_back=False, align_corners=getattr(opt, 'align_corners', False))
        self.render = render.cuda()
        self.detector = None
        if getattr(opt, 'has_detector', False):
            raise NotImplementedError("--has_detector (torchvision Mask-RCNN, utils/detectors.py) is outside the "
                                      "Imitator.forward() path")

    def _create_bgnet(self):
        """imitator.py:48-52: InpaintSANet(c_dim=4) with the deepfillv2 checkpoint."""
        net = NetworksFactory.get_by_name('deepfillv2', c_dim=4, image_size=self._opt.image_size)
        self._load_params(net, self._opt.bg_model, need_module=False)
        net.eval()
        return net.cuda()

    def _create_generator(self):
        opt = self._opt
        net = NetworksFactory.get_by_name(opt.gen_name, bg_dim=4, src_dim=3 + self._G_cond_nc,
                                          tsf_dim=3 + self._G_cond_nc, repeat_num=opt.repeat_num,
                                          image_size=opt.image_size, max_batch=max(1, opt.batch_size),
                                          align_corners=getattr(opt, 'align_corners', False))
        if opt.load_path:
            self._load_params(net, opt.load_path)
        else:
            raise ValueError('load_path {} is empty and load_epoch {} is 0'.format(opt.load_path, opt.load_epoch))
        net.eval()
        return net

    def _create_hmr(self):
        from ..networks.batch_smpl import HumanModelRecovery
        return HumanModelRecovery(self._opt.smpl_model).eval()

    def visualize(self, *args, **kwargs):
        visualizer = args[0]
        if visualizer is not None:
            for key, value in kwargs.items():
                visualizer.vis_named_img(key, value)

    # ------------------------------------------------------------------ once per source (imitator.py:82-145)
    def _load_image(self, src, size):
        """path | HxWx3 uint8 array | (3,H,W) float array in [-1,1] -> (1,3,size,size) cuda tensor, original image."""
        if isinstance(src, str):
            ori = cv_utils.read_cv2_img(src)
        else:
            ori = src
        if torch.is_tensor(ori):
            return ori.float().cuda().reshape(1, 3, size, size), ori
        ori = np.asarray(ori)
        if ori.dtype != np.uint8:
            return torch.tensor(ori, dtype=torch.float32).cuda().reshape(1, 3, size, size), ori
        img = cv_utils.transform_img(ori, size, transpose=True) * 2 - 1.0
        return torch.tensor(img, dtype=torch.float32).cuda()[None, ...], ori

    @torch.no_grad()
    def personalize(self, src_path, src_smpl=None, output_path='', visualizer=None, bg_img=None):
        opt = self._opt
        img, ori_img = self._load_image(src_path, opt.image_size)
        if src_smpl is None:
            raise NotImplementedError("estimating SMPL from the image needs the HMR regressor (networks/hmr.py), "
                                      "which is outside the Imitator.forward() path; pass src_smpl")
        src_smpl = torch.as_tensor(np.asarray(src_smpl), dtype=torch.float32).cuda().reshape(1, -1)

        src_info = self.hmr.get_details(src_smpl)
        src_f2verts, src_fim, src_wim = self.render.render_fim_wim(src_info['cam'], src_info['verts'])
        src_info['fim'] = src_fim
        src_info['wim'] = src_wim
        src_info['cond'], _ = self.render.encode_fim(src_info['cam'], src_info['verts'], fim=src_fim, transpose=True)
        src_info['f2verts'] = src_f2verts
        # hazard H9 (imitator.py:105-107): p2verts is a VIEW of f2verts and the y flip mutates f2verts too -- one liblwg launch
        # negates y in place and returns the contiguous (nf,3,2) copy the fused per-frame kernel reads
        p2verts_c = self.render.source_p2verts(src_f2verts)
        src_info['p2verts'] = src_f2verts[:, :, :, 0:2]
        if opt.only_vis:
            src_info['p2verts'] = p2verts_c = self.render.get_vis_f2pts(p2verts_c, src_fim)
        src_info['img'] = img
        src_info['image'] = ori_img

        # masks and network inputs of imitator.py:116-135 as liblwg launches (lwg_morph, lwg_mask_compose): bg is 1, ft is 0
        bg_cond = src_info['cond'][:, -1:, :, :]
        bg_done = None
        if bg_img is not None:
            src_info['bg'] = torch.as_tensor(bg_img, dtype=torch.float32).cuda().reshape(1, 3, opt.image_size, opt.image_size)
        elif getattr(opt, 'bg_model', 'ORIGINAL') != 'ORIGINAL' or self.bgnet is not self.generator.bg_model:
            body_mask = util.morph(bg_cond, ks=opt.bg_ks, mode='erode', complement=True)   # 1 - bg_mask
            # imitator.py:124-125.  The inpaintor has its own device handle (own scratch) and, at batch 1, launches of 64-128
            # workgroups: it runs on a side stream UNDERNEATH the source-stream encoder below (equally small launches) instead of in
            # front of it -- same kernels, same values; the streams meet again before personalize returns
            main = torch.cuda.current_stream()
            if os.environ.get("LWG_BG_SIDE_STREAM", "1") == "0":      # A/B switch: the inpaintor in front of the encoder, one stream
                src_info['bg'] = self.bgnet(img, masks=body_mask, only_x=True)
            else:
                if getattr(self, '_bg_stream', None) is None:
                    self._bg_stream = torch.cuda.Stream()
                fork = torch.cuda.Event()
                fork.record(main)
                with torch.cuda.stream(self._bg_stream):
                    self._bg_stream.wait_event(fork)
                    src_info['bg'] = self.bgnet(img, masks=body_mask, only_x=True)
                    bg_done = torch.cuda.Event()
                    bg_done.record(self._bg_stream)
                for t in (img, body_mask):
                    t.record_stream(self._bg_stream)
                src_info['bg'].record_stream(main)
        else:
            # imitator.py:126-132: BGNet on the masked image + mask
            bg_mask = util.morph(bg_cond, ks=opt.bg_ks, mode='erode')
            src_info['bg'] = self.bgnet(self.render.mask_compose(img, bg_mask, bg_mask))

        # ft_mask = 1 - erode(bg, ft_ks); src_inputs = cat([img * ft_mask, cond])
        ft_erode = util.morph(bg_cond, ks=opt.ft_ks, mode='erode')
        src_inputs = self.render.mask_compose(img, ft_erode, src_info['cond'], invert=True)
        src_info['feats'] = self.generator.encode_src(src_inputs)
        src_info['p2verts_c'] = p2verts_c
        if bg_done is not None:
            torch.cuda.current_stream().wait_event(bg_done)     # the background is complete for whoever uses src_info next
        self.src_info = src_info
        if getattr(self.generator, 'precision_policy', None) == 'auto' and self.generator.auto_pending():
            # conv arithmetic by probe (ImpersonatorGenerator.auto_probe; once per weight set): the source's own posed mesh as the
            # target frame (no second SMPL evaluation, `tsf_info` untouched), through both arithmetics
            probe = self.render.transfer(src_info['cam'], src_info['verts'], p2verts_c, img)
            self.generator.auto_probe(src_info['feats'][0], src_info['feats'][1], probe['tsf_inputs'], probe['T'], src_info['bg'])

        if visualizer is not None:
            visualizer.vis_named_img('src', img)
            visualizer.vis_named_img('bg', src_info['bg'])
        if output_path:
            cv_utils.save_cv2_img(np.asarray(src_info['image']), output_path, image_size=opt.image_size)

    # ------------------------------------------------------------------ per frame (imitator.py:216-268)
    def swap_smpl(self, src_cam, src_shape, tgt_smpl, cam_strategy='smooth'):
        """imitator.py:216-234.  All arguments may carry a batch of frames (the source rows broadcast)."""
        n = tgt_smpl.shape[0]
        tgt_cam = tgt_smpl[:, 0:3].contiguous()
        pose = tgt_smpl[:, 3:75].contiguous()
        if cam_strategy == 'smooth':
            cam = src_cam.expand(n, -1).clone()
            cam[:, 1:] += tgt_cam[:, 1:] - self.first_cam[:, 1:]
        elif cam_strategy == 'source':
            cam = src_cam.expand(n, -1)
        else:
            cam = tgt_cam
        return torch.cat([cam, pose, src_shape.expand(n, -1)], dim=1)

    @torch.no_grad()
    def transfer_params_by_smpl(self, tgt_smpl, cam_strategy='smooth', t=0):
        """imitator.py:236-268 for one frame (85,) or a batch (n,85); sets self.tsf_info, returns tsf_inputs."""
        src_info = self.src_info
        if isinstance(tgt_smpl, np.ndarray):
            tgt_smpl = torch.tensor(tgt_smpl).float()
        tgt_smpl = tgt_smpl.float().cuda()
        if tgt_smpl.dim() == 1:
            tgt_smpl = tgt_smpl[None, ...]
        if t == 0 and cam_strategy == 'smooth':
            self.first_cam = tgt_smpl[0:1, 0:3].clone()

        if tgt_smpl.is_cuda and hasattr(self.hmr, 'get_details_swapped') and (cam_strategy != 'smooth' or self.first_cam is not None):
            # swap_smpl + get_details as liblwg launches (same values; the tensor-op forms below stay the API and the CPU path)
            tsf_info = self.hmr.get_details_swapped(tgt_smpl, src_info['cam'], src_info['shape'], self.first_cam, cam_strategy)
        else:
            tsf_smpl = self.swap_smpl(src_info['cam'], src_info['shape'], tgt_smpl, cam_strategy=cam_strategy)
            tsf_info = self.hmr.get_details(tsf_smpl)
        out = self.render.transfer(tsf_info['cam'], tsf_info['verts'], src_info['p2verts_c'], src_info['img'])
        tsf_info['fim'] = out['fim']
        tsf_info['wim'] = out['wim']
        tsf_info['cond'] = out['cond']
        tsf_info['tsf_img'] = out['tsf_img']
        tsf_info['T'] = out['T']
        self.tsf_info = tsf_info
        return out['tsf_inputs']

    def transfer_params(self, tgt_path, tgt_smpl=None, cam_strategy='smooth', t=0):
        """imitator.py:270-283."""
        if tgt_smpl is None:
            raise NotImplementedError("estimating SMPL from target images needs the HMR regressor; pass tgt_smpl")
        tsf_inputs = self.transfer_params_by_smpl(tgt_smpl=tgt_smpl, cam_strategy=cam_strategy, t=t)
        self.tsf_info['image'] = cv_utils.read_cv2_img(tgt_path) if isinstance(tgt_path, str) and tgt_path else None
        return tsf_inputs

    @torch.no_grad()
    def forward(self, tsf_inputs, T, generator=None):
        """imitator.py:326-336: pred = mask*bg + (1-mask)*color, blend fused into the generator's last kernel.
        `generator` (extension): the engine replica to run on (predict_batches), default self.generator."""
        src_encoder_outs, src_resnet_outs = self.src_info['feats']
        generator = self.generator if generator is None else generator
        pred_imgs, _, tsf_mask = generator.inference(src_encoder_outs, src_resnet_outs, tsf_inputs, T,
                                                     bg_img=self.src_info['bg'])
        if self._opt.front_warp:
            pred_imgs = self.warp_front(pred_imgs, tsf_mask)
        return pred_imgs

    def warp_front(self, preds, mask):
        """imitator.py:338-342."""
        front_mask = self.render.encode_front_fim(self.tsf_info['fim'], transpose=True, front_fn=True)
        return (1 - front_mask) * preds + self.tsf_info['tsf_img'] * front_mask * (1 - mask)

    # ------------------------------------------------------------------ one frame per call, as ONE launch (latency form)
    @torch.no_grad()
    def frame_graph(self, batch=1, cam_strategy='smooth'):
        """(extension) `transfer_params_by_smpl` + `forward` for `batch` frames -- the body of the reference's per-frame loop,
        models/imitator.py:166-171 -- captured once as a HIP graph and replayed per call: at batch 1 the ~70 kernels of a frame are
        launch-bound when they are issued one by one from Python (about 20 us each on the host), one graph launch is not.
        Returns `run(tgt_smpl, t=1) -> preds`: the same values as the eager calls, bit for bit (same kernels, same order); `preds`
        and `self.tsf_info` are the graph's own static tensors, overwritten by the next call.  The source must be personalised
        first; personalising another source or changing the generator's precision needs a new graph."""
        if self.src_info is None:
            raise RuntimeError("frame_graph: personalize a source first")
        dev = self.src_info['img'].device
        width = 75 + int(self.src_info['shape'].shape[1])
        static_smpl = torch.zeros((batch, width), device=dev, dtype=torch.float32)
        static_first = torch.zeros((1, 3), device=dev, dtype=torch.float32)
        if self.first_cam is not None:
            static_first.copy_(self.first_cam.reshape(1, 3))
        self.first_cam = static_first              # the captured kernels read the camera reference through this pointer
        self.generator.reserve(batch)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):              # eager passes first: handles, scratch, per-device kernel attributes exist afterwards
            for _ in range(2):
                self.forward(self.transfer_params_by_smpl(static_smpl, cam_strategy, t=1), self.tsf_info['T'])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            tsf_inputs = self.transfer_params_by_smpl(static_smpl, cam_strategy, t=1)
            preds = self.forward(tsf_inputs, self.tsf_info['T'])
        info = self.tsf_info

        def run(tgt_smpl, t=1):
            tgt = torch.as_tensor(tgt_smpl, dtype=torch.float32).reshape(batch, width)
            if t == 0 and cam_strategy == 'smooth':
                static_first.copy_(tgt[0:1, 0:3], non_blocking=True)
            static_smpl.copy_(tgt, non_blocking=True)
            graph.replay()
            self.tsf_info = info
            return preds

        run.graph, run.static_smpl, run.preds = graph, static_smpl, preds
        return run

    # ------------------------------------------------------------------ stream pipeline over batches
    lanes = 2   # generator engines (and HIP streams) predict_batches deals consecutive batches to: 1 or 2; env LWG_LANES
    MAX_LANES = 2   # a third lane measured -3 % (profiles/r03_bench_lanes3.json) and is not offered
    # batches per lane and round: the geometry of lanes * round_depth batches is one launch sequence (the kernels are
    # latency-bound at these sizes) and the lanes drain once per round; tools/depth_bench.py at batch 8, two lanes:
    # depth 1 2709, 2 2772, 3 2799, 4 2811 frames/s in one process.  env LWG_ROUND_DEPTH
    round_depth = 4
    # consecutive batches of a round that run as ONE generator launch sequence (see predict_batches); env LWG_FUSE.
    # Two batches of 8 = 16 frames give the trunk convolutions 256 tiles of 8 x 32 pixels (eight waves sharing a weight
    # stage): conv kernels 0.42 -> 0.48 of the matrix-pipe peak, +2.5..4.5 % frames/s (profiles/r03_conv_experiments.md).
    # Results are bit-identical to unfused batches (every kernel is batch-invariant; tests/test_gpu_bench_config.py).
    # Round 5: four batches of 8 per launch sequence (32 frames: every layer's grid is a multiple of two full rounds of the chip, the
    # dependent launch chain -- conv, statistics, apply -- is paid once per 32 frames): +2.3..2.8 % frames/s over pairs on the same
    # box (profiles/r05_fuse_ab.md); with two lanes and round_depth 4 a round is exactly one sequence per lane.
    fuse = 4
    # entries of tsf_info with one row per frame (hmr.get_details + SMPLRenderer.transfer, imitator.py:236-268)
    PER_FRAME_KEYS = ('theta', 'cam', 'pose', 'shape', 'verts', 'j2d', 'j3d', 'fim', 'wim', 'cond', 'tsf_img', 'T')

    @staticmethod
    def _adjacent_rows(chunks):
        """`chunks` as one (sum of rows, width) view when they are consecutive row blocks of ONE contiguous tensor, else None."""
        c0 = chunks[0]
        if any(c.dim() != 2 or not c.is_contiguous() or c.dtype != torch.float32 or c.device != c0.device or c.shape[1] != c0.shape[1]
               for c in chunks):
            return None
        base, off, width = c0.untyped_storage().data_ptr(), c0.storage_offset(), c0.shape[1]
        for c in chunks:
            if c.untyped_storage().data_ptr() != base or c.storage_offset() != off:
                return None
            off += c.shape[0] * width
        return c0.as_strided((sum(c.shape[0] for c in chunks), width), (width, 1), c0.storage_offset())

    def _lanes(self, n):
        """n (stream, generator) pairs, each with its own HIP stream; lane 0 drives self.generator, the others an
        engine replica over the same parameters (ImpersonatorGenerator.replica: own scratch, own weight copy)."""
        have = getattr(self, '_lane_cache', None)
        if have is None or have[0][1] is not self.generator:
            have = self._lane_cache = [(torch.cuda.Stream(), self.generator)]
        while len(have) < n:
            have.append((torch.cuda.Stream(), self.generator.replica().cuda()))
        for _, g in have[1:n]:
            g.precision, g.align_corners = self.generator.precision, self.generator.align_corners
        return have[:n]

    @torch.no_grad()
    def predict_batches(self, batches, cam_strategy='smooth', lanes=None):
        """Yields (t, preds) for every (tgt_smpls_chunk, t) of `batches`, in order.  Frames are independent once the
        source is personalised, so consecutive batches are processed in rounds of `lanes * round_depth`:
          1. the geometry of the round's batches (swap_smpl, SMPL, projection, rasteriser, flow, image warp -- a dozen
             small, latency-bound kernels) runs as ONE launch sequence over all the round's frames on a side stream,
             after the generators of the previous round;
          2. their generators then run side by side, each on its own stream and engine (scratch): a layer is
             conv -> finalize -> apply, every launch waiting for the one before, and the idle tails and launch gaps
             of one chain are filled by the other's kernels (+10-15 % frames/s at batch 8 with two lanes).
        Events order every hand-over; round r+1 is enqueued before round r is yielded, so a consumer that synchronises on
        a result (device->host copy) does not drain the pipeline.  Same results as transfer_params_by_smpl + forward per
        batch, bit for bit.  (Rounds 2-3 also offered running round r+1's geometry UNDERNEATH round r's generators; at
        round_depth 4 it measured no gain -- profiles/r03_bench_overlap_ab.json -- and was removed in round 4.  What made it
        dangerous is in DESIGN_HISTORY.md: a packed-fp32 instruction form that miscomputes beside the bf16x3 conv kernels.)"""
        import os
        nl = min(self.MAX_LANES, max(1, int(lanes if lanes is not None else os.environ.get("LWG_LANES", self.lanes))))
        depth = max(1, int(os.environ.get("LWG_ROUND_DEPTH", self.round_depth)))
        fuse = max(1, int(os.environ.get("LWG_FUSE", self.fuse)))
        main = torch.cuda.current_stream()
        if getattr(self, '_side_stream', None) is None:
            self._side_stream = torch.cuda.Stream()
        side = self._side_stream
        # launch sequences of `fuse` batches: size every lane's scratch for that before the first one is enqueued
        want = max(1, int(self._opt.batch_size)) * (1 if self._opt.front_warp else fuse)
        self.generator.reserve(want)
        lane_list = self._lanes(nl)
        for _, g in lane_list[1:]:
            g.reserve(want)
        side.wait_stream(main)          # the personalised source (and the caller's smpl tensors) are main-stream work
        for st, _ in lane_list:
            st.wait_stream(main)

        def enqueue_round(items, prev_done):
            """geometry of `items` on the side stream, after the generators of the previous round, then one generator
            per lane; returns [(t, preds, info, done_event)]"""
            prepared = []
            with torch.cuda.stream(side):
                for ev in prev_done:
                    side.wait_event(ev)
                sizes = [int(chunk.shape[0]) if chunk.dim() > 1 else 1 for chunk, _ in items]
                if len(items) > 1 and all(t != 0 for _, t in items[1:]) and all(torch.is_tensor(c) for c, _ in items):
                    # one launch sequence for the whole round (the kernels are latency-bound at these sizes), then
                    # per-batch views of every result; a later chunk with t == 0 would reset the camera reference
                    # mid-way, that (unusual) order takes the chunk-by-chunk path below
                    # consecutive row blocks of one tensor (what _run_batches, sharding.imitate_sharded and bench.py hand over)
                    # are taken as ONE view: no copy kernel in the per-round sequence; anything else is concatenated
                    whole = self._adjacent_rows([c.reshape(n, -1) for (c, _), n in zip(items, sizes)])
                    if whole is None:
                        whole = torch.cat([c.reshape(n, -1) for (c, _), n in zip(items, sizes)], dim=0)
                    tsf_inputs = self.transfer_params_by_smpl(whole, cam_strategy, t=items[0][1])
                    info, k0 = self.tsf_info, 0
                    whole_inputs, whole_T = tsf_inputs, info['T']
                    for (_, t), n in zip(items, sizes):
                        # the per-frame entries are named, not inferred from a leading dimension
                        part = {k: (v[k0:k0 + n] if k in self.PER_FRAME_KEYS else v) for k, v in info.items()}
                        prepared.append((t, tsf_inputs[k0:k0 + n], part, k0, n))
                        k0 += n
                else:
                    whole_inputs = whole_T = None
                    for chunk, t in items:
                        tsf_inputs = self.transfer_params_by_smpl(chunk, cam_strategy, t=t)
                        prepared.append((t, tsf_inputs, self.tsf_info, 0, int(tsf_inputs.shape[0])))
                ready = torch.cuda.Event()
                ready.record(side)
            # generator calls: `fuse` consecutive batches of the round as ONE launch sequence (their inputs are adjacent
            # slices of the round's tensors): the trunk convolutions of 16 frames are 256 tiles of 8 x 32 pixels -- twice
            # the MFMAs per weight stage of the 4 x 32 tiles 8 frames leave room for (conv3x3_halo_bf16x3)
            # (a short round -- the tail of a sequence, or a timed window that is not a multiple of the round -- is still dealt to
            # EVERY lane: four batches left are two sequences of two, not one of four on one lane with the other idle)
            per = max(1, min(fuse, -(-len(prepared) // nl)))
            groups, k = [], 0
            while k < len(prepared):
                g = [k]
                while whole_inputs is not None and not self._opt.front_warp and len(g) < per and k + len(g) < len(prepared):
                    g.append(k + len(g))
                groups.append(g)
                k += len(g)
            out = []
            for gi, g in enumerate(groups):
                st, gen = lane_list[gi % nl]   # call gi of the round goes to lane gi mod lanes, in order
                with torch.cuda.stream(st):
                    st.wait_event(ready)
                    if len(g) == 1:
                        t, tsf_inputs, info, _, _ = prepared[g[0]]
                        self.tsf_info = info
                        preds = [self.forward(tsf_inputs, info['T'], generator=gen)]
                    else:
                        lo, hi = prepared[g[0]][3], prepared[g[-1]][3] + prepared[g[-1]][4]
                        self.tsf_info = prepared[g[0]][2]
                        both = self.forward(whole_inputs[lo:hi], whole_T[lo:hi], generator=gen)
                        preds = [both[prepared[j][3] - lo:prepared[j][3] - lo + prepared[j][4]] for j in g]
                    for j in g:
                        for v in [prepared[j][1]] + [x for x in prepared[j][2].values() if torch.is_tensor(x)]:
                            v.record_stream(st)   # allocated under the side stream, consumed here ...
                            v.record_stream(main)  # ... and by whoever reads tsf_info after the yield
                    done = torch.cuda.Event()
                    done.record(st)
                for j, p in zip(g, preds):
                    out.append((prepared[j][0], p, prepared[j][2], done))
            return out

        def rounds():
            group = []
            for item in batches:
                group.append(item)
                if len(group) == nl * depth:
                    yield group
                    group = []
            if group:
                yield group

        pending, prev_done = None, []
        try:
            for group in rounds():
                cur = enqueue_round(group, prev_done)
                prev_done = [e[3] for e in cur]
                if pending is not None:
                    for t, preds, info, done in pending:
                        main.wait_event(done)
                        preds.record_stream(main)
                        self.tsf_info = info
                        yield t, preds
                pending = cur
            if pending is not None:
                for t, preds, info, done in pending:
                    main.wait_event(done)
                    preds.record_stream(main)
                    self.tsf_info = info
                    yield t, preds
                pending = None
        finally:
            # whatever the consumer does next on its stream (also after leaving the loop early) comes after the lanes
            for ev in prev_done:
                main.wait_event(ev)
            main.wait_stream(side)

    # ------------------------------------------------------------------ drivers (imitator.py:157-214)
    def _run_batches(self, tgt_smpls, cam_strategy, on_batch):
        if len(tgt_smpls) == 0:      # the reference's loop over range(0) (imitator.py:166,196): nothing to do
            return []
        smpls = torch.as_tensor(np.asarray(tgt_smpls), dtype=torch.float32).reshape(len(tgt_smpls), -1)
        bs = max(1, int(self._opt.batch_size))
        outputs = []
        if cam_strategy == 'smooth' and len(smpls):
            self.first_cam = smpls[0:1, 0:3].clone().cuda()
        smpls = smpls.cuda()
        for s, preds in self.predict_batches(((smpls[s:s + bs], s) for s in range(0, len(smpls), bs)), cam_strategy):
            host = preds.permute(0, 2, 3, 1).cpu().numpy()   # one device->host copy per batch
            for i in range(host.shape[0]):
                outputs.append(host[i])
                on_batch(s + i, host[i], preds[i:i + 1])
        return outputs

    @torch.no_grad()
    def inference(self, tgt_paths, tgt_smpls=None, cam_strategy='smooth', output_dir='', visualizer=None,
                  verbose=True):
        """imitator.py:157-189 -> list of (H,W,3) float arrays in [-1,1]."""
        if tgt_smpls is None:
            raise NotImplementedError("estimating SMPL from target images needs the HMR regressor; pass tgt_smpls")

        def sink(t, pred, pred_dev):
            if visualizer is not None:
                visualizer.vis_named_img('pred_' + cam_strategy, pred_dev)
            if output_dir:
                filename = os.path.split(tgt_paths[t])[-1] if tgt_paths and tgt_paths[t] else 'pred_%.8d.jpg' % t
                cv_utils.save_cv2_img(pred, os.path.join(output_dir, 'pred_' + filename), normalize=True)

        return self._run_batches(tgt_smpls, cam_strategy, sink)

    @torch.no_grad()
    def inference_by_smpls(self, tgt_smpls, cam_strategy='smooth', output_dir='', visualizer=None):
        """imitator.py:191-214."""
        def sink(t, pred, pred_dev):
            if visualizer is not None:
                visualizer.vis_named_img('pred_' + cam_strategy, pred_dev)
            if output_dir:
                cv_utils.save_cv2_img(pred, os.path.join(output_dir, 'pred_%.8d.jpg' % t), normalize=True)

        return self._run_batches(tgt_smpls, cam_strategy, sink)

    def post_personalize(self, *args, **kwargs):
        raise NotImplementedError("post_personalize is a fine-tuning (training) loop (imitator.py:344-472): out of scope")
