"""Training model: one training iteration of the reference on MI355X.

Mirrors the reference's training model (models/impersonator_trainer.py, class Impersonator): `_create_generator` /
`_create_discriminator` (:215-222), the Adam settings of `_init_train_vars` (:224-232), `forward` (:329-348: the
three-stream generator pass and the background blends), `optimize_parameters` (:350-366), `_optimize_G` (:368-394) and
`_optimize_D` (:396-414).  The generator update is models/generator_trainer.py (hand-written backward pass on the
op-level HIP kernels), the discriminator update networks/discriminator.py.  Loss terms that need a downloaded network
(--use_vgg, --use_style, --use_face: VGG19, SphereFace) are not available and fail loudly."""
import torch

from .. import _lib
from ..networks.discriminator import PatchDiscriminator
from ..networks.generator import ImpersonatorGenerator
from ..networks.facenet import SphereFaceLoss
from ..networks.vgg import Vgg19Perceptual
from .generator_trainer import GeneratorTrainer
from .models import BaseModel


class BodyRecoveryFlow(torch.nn.Module):
    """models/impersonator_trainer.py:13-170: turns a (source image, target image, source SMPL, target SMPL) batch into
    the tensors a training iteration reads -- the three streams' generator inputs, the flow T, the crop masks and the
    head / body boxes.  Every device step is the inference path's: SMPL skinning (smpl.hip), the rasteriser with
    per-sample sources (raster.hip), `encode_fim`, `cal_bc_transform`, `grid_sample` (warp.hip) and `util.morph`."""

    def __init__(self, opt, hmr=None, render=None):
        super().__init__()
        self._name = 'BodyRecoveryFlow'
        self._opt = opt
        if hmr is None:
            from ..networks.batch_smpl import HumanModelRecovery
            hmr = HumanModelRecovery(smpl_pkl_path=opt.smpl_model)      # :22-27 (the image regressor's weights are not used)
        if render is None:
            from ..utils.nmr import SMPLRenderer
            render = SMPLRenderer(map_name=opt.map_name, uv_map_path=opt.uv_mapping, tex_size=opt.tex_size,
                                  image_size=opt.image_size, fill_back=False, anti_aliasing=True,
                                  background_color=(0, 0, 0), has_front=False)     # :29-36
        self._hmr, self._render = hmr, render

    @torch.no_grad()
    def forward(self, src_img, ref_img, src_smpl, ref_smpl):
        """:44-87, same return tuple."""
        from ..utils import util
        r = self._render
        src_info = self._hmr.get_details(src_smpl)
        ref_info = self._hmr.get_details(ref_smpl)
        src_f2verts, src_fim, _ = r.render_fim_wim(src_info['cam'], src_info['verts'])
        src_f2verts = src_f2verts[:, :, :, 0:2]
        src_f2verts[:, :, :, 1] *= -1
        src_cond, _ = r.encode_fim(src_info['cam'], src_info['verts'], fim=src_fim, transpose=True)
        src_crop_mask = util.morph(src_cond[:, -1:, :, :], ks=3, mode='erode')
        _, ref_fim, ref_wim = r.render_fim_wim(ref_info['cam'], ref_info['verts'])
        ref_cond, _ = r.encode_fim(ref_info['cam'], ref_info['verts'], fim=ref_fim, transpose=True)
        T = r.cal_bc_transform(src_f2verts, ref_fim, ref_wim)
        syn_img = r.grid_sample(src_img, T)
        input_G_src = torch.cat([src_img * (1 - src_crop_mask), src_cond], dim=1)
        input_G_tsf = torch.cat([syn_img, ref_cond], dim=1)
        src_bg_mask = util.morph(src_cond[:, -1:, :, :], ks=15, mode='erode')
        input_G_src_bg = torch.cat([src_img * src_bg_mask, src_bg_mask], dim=1)
        if getattr(self._opt, 'bg_both', False):
            ref_bg_mask = util.morph(ref_cond[:, -1:, :, :], ks=15, mode='erode')
            input_G_tsf_bg = torch.cat([ref_img * ref_bg_mask, ref_bg_mask], dim=1)
        else:
            input_G_tsf_bg = None
        tsf_crop_mask = util.morph(ref_cond[:, -1:, :, :], ks=3, mode='erode')
        return (input_G_src_bg, input_G_tsf_bg, input_G_src, input_G_tsf, T, src_crop_mask, tsf_crop_mask,
                self.cal_head_bbox(ref_info['j2d']), self.cal_body_bbox(ref_info['j2d']))

    def cal_head_bbox(self, kps):
        """:89-130: kps (N,19,2) in [-1,1] -> (N,4) long [min_x, max_x, min_y, max_y]; the head joints are 12.. (NECK_IDS)."""
        size = self._opt.image_size
        kps = (kps + 1) / 2.0
        zeros, ones = torch.zeros_like(kps[:, 12, 0]), torch.ones_like(kps[:, 12, 0])
        min_x = torch.max(torch.min(kps[:, 12:, 0] - 0.05, dim=1)[0], zeros)
        max_x = torch.min(torch.max(kps[:, 12:, 0] + 0.05, dim=1)[0], ones)
        min_y = torch.max(torch.min(kps[:, 12:, 1] - 0.05, dim=1)[0], zeros)
        max_y = torch.min(torch.max(kps[:, 12:, 1], dim=1)[0], ones)
        return torch.stack([(v * size).long() for v in (min_x, max_x, min_y, max_y)], dim=1)

    def cal_body_bbox(self, kps, factor=1.2):
        """:132-170."""
        size = self._opt.image_size
        kps = (kps + 1) / 2.0
        zeros = torch.zeros((kps.shape[0],), device=kps.device)
        ones = torch.ones((kps.shape[0],), device=kps.device)
        out = []
        for c in (0, 1):
            lo, hi = kps[:, :, c].min(dim=1)[0], kps[:, :, c].max(dim=1)[0]
            mid, ext = (lo + hi) / 2, (hi - lo) * factor
            out += [torch.max(zeros, mid - ext / 2), torch.min(ones, mid + ext / 2)]
        return torch.stack([(v * size).long() for v in out], dim=1)


class Impersonator(BaseModel):
    def __init__(self, opt, bdr=None):
        super(Impersonator, self).__init__(opt)
        self._bdr = bdr     # BodyRecoveryFlow (impersonator_trainer.py:201-207); built on first use when not injected
        self._head_bbox = self._body_bbox = None
        self._name = 'Impersonator'
        self._D_cond_nc = self._G_cond_nc          # models/models.py:85-94: same condition map for G and D
        if getattr(opt, 'lambda_D_prob', 1) != 1:
            raise NotImplementedError("lambda_D_prob != 1")
        self._G = self._create_generator()
        self._D = self._create_discriminator()
        self._current_lr_D = getattr(opt, 'lr_D', 0.0002)                                    # train_options.py:36
        self._D_betas = (getattr(opt, 'D_adam_b1', 0.5), getattr(opt, 'D_adam_b2', 0.999))   # train_options.py:37-38
        self._input_G_bg = self._input_G_src = self._input_G_tsf = self._T = None
        self._real_tsf = None
        self._d_loss = None
        # The loss networks are downloads of the reference; here their weights are handed in as files.
        #   --use_vgg : opt.vgg_weights = torch.save()d state_dict in torchvision's naming (what
        #               models.vgg19(pretrained=True).state_dict() is; the reference downloads it, networks/networks.py:133)
        #   --use_face: opt.face_model = the reference's own option (base_options.py:31, sphere20a_20171020.pth)
        #   --use_style: the Gram-matrix term over the same VGG19 (needs --vgg_weights too)
        self._face_state = None
        if getattr(opt, 'use_face', False):
            path = getattr(opt, 'face_model', None)
            if isinstance(path, dict):
                self._face_state = path
            else:
                import os
                if not path or not os.path.exists(path):
                    raise NotImplementedError("--use_face: --face_model %r does not exist (the Sphere20a checkpoint is a "
                                              "download of the reference)" % (path,))
                self._face_state = torch.load(path, map_location='cpu')
        self._vgg_state = None
        if getattr(opt, 'use_vgg', False) or getattr(opt, 'use_style', False):
            path = getattr(opt, 'vgg_weights', None)
            if not path:
                raise NotImplementedError("--use_vgg / --use_style: pass --vgg_weights <vgg19 state_dict .pth> (torchvision's "
                                          "vgg19(pretrained=True).state_dict(); there is no download here)")
            self._vgg_state = path if isinstance(path, dict) else torch.load(path, map_location='cpu')
        self._g_trainer = None
        self._graph = self._graph_terms = self._graph_lrs = None   # optimize_parameters_graphed
        self._graph_warm = 0
        self._graph_failed = False
        self._real_src = self._bg_mask = None

    def _generator_trainer(self):
        if self._g_trainer is None:
            o = self._opt
            self._g_trainer = GeneratorTrainer(
                self._G, self._D, lambda_D_prob=getattr(o, 'lambda_D_prob', 1), lambda_rec=getattr(o, 'lambda_rec', 10),
                lambda_tsf=getattr(o, 'lambda_tsf', 10), lambda_mask=getattr(o, 'lambda_mask', 0.1),
                lambda_mask_smooth=getattr(o, 'lambda_mask_smooth', 1e-5), lr=getattr(o, 'lr_G', 0.0002),
                betas=(getattr(o, 'G_adam_b1', 0.5), getattr(o, 'G_adam_b2', 0.999)),
                conv_precision=getattr(o, 'conv_precision', 'fp32'), mask_bce=getattr(o, 'mask_bce', False),
                bg_both=getattr(o, 'bg_both', False),
                vgg=(Vgg19Perceptual(self._vgg_state, getattr(o, 'conv_precision', 'fp32')) if self._vgg_state is not None else None),
                face=(SphereFaceLoss(self._face_state) if self._face_state is not None else None),
                lambda_face=getattr(o, 'lambda_face', 1), use_vgg=bool(getattr(o, 'use_vgg', False)),
                use_style=bool(getattr(o, 'use_style', False)), lambda_style=getattr(o, 'lambda_style', 5))
        return self._g_trainer

    def sync_generator(self):
        """Trained parameters -> the inference generator (self._G), e.g. before saving a checkpoint or running Imitator."""
        if self._g_trainer is not None:
            self._G.load_state_dict(self._g_trainer.state_dict())

    def _create_generator(self):
        # impersonator_trainer.py:215-217
        return ImpersonatorGenerator(bg_dim=4, src_dim=3 + self._G_cond_nc, tsf_dim=3 + self._G_cond_nc,
                                     repeat_num=getattr(self._opt, 'repeat_num', 6), image_size=self._opt.image_size,
                                     max_batch=getattr(self._opt, 'batch_size', 4)).cuda()

    def _create_discriminator(self):
        # impersonator_trainer.py:219-222
        return PatchDiscriminator(input_nc=3 + self._D_cond_nc, norm_type=getattr(self._opt, 'norm_type', 'instance'), ndf=64,
                                  n_layers=4, use_sigmoid=False, image_size=self._opt.image_size,
                                  max_batch=getattr(self._opt, 'batch_size', 4),
                                  conv_precision=getattr(self._opt, 'conv_precision', 'fp32')).cuda()

    @torch.no_grad()
    def set_input(self, input_G_tsf, real_tsf=None, input_G_bg=None, input_G_src=None, T=None, real_src=None, bg_mask=None,
                  head_bbox=None, body_bbox=None):
        """impersonator_trainer.py:289-319.  Called as the reference calls it -- `set_input(sample)` with
        sample['images'] (N,2,3,H,W) and sample['smpls'] (N,2,85): source / target pairs of a dataset batch -- the inputs
        are derived on the device by BodyRecoveryFlow.  Called with explicit tensors (extension) it takes what
        BodyRecoveryFlow would have produced: the generator inputs of the three streams, the flow T, the real images
        (`_optimize_D` alone needs input_G_tsf and real_tsf) and, for --use_face, `head_bbox` (N,4); the boxes of an
        earlier batch are never kept."""
        if isinstance(input_G_tsf, dict):
            sample = input_G_tsf
            images, smpls = sample['images'], sample['smpls']
            src_img, src_smpl = images[:, 0, ...].float().cuda(), smpls[:, 0, ...].float().cuda()
            tsf_img, tsf_smpl = images[:, 1, ...].float().cuda(), smpls[:, 1, ...].float().cuda()
            if self._bdr is None:
                self._bdr = BodyRecoveryFlow(self._opt)
            (input_G_src_bg, input_G_tsf_bg, input_G_src, input_G_tsf, T, src_crop_mask, tsf_crop_mask, self._head_bbox,
             self._body_bbox) = self._bdr(src_img, tsf_img, src_smpl, tsf_smpl)
            real_src, real_tsf = src_img, tsf_img
            bg_mask = torch.cat((src_crop_mask, tsf_crop_mask), dim=0)
            input_G_bg = (torch.cat([input_G_src_bg, input_G_tsf_bg], dim=0) if getattr(self._opt, 'bg_both', False)
                          else input_G_src_bg)   # impersonator_trainer.py:306-309
        else:
            self._head_bbox, self._body_bbox = head_bbox, body_bbox
        new = dict(_input_G_tsf=input_G_tsf, _real_tsf=real_tsf, _input_G_bg=input_G_bg, _input_G_src=input_G_src, _T=T,
                   _real_src=real_src, _bg_mask=bg_mask)
        if self._graph is not None:
            # a captured iteration reads the tensors it was captured on: the new batch is copied into them
            for k, v in new.items():
                cur = getattr(self, k)
                if (cur is None) != (v is None) or (v is not None and tuple(cur.shape) != tuple(v.shape)):
                    self.drop_graph()
                    break
            else:
                for k, v in new.items():
                    if v is not None:
                        getattr(self, k).copy_(v)
                return
        for k, v in new.items():
            setattr(self, k, v)

    @torch.no_grad()
    def forward(self, keep_data_for_visuals=False, return_estimates=False):
        """impersonator_trainer.py:329-348: the generator pass and the blends onto the inpainted background(s)
        -> (fake_bg, fake_src_imgs, fake_tsf_imgs, fake_masks).  No autograd graph."""
        fake_bg, fake_src_color, fake_src_mask, fake_tsf_color, fake_tsf_mask = \
            self._G.forward(self._input_G_bg, self._input_G_src, self._input_G_tsf, T=self._T)
        bs = fake_src_color.shape[0]
        fake_src_bg = fake_bg[0:bs]
        fake_tsf_bg = fake_bg[bs:] if getattr(self._opt, 'bg_both', False) else fake_src_bg
        fake_src_imgs = fake_src_mask * fake_src_bg + (1 - fake_src_mask) * fake_src_color
        fake_tsf_imgs = fake_tsf_mask * fake_tsf_bg + (1 - fake_tsf_mask) * fake_tsf_color
        return fake_bg, fake_src_imgs, fake_tsf_imgs, torch.cat([fake_src_mask, fake_tsf_mask], dim=0)

    def optimize_D_phase(self):
        """The second half of optimize_parameters (impersonator_trainer.py:350-366): generator pass, then the
        discriminator update on its output."""
        _, _, fake_tsf_imgs, _ = self.forward()
        return self._optimize_D(fake_tsf_imgs)

    @torch.no_grad()
    def _optimize_D(self, fake_tsf_imgs):
        """impersonator_trainer.py:396-411 followed by what optimize_parameters does with the result for D
        (:362-366: zero_grad, backward, optimizer step).  Returns the loss before the update."""
        tsf_cond = self._input_G_tsf[:, 3:]
        fake_input_D = torch.cat([fake_tsf_imgs.detach(), tsf_cond], dim=1)
        real_input_D = torch.cat([self._real_tsf, tsf_cond], dim=1)
        self._d_loss = self._D.optimize_D(real_input_D, fake_input_D, lr=self._current_lr_D, betas=self._D_betas)
        return self._d_loss

    def drop_graph(self):
        """Forget the captured iteration (new batch shape, new learning rate: both are baked into it)."""
        # the next graphed call warms up again before it captures: a new batch shape picks kernel variants (per-device function
        # attributes), scratch sizes and handles that must exist BEFORE a capture starts
        self._graph_warm = 0
        if self._graph is not None:
            self._graph = self._graph_terms = self._graph_batch = self._graph_lrs = None
            self._device_steps(False)

    def _device_steps(self, on):
        """Adam's step counts of G and D on the device (graph replay) or back on the host."""
        self._generator_trainer().use_device_step(on)
        b1, b2 = (float(self._D_betas[0]), float(self._D_betas[1])) if on else (0.0, 0.0)
        _lib.check(_lib.load().lwg_discriminator_use_device_step(self._D._ensure_handle(), 1 if on else 0, b1, b2))

    def optimize_parameters_graphed(self, warmup=2):
        """optimize_parameters() as ONE launch (extension): a training iteration is ~1500 kernel launches issued from Python at
        ~20 us each -- as long as the kernels themselves take.  The first `warmup` calls (after construction, and again after every
        drop_graph(): a new batch shape) run eagerly -- every lazily allocated buffer, handle and per-device kernel attribute exists
        afterwards -- the next one captures generator pass + update and discriminator update in a HIP graph (torch.cuda.graph;
        liblwg launches on torch's current stream, which is the capturing one), and every call from then on is a replay.
        What a replay cannot re-read from the host lives on the device: the batch (PRIVATE static tensors the capture reads;
        set_input copies every new batch into them, the caller's own tensors are never written), Adam's step counts
        (lwg_adam_update_device_step).  The learning rates are baked into the graph: a changed `_current_lr_D` / generator `lr`
        is noticed at the next call, which drops the graph and captures again.  Data-parallel jobs: the bucketed gradient
        all-reduces (sharding.GradientBuckets) and the discriminator's are captured with the iteration (RCCL collectives are
        graph-capturable; env LWG_GRAPH_COLLECTIVES=0 keeps multi-rank jobs eager).  If a capture fails for any reason the model
        falls back to eager iterations for good (host step counts restored).  Not with --use_face (its crops are host-side
        integers per batch).  Returns the same loss terms."""
        import os
        from .. import sharding
        if self._face_state is not None or self._graph_failed:
            return self.optimize_parameters()   # --use_face crops at the batch's head boxes: host integers, different every batch
        if sharding.collectives_active() and os.environ.get("LWG_GRAPH_COLLECTIVES", "1") == "0":
            return self.optimize_parameters()
        tr = self._generator_trainer()
        if self._graph is not None and self._graph_lrs != (float(self._current_lr_D), float(tr.lr)):
            self.drop_graph()                   # a learning-rate decay step: the rates are constants of the captured kernels
        if self._graph is None:
            if self._graph_warm < warmup:
                self._graph_warm += 1
                return self.optimize_parameters()
            names = ("_input_G_bg", "_input_G_src", "_input_G_tsf", "_T", "_real_src", "_real_tsf", "_bg_mask")
            caller = {k: getattr(self, k) for k in names}
            try:
                self._device_steps(True)
                # private static copies: what the graph reads and set_input overwrites
                for k, v in caller.items():
                    setattr(self, k, None if v is None else v.detach().clone())
                batch = dict(input_G_bg=self._input_G_bg, input_G_src=self._input_G_src, input_G_tsf=self._input_G_tsf, T=self._T,
                             real_src=self._real_src, real_tsf=self._real_tsf, bg_mask=self._bg_mask)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                # thread_local: ProcessGroupNCCL's watchdog thread polls the events of earlier collectives; under the default
                # (global) mode such a call from ANOTHER thread while this one captures is an error that kills the process
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    terms, (_, _, fake_tsf_imgs, _) = tr.optimize_G(batch)
                    terms = dict(terms, d_loss=self._optimize_D(fake_tsf_imgs))
                failed = None
            except RuntimeError as e:
                # what a capture can legitimately trip over is reported as RuntimeError (HIP / torch "operation not permitted when
                # stream is capturing", an allocation inside the capture, LwgError is one too); anything else -- a bug -- propagates
                failed = e
            if os.environ.get("LWG_GRAPH_STRICT") == "1" and failed is not None:
                raise failed
            # The fall-back is ONE decision for the whole job: a rank that replays a graph with captured collectives while another
            # issues them eagerly would hang both.  (The flag travels on the compute stream after the capture has ended.)
            if sharding.collectives_active():
                flag = torch.tensor([1.0 if failed is not None else 0.0], device=self._T.device)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
                any_failed = bool(flag.item() > 0)
            else:
                any_failed = failed is not None
            if any_failed:
                import warnings
                warnings.warn("optimize_parameters_graphed: capture failed (%s); continuing with eager iterations"
                              % ("%s: %s" % (type(failed).__name__, failed) if failed is not None else "on another rank"))
                self._graph = self._graph_terms = None
                self._graph_failed = True
                torch.cuda.synchronize()
                for k, v in caller.items():
                    setattr(self, k, v)
                self._device_steps(False)
                # the aborted capture advanced the generator trainer's pass state on the host (bucket counters, `untouched` keys):
                # the eager iteration below starts its own pass from scratch (backward() re-begins both)
                tr._buckets = None
                return self.optimize_parameters()
            self._graph, self._graph_terms = graph, terms
            self._graph_lrs = (float(self._current_lr_D), float(tr.lr))
        self._graph.replay()
        return {k: float(v) for k, v in self._graph_terms.items()}

    def optimize_parameters(self, trainable=True, keep_data_for_visuals=False):
        """impersonator_trainer.py:350-366: generator pass, generator update, then (trainable) the discriminator update on
        the images the generator produced before its update.  Returns the loss terms."""
        batch = dict(input_G_bg=self._input_G_bg, input_G_src=self._input_G_src, input_G_tsf=self._input_G_tsf, T=self._T,
                     real_src=self._real_src, real_tsf=self._real_tsf, bg_mask=self._bg_mask)
        if self._face_state is not None:
            if self._head_bbox is None:
                raise RuntimeError("--use_face needs the head boxes: call set_input(sample) (BodyRecoveryFlow) or set _head_bbox")
            batch['head_bbox'] = self._head_bbox
        terms, (_, _, fake_tsf_imgs, _) = self._generator_trainer().optimize_G(batch)
        losses = {k: float(v) for k, v in terms.items()}
        if trainable:
            losses['d_loss'] = float(self._optimize_D(fake_tsf_imgs))
        return losses
