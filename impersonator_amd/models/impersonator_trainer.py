"""Training model -- the first slice of SURVEY.md 8f row 4: everything of a training iteration except the generator's
own update.

Mirrors the reference's training model (models/impersonator_trainer.py, class Impersonator): `_create_generator` /
`_create_discriminator` (:215-222), the Adam settings of `_init_train_vars` (:224-232), `forward` (:329-348: the
three-stream generator pass and the background blends), `_optimize_D` (:396-411) and `_compute_loss_D` (:413-414).
The generator-side update (backward through the ResUnet, the Liquid Warping Block and the blends, VGG / face losses
with their absent pretrained nets) is not implemented yet; asking for it fails loudly."""
import torch

from ..networks.discriminator import PatchDiscriminator
from ..networks.generator import ImpersonatorGenerator
from .models import BaseModel


class Impersonator(BaseModel):
    def __init__(self, opt):
        super(Impersonator, self).__init__(opt)
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

    def _create_generator(self):
        # impersonator_trainer.py:215-217
        return ImpersonatorGenerator(bg_dim=4, src_dim=3 + self._G_cond_nc, tsf_dim=3 + self._G_cond_nc,
                                     repeat_num=getattr(self._opt, 'repeat_num', 6), image_size=self._opt.image_size,
                                     max_batch=getattr(self._opt, 'batch_size', 4)).cuda()

    def _create_discriminator(self):
        # impersonator_trainer.py:219-222
        return PatchDiscriminator(input_nc=3 + self._D_cond_nc, norm_type=getattr(self._opt, 'norm_type', 'instance'), ndf=64,
                                  n_layers=4, use_sigmoid=False, image_size=self._opt.image_size,
                                  max_batch=getattr(self._opt, 'batch_size', 4)).cuda()

    def set_input(self, input_G_tsf, real_tsf, input_G_bg=None, input_G_src=None, T=None):
        """The tensors a training iteration reads (impersonator_trainer.py:300-319), as the reference's BodyRecoveryFlow
        (`self._bdr`) produces them: the generator inputs of the three streams, the flow T and the real target image.
        `_optimize_D` alone needs input_G_tsf and real_tsf."""
        self._input_G_tsf, self._real_tsf = input_G_tsf, real_tsf
        self._input_G_bg, self._input_G_src, self._T = input_G_bg, input_G_src, T

    @torch.no_grad()
    def forward(self, keep_data_for_visuals=False, return_estimates=False):
        """impersonator_trainer.py:329-348 (bg_both=False): the generator pass and the blends onto the inpainted
        background -> (fake_bg, fake_src_imgs, fake_tsf_imgs, fake_masks).  No autograd graph."""
        fake_bg, fake_src_color, fake_src_mask, fake_tsf_color, fake_tsf_mask = \
            self._G.forward(self._input_G_bg, self._input_G_src, self._input_G_tsf, T=self._T)
        bs = fake_src_color.shape[0]
        fake_src_bg = fake_bg[0:bs]
        fake_src_imgs = fake_src_mask * fake_src_bg + (1 - fake_src_mask) * fake_src_color
        fake_tsf_imgs = fake_tsf_mask * fake_src_bg + (1 - fake_tsf_mask) * fake_tsf_color
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

    def optimize_parameters(self, *args, **kwargs):
        raise NotImplementedError("generator-side training is not implemented yet (SURVEY.md 8f row 4); "
                                  "the discriminator update is available as _optimize_D")
