"""Training model, discriminator side -- the first slice of SURVEY.md 8f row 4.

Mirrors the parts of the reference's training model (models/impersonator_trainer.py, class Impersonator) that concern
the PatchGAN discriminator: `_create_discriminator` (:219-222), the Adam settings of `_init_train_vars` (:224-232),
`_optimize_D` (:396-411) and `_compute_loss_D` (:413-414).  The generator-side update (backward through the ResUnet,
the Liquid Warping Block and the rasteriser-fed inputs, VGG / face losses) is not implemented yet; asking for it fails
loudly."""
import torch

from ..networks.discriminator import PatchDiscriminator
from .models import BaseModel


class Impersonator(BaseModel):
    def __init__(self, opt):
        super(Impersonator, self).__init__(opt)
        self._name = 'Impersonator'
        self._D_cond_nc = self._G_cond_nc          # models/models.py:85-94: same condition map for G and D
        if getattr(opt, 'lambda_D_prob', 1) != 1:
            raise NotImplementedError("lambda_D_prob != 1")
        self._D = self._create_discriminator()
        self._current_lr_D = getattr(opt, 'lr_D', 0.0002)                                    # train_options.py:36
        self._D_betas = (getattr(opt, 'D_adam_b1', 0.5), getattr(opt, 'D_adam_b2', 0.999))   # train_options.py:37-38
        self._input_G_tsf = None
        self._real_tsf = None
        self._d_loss = None

    def _create_discriminator(self):
        # impersonator_trainer.py:219-222
        return PatchDiscriminator(input_nc=3 + self._D_cond_nc, norm_type=getattr(self._opt, 'norm_type', 'instance'), ndf=64,
                                  n_layers=4, use_sigmoid=False, image_size=self._opt.image_size,
                                  max_batch=getattr(self._opt, 'batch_size', 4)).cuda()

    def set_input(self, input_G_tsf, real_tsf):
        """The two tensors `_optimize_D` reads (impersonator_trainer.py:397-399): the generator's tsf input
        (warped image + condition map, 3 + cond_nc channels) and the real target image."""
        self._input_G_tsf, self._real_tsf = input_G_tsf, real_tsf

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
