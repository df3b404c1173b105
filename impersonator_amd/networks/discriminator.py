"""PatchGAN discriminator on MI355X: the reference's module surface plus its optimiser step, liblwg underneath.

Mirrors networks/discriminator.py:8-57 of the reference (`PatchDiscriminator`, as the trainer builds it at
models/impersonator_trainer.py:219-222: norm_type='instance', n_layers=4, use_sigmoid=False) with identical
`state_dict` keys, and folds what the trainer does around it for the discriminator update -- `_optimize_D` (:396-411),
`loss.backward()`, `torch.optim.Adam.step()` (:231-232) -- into `optimize_D`.  The nn modules below only hold the
parameters; forward, backward and Adam run in train.hip on flat device buffers.  In a data-parallel job the gradient
buffer is all-reduced with torch.distributed (RCCL on GPUs) between backward and the Adam step: the one collective of
the training path."""
import ctypes

import torch
import torch.nn as nn

from .. import _lib, sharding
from .networks import NetworkBase


class _DeviceView(object):
    """Zero-copy torch view of a raw device buffer (float32, 1-D) through __cuda_array_interface__."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class PatchDiscriminator(NetworkBase):
    def __init__(self, input_nc, ndf=64, n_layers=3, norm_type='batch', use_sigmoid=False, image_size=256, max_batch=8,
                 conv_precision='fp32'):
        """conv_precision (extension): 'fp32' or 'bf16x3' -- the arithmetic of the convolutions and their gradients
        (include/lwg.h, lwg_discriminator_set_precision); norms, activations, loss and Adam are fp32 in both."""
        super().__init__()
        if conv_precision not in ('fp32', 'bf16x3'):
            raise ValueError("conv_precision must be 'fp32' or 'bf16x3'")
        self.conv_precision = conv_precision
        self._name = 'discriminator_patch_gan'
        if norm_type != 'instance':
            raise NotImplementedError("normalization layer [%s]: the MI355X build implements the trainer's default, "
                                      "'instance' (options/base_options.py:51)" % norm_type)
        if use_sigmoid:
            raise NotImplementedError("use_sigmoid=True is not used by the trainer (impersonator_trainer.py:221)")
        self.input_nc, self.ndf, self.n_layers = input_nc, ndf, n_layers
        self.image_size, self.max_batch = image_size, max_batch
        # same nn.Sequential indices as discriminator.py:29-49, so the state_dict keys are the reference's
        seq = [nn.Conv2d(input_nc, ndf, 4, 2, 1), nn.LeakyReLU(0.2, True)]
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            seq += [nn.Conv2d(ndf * prev, ndf * mult, 4, 2, 1), nn.InstanceNorm2d(ndf * mult, affine=False), nn.LeakyReLU(0.2, True)]
        prev, mult = mult, min(2 ** n_layers, 8)
        seq += [nn.Conv2d(ndf * prev, ndf * mult, 4, 1, 1), nn.InstanceNorm2d(ndf * mult, affine=False), nn.LeakyReLU(0.2, True)]
        seq += [nn.Conv2d(ndf * mult, 1, 4, 1, 1)]
        self.model = nn.Sequential(*seq)
        self._handle = None
        self._uploaded = None

    # ---- handle / weights
    def _version(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _ensure_handle(self):
        lib = _lib.load()
        if self._handle is None:
            h = ctypes.c_void_p()
            _lib.check(lib.lwg_discriminator_create(ctypes.byref(h), self.input_nc, self.ndf, self.n_layers, self.image_size,
                                                    self.max_batch))
            self._handle = h
            self._uploaded = None
            _lib.check(lib.lwg_discriminator_set_precision(h, 1 if self.conv_precision == 'bf16x3' else 0))
        if self._uploaded != self._version():
            self.push_parameters()
        return self._handle

    def push_parameters(self):
        """Python-side parameters -> device master copy (resets nothing else: Adam moments are kept)."""
        lib = _lib.load()
        if self._handle is None:
            return self._ensure_handle()
        for k, v in self.state_dict().items():
            a = v.detach().float().cpu().contiguous()
            shape = (ctypes.c_int64 * a.dim())(*a.shape)
            _lib.check(lib.lwg_discriminator_load_weight(self._handle, k.encode(), ctypes.c_void_p(a.data_ptr()), shape, a.dim()))
        self._uploaded = self._version()

    def _read(self, from_grads):
        lib = _lib.load()
        out = {}
        for k, v in self.state_dict().items():
            a = torch.empty(v.shape, dtype=torch.float32)
            _lib.check(lib.lwg_discriminator_read_weight(self._handle, k.encode(), int(from_grads), ctypes.c_void_p(a.data_ptr()),
                                                         a.numel()))
            out[k] = a
        return out

    def pull_parameters(self):
        """Device master copy -> the nn.Parameters (call before state_dict()/saving a checkpoint after training steps)."""
        self._ensure_handle()
        with torch.no_grad():
            for k, a in self._read(False).items():
                self.state_dict()[k].copy_(a)
        self._uploaded = self._version()

    def gradients(self):
        """Last backward's gradients as {state_dict key: CPU tensor} (tests, debugging)."""
        self._ensure_handle()
        return self._read(True)

    def flat_buffers(self):
        """(params, grads): zero-copy 1-D CUDA tensors over the handle's flat device buffers (padding entries are 0)."""
        h = self._ensure_handle()
        p, g, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t()
        _lib.check(_lib.load().lwg_discriminator_buffers(h, ctypes.byref(p), ctypes.byref(g), ctypes.byref(n)))
        dev = torch.device("cuda", torch.cuda.current_device())
        return (torch.as_tensor(_DeviceView(p.value, n.value), device=dev), torch.as_tensor(_DeviceView(g.value, n.value), device=dev))

    def release(self):
        if self._handle is not None:
            _lib.load().lwg_discriminator_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    # ---- the reference's surface
    @torch.no_grad()
    def forward(self, input):
        """discriminator.py:55-57: (bs, input_nc, H, W) -> (bs, 1, h, h) patch map."""
        if not input.is_cuda:
            raise RuntimeError("PatchDiscriminator runs on the MI355X only (no CPU fallback)")
        h = self._ensure_handle()
        x = input.float().contiguous()
        if x.shape[1:] != (self.input_nc, self.image_size, self.image_size) or x.shape[0] > 2 * self.max_batch:
            raise ValueError("expected (<=%d, %d, %d, %d) input" % (2 * self.max_batch, self.input_nc, self.image_size, self.image_size))
        lib = _lib.load()
        ho = ctypes.c_int()
        _lib.check(lib.lwg_discriminator_output_size(h, ctypes.byref(ho)))
        outs = []
        for s in range(0, x.shape[0], self.max_batch):   # the handle's scratch is sized for 2*max_batch images per pass
            xb = x[s:s + self.max_batch].contiguous()
            out = torch.empty((xb.shape[0], 1, ho.value, ho.value), device=x.device, dtype=torch.float32)
            _lib.check(lib.lwg_discriminator_forward(h, _lib.ptr(xb), xb.shape[0], _lib.ptr(out), _lib.stream_ptr()))
            outs.append(out)
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)

    @torch.no_grad()
    def input_grad(self, x, target=0.0):
        """The generator's adversarial term (impersonator_trainer.py:369-371): (loss, d loss / d x) for
        loss = mean((D(x) - target)^2); the discriminator's own parameters get no gradient."""
        if not x.is_cuda:
            raise RuntimeError("PatchDiscriminator runs on the MI355X only (no CPU fallback)")
        h = self._ensure_handle()
        x = x.detach().float().contiguous()
        if x.shape[1:] != (self.input_nc, self.image_size, self.image_size) or x.shape[0] > self.max_batch:
            raise ValueError("expected (<=%d, %d, %d, %d) input" % (self.max_batch, self.input_nc, self.image_size, self.image_size))
        loss = torch.empty((), device=x.device, dtype=torch.float32)
        dx = torch.empty_like(x)
        _lib.check(_lib.load().lwg_discriminator_input_grad(h, _lib.ptr(x), x.shape[0], float(target), _lib.ptr(loss), _lib.ptr(dx),
                                                            _lib.stream_ptr()))
        return loss, dx

    @torch.no_grad()
    def optimize_D(self, real_input_D, fake_input_D, lr=0.0002, betas=(0.5, 0.999), eps=1e-8, all_reduce=True):
        """One discriminator update, impersonator_trainer.py:396-411 + backward + Adam (:231-232, train_options.py:36-38):
        loss = mean((D(real) - 1)^2) + mean((D(fake) + 1)^2).  Returns the loss (0-d CUDA tensor, before the update).
        With torch.distributed initialised and world_size > 1 the gradients are averaged over the ranks first."""
        if not (real_input_D.is_cuda and fake_input_D.is_cuda):
            raise RuntimeError("PatchDiscriminator runs on the MI355X only (no CPU fallback)")
        h = self._ensure_handle()
        real = real_input_D.detach().float().contiguous()
        fake = fake_input_D.detach().float().contiguous()
        bs = real.shape[0]
        if real.shape != fake.shape or real.shape[1:] != (self.input_nc, self.image_size, self.image_size):
            raise ValueError("real/fake must both be (bs, %d, %d, %d)" % (self.input_nc, self.image_size, self.image_size))
        lib = _lib.load()
        loss = torch.empty((), device=real.device, dtype=torch.float32)
        _lib.check(lib.lwg_discriminator_backward(h, _lib.ptr(real), _lib.ptr(fake), bs, _lib.ptr(loss), _lib.stream_ptr()))
        if all_reduce and sharding.collectives_active():
            sharding.average_gradients(self.flat_buffers()[1])
        _lib.check(lib.lwg_discriminator_adam_step(h, float(lr), float(betas[0]), float(betas[1]), float(eps), _lib.stream_ptr()))
        return loss
