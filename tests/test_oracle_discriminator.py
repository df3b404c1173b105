"""CPU: the discriminator-update oracle (oracle/torch_ref.py) against the golden vectors produced by the REAL reference
code (tests/golden/make_golden.py::make_discriminator), and the data-parallel gradient averaging over gloo."""
import numpy as np
import torch

from oracle import torch_ref
from tests import helpers


def _inputs():
    gen = torch.Generator().manual_seed(1)
    real = torch.rand(2, 6, 64, 64, generator=gen) * 2 - 1
    fake = torch.rand(2, 6, 64, 64, generator=gen) * 2 - 1
    return real, fake


def test_oracle_reproduces_reference_golden():
    g = helpers.golden("discriminator_golden.npz")
    sd = helpers.discriminator_state_dict(seed=3)
    real, fake = _inputs()
    with torch.no_grad():
        assert np.allclose(torch_ref.discriminator_forward(sd, real).numpy(), g["d_real"], atol=1e-6, rtol=1e-5)
    losses, grads, params = torch_ref.discriminator_train_steps(sd, [(real, fake)])
    assert abs(losses[0] - float(g["loss"][0])) < 1e-6
    for k in sd:
        gr = grads[k].double()
        ref = g["gnorm/" + k]
        assert abs(gr.abs().sum().item() - ref[0]) <= 1e-4 * ref[0] + 1e-9, k
        assert np.allclose(grads[k].flatten()[::97].numpy(), g["gsample/" + k], atol=1e-8, rtol=1e-4), k
        assert np.allclose(params[k].flatten()[::97].numpy(), g["psample/" + k], atol=1e-7, rtol=1e-5), k


def _avg_worker(rank, world, port, ret):
    import torch.distributed as dist
    from impersonator_amd import sharding
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    g = torch.full((1000,), float(rank + 1))
    sharding.average_gradients(g)
    ret[rank] = g.clone()
    dist.destroy_process_group()


def test_gradient_averaging_gloo_world2():
    """What PatchDiscriminator.optimize_D does between backward and the Adam step in a data-parallel job."""
    import torch.multiprocessing as mp
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_avg_worker, args=(2, port, ret), nprocs=2, join=True)
    assert torch.equal(ret[0], ret[1]) and torch.allclose(ret[0], torch.full((1000,), 1.5))
