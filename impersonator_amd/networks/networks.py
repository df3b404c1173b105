"""Network factory / base class with the reference's names (networks/networks.py:9-65)."""
import torch.nn as nn


class NetworkBase(nn.Module):
    def __init__(self):
        super().__init__()
        self._name = 'BaseNetwork'

    @property
    def name(self):
        return self._name

    def init_weights(self):
        """networks/networks.py:54-65: conv weights ~ N(0, 0.02); InstanceNorm affine left at (1, 0)."""
        for m in self.modules():
            cls = m.__class__.__name__
            if 'Conv' in cls and hasattr(m, 'weight'):
                m.weight.data.normal_(0.0, 0.02)
            elif 'BatchNorm2d' in cls:
                m.weight.data.normal_(1.0, 0.02)
                m.bias.data.fill_(0)


class NetworksFactory(object):
    """networks/networks.py:9-43, restricted to the networks on the Imitator.forward path."""

    @staticmethod
    def get_by_name(network_name, *args, **kwargs):
        if network_name == 'impersonator':
            from .generator import ImpersonatorGenerator
            return ImpersonatorGenerator(*args, **kwargs)
        if network_name == 'deepfillv2':
            from .inpaintor import InpaintSANet
            return InpaintSANet(*args, **kwargs)
        raise ValueError("Network %s is not part of the MI355X Imitator.forward path" % network_name)
