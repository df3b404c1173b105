"""Model factory / base class with the reference's names (models/models.py:7-203), inference subset."""
import os
from collections import OrderedDict

import torch


class ModelsFactory(object):
    @staticmethod
    def get_by_name(model_name, *args, **kwargs):
        if model_name == 'imitator':
            from .imitator import Imitator
            return Imitator(*args, **kwargs)
        if model_name == 'swapper':
            from .swapper import Swapper
            return Swapper(*args, **kwargs)
        if model_name == 'viewer':
            from .viewer import Viewer
            return Viewer(*args, **kwargs)
        raise ValueError("Model %s is not part of the MI355X Imitator.forward path" % model_name)


class BaseModel(object):
    def __init__(self, opt):
        self._name = 'BaseModel'
        self._opt = opt
        self._is_train = getattr(opt, 'is_train', False)
        self._G_cond_nc = self._cond_nc(getattr(opt, 'map_name', 'uv_seg'))

    @staticmethod
    def _cond_nc(map_name):
        # models/models.py:85-94 -> utils/mesh.py:446-473 (get_map_fn_dim): channel count of the face->condition table.
        # (3 + cond_nc <= 8 takes the fused NHWC8 input path; 'par' and 'binary' run through the NCHW input and the
        # 16- / 32-channel padded stem, lwg_generator_create.)
        dims = {'seg': 1, 'uv': 2, 'uv_seg': 3, 'par': 11, 'ids': 1, 'binary': 15}
        if map_name not in dims:
            raise ValueError('map name error {}'.format(map_name))
        return dims[map_name]

    @property
    def name(self):
        return self._name

    @staticmethod
    def _load_params(network, load_path, need_module=False):
        """models/models.py:159-179: load a checkpoint, stripping DataParallel's 'module.' prefix."""
        assert os.path.exists(load_path), \
            'Weights file not found. Have you trained a model!? We are not providing one %s' % load_path
        save_data = torch.load(load_path, map_location='cpu')
        if need_module:
            network.load_state_dict(save_data)
        else:
            state_dict = OrderedDict()
            for k, v in save_data.items():
                state_dict[k[7:] if 'module' in k else k] = v
            network.load_state_dict(state_dict)
        print('Loading net: %s' % load_path)
