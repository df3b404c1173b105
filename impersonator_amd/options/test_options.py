"""Command-line flags of the inference path, names and defaults as in the reference
(options/base_options.py:13-70, options/test_options.py:5-49)."""
import argparse
import os


class BaseOptions(object):
    def __init__(self):
        self._parser = argparse.ArgumentParser()
        self._initialized = False

    def initialize(self):
        p = self._parser
        p.add_argument('--data_dir', type=str, default='', help='path to dataset')
        p.add_argument('--dataset_mode', type=str, default='iPER')
        p.add_argument('--checkpoints_dir', type=str, default='./outputs/checkpoints/')
        p.add_argument('--load_epoch', type=int, default=-1)
        p.add_argument('--load_path', type=str, default='./outputs/checkpoints/lwb_imper_fashion_place/net_epoch_30_id_G.pth')
        p.add_argument('--batch_size', type=int, default=4, help='frames per launch sequence')
        p.add_argument('--time_step', type=int, default=10)
        p.add_argument('--tex_size', type=int, default=3)
        p.add_argument('--image_size', type=int, default=256)
        p.add_argument('--repeat_num', type=int, default=6)
        p.add_argument('--cond_nc', type=int, default=3)
        p.add_argument('--map_name', type=str, default='uv_seg')
        p.add_argument('--uv_mapping', type=str, default='./assets/pretrains/mapper.txt')
        p.add_argument('--part_info', type=str, default='./assets/pretrains/smpl_part_info.json')
        p.add_argument('--front_info', type=str, default='./assets/pretrains/front_facial.json')
        p.add_argument('--head_info', type=str, default='./assets/pretrains/head.json')
        p.add_argument('--hmr_model', type=str, default='./assets/pretrains/hmr_tf2pt.pth')
        p.add_argument('--smpl_model', type=str, default='./assets/pretrains/smpl_model.pkl')
        p.add_argument('--face_model', type=str, default='./assets/pretrains/sphere20a_20171020.pth')
        p.add_argument('--gpu_ids', type=str, default='0')
        p.add_argument('--name', type=str, default='running')
        p.add_argument('--model', type=str, default='impersonator')
        p.add_argument('--gen_name', type=str, default='impersonator')
        p.add_argument('--norm_type', type=str, default='instance')
        p.add_argument('--serial_batches', action='store_true')
        self._initialized = True

    def parse(self, args=None):
        if not self._initialized:
            self.initialize()
        self._opt = self._parser.parse_args(args)
        self._opt.is_train = self.is_train
        # options/base_options.py:119-125: the reference exports CUDA_VISIBLE_DEVICES from --gpu_ids
        if self._opt.gpu_ids and 'LOCAL_RANK' not in os.environ:
            os.environ.setdefault('HIP_VISIBLE_DEVICES', self._opt.gpu_ids)
        return self._opt


class TestOptions(BaseOptions):
    is_train = False

    def initialize(self):
        super().initialize()
        p = self._parser
        p.add_argument('--output_dir', type=str, default='./outputs/results/')
        p.add_argument('--src_path', type=str, default='')
        p.add_argument('--tgt_path', type=str, default='')
        p.add_argument('--pri_path', type=str, default='./assets/samples/A_priors/imgs')
        p.add_argument('--bg_model', type=str, default='./outputs/checkpoints/deepfillv2/net_epoch_50_id_G.pth')
        p.add_argument('--bg_ks', default=13, type=int)
        p.add_argument('--ft_ks', default=3, type=int)
        p.add_argument('--only_vis', action='store_true', default=False)
        p.add_argument('--has_detector', action='store_true', default=False)
        p.add_argument('--bg_replace', action='store_true', default=False)
        p.add_argument('--post_tune', action='store_true', default=False)
        p.add_argument('--front_warp', action='store_true', default=False)
        p.add_argument('--cam_strategy', type=str, default='smooth', choices=['smooth', 'source', 'copy'])
        p.add_argument('--save_res', action='store_true', default=False)
        p.add_argument('--swap_part', type=str, default='body')
        p.add_argument('--ip', type=str, default='')
        p.add_argument('--port', type=int, default=31100)
        p.add_argument('--align_corners', action='store_true', default=False,
                       help='grid_sample semantics of torch 1.2 (what 2019 checkpoints were trained with)')
        p.add_argument('--synthetic', action='store_true', default=False,
                       help='run with the seeded synthetic body model / weights (no downloaded assets)')
        p.add_argument('--num_frames', type=int, default=16, help='synthetic target frames')
