"""ctypes binding of liblwg.so (include/lwg.h).  Fails loudly: no library, no product path."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# LWG_LIB=exp: the measurement build (python -m impersonator_amd.build --experiments; knock-out switches, tools/ only);
# LWG_LIB=<tag>: _C/liblwg_<tag>.so, e.g. a copy of the previous build kept for an A/B on one box (with LWG_ALLOW_STALE_LIB=1)
_LIB_TAG = re.sub(r"[^A-Za-z0-9]", "", os.environ.get("LWG_LIB", ""))
LIB_PATH = os.path.join(_HERE, "_C", "liblwg_%s.so" % _LIB_TAG if _LIB_TAG else "liblwg.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "lwg.h")

LWG_OK = 0
ERR_NAMES = {-1: "LWG_ERR_INVALID_ARG", -2: "LWG_ERR_UNSUPPORTED", -3: "LWG_ERR_WORKSPACE", -4: "LWG_ERR_HIP",
             -5: "LWG_ERR_STATE"}


class LwgError(RuntimeError):
    """A liblwg entry point returned a negative status (the reference raised RuntimeError via AT_CHECK)."""

    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "LWG_ERR"), code, message))
        self.code = code


_c = ctypes
_vp, _i, _f, _sz = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t
_PROTOS = {
    "lwg_conv_trace": (_i, [_vp, _sz]),
    "lwg_conv_trace_launch": (_i, [_i, _c.POINTER(_c.c_longlong)]),
    "lwg_version": (_i, []),
    "lwg_last_error": (_c.c_char_p, []),
    "lwg_device_info": (_i, [_c.POINTER(_i), _c.POINTER(_sz), _c.c_char_p, _sz]),
    "lwg_project_faces": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp]),
    "lwg_rasterize_workspace_bytes": (_sz, [_i, _i, _i]),
    "lwg_rasterize_fim_wim": (_i, [_vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "lwg_encode_fim": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "lwg_cal_bc_transform": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "lwg_grid_sample": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "lwg_resize_flow": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "lwg_transfer_workspace_bytes": (_sz, [_i, _i, _i]),
    "lwg_transfer_frame": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp, _vp, _sz, _vp]),
    "lwg_smpl_swap": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lwg_smpl_project_joints": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "lwg_rotate_translate": (_i, [_vp, _c.c_long, _vp, _vp, _vp, _vp]),
    "lwg_smpl_workspace_bytes": (_sz, [_i]),
    "lwg_smpl_forward": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "lwg_smpl_forward_f64": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "lwg_morph": (_i, [_vp, _i, _i, _i, _c.c_long, _i, _i, _i, _vp, _vp]),
    "lwg_mask_compose": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "lwg_source_p2verts": (_i, [_vp, _i, _i, _vp, _vp]),
    "lwg_vis_f2pts_workspace_bytes": (_sz, [_i, _i]),
    "lwg_vis_f2pts": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "lwg_pack_nhwc": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "lwg_unpack_nchw": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "lwg_generator_create": (_i, [_c.POINTER(_vp), _i, _i, _i, _i, _i, _i]),
    "lwg_generator_destroy": (None, [_vp]),
    "lwg_generator_set_precision": (_i, [_vp, _i]),
    "lwg_generator_encode_src_n": (_i, [_vp, _vp, _i, _c.POINTER(_vp), _vp]),
    "lwg_generator_decode_src": (_i, [_vp, _c.POINTER(_vp), _i, _vp, _vp, _vp]),
    "lwg_generator_inference_n": (_i, [_vp, _vp, _i, _vp, _i, _c.POINTER(_vp), _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "lwg_generator_enable_bg": (_i, [_vp, _i]),
    "lwg_generator_bg_forward": (_i, [_vp, _vp, _i, _vp, _vp]),
    "lwg_generator_load_weight": (_i, [_vp, _c.c_char_p, _vp, _c.POINTER(_c.c_int64), _i]),
    "lwg_generator_missing_weights": (_i, [_vp]),
    "lwg_generator_num_src_features": (_i, [_vp]),
    "lwg_generator_src_feature_shape": (_i, [_vp, _i, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i)]),
    "lwg_generator_encode_src": (_i, [_vp, _vp, _c.POINTER(_vp), _vp]),
    "lwg_generator_inference": (_i, [_vp, _vp, _i, _vp, _i, _c.POINTER(_vp), _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "lwg_generator_swap": (_i, [_vp, _vp, _i, _vp, _vp, _i, _c.POINTER(_vp), _c.POINTER(_vp), _i, _vp, _vp, _vp, _i, _vp,
                                _vp]),
    "lwg_inpaint_create": (_i, [_c.POINTER(_vp), _i, _i]),
    "lwg_inpaint_destroy": (None, [_vp]),
    "lwg_inpaint_load_weight": (_i, [_vp, _c.c_char_p, _vp, _c.POINTER(_c.c_int64), _i]),
    "lwg_inpaint_missing_weights": (_i, [_vp]),
    "lwg_inpaint_set_precision": (_i, [_vp, _i]),
    "lwg_inpaint_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lwg_conv2d_workspace_bytes": (_c.c_size_t, [_vp]),
    "lwg_conv2d_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _c.c_size_t, _vp]),
    "lwg_conv2d_backward_data": (_i, [_vp, _vp, _vp, _vp, _vp, _c.c_size_t, _vp]),
    "lwg_conv2d_backward_weight": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _c.c_size_t, _vp]),
    "lwg_heads_workspace_bytes": (_c.c_size_t, [_i, _i, _i]),
    "lwg_heads_forward": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _c.c_size_t, _vp]),
    "lwg_heads_backward_weight": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _c.c_size_t, _vp]),
    "lwg_discriminator_input_grad": (_i, [_vp, _vp, _i, _c.c_float, _vp, _vp, _vp]),
    "lwg_grid_sample_nhwc": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "lwg_instance_norm_scratch_bytes": (_c.c_size_t, [_i, _i, _i]),
    "lwg_instance_norm_forward": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "lwg_instance_norm_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "lwg_grid_sample_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "lwg_swap_masks": (_i, [_vp, _i, _i, _i, _c.c_uint, _c.c_uint, _vp, _vp, _vp, _vp, _vp]),
    "lwg_mask_faces": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "lwg_swap_compose": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "lwg_clamp": (_i, [_vp, _c.c_size_t, _c.c_float, _c.c_float, _vp]),
    "lwg_max_abs_diff": (_i, [_vp, _vp, _c.c_size_t, _vp, _vp]),
    "lwg_grid_sample_plan_bytes": (_c.c_size_t, [_i, _i, _i, _i, _i, _i]),
    "lwg_grid_sample_plan": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _c.c_size_t, _vp]),
    "lwg_grid_sample_backward_planned": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _c.c_size_t, _vp, _vp]),
    "lwg_adam_update": (_i, [_vp, _vp, _vp, _vp, _c.c_size_t, _c.c_long, _c.c_float, _c.c_float, _c.c_float, _c.c_float, _vp]),
    "lwg_adam_update_device_step": (_i, [_vp, _vp, _vp, _vp, _c.c_size_t, _vp, _c.c_float, _c.c_float, _c.c_float, _c.c_float, _vp]),
    "lwg_discriminator_use_device_step": (_i, [_vp, _i, _c.c_float, _c.c_float]),
    "lwg_discriminator_create": (_i, [_c.POINTER(_vp), _i, _i, _i, _i, _i]),
    "lwg_discriminator_set_precision": (_i, [_vp, _i]),
    "lwg_discriminator_destroy": (None, [_vp]),
    "lwg_discriminator_load_weight": (_i, [_vp, _c.c_char_p, _vp, _c.POINTER(_c.c_int64), _i]),
    "lwg_discriminator_read_weight": (_i, [_vp, _c.c_char_p, _i, _vp, _c.c_size_t]),
    "lwg_discriminator_num_params": (_i, [_vp, _c.POINTER(_c.c_size_t)]),
    "lwg_discriminator_output_size": (_i, [_vp, _c.POINTER(_i)]),
    "lwg_discriminator_forward": (_i, [_vp, _vp, _i, _vp, _vp]),
    "lwg_discriminator_backward": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "lwg_discriminator_buffers": (_i, [_vp, _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_c.c_size_t)]),
    "lwg_discriminator_adam_step": (_i, [_vp, _c.c_float, _c.c_float, _c.c_float, _c.c_float, _vp]),
    "lwg_generator_peek": (_i, [_vp, _i, _vp, _sz, _vp]),
    "lwg_generator_profile": (_i, [_vp, _i]),
    "lwg_generator_profile_variants": (_i, []),
    "lwg_generator_profile_variant_name": (_c.c_char_p, [_i]),
    "lwg_generator_profile_read": (_i, [_vp, _i, _c.POINTER(_i), _c.POINTER(_c.c_double), _c.POINTER(_c.c_double)]),
}

_lib = None


def header_symbols():
    """Every entry point include/lwg.h declares (used by the ABI test)."""
    with open(HEADER_PATH) as fh:
        text = fh.read()
    return sorted(set(re.findall(r"LWG_API[^;]*?\b(lwg_[a-z0-9_]+)\s*\(", text)))


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "liblwg.so is missing (%s). Build it with `python -m impersonator_amd.build` (needs hipcc); "
            "there is no CPU fallback for the hot path." % LIB_PATH)
    # liblwg works on PyTorch's device memory and streams, so it must share PyTorch's HIP runtime instance:
    # import torch first so that its libamdhip64 is the one already mapped when liblwg's dependency is resolved
    # (loading liblwg first binds a second runtime that sees no device context: "no ROCm-capable device").
    import torch  # noqa: F401
    # a library older than the kernel sources next to it must not run silently (the build stamps what it compiled)
    stamp = os.path.join(_HERE, "_C", "liblwg_%s.sha256" % _LIB_TAG if _LIB_TAG else "liblwg.sha256")
    if os.path.isdir(os.path.join(_HERE, "csrc")) and os.path.exists(stamp) and not os.environ.get("LWG_ALLOW_STALE_LIB"):
        from . import build as _build
        if open(stamp).read().strip() != _build._digest():
            raise ImportError("liblwg.so was built from other sources than impersonator_amd/csrc + include/ hold now: "
                              "run `python -m impersonator_amd.build` (LWG_ALLOW_STALE_LIB=1 overrides)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != LWG_OK:
        raise LwgError(rc, load().lwg_last_error().decode(errors="replace"))


def ptr(t):
    """Device (or host) address of a torch tensor, None -> NULL."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr
