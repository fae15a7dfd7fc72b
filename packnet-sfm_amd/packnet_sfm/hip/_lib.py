"""ctypes binding of libpnsfm_hip.so -- the C ABI declared in include/pnsfm.h.

The library is the product: there is NO fallback.  If the gfx950 build is missing (or is not a gfx950
build) importing any HIP-backed module fails loudly here.
"""
import ctypes
import os

import torch  # noqa: F401  -- MUST come first: it loads PyTorch-ROCm's own libamdhip64.so.7, which this library then
#                              shares; loading ours first would pull a second HIP runtime (/opt/rocm) into the process

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.abspath(os.path.join(_HERE, "..", "..", "csrc"))
LIB_PATH = os.path.join(_CSRC, "libpnsfm_hip.so")

_p = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/pnsfm.h one to one
SIGNATURES = {
    "pnsfm_version": (_i, []),
    "pnsfm_last_error": (ctypes.c_char_p, []),
    "pnsfm_build_target": (ctypes.c_char_p, []),
    "pnsfm_conv2d_packed_elems_fwd": (_sz, [_i, _i, _i]),
    "pnsfm_conv2d_packed_elems_bwd": (_sz, [_i, _i, _i]),
    "pnsfm_conv2d_pack_weights": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "pnsfm_conv2d_pack_item_bytes": (_sz, []),
    "pnsfm_conv2d_pack_item_fill": (_i, [_p, _p, _p, _p, _i, _i, _i, _i]),
    "pnsfm_conv2d_pack_table": (_i, [_p, _i, _i, _p]),
    "pnsfm_conv2d_forward": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "pnsfm_conv2d_backward_data": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "pnsfm_conv2d_backward_data_add": (_i, [_p, _p, _p, _p, ctypes.c_longlong, _i, _i, _i, _i, _i, _i, _p]),
    "pnsfm_conv2d_backward_weight": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "pnsfm_conv2d_forward_cat": (_i, [_p, _i, _p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "pnsfm_conv2d_backward_weight_cat": (_i, [_p, _i, _p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "pnsfm_conv2d_cat_wgrad_supported": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "pnsfm_conv2d_forward_strided": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "pnsfm_conv2d_backward_weight_strided": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "pnsfm_groupnorm_ws_doubles": (_sz, [_i, _i, _i]),
    "pnsfm_groupnorm_act_forward": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p]),
    "pnsfm_groupnorm_act_backward": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "pnsfm_set_gn_fused": (_i, [_i]),
    "pnsfm_space_to_depth": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "pnsfm_space_to_depth_strided": (_i, [_p, _p, _i, _i, _i, _i, _sz, _p]),
    "pnsfm_depth_to_space": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "pnsfm_conv3d_1to8_forward": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "pnsfm_conv3d_1to8_backward_data": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "pnsfm_conv3d_1to8_backward_weight": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "pnsfm_conv3d_forward": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "pnsfm_conv3d_backward_data": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "pnsfm_conv3d_backward_weight": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "pnsfm_invdepth_act_forward": (_i, [_p, _p, _sz, _f, _p]),
    "pnsfm_invdepth_act_backward": (_i, [_p, _p, _p, _sz, _f, _p]),
    "pnsfm_pose_vec2mat_forward": (_i, [_p, _p, _i, _p]),
    "pnsfm_upsample_nearest_forward": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "pnsfm_upsample_nearest_backward": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "pnsfm_loss_combine_forward": (_i, [_p, _i, _p, _i, _f, _p, _p]),
    "pnsfm_loss_combine_backward": (_i, [_p, _i, _i, _f, _p, _p]),
    "pnsfm_pack_bias_eff_forward": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "pnsfm_pack_bias_eff_backward": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "pnsfm_pose_vec2mat_backward": (_i, [_p, _p, _p, _i, _p]),
    "pnsfm_supervised_loss_forward": (_i, [_p, _p, _p, _p, _sz, _i, _i, _p]),
    "pnsfm_supervised_loss_backward": (_i, [_p, _p, _p, _p, _p, _sz, _i, _i, _p]),
    "pnsfm_invdepth_conv_forward": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "pnsfm_invdepth_conv_backward": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "pnsfm_view_synthesis_forward": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "pnsfm_view_synthesis_backward": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "pnsfm_view_synthesis_forward_pad": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "pnsfm_view_synthesis_backward_pad": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "pnsfm_photometric_forward": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _f, _i, _i, _p]),
    "pnsfm_photometric_backward": (_i, [_p, _p, _p, _p, _f, _i, _i, _i, _i, _f, _f, _f, _i, _i, _p]),
    "pnsfm_photometric_forward_clip": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _f, _i, _i, _f, _p, _p, _p]),
    "pnsfm_photometric_backward_clip": (_i, [_p, _p, _p, _p, _f, _i, _i, _i, _i, _f, _f, _f, _i, _i, _p]),
    "pnsfm_photometric_forward_mean": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _f, _i, _i, _p]),
    "pnsfm_photometric_backward_dev": (_i, [_p, _p, _p, _p, _f, _p, _i, _i, _i, _i, _f, _f, _f, _i, _i, _i, _p]),
    "pnsfm_smoothness_norm_forward": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "pnsfm_smoothness_norm_backward": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "pnsfm_region_ops": (_i, [_p, _i, _p]),
    "pnsfm_photometric_l1_forward": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p]),
    "pnsfm_photometric_l1_backward": (_i, [_p, _p, _p, _p, _f, _p, _i, _i, _i, _i, _i, _i, _p]),
    "pnsfm_smoothness_forward": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "pnsfm_smoothness_backward": (_i, [_p, _p, _p, _f, _f, _i, _i, _i, _p]),
    "pnsfm_adam_step": (_i, [_p, _p, _p, _p, _sz, _f, _f, _f, _f, _f, _f, _i, _p]),
    "pnsfm_adam_flat_step": (_i, [_p, _p, _p, _p, _sz, _p, _p]),
    "pnsfm_adam_flat_update": (_i, [_p, _p, _p, _p, _sz, _p, _i, _p]),
    "pnsfm_stream_wait_stream": (_i, [_p, _p]),
    "pnsfm_adam_pack_item_bytes": (_sz, []),
    "pnsfm_adam_pack_item_fill": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i]),
    "pnsfm_adam_pack_table": (_i, [_p, _i, _i, _p]),
    "pnsfm_adam_seg_bytes": (_sz, []),
    "pnsfm_adam_seg_fill": (_i, [_p, _p, _p, _p, _p, _p, _sz, _i]),
    "pnsfm_adam_segments": (_i, [_p, _i, _i, _p]),
    "pnsfm_resample8": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "pnsfm_jitter_totensor": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "pnsfm_nrs_project_forward": (_i, [_p, _p, _p, _p, _i, _i, _f, _p]),
    "pnsfm_nrs_project_backward": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "pnsfm_sparse_compact_ws_ints": (_sz, [_i]),
    "pnsfm_sparse_compact": (_i, [_p, _i, _p, _p, _i, _p, _p, _p]),
    "pnsfm_sparse_pool_cells": (_i, [_p, _i, _i, _i, _p, _p]),
    "pnsfm_sparse_neighbors": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "pnsfm_sparse_conv": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "pnsfm_sparse_conv_backward_weight": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "pnsfm_sparse_maxpool_forward": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "pnsfm_sparse_maxpool_backward": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "pnsfm_sparse_densify": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "pnsfm_sparse_gather": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "pnsfm_set_autotune": (_i, [_i]),
    "pnsfm_set_conv_variant": (_i, [_i]),
    "pnsfm_tune_shipped_entries": (_i, []),
    "pnsfm_set_conv_math": (_i, [_i]),
    "pnsfm_get_conv_math": (_i, []),
    "pnsfm_set_wgrad_variant": (_i, [_i]),
    "pnsfm_tune_set": (_i, [ctypes.POINTER(ctypes.c_int), _i, _i]),
    "pnsfm_conv2d_last_config": (_i, [ctypes.POINTER(ctypes.c_int)]),
    "pnsfm_calib_mfma": (_i, [_p, _i, _i, _p]),
    "pnsfm_calib_copy": (_i, [_p, _p, _sz, _p]),
    "pnsfm_prof_enable": (_i, [_i]),
    "pnsfm_prof_reset": (_i, []),
    "pnsfm_prof_collect": (_i, [_i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                ctypes.POINTER(ctypes.c_longlong)]),
    "pnsfm_prof_dump": (_i, [ctypes.c_char_p]),
}


def bind(lib):
    """Attach restype/argtypes for every symbol of include/pnsfm.h; raises if one is missing."""
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


# The product path requires device tensors.  (tests/emu flips this to run the same wrappers on the
# host-emulated build; nothing under packnet-sfm_amd/ ever does.)
REQUIRE_CUDA = True
_LIB = None


def get():
    """The loaded gfx950 library; raises ImportError if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libpnsfm_hip.so not found at %s -- build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the HIP ops." % LIB_PATH)
        lib = bind(ctypes.CDLL(LIB_PATH))
        target = lib.pnsfm_build_target().decode()
        if target != "gfx950":
            raise ImportError("libpnsfm_hip.so was built for %r, expected 'gfx950'" % target)
        _LIB = lib
    return _LIB


class HipError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise HipError("%s failed (rc=%d): %s" % (what, rc, get().pnsfm_last_error().decode()))
