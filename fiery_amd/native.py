"""ctypes binding of libfiery_hip.so (include/fiery_hip.h).

PyTorch is plumbing here: tensors supply device memory and the current HIP stream; every kernel is
the hand-written gfx950 code in `fiery_amd/csrc`.  There is no fallback: if the shared library is not
built, `get()` raises - a silently different code path would void every parity claim.
"""
import ctypes as C
import os
import subprocess
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libfiery_hip.so')
ABI_VERSION = 25

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_uint8_p = C.POINTER(C.c_uint8)

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SWISH = 0, 1, 2, 3
PRECISION_F32, PRECISION_BF16, PRECISION_F32_SPLIT = 0, 1, 2
EPI_PLAIN, EPI_GRU_GATES, EPI_GRU_OUT, EPI_HEADS = 0, 1, 2, 3
CONV_FORM_TILE, CONV_FORM_STREAM_K, CONV_FORM_WINOGRAD, CONV_FORM_WINOGRAD_SPLIT = 0, 1, 2, 3
WINOGRAD_SPLIT_TERMS = 3                    # fiery_conv_desc.winograd: the split image is in weights_winograd
POOL_DETERMINISTIC = 1
POOL_WORKSPACE_CLEAN = 2        # the workspace is a zero-filled allocation or was left by a successful pooling call
POOL_NO_RANKS = 4               # inference: the call leaves no voxel ranks in its workspace (nobody runs backward)
WARP_FUSED_GRID_PRODUCT = 1


class BevGrid(C.Structure):
    _fields_ = [('origin', C.c_float * 3), ('resolution', C.c_float * 3), ('dim', C.c_int32 * 3)]


class ConvSrc(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('ld', C.c_int32), ('units', C.c_int32),
                ('batch_stride', C.c_int64), ('time_stride', C.c_int64)]


class Nhwc(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('ld', C.c_int32), ('img_stride', C.c_int64)]


MAX_HEAD_OUTPUTS = 8


class ConvHeads(C.Structure):
    _fields_ = [('w', C.c_void_p), ('bias', C.c_void_p), ('n_out', C.c_int32),
                ('group', C.c_int32 * MAX_HEAD_OUTPUTS), ('sigmoid', C.c_int32 * MAX_HEAD_OUTPUTS),
                ('out', C.c_void_p * MAX_HEAD_OUTPUTS), ('img_stride', C.c_int64 * MAX_HEAD_OUTPUTS)]


class ConvDesc(C.Structure):
    _fields_ = [
        ('src', ConvSrc * 2),
        ('Hin', C.c_int32), ('Win', C.c_int32), ('Hout', C.c_int32), ('Wout', C.c_int32),
        ('n_img_out', C.c_int32), ('T_out', C.c_int32), ('t_out0', C.c_int32), ('t_in_add', C.c_int32),
        ('kT', C.c_int32), ('kH', C.c_int32), ('kW', C.c_int32), ('stride', C.c_int32),
        ('padH', C.c_int32), ('padW', C.c_int32),
        ('weights', C.c_void_p), ('cout_pad', C.c_int32),
        ('scale', C.c_void_p), ('shift', C.c_void_p), ('img_bias', C.c_void_p),
        ('act', C.c_int32), ('epi', C.c_int32), ('res_before_act', C.c_int32),
        ('res', Nhwc), ('out', Nhwc), ('cout_store', C.c_int32), ('out2', Nhwc),
        ('aux0', Nhwc), ('aux1', Nhwc),
        ('weights2', C.c_void_p), ('scale2', C.c_void_p), ('shift2', C.c_void_p), ('act2', C.c_int32),
        ('weights3', C.c_void_p), ('scale3', C.c_void_p), ('shift3', C.c_void_p), ('act3', C.c_int32), ('out3', Nhwc),
        ('tile_m', C.c_int32), ('img_bias_border', C.c_int32), ('heads', ConvHeads),
        ('weights_bf16', C.c_void_p), ('precision', C.c_int32),
        ('weights_winograd', C.c_void_p), ('winograd', C.c_int32),
        ('stream_k', C.c_int32), ('sk_workspace', C.c_void_p), ('sk_workspace_bytes', C.c_int64),
        ('sk_counters', C.c_void_p), ('sk_counters_len', C.c_int32),
        ('weights2_split', C.c_void_p), ('weights3_split', C.c_void_p),
    ]


class NativeError(RuntimeError):
    pass


def _ptr(t):
    """Device address of a tensor, of a pixel-major view object exposing `.ptr` (ops.Buf), or a raw int."""
    if t is None:
        return None
    if isinstance(t, int):
        return C.c_void_p(t)
    if isinstance(t, torch.Tensor):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ptr)


def _stream_of(*tensors):
    for t in tensors:
        t = getattr(t, 'tensor', t)
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return C.c_void_p(0)


_SIGNATURES = {
    'fiery_abi_version': (C.c_int, []),
    'fiery_last_error': (C.c_char_p, []),
    'fiery_camera_matrices': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'fiery_camera_matrices_cached': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                               C.c_void_p, C.c_void_p]),
    'fiery_lift_geometry': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'fiery_voxel_index': (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(BevGrid), C.c_void_p, C.c_void_p, C.c_void_p]),
    'fiery_voxel_pool_workspace_bytes': (C.c_size_t, [C.c_int] * 7 + [C.c_uint32]),
    'fiery_voxel_pool_occupied_offset': (C.c_size_t, [C.c_int] * 7 + [C.c_uint32]),
    'fiery_voxel_pool_fwd': (C.c_int, [C.c_void_p, c_int64_p, C.c_void_p] + [C.c_int] * 6 +
                             [C.POINTER(BevGrid), C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_void_p]),
    'fiery_lift_splat_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 +
                             [C.POINTER(BevGrid), C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_void_p]),
    'fiery_voxel_pool_bwd': (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, c_int64_p, C.c_void_p]),
    'fiery_lift_splat_bwd_workspace_bytes': (C.c_size_t, [C.c_int] * 3),
    'fiery_lift_splat_bwd': (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'fiery_depth_softmax': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'fiery_depth_softmax_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'fiery_warp_params': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fiery_warp_params_reverse': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    'fiery_bev_warp_nearest_nchw': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'fiery_bev_warp_nchw_to_nhwc': (C.c_int, [C.c_void_p, C.c_void_p, c_uint8_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p]),
    'fiery_bev_warp_bwd_nhwc_to_nchw': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, c_uint8_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.c_void_p, C.c_int, C.c_void_p]),
    'fiery_maxpool2x2_bwd_nhwc': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_int, C.c_void_p]),
    'fiery_conv_packed_floats': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'fiery_conv_pack_weights': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_int32_p, C.c_int, C.c_void_p, C.c_void_p]),
    'fiery_conv_fwd': (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    'fiery_conv_pack_weights_bf16': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_int32_p, C.c_int, C.c_void_p, C.c_void_p]),
    'fiery_conv_pack_weights_split': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_int32_p, C.c_int, C.c_void_p, C.c_void_p]),
    'fiery_conv_precision_used': (C.c_int, [C.POINTER(ConvDesc)]),
    'fiery_conv_form_used': (C.c_int, [C.POINTER(ConvDesc)]),
    'fiery_conv_winograd_packed_floats': (C.c_size_t, [C.c_int, C.c_int]),
    'fiery_conv_pack_weights_winograd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_void_p, C.c_void_p]),
    'fiery_conv_winograd_split_packed_floats': (C.c_size_t, [C.c_int, C.c_int]),
    'fiery_conv_pack_weights_winograd_split': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_void_p, C.c_void_p]),
    'fiery_conv_stream_k_plan': (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'fiery_conv_wgrad': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int64] + [C.c_int] * 11 +
                         [C.c_void_p, C.c_void_p]),
    'fiery_conv_wgrad_prec': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int64] + [C.c_int] * 12 +
                              [C.c_void_p, C.c_void_p]),
    'fiery_heads_1x1_nchw': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       c_int32_p, c_uint8_p, C.c_void_p, C.c_void_p]),
    'fiery_spatial_mean': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    'fiery_rowwise_dense': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int,
                                      C.c_void_p]),
    'fiery_sequential_window_mean': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'fiery_latent_sample': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                      C.c_void_p]),
    'fiery_maxpool2x2_nhwc': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'fiery_upsample2x_add_nhwc': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'fiery_bn_workspace_floats': (C.c_int64, [C.c_int]),
    'fiery_bn_train_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    'fiery_bn_train_bwd': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    'fiery_bn_train_stats': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fiery_bn_apply': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_int, C.c_int, C.c_void_p]),
    'fiery_bn_train_bwd_sums': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fiery_bn_train_bwd_dx': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'fiery_gru_reset_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_void_p]),
    'fiery_gru_reset_bwd': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_void_p]),
    'fiery_gru_out_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_void_p]),
    'fiery_gru_out_bwd': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'fiery_instance_labels_workspace_ints': (C.c_int64, [C.c_int, C.c_int]),
    'fiery_instance_labels': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fiery_image_resize_crop_normalise': (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] +
                                          [C.c_int] * 6 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fiery_upsample2x_bwd_nhwc': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'fiery_depthwise_conv_nhwc': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int] + [C.c_int] * 6 +
                                  [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'fiery_depthwise_conv_wgrad_nhwc': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int] + [C.c_int] * 6 +
                                        [C.c_void_p, C.c_int, C.c_void_p]),
    'fiery_instance_segmentation': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fiery_se_gate': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                      C.c_int, C.c_void_p]),
    'fiery_se_gate_nhwc': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'fiery_scale_channels_nhwc': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'fiery_broadcast_nhwc': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    'fiery_nchw_to_nhwc': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    'fiery_nhwc_to_nchw': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def require_usable_gpu_process(what):
    """The input-pipeline entry points keep the signatures of functions the reference calls inside `Dataset.__getitem__`,
    i.e. in DataLoader worker processes.  With the default `fork` start method such a worker inherits an initialised HIP
    context it cannot use; PyTorch reports that as 'Cannot re-initialize CUDA in forked subprocess' from somewhere deep in
    the first tensor move.  Say it here, with the remedies."""
    # (torch.cuda.is_initialized() is False in a forked child BY DEFINITION - it is `_initialized and not _is_in_bad_fork()` -
    # so the fork test stands alone)
    if getattr(torch.cuda, '_is_in_bad_fork', lambda: False)():
        raise RuntimeError(
            f'{what} launches HIP kernels, but this process was forked from one that had already initialised the GPU '
            f'(a DataLoader worker with the fork start method).  Create the DataLoader with '
            f"multiprocessing_context='spawn', or keep num_workers=0 for these labels, or call the function on the collated "
            f'batch in the training process.')

_WARP_FLAGS = {}


def warp_flags_of_this_host(h, w):
    """`flags` for the resampling kernels such that they round like THIS host's ATen: affine_grid multiplies its base grid
    with theta^T through MKL's sgemm, which fuses the three-term product on Intel CPUs and rounds every operation on AMD
    ones (include/fiery_hip.h, FIERY_WARP_FUSED_GRID_PRODUCT).  The reference's CPU path is whatever the local ATen does,
    so the local ATen is asked once per map size: one affine_grid of that size on a fixed transform, compared with the
    two candidate evaluations.  `FIERY_WARP_GRID_PRODUCT=fused|plain` overrides (e.g. to reproduce another host's bits)."""
    forced = os.environ.get('FIERY_WARP_GRID_PRODUCT')
    if forced:
        return WARP_FUSED_GRID_PRODUCT if forced == 'fused' else 0
    key = (int(h), int(w))
    if key not in _WARP_FLAGS:
        import numpy as np
        theta = torch.tensor([[[0.99981, -0.01937, 0.01234], [0.01937, 0.99981, -0.05678]]], dtype=torch.float32)
        grid = torch.nn.functional.affine_grid(theta, (1, 1, key[0], key[1]), align_corners=False)[0, ..., 0].numpy()
        lin = lambda n: (torch.linspace(-1, 1, n) * (n - 1) / n).numpy() if n > 1 else np.zeros(1, np.float32)
        xs, ys = np.meshgrid(lin(key[1]), lin(key[0]))
        t0, t1, t2 = (np.float32(v) for v in theta[0, 0].tolist())
        first = (xs * t0).astype(np.float32)
        plain = ((first + (ys * t1).astype(np.float32)).astype(np.float32) + t2).astype(np.float32)
        fused = ((ys.astype(np.float64) * np.float64(t1) + first.astype(np.float64)).astype(np.float32) + t2).astype(np.float32)
        if np.array_equal(grid, fused) and not np.array_equal(grid, plain):
            _WARP_FLAGS[key] = WARP_FUSED_GRID_PRODUCT
        elif np.array_equal(grid, plain):
            _WARP_FLAGS[key] = 0
        else:            # an ATen / BLAS build that does neither: keep the fused form (documented, 1 ulp of a position apart)
            _WARP_FLAGS[key] = WARP_FUSED_GRID_PRODUCT
    return _WARP_FLAGS[key]


# `CALL_SINK = []`: every launching entry point of the library is bracketed by HIP events on the current stream and filed as
# (entry point, start, end) - bench.py's per-kernel table of the pass from images (None: the calls go straight through)
CALL_SINK = None
_HOST_ONLY = ('version', 'last_error', 'workspace_', 'occupied_offset', 'packed_floats', '_plan', '_used')


class _EntryPoints:
    """The loaded library's functions; with `CALL_SINK` set, the launching ones are timed."""

    def __init__(self, dll):
        self._dll = dll

    def __getattr__(self, name):
        fn = getattr(self._dll, name)
        if not name.startswith('fiery_') or any(t in name for t in _HOST_ONLY):
            self.__dict__[name] = fn
            return fn

        def call(*args):
            sink = CALL_SINK
            if sink is None or not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
                return fn(*args)
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            rc = fn(*args)
            end.record()
            sink.append((name, start, end))
            return rc
        self.__dict__[name] = call
        return call


class Lib:
    """One loaded copy of the C-ABI library."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise NativeError(
                f'{path} is missing: build the HIP extension first '
                f'(`python -c "import __graft_entry__ as g; g.build()"` or `python -m fiery_amd.build`). '
                f'There is deliberately no fallback path.')
        self.path = path
        self.dll = C.CDLL(path)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(self.dll, name)          # AttributeError here = symbol missing from the build
            fn.restype, fn.argtypes = restype, argtypes
        self.dll = _EntryPoints(self.dll)
        got = self.dll.fiery_abi_version()
        if got != ABI_VERSION:
            raise NativeError(f'{path}: ABI version {got}, bindings expect {ABI_VERSION}; rebuild')

    def check(self, rc):
        if rc != 0:
            raise NativeError(f'libfiery_hip: error {rc}: {self.dll.fiery_last_error().decode()}')

    # -- lift -------------------------------------------------------------------------------------
    def camera_matrices(self, intrinsics, extrinsics):
        n = intrinsics.numel() // 9
        cam = torch.empty(n, 12, dtype=torch.float32, device=intrinsics.device)
        self.check(self.dll.fiery_camera_matrices(_ptr(intrinsics), _ptr(extrinsics), n, _ptr(cam), _stream_of(cam)))
        return cam

    def camera_matrices_cached(self, intrinsics, extrinsics, table, slots, misses, miss_capacity):
        """Camera matrices through a calibration table (fiery_amd/calibration.py owns `table` and `misses`)."""
        n = intrinsics.numel() // 9
        assert extrinsics.numel() == n * 16 and table.numel() == slots * 36 and table.dtype == torch.int32
        cam = torch.empty(n, 12, dtype=torch.float32, device=intrinsics.device)
        self.check(self.dll.fiery_camera_matrices_cached(_ptr(intrinsics), _ptr(extrinsics), n, _ptr(table), slots, _ptr(misses),
                                                         miss_capacity, _ptr(cam), _stream_of(cam)))
        return cam

    def lift_geometry(self, frustum, cam):
        d, h, w, _ = frustum.shape
        n = cam.shape[0]
        out = torch.empty(n, d, h, w, 3, dtype=torch.float32, device=cam.device)
        self.check(self.dll.fiery_lift_geometry(_ptr(frustum), _ptr(cam), n, d, h, w, _ptr(out), _stream_of(out)))
        return out

    # -- splat ------------------------------------------------------------------------------------
    def voxel_index(self, geometry, grid, want_idx=True):
        n = geometry.numel() // 3
        rank = torch.empty(n, dtype=torch.int32, device=geometry.device)
        idx = torch.empty(n, 3, dtype=torch.int32, device=geometry.device) if want_idx else None
        self.check(self.dll.fiery_voxel_index(_ptr(geometry), n, C.byref(grid), _ptr(rank), _ptr(idx), _stream_of(rank)))
        return rank, idx

    def pool_workspace(self, frames, n_cam, d, h, w, device, grid, tile_voxels=0, flags=0, zeroed=False):
        """`zeroed`: a zero-filled allocation - what a caller that keeps the workspace across calls needs in order to pass
        POOL_WORKSPACE_CLEAN (every successful call leaves the workspace clean again)."""
        nbytes = self.dll.fiery_voxel_pool_workspace_bytes(frames, n_cam, d, h, w, grid.dim[0] * grid.dim[1],
                                                           tile_voxels, flags)
        if nbytes == 0:
            raise NativeError('libfiery_hip: unusable pooling problem size (see fiery_voxel_pool_workspace_bytes)')
        return (torch.zeros if zeroed else torch.empty)((nbytes + 3) // 4, dtype=torch.int32, device=device)

    def pool_occupied(self, workspace, frames, n_cam, d, h, w, grid, tile_voxels=0, flags=0):
        """Occupied voxels per frame, as the last compact-plane pooling call on `workspace` counted them (int32 view of the
        workspace: read it after the stream has finished)."""
        off = self.dll.fiery_voxel_pool_occupied_offset(frames, n_cam, d, h, w, grid.dim[0] * grid.dim[1], tile_voxels, flags)
        if off == 0:
            raise NativeError('libfiery_hip: unusable pooling problem size (see fiery_voxel_pool_occupied_offset)')
        return workspace[off // 4:off // 4 + frames]

    def voxel_pool(self, x, strides, geometry, frames, n_cam, d, h, w, c, grid, out=None, workspace=None,
                   tile_voxels=0, flags=0):
        if workspace is None:
            workspace = self.pool_workspace(frames, n_cam, d, h, w, x.device, grid, tile_voxels, flags)
        if out is None:
            out = torch.empty(frames, c, grid.dim[0], grid.dim[1], dtype=torch.float32, device=x.device)
        xs = (C.c_int64 * 6)(*strides)
        self.check(self.dll.fiery_voxel_pool_fwd(
            _ptr(x), xs, _ptr(geometry), frames, n_cam, d, h, w, c, C.byref(grid), _ptr(out),
            _ptr(workspace), workspace.numel() * 4, tile_voxels, flags, _stream_of(out)))
        return out

    def lift_splat(self, depth_prob, features, geometry, frames, n_cam, d, h, w, c, grid, out=None, workspace=None,
                   tile_voxels=0, flags=0):
        if workspace is None:
            workspace = self.pool_workspace(frames, n_cam, d, h, w, features.device, grid, tile_voxels, flags)
        if out is None:
            out = torch.empty(frames, c, grid.dim[0], grid.dim[1], dtype=torch.float32, device=features.device)
        self.check(self.dll.fiery_lift_splat_fwd(
            _ptr(depth_prob), _ptr(features), _ptr(geometry), frames, n_cam, d, h, w, c, C.byref(grid), _ptr(out),
            _ptr(workspace), workspace.numel() * 4, tile_voxels, flags, _stream_of(out)))
        return out

    def voxel_pool_bwd(self, grad_out, rank, frames, n_cam, d, h, w, c, grad_x):
        """grad_x: a (frames, n_cam, d, h, w, c) tensor of any strides, fully overwritten."""
        n_vox = grad_out[0, 0].numel()
        gs = (C.c_int64 * 6)(*grad_x.stride())
        self.check(self.dll.fiery_voxel_pool_bwd(_ptr(grad_out), _ptr(rank), frames, n_cam, d, h, w, c, n_vox,
                                                 _ptr(grad_x), gs, _stream_of(grad_x)))
        return grad_x

    def lift_splat_bwd(self, grad_out, rank, depth_prob, features, frames, n_cam, d, h, w, c, want_depth=True,
                       want_features=True):
        n_vox = grad_out[0, 0].numel()
        gd = torch.empty_like(depth_prob) if want_depth else None
        gf = torch.empty_like(features) if want_features else None
        ws = torch.empty(self.dll.fiery_lift_splat_bwd_workspace_bytes(frames, c, n_vox) // 4, dtype=torch.float32,
                         device=grad_out.device)
        self.check(self.dll.fiery_lift_splat_bwd(_ptr(grad_out), _ptr(rank), _ptr(depth_prob), _ptr(features), frames,
                                                 n_cam, d, h, w, c, n_vox, _ptr(gd), _ptr(gf), _ptr(ws), ws.numel() * 4,
                                                 _stream_of(grad_out)))
        return gd, gf

    def depth_softmax(self, logits):
        n, d = logits.shape[:2]
        hw = logits[0, 0].numel()
        out = torch.empty_like(logits)
        self.check(self.dll.fiery_depth_softmax(_ptr(logits), n, d, hw, _ptr(out), _stream_of(out)))
        return out

    def depth_softmax_bwd(self, prob, grad_prob):
        n, d = prob.shape[:2]
        hw = prob[0, 0].numel()
        out = torch.empty_like(prob)
        self.check(self.dll.fiery_depth_softmax_bwd(_ptr(prob), _ptr(grad_prob), n, d, hw, _ptr(out), _stream_of(out)))
        return out

    # -- warp -------------------------------------------------------------------------------------
    def warp_params(self, future_egomotion, extent, theta=None, ego_shifted=None):
        """-> theta (B, S, 6); `ego_shifted` (B, S, 6), when given, receives the temporal model's ego-pose input."""
        b, s, _ = future_egomotion.shape
        if theta is None:
            theta = torch.empty(b, s, 6, dtype=torch.float32, device=future_egomotion.device)
        self.check(self.dll.fiery_warp_params(_ptr(future_egomotion), b, s, float(extent[0]), float(extent[1]),
                                              _ptr(theta), _ptr(ego_shifted), _stream_of(theta)))
        return theta

    def warp_params_reverse(self, future_egomotion, extent):
        b, s, _ = future_egomotion.shape
        theta = torch.empty(b, s, 6, dtype=torch.float32, device=future_egomotion.device)
        self.check(self.dll.fiery_warp_params_reverse(_ptr(future_egomotion), b, s, float(extent[0]), float(extent[1]),
                                                      _ptr(theta), _stream_of(theta)))
        return theta

    def bev_warp_nearest(self, x, theta, flags=None):
        """x (n, C, H, W) f32 contiguous, theta (n, 6) -> the nearest-neighbour resampling (n, C, H, W)."""
        n, c, h, w = x.shape
        out = torch.empty_like(x)
        flags = warp_flags_of_this_host(h, w) if flags is None else flags
        self.check(self.dll.fiery_bev_warp_nearest_nchw(_ptr(x), _ptr(theta), n, c, h, w, _ptr(out), flags, _stream_of(out)))
        return out

    def bev_warp_nchw_to_nhwc(self, x, theta, identity, out, out_ld, out_img_stride, flags=None):
        n, c, h, w = x.shape
        ident = (C.c_uint8 * n)(*[1 if v else 0 for v in identity]) if identity is not None else None
        flags = warp_flags_of_this_host(h, w) if flags is None else flags
        self.check(self.dll.fiery_bev_warp_nchw_to_nhwc(_ptr(x), _ptr(theta), ident, n, c, h, w, _ptr(out), out_ld,
                                                        out_img_stride, flags, _stream_of(out)))

    def bev_warp_bwd(self, grad_out, g_ld, g_img_stride, theta, identity, n, c, h, w, flags=None):
        """Adjoint of `bev_warp_nchw_to_nhwc` in its input: pixel-major output gradient -> (n, c, h, w)."""
        ident = (C.c_uint8 * n)(*[1 if v else 0 for v in identity]) if identity is not None else None
        flags = warp_flags_of_this_host(h, w) if flags is None else flags
        gx = torch.empty(n, c, h, w, dtype=torch.float32, device=grad_out.device)
        self.check(self.dll.fiery_bev_warp_bwd_nhwc_to_nchw(_ptr(grad_out), g_ld, g_img_stride, _ptr(theta), ident, n, c, h, w,
                                                            _ptr(gx), flags, _stream_of(gx)))
        return gx

    def maxpool2x2_bwd(self, x, in_ld, grad_out, g_ld, n_img, h, w, c):
        """x dense pixel-major (n_img, h, w, in_ld), grad_out (n_img, ceil(h/2), ceil(w/2), g_ld) -> (n_img, h, w, in_ld)."""
        gx = torch.zeros(n_img, h, w, in_ld, dtype=torch.float32, device=grad_out.device)
        self.check(self.dll.fiery_maxpool2x2_bwd_nhwc(_ptr(x), in_ld, 0, _ptr(grad_out), g_ld, n_img, h, w, c, _ptr(gx), in_ld,
                                                      _stream_of(gx)))
        return gx

    # -- conv -------------------------------------------------------------------------------------
    def conv_pack_weights(self, w, cout, cin_total, taps, chan_map, cin_units):
        n = self.dll.fiery_conv_packed_floats(cout, cin_units, taps)
        packed = torch.empty(n, dtype=torch.float32, device=w.device)
        cmap = (C.c_int32 * cin_total)(*chan_map)
        self.check(self.dll.fiery_conv_pack_weights(_ptr(w), cout, cin_total, taps, cmap, cin_units, _ptr(packed),
                                                    _stream_of(packed)))
        return packed

    def conv_pack_weights_bf16(self, w, cout, cin_total, taps, chan_map, cin_units):
        n = self.dll.fiery_conv_packed_floats(cout, cin_units, taps)
        packed = torch.empty(n, dtype=torch.bfloat16, device=w.device)
        cmap = (C.c_int32 * cin_total)(*chan_map)
        self.check(self.dll.fiery_conv_pack_weights_bf16(_ptr(w), cout, cin_total, taps, cmap, cin_units, _ptr(packed),
                                                         _stream_of(packed)))
        return packed

    def conv_pack_weights_split(self, w, cout, cin_total, taps, chan_map, cin_units):
        """The split image (PRECISION_F32_SPLIT): every weight as three bf16 terms that add up to it exactly."""
        n = self.dll.fiery_conv_packed_floats(cout, cin_units, taps)
        packed = torch.empty(3 * n, dtype=torch.bfloat16, device=w.device)
        cmap = (C.c_int32 * cin_total)(*chan_map)
        self.check(self.dll.fiery_conv_pack_weights_split(_ptr(w), cout, cin_total, taps, cmap, cin_units, _ptr(packed),
                                                          _stream_of(packed)))
        return packed

    def conv_pack_weights_winograd(self, w, cout, cin_total, chan_map, cin_units):
        """w (cout, cin_total, 9) f32 -> the Winograd F(2x2, 3x3) image of the weights (G g G^T in fp64, rounded once)."""
        n = self.dll.fiery_conv_winograd_packed_floats(cout, cin_units)
        packed = torch.empty(n, dtype=torch.float32, device=w.device)
        cmap = (C.c_int32 * cin_total)(*chan_map)
        self.check(self.dll.fiery_conv_pack_weights_winograd(_ptr(w), cout, cin_total, cmap, cin_units, _ptr(packed), _stream_of(packed)))
        return packed

    def conv_pack_weights_winograd_split(self, w, cout, cin_total, chan_map, cin_units):
        """w (cout, cin_total, 9) f32 -> the split Winograd image: every transformed weight as three bf16 terms."""
        n = self.dll.fiery_conv_winograd_split_packed_floats(cout, cin_units)
        packed = torch.empty(n, dtype=torch.float32, device=w.device)
        cmap = (C.c_int32 * cin_total)(*chan_map)
        self.check(self.dll.fiery_conv_pack_weights_winograd_split(_ptr(w), cout, cin_total, cmap, cin_units, _ptr(packed), _stream_of(packed)))
        return packed

    def conv_precision_used(self, desc):
        rc = self.dll.fiery_conv_precision_used(C.byref(desc))
        if rc < 0:
            self.check(rc)
        return rc

    def conv_form_used(self, desc):
        """CONV_FORM_TILE / _STREAM_K / _WINOGRAD: the form `conv_fwd` runs this descriptor in (requests the form does not
        cover fall back to the tile form)."""
        rc = self.dll.fiery_conv_form_used(C.byref(desc))
        if rc < 0:
            self.check(rc)
        return rc

    def conv_stream_k_plan(self, desc):
        """(workspace bytes, counters, workgroups) of the stream-K form of this launch; workgroups = 0: not covered."""
        nbytes, n_cnt, n_wg = C.c_int64(0), C.c_int32(0), C.c_int32(0)
        self.check(self.dll.fiery_conv_stream_k_plan(C.byref(desc), C.byref(nbytes), C.byref(n_cnt), C.byref(n_wg)))
        return nbytes.value, n_cnt.value, n_wg.value

    ZERO_CHUNK_FLOATS = 4 << 20

    def zeros_f32(self, shape, device):
        """A zeroed fp32 tensor for a kernel that accumulates into its output (the weight gradients' split-K atomics): a slice of
        a 16 MB chunk zeroed in ONE fill when it was allocated, instead of a fill launch per tensor (150 per training step).
        Chunks are never re-zeroed or handed out twice - the views keep theirs alive, the caching allocator takes it back after
        the last one - so a view held by autograd (a shared weight's first contribution, a retained graph) stays valid.  One
        chunk per (device, stream): the fill is ordered with its users.  Inside a stream capture: a plain `torch.zeros`
        (a replay has to zero again)."""
        n = 1
        for v in shape:
            n *= int(v)
        step = (n + 63) // 64 * 64                                                          # 256-byte aligned slices
        if device.type != 'cuda' or step > self.ZERO_CHUNK_FLOATS or torch.cuda.is_current_stream_capturing():
            return torch.zeros(shape, dtype=torch.float32, device=device)
        if not hasattr(self, '_zero_chunks'):
            self._zero_chunks = {}
        key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        chunk = self._zero_chunks.get(key)
        if chunk is None or chunk[1] + step > chunk[0].numel():
            chunk = self._zero_chunks[key] = [torch.zeros(self.ZERO_CHUNK_FLOATS, dtype=torch.float32, device=device), 0]
        view = chunk[0][chunk[1]:chunk[1] + n].view(shape)
        chunk[1] += step
        return view

    def conv_wgrad(self, x, grad_out, cout, k, stride, pad, precision=PRECISION_F32):
        """x: pixel-major (n, Hin, Win, cin_pad) f32 (cin_pad a multiple of 8), grad_out: (n, Hout, Wout, >= cout) f32, both
        with contiguous rows -> dw (cout, k*k, cin_pad) f32.  precision = PRECISION_BF16: operands rounded on chip where the
        bf16 kernel applies (3 x 3 / stride 1), fp32 accumulation."""
        n, hin, win, cin_pad = x.shape
        _, hout, wout, g_ld = grad_out.shape
        # (strides of size-1 dimensions are arbitrary in torch: derive them from the shapes of the dense tensors)
        assert cin_pad % 8 == 0 and x.is_contiguous() and grad_out.is_contiguous()
        dw = self.zeros_f32((cout, k * k, cin_pad), x.device)
        self.check(self.dll.fiery_conv_wgrad_prec(_ptr(x), cin_pad, hin * win * cin_pad, cin_pad // 8, _ptr(grad_out), g_ld,
                                                  hout * wout * g_ld, cout, n, hin, win, hout, wout, k, k, stride, pad, pad, precision,
                                                  _ptr(dw), _stream_of(dw)))
        return dw

    def conv_fwd(self, desc, stream_tensor):
        self.check(self.dll.fiery_conv_fwd(C.byref(desc), _stream_of(stream_tensor)))

    def heads_1x1_nchw(self, x, in_ld, n_img, hw, c, head_c, w, bias, c_off, sigmoid, out):
        n_out = len(c_off)
        offs = (C.c_int32 * n_out)(*c_off)
        sig = (C.c_uint8 * n_out)(*[1 if v else 0 for v in sigmoid])
        self.check(self.dll.fiery_heads_1x1_nchw(_ptr(x), in_ld, n_img, hw, c, head_c, n_out, _ptr(w), _ptr(bias), offs, sig,
                                                 _ptr(out), _stream_of(out)))

    # -- helpers ----------------------------------------------------------------------------------
    def spatial_mean(self, x_ptr, in_ld, outer_stride, n_outer, inner_stride, n_inner, n_pixels, c, out, workspace):
        self.check(self.dll.fiery_spatial_mean(_ptr(x_ptr), in_ld, outer_stride, n_outer, inner_stride, n_inner, n_pixels, c,
                                               _ptr(out), _ptr(workspace), _stream_of(out)))

    def rowwise_dense(self, v, v_ld, rows, n_in, w, w_ld, w_col0, n_out, scale, shift, act, accumulate, y, y_ld,
                      w_mul=1.0, lo=float('-inf'), hi=float('inf')):
        self.check(self.dll.fiery_rowwise_dense(_ptr(v), v_ld, rows, n_in, _ptr(w), w_ld, w_col0, n_out, w_mul, _ptr(scale),
                                                _ptr(shift), act, int(accumulate), lo, hi, _ptr(y), y_ld, _stream_of(y)))

    def sequential_window_mean(self, prev, cur, rows, n, count_each, out, out_ld):
        self.check(self.dll.fiery_sequential_window_mean(_ptr(prev), _ptr(cur), rows, n, count_each, _ptr(out), out_ld,
                                                         _stream_of(out)))

    def latent_sample(self, mu, log_sigma, noise, ld, rows, n, sample, sample_ld):
        self.check(self.dll.fiery_latent_sample(_ptr(mu), _ptr(log_sigma), _ptr(noise), ld, rows, n, _ptr(sample), sample_ld,
                                                _stream_of(sample)))

    def maxpool2x2(self, x, in_ld, n_img, h, w, c, out, out_ld):
        self.check(self.dll.fiery_maxpool2x2_nhwc(_ptr(x), in_ld, getattr(x, 'img_stride', 0), n_img, h, w, c, _ptr(out), out_ld,
                                                  _stream_of(out)))

    def upsample2x_add(self, x, in_ld, n_img, h, w, c, shift, skip, skip_ld, out, out_ld):
        self.check(self.dll.fiery_upsample2x_add_nhwc(_ptr(x), in_ld, n_img, h, w, c, _ptr(shift), _ptr(skip), skip_ld,
                                                      _ptr(out), out_ld, _stream_of(out)))

    def _bn_workspace(self, c, device):
        """Partial-sum rows of the BatchNorm passes: one buffer per (width, device, STREAM) - launches on different streams
        of a device (per-sample side streams, replicas in threads) must not share partial sums."""
        stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == 'cuda' else 0
        key = (int(c), str(device), stream)
        cache = self.__dict__.setdefault('_bn_ws', {})
        if key not in cache:
            while len(cache) >= 64:                      # short-lived streams (lanes rebuilt, thread replicas) must not pile up
                cache.pop(next(iter(cache)))             # oldest first (insertion order); a live stream's entry is re-made on demand
            cache[key] = torch.empty(self.dll.fiery_bn_workspace_floats(int(c)), dtype=torch.float32, device=device)
        else:
            cache[key] = cache.pop(key)                  # most recently used last
        return cache[key]

    def bn_train_fwd(self, x, ld, n_pixels, c, gamma, beta, running_mean, running_var, batch_stats, momentum, eps, relu, c_store):
        """x: rows of `ld` floats (a tensor whose data pointer is the first row) -> (y dense [n_pixels, c_store], mean, invstd)."""
        y = torch.empty(n_pixels, c_store, dtype=torch.float32, device=x.device)
        mean = torch.empty(c, dtype=torch.float32, device=x.device)
        invstd = torch.empty(c, dtype=torch.float32, device=x.device)
        self.check(self.dll.fiery_bn_train_fwd(_ptr(x), ld, n_pixels, c, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                               int(bool(batch_stats)), float(momentum), float(eps), int(bool(relu)), _ptr(y), c_store,
                                               c_store, _ptr(mean), _ptr(invstd), _ptr(self._bn_workspace(c, x.device)), _stream_of(y)))
        return y, mean, invstd

    def bn_train_bwd(self, grad_out, g_ld, x, ld, y, y_ld, n_pixels, c, gamma, mean, invstd, batch_stats, c_store):
        gx = torch.empty(n_pixels, c_store, dtype=torch.float32, device=x.device)
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
        self.check(self.dll.fiery_bn_train_bwd(_ptr(grad_out), g_ld, _ptr(x), ld, _ptr(y), y_ld, n_pixels, c, _ptr(gamma), _ptr(mean),
                                               _ptr(invstd), int(bool(batch_stats)), _ptr(gx), c_store, c_store, _ptr(dgamma), _ptr(dbeta),
                                               _ptr(self._bn_workspace(c, x.device)), _stream_of(gx)))
        return gx, dgamma, dbeta

    def bn_train_stats(self, x, ld, n_pixels, c):
        """-> (mean, biased variance) of this process's rows."""
        mean = torch.empty(c, dtype=torch.float32, device=x.device)
        var = torch.empty(c, dtype=torch.float32, device=x.device)
        self.check(self.dll.fiery_bn_train_stats(_ptr(x), ld, n_pixels, c, _ptr(mean), _ptr(var), _ptr(self._bn_workspace(c, x.device)),
                                                 _stream_of(mean)))
        return mean, var

    def bn_apply(self, x, ld, n_pixels, c, mean, invstd, gamma, beta, relu, c_store):
        y = torch.empty(n_pixels, c_store, dtype=torch.float32, device=x.device)
        self.check(self.dll.fiery_bn_apply(_ptr(x), ld, n_pixels, c, _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), int(bool(relu)), _ptr(y),
                                           c_store, c_store, _stream_of(y)))
        return y

    def bn_train_bwd_sums(self, grad_out, g_ld, x, ld, y, y_ld, n_pixels, c, mean, invstd):
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
        self.check(self.dll.fiery_bn_train_bwd_sums(_ptr(grad_out), g_ld, _ptr(x), ld, _ptr(y), y_ld, n_pixels, c, _ptr(mean), _ptr(invstd),
                                                    _ptr(dgamma), _ptr(dbeta), _ptr(self._bn_workspace(c, x.device)), _stream_of(dgamma)))
        return dgamma, dbeta

    def bn_train_bwd_dx(self, grad_out, g_ld, x, ld, y, y_ld, n_pixels, c, gamma, mean, invstd, dgamma, dbeta, total_pixels, c_store):
        gx = torch.empty(n_pixels, c_store, dtype=torch.float32, device=x.device)
        self.check(self.dll.fiery_bn_train_bwd_dx(_ptr(grad_out), g_ld, _ptr(x), ld, _ptr(y), y_ld, n_pixels, c, _ptr(gamma), _ptr(mean),
                                                  _ptr(invstd), _ptr(dgamma), _ptr(dbeta), total_pixels, _ptr(gx), c_store, c_store,
                                                  _stream_of(gx)))
        return gx

    def _rows_out(self, n_pixels, c_store, like, count):
        return [torch.empty(n_pixels, c_store, dtype=torch.float32, device=like.device) for _ in range(count)]

    def gru_reset_fwd(self, pre, pre_ld, bias, h, h_ld, n_pixels, c, c_store):
        r, rh = self._rows_out(n_pixels, c_store, pre, 2)
        self.check(self.dll.fiery_gru_reset_fwd(_ptr(pre), pre_ld, _ptr(bias), _ptr(h), h_ld, n_pixels, c, _ptr(r), _ptr(rh), c_store, _stream_of(r)))
        return r, rh

    def gru_reset_bwd(self, d_rh, g_ld, r, h, h_ld, n_pixels, c, c_store):
        d_pre, dh = self._rows_out(n_pixels, c_store, r, 2)
        self.check(self.dll.fiery_gru_reset_bwd(_ptr(d_rh), g_ld, _ptr(r), _ptr(h), h_ld, n_pixels, c, _ptr(d_pre), _ptr(dh), c_store, _stream_of(dh)))
        return d_pre, dh

    def gru_out_fwd(self, pre, pre_ld, bias, h, h_ld, cand, cand_ld, n_pixels, c, c_store):
        u, hn = self._rows_out(n_pixels, c_store, pre, 2)
        self.check(self.dll.fiery_gru_out_fwd(_ptr(pre), pre_ld, _ptr(bias), _ptr(h), h_ld, _ptr(cand), cand_ld, n_pixels, c, _ptr(u), _ptr(hn), c_store,
                                              _stream_of(u)))
        return u, hn

    def gru_out_bwd(self, d_hn, g_ld, u, h, h_ld, cand, cand_ld, n_pixels, c, c_store):
        d_pre, dh, dcand = self._rows_out(n_pixels, c_store, u, 3)
        self.check(self.dll.fiery_gru_out_bwd(_ptr(d_hn), g_ld, _ptr(u), _ptr(h), h_ld, _ptr(cand), cand_ld, n_pixels, c, _ptr(d_pre), _ptr(dh), _ptr(dcand),
                                              c_store, _stream_of(dh)))
        return d_pre, dh, dcand

    def instance_labels(self, ids, warped_ids, n_instances, sigma, ignore_index):
        """ids, warped_ids: (T, H, W) int32 -> centerness (T, 1, H, W), offset (T, 2, H, W), flow (T, 2, H, W)."""
        t, h, w = ids.shape
        assert ids.dtype == torch.int32 and warped_ids.dtype == torch.int32 and ids.is_contiguous() and warped_ids.is_contiguous()
        f32 = dict(dtype=torch.float32, device=ids.device)
        center, offset, flow = torch.empty(t, 1, h, w, **f32), torch.empty(t, 2, h, w, **f32), torch.empty(t, 2, h, w, **f32)
        ws = torch.empty(self.dll.fiery_instance_labels_workspace_ints(t, n_instances), dtype=torch.int32, device=ids.device)
        self.check(self.dll.fiery_instance_labels(_ptr(ids), _ptr(warped_ids), t, h, w, n_instances, float(sigma), float(ignore_index),
                                                  _ptr(center), _ptr(offset), _ptr(flow), _ptr(ws), _stream_of(center)))
        return center, offset, flow

    def image_resize_crop_normalise(self, images, res_hw, tables, window, mean, std):
        """images (n, H, W, 3) uint8 on the device; tables = (bounds_h, kk_h, bounds_v, kk_v) int32 device tensors, window =
        (y_first, tmp_h, crop_left, crop_top, crop_w, crop_h) -> (n, 3, crop_h, crop_w) float32."""
        n, in_h, in_w, _ = images.shape
        assert images.dtype == torch.uint8 and images.is_contiguous() and images.shape[3] == 3
        bounds_h, kk_h, bounds_v, kk_v = tables
        y_first, tmp_h, crop_left, crop_top, crop_w, crop_h = window
        tmp = torch.empty(n, tmp_h, crop_w, 3, dtype=torch.uint8, device=images.device)
        out = torch.empty(n, 3, crop_h, crop_w, dtype=torch.float32, device=images.device)
        mean3, std3 = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
        self.check(self.dll.fiery_image_resize_crop_normalise(_ptr(images), n, in_h, in_w, res_hw[0], res_hw[1], _ptr(bounds_h), _ptr(kk_h),
                                                              kk_h.shape[1], _ptr(bounds_v), _ptr(kk_v), kk_v.shape[1], y_first, tmp_h, crop_left,
                                                              crop_top, crop_w, crop_h, mean3, std3, _ptr(tmp), _ptr(out), _stream_of(out)))
        return out

    def upsample2x_bwd(self, grad_out, n_img, h, w, c):
        """grad_out: dense pixel-major (n_img, 2h, 2w, c) -> (n_img, h, w, c)."""
        assert grad_out.is_contiguous() and tuple(grad_out.shape) == (n_img, 2 * h, 2 * w, c)
        gx = torch.empty(n_img, h, w, c, dtype=torch.float32, device=grad_out.device)
        self.check(self.dll.fiery_upsample2x_bwd_nhwc(_ptr(grad_out), c, n_img, h, w, c, _ptr(gx), c, _stream_of(gx)))
        return gx

    def depthwise_conv(self, x, in_ld, n_img, h, w, c, weights, w_ld, k, stride, pad_top, pad_left, ho, wo, scale, shift, act,
                       out, out_ld):
        self.check(self.dll.fiery_depthwise_conv_nhwc(_ptr(x), in_ld, n_img, h, w, c, _ptr(weights), w_ld, k, stride, pad_top,
                                                      pad_left, ho, wo, _ptr(scale), _ptr(shift), act, _ptr(out), out_ld,
                                                      _stream_of(out)))

    def depthwise_conv_wgrad(self, x, in_ld, n_img, h, w, c, grad_out, g_ld, ho, wo, k, stride, pad_top, pad_left):
        """-> dw [k*k][c] (tap-major, like the forward's weights)."""
        dw = self.zeros_f32((k * k, c), x.device)
        self.check(self.dll.fiery_depthwise_conv_wgrad_nhwc(_ptr(x), in_ld, n_img, h, w, c, _ptr(grad_out), g_ld, ho, wo, k, stride,
                                                            pad_top, pad_left, _ptr(dw), c, _stream_of(dw)))
        return dw

    def instance_segmentation(self, center, offset, foreground, conf_threshold=0.1, max_centers=100):
        """center (n, H, W) f32, offset (n, 2, H, W) f32, foreground (n, H, W) uint8 ->
        (instance ids (n, H, W) int32, centres (n, max_centers, 2) int32, n_centres (n,) int32)."""
        n, h, w = center.shape
        seg = torch.empty(n, h, w, dtype=torch.int32, device=center.device)
        centers = torch.empty(n, max_centers, 2, dtype=torch.int32, device=center.device)
        count = torch.empty(n, dtype=torch.int32, device=center.device)
        self.check(self.dll.fiery_instance_segmentation(_ptr(center), _ptr(offset), _ptr(foreground), n, h, w, conf_threshold,
                                                        max_centers, _ptr(seg), _ptr(centers), _ptr(count), _stream_of(seg)))
        return seg, centers, count

    def se_gate(self, mean, mean_ld, n_img, c, w1, b1, hidden, w2, b2, gate, gate_ld):
        self.check(self.dll.fiery_se_gate(_ptr(mean), mean_ld, n_img, c, _ptr(w1), _ptr(b1), hidden, _ptr(w2), _ptr(b2),
                                          _ptr(gate), gate_ld, _stream_of(gate)))

    def se_gate_nhwc(self, x, ld, img_stride, n_img, n_pixels, c, w1, b1, hidden, w2, b2, gate, gate_ld, workspace):
        self.check(self.dll.fiery_se_gate_nhwc(_ptr(x), ld, img_stride, n_img, n_pixels, c, _ptr(w1), _ptr(b1), hidden, _ptr(w2),
                                               _ptr(b2), _ptr(gate), gate_ld, _ptr(workspace), _stream_of(gate)))

    def scale_channels(self, x, ld, n_img, hw, c, gate, gate_ld):
        self.check(self.dll.fiery_scale_channels_nhwc(_ptr(x), ld, n_img, hw, c, _ptr(gate), gate_ld, _stream_of(x)))

    def broadcast(self, v, v_ld, n_img, hw, c, out, out_ld, out_img_stride):
        self.check(self.dll.fiery_broadcast_nhwc(_ptr(v), v_ld, n_img, hw, c, _ptr(out), out_ld, out_img_stride,
                                                 _stream_of(out)))

    def nchw_to_nhwc(self, x, n_img, c, hw, out, out_ld, out_img_stride):
        self.check(self.dll.fiery_nchw_to_nhwc(_ptr(x), n_img, c, hw, _ptr(out), out_ld, out_img_stride, _stream_of(out)))

    def nhwc_to_nchw(self, x, in_ld, in_img_stride, n_img, c, hw, out):
        self.check(self.dll.fiery_nhwc_to_nchw(_ptr(x), in_ld, in_img_stride, n_img, c, hw, _ptr(out), _stream_of(out)))


def make_grid(origin, resolution, dim):
    g = BevGrid()
    for i in range(3):
        g.origin[i] = float(origin[i])
        g.resolution[i] = float(resolution[i])
        g.dim[i] = int(dim[i])
    return g


_LIB = None


def get():
    """The product library; raises NativeError when it has not been built."""
    global _LIB
    if _LIB is None:
        # FIERY_HIP_LIB: another build of the same library (A/B runs of two kernel versions on one GPU box)
        _LIB = Lib(os.environ.get('FIERY_HIP_LIB') or LIB_PATH)
    return _LIB


def is_built():
    return os.path.exists(LIB_PATH)
