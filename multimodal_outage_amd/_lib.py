"""ctypes binding of the C-ABI declared in include/mo_hip.h.

The product has NO CPU fallback: if libmo_hip.so is missing the import of any op raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmo_hip.so')

_lib = None
ABI_VERSION = 6          # == MO_ABI_VERSION of include/mo_hip.h; bump both when an entry point's arguments change

vp, i32, i64, f32, u32 = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint32

# name -> (restype, argtypes); mirrors include/mo_hip.h one to one
SIGNATURES = {
    'mo_strerror': (C.c_char_p, [i32]),
    'mo_version': (i32, []),
    'mo_set_option': (i32, [C.c_char_p, i32]),
    'mo_unet_set_option': (i32, [C.c_char_p, i32]),
    'mo_nchw_to_nbtc': (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
    'mo_nbtc_to_nchw': (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
    'mo_conv1x1_fwd': (i32, [vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, i64, i32, i32, vp]),
    'mo_skip_fwd': (i32, [vp, vp, vp, i32, vp, i32, i64, i32, vp, i32, i32, vp, vp]),
    'mo_conv1x1_bwd_data_smallk': (i32, [vp, i32, i64, vp, i32, vp, vp, vp, vp]),
    'mo_linear_splitk_ws_floats': (i64, [i64, i32, i32]),
    'mo_conv1x1_fwd_splitk': (i32, [vp, i32, vp, vp, i32, vp, i64, i32, vp, vp]),
    'mo_conv1x1_bwd_data_splitk': (i32, [vp, i32, i64, vp, i32, vp, vp, vp]),
    'mo_conv1x1_bwd_data': (i32, [vp, i32, i64, vp, i32, vp, i32, i32, i32, vp, i32, vp]),
    'mo_wgrad_ws_floats': (i64, [i32, i32, i64]),
    'mo_conv1x1_bwd_weight': (i32, [vp, i32, i64, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    'mo_adp_fwd': (i32, [vp, vp, i32, i32, vp, vp, vp]),
    'mo_adp_bwd': (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, i64, vp]),
    'mo_tcn_pack_weights': (i32, [vp, vp, i32, vp, vp]),
    'mo_tcn_fwd': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i64, i32, vp, vp, i32, vp, i32, vp]),
    'mo_tcn_bwd': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i64, i32, vp, vp, vp, vp, vp, vp, vp,
                         vp, vp, i32, i32, vp]),
    'mo_spmm_csr': (i32, [vp, vp, vp, i32, vp, vp, i64, i32, i32, i32, vp]),
    'mo_adj_gemm': (i32, [vp, i32, vp, vp, i64, i32, vp]),
    'mo_adj_grad': (i32, [vp, vp, i32, i64, vp, i32, vp]),
    'mo_gemm_bf16': (i32, [vp, i32, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp]),
    'mo_gemm_bf16_ex': (i32, [vp, i32, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp]),
    'mo_f32_to_bf16': (i32, [vp, vp, i64, vp]),
    'mo_gemm_bf16_256': (i32, [vp, i32, i32, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp]),
    'mo_skip_bwd_add': (i32, [vp, i32, i32, i64, i32, i32, vp, vp]),
    'mo_spmm_blk': (i32, [vp, vp, vp, vp, vp, i32, i32, vp, vp, i64, i32, i32, vp]),
    'mo_spmm_blk2': (i32, [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, i64, i32, i32, vp]),
    'mo_skip_gather_bf16': (i32, [vp, vp, i32, i64, i32, vp, vp]),
    'mo_skip_wsplit': (i32, [vp, i32, i32, vp, vp]),
    'mo_wgrad_bf16_kk_ws_floats': (i64, [i32, i32, i64]),
    'mo_wgrad_bf16_kk': (i32, [vp, i32, vp, i32, i64, i32, i32, vp, vp, vp]),
    'mo_gemm_bf16_256_ex': (i32, [vp, i32, i32, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp]),
    'mo_f32_to_bf16_padded': (i32, [vp, i32, i32, vp, i32, vp]),
    'mo_mlp_partial_floats': (i64, [i64]),
    'mo_gcn_mlp_fwd': (i32, [vp, i32, vp, vp, i64, i32, i32, vp, vp, vp, u32, u32, f32, vp, vp, i32, vp]),
    'mo_bn_finalize': (i32, [vp, i64, i64, vp, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp]),
    'mo_bn_bwd': (i32, [vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp]),
    'mo_gcn_mlp_bwd': (i32, [vp, vp, vp, i32, vp, i64, u32, u32, f32, vp, vp, vp, vp, i32, i32, i32, vp]),
    'mo_metrics_ws_floats': (i64, [i64]),
    'mo_mse_metrics': (i32, [vp, vp, i64, vp, vp, vp, vp]),
    'mo_date2vec_encode': (i32, [vp, i64, vp, vp, i32, vp, vp, i32, vp, vp]),
    'mo_colsum': (i32, [vp, i64, i32, vp, vp, vp]),
    'mo_colsum_ws_floats': (i64, [i64, i32]),
    'mo_adam_step': (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, f32, vp]),
    # ---- small-graph Graph WaveNet body (one workgroup per call)
    'mo_gwnet_small_supported': (i32, [i32, i32, i32, i32, i32]),
    'mo_gwnet_small_fwd': (i32, [i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, u32, u32, f32, vp]),
    'mo_gwnet_small_bwd_ws_floats': (i64, [i32, i32, i32, i32, i32, i32]),
    'mo_gwnet_small_slab_floats': (i64, [i32]),
    'mo_gwnet_small_bwd': (i32, [i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, f32, u32, u32, f32,
                                 vp, vp, vp, vp, vp]),
    # ---- UNet
    'mo_conv3x3_fwd': (i32, [vp, i32, i64, vp, vp, i32, vp, i32, i64, vp, vp, i32, i32, vp, i32, i64, i32, i32,
                             vp, i64, vp, i32, vp, vp]),
    'mo_conv3x3_stats_tiles': (i32, [i32, i64, i32, i32]),
    'mo_conv3x3_stats_tiles2': (i32, [i32, i32, i32, i64, i32, i32, i32]),
    'mo_conv3x3_bf16_route': (i32, [i32, i32, i64, i32, i32]),
    'mo_conv3x3_flip_weights': (i32, [vp, i32, i32, vp, vp]),
    'mo_unet_wgrad_ws_floats': (i64, [i32, i32, i64]),
    'mo_conv3x3_bwd_weight': (i32, [vp, i64, i32, vp, i32, i64, vp, vp, i32, vp, i32, i64, vp, vp, i32, i32,
                                    i64, i32, i32, vp, vp, i32, vp, vp]),
    'mo_nchw_conv1x1_fwd': (i32, [vp, i64, i32, vp, vp, i32, i32, vp, vp, i32, i64, i32, vp, i64, i32, vp]),
    'mo_nchw_conv1x1_bwd_data': (i32, [vp, i64, i32, vp, i32, i64, i32, vp, i64, i32, vp]),
    'mo_nchw_conv1x1_bwd_weight': (i32, [vp, i64, i32, vp, i64, i32, vp, vp, i32, i32, i64, i32, vp, vp, vp, i32, vp]),
    'mo_convt2x2_bf16_route': (i32, [i32, i32, i64]),
    'mo_convt2x2_fwd': (i32, [vp, i64, i32, vp, vp, i32, i32, vp, vp, i32, i64, i32, i32, vp, i64, i32, vp]),
    'mo_convt2x2_bwd_data': (i32, [vp, i64, i32, vp, i32, i64, i32, i32, vp, i64, i32, vp]),
    'mo_convt2x2_bwd_weight': (i32, [vp, i64, i32, vp, i64, i32, vp, vp, i32, i32, i64, i32, i32, vp, vp, vp, i32, vp]),
    'mo_fc3_supported': (i32, [i64, i32, i32]),
    'mo_fc3_ws_floats': (i64, [i64, i32, i32]),
    'mo_fc3_fwd': (i32, [vp, i64, i32, vp, vp, i32, i32, vp, vp, vp]),
    'mo_fc3_bwd_data': (i32, [vp, i64, i32, vp, i32, vp, vp, vp]),
    'mo_fc3_wgrad_ws_floats': (i64, [i64, i32, i32]),
    'mo_fc3_bwd_weight': (i32, [vp, i64, i32, vp, i32, vp, vp, vp, vp]),
    'mo_nchw_stats': (i32, [vp, i64, i32, i64, i32, vp, vp]),
    'mo_group_bn_finalize': (i32, [vp, i64, i32, i32, i32, i32, vp, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp]),
    'mo_group_bn_finalize2': (i32, [vp, i64, i32, i32, i32, i32, vp, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp, vp]),
    'mo_unet_act': (i32, [vp, i64, i32, i64, i32, i32, vp, vp, i32, i32, vp, i64, i32, vp]),
    'mo_unet_act_bwd_ws_floats': (i64, [i64, i32]),
    'mo_unet_act_bwd': (i32, [vp, i64, i32, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp, i64,
                              vp, vp, vp, i32, vp, vp]),
    'mo_outc_loss_ws_floats': (i64, [i64, i32, i32, i32]),
    'mo_outc_loss_fwd': (i32, [vp, i64, i32, vp, vp, i32, i32, vp, vp, i32, vp, vp, i64, i32, vp, vp, i64, vp, vp, i32,
                               vp]),
    'mo_outc_loss_bwd': (i32, [vp, i64, i32, vp, vp, i32, i32, vp, vp, i32, vp, vp, i64, i32, vp, vp, vp, vp, i32, vp]),
    'mo_nchw_channel_sum': (i32, [vp, i64, i32, i64, i32, vp, vp, vp]),
    'mo_maxpool2_bwd': (i32, [vp, i64, i32, i64, i32, i32, vp, i64, vp, i64, vp]),
    'mo_raster_prepare': (i32, [vp, i64, i32, i32, f32, f32, f32, vp, i32, i32, vp]),
    'mo_dropout': (i32, [vp, vp, i64, u32, u32, f32, vp]),
    'mo_relu_bwd': (i32, [vp, vp, vp, i64, vp]),
    # ---- RCCL exchange step / host CSR builder
    'mo_allreduce_unique_id': (i32, [vp]),
    'mo_allreduce_init': (i32, [vp, i32, i32, C.POINTER(vp)]),
    'mo_allreduce_launch': (i32, [vp, vp, i64, i32, vp]),
    'mo_allreduce_wait': (i32, [vp, vp]),
    'mo_allreduce_destroy': (i32, [vp]),
    'mo_csr_from_dense': (i32, [vp, i32, i32, vp, vp, vp, C.POINTER(i64)]),
}


def load():
    """Load libmo_hip.so (built in-tree by multimodal_outage_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} not found: the HIP extension is required (no CPU fallback). '
            'Build it with `python -m multimodal_outage_amd.build`.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    have = lib.mo_version()
    if have != ABI_VERSION:
        raise RuntimeError(f'{LIB_PATH} is stale: it reports ABI version {have}, this package binds version '
                           f'{ABI_VERSION}; rebuild it with `python -m multimodal_outage_amd.build --force`')
    _lib = lib
    return lib


def check(code, what=''):
    if code != 0:
        msg = load().mo_strerror(code).decode()
        raise RuntimeError(f'mo_hip {what}: {msg} (code {code})')


BF_IN0, BF_IN1, BF_OUT, BF_DY, BF_DP = 1, 2, 4, 8, 16     # `dtypes` flags of the UNet entry points (include/mo_hip.h)
BF_MATH, W_FLIP = 32, 64                                  # bf16 matrix-pipe arithmetic; weights read transposed + flipped


def ptr(t):
    """Device pointer of a contiguous fp32/int32 CUDA(HIP) tensor, or NULL."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), 'mo_hip ops need contiguous device tensors'
    return t.data_ptr()


def stream():
    """hipStream_t of torch's current stream on the current device (the raw query: torch.cuda.current_stream() builds a
    Stream object, ~10 us per call and one call per launch)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


_fns = {}


def call(name, *args):
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        check(rc, name)
