"""ctypes binding of libm355seg.so (include/m355seg.h).

There is NO fallback: if the HIP library has not been built the import of any
op raises, and ops called with non-GPU tensors raise.  The product path never
touches oracle/.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# M355_LIB_PATH: load another build of the SAME library (the diagnostic build with in-kernel cycle stamps,
# build.py --stamps); never a fallback -- a missing file fails exactly like a missing product library.
LIB_PATH = os.environ.get("M355_LIB_PATH") or os.path.join(_HERE, "libm355seg.so")

M355_OK = 0
ACT_NONE, ACT_RELU, ACT_LEAKY_RELU = 0, 1, 2
COMPUTE_F32, COMPUTE_BF16, COMPUTE_F16, COMPUTE_F32X3 = 0, 1, 2, 3
CONV_W_PACKED, CONV_SOFTMAX = 1, 2


class ConvDesc(C.Structure):
    """m355_conv3d_desc"""
    _fields_ = [
        ("N", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
        ("D", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("k", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("out_pad", C.c_int32),
        ("x_batch_stride", C.c_int64), ("y_batch_stride", C.c_int64),
        ("compute", C.c_int32), ("flags", C.c_int32),
    ]


class PackItem(C.Structure):
    """m355_pack_item"""
    _fields_ = [("desc", ConvDesc), ("which", C.c_int32), ("w", C.c_void_p), ("packed", C.c_void_p)]


class NormDesc(C.Structure):
    """m355_norm_desc"""
    _fields_ = [
        ("N", C.c_int32), ("C", C.c_int32), ("S", C.c_int64),
        ("groups", C.c_int32), ("act", C.c_int32),
        ("eps", C.c_float), ("act_slope", C.c_float),
        ("x_batch_stride", C.c_int64), ("y_batch_stride", C.c_int64), ("add_batch_stride", C.c_int64),
    ]


_P = C.c_void_p
ABI_VERSION = 3   # M355_ABI_VERSION of include/m355seg.h this binding was written against
_i32, _i64, _f32, _sz = C.c_int32, C.c_int64, C.c_float, C.c_size_t
_CD, _ND = C.POINTER(ConvDesc), C.POINTER(NormDesc)

# name -> (restype, argtypes); mirrors include/m355seg.h one to one
SIGNATURES = {
    "m355_version": (C.c_int, []),
    "m355_last_error": (C.c_char_p, []),
    "m355_reload_tuning": (None, []),
    "m355_queue_pool_bytes": (_sz, []),
    "m355_queue_pool_set": (C.c_int, [_P, _sz, _i32]),
    "m355_overflow_flag_set": (C.c_int, [_P, _i32]),
    "m355_conv3d_fuses_softmax": (_i32, [_CD]),
    "m355_conv3d_packed_bytes": (_sz, [_CD, _i32]),
    "m355_conv3d_pack": (C.c_int, [_CD, _i32, _P, _P, _P]),
    "m355_conv3d_pack_batch": (C.c_int, [_P, _i32, _P]),
    "m355_conv3d_fwd_workspace": (_sz, [_CD]),
    "m355_conv3d_fwd": (C.c_int, [_CD, _P, _P, _P, _P, _P, _P, _sz, _P]),
    "m355_conv3d_stats_slots": (_i64, [_CD]),
    "m355_conv3d_fwd_stats": (C.c_int, [_CD, _P, _P, _P, _P, _P, _P, _P, _sz, _P]),
    "m355_act16_bytes": (_sz, [_i32, _i32, _i64]),
    "m355_act16_pack": (C.c_int, [_P, _P, _i32, _i32, _i64, _i64, _i64, _i32, _P]),
    "m355_act16_unpack": (C.c_int, [_P, _P, _i32, _i32, _i64, _i64, _i64, _i32, _P]),
    "m355_conv3d_h16_workspace": (_sz, [_CD, _i32]),
    "m355_conv3d_fwd_h16": (C.c_int, [_CD, _P, _i64, _P, _P, _P, _P, _P, _P, _sz, _P]),
    "m355_conv3d_stats_slots_c8": (_i64, [_CD]),
    "m355_conv3d_fwd_h16_c8": (C.c_int, [_CD, _P, _i64, _P, _P, _P, _i64, _P, _P, _sz, _P]),
    "m355_conv3d_bwd_data_h16": (C.c_int, [_CD, _P, _i64, _P, _P, _P, _sz, _P]),
    "m355_conv3d_bwd_weight_h16_workspace": (_sz, [_CD]),
    "m355_conv3d_bwd_weight_h16": (C.c_int, [_CD, _P, _i64, _P, _i64, _P, _P, _P, _P, _sz, _P]),
    "m355_norm_act_fwd_h16": (C.c_int, [_ND, _P, _P, _P, _P, _P, _P, _P, _P, _i64, _i32, _P]),
    "m355_norm_act_fwd_c8": (C.c_int, [_ND, _P, _i64, _P, _P, _P, _P, _P, _i64, _P, _i64, _i32, _P]),
    "m355_act16_partials_slots": (_i64, [_i64]),
    "m355_act16_channel_partials": (C.c_int, [_P, _i64, _i32, _i32, _i64, _i32, _P, _P]),
    "m355_avgpool3d_2x_fwd_h16": (C.c_int, [_P, _P, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i32, _P]),
    "m355_act16_pack_scaled": (C.c_int, [_P, _P, _i32, _i32, _i64, _i64, _i64, _i32, _f32, _P]),
    "m355_act16_unpack_scaled": (C.c_int, [_P, _P, _i32, _i32, _i64, _i64, _i64, _i32, _f32, _P]),
    "m355_conv3d_bwd_data_h16_c8": (C.c_int, [_CD, _P, _i64, _P, _P, _i64, _P, _sz, _P]),
    "m355_conv3d_bwd_weight_c8_workspace": (_sz, [_CD]),
    "m355_conv3d_bwd_weight_c8": (C.c_int, [_CD, _P, _i64, _P, _i64, _P, _P, _f32, _P, _sz, _P]),
    "m355_norm_act_bwd_c8": (C.c_int, [_ND, _P, _i64, _P, _i64, _P, _i64, _i32, _i32, _i32, _P, _P, _P, _P, _P, _i64, _P, _P,
                                       C.c_int, _f32, _i32, _P, _sz, _P]),
    "m355_norm_act_bwd_c8_reduce": (C.c_int, [_ND, _P, _i64, _P, _i64, _P, _i64, _i32, _i32, _i32, _P, _P, _P, _P, _P, _P,
                                              C.c_int, _P, _f32, _P, _i32, _P, _sz, _P]),
    "m355_norm_act_bwd_c8_apply": (C.c_int, [_ND, _P, _i64, _P, _i64, _P, _i64, _i32, _i32, _i32, _P, _P, _P, _P, _P, _P, _i64,
                                             _i32, _P]),
    "m355_avgpool3d_2x_bwd_h16": (C.c_int, [_P, _P, _P, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _P]),
    "m355_upsample_trilinear2x_fwd_h16": (C.c_int, [_P, _P, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i32, _P]),
    "m355_upsample_trilinear2x_bwd_h16": (C.c_int, [_P, _P, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i32, _P]),
    "m355_act16_channel_scale": (C.c_int, [_P, _P, _P, _i32, _i32, _i64, _i64, _i64, _i32, _P]),
    "m355_space_to_depth2_h16": (C.c_int, [_P, _P, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i32, _P]),
    "m355_depth_to_space2_h16": (C.c_int, [_P, _P, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i32, _P]),
    "m355_conv_transpose3d_h16_bwd_supported": (_i32, [_CD]),
    "m355_conv_transpose3d_h16_bwd_workspace": (_sz, [_CD]),
    "m355_conv_transpose3d_bwd_data_h16": (C.c_int, [_CD, _P, _i64, _P, _P, _i64, _i32, _P]),
    "m355_conv_transpose3d_bwd_weight_h16": (C.c_int, [_CD, _P, _i64, _P, _i64, _P, _P, _f32, _i32, _P, _sz, _P]),
    "m355_conv3d_plan": (C.c_int, [_CD, _i32, C.POINTER(C.c_int32)]),
    "m355_conv3d_bwd_data_workspace": (_sz, [_CD]),
    "m355_conv3d_bwd_data": (C.c_int, [_CD, _P, _P, _P, _P, _sz, _P]),
    "m355_conv3d_bwd_weight_workspace": (_sz, [_CD]),
    "m355_conv3d_bwd_weight": (C.c_int, [_CD, _P, _P, _P, _P, _P, _sz, _P]),
    "m355_conv_transpose3d_workspace": (_sz, [_CD]),
    "m355_conv_transpose3d_fwd": (C.c_int, [_CD, _P, _P, _P, _P, _P, _sz, _P]),
    "m355_conv_transpose3d_bwd_data": (C.c_int, [_CD, _P, _P, _P, _P, _sz, _P]),
    "m355_conv_transpose3d_bwd_weight": (C.c_int, [_CD, _P, _P, _P, _P, _P, _sz, _P]),
    "m355_conv_transpose3d_fwd_h16": (C.c_int, [_CD, _P, _i64, _P, _P, _P, _i64, _i32, _P]),
    "m355_norm_num_stats": (_i64, [_ND]),
    "m355_norm_workspace": (_sz, [_ND]),
    "m355_norm_stats": (C.c_int, [_ND, _P, _P, _P, _P, _P, _f32, _P, _sz, _P]),
    "m355_norm_stats_from_partials": (C.c_int, [_ND, _P, _i64, _P, _P, _P, _P, _f32, _P, _sz, _P]),
    "m355_norm_stats_from_running": (C.c_int, [_ND, _P, _P, _P, _P, _P]),
    "m355_norm_act_fwd": (C.c_int, [_ND, _P, _P, _P, _P, _P, _P, _P, _P]),
    "m355_norm_act_pool_fwd": (C.c_int, [_ND, _P, _P, _P, _P, _P, _P, _P, _i64, _i32, _i32, _i32, _P]),
    "m355_norm_act_bwd": (C.c_int, [_ND, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P, _sz, _P]),
    "m355_norm_sums": (C.c_int, [_ND, _P, _P, _i64, _P, _P, _sz, _P]),
    "m355_norm_stats_from_sums": (C.c_int, [_ND, _P, _P, _P, _P, _P, _f32, _P]),
    "m355_norm_act_bwd_reduce": (C.c_int, [_ND, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _sz, _P]),
    "m355_norm_act_bwd_apply": (C.c_int, [_ND, _P, _P, _P, _P, _P, _P, _P, _P, _P, _i64, _i32, _P]),
    "m355_norm_act_bwd_h16": (C.c_int, [_ND, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P, _i64, _i32, _P, _sz, _P]),
    "m355_avgpool3d_2x_fwd": (C.c_int, [_P, _P, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _P]),
    "m355_avgpool3d_2x_bwd": (C.c_int, [_P, _P, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _P]),
    "m355_avgpool3d_2x_bwd_add": (C.c_int, [_P, _P, _P, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _P]),
    "m355_upsample_trilinear2x_fwd": (C.c_int, [_P, _P, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _P]),
    "m355_upsample_trilinear2x_bwd": (C.c_int, [_P, _P, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _P]),
    "m355_softmax_fwd": (C.c_int, [_P, _P, _i32, _i32, _i32, _i64, _f32, _P]),
    "m355_softmax_bwd": (C.c_int, [_P, _P, _P, _i32, _i32, _i32, _i64, _P]),
    "m355_hybrid_loss_workspace": (_sz, [_i32, _i32, _i64]),
    "m355_hybrid_loss_fwd": (C.c_int, [_P, _P, _i32, _i32, _i64, _f32, _P, _i32, _P, _P, _P, _sz, _P]),
    "m355_hybrid_loss_bwd": (C.c_int, [_P, _P, _P, _P, _i32, _i32, _i64, _f32, _P, _i32, _P, _P]),
    "m355_copy_channels": (C.c_int, [_P, _P, _i32, _i32, _i64, _i64, _i64, _P]),
    "m355_channel_scale": (C.c_int, [_P, _P, _P, _i32, _i32, _i64, _P]),
    "m355_add": (C.c_int, [_P, _P, _P, _i64, _P]),
    "m355_space_to_depth2": (C.c_int, [_P, _P, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _P]),
    "m355_depth_to_space2": (C.c_int, [_P, _P, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _P]),
    "m355_blur_weight_fwd": (C.c_int, [_P, _P, _P, _P, _i32, _i32, _i32, _i32, _P]),
    "m355_blur_weight_bwd": (C.c_int, [_P, _P, _P, _P, _P, _i32, _i32, _i32, _i32, _P]),
    "m355_weight_standardize_fwd": (C.c_int, [_P, _P, _P, _i32, _i32, _P]),
    "m355_weight_standardize_bwd": (C.c_int, [_P, _P, _P, _P, _i32, _i32, _P]),
    "m355_sampler_table_bytes": (_sz, [_i32, _i32, _i32]),
    "m355_sampler_build": (C.c_int, [_P] + [_i32] * 6 + [_P, _P]),
    "m355_sampler_draw": (C.c_int, [_P, _P] + [_i32] * 6 + [_P, _i32, _P, _P]),
    "m355_patch_gather": (C.c_int, [_P, _P, _P] + [_i32] * 8 + [_P]),
    "m355_patch_aggregate_grid": (C.c_int, [_P, _P, _i32, _i32, _i32, _P] + [_i32] * 10 + [_P]),
    "m355_patch_accumulate": (C.c_int, [_P, _P, _P, _P] + [_i32] * 8 + [_P]),
    "m355_patch_gather_padded": (C.c_int, [_P, _P, _P] + [_i32] * 12 + [_f32, _P]),
    "m355_patch_finalize_crop": (C.c_int, [_P, _P, _P] + [_i32] * 7 + [_P]),
    "m355_patch_finalize": (C.c_int, [_P, _P, _P, _i32, _i64, _P]),
    "m355_flip_permute": (C.c_int, [_P, _P, _i32, _i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _i32, _P]),
    "m355_ensemble_accumulate": (C.c_int, [_P, _P, _P, _i32, _i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _i32, _i32,
                                           _i32, _P]),
    "m355_ensemble_finalize": (C.c_int, [_P, _P, _P, _P, _i32, _i32, _i64, _i32, _i32, _P]),
    "m355_argmax_confusion": (C.c_int, [_P, _P, _P, _P, _i32, _i32, _i64, _P]),
}


class M355Error(RuntimeError):
    pass


def bind(lib, prefix="m355_"):
    """Declare argtypes/restype of every ABI symbol on a loaded CDLL.

    `prefix` lets the test-suite bind the CPU oracle (m355o_*) with the same
    signatures; the product only ever binds "m355_".
    """
    for name, (res, args) in SIGNATURES.items():
        sym = prefix + name[len("m355_"):]
        fn = getattr(lib, sym, None)
        if fn is None:
            if prefix == "m355_":
                raise M355Error(f"{LIB_PATH} does not export {sym}")
            continue
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def lib():
    """The loaded HIP library; raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise M355Error(
                f"HIP extension {LIB_PATH} is missing. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc, no GPU). "
                "There is no CPU or PyTorch fallback for the hot path.")
        _lib = bind(C.CDLL(LIB_PATH))
        if _lib.m355_version() != ABI_VERSION:
            raise M355Error(f"ABI version mismatch: library reports {_lib.m355_version()}, expected {ABI_VERSION}")
    return _lib


_queue_pools = {}   # device index -> the zero-filled tensor handed to m355_queue_pool_set (kept alive here)


def ensure_queue_pool(device):
    """Hand the library its work-queue pool for `device` from torch's allocator (once per device): with this host the
    library never allocates device memory itself (include/m355seg.h, "Conventions")."""
    idx = device.index if device.index is not None else 0
    if idx in _queue_pools:
        return
    import torch
    L = lib()
    n = int(L.m355_queue_pool_bytes())
    with torch.cuda.device(idx):
        if torch.cuda.is_current_stream_capturing():
            return   # (a capture's warm-up ran eagerly before: the pool exists unless the caller skipped the warm-up)
        buf = torch.zeros(n, dtype=torch.uint8, device=device)
        torch.cuda.current_stream().synchronize()
    rc = L.m355_queue_pool_set(C.c_void_p(buf.data_ptr()), n, idx)
    # M355_EUNSUPPORTED: another binding in this process already set / allocated one -- fine, it is in use
    _queue_pools[idx] = buf if rc == M355_OK else None


def reload_tuning():
    """Re-read the M355_* environment overrides (the library caches them at load time)."""
    lib().m355_reload_tuning()


def check(rc, what):
    if rc != M355_OK:
        msg = lib().m355_last_error().decode("utf-8", "replace")
        raise M355Error(f"{what} failed (status {rc}): {msg}")
