"""ctypes binding of libhpfg_hip.so (C ABI declared in include/hpfg_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If it is missing or a call fails the error is raised
immediately (``HipLibraryError``), never swallowed.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libhpfg_hip.so")

# enums from include/hpfg_hip.h
BN_MEAN, BN_RSTD, BN_SCALE, BN_SHIFT, BN_K1, BN_K2, BN_K3, BN_SPARE, BN_ROWS = range(9)
ACT_NONE, ACT_PLAIN, ACT_STRIDED, ACT_BNACT, ACT_BNACT_POOL, ACT_UP2X, ACT_DZ, ACT_SPLIT16, ACT_UPBWD = range(9)
OPT_CONV_THIN, OPT_FIRST_MFMA, OPT_FIRST_WGRAD, OPT_NARROW_DEEP = 0, 1, 2, 3          # hpfg_set_option
LOSS_NSUM = 32
ACC_MAX_SHARDS = 8          # HPFG_ACC_MAX_SHARDS: a BatchNorm sum accumulator is long long [shards][2][C][2]
VERSION = 132
MATH_F32, MATH_BF16X3 = 0, 1


class HipLibraryError(RuntimeError):
    pass


class Act(C.Structure):
    _fields_ = [("z", C.c_void_p), ("bn", C.c_void_p), ("aux", C.c_void_p), ("mode", C.c_int32), ("C", C.c_int32),
                ("Hs", C.c_int32), ("Ws", C.c_int32), ("pstride", C.c_int32), ("aux_pstride", C.c_int32),
                ("bn_stride", C.c_int32), ("bn_coff", C.c_int32), ("sn", C.c_int32), ("sc", C.c_int32), ("sy", C.c_int32),
                ("sx", C.c_int32), ("drop_p", C.c_float), ("drop_seed", C.c_uint32), ("drop_mask", C.c_void_p), ("seed_dev", C.c_void_p),
                ("bn_acc", C.c_void_p), ("bn_gamma", C.c_void_p), ("bn_beta", C.c_void_p), ("bn_count", C.c_float), ("bn_eps", C.c_float),
                ("bn_shards", C.c_int32), ("reserved1", C.c_int32)]


class ConvArgs(C.Structure):
    _fields_ = [("a0", Act), ("a1", Act), ("wpk", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
                ("stat_partials", C.c_void_p), ("out_pstride", C.c_int32), ("Cout", C.c_int32), ("CoutPad", C.c_int32),
                ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("taps", C.c_int32), ("math", C.c_int32),
                ("bwd_stats", C.c_int32), ("bwd_of", Act), ("out2", C.c_void_p), ("out_split", C.c_int32), ("out2_pstride", C.c_int32),
                ("stat_acc", C.c_void_p), ("stat_shards", C.c_int32), ("reserved2", C.c_int32), ("stage_out", C.c_void_p), ("side_sums", C.c_void_p)]


class WgradArgs(C.Structure):
    _fields_ = [("a0", Act), ("a1", Act), ("g", Act), ("slab", C.c_void_p), ("dw_oihw", C.c_void_p), ("Cin", C.c_int32),
                ("CinPad", C.c_int32), ("Cout", C.c_int32), ("CoutPad", C.c_int32), ("N", C.c_int32), ("H", C.c_int32),
                ("W", C.c_int32), ("taps", C.c_int32), ("S", C.c_int32), ("math", C.c_int32), ("defer_reduce", C.c_int32)]


class PeerX(C.Structure):
    _fields_ = [("mbox", C.c_void_p * 8), ("epoch", C.c_void_p), ("err", C.c_void_p), ("world", C.c_int32), ("rank", C.c_int32),
                ("slot", C.c_int32), ("cap", C.c_int32), ("slot_bytes", C.c_int64)]


class PeerBuf(C.Structure):
    _fields_ = [("win", C.c_void_p * 8), ("epoch", C.c_void_p), ("err", C.c_void_p), ("world", C.c_int32), ("rank", C.c_int32),
                ("slice", C.c_int64), ("n", C.c_int64), ("stride", C.c_int64)]


class FusedBwdArgs(C.Structure):
    _fields_ = [("d", ConvArgs), ("xa0", Act), ("xa1", Act), ("slab", C.c_void_p), ("Cin", C.c_int32), ("CinPad", C.c_int32),
                ("Cout", C.c_int32), ("CoutPad", C.c_int32)]


class AugSample(C.Structure):
    _fields_ = [("img_off", C.c_int64), ("lab_off", C.c_int64), ("h", C.c_int32), ("w", C.c_int32), ("mode", C.c_int32), ("k", C.c_int32),
                ("axis", C.c_int32), ("tab_off", C.c_int32), ("m00", C.c_double), ("m01", C.c_double), ("m10", C.c_double),
                ("m11", C.c_double), ("off_y", C.c_double), ("off_x", C.c_double)]


class BnAccDesc(C.Structure):
    _fields_ = [("acc", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("bn", C.c_void_p), ("C", C.c_int32), ("count", C.c_float), ("shards", C.c_int32), ("reserved", C.c_int32)]


class BnAccBwdDesc(C.Structure):
    _fields_ = [("acc", C.c_void_p), ("gamma", C.c_void_p), ("bn", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("C", C.c_int32), ("count", C.c_float), ("shards", C.c_int32), ("reserved", C.c_int32)]


class SlabDesc(C.Structure):
    _fields_ = [("slab", C.c_void_p), ("dw_oihw", C.c_void_p), ("S", C.c_int32), ("taps", C.c_int32), ("Cin", C.c_int32),
                ("CinPad", C.c_int32), ("Cout", C.c_int32), ("CoutPad", C.c_int32)]


class PackDesc(C.Structure):
    _fields_ = [("w_oihw", C.c_void_p), ("b", C.c_void_p), ("wpk_fwd", C.c_void_p), ("wpk_dgrad", C.c_void_p),
                ("bias_pad", C.c_void_p), ("wpk16_fwd", C.c_void_p), ("wpk16_dgrad", C.c_void_p), ("Cout", C.c_int32), ("Cin", C.c_int32),
                ("CoutPad", C.c_int32), ("CinPad", C.c_int32), ("taps", C.c_int32), ("kc", C.c_int32)]


class LossArgs(C.Structure):
    _fields_ = [("logits", C.c_void_p), ("t_logits", C.c_void_p), ("labels0", C.c_void_p), ("labels1", C.c_void_p),
                ("coef", C.c_void_p), ("partials", C.c_void_p), ("sums", C.c_void_p), ("out", C.c_void_p),
                ("dlogits", C.c_void_p), ("N", C.c_int32), ("n_lab", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("C", C.c_int32), ("world", C.c_int32), ("input_is_prob", C.c_int32), ("teacher_is_prob", C.c_int32),
                ("t_unlab_only", C.c_int32), ("reserved0", C.c_int32), ("cons_mask", C.c_void_p)]


class PredBlocks(C.Structure):
    _fields_ = [("p", C.c_void_p * 8), ("n_blocks", C.c_int32), ("per_block", C.c_int32)]


_i, _l, _f, _d, _p, _u32 = C.c_int, C.c_long, C.c_float, C.c_double, C.c_void_p, C.c_uint32

# name -> (restype, argtypes); must list every symbol include/hpfg_hip.h declares (checked by tests/test_abi.py)
PROTOTYPES = {
    "hpfg_version": (_i, []),
    "hpfg_last_error": (C.c_char_p, []),
    "hpfg_set_option": (_i, [_i, _i]),
    "hpfg_conv3x3_first_fwd": (_i, [C.POINTER(Act), _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "hpfg_conv_fwd": (_i, [C.POINTER(ConvArgs), _p]),
    "hpfg_conv_stat_blocks": (_i, [_i, _i, _i]),
    "hpfg_conv_stat_rows": (_i, [C.POINTER(ConvArgs)]),
    "hpfg_bn_fwd_finalize": (_i, [_p, _i, _p, _d, _p, _p, _p, _p, _f, _f, _p, _i, _p]),
    "hpfg_reduce_partials": (_i, [_p, _i, _i, _p, _p]),
    "hpfg_conv_kc": (_i, [_i, _i, _i]),
    "hpfg_wpk16_elems": (_l, [_i, _i, _i, _i]),
    "hpfg_pack_weights": (_i, [_p, C.POINTER(PackDesc), _i, _p]),
    "hpfg_pack_weights_bump": (_i, [_p, C.POINTER(PackDesc), _i, _p, _i, _p, _i, _p, _l, _p]),
    "hpfg_conv3x3_first_fwd_acc": (_i, [C.POINTER(Act), _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "hpfg_bn_acc_finalize": (_i, [_p, C.POINTER(BnAccDesc), _i, _f, _f, _p]),
    "hpfg_bn_acc_bwd_finalize": (_i, [_p, C.POINTER(BnAccBwdDesc), _i, _p]),
    "hpfg_bn_bwd_reduce_acc": (_i, [C.POINTER(Act), _i, _i, _i, _p, _i, _p]),
    "hpfg_bn_bwd_reduce_pool_acc": (_i, [C.POINTER(Act), _p, _i, _i, _i, _i, _p, _i, _p]),
    "hpfg_act_materialize": (_i, [C.POINTER(Act), C.POINTER(Act), _i, _i, _i, _p, _p]),
    "hpfg_dropout_mask": (_i, [_p, _l, _f, _u32, _p, _p]),
    "hpfg_bn_eval_table": (_i, [_p, _p, _p, _p, _f, _p, _i, _p]),
    "hpfg_bn_bwd_reduce": (_i, [C.POINTER(Act), _i, _i, _i, _p, _p]),
    "hpfg_bn_bwd_reduce_pool": (_i, [C.POINTER(Act), _p, _i, _i, _i, _i, _p, _p]),
    "hpfg_bn_bwd_pool_blocks": (_i, [_i, _i, _i, _i]),
    "hpfg_bn_bwd_blocks": (_i, [_i, _i, _i, _i]),
    "hpfg_bn_bwd_finalize": (_i, [_p, _i, _p, _d, _p, _p, _p, _p, _i, _f, _p]),
    "hpfg_wgrad": (_i, [C.POINTER(WgradArgs), _p]),
    "hpfg_fused_bwd": (_i, [C.POINTER(FusedBwdArgs), _p]),
    "hpfg_fused_bwd_grid": (_i, [C.POINTER(FusedBwdArgs)]),
    "hpfg_peer_slot_bytes": (_l, [_i, _i]),
    "hpfg_peer_alloc": (_i, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "hpfg_peer_free": (_i, [_p]),
    "hpfg_peer_handle": (_i, [_p, C.c_char_p]),
    "hpfg_peer_open": (_i, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "hpfg_peer_close": (_i, [_p]),
    "hpfg_word_add": (_i, [_p, _i, _p]),
    "hpfg_peer_buf_slice": (_l, [_i, _l]),
    "hpfg_peer_buf_bytes": (_l, [_i, _l]),
    "hpfg_peer_allreduce_f32": (_i, [C.POINTER(PeerBuf), _p, _p]),
    "hpfg_bn_fwd_finalize_x": (_i, [_p, _i, C.POINTER(PeerX), _d, _p, _p, _p, _p, _f, _f, _p, _i, _p]),
    "hpfg_bn_bwd_finalize_x": (_i, [_p, _i, C.POINTER(PeerX), _d, _p, _p, _p, _p, _i, _f, _p]),
    "hpfg_seg_loss_partials_x": (_i, [C.POINTER(LossArgs), C.POINTER(PeerX), _p]),
    "hpfg_slab_reduce_multi": (_i, [_p, C.POINTER(SlabDesc), _i, _p]),
    "hpfg_wgrad_splits": (_i, [_i, _i, _i, _i, _i, _i]),
    "hpfg_wgrad_slab_floats": (_l, [_i, _i, _i, _i, _i, _i]),
    "hpfg_channel_sum": (_i, [_p, _i, _l, _i, _p, _p, _p]),
    "hpfg_channel_sum_partials": (_i, [_p, _i, _l, _i, _p, _p]),
    "hpfg_channel_sum_blocks": (_i, [_l, _i]),
    "hpfg_conv_first_rows": (_i, [_i, _i, _i]),
    "hpfg_confusion_counts": (_i, [_p, _p, _l, _i, _p, _p]),
    "hpfg_box_masks": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "hpfg_augment_batch": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p, _p]),
    "hpfg_timestamp": (_i, [_p, _p]),
    "hpfg_ln_fwd": (_i, [_p, _p, _p, _p, _p, _p, _l, _i, _p]),
    "hpfg_ln_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _l, _i, _p]),
    "hpfg_ln_bwd_blocks": (_i, [_l]),
    "hpfg_attn_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "hpfg_attn_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "hpfg_dwgelu_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "hpfg_dwgelu_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "hpfg_dwgelu_bwd_blocks": (_i, [_i, _i, _i]),
    "hpfg_resize_bilinear_fwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "hpfg_resize_sum_fwd": (_i, [_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), _i, _p, _i, _i, _i, _i, _p]),
    "hpfg_resize_bilinear_bwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "hpfg_residual_scale": (_i, [_p, _p, _p, _p, _i, _l, _p]),
    "hpfg_scale_rows": (_i, [_p, _p, _p, _i, _l, _p]),
    "hpfg_im2col_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "hpfg_col2im_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "hpfg_linear_wgrad": (_i, [_p, _p, _p, _p, _l, _i, _i, _p]),
    "hpfg_linear_wgrad_splits": (_i, [_l, _i, _i]),
    "hpfg_tok_col_stats": (_i, [_p, _l, _i, _p, _p, _p]),
    "hpfg_tok_stat_blocks": (_i, [_l]),
    "hpfg_bnrelu_apply": (_i, [_p, _p, _p, _p, _p, _p, _f, _l, _p, _l, _i, _p]),
    "hpfg_bnrelu_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _f, _l, _p, _p, _p, _l, _i, _p]),
    "hpfg_noise_add": (_i, [_p, _p, _p, _l, _l, _f, _f, _f, _p]),
    "hpfg_uncertainty_mask": (_i, [C.POINTER(PredBlocks), _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "hpfg_mix_samples": (_i, [_p, _p, _p, _p, _i, _l, _p]),
    "hpfg_softmax_mix": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "hpfg_pool_scatter_add": (_i, [C.POINTER(Act), _p, _i, _p, _i, _i, _i, _i, _p]),
    "hpfg_upsample2x_bwd": (_i, [_p, _i, _p, _i, _i, _i, _i, _p]),
    "hpfg_upsample2x_bwd_sums": (_i, [_p, _i, _p, _i, _i, _i, _i, _p, _p]),
    "hpfg_upsample2x_bwd_blocks": (_i, [_i, _i, _i, _i]),
    "hpfg_loss_blocks": (_i, [_i, _i, _i]),
    "hpfg_seg_loss_partials": (_i, [C.POINTER(LossArgs), _p]),
    "hpfg_seg_loss_finalize": (_i, [C.POINTER(LossArgs), _p]),
    "hpfg_seg_loss_bwd": (_i, [C.POINTER(LossArgs), _p, _p]),
    "hpfg_argmax_labels": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p]),
    "hpfg_cutmix_blend": (_i, [_p, _p, _p, _p, _l, _p]),
    "hpfg_sgd_step": (_i, [_p, _p, _p, _l, _p, _f, _f, _f, _p]),
    "hpfg_adamw_step": (_i, [_p, _p, _p, _p, _l, _p, _p, _f, _f, _f, _f, _f, _p]),
    "hpfg_ema_update": (_i, [_p, _p, _l, _p, _p]),
    "hpfg_sgd_ema_step": (_i, [_p, _p, _p, _l, _p, _f, _f, _f, _p, _l, _p, _p]),
    "hpfg_attn_mfma_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "hpfg_attn_mfma_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "hpfg_attn_mfma_blocks": (_i, [_i]),
    "hpfg_gemm_f32": (_i, [_p, _l, _l, _p, _l, _l, _p, _l, _i, _i, _i, _p, _i, _i, _p]),
    "hpfg_gemm_f32_splitk": (_i, [_p, _l, _l, _p, _l, _l, _p, _l, _i, _i, _i, _p, _i, _i, _p, _p]),
    "hpfg_gemm_f32_splits": (_i, [_i, _i, _i]),
    "hpfg_gemm_bf16x3": (_i, [_p, _l, _l, _p, _l, _l, _p, _l, _i, _i, _i, _p, _i, _i, _p]),
    "hpfg_gemm_bf16x3_splits": (_i, [_i, _i, _i]),
    "hpfg_gemm_bf16x3_splitk": (_i, [_p, _l, _l, _p, _l, _l, _p, _l, _i, _i, _i, _p, _i, _i, _p, _p]),
    "hpfg_gemm_bf16x3_ok": (_i, [_p, _l, _l, _p, _l, _l, _i, _i, _i]),
    "hpfg_gemm_tn_bf16x3": (_i, [_p, _p, _p, _p, _l, _i, _i, _i, _p]),
    "hpfg_gemm_tn_splits": (_i, [_l, _i, _i]),
    "hpfg_col_sum2": (_i, [_p, _l, _i, _l, _p, _p, _p]),
    "hpfg_col_sum_splits": (_i, [_l]),
    "hpfg_col_sum": (_i, [_p, _l, _i, _l, _p, _p]),
    "hpfg_relu_bwd": (_i, [_p, _p, _l, _p]),
    "hpfg_neck_pool_fwd": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "hpfg_neck_pool_bwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "hpfg_l2norm_fwd": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "hpfg_l2norm_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "hpfg_ntxent_rows": (_i, [_p, _i, _f, _p, _p, _p]),
}

_lock = threading.Lock()
_lib = None


def load() -> C.CDLL:
    """Load the shared library (once) and attach prototypes.  Raises HipLibraryError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C hpfg_amd/csrc`. There is no CPU fallback for the HIP path.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.hpfg_version() != VERSION:
            raise HipLibraryError(f"libhpfg_hip.so version {lib.hpfg_version()} != binding version {VERSION}; rebuild")
        if os.environ.get("HPFG_FIRST_WGRAD", "1") == "0":      # A/B runs: the first layer's weight gradient on the tile kernel
            lib.hpfg_set_option(OPT_FIRST_WGRAD, 0)
        if os.environ.get("HPFG_NARROW_DEEP") is not None:      # A/B runs: workgroup threshold below which a 3x3 launch takes 32-wide slices (0: never)
            lib.hpfg_set_option(OPT_NARROW_DEEP, int(os.environ["HPFG_NARROW_DEEP"]))
        _lib = lib
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().hpfg_last_error().decode("utf-8", "replace")
        raise HipLibraryError(f"{what} failed (rc={rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def none_act() -> Act:
    return Act()
