"""ctypes binding of libcwf_hip.so (include/cwf_hip.h).  The product path has NO fallback: if the
library is missing this raises, and every wrapper raises on a non-zero status."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcwf_hip.so")

P = C.c_void_p
I = C.c_int
L = C.c_int64
F = C.c_float
U = C.c_uint32
U64 = C.c_uint64

# name -> argtypes, exactly as declared in include/cwf_hip.h
SIGNATURES = {
    "cwf_version": [],
    "cwf_conv_mfma": [I, P, I, P, P, P, I, P, P, F, P, I, P, P, I, I, I, I, I, I, I, I, I, P],
    "cwf_conv_mfma_bf16": [I, I, P, I, P, P, P, I, P, P, F, P, I, P, P, I, I, I, I, I, I, I, I, I, P],
    "cwf_conv_mfma_bf16_nb": [I, I, P, I, P, P, P, I, P, P, F, P, I, P, P, P, I, P, P, F, I, I, I, I, I, I, I, I, I, P],
    "cwf_conv_s2c16_bf16": [I, P, I, P, P, P, I, P, I, I, I, I, P],
    "cwf_conv_mfma_bf16_y16": [I, I, P, I, P, P, P, I, P, P, P, F, P, I, P, I, I, I, I, I, I, I, I, I, P],
    "cwf_conv_stem_bf16": [I, P, I, P, P, P, I, P, P, I, I, I, I, P],
    "cwf_wgrad_mfma_bf16": [I, I, P, I, P, P, F, P, I, P, I, I, I, I, I, I, I, I, I, P, P],
    "cwf_gather_split_bf16": [P, I, L, P],
    "cwf_wgrad_nsplit": [I, I, I, I, I, I, I],
    "cwf_wgrad_partial_floats": [I, I, I, I, I, I, I],
    "cwf_wgrad_slab_floats": [I, I, I],
    "cwf_wgrad_mfma": [I, P, I, P, P, F, P, I, P, I, I, I, I, I, I, I, I, I, P],
    "cwf_wgrad_reduce": [P, I, L, P, P, P, P],
    "cwf_gather_batched": [P, I, L, P],
    "cwf_in_finalize": [P, P, P, I, L, F, P],
    "cwf_in_stats": [P, I, P, I, L, I, P],
    "cwf_norm_act_add": [P, I, P, P, F, P, I, P, I, I, L, I, P],
    "cwf_in_bwd_stats": [P, I, P, I, P, P, F, P, I, L, I, P],
    "cwf_in_bwd_apply": [P, I, P, I, P, P, F, P, P, I, P, I, I, L, I, P],
    "cwf_in_bwd_apply_ex": [P, I, P, I, P, P, F, P, P, I, P, I, P, P, I, L, I, P],
    "cwf_norm_act_add_ex": [P, I, P, P, F, P, I, P, I, P, I, L, I, P],
    "cwf_to_bf16": [P, I, P, P, F, P, I, L, I, P],
    "cwf_wgrad16_bf16": [P, P, P, P, I, I, I, I, P, P],
    "cwf_wgrad_mfma_bf16_dys": [I, I, P, I, P, P, F, P, I, P, P, I, I, I, I, I, I, I, I, I, P, P],
    "cwf_wgrad_s1_bf16": [P, P, P, P, I, I, I, I, I, I, P, P],
    "cwf_conv_mfma_bf16_in16": [I, P, P, P, P, P, I, P, I, P, P, I, P, P, F, I, I, I, I, P],
    "cwf_gemm": [P, L, L, L, L, P, L, L, L, L, P, L, L, L, P, P, L, L, L, I, I, I, I, I, F, I, I, P],
    "cwf_layernorm_fwd": [P, P, P, P, P, P, I, I, F, P],
    "cwf_layernorm_bwd": [P, P, P, P, P, P, P, P, I, I, I, P],
    "cwf_softmax_rows": [P, L, I, I, P],
    "cwf_softmax_rows_bwd": [P, P, L, I, I, P],
    "cwf_gelu_bwd": [P, P, P, L, P],
    "cwf_colsum": [P, L, I, I, P, I, P],
    "cwf_window_to_tokens": [P, I, P, I, I, I, I, I, I, I, I, P],
    "cwf_tokens_to_window": [P, P, I, I, I, I, I, I, I, I, I, I, P],
    "cwf_token_scores": [P, P, L, P, I, I, I, P],
    "cwf_topk": [P, P, I, I, I, P],
    "cwf_gather_tokens": [P, P, P, L, P, F, P, I, I, I, I, P],
    "cwf_gather_tokens_bwd": [P, P, P, P, P, L, I, I, I, I, P],
    "cwf_scatter_rows": [P, P, P, L, L, P, L, P, I, I, I, I, P],
    "cwf_scatter_rows_bwd": [P, P, P, P, L, P, I, P, L, L, P, L, I, I, I, I, P],
    "cwf_upsample_softmax": [P, I, P, I, I, I, I, I, I, P],
    "cwf_upsample_softmax_bwd": [P, P, P, I, I, I, I, I, I, I, P, P],
    "cwf_channel_softmax": [P, I, P, L, I, P],
    "cwf_channel_softmax_bwd": [P, P, P, I, L, I, P],
    "cwf_dice_ce_sums": [P, P, U, P, I, L, I, P],
    "cwf_dice_ce_finalize": [P, P, P, I, L, I, P],
    "cwf_dice_ce_bwd": [P, P, U, P, P, P, I, L, I, P],
    "cwf_adam_amsgrad": [P, I, L, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, I, I, P, P],
    "cwf_adam_amsgrad_scaled": [P, I, L, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, I, I, P, F, P],
    "cwf_wgrad_reduce_batched": [P, I, P],
    "cwf_dropout_mask": [P, L, F, F, C.c_uint64, C.c_uint64, P],
    "cwf_mul": [P, P, P, L, P],
    "cwf_add": [P, P, P, L, P],
    "cwf_channel_scale": [P, I, P, P, I, I, L, I, P],
    "cwf_copy_strided": [P, I, P, I, L, I, P],
    "cwf_add3": [P, P, P, P, L, P],
    "cwf_bcast3": [P, P, L, P],
    "cwf_stats_channel_sum": [P, P, I, I, P],
    "cwf_gemm_ex": [P, P],
    "cwf_attn_fwd": [P, L, P, L, I, I, I, I, F, P, U64, F, P],
    "cwf_attn_bwd": [P, L, P, L, P, I, I, I, I, F, P, U64, F, P],
    "cwf_ln_pair_fwd": [P, P, I, P, P, P, P, P, P, P, I, I, F, P],
    "cwf_ln_pair_bwd": [P, P, P, P, P, I, P, P, P, P, P, P, P, P, P, I, I, I, P],
    "cwf_gelu_bwd_drop": [P, P, P, L, P, U64, F, P],
    "cwf_token_scores2": [P, P, L, P, L, P, P, I, I, I, P],
    "cwf_topk_inv": [P, P, P, P, P, P, I, I, I, P],
    "cwf_index_inv": [P, P, I, I, I, P],
    "cwf_gather_multi": [P, I, I, I, I, F, P, F, P],
    "cwf_scatter_inv": [P, P, P, L, L, P, L, P, P, I, I, I, P],
    "cwf_scatter_bwd": [P, P, P, P, P, P, L, L, P, L, P, L, P, L, L, P, L, I, I, I, I, P],
    "cwf_token_grad": [P, P, P, L, P, P, P, L, P, L, P, U64, U64, F, P, I, I, I, I, P],
    "cwf_head_grad": [P, P, P, P, L, P, P, I, I, P],
    "cwf_dice_ce_finalize_multi": [P, P, P, P, I, I, L, I, P],
    "cwf_head_loss_sums": [P, I, I, P, P, P, I, I, I, I, I, P],
    "cwf_head_loss_bwd": [P, I, I, P, P, P, P, P, I, P, I, I, I, I, I, P],
    "cwf_head_loss_bwd_ex": [P, I, I, P, P, P, P, P, I, I, P, I, I, I, I, I, P],
    "cwf_wgrad_mfma_bf16_grouped": [I, I, P, I, P, I, P, I, I, I, I, I, I, I, I, I, I, P, P],
    "cwf_conv_mfma_bf16_grouped": [I, I, P, I, I, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P],
    "cwf_ln_pair_fwd_g": [P, P, I, P, I, P, P, P, I, I, F, P],
    "cwf_ln_pair_bwd_g": [P, P, P, P, P, I, P, I, P, P, P, I, I, I, P],
    "cwf_token_scores2_g": [P, P, P, I, I, P, P, I, I, I, P],
    "cwf_head_grad_g": [P, P, P, P, L, P, P, I, I, I, P],
    "cwf_window_to_tokens_g": [P, I, P, I, I, I, I, I, I, I, I, I, P],
    "cwf_tokens_to_window_g": [P, P, I, I, I, I, I, I, I, I, I, I, P],
    "cwf_cat3_channels": [P, P, P, P, L, I, P],
    "cwf_stitch_windows": [P, P, I, P],
    "cwf_argmax_dice": [P, L, L, L, P, P, P, I, L, P],
    "cwf_argmax_metrics": [P, L, L, L, P, P, P, I, L, P],
    "cwf_rng_advance": [P, P],
    "cwf_dropout_mask_rng": [P, L, F, F, P, U64, P],
    "cwf_plan_create": [P, P],
    "cwf_plan_info": [P, P],
    "cwf_plan_run": [P, P, P, I, P, P],
    "cwf_plan_destroy": [P],
    "cwf_plan_marker": [I, P],
}


class GemmArgs(C.Structure):
    """struct cwf_gemm_args (include/cwf_hip.h)"""
    _fields_ = [("A", P), ("sa_m", L), ("sa_k", L), ("sa_zb", L), ("sa_zh", L),
                ("B", P), ("sb_k", L), ("sb_n", L), ("sb_zb", L), ("sb_zh", L),
                ("C", P), ("sc_m", L), ("sc_zb", L), ("sc_zh", L),
                ("bias", P), ("residual", P), ("sr_m", L), ("sr_zb", L), ("sr_zh", L),
                ("M", I), ("N", I), ("K", I), ("ZB", I), ("ZH", I), ("alpha", F), ("act", I), ("accumulate", I),
                ("A2", P), ("split_n", I), ("B2", P), ("split_m", I), ("C2", P), ("rowsum", P), ("rowsum_acc", I), ("rng", P),
                ("a_drop_off", C.c_uint64), ("a_drop_n", C.c_uint64), ("a_drop_p", F), ("a_drop_p2", F),
                ("c_drop_off", C.c_uint64), ("c_drop_n", C.c_uint64), ("c_drop_p", F), ("c_drop_p2", F),
                ("B_tab", P * 4), ("bias_tab", P * 4), ("C_tab", P * 4), ("rowsum_tab", P * 4)]


class LnGroupParams(C.Structure):
    """struct cwf_ln_group_params (include/cwf_hip.h)"""
    _fields_ = [(n, P * 4) for n in ("g1", "b1", "g2", "b2", "dg1", "db1", "dg2", "db2")]


class GatherJob(C.Structure):
    """struct cwf_gather_job (include/cwf_hip.h)"""
    _fields_ = [("feats", P), ("index", P), ("head", P), ("out", P), ("head_bstride", L), ("out_bstride", L), ("T", I),
                ("drop_off", C.c_uint64), ("head_g", P * 4), ("group_B", I)]
RESTYPE_INT64 = {"cwf_wgrad_partial_floats", "cwf_wgrad_slab_floats"}

_lib = None


class CwfError(RuntimeError):
    pass


def load():
    """dlopen the in-tree library and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CwfError("libcwf_hip.so is not built (%s); run `python __graft_entry__.py` or `make -C csrc`. "
                       "There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = L if name in RESTYPE_INT64 else I
    lib.cwf_arch.restype = C.c_char_p
    lib.cwf_arch.argtypes = []
    lib.cwf_plan_last_error.restype = C.c_char_p
    lib.cwf_plan_last_error.argtypes = []
    _lib = lib
    return lib


def check(rc, name):
    if rc != 0:
        raise CwfError("%s failed with status %d" % (name, rc))
