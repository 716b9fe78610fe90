"""ctypes binding of libmpv_hip.so (the C ABI declared in include/mpv.h).

The product path has NO CPU / PyTorch fallback: if the library is missing or a call fails
this module raises.  Build with `python __graft_entry__.py` (or `make -C youku-mplug_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPV_LIB_PATH") or os.path.join(_HERE, "libmpv_hip.so")   # override: instrumented measurement builds only

c_void_p, c_int, c_int64, c_float, c_uint64, c_size_t = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64, C.c_size_t


class GemmEpilogue(C.Structure):
    _fields_ = [
        ("a_group", c_int), ("a_stride", c_int), ("a_offset", c_int),
        ("c_group", c_int), ("c_stride", c_int), ("c_offset", c_int),
        ("k_group", c_int), ("k_stride", c_int), ("k_offset", c_int),
        ("bias", c_void_p), ("act", c_int), ("preact_out", c_void_p),
        ("residual", c_void_p), ("ldr", c_int64),
        ("act_bwd_z", c_void_p), ("ldz", c_int64), ("act_bwd", c_int),
        ("dropout_p", c_float), ("seed", c_uint64), ("offset", c_uint64),
        ("alpha_dev", c_void_p), ("alpha", c_float), ("out_f32", c_int), ("accumulate", c_int),
        ("colsum_out", c_void_p), ("tile_hint", c_int), ("row_tap_out", c_void_p), ("row_tap_group", c_int), ("split_hint", c_int), ("gm_hint", c_int), ("preact_deriv", c_int), ("keep_output", c_int),
        ("colscale", c_float), ("colscale_cols", c_int),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("o", c_void_p), ("lse", c_void_p),
        ("q_bs", c_int64), ("q_hs", c_int64), ("q_rs", c_int64),
        ("k_bs", c_int64), ("k_hs", c_int64), ("k_rs", c_int64),
        ("v_bs", c_int64), ("v_hs", c_int64), ("v_rs", c_int64),
        ("o_bs", c_int64), ("o_hs", c_int64), ("o_rs", c_int64),
        ("batch", c_int), ("heads", c_int), ("sq", c_int), ("sk", c_int), ("head_dim", c_int),
        ("causal", c_int), ("scale", c_float), ("scale_q_bf16", c_int),
        ("dropout_p", c_float), ("seed", c_uint64), ("offset", c_uint64),
    ]


_RM = [c_int, c_int, c_int]
_SIGS = {
    "mpv_version": (c_int, []),
    "mpv_last_error": (C.c_char_p, []),
    "mpv_check_device": (c_int, []),
    "mpv_check_arch_name": (c_int, [C.c_char_p]),
    "mpv_gemm_workspace_size": (c_size_t, [c_int64, c_int64, c_int64, c_int, c_int]),
    "mpv_gemm_plan_bands": (c_int, [c_int64, c_int64, c_int64, c_int, c_int, c_int, C.POINTER(c_int)]),
    "mpv_gemm_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64,
                              c_int, c_int, C.POINTER(GemmEpilogue), c_void_p, c_size_t, c_void_p]),
    "mpv_layernorm_fwd": (c_int, [c_void_p] * 6 + [c_int64] * 4 + [c_float] + _RM + _RM + [c_void_p]),
    "mpv_layernorm_bwd_workspace_size": (c_size_t, [c_int64]),
    "mpv_layernorm_bwd": (c_int, [c_void_p] * 8 + [c_float, c_uint64, c_uint64, c_void_p, c_void_p, c_int] +
                          [c_int64] * 4 + _RM + _RM + [c_void_p, c_size_t, c_void_p]),
    "mpv_ln_stream_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int64] * 5 +
                          [c_float] + _RM + _RM + _RM + [c_void_p]),
    "mpv_ln_stream_fwd_drop": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int64] * 5 +
                               [c_float] + _RM + _RM + _RM + [c_float, c_uint64, c_uint64, c_void_p]),
    "mpv_ln_stream_bwd": (c_int, [c_void_p] * 8 + [c_float, c_uint64, c_uint64] + [c_int64] * 4 + _RM + _RM + [c_void_p]),
    "mpv_attn_fwd": (c_int, [C.POINTER(AttnDesc), c_void_p]),
    "mpv_attn_bwd": (c_int, [C.POINTER(AttnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mpv_temporal_attn_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int64, c_int64, c_int, c_int, c_int,
                                      c_float, c_void_p]),
    "mpv_temporal_attn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int64, c_int64, c_int,
                                      c_int, c_int, c_float, c_void_p]),
    "mpv_im2col_patches": (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "mpv_vit_embed_assemble_fwd": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "mpv_vit_embed_assemble_bwd": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "mpv_vit_cls_merge_fwd": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "mpv_vit_cls_merge_bwd": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_void_p]),
    "mpv_vit_cls_fix_fwd": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "mpv_vit_cls_merge_bwd_inplace": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_void_p]),
    "mpv_copy_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64] + _RM + _RM + [c_void_p]),
    "mpv_colsum_workspace_size": (c_size_t, [c_int64]),
    "mpv_colsum": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64] + _RM + [c_int, c_void_p, c_size_t, c_void_p]),
    "mpv_add": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mpv_accum_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mpv_store_words": (c_int, [c_void_p, C.POINTER(C.c_uint32), c_int, c_void_p]),
    "mpv_f32_to_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "mpv_copy_segments": (c_int, [C.POINTER(c_void_p), C.POINTER(c_void_p), C.POINTER(c_int64), c_int, c_void_p]),
    "mpv_vit_compose_bwd_finish": (c_int, [c_void_p] * 6 + [c_int, c_void_p]),
    "mpv_vit_compose_bias_batched": (c_int, [C.POINTER(c_void_p)] * 4 + [c_int, c_int, c_void_p]),
    "mpv_vit_compose_bwd_finish_batched": (c_int, [C.POINTER(c_void_p)] * 6 + [c_int, c_int, c_void_p]),
    "mpv_gemm_bf16_batched": (c_int, [C.POINTER(c_void_p)] * 3 + [c_int] + [c_int64] * 6 + [c_int, c_int, c_void_p]),
    "mpv_caption_targets": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mpv_layernorm_bwd_partial_rows": (c_int, [c_int64]),
    "mpv_layernorm_dparam_finish": (c_int, [C.POINTER(c_void_p), C.POINTER(c_int), C.POINTER(c_void_p), C.POINTER(c_void_p),
                                            C.POINTER(c_int), c_int, c_int64, c_void_p]),
    "mpv_gpt_embed_fwd": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_float, c_uint64, c_uint64, c_void_p]),
    "mpv_gpt_embed_bwd": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_float, c_uint64, c_uint64, c_void_p]),
    "mpv_gpt_embed_bwd_full": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_uint64, c_uint64, c_void_p]),
    "mpv_cross_entropy": (c_int, [c_void_p] * 6 + [c_int64] * 3 + [c_void_p]),
    "mpv_l2norm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float, c_void_p]),
    "mpv_l2norm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "mpv_gather_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "mpv_video_resized_crop_normalize": (c_int, [c_void_p] + [c_int] * 11 + [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "mpv_video_resized_crop_u8": (c_int, [c_void_p] + [c_int] * 11 + [c_void_p, c_void_p]),
    "mpv_video_aug_pointwise": (c_int, [c_void_p, c_int, c_int, c_int, c_int, C.c_double, c_void_p, c_void_p]),
    "mpv_video_aug_sharpness": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, C.c_double, c_void_p]),
    "mpv_video_aug_warp_affine": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint8), c_void_p]),
    "mpv_video_u8_normalize": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "mpv_scatter_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "mpv_gather_rows_ld": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "mpv_logprob_topk_workspace_size": (c_size_t, [c_int64, c_int]),
    "mpv_logprob_topk": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mpv_soft_target_ce": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "mpv_grad_sumsq_workspace_size": (c_size_t, []),
    "mpv_grad_sumsq": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mpv_adamw_step": (c_int, [c_void_p] * 5 + [c_int64] + [c_float] * 5 + [c_int, c_float, c_void_p, c_float, c_void_p]),
    "mpv_adamw_hyper_pack": (c_int, [C.POINTER(c_float), C.POINTER(c_float), c_int, c_float, c_float, c_int, C.POINTER(c_float)]),
    "mpv_adamw_step_grouped_dev": (c_int, [c_void_p] * 5 + [c_int64, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_void_p,
                                           c_float, c_void_p]),
    "mpv_adamw_step_grouped": (c_int, [c_void_p] * 5 + [c_int64, c_void_p, C.POINTER(c_float), C.POINTER(c_float), c_int,
                                       c_float, c_float, c_float, c_int, c_float, c_void_p, c_float, c_void_p]),
}
class GptLayerWeights(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "dense_w", "dense_b", "ln2_w", "ln2_b", "fc1_w", "fc1_b",
                                        "fc2_w", "fc2_b")]


class GptWeights(C.Structure):
    _fields_ = [("layers", c_int), ("hidden", c_int), ("heads", c_int), ("ffn", c_int), ("vocab", c_int), ("ln_eps", c_float),
                ("layer", C.POINTER(GptLayerWeights)), ("wte", c_void_p), ("wpe", c_void_p), ("lnf_w", c_void_p), ("lnf_b", c_void_p),
                ("max_positions", c_int)]


_SIGS.update({
    "mpv_gpt_decode_workspace_size": (c_size_t, [C.POINTER(GptWeights), c_int, c_int]),
    "mpv_gpt_decode_step": (c_int, [C.POINTER(GptWeights), C.POINTER(c_void_p), c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                                    c_void_p, c_size_t, c_void_p, c_void_p]),
    "mpv_kv_reorder": (c_int, [C.POINTER(c_void_p), C.POINTER(c_void_p), c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
})

EXPORTS = tuple(_SIGS)

_lib = None


class MpvError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise MpvError(f"{LIB_PATH} not found: the HIP kernels are not built. "
                           "Run `python __graft_entry__.py` (build()) first; there is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)      # AttributeError if a declared symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


n_calls = 0      # C-ABI calls checked so far (engine.graph_step: "has this graph segment captured anything yet?")


def check(rc: int, what: str = ""):
    global n_calls
    n_calls += 1
    if rc != 0:
        raise MpvError(f"{what} failed ({rc}): {lib().mpv_last_error().decode()}")
