"""ctypes binding of libspider_hip.so (C ABI declared in include/spider_hip.h).

The product path has no CPU or PyTorch fallback: if the shared library is missing or a call fails, this
module raises. Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C spider_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

# torch must be imported (and with it the HIP runtime it bundles, in the global symbol scope) BEFORE the
# kernel library is dlopen'ed: both then talk to ONE HIP runtime, so torch's streams, graphs and device
# pointers are valid inside libspider_hip.so.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspider_hip.so")

_lib = None
_lock = threading.Lock()

_vp, _i, _f, _l = C.c_void_p, C.c_int, C.c_float, C.c_long

# name -> (restype, argtypes); mirrors include/spider_hip.h one-to-one
SIGNATURES = {
    "spider_abi_version": (_i, []),
    "spider_target_arch": (C.c_char_p, []),
    "spider_last_error": (C.c_char_p, []),
    "spider_embed_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "spider_rmsnorm_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "spider_gemv_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _vp]),
    "spider_gemv_swiglu_bf16": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _i, _vp]),
    "spider_lm_head_nparts": (_i, [_i]),
    "spider_decode_advance_i32": (_i, [_vp] * 7 + [_i, _i, _vp]),
    "spider_lm_head_argmax_bf16": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "spider_gemv_fm_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "spider_gemv_swiglu_fm_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "spider_lm_head_argmax_fm_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "spider_rope_kv_append_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "spider_rope_kv_append_mrope_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "spider_attn_decode_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "spider_attn_decode_fused_bf16": (_i, [_vp] * 11 + [_i, _i, _i, _i, _i, _f, _i, _vp]),
    "spider_set_attn_inline": (_i, [_i]),
    "spider_set_ws_inlaunch": (_i, [_i]),
    "spider_gemm_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _l, _vp]),
    "spider_gemm_ln_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _l, _vp]),
    "spider_xattn_fused_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "spider_conv2d_nhwc_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _l, _vp]),
    "spider_conv_nhwc_ex_bf16": (_i, [_vp] * 6 + [_i] * 14 + [_f, _f, _i, _vp, _vp, _vp, _l, _vp]),
    "spider_attn_bf16": (_i, [_vp, _vp, _vp, _vp] + [_l] * 12 + [_i] * 6 + [_f, _i, _i, _vp, _vp, _i, _i, _vp]),
    "spider_story_key_lists_i32": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "spider_attn_keylist_bf16": (_i, [_vp, _vp, _vp, _vp] + [_l] * 12 + [_i] * 6 + [_f, _vp, _i, _vp, _i, _vp]),
    "spider_attn_varlen_bf16": (_i, [_vp, _vp, _vp, _vp, _l, _l, _l, _l, _i, _i, _i, _i, _f, _vp, _i, _vp]),
    "spider_rope_rows_bf16": (_i, [_vp, _vp, _l, _i, _i, _i, _vp]),
    "spider_groupnorm_nchunk": (_i, [_i]),
    "spider_groupnorm_nhwc_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "spider_groupnorm_cat_nhwc_bf16": (_i, [_vp] * 7 + [_i, _i, _i, _i, _i, _f, _i, _vp]),
    "spider_conv_nhwc_gn_bf16": (_i, [_vp] * 6 + [_i] * 14 + [_f, _f, _i, _vp, _vp, _vp, _l, _vp] + [_vp, _i, _vp]),
    "spider_groupnorm_stats_nhwc_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "spider_groupnorm_apply_nhwc_bf16": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "spider_gemm_gn_in_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _f, _i, _vp, _vp]),
    "spider_layernorm_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "spider_geglu_bf16": (_i, [_vp, _vp, _i, _i, _vp]),
    "spider_swiglu_bf16": (_i, [_vp, _vp, _i, _i, _vp]),
    "spider_concat_channels_bf16": (_i, [_vp, _vp, _vp, _l, _i, _i, _vp]),
    "spider_act_bf16": (_i, [_vp, _vp, _l, _i, _vp]),
    "spider_add_bf16": (_i, [_vp, _vp, _vp, _l, _vp]),
    "spider_act_ex_bf16": (_i, [_vp, _vp, _l, _i, _f, _vp]),
    "spider_add_scaled_bf16": (_i, [_vp, _vp, _vp, _l, _f, _vp]),
    "spider_axpby_bf16": (_i, [_vp, _vp, _vp, _l, _f, _f, _vp]),
    "spider_mean_tokens_bf16": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "spider_moe_combine_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _l, _vp]),
    "spider_col2im1d_f32_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "spider_l2_normalize_rows_bf16": (_i, [_vp, _vp, _i, _i, _f, _vp]),
    "spider_conv2d_small_cin_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "spider_conv2d_small_cout_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "spider_latent_to_nhwc_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "spider_cfg_combine_f32": (_i, [_vp, _vp, _i, _i, _i, _f, _vp]),
    "spider_lincomb_f32": (_i, [_vp, _vp, _i, _vp, _l, _vp]),
    "spider_softmax_rows_f32_bf16": (_i, [_vp, _vp, _i, _i, _i, _f, _vp]),
    "spider_pack_keep_bits_f32": (_i, [_vp, _vp, _i, _i, _f, _vp]),
    "spider_nhwc_to_nchw_f32": (_i, [_vp, _vp, _i, _i, _i, _f, _f, _i, _vp]),
    # "precise" fp32-operand forms (ABI v4)
    "spider_gemm_a32_bf16": (_i, [_vp] * 5 + [_i] * 5 + [_f, _i, _vp, _vp, _vp, _l, _vp]),
    "spider_gemm_ln_a32_bf16": (_i, [_vp] * 6 + [_i, _i, _i, _i, _i, _f, _i, _vp]),
    "spider_gemm_gn_in_a32_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _f, _i, _vp, _vp]),
    "spider_conv_nhwc_a32_bf16": (_i, [_vp] * 6 + [_i] * 13 + [_f, _i, _vp, _vp, _vp, _l, _vp]),
    "spider_groupnorm_f32in_nhwc_bf16": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "spider_split_hilo_f32_bf16": (_i, [_vp, _vp, _vp, _l, _vp]),
    "spider_row_split_f32_bf16": (_i, [_vp, _vp, _vp, _vp, _l, _i, _f, _vp]),
    "spider_groupnorm_f32in_split_nhwc_bf16": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "spider_conv2d_small_cin_f32in_bf16": (_i, [_vp] * 5 + [_i] * 6 + [_vp]),
    "spider_conv2d_small_cout_f32in_bf16": (_i, [_vp] * 4 + [_i] * 6 + [_vp]),
    "spider_latent_to_nhwc_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
}

# the diffusion-side operators (gemm / conv / attention / UNet ops / fused cross-attention translation units) also exist as
# IEEE-half instantiations with the same signatures: spider_<op>_f16 (include/spider_hip.h, last section)
F16_OPS = (
    "spider_gemm", "spider_gemm_ln", "spider_xattn_fused", "spider_conv2d_nhwc", "spider_conv_nhwc_ex", "spider_attn",
    "spider_conv_nhwc_gn", "spider_groupnorm_stats_nhwc", "spider_groupnorm_apply_nhwc", "spider_gemm_gn_in",
    "spider_attn_keylist", "spider_attn_varlen", "spider_groupnorm_nhwc", "spider_groupnorm_cat_nhwc", "spider_layernorm",
    "spider_geglu", "spider_swiglu", "spider_concat_channels", "spider_act", "spider_add", "spider_act_ex", "spider_add_scaled",
    "spider_axpby", "spider_mean_tokens", "spider_moe_combine", "spider_col2im1d_f32", "spider_l2_normalize_rows",
    "spider_conv2d_small_cin", "spider_conv2d_small_cout", "spider_latent_to_nhwc", "spider_softmax_rows_f32",
    "spider_gemm_a32", "spider_gemm_ln_a32", "spider_gemm_gn_in_a32", "spider_conv_nhwc_a32", "spider_groupnorm_f32in_nhwc", "spider_conv2d_small_cin_f32in",
    "spider_conv2d_small_cout_f32in", "spider_split_hilo_f32", "spider_row_split_f32", "spider_groupnorm_f32in_split_nhwc",
)
for _n in F16_OPS:
    SIGNATURES[_n + "_f16"] = SIGNATURES[_n + "_bf16"]


class SpiderHipError(RuntimeError):
    pass


def load():
    """Load libspider_hip.so (once) and attach the prototypes. Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise SpiderHipError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path."
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if lib.spider_abi_version() != 4:
            raise SpiderHipError("libspider_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def call(name: str, *args) -> None:
    """Invoke an int-returning entry point; raise SpiderHipError with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.spider_last_error().decode("utf-8", "replace")
        raise SpiderHipError(f"{name} failed (rc={rc}): {msg}")
