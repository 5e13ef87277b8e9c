"""ctypes binding of ``libldit_hip.so`` (C ABI: ``include/ldit.h``).

The library is the product; this module only loads it and describes its signatures.  There is deliberately no
fallback: if the shared object is missing or a symbol is absent, import of the compute path fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LDIT_LIB_PATH (diagnostic): load another BUILD of this same library - an A/B variant compiled with extra -D flags by
# scripts/build_alt.sh - instead of the shipped one.  Still the HIP library or nothing: there is no other implementation.
LIB_PATH = os.environ.get("LDIT_LIB_PATH") or os.path.join(_HERE, "libldit_hip.so")

LDIT_ABI_VERSION = 5
LDIT_MAX_TAPS = 8
DTYPE_F32, DTYPE_BF16, DTYPE_FP8, DTYPE_F32X3, DTYPE_F32X6 = 0, 1, 3, 4, 5
FP8_A_COUNT = 4
LDIT_OK, LDIT_EINVAL, LDIT_EWORKSPACE, LDIT_EHIP, LDIT_EUNSUPPORTED = 0, -1, -2, -3, -4
EPI_BIAS, EPI_BIAS_GELU, EPI_SCALE_RESID, EPI_F32, EPI_GELU_BWD = 0, 1, 2, 4, 5
K_GEMM, K_ATTENTION, K_LAYERNORM, K_OTHER, K_COUNT = 0, 1, 2, 3, 4
KERNEL_FAMILIES = ("gemm", "attention", "layernorm", "other")


class LditCfg(C.Structure):
    _fields_ = [("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32), ("mlp", C.c_int32),
                ("patch", C.c_int32), ("in_ch", C.c_int32), ("img_h", C.c_int32), ("img_w", C.c_int32),
                ("n_taps", C.c_int32), ("taps", C.c_int32 * LDIT_MAX_TAPS), ("ln_eps", C.c_float),
                ("dtype", C.c_int32), ("flags", C.c_int32)]


LAYER_FIELDS = ("ln1_w", "ln1_b", "wq", "bq", "wk", "wv", "bv", "wo", "bo", "lam1",
                "ln2_w", "ln2_b", "w1", "b1", "w2", "b2", "lam2")


class LditLayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in LAYER_FIELDS]


class LditWeights(C.Structure):
    _fields_ = [("patch_w", C.c_void_p), ("patch_b", C.c_void_p), ("cls", C.c_void_p), ("pos", C.c_void_p),
                ("layer", C.POINTER(LditLayerWeights))]


# name -> (restype, argtypes); every symbol include/ldit.h declares
_vp, _i64, _i32, _f32, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_size_t
SIGNATURES = {
    "ldit_abi_version": (C.c_int, []),
    "ldit_last_error": (C.c_char_p, []),
    "ldit_debug_reload_env": (C.c_int, []),
    "ldit_packed_bytes": (_sz, [C.POINTER(LditCfg)]),
    "ldit_pack_weights": (C.c_int, [C.POINTER(LditCfg), C.POINTER(LditWeights), _vp, _sz, _vp]),
    "ldit_workspace_bytes": (_sz, [C.POINTER(LditCfg), _i32]),
    "ldit_vit_forward": (C.c_int, [C.POINTER(LditCfg), _vp, _vp, _i32, C.POINTER(_vp), _vp, _sz, _vp]),
    "ldit_vit_forward_images": (C.c_int, [C.POINTER(LditCfg), _vp, C.POINTER(_vp), C.POINTER(_i32), C.POINTER(_i32), _i32, _f32, _f32, _i32,
                                          C.POINTER(_vp), _vp, _sz, _vp]),
    "ldit_vit_forward_timed": (C.c_int, [C.POINTER(LditCfg), _vp, _vp, _i32, C.POINTER(_vp), _vp, _sz, _vp,
                                         C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "ldit_linear_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp]),
    "ldit_layernorm_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _f32, _vp]),
    "ldit_attention_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _vp]),
    "ldit_embed_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    "ldit_attention_planes": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp]),
    "ldit_split_f32_planes": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _vp]),
    "ldit_layernorm_f32_planes": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, C.c_float, _i32, _vp]),
    "ldit_linear_planes": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _i32, _vp]),
    "ldit_embed_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    "ldit_embed_bf16_images": (C.c_int, [C.POINTER(_vp), C.POINTER(_i32), C.POINTER(_i32), _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp,
                                         _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    "ldit_tap_to_map_f32": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _f32, _vp]),
    "ldit_tap_to_map_bwd_f32": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _f32, _vp]),
    "ldit_linear_bf16": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp]),
    "ldit_attention_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _vp]),
    "ldit_cast_f32_bf16": (C.c_int, [_vp, _vp, _i64, _vp]),
    "ldit_linear_fp8": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _f32, _f32, _vp, _vp]),
    "ldit_quant_rows_f32_fp8": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "ldit_set_fp8_act_scales": (C.c_int, [C.POINTER(LditCfg), _vp, C.c_size_t, C.POINTER(C.c_float), _vp]),
    "ldit_quant_f32_fp8": (C.c_int, [_vp, _vp, _i64, _f32, _vp]),
    "ldit_amax_f32": (C.c_int, [_vp, _i64, _vp, _vp]),
    "ldit_preprocess_f32": (C.c_int, [C.POINTER(_vp), C.POINTER(_i32), C.POINTER(_i32), _i32, _i32, _f32, _f32, _i32, _i32,
                                      _vp, _vp]),
    "ldit_preprocess_f16": (C.c_int, [C.POINTER(_vp), C.POINTER(_i32), C.POINTER(_i32), _i32, _i32, _f32, _f32, _i32, _i32,
                                      _vp, _vp]),
    "ldit_cast_f16_f32": (C.c_int, [_vp, _vp, _i64, _vp]),
    "ldit_cast_f32_f16": (C.c_int, [_vp, _vp, _i64, _vp]),
    "ldit_fpn_merge_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _i64, _i64, _vp]),
    "ldit_conv3x3_nhwc_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp]),
    "ldit_fpn_merge_bwd_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _i64, _i64, _vp]),
    "ldit_pad_nhwc_f32_bf16": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "ldit_colsum_scratch_bytes": (_sz, [_i64, _i64]),
    "ldit_colsum_f32": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "ldit_colamax_f32": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    # train step
    "ldit_flat_param_bytes": (_sz, [C.POINTER(LditCfg)]),
    "ldit_flat_param_layout": (C.c_int, [C.POINTER(LditCfg), C.POINTER(_i64), _i32]),
    "ldit_train_saved_bytes": (_sz, [C.POINTER(LditCfg), _i32]),
    "ldit_train_workspace_bytes": (_sz, [C.POINTER(LditCfg), _i32]),
    "ldit_train_mirror_bytes": (_sz, [C.POINTER(LditCfg)]),
    "ldit_pack_train": (C.c_int, [C.POINTER(LditCfg), _vp, _vp, _sz, _vp]),
    "ldit_vit_forward_train": (C.c_int, [C.POINTER(LditCfg), _vp, _vp, _vp, _i32, C.POINTER(_vp), _vp, _vp, _sz, _vp,
                                         C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "ldit_vit_backward": (C.c_int, [C.POINTER(LditCfg), _vp, _vp, _vp, _i32, C.POINTER(_vp), _vp, _vp, _sz, _vp, _sz, _vp, _sz,
                                    _i32, _i32, _vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "ldit_adamw_step": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _vp, _vp]),
    "ldit_attention_fwd_lse_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _vp]),
    "ldit_attention_bwd_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                          _i64, _f32, _vp]),
    "ldit_layernorm_bwd_scratch_bytes": (_sz, [_i64, _i64]),
    "ldit_layernorm_bwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _f32, _vp, _vp, _vp, _sz, _vp]),
    "ldit_linear_bf16_ex": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                      _i32, _vp]),
    "ldit_reduce_slabs_f32": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "ldit_linear_bf16_tr": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _i32, _vp, _vp]),
}

_lib = None


class LditError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libldit_hip: error {code}: {message}")
        self.code = code


def load() -> C.CDLL:
    """Load the library (once).  Raises if it is missing - there is no other implementation to fall back to."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build it with `make -C layoutdit_amd/csrc` "
                              f"(or __graft_entry__.build()); layoutdit_amd has no CPU / eager fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        got = lib.ldit_abi_version()
        if got != LDIT_ABI_VERSION:
            raise ImportError(f"libldit_hip ABI {got} != expected {LDIT_ABI_VERSION}")
        _lib = lib
    return _lib


def set_switch(name: str, value) -> None:
    """Set (``value`` a string) or remove (``None``) one of the library's diagnostic ``LDIT_*`` environment switches and make
    the library re-read them: it reads its switches once, not per launch (``ldit_debug_reload_env``, include/ldit.h)."""
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)
    load().ldit_debug_reload_env()


def check(rc: int) -> None:
    if rc != LDIT_OK:
        raise LditError(rc, load().ldit_last_error().decode(errors="replace"))
