"""ctypes front end of the CPU oracle (``oracle/vit_oracle.c``).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this module; nothing
under ``layoutdit_amd/`` does.  The shared objects are built by ``oracle/Makefile`` (``__graft_entry__.build()``
runs it); if one is missing it is compiled on first use with the same recipe (gcc is in the image).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS: Dict[str, C.CDLL] = {}


class OracleCfg(C.Structure):
    _fields_ = [("C", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("F", C.c_int32), ("patch", C.c_int32),
                ("in_ch", C.c_int32), ("n_taps", C.c_int32), ("taps", C.c_int32 * 8), ("ln_eps", C.c_float)]


_LAYER_FIELDS = ["ln1_w", "ln1_b", "wq", "bq", "wk", "wv", "bv", "wo", "bo", "lam1",
                 "ln2_w", "ln2_b", "w1", "b1", "w2", "b2", "lam2"]

# transformers-4.49 BEiT parameter names -> oracle_layer fields
_LAYER_KEYS = {
    "ln1_w": "layernorm_before.weight", "ln1_b": "layernorm_before.bias",
    "wq": "attention.attention.query.weight", "bq": "attention.attention.query.bias",
    "wk": "attention.attention.key.weight",
    "wv": "attention.attention.value.weight", "bv": "attention.attention.value.bias",
    "wo": "attention.output.dense.weight", "bo": "attention.output.dense.bias", "lam1": "lambda_1",
    "ln2_w": "layernorm_after.weight", "ln2_b": "layernorm_after.bias",
    "w1": "intermediate.dense.weight", "b1": "intermediate.dense.bias",
    "w2": "output.dense.weight", "b2": "output.dense.bias", "lam2": "lambda_2",
}


class OracleLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _LAYER_FIELDS]


class OracleWeights(C.Structure):
    _fields_ = [("patch_w", C.c_void_p), ("patch_b", C.c_void_p), ("cls", C.c_void_p), ("pos", C.c_void_p),
                ("layers", C.POINTER(OracleLayer))]


def build(force: bool = False) -> None:
    targets = ["libvit_oracle.so", "libvit_oracle_f32.so"]
    src = os.path.join(_HERE, "vit_oracle.c")
    stale = force or any(
        not os.path.exists(os.path.join(_HERE, t)) or os.path.getmtime(os.path.join(_HERE, t)) < os.path.getmtime(src)
        for t in targets)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B"] + targets, check=True, capture_output=True)


def lib(f32acc: bool = False) -> C.CDLL:
    name = "libvit_oracle_f32.so" if f32acc else "libvit_oracle.so"
    if name not in _LIBS:
        build()
        L = C.CDLL(os.path.join(_HERE, name))
        L.oracle_abi_version.restype = C.c_int
        L.oracle_acc_bytes.restype = C.c_int
        i64, fp, f32 = C.c_int64, C.c_void_p, C.c_float
        L.oracle_linear.argtypes = [fp, fp, fp, i64, i64, i64, fp]
        L.oracle_linear.restype = None
        L.oracle_layernorm.argtypes = [fp, fp, fp, i64, i64, f32, fp]
        L.oracle_layernorm.restype = None
        L.oracle_gelu.argtypes = [fp, i64, fp]
        L.oracle_gelu.restype = None
        L.oracle_attention.argtypes = [fp, fp, fp, i64, i64, i64, i64, i64, i64, i64, i64, f32, fp]
        L.oracle_attention.restype = None
        L.oracle_embed.argtypes = [fp, fp, fp, fp, fp, i64, i64, i64, i64, i64, i64, fp]
        L.oracle_embed.restype = None
        L.oracle_vit_forward.argtypes = [C.POINTER(OracleCfg), C.POINTER(OracleWeights), fp, i64, i64, i64, fp,
                                         C.POINTER(C.c_void_p)]
        L.oracle_vit_forward.restype = C.c_int
        L.oracle_tap_to_map.argtypes = [fp, i64, i64, i64, i64, C.c_double, fp]
        L.oracle_tap_to_map.restype = None
        L.oracle_preprocess.argtypes = [fp, i64, i64, i64, C.c_double, C.c_double, i64, i64, fp]
        L.oracle_preprocess.restype = None
        _LIBS[name] = L
    return _LIBS[name]


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def linear(x, w, b=None, f32acc=False) -> np.ndarray:
    x, w = _f32(x), _f32(w)
    b = None if b is None else _f32(b)
    M, K = x.shape
    N = w.shape[0]
    y = np.empty((M, N), np.float32)
    lib(f32acc).oracle_linear(_p(x), _p(w), _p(b), M, K, N, _p(y))
    return y


def layernorm(x, g, b, eps=1e-12, f32acc=False) -> np.ndarray:
    x, g, b = _f32(x), _f32(g), _f32(b)
    y = np.empty_like(x)
    lib(f32acc).oracle_layernorm(_p(x), _p(g), _p(b), x.shape[0], x.shape[1], eps, _p(y))
    return y


def gelu(x, f32acc=False) -> np.ndarray:
    x = _f32(x)
    y = np.empty_like(x)
    lib(f32acc).oracle_gelu(_p(x), x.size, _p(y))
    return y


def attention(q, k, v, heads: int, scale: Optional[float] = None, f32acc=False) -> np.ndarray:
    """q, k, v: [B, N, H*D] token-major (heads interleaved along the last axis)."""
    q, k, v = _f32(q), _f32(k), _f32(v)
    B, N, Cc = q.shape
    D = Cc // heads
    o = np.empty_like(q)
    lib(f32acc).oracle_attention(_p(q), _p(k), _p(v), B, N, heads, D, Cc, Cc, Cc, Cc,
                                 float(D ** -0.5 if scale is None else scale), _p(o))
    return o


def tap_to_map(tap, gh: int, gw: int, scale: float) -> np.ndarray:
    tap = _f32(tap)
    B, N, Cc = tap.shape
    assert N == gh * gw + 1
    oh, ow = int(np.floor(gh * scale)), int(np.floor(gw * scale))
    out = np.empty((B, Cc, oh, ow), np.float32)
    lib().oracle_tap_to_map(_p(tap), B, gh, gw, Cc, float(scale), _p(out))
    return out


def preprocess(img, size: int = 224, mean: float = 0.5, std: float = 0.5) -> np.ndarray:
    img = _f32(img)
    ch, h, w = img.shape
    out = np.empty((ch, size, size), np.float32)
    lib().oracle_preprocess(_p(img), ch, h, w, mean, std, size, size, _p(out))
    return out


def vit_forward(cfg, weights: Dict[str, np.ndarray], x, pos: Optional[np.ndarray] = None,
                all_hidden: bool = False, f32acc: bool = False):
    """Run the encoder.  ``cfg``: a layoutdit_amd.config.DiTConfig-like object; ``weights``: 4.49-named fp32 arrays;
    ``pos``: optional ``[1+Gh*Gw, C]`` position table already resampled for this input size (default: the
    model's own table, valid when the input grid equals the table's grid).
    Returns ``(taps, hidden)``: list of ``[B,N,C]`` arrays for ``cfg.taps`` and (optionally) ``[L+1,B,N,C]``."""
    x = _f32(x)
    B, in_ch, Himg, Wimg = x.shape
    Cc, Lr = cfg.hidden_size, cfg.num_hidden_layers
    N = (Himg // cfg.patch_size) * (Wimg // cfg.patch_size) + 1
    keep: List[np.ndarray] = []

    def hold(a):
        a = _f32(a)
        keep.append(a)
        return a.ctypes.data

    ocfg = OracleCfg(C=Cc, L=Lr, H=cfg.num_attention_heads, F=cfg.intermediate_size, patch=cfg.patch_size,
                     in_ch=in_ch, n_taps=len(cfg.taps), ln_eps=cfg.layer_norm_eps)
    for i, t in enumerate(cfg.taps):
        ocfg.taps[i] = t
    layers = (OracleLayer * Lr)()
    for l in range(Lr):
        for f, key in _LAYER_KEYS.items():
            setattr(layers[l], f, hold(weights[f"encoder.layer.{l}.{key}"]))
    if pos is None:
        pos = weights["embeddings.position_embeddings"].reshape(-1, Cc)
    pos = _f32(pos)
    if pos.shape != (N, Cc):
        raise ValueError(f"position table {pos.shape} does not match the {N}-token input; pass a resampled `pos`")
    ow = OracleWeights(patch_w=hold(weights["embeddings.patch_embeddings.projection.weight"]),
                       patch_b=hold(weights["embeddings.patch_embeddings.projection.bias"]),
                       cls=hold(weights["embeddings.cls_token"]), pos=hold(pos), layers=layers)
    taps = [np.empty((B, N, Cc), np.float32) for _ in cfg.taps]
    tap_ptrs = (C.c_void_p * len(taps))(*[t.ctypes.data for t in taps])
    hidden = np.empty((Lr + 1, B, N, Cc), np.float32) if all_hidden else None
    rc = lib(f32acc).oracle_vit_forward(C.byref(ocfg), C.byref(ow), _p(x), B, Himg, Wimg, _p(hidden), tap_ptrs)
    if rc != 0:
        raise RuntimeError(f"oracle_vit_forward failed ({rc})")
    return taps, hidden
