"""Deterministic synthetic weights and page images (no network: neither ``microsoft/dit-base``
nor PubLayNet exist offline).

A counter-based generator (SplitMix64 finaliser over ``seed, stream, index``) so the same tensors can be
rebuilt anywhere from three integers - nothing large is ever committed or shipped.  Parameter
distributions follow SURVEY.md 8(c): HF's own init leaves every bias / position / cls entry at zero and
LayerScale at 0.1 (transformers ``models/beit/modeling_beit.py:467-482``), which would hide bias and
scale bugs, so everything is re-randomised.

Parameter names are the ``transformers==4.49.0`` BEiT key names the reference pins
(ref ``uv.lock:1771-1772``; keys listed in SURVEY.md 5.4).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .config import DiTConfig

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _bits(seed: int, stream: int, n: int, lane: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([np.uint64(seed) * np.uint64(0x2545F4914F6CDD1D)
                                     + np.uint64(stream) * np.uint64(0x9E3779B97F4A7C15)
                                     + np.uint64(lane) * np.uint64(0xD1B54A32D192ED03)], dtype=np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64)
        return _splitmix64(base + idx * np.uint64(0xA0761D6478BD642F))


def uniform01(seed: int, stream: int, n: int, lane: int = 0) -> np.ndarray:
    """float64 uniforms in (0, 1): 53 random bits, never exactly 0."""
    b = _bits(seed, stream, n, lane) >> np.uint64(11)
    return (b.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(seed: int, stream: int, n: int) -> np.ndarray:
    """float64 standard normals (Box-Muller over two independent lanes)."""
    u1 = uniform01(seed, stream, n, 0)
    u2 = uniform01(seed, stream, n, 1)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def _stream_of(name: str) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h & 0x7FFFFFFF


def param_shapes(cfg: DiTConfig) -> Dict[str, tuple]:
    C, F, p, ch = cfg.hidden_size, cfg.intermediate_size, cfg.patch_size, cfg.num_channels
    shapes = {
        "embeddings.cls_token": (1, 1, C),
        "embeddings.mask_token": (1, 1, C),
        "embeddings.position_embeddings": (1, cfg.num_patches + 1, C),
        "embeddings.patch_embeddings.projection.weight": (C, ch, p, p),
        "embeddings.patch_embeddings.projection.bias": (C,),
    }
    for i in range(cfg.num_hidden_layers):
        pre = f"encoder.layer.{i}."
        shapes.update({
            pre + "lambda_1": (C,),
            pre + "lambda_2": (C,),
            pre + "layernorm_before.weight": (C,),
            pre + "layernorm_before.bias": (C,),
            pre + "attention.attention.query.weight": (C, C),
            pre + "attention.attention.query.bias": (C,),
            pre + "attention.attention.key.weight": (C, C),      # no key bias (modeling_beit.py:306)
            pre + "attention.attention.value.weight": (C, C),
            pre + "attention.attention.value.bias": (C,),
            pre + "attention.output.dense.weight": (C, C),
            pre + "attention.output.dense.bias": (C,),
            pre + "layernorm_after.weight": (C,),
            pre + "layernorm_after.bias": (C,),
            pre + "intermediate.dense.weight": (F, C),
            pre + "intermediate.dense.bias": (F,),
            pre + "output.dense.weight": (C, F),
            pre + "output.dense.bias": (C,),
        })
    shapes["pooler.layernorm.weight"] = (C,)
    shapes["pooler.layernorm.bias"] = (C,)
    return shapes


def synth_weights(cfg: DiTConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """fp32 parameters keyed by transformers-4.49 BEiT names."""
    out: Dict[str, np.ndarray] = {}
    for name, shape in param_shapes(cfg).items():
        n = int(np.prod(shape))
        s = _stream_of(name)
        if "lambda_" in name:
            v = 0.05 + 0.45 * uniform01(seed, s, n)            # U(0.05, 0.5)
        elif "layernorm" in name and name.endswith(".weight"):
            v = 1.0 + 0.1 * normal(seed, s, n)
        elif "layernorm" in name and name.endswith(".bias"):
            v = 0.1 * normal(seed, s, n)
        else:
            v = 0.02 * normal(seed, s, n)
        out[name] = v.astype(np.float32).reshape(shape)
    return out


def synth_images(batch: int, height: int, width: int, seed: int = 1234, kind: str = "doc",
                 first_index: int = 0) -> np.ndarray:
    """``[batch, 3, height, width]`` fp32 NCHW in [-1, 1], i.e. already ``(img - 0.5) / 0.5``
    (ref ``src/layoutdit/modeling/model.py:53-54``).

    ``doc``: white page (+1) with 20-60 dark axis-aligned rectangles ("text lines"), gaussian noise
    sigma 0.02, clipped - the stand-in for PubLayNet crops.  ``uniform``: U(-1, 1).
    Sample ``i`` depends only on ``(seed, first_index + i)`` so a rank's shard equals the same rows of
    the global batch.
    """
    x = np.empty((batch, 3, height, width), dtype=np.float32)
    n = 3 * height * width
    for i in range(batch):
        sid = first_index + i
        if kind == "uniform":
            x[i] = (2.0 * uniform01(seed, sid, n) - 1.0).astype(np.float32).reshape(3, height, width)
            continue
        if kind != "doc":
            raise ValueError(f"unknown image kind {kind!r}")
        page = np.ones((height, width), dtype=np.float64)
        r = uniform01(seed, sid, 1 + 5 * 60, lane=2)
        nrect = 20 + int(r[0] * 41)
        for j in range(nrect):
            a = r[1 + 5 * j: 6 + 5 * j]
            y0 = int(a[0] * height)
            x0 = int(a[1] * width)
            hh = 1 + int(a[2] * max(1, height // 16))
            ww = 4 + int(a[3] * max(1, width // 2))
            page[y0:y0 + hh, x0:x0 + ww] = -a[4]               # U(-1, 0)
        img = page[None, :, :] + 0.02 * normal(seed, sid, n).reshape(3, height, width)
        x[i] = np.clip(img, -1.0, 1.0).astype(np.float32)
    return x
