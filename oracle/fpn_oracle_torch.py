"""CPU restatement of the FPN stage behind ``DiTWithFPN`` (ref src/layoutdit/modeling/dit_backbone.py:65-90).
TEST INFRASTRUCTURE ONLY - never imported by ``layoutdit_amd/``.

The arithmetic lives in ``torchvision==0.19.0`` (``ops/feature_pyramid_network.py``: ``FeaturePyramidNetwork.forward`` and
``LastLevelMaxPool``; pinned at ref uv.lock:1709-1742), which is NOT installed offline and whose source is not under
/root/reference: **parity unpinned**.  This file restates that module's published forward with the torch functional ops it
is built from, in the reference's own order (rescale the 768-channel map first, THEN the 1x1 lateral):

    last_inner = inner[-1](x[-1]);  results = [layer[-1](last_inner)]
    for idx = n-2 .. 0:  lateral = inner[idx](x[idx]);  top_down = interpolate(last_inner, size=lateral.shape[-2:], "nearest")
                         last_inner = lateral + top_down;  results.insert(0, layer[idx](last_inner))
    LastLevelMaxPool: results.append(max_pool2d(results[-1], kernel_size=1, stride=2, padding=0)), name "pool"
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def backbone_maps(taps: Sequence[np.ndarray], gh: int, gw: int, scales=(4.0, 2.0, 1.0, 0.5)) -> List[torch.Tensor]:
    """ref dit_backbone.py:50-61 on the four tapped hidden states ([B, 1+P, C] each)."""
    maps = []
    for tap, s in zip(taps, scales):
        t = torch.from_numpy(np.ascontiguousarray(tap, dtype=np.float32)).double()[:, 1:, :]
        t = t.permute(0, 2, 1).reshape(t.shape[0], t.shape[2], gh, gw)
        if s != 1.0:
            t = F.interpolate(t, scale_factor=s, mode="bilinear", align_corners=False)
        maps.append(t)
    return maps


def fpn_forward_t(maps: Sequence[torch.Tensor], w: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    """The stage on torch tensors (float64), graph kept: autograd of this function is the gradient oracle of the FPN backward
    (tests/test_gpu_fpn.py)."""
    n = len(maps)
    inner = lambda i, t: F.conv2d(t, w[f"inner_blocks.{i}.0.weight"], w[f"inner_blocks.{i}.0.bias"])            # noqa: E731
    layer = lambda i, t: F.conv2d(t, w[f"layer_blocks.{i}.0.weight"], w[f"layer_blocks.{i}.0.bias"], padding=1)  # noqa: E731
    last_inner = inner(n - 1, maps[-1])
    results = [layer(n - 1, last_inner)]
    for idx in range(n - 2, -1, -1):
        lateral = inner(idx, maps[idx])
        top_down = F.interpolate(last_inner, size=lateral.shape[-2:], mode="nearest")
        last_inner = lateral + top_down
        results.insert(0, layer(idx, last_inner))
    results.append(F.max_pool2d(results[-1], kernel_size=1, stride=2, padding=0))
    names = [f"p{i + 2}" for i in range(n)] + ["pool"]
    return OrderedDict(zip(names, results))


def backbone_maps_t(taps: Sequence[torch.Tensor], gh: int, gw: int, scales=(4.0, 2.0, 1.0, 0.5)) -> List[torch.Tensor]:
    """ref dit_backbone.py:50-61 on float64 tensors [B, 1+P, C], graph kept."""
    maps = []
    for t, s in zip(taps, scales):
        t = t[:, 1:, :].permute(0, 2, 1).reshape(t.shape[0], t.shape[2], gh, gw)
        if s != 1.0:
            t = F.interpolate(t, scale_factor=s, mode="bilinear", align_corners=False)
        maps.append(t)
    return maps


def fpn_forward(maps: Sequence[torch.Tensor], weights: Dict[str, np.ndarray]) -> "OrderedDict[str, np.ndarray]":
    """``weights``: torchvision-named arrays ``inner_blocks.{i}.0.weight/bias``, ``layer_blocks.{i}.0.weight/bias``."""
    w = {k: torch.from_numpy(np.ascontiguousarray(v)).double() for k, v in weights.items()}
    return OrderedDict((k, r.to(torch.float32).numpy()) for k, r in fpn_forward_t(maps, w).items())
