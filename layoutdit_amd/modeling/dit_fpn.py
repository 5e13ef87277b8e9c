"""``DiTWithFPN`` with the reference's surface (ref ``src/layoutdit/modeling/dit_backbone.py:65-90``): ``DiTBackbone`` +
torchvision's ``FeaturePyramidNetwork([C] * 4, 256, extra_blocks=LastLevelMaxPool())``, ``out_channels = 256``,
``forward(x) -> OrderedDict{p2, p3, p4, p5, pool}`` - re-designed for the MI355X instead of translated:

* the four 1x1 lateral convolutions run on the TOKENS of the taps (``ldit_linear_f32`` on ``[B * 197, C]``), not on the
  rescaled maps: a 1x1 convolution commutes with the bilinear rescale of ``DiTBackbone.forward`` (both are linear and the
  rescale's weights sum to one), so the 768-channel ``p2`` map (616 MB at bs=64) is never materialised and the p2 lateral
  costs 16x fewer FLOPs;
* rescale + top-down nearest add is one NHWC kernel per level (``ldit_fpn_merge_f32``);
* the 3x3 output convolutions are implicit-im2col fp32 MFMA GEMMs on the NHWC maps (``ldit_conv3x3_nhwc_f32``); their
  results are returned as ``[B, 256, H, W]`` tensors in channels-last memory (exactly the kind of strided NCHW view the
  reference's own ``p4`` is, ref dit_backbone.py:52-54);
* ``LastLevelMaxPool`` (``max_pool2d(kernel 1, stride 2)``) is a strided view.

torchvision is not installed offline and its source is not part of the reference: the FPN arithmetic is
restated from its documented forward (``oracle/fpn_oracle_torch.py``) - **parity unpinned** with respect to torchvision itself.
Parameter names follow torchvision (``fpn.inner_blocks.{i}.0.weight`` ...), so detector checkpoints load.
Inference only: the FPN has no backward here (the encoder below it has: ``layoutdit_amd.training``).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional

import torch
import torch.nn as nn

from .. import ops
from ..config import DiTConfig
from .dit_backbone import DiTBackbone


class _Conv(nn.Sequential):
    """``Conv2dNormActivation(in, out, k, padding, norm_layer=None, activation_layer=None)``: a Sequential holding the
    convolution at index 0 - gives the parameters torchvision's key names (``...{i}.0.weight``)."""

    def __init__(self, cin: int, cout: int, k: int):
        super().__init__(nn.Conv2d(cin, cout, kernel_size=k, padding=k // 2))


class _FPNParams(nn.Module):
    def __init__(self, in_channels_list, out_channels: int):
        super().__init__()
        self.inner_blocks = nn.ModuleList([_Conv(c, out_channels, 1) for c in in_channels_list])
        self.layer_blocks = nn.ModuleList([_Conv(out_channels, out_channels, 3) for _ in in_channels_list])
        for m in self.modules():                     # torchvision's init: kaiming_uniform_(a=1), zero bias
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)


class DiTWithFPN(nn.Module):
    def __init__(self, pretrained: bool = False, config: Optional[DiTConfig] = None, checkpoint: Optional[str] = None,
                 compute_dtype: str = "f32"):
        super().__init__()
        self.backbone = DiTBackbone(pretrained=pretrained, config=config, checkpoint=checkpoint, compute_dtype=compute_dtype)
        self.fpn = _FPNParams([self.backbone.hidden_size] * 4, 256)
        self.out_channels = 256
        self._cache: Dict[int, tuple] = {}

    def _ohwi(self, i: int) -> torch.Tensor:
        """3x3 weight of level i as [Cout, 3, 3, Cin] (GEMM operand layout), re-laid when the parameter changes."""
        w = self.fpn.layer_blocks[i][0].weight
        key = (w.data_ptr(), w._version, str(w.device))
        hit = self._cache.get(i)
        if hit is None or hit[0] != key:
            hit = (key, w.detach().permute(0, 2, 3, 1).contiguous())
            self._cache[i] = hit
        return hit[1]

    def forward(self, x: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.fpn.parameters()) and self.training:
            raise NotImplementedError("DiTWithFPN: the FPN has no backward in this library (inference only); the encoder "
                                      "below it trains through layoutdit_amd.training")
        bb = self.backbone
        B, _, H, W = x.shape
        p = bb.dit.config.patch_size
        gh, gw = H // p, W // p
        hs = bb.dit(x, taps=bb.layer_idxs).hidden_states
        Cc = bb.hidden_size
        lats = []
        for i, idx in enumerate(bb.layer_idxs):
            conv = self.fpn.inner_blocks[i][0]
            tok = hs[idx]
            if tok.dtype != torch.float32:
                tok = ops.widen_f16(tok.contiguous())
            y = ops.linear(tok.reshape(B * (gh * gw + 1), Cc), conv.weight.detach().reshape(256, Cc), conv.bias.detach())
            lats.append(y.view(B, gh * gw + 1, 256))
        inner, results = None, [None] * 4
        for i in (3, 2, 1, 0):                                       # coarsest first (top-down)
            inner = ops.fpn_merge(lats[i], gh, gw, bb.scales[i], top=inner)
            out = ops.conv3x3_nhwc(inner, self._ohwi(i), self.fpn.layer_blocks[i][0].bias.detach())
            results[i] = out.permute(0, 3, 1, 2)                      # [B, 256, h, w], channels-last memory
        feats = OrderedDict((f"p{i + 2}", r) for i, r in enumerate(results))
        feats["pool"] = results[3][:, :, ::2, ::2]                    # LastLevelMaxPool: max_pool2d(kernel 1, stride 2)
        if x.dtype == torch.float16:
            feats = OrderedDict((k, ops.narrow_f16(v.contiguous())) for k, v in feats.items())
        return feats
