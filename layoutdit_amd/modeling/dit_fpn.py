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

Training (round 3): ``self.fpn(feats)`` is differentiated as the reference's loop does it (ref dit_backbone.py:87-90 under
ref trainer.py:169-178) - :class:`_FPNFn` is one ``torch.autograd.Function`` over the whole stage whose backward runs on the
library's kernels: 3x3 dgrad = the same implicit-im2col fp32 GEMM on the flipped weight, 3x3 and lateral wgrad = bf16 MFMA
GEMMs on reduction-major operands (fp32 accumulation; gradients of the FPN weights carry bf16 operand rounding, like the
encoder's), the merge adjoint and the bias column sums in ``csrc/fpn_bwd.hip``.  Gradients reach the four taps, hence the
encoder (``layoutdit_amd.training``).  Oracle: autograd of ``oracle/fpn_oracle_torch.py`` (parity unpinned, as above).
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


def _flip_ihwo(w: torch.Tensor) -> torch.Tensor:
    """OIHW 3x3 weight -> the operand of its dgrad convolution: [Cin, 3, 3, Cout] with the taps flipped."""
    return w.detach().flip(2, 3).permute(1, 2, 3, 0).contiguous()


def _fpn_forward(bb_scales, gh: int, gw: int, toks, lat_w, lat_b, conv_w_ohwi, conv_b, keep_inner: bool):
    """The stage on fp32 contiguous token tensors [B, 1+P, C]; returns NHWC outputs (coarsest last) and, for the backward,
    the four merged maps."""
    B, T, Cc = toks[0].shape
    lats = [ops.linear(toks[i].reshape(B * T, Cc), lat_w[i], lat_b[i]).view(B, T, -1) for i in range(4)]
    inner, outs, inners = None, [None] * 4, [None] * 4
    for i in (3, 2, 1, 0):                                            # coarsest first (top-down)
        inner = ops.fpn_merge(lats[i], gh, gw, bb_scales[i], top=inner)
        outs[i] = ops.conv3x3_nhwc(inner, conv_w_ohwi[i], conv_b[i])
        if keep_inner:
            inners[i] = inner
    return outs, inners


class _FPNFn(torch.autograd.Function):
    """p2..p5 = FPN(taps) with gradients for the taps and for every FPN parameter.
    apply(geom, t0..t3, lw0, lb0, .. lw3, lb3, cw0, cb0, .. cw3, cb3) -> (p2, p3, p4, p5) as [B, 256, h, w] channels-last views."""

    @staticmethod
    def forward(ctx, geom, *args):
        gh, gw, scales = geom
        toks = [_f32c(t.detach()) for t in args[0:4]]
        lw = [args[4 + 2 * i].detach() for i in range(4)]
        lb = [args[5 + 2 * i].detach() for i in range(4)]
        cw = [args[12 + 2 * i].detach() for i in range(4)]
        cb = [args[13 + 2 * i].detach() for i in range(4)]
        Ch, Cc = lw[0].shape[0], toks[0].shape[2]
        outs, inners = _fpn_forward(scales, gh, gw, toks, [w.reshape(Ch, Cc) for w in lw], lb,
                                    [w.permute(0, 2, 3, 1).contiguous() for w in cw], cb, keep_inner=True)
        ctx.geom = geom
        ctx.toks, ctx.inners, ctx.lw, ctx.cw = toks, inners, lw, cw
        return tuple(o.permute(0, 3, 1, 2) for o in outs)

    @staticmethod
    def backward(ctx, *dps):
        gh, gw, scales = ctx.geom
        if ctx.inners is None:
            raise RuntimeError("DiTWithFPN: backward through the FPN a second time (its saved maps are freed after the first)")
        toks, inners, lw, cw = ctx.toks, ctx.inners, ctx.lw, ctx.cw
        ctx.inners = None
        B, T, Cc = toks[0].shape
        Ch = lw[0].shape[0]
        need = ctx.needs_input_grad[1:]                              # lines up with ``args``: t0..t3, (lw, lb) x 4, (cw, cb) x 4
        d_inner = [None] * 4
        g_cw, g_cb = [None] * 4, [None] * 4
        for i in range(4):                                            # finest first: the top-down adjoint flows fine -> coarse
            h, w = inners[i].shape[1], inners[i].shape[2]
            dp = dps[i]
            if dp is not None:
                dy = dp.permute(0, 2, 3, 1)
                dy = dy.float() if dy.dtype != torch.float32 else dy
                dy = dy.contiguous()                                  # NHWC [B, h, w, Ch]; a no-op for channels-last gradients
                d_inner[i] = ops.conv3x3_nhwc(dy, _flip_ihwo(cw[i]))                    # dgrad: same GEMM, flipped weight
                if need[13 + 2 * i]:
                    g_cb[i] = ops.colsum(dy.view(B * h * w, Ch))
                if need[12 + 2 * i]:
                    slack = w + 3
                    dy_p = ops.pad_nhwc_bf16(dy)                                         # [B (h+2)(w+2), Ch]
                    in_p = ops.pad_nhwc_bf16(inners[i], slack_rows=slack)
                    K = dy_p.shape[0]
                    taps = []
                    for ky in range(3):
                        for kx in range(3):
                            shift = (ky - 1) * (w + 2) + (kx - 1)
                            taps.append(ops.wgrad_bf16(dy_p, in_p, K, w_row_offset=slack + shift))   # [co, ci]
                    g_cw[i] = torch.stack(taps, dim=2).view(Ch, Ch, 3, 3)               # [co, ci, ky, kx] (layout plumbing)
                    del dy_p, in_p
            else:
                d_inner[i] = torch.zeros_like(inners[i])
            if i > 0:
                ops.fpn_merge_bwd(d_inner[i - 1], gh, gw, scales[i - 1], d_top=d_inner[i], want_lat=False)
            inners[i] = None
        g_tok, g_lw, g_lb = [None] * 4, [None] * 4, [None] * 4
        for i in range(4):
            d_lat = ops.fpn_merge_bwd(d_inner[i], gh, gw, scales[i])                     # [B, T, Ch], CLS row zero
            d_inner[i] = None
            d2 = d_lat.view(B * T, Ch)
            if need[5 + 2 * i]:
                g_lb[i] = ops.colsum(d2)
            if need[4 + 2 * i]:
                g_lw[i] = ops.wgrad_bf16(ops.cast_bf16(d2), ops.cast_bf16(toks[i].view(B * T, Cc)), B * T).view(Ch, Cc, 1, 1)
            if need[i]:
                wt = lw[i].reshape(Ch, Cc).t().contiguous()                              # [C, Ch]: the dgrad's K-contiguous operand
                g_tok[i] = ops.linear(d2, wt).view(B, T, Cc)
        grads = [None] + g_tok
        for i in range(4):
            grads += [g_lw[i], g_lb[i]]
        for i in range(4):
            grads += [g_cw[i], g_cb[i]]
        return tuple(grads)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = ops.widen_f16(t.contiguous()) if t.dtype == torch.float16 else t.float()
    return t.contiguous()


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
        bb = self.backbone
        B, _, H, W = x.shape
        p = bb.dit.config.patch_size
        gh, gw = H // p, W // p
        hs = bb.dit(x, taps=bb.layer_idxs).hidden_states
        toks = [hs[idx] for idx in bb.layer_idxs]
        lat = [self.fpn.inner_blocks[i][0] for i in range(4)]
        conv = [self.fpn.layer_blocks[i][0] for i in range(4)]
        wants_grad = torch.is_grad_enabled() and (any(t.requires_grad for t in toks)
                                                  or any(q.requires_grad for q in self.fpn.parameters()))
        if wants_grad:
            # one differentiable stage: gradients for the taps (hence the encoder) and for every FPN parameter
            args = list(toks)
            for m in lat:
                args += [m.weight, m.bias]
            for m in conv:
                args += [m.weight, m.bias]
            results = list(_FPNFn.apply((gh, gw, tuple(bb.scales)), *args))
        else:
            Cc = bb.hidden_size
            outs, _ = _fpn_forward(bb.scales, gh, gw, [_f32c(t.detach()) for t in toks],
                                   [m.weight.detach().reshape(256, Cc) for m in lat], [m.bias.detach() for m in lat],
                                   [self._ohwi(i) for i in range(4)], [m.bias.detach() for m in conv], keep_inner=False)
            results = [o.permute(0, 3, 1, 2) for o in outs]          # [B, 256, h, w], channels-last memory
        feats = OrderedDict((f"p{i + 2}", r) for i, r in enumerate(results))
        feats["pool"] = results[3][:, :, ::2, ::2]                    # LastLevelMaxPool: max_pool2d(kernel 1, stride 2)
        if x.dtype == torch.float16 and not wants_grad:
            feats = OrderedDict((k, ops.narrow_f16(v.contiguous())) for k, v in feats.items())
        return feats
