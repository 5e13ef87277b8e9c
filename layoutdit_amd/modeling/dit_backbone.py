"""``DiTBackbone`` with the reference's surface (ref ``src/layoutdit/modeling/dit_backbone.py:16-62``), its ``dit``
attribute being the MI355X-native :class:`DiTEncoder` instead of ``AutoModel.from_pretrained("microsoft/dit-base")``.

Same contract as the reference class: ``forward(x: [B,3,H,W]) -> OrderedDict{p2, p3, p4, p5}`` with
``p_i = bilinear(scale_i)(tap_i without CLS, viewed as [B, C, Gh, Gw])``, taps ``[d/3, d/2, 2d/3, d]``, scales
``[4, 2, 1, 0.5]``.  The rescale runs in ``ldit_tap_to_map_f32`` (HIP); the x1 level is the same zero-copy strided
view the reference returns.

A maintainer of the reference can instead keep their own ``DiTBackbone`` and only swap the attribute:
``backbone.dit = layoutdit_amd.DiTEncoder(cfg).to("cuda")``.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from ..config import DiTConfig
from .dit_encoder import DiTEncoder


class DiTBackbone(nn.Module):
    def __init__(self, pretrained: bool = False, config: Optional[DiTConfig] = None,
                 checkpoint: Optional[str] = None, compute_dtype: str = "f32"):
        """``pretrained=True`` in the reference means a hub download; offline it must come with a local
        ``checkpoint`` path (a ``state_dict`` in any of the accepted BEiT key layouts, loaded with
        ``weights_only=True``).  ``compute_dtype`` selects the encoder build (``"f32"``, ``"bf16"``, ``"fp8"`` - the
        last needs ``self.dit.calibrate_fp8(sample)`` once); the returned maps are fp32 in every build."""
        super().__init__()
        self.dit = DiTEncoder(config, compute_dtype=compute_dtype)
        if pretrained and checkpoint is None:
            raise ValueError("pretrained=True needs checkpoint=<local state_dict path>: there is no hub access "
                             "(the reference fetches microsoft/dit-base, ref dit_backbone.py:25-31)")
        if checkpoint is not None:
            sd = torch.load(checkpoint, map_location="cpu", weights_only=True)
            self.dit.load_state_dict(sd, strict=False)
        d = self.dit.config.num_hidden_layers
        self.layer_idxs = [d // 3, d // 2, 2 * d // 3, d]       # ref dit_backbone.py:33-34
        self.scales = [4.0, 2.0, 1.0, 0.5]
        self.hidden_size = self.dit.config.hidden_size

    def forward(self, x: torch.Tensor):
        B, _, H, W = x.shape
        patch_size = self.dit.config.patch_size
        Gh, Gw = H // patch_size, W // patch_size
        hs = self.dit(x, taps=self.layer_idxs).hidden_states
        feats = OrderedDict()
        for i, (idx, scale) in enumerate(zip(self.layer_idxs, self.scales), start=2):
            h = hs[idx]
            if scale == 1.0:
                # zero-copy, physically NHWC view - exactly what the reference hands to the FPN (dit_backbone.py:52-54)
                t = h[:, 1:, :].permute(0, 2, 1).unflatten(2, (Gh, Gw))
            else:
                # the encoder returns fp32 contiguous taps for fp32 input: no host-side copies on that path
                src = h if (h.dtype == torch.float32 and h.is_contiguous()) else h.float().contiguous()
                # under autograd (train mode) the rescale carries its adjoint kernel: gradients reach the encoder
                t = ops.tap_to_map_autograd(src, Gh, Gw, scale) if src.requires_grad else ops.tap_to_map(src, Gh, Gw, scale)
                if t.dtype != h.dtype:
                    t = t.to(h.dtype)
            feats[f"p{i}"] = t
        return feats
