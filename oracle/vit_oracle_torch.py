"""The same restatement as ``vit_oracle.c`` written with plain ``torch`` CPU ops.  TEST / BASELINE INFRASTRUCTURE ONLY.

Purpose: the *timed CPU baseline* of ``bench.py`` (``cpu_baseline.kind = "port"``).  The reference's CPU path is
HuggingFace ``BeitModel`` on ATen CPU kernels (oneDNN / MKL GEMMs, ``F.scaled_dot_product_attention``); this file
issues the very same ATen ops in the same order without importing ``transformers`` or anything from the reference,
so its speed is the reference CPU path's speed on whatever host it runs on.  It is also checked against the golden
vectors (tests/test_oracle_golden.py) so that it cannot drift from the C oracle.

Follows TF:models/beit/modeling_beit.py:81-90,153-176,296-357,406-444,504-506 (see vit_oracle.c for the line map).
Never imported by anything under ``layoutdit_amd/``.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def _t(a) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


class TorchOracle:
    def __init__(self, cfg, weights: Dict[str, np.ndarray]):
        self.cfg = cfg
        self.w = {k: _t(v) for k, v in weights.items()}

    @torch.no_grad()
    def forward(self, x, taps: Optional[Sequence[int]] = None, pos: Optional[np.ndarray] = None) -> List[torch.Tensor]:
        cfg, w = self.cfg, self.w
        x = x if isinstance(x, torch.Tensor) else _t(x)
        taps = list(cfg.taps if taps is None else taps)
        B = x.shape[0]
        C, H = cfg.hidden_size, cfg.num_attention_heads
        D = C // H
        e = F.conv2d(x, w["embeddings.patch_embeddings.projection.weight"],
                     w["embeddings.patch_embeddings.projection.bias"], stride=cfg.patch_size)
        e = e.flatten(2).transpose(1, 2)
        h = torch.cat((w["embeddings.cls_token"].expand(B, -1, -1), e), dim=1)
        table = w["embeddings.position_embeddings"] if pos is None else _t(pos).unsqueeze(0)
        h = h + table
        out = {0: h} if 0 in taps else {}
        for l in range(cfg.num_hidden_layers):
            p = f"encoder.layer.{l}."
            y = F.layer_norm(h, (C,), w[p + "layernorm_before.weight"], w[p + "layernorm_before.bias"], cfg.layer_norm_eps)
            q = F.linear(y, w[p + "attention.attention.query.weight"], w[p + "attention.attention.query.bias"])
            k = F.linear(y, w[p + "attention.attention.key.weight"])
            v = F.linear(y, w[p + "attention.attention.value.weight"], w[p + "attention.attention.value.bias"])
            q, k, v = (t.view(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
            a = F.scaled_dot_product_attention(q, k, v, scale=D ** -0.5)
            a = a.transpose(1, 2).reshape(B, -1, C)
            a = F.linear(a, w[p + "attention.output.dense.weight"], w[p + "attention.output.dense.bias"])
            h = w[p + "lambda_1"] * a + h
            y = F.layer_norm(h, (C,), w[p + "layernorm_after.weight"], w[p + "layernorm_after.bias"], cfg.layer_norm_eps)
            m = F.linear(y, w[p + "intermediate.dense.weight"], w[p + "intermediate.dense.bias"])
            m = F.gelu(m)
            m = F.linear(m, w[p + "output.dense.weight"], w[p + "output.dense.bias"])
            h = w[p + "lambda_2"] * m + h
            if (l + 1) in taps:
                out[l + 1] = h
        return [out[t] for t in taps]


def drop_path_rates(cfg, drop_path_rate: float = 0.1) -> List[float]:
    """Stochastic-depth rate of layer i: ``rate * i / (L - 1)`` (TF:models/beit/modeling_beit.py:499-502)."""
    L = cfg.num_hidden_layers
    return [drop_path_rate * i / max(L - 1, 1) for i in range(L)]


def train_reference(cfg, weights: Dict[str, np.ndarray], x, dtaps: Sequence[np.ndarray],
                    drop_scales: Optional[np.ndarray] = None, taps: Optional[Sequence[int]] = None,
                    dtype=torch.float64):
    """Forward + backward of the encoder by ``torch.autograd`` on the CPU - the gradient oracle of the train step.

    Same ATen ops in the same order as :meth:`TorchOracle.forward` (i.e. as HF ``BeitModel``), in ``dtype`` (float64 by
    default: a reference, not a speed baseline), with the train-mode extra of TF:360-378,432-434,440-442: each residual
    branch is multiplied per SAMPLE by ``drop_scales[l, branch, b]`` (= 0 or 1 / keep_prob; ``None`` = all ones = eval).
    The loss is ``sum_t <tap_t, dtaps_t>``, so ``dtaps`` are exactly the upstream gradients a detector head would send
    back into ``hidden_states``.  Returns ``(taps, grads)``: list of [B,N,C] arrays and a dict of gradients keyed like
    ``weights`` (4.49 names; the inert mask_token / pooler parameters get no entry)."""
    taps = list(cfg.taps if taps is None else taps)
    w = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype).requires_grad_(True) for k, v in weights.items()
         if "mask_token" not in k and not k.startswith("pooler.")}
    xt = (x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))).to(dtype)
    B = xt.shape[0]
    C, H = cfg.hidden_size, cfg.num_attention_heads
    D = C // H
    s = None if drop_scales is None else torch.from_numpy(np.ascontiguousarray(drop_scales)).to(dtype)
    e = F.conv2d(xt, w["embeddings.patch_embeddings.projection.weight"], w["embeddings.patch_embeddings.projection.bias"],
                 stride=cfg.patch_size)
    e = e.flatten(2).transpose(1, 2)
    pe = w["embeddings.position_embeddings"]
    gh, gw = xt.shape[2] // cfg.patch_size, xt.shape[3] // cfg.patch_size
    g0 = int(round((pe.shape[1] - 1) ** 0.5))
    if (gh, gw) != (g0, g0):
        # interpolate_pos_encoding, TF:113-151: bicubic resample (align_corners=False) of the patch part of the table, inside
        # the graph - the gradient reaches the table through the resample's adjoint
        patch = pe[:, 1:].reshape(1, g0, g0, C).permute(0, 3, 1, 2)
        patch = F.interpolate(patch, size=(gh, gw), mode="bicubic", align_corners=False)
        pe = torch.cat((pe[:, :1], patch.permute(0, 2, 3, 1).reshape(1, gh * gw, C)), dim=1)
    h = torch.cat((w["embeddings.cls_token"].expand(B, -1, -1), e), dim=1) + pe
    out = {0: h} if 0 in taps else {}
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{l}."
        y = F.layer_norm(h, (C,), w[p + "layernorm_before.weight"], w[p + "layernorm_before.bias"], cfg.layer_norm_eps)
        q = F.linear(y, w[p + "attention.attention.query.weight"], w[p + "attention.attention.query.bias"])
        k = F.linear(y, w[p + "attention.attention.key.weight"])
        v = F.linear(y, w[p + "attention.attention.value.weight"], w[p + "attention.attention.value.bias"])
        q, k, v = (t.view(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax((q @ k.transpose(-1, -2)) * D ** -0.5, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, -1, C)
        a = F.linear(a, w[p + "attention.output.dense.weight"], w[p + "attention.output.dense.bias"])
        a = w[p + "lambda_1"] * a
        h = (a if s is None else a * s[l, 0].view(B, 1, 1)) + h
        y = F.layer_norm(h, (C,), w[p + "layernorm_after.weight"], w[p + "layernorm_after.bias"], cfg.layer_norm_eps)
        m = F.gelu(F.linear(y, w[p + "intermediate.dense.weight"], w[p + "intermediate.dense.bias"]))
        m = w[p + "lambda_2"] * F.linear(m, w[p + "output.dense.weight"], w[p + "output.dense.bias"])
        h = (m if s is None else m * s[l, 1].view(B, 1, 1)) + h
        if (l + 1) in taps:
            out[l + 1] = h
    loss = sum((out[t] * torch.from_numpy(np.ascontiguousarray(d)).to(dtype)).sum() for t, d in zip(taps, dtaps))
    names = list(w)
    grads = torch.autograd.grad(loss, [w[n] for n in names], allow_unused=True)
    return ([out[t].detach().to(torch.float32).numpy() for t in taps],
            {n: (torch.zeros_like(w[n]) if g is None else g).to(torch.float32).numpy() for n, g in zip(names, grads)})
