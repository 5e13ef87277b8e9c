#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference arithmetic.

Run in the BUILD CONTAINER only (needs `transformers`; the GPU box never runs this):

    python tests/golden/make_golden.py

The reference's hot path is `hs = self.dit(x).hidden_states` (ref src/layoutdit/modeling/dit_backbone.py:47) with
`self.dit` = HuggingFace `BeitModel`.  `from_pretrained("microsoft/dit-base")` is a hub fetch and is not attempted;
the model is built from a LOCAL `BeitConfig` (DiT-base = BEiT-base + absolute position embeddings + mask token,
SURVEY.md 8(b)) and loaded with the deterministic synthetic parameters of `layoutdit_amd.synth` (HF's own init
zeroes every bias / position / cls entry, which would hide bugs).  CPU, fp32, eval, SDPA attention (HF default).

Outputs are DATA only (inputs, expected outputs, sub-samples and statistics); no reference source is stored.
Files: g0_micro.npz (full tensors), g1_tiny.npz, g2_base.npz, g3_large512.npz (strided sub-samples + statistics),
g4_ops.npz (per-op known answers), g5_maps.npz (DiTBackbone tap post-processing).
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from layoutdit_amd import config as cfgs            # noqa: E402
from layoutdit_amd import synth                     # noqa: E402
from layoutdit_amd.modeling.keys import to_v5       # noqa: E402
from tests.golden.kat_inputs import attn_inputs, ln_inputs  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.manual_seed(0)
torch.set_grad_enabled(False)


def build_hf(cfg, weights):
    from transformers import BeitConfig, BeitModel
    hc = BeitConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                    num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                    image_size=cfg.image_size, patch_size=cfg.patch_size, num_channels=cfg.num_channels,
                    layer_norm_eps=cfg.layer_norm_eps, layer_scale_init_value=cfg.layer_scale_init_value,
                    use_absolute_position_embeddings=True, use_mask_token=True, output_hidden_states=True)
    m = BeitModel(hc).eval()
    sd = {to_v5(k): torch.from_numpy(v.copy()) for k, v in weights.items()}
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    return m


def weight_fingerprint(weights):
    tot = 0.0
    for k in sorted(weights):
        tot += float(np.sum(weights[k].astype(np.float64) * (1.0 + (len(k) % 7))))
    return np.array([tot], dtype=np.float64)


def stats(a):
    a64 = a.astype(np.float64)
    return np.array([a64.mean(), a64.std(), np.sqrt((a64 ** 2).sum()), np.abs(a64).max()], dtype=np.float64)


def run_model(cfg, weights, x, interpolate=False):
    m = build_hf(cfg, weights)
    out = m(torch.from_numpy(x), interpolate_pos_encoding=interpolate)
    hs = [h.numpy() for h in out.hidden_states]
    assert len(hs) == cfg.num_hidden_layers + 1
    return m, hs


def emit_sampled(name, cfg, B, size, wseed, xseed, stride, kind="doc"):
    w = synth.synth_weights(cfg, wseed)
    x = synth.synth_images(B, size, size, seed=xseed, kind=kind)
    interp = size != cfg.image_size
    m, hs = run_model(cfg, w, x, interpolate=interp)
    rec = dict(geometry=np.array([cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads,
                                  cfg.intermediate_size, cfg.patch_size, cfg.image_size, B, size], dtype=np.int64),
               seeds=np.array([wseed, xseed], dtype=np.int64), stride=np.array([stride], dtype=np.int64),
               taps=np.array(cfg.taps, dtype=np.int64), weight_fingerprint=weight_fingerprint(w),
               x_fingerprint=stats(x))
    for t in cfg.taps:
        a = hs[t]
        rec[f"tap{t}_sample"] = a.reshape(-1)[::stride].copy()
        rec[f"tap{t}_head"] = a[0, :8, :8].copy()
        rec[f"tap{t}_stats"] = stats(a)
    rec["emb_sample"] = hs[0].reshape(-1)[::stride].copy()
    if interp:
        pos = m.embeddings.interpolate_pos_encoding(out_like(hs[0]), size, size).numpy()[0]
        rec["pos_resampled_sample"] = pos.reshape(-1)[::stride].copy()
        rec["pos_resampled_stats"] = stats(pos)
    np.savez_compressed(os.path.join(OUT, name), **rec)
    print(name, {k: v.shape for k, v in rec.items() if k.endswith("_sample")})
    return w, x, hs


def out_like(h):
    return torch.from_numpy(h)


def main():
    # ---- G0: micro, full tensors -------------------------------------------------------------------------------
    cfg = cfgs.vit_micro()
    w = synth.synth_weights(cfg, seed=7)
    x = synth.synth_images(2, 64, 64, seed=99, kind="uniform")
    _, hs = run_model(cfg, w, x)
    np.savez_compressed(os.path.join(OUT, "g0_micro.npz"), x=x, hidden=np.stack(hs),
                        taps=np.array(cfg.taps, dtype=np.int64), seeds=np.array([7, 99], dtype=np.int64),
                        weight_fingerprint=weight_fingerprint(w))
    micro_hs = hs
    print("g0_micro", np.stack(hs).shape)

    # ---- G1..G3: sub-sampled taps ------------------------------------------------------------------------------
    _, _, tiny_hs = emit_sampled("g1_tiny.npz", cfgs.vit_tiny(), 2, 224, wseed=1, xseed=1234, stride=7)
    emit_sampled("g2_base.npz", cfgs.vit_base(), 2, 224, wseed=0, xseed=1234, stride=31)
    emit_sampled("g3_large512.npz", cfgs.vit_large(), 1, 512, wseed=3, xseed=1234, stride=101)

    # ---- G4: per-op known answers (torch functional ops the HF modules call) -------------------------------------
    rec = {}
    Cc = 768
    rows, g, b = ln_inputs(Cc)
    rec["ln_x"], rec["ln_g"], rec["ln_b"] = rows, g, b
    rec["ln_y"] = F.layer_norm(torch.from_numpy(rows), (Cc,), torch.from_numpy(g), torch.from_numpy(b), 1e-12).numpy()
    grid = np.concatenate([np.linspace(-8, 8, 2049), [0.0, -0.0, 1e-8, -1e-8, 30.0, -30.0]]).astype(np.float32)
    rec["gelu_x"] = grid
    rec["gelu_y"] = F.gelu(torch.from_numpy(grid)).numpy()
    for n_tok, tag in ((197, "197"), (1025, "1025")):
        H, D, Bq = 3, 64, 1
        q, k, v = attn_inputs(n_tok, H, D, Bq)
        tq, tk, tv = (torch.from_numpy(a).view(Bq, n_tok, H, D).transpose(1, 2) for a in (q, k, v))
        o = F.scaled_dot_product_attention(tq, tk, tv, scale=D ** -0.5).transpose(1, 2).reshape(Bq, n_tok, H * D)
        rec[f"attn{tag}_o"] = o.numpy()
    xl = synth.normal(15, 1, 37 * 96).reshape(37, 96).astype(np.float32)
    wl = (0.05 * synth.normal(15, 2, 50 * 96)).reshape(50, 96).astype(np.float32)
    bl = (0.05 * synth.normal(15, 3, 50)).astype(np.float32)
    rec["lin_x"], rec["lin_w"], rec["lin_b"] = xl, wl, bl
    rec["lin_y"] = F.linear(torch.from_numpy(xl), torch.from_numpy(wl), torch.from_numpy(bl)).numpy()
    np.savez_compressed(os.path.join(OUT, "g4_ops.npz"), **rec)
    print("g4_ops", sorted(rec))

    # ---- G5: tap post-processing exactly as DiTBackbone.forward applies it (ref dit_backbone.py:50-61) -----------
    rec = {}
    for tag, hs_, cfg_, size in (("micro", micro_hs, cfgs.vit_micro(), 64), ("tiny", tiny_hs, cfgs.vit_tiny(), 224)):
        g_ = size // 16
        for i, (idx, scale) in enumerate(zip(cfg_.taps, [4.0, 2.0, 1.0, 0.5]), start=2):
            t = torch.from_numpy(hs_[idx])[:, 1:, :]
            t = t.permute(0, 2, 1).reshape(t.shape[0], cfg_.hidden_size, g_, g_)
            if scale != 1.0:
                t = F.interpolate(t, scale_factor=scale, mode="bilinear", align_corners=False)
            a = t.contiguous().numpy()
            if tag == "micro":
                rec[f"{tag}_p{i}"] = a
            else:
                rec[f"{tag}_p{i}_sample"] = a.reshape(-1)[::13].copy()
                rec[f"{tag}_p{i}_shape"] = np.array(a.shape, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "g5_maps.npz"), **rec)
    print("g5_maps", sorted(rec))


if __name__ == "__main__":
    main()
