#!/usr/bin/env python3
"""Gradient goldens for the train step (BASELINE.json configs[2]), from the reference arithmetic.

Run in the BUILD CONTAINER only (needs `transformers`):

    python tests/golden/make_golden_grad.py

The reference trains `self.dit` = HF `BeitModel` through `loss.backward()` (ref src/layoutdit/training/trainer.py:169-178);
what reaches the encoder is the gradient of the loss with respect to `hidden_states[4, 6, 8, 12]`.  Here the model is
built from a LOCAL `BeitConfig` with the synthetic parameters of `layoutdit_amd.synth`, run on the CPU in fp32, and
differentiated with `torch.autograd` for upstream gradients `dtaps` drawn from the same counter-based generator
(loss = sum_t <hidden_states[t], dtaps[t]>).  Two passes per geometry:
  * eval-mode arithmetic (stochastic depth = identity), and
  * train mode (`model.train()`, `drop_path_rate = 0.1`): the per-sample stochastic-depth factors HF actually applied are
    CAPTURED with forward hooks on its `BeitDropPath` modules and stored next to the gradients (DATA: 0 or 1/keep_prob
    per layer, branch and sample), so the build can replay exactly the same masks.
Outputs (data only): g6_grad_micro.npz (every third element of every gradient), g7_grad_tiny.npz (strided samples + norms),
g8_grad_base.npz (ViT-B/16 224x224 bs=2 - the geometry of BASELINE.json configs[2] itself; strided samples + norms).
`python tests/golden/make_golden_grad.py g8` regenerates one file only.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from layoutdit_amd import config as cfgs            # noqa: E402
from layoutdit_amd import synth                     # noqa: E402
from layoutdit_amd.modeling.keys import to_v4, to_v5       # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
DROP_PATH_RATE = 0.1        # BeitConfig default (TF:models/beit/configuration_beit.py:90)


def build_hf(cfg, weights):
    from transformers import BeitConfig, BeitModel
    hc = BeitConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                    num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                    image_size=cfg.image_size, patch_size=cfg.patch_size, num_channels=cfg.num_channels,
                    layer_norm_eps=cfg.layer_norm_eps, layer_scale_init_value=cfg.layer_scale_init_value,
                    use_absolute_position_embeddings=True, use_mask_token=True, output_hidden_states=True,
                    drop_path_rate=DROP_PATH_RATE)
    m = BeitModel(hc)
    m.load_state_dict({to_v5(k): torch.from_numpy(v.copy()) for k, v in weights.items()}, strict=True)
    return m


def upstream(cfg, B, N, seed):
    """dtaps[i]: [B, N, C] ~ N(0, 1) / sqrt(B N C): the scale a mean-reduced loss sends back."""
    C = cfg.hidden_size
    return [(synth.normal(seed, 100 + t, B * N * C) / np.sqrt(B * N * C)).astype(np.float32).reshape(B, N, C)
            for t in cfg.taps]


def run(cfg, w, x, dtaps, train: bool, seed: int):
    m = build_hf(cfg, w)
    m.train(train)
    scales = np.ones((cfg.num_hidden_layers, 2, x.shape[0]), dtype=np.float32)
    hooks, calls = [], {}

    def make_hook(l):
        def hook(mod, inp, out):
            b = calls.get(l, 0)
            calls[l] = b + 1
            i, o = inp[0].detach(), out.detach()
            # factor per sample = out / in wherever in != 0 (it is constant over a sample)
            num = (o * i).flatten(1).sum(1)
            den = (i * i).flatten(1).sum(1)
            scales[l, b] = (num / den).numpy()
        return hook

    for l, layer in enumerate(m.layers):
        hooks.append(layer.drop_path.register_forward_hook(make_hook(l)))
    torch.manual_seed(seed)
    out = m(torch.from_numpy(x))
    loss = sum((out.hidden_states[t] * torch.from_numpy(d)).sum() for t, d in zip(cfg.taps, dtaps))
    loss.backward()
    for h in hooks:
        h.remove()
    grads = {}
    for k, p in m.named_parameters():
        k4 = to_v4(k)
        if "mask_token" in k4 or k4.startswith("pooler."):
            continue
        grads[k4] = (torch.zeros_like(p) if p.grad is None else p.grad).numpy().copy()
    taps = [out.hidden_states[t].detach().numpy() for t in cfg.taps]
    return taps, grads, scales


def stats(a):
    a64 = a.astype(np.float64)
    return np.array([np.sqrt((a64 ** 2).sum()), np.abs(a64).max()], dtype=np.float64)


def sample_stride(g, size: int) -> int:
    """Stride of a gradient's committed sample: `stride`, except that files carrying `small_full` store every tensor of at
    most that many elements whole (a 768-element bias sampled every 499th element is two numbers - no statistic)."""
    if "small_full" in g and size <= int(g["small_full"][0]):
        return 1
    return int(g["stride"][0])


def emit(name, cfg, B, size, wseed, xseed, gseed, stride, small_full=0):
    w = synth.synth_weights(cfg, wseed)
    x = synth.synth_images(B, size, size, seed=xseed, kind="uniform" if size < 224 else "doc")
    N = cfg.tokens(size, size)
    dtaps = upstream(cfg, B, N, gseed)
    rec = dict(geometry=np.array([cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.intermediate_size,
                                  cfg.patch_size, cfg.image_size, B, size], dtype=np.int64),
               seeds=np.array([wseed, xseed, gseed], dtype=np.int64), stride=np.array([stride], dtype=np.int64),
               taps=np.array(cfg.taps, dtype=np.int64), drop_path_rate=np.array([DROP_PATH_RATE]))
    if small_full:
        rec["small_full"] = np.array([small_full], dtype=np.int64)
    for mode, train in (("eval", False), ("train", True)):
        # train mode: first torch seed >= 1000 + gseed under which HF drops at least one (layer, branch, sample)
        for rng_seed in range(1000 + gseed, 1100 + gseed):
            taps, grads, scales = run(cfg, w, x, dtaps, train, seed=rng_seed)
            if not train or (scales == 0.0).any():
                break
        rec[f"{mode}_rng_seed"] = np.array([rng_seed], dtype=np.int64)
        rec[f"{mode}_drop_scales"] = scales
        for t, a in zip(cfg.taps, taps):
            rec[f"{mode}_tap{t}_sample"] = a.reshape(-1)[::max(stride, 7)].copy()
        for k, g in grads.items():
            rec[f"{mode}_grad/{k}"] = g.reshape(-1)[::sample_stride(rec, g.size)].copy()
            rec[f"{mode}_gstat/{k}"] = stats(g)
        if train:
            assert set(np.unique(np.round(scales, 4))) - {0.0, 1.0} != set() or cfg.num_hidden_layers < 2, \
                "no stochastic-depth factor other than 0 / 1 captured"
            print(name, "dropped (layer, branch, sample):", [tuple(int(i) for i in ix) for ix in np.argwhere(scales == 0.0)])
    np.savez_compressed(os.path.join(OUT, name), **rec)
    print(name, len(rec), "arrays")


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    if only in ("", "g6"):
        emit("g6_grad_micro.npz", cfgs.vit_micro(), 4, 64, wseed=7, xseed=99, gseed=5, stride=3)
    if only in ("", "g7"):
        emit("g7_grad_tiny.npz", cfgs.vit_tiny(), 2, 224, wseed=1, xseed=1234, gseed=6, stride=53)
    if only in ("", "g8"):
        # configs[2]'s own geometry (C = 768, 12 heads, F = 3072); the same weight / image seeds as g2_base.npz
        emit("g8_grad_base.npz", cfgs.vit_base(), 2, 224, wseed=0, xseed=1234, gseed=8, stride=499, small_full=8192)


if __name__ == "__main__":
    main()
