#!/usr/bin/env python3
"""Where does the fp8 build lose its accuracy under the activation-outlier stress (tests/test_gpu_lowp_pinning.py: three LayerNorm
gamma channels and two fc1 rows x 60 in every layer, ViT-B)?  A torch emulation of the build's operand rounding - per-tensor e4m3
activations (scale = amax / 448 of the same batch), per-output-channel e4m3 weights, bf16 q|k|v and P, fp32 everything else - with
the four GEMMs of a layer switched between fp8 and bf16 operands one group at a time, against the fp32 forward.  Measurement only."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import config as cfgs, synth  # noqa: E402

dev = "cuda"
cfg = cfgs.vit_base()
GAIN = float(os.environ.get("GAIN", 60))
w = synth.synth_weights(cfg, 4)
ch = [5, 77, 130]
for l in range(cfg.num_hidden_layers):
    p = f"encoder.layer.{l}."
    for k in ("layernorm_before.weight", "layernorm_after.weight"):
        w[p + k] = w[p + k].copy(); w[p + k][ch] *= GAIN
    w[p + "intermediate.dense.weight"] = w[p + "intermediate.dense.weight"].copy(); w[p + "intermediate.dense.weight"][[11, 300]] *= GAIN
W = {k: torch.from_numpy(v).to(dev) for k, v in w.items()}
x = torch.from_numpy(synth.synth_images(1, 224, 224, seed=1234)).to(dev)
C, H, L = cfg.hidden_size, cfg.num_attention_heads, cfg.num_hidden_layers
F8 = torch.float8_e4m3fn

def q_act(t, mode):
    if mode == "fp8":
        s = t.abs().amax() / 448.0
        return (t / s).to(F8).float() * s
    if mode == "bf16":
        return t.bfloat16().float()
    return t

def q_w(t, mode):
    if mode == "fp8":
        s = t.abs().amax(dim=1, keepdim=True) / 448.0
        return (t / s).to(F8).float() * s
    if mode == "bf16":
        return t.bfloat16().float()
    return t

def forward(modes):           # modes: dict group -> "fp8" | "bf16" | "f32" for groups qkv, o, fc1, fc2; "attn": bf16 | f32
    B = x.shape[0]
    pw = W["embeddings.patch_embeddings.projection.weight"]
    e = torch.nn.functional.conv2d(x, pw, W["embeddings.patch_embeddings.projection.bias"], stride=16).flatten(2).transpose(1, 2)
    h = torch.cat([W["embeddings.cls_token"].expand(B, 1, C), e], 1) + W["embeddings.position_embeddings"]
    taps = {}
    for l in range(L):
        p = f"encoder.layer.{l}."
        y = torch.nn.functional.layer_norm(h, (C,), W[p + "layernorm_before.weight"], W[p + "layernorm_before.bias"], 1e-12)
        ya = q_act(y, modes["qkv"])
        q = ya @ q_w(W[p + "attention.attention.query.weight"], modes["qkv"]).t() + W[p + "attention.attention.query.bias"]
        k = ya @ q_w(W[p + "attention.attention.key.weight"], modes["qkv"]).t()
        v = ya @ q_w(W[p + "attention.attention.value.weight"], modes["qkv"]).t() + W[p + "attention.attention.value.bias"]
        if modes["attn"] == "bf16":
            q, k, v = q.bfloat16().float(), k.bfloat16().float(), v.bfloat16().float()
        sh = lambda t: t.view(B, -1, H, C // H).transpose(1, 2)
        a = torch.softmax(sh(q) @ sh(k).transpose(-1, -2) / 8.0, -1)
        if modes["attn"] == "bf16":
            a = a.bfloat16().float()
        o = (a @ sh(v)).transpose(1, 2).reshape(B, -1, C)
        o = q_act(o, modes["o"]) @ q_w(W[p + "attention.output.dense.weight"], modes["o"]).t() + W[p + "attention.output.dense.bias"]
        h = h + W[p + "lambda_1"] * o
        y2 = torch.nn.functional.layer_norm(h, (C,), W[p + "layernorm_after.weight"], W[p + "layernorm_after.bias"], 1e-12)
        g = torch.nn.functional.gelu(q_act(y2, modes["fc1"]) @ q_w(W[p + "intermediate.dense.weight"], modes["fc1"]).t() + W[p + "intermediate.dense.bias"])
        z = q_act(g, modes["fc2"]) @ q_w(W[p + "output.dense.weight"], modes["fc2"]).t() + W[p + "output.dense.bias"]
        h = h + W[p + "lambda_2"] * z
        if l + 1 in cfg.taps:
            taps[l + 1] = h.clone()
    return taps

ref = forward(dict(qkv="f32", o="f32", fc1="f32", fc2="f32", attn="f32"))
def report(name, modes):
    t = forward(modes)
    errs = [float((t[k].double() - ref[k].double()).norm() / ref[k].double().norm()) for k in cfg.taps]
    print(f"{name:44s} rel-L2 per tap " + "  ".join(f"{e:.3e}" for e in errs), flush=True)

with torch.no_grad():
    print(f"outlier gain {GAIN}")
    report("all bf16 operands", dict(qkv="bf16", o="bf16", fc1="bf16", fc2="bf16", attn="bf16"))
    report("all four GEMMs fp8 (the build)", dict(qkv="fp8", o="fp8", fc1="fp8", fc2="fp8", attn="bf16"))
    report("q|k|v on bf16 operands, rest fp8", dict(qkv="bf16", o="fp8", fc1="fp8", fc2="fp8", attn="bf16"))
    report("o_proj on bf16, rest fp8", dict(qkv="fp8", o="bf16", fc1="fp8", fc2="fp8", attn="bf16"))
    report("fc1 on bf16, rest fp8", dict(qkv="fp8", o="fp8", fc1="bf16", fc2="fp8", attn="bf16"))
    report("fc2 on bf16, rest fp8", dict(qkv="fp8", o="fp8", fc1="fp8", fc2="bf16", attn="bf16"))
    report("ONLY q|k|v fp8", dict(qkv="fp8", o="bf16", fc1="bf16", fc2="bf16", attn="bf16"))
    report("ONLY fc1 fp8", dict(qkv="bf16", o="bf16", fc1="fp8", fc2="bf16", attn="bf16"))
    report("ONLY fc2 fp8", dict(qkv="bf16", o="bf16", fc1="bf16", fc2="fp8", attn="bf16"))
    report("ONLY o_proj fp8", dict(qkv="bf16", o="fp8", fc1="bf16", fc2="bf16", attn="bf16"))
