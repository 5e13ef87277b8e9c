#!/usr/bin/env python3
"""bf16 fused attention on the two BASELINE sequence lengths (GPU box): ViT-L/16 512x512 bs=16 and ViT-B/16 224x224 bs=64.
Both entry modes: plain (the kernel applies scale * log2 e) and PRE (scale = 0: q already carries that factor, as the packed
inference path delivers it), each checked against a float64 reference on a slice before it is timed.  DATA=model (default)
draws q, k, v with the spread the encoder produces (|score| of a few units); DATA=randn is unit-variance noise (scores up
to +-30: the rescale branch fires often)."""
import ctypes as C
import math
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import _lib  # noqa: E402
lib = _lib.load()
amp = 1.0 if os.environ.get("DATA", "model") == "randn" else 0.35
for B, N, H in ((16, 1025, 16), (64, 197, 12), (32, 197, 12)):
    Cc = 64 * H
    qkv = (amp * torch.randn(B, N, 3 * Cc, device="cuda")).to(torch.bfloat16)
    q, k, v = qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:]
    c = 0.125 * math.log2(math.e)
    qpre = torch.cat([(q.float() * c).to(torch.bfloat16), qkv[..., Cc:]], dim=-1).contiguous()
    o = torch.empty(B, N, Cc, device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream

    def run(src, scale):
        _lib.check(lib.ldit_attention_bf16(src.data_ptr(), src.data_ptr() + 2 * Cc, src.data_ptr() + 4 * Cc, o.data_ptr(), B, N, H, 64,
                                           3 * Cc, 3 * Cc, 3 * Cc, Cc, scale, st))
    # reference on image 0, heads 0..1
    qq, kk, vv = (t[0, :, :128].double().view(N, 2, 64).transpose(0, 1) for t in (q, k, v))
    ref = (torch.softmax(qq @ kk.transpose(-1, -2) * 0.125, -1) @ vv).transpose(0, 1).reshape(N, 128)
    for name, src, scale in (("plain", qkv, 0.125), ("PRE", qpre, 0.0)):
        run(src, scale)
        torch.cuda.synchronize()
        err = float((o[0, :, :128].double() - ref).norm() / ref.norm())
        for _ in range(5):
            run(src, scale)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run(src, scale)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        fl = 4.0 * B * H * N * N * 64
        print(f"B={B} N={N} H={H} {name:5s}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s ({fl / ms / 1e9 / 2500 * 100:.1f}% of 2.5 PF)  rel-L2 vs f64 {err:.2e}")
