#!/usr/bin/env python3
"""fp32 fused attention on the headline shape (GPU box): ViT-B/16 224x224 bs=64, and ViT-L/16 512x512 bs=16."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import ops  # noqa: E402
for B, N, H in ((64, 197, 12), (16, 1025, 16)):
    C = 64 * H
    qkv = torch.randn(B, N, 3 * C, device="cuda")
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    for _ in range(5):
        ops.attention(q, k, v, H)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.attention(q, k, v, H)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    fl = 4.0 * B * H * N * N * 64
    print(f"B={B} N={N} H={H}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s ({fl / ms / 1e9 / 157.3 * 100:.1f}% of 157.3 TF)")
