#!/usr/bin/env python3
"""bf16 MFMA GEMM on the ViT-L/16 512x512 bs=16 shapes (GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import _lib, ops  # noqa: E402
M = int(os.environ.get("M", 16 * 1025))
C = int(os.environ.get("C", 1024)); F = 4 * C
tot_f = tot_t = 0.0
for name, n, k, epi in (("qkv", 3 * C, C, _lib.EPI_BIAS), ("o_proj", C, C, _lib.EPI_SCALE_RESID), ("fc1", F, C, _lib.EPI_BIAS_GELU), ("fc2", C, F, _lib.EPI_SCALE_RESID)):
    x = torch.randn(M, k, device="cuda").to(torch.bfloat16); w = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(n, device="cuda"); lam = torch.rand(n, device="cuda"); r = torch.randn(M, n, device="cuda")
    kw = dict(epilogue=epi)
    if epi == _lib.EPI_SCALE_RESID:
        kw.update(lam=lam, residual=r, out=r)
    for _ in range(5):
        ops.linear_bf16(x, w, b, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.linear_bf16(x, w, b, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    fl = 2.0 * M * n * k
    tot_f += fl; tot_t += ms
    print(f"{name:8s} M={M} N={n} K={k}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s ({fl / ms / 1e9 / 2500 * 100:.1f}% of 2.5 PF)")
print(f"layer GEMMs: {tot_t * 1e3:.1f} us, {tot_f / tot_t / 1e9:.1f} TFLOP/s")
