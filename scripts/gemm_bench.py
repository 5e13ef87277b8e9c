#!/usr/bin/env python3
"""Time the fp32 MFMA GEMM on the ViT-B/16 bs=64 shapes (run on the GPU box).  Prints TFLOP/s per shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import _lib, ops  # noqa: E402

M = int(os.environ.get("M", 64 * 197))
SHAPES = [("qkv", M, 2304, 768, _lib.EPI_BIAS), ("o_proj", M, 768, 768, _lib.EPI_SCALE_RESID),
          ("fc1", M, 3072, 768, _lib.EPI_BIAS_GELU), ("fc2", M, 768, 3072, _lib.EPI_SCALE_RESID)]
dev = "cuda:0"
torch.manual_seed(0)
tot_f, tot_t = 0.0, 0.0
for name, m, n, k, epi in SHAPES:
    x = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) * 0.05
    b = torch.randn(n, device=dev)
    lam = torch.rand(n, device=dev)
    r = torch.randn(m, n, device=dev)
    out = torch.empty(m, n, device=dev)
    kw = dict(epilogue=epi, out=out)
    if epi == _lib.EPI_SCALE_RESID:
        kw.update(lam=lam, residual=r)
    for _ in range(3):
        ops.linear(x, w, b, **kw)
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.linear(x, w, b, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * m * n * k
    tot_f += fl
    tot_t += ms
    print(f"{name:8s} M={m} N={n} K={k}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s  ({fl / ms / 1e9 / 157.3 * 100:.1f}% of 157.3)")
print(f"layer GEMMs: {tot_t * 1e3:.1f} us, {tot_f / tot_t / 1e9:.1f} TFLOP/s")
