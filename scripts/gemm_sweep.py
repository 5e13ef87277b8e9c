#!/usr/bin/env python3
"""Fixed-overhead vs per-k-tile time of the GEMM: sweep K at M=12608, N=768 for each epilogue (GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import _lib, ops  # noqa: E402
dev = "cuda:0"
M = 12608
N = int(os.environ.get("N", 768))
for epi, name in ((_lib.EPI_BIAS, "bias"), (_lib.EPI_BIAS_GELU, "gelu"), (_lib.EPI_SCALE_RESID, "resid")):
    res = []
    for K in (32, 64, 256, 768, 1536, 3072):
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev); lam = torch.rand(N, device=dev); r = torch.randn(M, N, device=dev)
        out = torch.empty(M, N, device=dev)
        kw = dict(epilogue=epi, out=out)
        if epi == _lib.EPI_SCALE_RESID:
            kw.update(lam=lam, residual=r)
        for _ in range(3):
            ops.linear(x, w, b, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.linear(x, w, b, **kw)
        e1.record(); torch.cuda.synchronize()
        res.append((K, e0.elapsed_time(e1) / 20 * 1e3))
    (k0, t0), (k1, t1) = res[3], res[5]
    per = (t1 - t0) / ((k1 - k0) / 32)
    print(name, " ".join(f"K={k}:{t:.1f}us" for k, t in res), f"| per-k-tile {per:.2f} us, fixed {t0 - per * k0 / 32:.1f} us")
