#!/usr/bin/env python3
"""The peeled-tail kernel of the bf16 GEMM (gemm_bf16_tail: <= 64 rows, one wave per 32 x 32 tile over all of K) on the ViT-L
shapes, M = 16 rows, behind a main-part GEMM of the same weight (so the weight is as warm as it is inside the model)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import _lib, ops  # noqa: E402
dev = "cuda"
for name, N, K, epi in (("qkv", 3072, 1024, _lib.EPI_BIAS), ("o_proj", 1024, 1024, _lib.EPI_SCALE_RESID), ("fc1", 4096, 1024, _lib.EPI_BIAS_GELU),
                        ("fc2", 1024, 4096, _lib.EPI_SCALE_RESID)):
    M = 16384
    x = torch.randn(M + 16, K, device=dev).to(torch.bfloat16)
    w = (0.05 * torch.randn(N, K, device=dev)).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    lam, r = torch.rand(N, device=dev), torch.randn(M + 16, N, device=dev)
    kw = dict(lam=lam, residual=r, out=r) if epi == _lib.EPI_SCALE_RESID else {}
    kt = dict(lam=lam, residual=r[M:], out=r[M:]) if epi == _lib.EPI_SCALE_RESID else {}
    xm, xt = x[:M], x[M:]
    for _ in range(3):
        ops.linear_bf16(xm, w, b, epilogue=epi, **({k: (v[:M] if v.dim() == 2 else v) for k, v in kw.items()}))
        ops.linear_bf16(xt, w, b, epilogue=epi, **kt)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tm, tt = 0.0, 0.0
    for _ in range(20):
        ev[0].record()
        ops.linear_bf16(xm, w, b, epilogue=epi, **({k: (v[:M] if v.dim() == 2 else v) for k, v in kw.items()}))
        ev[1].record()
        ops.linear_bf16(xt, w, b, epilogue=epi, **kt)
        ev[2].record()
        torch.cuda.synchronize()
        tm += ev[0].elapsed_time(ev[1]); tt += ev[1].elapsed_time(ev[2])
    print(f"{name:7s} N={N:5d} K={K:5d}: main {tm / 20 * 1e3:7.1f} us   tail(16 rows) {tt / 20 * 1e3:6.1f} us")
