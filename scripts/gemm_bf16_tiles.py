#!/usr/bin/env python3
"""A/B of the bf16 / fp8 (DT=fp8) GEMM tilings on the layer shapes of the bench configs, interleaved rounds in ONE process
(LDIT_GEMM_BF16_TILE / LDIT_GEMM_FP8_TILE are read at every launch).  bf16: 2 = 128x128, 3 = 256x256, 4 = 192x256,
5 = 320x256; fp8: 2 = 128x128, 0 = 256x256, 3 = 192x256, 4 = 320x256; auto = the picker."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import _lib, ops  # noqa: E402
FP8 = os.environ.get("DT", "bf16") == "fp8"
ENV = "LDIT_GEMM_FP8_TILE" if FP8 else "LDIT_GEMM_BF16_TILE"
TILES = tuple(os.environ.get("TILES", "2,0,3,4,5,6,auto" if FP8 else "2,3,4,5,6,7,auto").split(","))
DT = torch.float8_e4m3fn if FP8 else torch.bfloat16
def run(x, w, b, **kw):
    return ops.linear_fp8(x, w, 0.01, b, **kw) if FP8 else ops.linear_bf16(x, w, b, **kw)

def shapes(M, C):
    F = 4 * C
    return [("qkv", M, 3 * C, C, _lib.EPI_BIAS), ("o_proj", M, C, C, _lib.EPI_SCALE_RESID), ("fc1", M, F, C, _lib.EPI_BIAS_GELU),
            ("fc2", M, C, F, _lib.EPI_SCALE_RESID)]

for M, C in ((64 * 197, 768), (32 * 197, 768), (16 * 1024, 1024)):
    for name, m, n, k, epi in shapes(M, C):
        x = torch.randn(m, k, device="cuda").to(DT); w = (torch.randn(n, k, device="cuda") * (1.0 if FP8 else 0.05)).to(DT)
        b = torch.randn(n, device="cuda"); lam = torch.rand(n, device="cuda"); r = torch.randn(m, n, device="cuda")
        kw = dict(epilogue=epi)
        if epi == _lib.EPI_SCALE_RESID:
            kw.update(lam=lam, residual=r, out=r)
        res = {t: [] for t in TILES}
        for rnd in range(5):
            for t in res:
                _lib.set_switch(ENV, None if t == "auto" else t)
                for _ in range(3): run(x, w, b, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): run(x, w, b, **kw)
                e1.record(); torch.cuda.synchronize()
                res[t].append(e0.elapsed_time(e1) / 20 * 1e3)
        _lib.set_switch(ENV, None)
        med = {t: statistics.median(v) for t, v in res.items()}
        fl = 2.0 * m * n * k
        print(f"M={m:6d} {name:7s} N={n:5d} K={k:5d}  " + "  ".join(f"{t}:{med[t]:7.1f}us" for t in res) +
              f"   best {fl / min(med.values()) / 1e6:7.1f} TF/s", flush=True)
