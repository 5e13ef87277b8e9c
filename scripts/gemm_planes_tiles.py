#!/usr/bin/env python3
"""A/B of the bf16 GEMM tilings on the split-fp32 layer GEMMs (ViT-B bs=64: M = 12 608; PLANES=2 -> three plane products,
PLANES=3 -> six), interleaved rounds in one process.  2 = 128x128, 3 = 256x256, 4 = 192x256, 5 = 320x256, auto = the picker."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import _lib, ops  # noqa: E402
PL = int(os.environ.get("PLANES", "2"))
TILES = tuple(os.environ.get("TILES", "2,3,4,5,auto").split(","))
ORDER = os.environ.get("ORDER_AB") == "1"        # A/B of the segment order (outermost / innermost) on the picker's tiling instead
if ORDER:
    TILES = ("outer", "inner")
M, C = int(os.environ.get("M", 64 * 197)), 768
F = 4 * C
for name, n, k, epi in (("qkv", 3 * C, C, _lib.EPI_BIAS), ("o_proj", C, C, _lib.EPI_SCALE_RESID), ("fc1", F, C, _lib.EPI_BIAS_GELU),
                        ("fc2", C, F, _lib.EPI_SCALE_RESID)):
    xp = ops.split_planes(torch.randn(M, k, device="cuda"), PL)
    wp = ops.split_planes(torch.randn(n, k, device="cuda") * 0.05, PL)
    b = torch.randn(n, device="cuda"); lam = torch.rand(n, device="cuda"); r = torch.randn(M, n, device="cuda")
    kw = dict(epilogue=epi)
    if epi == _lib.EPI_SCALE_RESID:
        kw.update(lam=lam, residual=r, out=r)
    else:
        kw.update(out=ops.linear_planes(xp, wp, PL, b, **kw))
    res = {t: [] for t in TILES}
    for rnd in range(5):
        for t in res:
            if ORDER:
                _lib.set_switch("LDIT_GEMM_SEG_ORDER", "0" if t == "outer" else "1")
            else:
                _lib.set_switch("LDIT_GEMM_BF16_TILE", None if t == "auto" else t)
            for _ in range(3): ops.linear_planes(xp, wp, PL, b, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.linear_planes(xp, wp, PL, b, **kw)
            e1.record(); torch.cuda.synchronize()
            res[t].append(e0.elapsed_time(e1) / 20 * 1e3)
    _lib.set_switch("LDIT_GEMM_BF16_TILE", None)
    _lib.set_switch("LDIT_GEMM_SEG_ORDER", None)
    med = {t: statistics.median(v) for t, v in res.items()}
    fl = 2.0 * M * n * k * (3 if PL == 2 else 6)
    print(f"planes={PL} M={M:6d} {name:7s} N={n:5d} K={k:5d}  " + "  ".join(f"{t}:{med[t]:7.1f}us" for t in res) +
          f"   best {fl / min(med.values()) / 1e6:7.1f} bf16 TF/s", flush=True)
