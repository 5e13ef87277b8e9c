#!/usr/bin/env python3
"""fp32 panel GEMM: persistent tile loop (LDIT_GEMM_PERSIST=1) against one workgroup per tile (the default), same process,
interleaved rounds, on the ViT-B/16 bs=64 layer shapes; outputs compared bit for bit."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import _lib, ops  # noqa: E402

M = int(os.environ.get("M", 64 * 197))
SHAPES = [("qkv", M, 2304, 768, _lib.EPI_BIAS), ("o_proj", M, 768, 768, _lib.EPI_SCALE_RESID),
          ("fc1", M, 3072, 768, _lib.EPI_BIAS_GELU), ("fc2", M, 768, 3072, _lib.EPI_SCALE_RESID)]

def t(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

tot = {"persist": 0.0, "classic": 0.0}
for name, m, n, k, epi in SHAPES:
    x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * 0.05; b = torch.randn(n, device="cuda")
    lam = torch.rand(n, device="cuda"); r = torch.randn(m, n, device="cuda")
    kw = dict(epilogue=epi)
    if epi == _lib.EPI_SCALE_RESID:
        kw.update(lam=lam, residual=r)
    outs, times = {}, {"persist": [], "classic": []}
    for mode, val in (("persist", "1"), ("classic", None)):
        _lib.set_switch("LDIT_GEMM_PERSIST", val)
        outs[mode] = ops.linear(x, w, b, **kw).clone()
    assert torch.equal(outs["persist"], outs["classic"]), name
    scratch = torch.empty(m, n, device="cuda")
    for _ in range(5):
        for mode, val in (("persist", "1"), ("classic", None)):
            _lib.set_switch("LDIT_GEMM_PERSIST", val)
            times[mode].append(t(lambda: ops.linear(x, w, b, out=scratch, **kw)))
    _lib.set_switch("LDIT_GEMM_PERSIST", None)
    a, c = statistics.median(times["persist"]), statistics.median(times["classic"])
    tot["persist"] += a; tot["classic"] += c
    fl = 2.0 * m * n * k
    print(f"{name:7s} M={m} N={n} K={k}: persistent {a:7.1f} us ({fl / a / 1e6:6.1f} TF/s)   one tile per workgroup {c:7.1f} us ({fl / c / 1e6:6.1f} TF/s)   {c / a:.3f}x   bit-equal", flush=True)
print(f"layer: persistent {tot['persist']:.1f} us, classic {tot['classic']:.1f} us ({tot['classic'] / tot['persist']:.3f}x)")
