#!/usr/bin/env python3
"""Known-good reference for the GEMM shapes of the bench configs: the vendor library behind torch (hipBLASLt / rocBLAS) on the same
random bf16 operands, plain C = A . W^T without epilogue, beside this library's kernel with its cheapest epilogue (bias, bf16 out).
Measurement only - nothing in the product calls a vendor GEMM (hand-written kernels).  Interleaved rounds, medians."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import _lib, ops  # noqa: E402

def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

shapes = [("ViT-L qkv", 16384, 3072, 1024), ("ViT-L o_proj", 16384, 1024, 1024), ("ViT-L fc1", 16384, 4096, 1024), ("ViT-L fc2", 16384, 1024, 4096),
          ("ViT-B qkv", 12608, 2304, 768), ("ViT-B o_proj", 12608, 768, 768), ("ViT-B fc1", 12608, 3072, 768), ("ViT-B fc2", 12608, 768, 3072),
          ("x3 qkv K'=2304", 12608, 2304, 2304), ("x3 fc2 K'=9216", 12608, 768, 9216), ("square", 4096, 4096, 4096), ("square", 8192, 8192, 8192)]
for name, m, n, k in shapes:
    x = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16)
    b = torch.zeros(n, device="cuda")
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    ours, ref = [], []
    for r in range(5):
        ours.append(t(lambda: ops.linear_bf16(x, w, b, out=out)))
        ref.append(t(lambda: torch.mm(x, w.t(), out=out)))
    a, v = statistics.median(ours), statistics.median(ref)
    fl = 2.0 * m * n * k
    print(f"{name:16s} M={m:6d} N={n:5d} K={k:5d}: this library {a:8.1f} us {fl / a / 1e6:7.1f} TF/s   vendor {v:8.1f} us {fl / v / 1e6:7.1f} TF/s   ratio {v / a:.2f}", flush=True)
