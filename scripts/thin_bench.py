#!/usr/bin/env python3
"""Serving-size fp32 GEMMs (M = B*197): thin tiling (LDIT_GEMM_TILE=4) vs the 64x64 tiling (2), with the weights warm
(one W, L2-resident) and cold (a rotation of W buffers larger than L2 + Infinity Cache, as in a real forward)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import _lib, ops  # noqa: E402

dev = "cuda:0"
shapes = [(197, 2304, 768), (197, 768, 768), (197, 3072, 768), (197, 768, 3072)]
Bs = [int(a) for a in sys.argv[1:]] or [1]
for B in Bs:
    for (M0, N, K) in shapes:
        M = M0 * B
        x = torch.randn(M, K, device=dev)
        nbuf = max(2, int(600e6 // (N * K * 4)))
        ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(nbuf)]
        b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        line = f"M={M:5d} N={N:5d} K={K:5d}:"
        for tile in ("4", "2"):
            _lib.set_switch("LDIT_GEMM_TILE", tile)
            for mode in ("warm", "cold"):
                reps = 40
                for i in range(5):
                    ops.linear(x, ws[0] if mode == "warm" else ws[i % nbuf], b, out=y)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(reps):
                    ops.linear(x, ws[0] if mode == "warm" else ws[(i + 5) % nbuf], b, out=y)
                e1.record()
                torch.cuda.synchronize()
                line += f"  tile{tile} {mode} {e0.elapsed_time(e1) / reps * 1e3:7.1f} us"
        print(line, flush=True)
