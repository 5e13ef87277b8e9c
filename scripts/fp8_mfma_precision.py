#!/usr/bin/env python3
"""How many bits does the fp8 MFMA keep inside one instruction?  Row k=0 carries one big product (256 * 1), the other 127
products of the k-tile are 2^-6 * 2^-s; the exact sum is representable in fp32 for every s.  Prints got / exact per s."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import _lib, ops  # noqa: E402
F8 = torch.float8_e4m3fn
M, N, K = 32, 32, 128
x = torch.full((M, K), 2.0 ** -6); x[:, 0] = 256.0
w = torch.zeros(N, K); w[:, 0] = 1.0
for s in range(10):
    w[s, 1:] = 2.0 ** -s
xq, wq = x.cuda().to(F8), w.cuda().to(F8)
r = torch.zeros(M, N, device="cuda"); lam = torch.ones(N, device="cuda")
y = ops.linear_fp8(xq, wq, 1.0, None, epilogue=_lib.EPI_SCALE_RESID, lam=lam, residual=r).cpu()
for s in range(10):
    exact = 256.0 + 127 * 2.0 ** (-6 - s)
    print(f"small/big = 2^-{14 + s}: got {y[0, s].item():.10f} exact {exact:.10f} lost {(exact - y[0, s].item()) / (127 * 2.0 ** (-6 - s)) * 100:.1f}% of the small terms")

# second sweep: ratios 2^-1 .. 2^-13 (big = 2^-6 * 2^t against 127 terms of 2^-6, weights 1)
for t in range(1, 15):
    x = torch.full((M, K), 2.0 ** -6); x[:, 0] = 2.0 ** (t - 6)
    w = torch.ones(N, K)
    y = ops.linear_fp8(x.cuda().to(F8), w.cuda().to(F8), 1.0, None, epilogue=_lib.EPI_SCALE_RESID, lam=lam,
                       residual=torch.zeros(M, N, device="cuda")).cpu()
    exact = 2.0 ** (t - 6) + 127 * 2.0 ** -6
    print(f"small/big = 2^-{t}: got {y[0, 0].item():.10f} exact {exact:.10f} lost {(exact - y[0, 0].item()) / 2.0 ** -6:.3f} small terms")
