#!/usr/bin/env python3
"""In-kernel phase timing of the GEMM (diagnostic build libldit_hip_dbg.so; GPU box only).
Per block: prologue / main loop / epilogue issue / store drain in cycles, plus wall-clock start/end per CU."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "layoutdit_amd", "csrc", "build", os.environ.get("DBGLIB", "libldit_hip_dbg.so")))
lib.ldit_dbg_linear_stamps.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p] * 4
dev = "cuda:0"
M = 12608
for name, N, K, epi in (("qkv", 2304, 768, 0), ("o_proj", 768, 768, 2), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    mode = os.environ.get("DATA", "normal")
    if mode == "zero":
        x.zero_(); w.zero_()
    elif mode == "const":
        x.fill_(1.0); w.fill_(0.5)
    elif mode == "uniform":
        x.uniform_(-1, 1); w.uniform_(-1, 1)
    lam = torch.rand(N, device=dev); r = torch.randn(M, N, device=dev); y = torch.empty(M, N, device=dev)
    bm = 304 if os.environ.get('LDIT_GEMM_TILE', '3') == '3' else 320
    nblk = ((M + bm - 1) // bm) * ((N + 127) // 128)
    st = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
    for _ in range(int(os.environ.get('REPS', 3))):
        rc = lib.ldit_dbg_linear_stamps(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, epi,
                                        lam.data_ptr(), r.data_ptr(), st.data_ptr(), None)
        assert rc == 0
    torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(nblk, 8).astype(np.int64)
    real0, real1 = s[:, 0], s[:, 1]
    t0 = real0.min()
    print(f"== {name} N={N} K={K}: {nblk} blocks; kernel span {(real1.max() - t0) / 100:.1f} us (100 MHz clock)")
    for i, lab in enumerate(("prologue", "mainloop", "epi-issue", "store-drain")):
        v = s[:, 2 + i]
        print(f"   {lab:12s} cycles: median {np.median(v):9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f}")
    dur = (real1 - real0) / 100.0
    print(f"   block wall us: median {np.median(dur):.1f} min {dur.min():.1f} max {dur.max():.1f}; "
          f"start offsets us: p50 {np.median(real0 - t0) / 100:.1f} max {(real0.max() - t0) / 100:.1f}")
    # per-CU timeline: gaps between consecutive blocks on the same CU
    cu = s[:, 6]
    gaps = []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        o = idx[np.argsort(real0[idx])]
        for a, bb in zip(o[:-1], o[1:]):
            gaps.append((real0[bb] - real1[a]) / 100.0)
    if gaps:
        print(f"   same-CU gap between blocks us: median {np.median(gaps):.2f} max {np.max(gaps):.2f} (n={len(gaps)}); distinct CUs {len(np.unique(cu))}")
    clk = (s[:, 2] + s[:, 3] + s[:, 4] + s[:, 5]) / np.maximum(dur, 1e-9) / 1e3
    print(f"   in-kernel clock GHz: median {np.median(clk):.2f}")
