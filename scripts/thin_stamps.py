#!/usr/bin/env python3
"""In-kernel phase timing of the serving-size GEMM (diagnostic build libldit_hip_dbg.so, `make -C layoutdit_amd/csrc dbg`;
GPU box only).  Per block: prologue (to the first fragments) / main loop / DMA drain in shader cycles + 100 MHz wall stamps."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "layoutdit_amd", "csrc", "build", "libldit_hip_dbg.so"))
lib.ldit_dbg_linear_stamps.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p] * 4
dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 197
for name, N, K, epi in (("qkv", 2304, 768, 0), ("o_proj", 768, 768, 2), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    lam = torch.rand(N, device=dev); r = torch.randn(M, N, device=dev); y = torch.empty(M, N, device=dev)
    nbm, nbn = (M + 31) // 32, (N + 31) // 32
    nblk = nbm * ((nbn + 7) // 8) * 8
    st = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
    for _ in range(3):
        st.zero_()
        rc = lib.ldit_dbg_linear_stamps(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, epi,
                                        lam.data_ptr(), r.data_ptr(), st.data_ptr(), None)
        assert rc == 0
    torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(nblk, 8).astype(np.int64)
    s = s[s[:, 0] > 0]
    real0, real1 = s[:, 0], s[:, 1]
    t0 = real0.min()
    nk = K // 32
    print(f"== {name} M={M} N={N} K={K}: {len(s)} blocks; kernel span {(real1.max() - t0) / 100:.1f} us (100 MHz clock)")
    for i, lab in enumerate(("prologue", "mainloop", "drain")):
        v = s[:, 2 + i]
        extra = f"  = {np.median(v) / nk:6.0f} cycles per k-tile" if i == 1 else ""
        print(f"   {lab:10s} cycles: median {np.median(v):9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f}{extra}")
    dur = (real1 - real0) / 100.0
    print(f"   block wall us: median {np.median(dur):.1f} min {dur.min():.1f} max {dur.max():.1f}; "
          f"start offsets us: p50 {np.median(real0 - t0) / 100:.1f} max {(real0.max() - t0) / 100:.1f}; distinct CUs {len(np.unique(s[:, 6]))}")
    simd = [tuple((int(v) >> (4 * w)) & 3 for w in range(4)) for v in s[:, 5]]
    distinct = np.array([len(set(t)) for t in simd])
    print(f"   SIMDs used by the 4 waves of a block: " + ", ".join(f"{k} distinct: {int((distinct == k).sum())}" for k in (1, 2, 3, 4)) + f"; e.g. {simd[0]} {simd[1]}")
    clk = (s[:, 2] + s[:, 3]) / np.maximum(dur, 1e-9) / 1e3
    print(f"   in-kernel clock GHz: median {np.median(clk):.2f}")
