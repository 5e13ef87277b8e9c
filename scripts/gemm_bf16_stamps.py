#!/usr/bin/env python3
"""In-kernel phase timing of the bf16 GEMM main kernel (diagnostic build csrc/build/libldit_hip_dbg.so; GPU box only):
per workgroup, wave 0 - cycles in the k-loop, of which waiting for its own LDS-DMA pieces and for the other waves at the
per-k-tile hand-over, and cycles in the epilogue."""
import ctypes as C
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["LDIT_GEMM_BF16_TILE"] = os.environ.get("LDIT_GEMM_BF16_TILE", "3")
import sys
lib = C.CDLL(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "layoutdit_amd", "csrc", "build", "libldit_hip_dbg.so"))
vp, i64 = C.c_void_p, C.c_int64
lib.ldit_linear_bf16.argtypes = [vp, i64, vp, vp, vp, i64, i64, i64, i64, C.c_int32, vp, vp, vp, vp]
lib.ldit_dbg_set_gemm_bf16_stamps.argtypes = [vp]
M = int(os.environ.get("M", 16384))
for name, N, K, epi in (("qkv", 3072, 1024, 0), ("o_proj", 1024, 1024, 2), ("fc1", 4096, 1024, 1), ("fc2", 1024, 4096, 2)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); lam = torch.rand(N, device="cuda"); r = torch.randn(M, N, device="cuda")
    y = r if epi == 2 else torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    nwg = ((M + 255) // 256) * ((N + 255) // 256)
    st = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
    assert lib.ldit_dbg_set_gemm_bf16_stamps(st.data_ptr()) == 0
    for _ in range(30):
        rc = lib.ldit_linear_bf16(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, M, N, K, epi,
                                  lam.data_ptr(), r.data_ptr() if epi == 2 else None, None, None)
        assert rc == 0
    torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(nwg, 2, 8)
    nk = K // 64
    ideal = nk * 2 * 32 * 32          # two waves per SIMD x 32 MFMAs x 32 cycles per k-tile
    for wv, tag in ((0, "wave 0 (loader) "), (1, "wave 4 (partner)")):
        med = np.median(s[:, wv, :], axis=0)
        ghz = med[0] / max(med[5], 1) * 0.1
        print(f"{name:7s} N={N} K={K} {tag}: {nwg} tiles; median cycles  k-loop {med[0]:8.0f} = {med[0] / nk:5.0f}/k-tile (MFMA-bound {ideal}, {ideal / med[0] * 100:.0f} %)  "
              f"own-DMA wait {med[1] / nk:5.0f}/k-tile  barrier {med[2] / nk:5.0f}/k-tile  epilogue {med[3]:7.0f}  entry->loop {med[4]:6.0f}  clock {ghz:.2f} GHz")
