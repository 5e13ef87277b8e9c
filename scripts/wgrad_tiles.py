#!/usr/bin/env python3
"""wgrad form of gemm_bf16_tr (both operands reduction-major, K = tokens split over workgroups) on the ViT-B bs=64 shapes: time of
the GEMM + the slab reduction per (tile, splits) choice.  LDIT_GEMM_BF16_TR_TILE: 2 = 128 x 128 (4 waves, two per CU), 3 = 256 x 256, 6 = 256 x 128 (8 waves)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import _lib, ops  # noqa: E402

K = int(os.environ.get("TOKENS", 64 * 197))
lib = _lib.load()
def t(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

zero = torch.zeros(4096, device="cuda", dtype=torch.uint8)
for name, M, N in (("W2 [768 x 3072]", 768, 3072), ("W1 [3072 x 768]", 3072, 768), ("Wo [768 x 768]", 768, 768), ("Wqkv [2304 x 768]", 2304, 768)):
    a = torch.randn(K, M, device="cuda").to(torch.bfloat16)
    w = torch.randn(K, N, device="cuda").to(torch.bfloat16)
    ref = None
    res = []
    for tile in ("2", "3", "6"):
        _lib.set_switch("LDIT_GEMM_BF16_TR_TILE", tile)
        for splits in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12):
            nk = (K + 63) // 64
            if ((nk + splits - 1) // splits) * (splits - 1) >= nk:
                continue
            slabs = torch.empty((splits, M, N), device="cuda", dtype=torch.float32)
            out = torch.empty((M, N), device="cuda", dtype=torch.float32)
            def run():
                rc = lib.ldit_linear_bf16_tr(a.data_ptr(), M, 1, w.data_ptr(), N, slabs.data_ptr(), N, M, N, K, _lib.EPI_F32, None, splits, zero.data_ptr(), torch.cuda.current_stream().cuda_stream)
                assert rc == 0, lib.ldit_last_error()
                if splits > 1:
                    lib.ldit_reduce_slabs_f32(slabs.data_ptr(), out.data_ptr(), M * N, splits, torch.cuda.current_stream().cuda_stream)
            us = statistics.median(t(run) for _ in range(3))
            got = (out if splits > 1 else slabs[0]).clone()
            if ref is None:
                ref = (a[:, :64].double().t() @ w.double())
            err = float((got[:64].double() - ref).norm() / ref.norm())
            assert err < 1e-4, (name, tile, splits, err)
            res.append((us, tile, splits))
    _lib.set_switch("LDIT_GEMM_BF16_TR_TILE", None)
    res.sort()
    fl = 2.0 * M * N * K
    print(f"{name}: " + "  ".join(f"tile {tl} x{sp}: {us:6.1f}us" for us, tl, sp in res[:6]) + f"   | best {fl / res[0][0] / 1e6:.0f} TF/s", flush=True)
