#!/usr/bin/env python3
"""Fixed cost and per-k cost of the backward's dgrad GEMM (gemm_bf16_tr, K-contiguous A) beside the forward GEMM of the same output
shape: time = a + b K fitted over K = 768 ... 6144 at M = 12 608 (ViT-B bs=64), per tile height (LDIT_GEMM_BF16_TR_TILE / LDIT_GEMM_BF16_TILE
3 = 256 rows, 5 = 320 rows).  Isolated back-to-back launches (they throttle harder than inside a model: read the RATIOS)."""
import os, sys, statistics
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import _lib  # noqa: E402

lib = _lib.load()
BF = torch.bfloat16
stream = torch.cuda.current_stream().cuda_stream

def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

M = 12608
zeros = torch.zeros(64, device="cuda")
for N, epi_tr, epi_fw, label in ((3072, _lib.EPI_GELU_BWD, _lib.EPI_BIAS_GELU, "N=3072 (fc2 dgrad: x gelu', bf16 out | fc1 forward: bias + GELU, two bf16 outs)"),
                                 (768, _lib.EPI_F32, _lib.EPI_BIAS, "N=768 (fc1 / qkv dgrad: fp32 out | forward with a bf16 out)")):
    print(label)
    for tile in (3, 5):
        rows = []
        for K in (768, 1536, 3072, 6144):
            dy = (torch.rand(M, K, device="cuda") * 2 - 1).to(BF)
            w_tr = (torch.rand(K, N, device="cuda") * 0.1).to(BF)            # [reduction, output]: nn.Linear's layout for the dgrad
            w_fw = (torch.rand(N, K, device="cuda") * 0.1).to(BF)            # [output, reduction]: the forward's
            bias = torch.zeros(N, device="cuda")
            aux = torch.rand(M, N, device="cuda").to(BF)
            out_tr = torch.empty((M, N), dtype=torch.float32 if epi_tr == _lib.EPI_F32 else BF, device="cuda")
            out_fw = torch.empty((M, N), dtype=BF, device="cuda")
            pre = torch.empty((M, N), dtype=BF, device="cuda")
            _lib.set_switch("LDIT_GEMM_BF16_TR_TILE", tile); _lib.set_switch("LDIT_GEMM_BF16_TILE", tile)
            f_tr = lambda: _lib.check(lib.ldit_linear_bf16_tr(dy.data_ptr(), K, 0, w_tr.data_ptr(), N, out_tr.data_ptr(), N, M, N, K, epi_tr,
                                                            aux.data_ptr() if epi_tr == _lib.EPI_GELU_BWD else None, 1, zeros.data_ptr(), stream))
            f_fw = lambda: _lib.check(lib.ldit_linear_bf16_ex(dy.data_ptr(), K, w_fw.data_ptr(), bias.data_ptr(), out_fw.data_ptr(), N, M, N, K, epi_fw,
                                                            None, None, None, pre.data_ptr() if epi_fw == _lib.EPI_BIAS_GELU else None, None, None, 0, 1, stream))
            a, b = [], []
            for r in range(3):
                a.append(t(f_tr)); b.append(t(f_fw))
            rows.append((K, statistics.median(a), statistics.median(b)))
        _lib.set_switch("LDIT_GEMM_BF16_TR_TILE", None); _lib.set_switch("LDIT_GEMM_BF16_TILE", None)
        ks = np.array([r[0] for r in rows], float)
        for name, col in (("dgrad  ", 1), ("forward", 2)):
            ys = np.array([r[col] for r in rows])
            b_, a_ = np.polyfit(ks, ys, 1)
            print(f"  tile {256 if tile == 3 else 320} rows  {name}: " + "  ".join(f"K={int(k)}: {y:6.1f} us" for k, y in zip(ks, ys)) + f"   fit {a_:5.1f} + {b_ * 1e3:5.2f}e-3 K")
