#!/usr/bin/env python3
"""In-kernel phase timing of the bf16 attention (diagnostic build csrc/build/libldit_hip_dbg.so, `make dbg`; GPU box only).
Per workgroup, wave 0: cycles spent in (DMA wait + barrier, DMA issue, S = K Q^T issue, softmax incl. the wait for S,
P V incl. its LDS waits) summed over the chunks, and the kernel-lifetime total."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "layoutdit_amd", "csrc", "build", "libldit_hip_dbg.so"))
vp, i64 = C.c_void_p, C.c_int64
lib.ldit_attention_bf16.argtypes = [vp] * 4 + [i64] * 8 + [C.c_float, vp]
lib.ldit_dbg_set_attn_stamps.argtypes = [vp]
PRE = os.environ.get("PRE", "0") == "1"      # PRE=1: q carries scale * log2(e), the kernel is called with scale = 0
for B, N, H in ((16, 1025, 16), (64, 197, 12)):
    Cc = 64 * H
    qkv = (0.35 * torch.randn(B, N, 3 * Cc, device="cuda"))
    if PRE:
        qkv[..., :Cc] *= 0.125 * 1.4426950408889634
    qkv = qkv.to(torch.bfloat16)
    o = torch.empty(B, N, Cc, device="cuda", dtype=torch.bfloat16)
    nwg = B * H * ((((N + 31) // 32) + 3) // 4)
    st = torch.zeros(nwg * 6, dtype=torch.int64, device="cuda")
    assert lib.ldit_dbg_set_attn_stamps(st.data_ptr()) == 0
    for _ in range(20):
        rc = lib.ldit_attention_bf16(qkv.data_ptr(), qkv.data_ptr() + 2 * Cc, qkv.data_ptr() + 4 * Cc, o.data_ptr(), B, N, H, 64,
                                     3 * Cc, 3 * Cc, 3 * Cc, Cc, 0.0 if PRE else 0.125, None)
        assert rc == 0
    torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(nwg, 6)
    nch = (N + 63) // 64
    print(f"B={B} N={N} H={H}: {nwg} workgroups, {nch} chunks of 64 keys; median cycles per workgroup (wave 0) and per chunk:")
    for i, lab in enumerate(("wait+barrier", "DMA issue", "S issue", "softmax(+S wait)", "PV", "kernel total")):
        v = s[:, i]
        print(f"   {lab:18s} {np.median(v):10.0f}   per chunk {np.median(v) / nch:8.0f}   p10 {np.percentile(v, 10):9.0f} p90 {np.percentile(v, 90):9.0f}")
