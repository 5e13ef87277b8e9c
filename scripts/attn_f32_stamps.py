#!/usr/bin/env python3
"""In-kernel phase timing of the fp32 attention (diagnostic build csrc/build/libldit_hip_dbg.so, `make dbg`; GPU box only).
Per workgroup, wave 0: cycles in (Q load + K/V staging, S = K Q^T, softmax, P V) and the kernel-lifetime total."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "layoutdit_amd", "csrc", "build", "libldit_hip_dbg.so"))
vp, i64 = C.c_void_p, C.c_int64
lib.ldit_attention_f32.argtypes = [vp] * 4 + [i64] * 8 + [C.c_float, vp]
lib.ldit_dbg_set_attn32_stamps.argtypes = [vp]
for B, N, H in ((1, 197, 12), (4, 197, 12), (64, 197, 12)):
    Cc = 64 * H
    qkv = torch.randn(B, N, 3 * Cc, device="cuda")
    o = torch.empty(B, N, Cc, device="cuda")
    nwg = B * H * 2
    st = torch.zeros(nwg * 6, dtype=torch.int64, device="cuda")
    assert lib.ldit_dbg_set_attn32_stamps(st.data_ptr()) == 0
    for _ in range(5):
        rc = lib.ldit_attention_f32(qkv.data_ptr(), qkv.data_ptr() + 4 * Cc, qkv.data_ptr() + 8 * Cc, o.data_ptr(), B, N, H, 64,
                                    3 * Cc, 3 * Cc, 3 * Cc, Cc, 0.125, None)
        assert rc == 0
    torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(nwg, 6)
    s = s[s[:, 5] > 0]
    print(f"B={B} N={N} H={H}: {len(s)} workgroups stamped; median cycles per workgroup (wave 0):")
    for i, lab in ((0, "Q load + staging"), (1, "S = K Q^T"), (2, "softmax"), (3, "P V"), (5, "kernel total")):
        v = s[:, i]
        print(f"   {lab:18s} {np.median(v):10.0f}   p10 {np.percentile(v, 10):9.0f} p90 {np.percentile(v, 90):9.0f}")
