#!/usr/bin/env python3
"""fp32 patch embedding (ldit_embed_f32: NCHW gather on the LDS-DMA source) against the bf16 one (ldit_embed_bf16: bf16 im2col
pass + bf16 MFMA GEMM) on the bench geometries, interleaved in one process."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import ops  # noqa: E402

def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for B, S, C in ((64, 224, 768), (32, 224, 768), (16, 512, 1024)):
    x = torch.rand(B, 3, S, S, device="cuda") * 2 - 1
    pw = torch.randn(C, 3, 16, 16, device="cuda") * 0.02
    pb, cls = torch.randn(C, device="cuda") * 0.02, torch.randn(C, device="cuda") * 0.02
    T = (S // 16) ** 2 + 1
    pos = torch.randn(T, C, device="cuda") * 0.02
    pw16 = pw.to(torch.bfloat16).reshape(C, -1).contiguous()
    a, b = [], []
    for r in range(5):
        a.append(t(lambda: ops.embed(x, pw, pb, cls, pos, 16)))
        b.append(t(lambda: ops.embed_bf16(x, pw16, pb, cls, pos, 16)))
    print(f"B={B} {S}x{S} C={C}: fp32 {statistics.median(a):7.1f} us   bf16 {statistics.median(b):7.1f} us (incl. the output / scratch allocation of the wrapper)", flush=True)

# the detector's input transform fused into the im2col pass (ldit_embed_bf16_images) against the two-step path
# (ldit_preprocess_f32 -> fp32 batch -> ldit_embed_bf16), ragged page-like images, interleaved
print("input transform + bf16 embedding: two launches through an fp32 batch vs fused into the im2col pass")
for B, S, C in ((64, 224, 768), (32, 224, 768), (16, 512, 1024)):
    g = torch.Generator(device="cuda").manual_seed(1)
    imgs = [torch.rand(3, 600 + 37 * (i % 7), 450 + 29 * (i % 5), device="cuda", generator=g) for i in range(B)]
    pw16 = (torch.randn(C, 768, device="cuda") * 0.02).to(torch.bfloat16)
    pb, cls = torch.randn(C, device="cuda") * 0.02, torch.randn(C, device="cuda") * 0.02
    pos = torch.randn((S // 16) ** 2 + 1, C, device="cuda") * 0.02
    assert torch.equal(ops.embed_bf16(ops.preprocess(imgs, size=S), pw16, pb, cls, pos, 16), ops.embed_bf16_images(imgs, pw16, pb, cls, pos, 16, size=S))
    a, b, c = [], [], []
    for r in range(5):
        a.append(t(lambda: ops.embed_bf16(ops.preprocess(imgs, size=S), pw16, pb, cls, pos, 16)))
        b.append(t(lambda: ops.embed_bf16_images(imgs, pw16, pb, cls, pos, 16, size=S)))
        c.append(t(lambda: ops.preprocess(imgs, size=S)))
    print(f"B={B} -> {S}x{S} C={C}: two-step {statistics.median(a):7.1f} us (of which the transform alone {statistics.median(c):6.1f})   "
          f"fused {statistics.median(b):7.1f} us   (bit-equal outputs)", flush=True)
