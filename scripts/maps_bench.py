#!/usr/bin/env python3
"""Throughput of DiTBackbone's tap post-processing (ldit_tap_to_map_f32) at ViT-B bs=64 (GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import ops  # noqa: E402
tap = torch.randn(64, 197, 768, device="cuda")
for scale in (4.0, 2.0, 0.5):
    for _ in range(3):
        out = ops.tap_to_map(tap, 14, 14, scale)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = ops.tap_to_map(tap, 14, 14, scale)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    by = out.numel() * 4 + tap.numel() * 4
    print(f"scale {scale}: out {tuple(out.shape)} {ms * 1e3:.1f} us  {by / ms / 1e6:.0f} GB/s (read tap once + write map)")
