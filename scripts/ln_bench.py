#!/usr/bin/env python3
"""LayerNorm kernel (fp32 in; fp32 / bf16 out through the whole-path entry is not exposed: fp32 out here) on the bench shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import ops  # noqa: E402
for rows, C in ((16400, 1024), (12608, 768), (6304, 768), (197, 768)):
    x = torch.randn(rows, C, device="cuda"); g = torch.rand(C, device="cuda"); b = torch.randn(C, device="cuda")
    for _ in range(5): ops.layernorm(x, g, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): ops.layernorm(x, g, b)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    print(f"rows={rows:6d} C={C:5d}: {us:7.1f} us  {rows * C * 8 / us / 1e6:6.2f} TB/s (fp32 in + fp32 out)")
