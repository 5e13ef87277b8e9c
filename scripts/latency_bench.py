#!/usr/bin/env python3
"""Small-batch latency of the forward, eager launches vs one captured HIP graph (GPU box).  DTYPE=f32 (default) | f32x3 | f32x6 | bf16."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import config as cfgs, synth  # noqa: E402
from layoutdit_amd.modeling import DiTEncoder     # noqa: E402
cfg = cfgs.vit_base()
DT = os.environ.get("DTYPE", "f32")
m = DiTEncoder(cfg, compute_dtype=DT).load_numpy(synth.synth_weights(cfg, 0)).to("cuda").eval()
for B in (1, 2, 4, 8, 16):
    x = torch.from_numpy(synth.synth_images(B, 224, 224)).to("cuda")
    with torch.no_grad():
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            m(x)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 50
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m(x)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m(x)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 50
    print(f"{DT} bs={B:3d}: eager {eager * 1e3:7.3f} ms ({B / eager:7.0f} img/s)   graph {graph * 1e3:7.3f} ms ({B / graph:7.0f} img/s)")
