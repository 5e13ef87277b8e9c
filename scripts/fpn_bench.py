#!/usr/bin/env python3
"""Time of the callers either side of the encoder at ViT-B/16 224x224 bs=64 fp32 (GPU box): DiTBackbone (encoder + the four
feature maps) and DiTWithFPN (+ laterals on the tokens, top-down merges, four 3x3 convolutions, pool), against the bare
encoder; and the detector input transform on a ragged list of 64 images."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import config as cfgs, synth  # noqa: E402
from layoutdit_amd.modeling import DiTBackbone, DiTEncoder, DiTWithFPN, DetectorInputTransform  # noqa: E402

dev = "cuda:0"
cfg = cfgs.vit_base()
w = synth.synth_weights(cfg, 0)
x = torch.from_numpy(synth.synth_images(64, 224, 224)).to(dev)


def timed(fn, n=10):
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


enc = DiTEncoder(cfg).load_numpy(w).to(dev).eval()
bb = DiTBackbone(config=cfg)
bb.dit.load_numpy(w)
bb = bb.to(dev).eval()
fpn = DiTWithFPN(config=cfg)
fpn.backbone.dit.load_numpy(w)
fpn = fpn.to(dev).eval()
t_enc, t_bb, t_fpn = timed(lambda: enc(x)), timed(lambda: bb(x)), timed(lambda: fpn(x))
conv_flops = 2 * 9 * 256 * 256 * (56 * 56 + 28 * 28 + 14 * 14 + 7 * 7) * 64
lat_flops = 2 * 768 * 256 * 4 * 197 * 64
print(f"encoder {t_enc:.2f} ms | DiTBackbone {t_bb:.2f} ms (+{t_bb - t_enc:.2f} for p2..p5) | DiTWithFPN {t_fpn:.2f} ms "
      f"(+{t_fpn - t_enc:.2f}: {lat_flops / 1e9:.0f} GFLOP laterals + {conv_flops / 1e9:.0f} GFLOP 3x3 = "
      f"{(lat_flops + conv_flops) / (t_fpn - t_enc) / 1e9:.0f} TFLOP/s incl. the merges)")
imgs = [torch.rand(3, 180 + 7 * (i % 11), 200 + 5 * (i % 13), device=dev) for i in range(64)]
tr = DetectorInputTransform(fixed_size=(224, 224)).to(dev)
t_tr = timed(lambda: tr(imgs))
print(f"DetectorInputTransform on 64 ragged images -> [64, 3, 224, 224]: {t_tr * 1e3:.0f} us")
