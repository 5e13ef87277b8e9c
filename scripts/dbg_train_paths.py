import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from layoutdit_amd import config as cfgs, synth, training
from layoutdit_amd.modeling import DiTEncoder
from tests.golden.make_golden_grad import upstream
DEV = "cuda:0"
cfg = cfgs.vit_micro(); cfg.drop_path_rate = 0.0
w = synth.synth_weights(cfg, 3)
x = torch.from_numpy(synth.synth_images(4, 64, 64, seed=5, kind="uniform")).to(DEV)
dt = [torch.from_numpy(d).to(DEV) for d in upstream(cfg, 4, cfg.tokens(64, 64), 9)]
a = DiTEncoder(cfg, compute_dtype="bf16").load_numpy(w).to(DEV).train()
opt = torch.optim.AdamW([p for p in a.parameters()], lr=1e-3, weight_decay=0.0)
b = DiTEncoder(cfg, compute_dtype="bf16").load_numpy(w).to(DEV).train()
fused = training.TrainStep(b, lr=1e-3, weight_decay=0.0, dtaps=dt, drop_path_rate=0.0, img_size=(64, 64))
for it in range(3):
    opt.zero_grad()
    out = a(x)
    sum((out.hidden_states[t] * d).sum() for t, d in zip(cfg.taps, dt)).backward()
    ga = a._flat_state.grads.clone()
    va = a._flat_state.version()[:3]
    opt.step()
    fused.step(x)
    gb = fused.state.grads
    print(it, "grad equal:", torch.equal(ga, gb), float((ga - gb).abs().max()), "versions", va, a._flat_state.version()[:3],
          "param diff", float((a._flat_state.params - fused.state.params).abs().max()))
sa, sb = a.state_dict(), b.state_dict()
for k in list(sa)[:8]:
    print(k, float((sa[k] - sb[k]).abs().max()), float(sa[k].abs().max()))
