"""The gradient oracle (oracle/vit_oracle_torch.py: train_reference, float64 autograd of the restated encoder) against
the committed gradient goldens generated from HF ``BeitModel`` (tests/golden/make_golden_grad.py): eval-mode arithmetic
and train mode with the stochastic-depth factors HF actually drew.  CPU only."""
import os

import numpy as np
import pytest

from layoutdit_amd import config as cfgs, synth
from oracle.vit_oracle_torch import drop_path_rates, train_reference
from tests.golden.make_golden_grad import sample_stride, upstream
from tests.util import rel_l2


@pytest.mark.parametrize("name,geom", [("g6_grad_micro.npz", "micro"), ("g7_grad_tiny.npz", "tiny"), ("g8_grad_base.npz", "base")])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_gradient_oracle_vs_hf_goldens(golden_dir, name, geom, mode):
    g = np.load(os.path.join(golden_dir, name))
    cfg = cfgs.GEOMETRIES[geom]()
    B, size = int(g["geometry"][6]), int(g["geometry"][7])
    wseed, xseed, gseed = (int(v) for v in g["seeds"])
    w = synth.synth_weights(cfg, wseed)
    x = synth.synth_images(B, size, size, seed=xseed, kind="uniform" if size < 224 else "doc")
    dtaps = upstream(cfg, B, cfg.tokens(size, size), gseed)
    scales = g[f"{mode}_drop_scales"]
    if mode == "eval":
        assert (scales == 1.0).all()
    else:
        rates = drop_path_rates(cfg, float(g["drop_path_rate"][0]))
        for l, r in enumerate(rates):                       # captured factors are exactly 0 or 1 / keep_prob of the layer
            ok = np.isclose(scales[l], 0.0) | np.isclose(scales[l], 1.0 / (1.0 - r), rtol=1e-5)
            assert ok.all(), l
        assert (scales == 0.0).any()
    taps, grads = train_reference(cfg, w, x, dtaps, drop_scales=None if mode == "eval" else scales)
    stride = int(g["stride"][0])
    for t, a in zip(cfg.taps, taps):
        assert rel_l2(a.reshape(-1)[::max(stride, 7)], g[f"{mode}_tap{t}_sample"]) < 2e-6, t
    checked = 0
    for k, gr in grads.items():
        ref = g[f"{mode}_grad/{k}"]
        norm = float(g[f"{mode}_gstat/{k}"][0])
        if k.endswith("attention.attention.key.bias"):
            continue
        # fp32 HF autograd vs float64 oracle: gate relative to the gradient's own norm
        err = np.linalg.norm(gr.reshape(-1)[::sample_stride(g, gr.size)].astype(np.float64) - ref) / max(np.linalg.norm(ref), 1e-30)
        assert err < 2e-4, (k, err)
        assert abs(np.linalg.norm(gr.astype(np.float64)) - norm) <= 2e-4 * norm + 1e-12, k
        checked += 1
    assert checked == len([k for k in synth.param_shapes(cfg) if "mask_token" not in k and not k.startswith("pooler.")])
