"""Whole-path parity on the GPU: ``DiTEncoder`` / ``DiTBackbone`` (-> ctypes -> ldit_vit_forward) against the CPU
oracle on the same seeded inputs, against the committed golden vectors generated from transformers.BeitModel, and -
at BASELINE.json's full size (ViT-B/16, bs=64) - through size-independent properties.

fp32 tolerance: north-star 1e-3 relative; gates here: relative-L2 <= 2e-5 and |err| <= 1e-4 * max(|ref|, 1) per tap."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from layoutdit_amd import config as cfgs, synth          # noqa: E402
from layoutdit_amd.modeling import DiTBackbone, DiTEncoder  # noqa: E402
from oracle import oracle                                # noqa: E402
from tests.util import max_rel, rel_l2, resample_pos     # noqa: E402

DEV = "cuda:0"


def _model(cfg, wseed):
    w = synth.synth_weights(cfg, wseed)
    m = DiTEncoder(cfg).load_numpy(w).to(DEV).eval()
    return m, w


def _run(m, x, taps=None):
    with torch.no_grad():
        out = m(torch.from_numpy(x).to(DEV), taps=taps)
    torch.cuda.synchronize()
    return out


def test_micro_every_hidden_state_vs_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g0_micro.npz"))
    cfg = cfgs.vit_micro()
    m, _ = _model(cfg, int(g["seeds"][0]))
    out = _run(m, g["x"], taps=[0, 1, 2, 3])
    for l in range(4):
        h = out.hidden_states[l].cpu().numpy()
        assert rel_l2(h, g["hidden"][l]) < 5e-6, l
        assert max_rel(h, g["hidden"][l]) < 2e-5, l
    assert out.last_hidden_state is out.hidden_states[3]


@pytest.mark.parametrize("name,geom", [("g1_tiny.npz", "tiny"), ("g2_base.npz", "base")])
def test_taps_vs_golden_and_oracle(golden_dir, name, geom):
    g = np.load(os.path.join(golden_dir, name))
    cfg = cfgs.GEOMETRIES[geom]()
    B, size = int(g["geometry"][6]), int(g["geometry"][7])
    m, w = _model(cfg, int(g["seeds"][0]))
    x = synth.synth_images(B, size, size, seed=int(g["seeds"][1]))
    out = _run(m, x)
    ref_taps, _ = oracle.vit_forward(cfg, w, x)
    stride = int(g["stride"][0])
    for t, ref in zip(cfg.taps, ref_taps):
        h = out.hidden_states[t].cpu().numpy()
        assert rel_l2(h, ref) < 2e-5, t                       # vs the oracle, every element
        assert max_rel(h, ref) < 1e-4, t
        assert rel_l2(h.reshape(-1)[::stride], g[f"tap{t}_sample"]) < 2e-5, t   # vs HF BeitModel
        assert max_rel(h[0, :8, :8], g[f"tap{t}_head"]) < 1e-4, t
    untapped = [i for i in range(cfg.num_hidden_layers + 1) if i not in cfg.taps]
    assert all(out.hidden_states[i] is None for i in untapped)


def test_large_512_long_sequence_vs_golden(golden_dir):
    """ViT-L/16 at 512x512, N = 1025: chunked K/V attention + bicubically resampled position table."""
    g = np.load(os.path.join(golden_dir, "g3_large512.npz"))
    cfg = cfgs.vit_large()
    m, w = _model(cfg, int(g["seeds"][0]))
    x = synth.synth_images(1, 512, 512, seed=int(g["seeds"][1]))
    out = _run(m, x)
    stride = int(g["stride"][0])
    for t in cfg.taps:
        h = out.hidden_states[t].cpu().numpy()
        assert h.shape == (1, 1025, 1024)
        assert rel_l2(h.reshape(-1)[::stride], g[f"tap{t}_sample"]) < 3e-5, t
        assert max_rel(h[0, :8, :8], g[f"tap{t}_head"]) < 1e-4, t
    pos = m._position_table(32, 32).cpu().numpy()
    assert rel_l2(pos.reshape(-1)[::stride], g["pos_resampled_sample"]) < 1e-6


def test_base_bs64_properties():
    """BASELINE config 2 at full size.  The oracle would need minutes here, so use what the arithmetic guarantees:
    rows of different images never mix and the k-order of every dot product is fixed, hence
    (1) image i of a 64-batch gives BIT-IDENTICAL taps to the same image run in a batch of 2, 1 or 5 (the batch of 2 is
        what the previous test pins to the oracle and to HF; 1 and 2 take the serving-size GEMM tiling), (2) a permutation of the batch permutes the outputs bit-exactly,
    (3) repeated runs are bit-identical (no atomics, no race)."""
    cfg = cfgs.vit_base()
    m, _ = _model(cfg, 0)
    x = synth.synth_images(64, 224, 224, seed=1234)
    big = [h.cpu().numpy() for h in _run(m, x).hidden_states if h is not None]
    small = [h.cpu().numpy() for h in _run(m, x[:2]).hidden_states if h is not None]
    for a, b in zip(big, small):
        np.testing.assert_array_equal(a[:2], b)
    pair = [h.cpu().numpy() for h in _run(m, x[[37, 5]]).hidden_states if h is not None]
    for a, b in zip(big, pair):
        np.testing.assert_array_equal(a[[37, 5]], b)
    # serving sizes run the 32x32 / 16x16x4 GEMM tiling (gemm_thin_f32.hip: all four GEMMs at one image, o_proj and fc2 up to
    # four) and the query-split attention launch (up to ten images): one image alone and five still reproduce their rows of
    # the 64-batch
    for sel in ([11], [3, 60, 17, 41, 8], list(range(20, 36))):       # 16 images: the 80-row panels on o_proj / fc2
        few = [h.cpu().numpy() for h in _run(m, x[sel]).hidden_states if h is not None]
        for a, b in zip(big, few):
            np.testing.assert_array_equal(a[sel], b)
    again = [h.cpu().numpy() for h in _run(m, x).hidden_states if h is not None]
    for a, b in zip(big, again):
        np.testing.assert_array_equal(a, b)
    assert all(np.isfinite(a).all() for a in big)


def test_ragged_batches_and_rectangular_input():
    """Batch sizes around the 320-row GEMM tile (B*197 rows) and a non-square grid."""
    cfg = cfgs.vit_tiny()
    m, w = _model(cfg, 1)
    for B in (1, 3):
        x = synth.synth_images(B, 224, 224, seed=77)
        out = _run(m, x)
        ref, _ = oracle.vit_forward(cfg, w, x)
        for t, r in zip(cfg.taps, ref):
            assert rel_l2(out.hidden_states[t].cpu().numpy(), r) < 2e-5
    x = synth.synth_images(2, 160, 96, seed=78)                 # 10 x 6 grid -> resampled positions
    out = _run(m, x)
    pos = resample_pos(w["embeddings.position_embeddings"], 14, 10, 6)
    ref, _ = oracle.vit_forward(cfg, w, x, pos=pos)
    for t, r in zip(cfg.taps, ref):
        assert rel_l2(out.hidden_states[t].cpu().numpy(), r) < 2e-5


def test_weight_update_repacks():
    cfg = cfgs.vit_micro()
    m, w = _model(cfg, 7)
    x = synth.synth_images(2, 64, 64, seed=3, kind="uniform")
    a = _run(m, x).hidden_states[3].cpu().numpy()
    with torch.no_grad():
        m.encoder.layer[0].lambda_1.mul_(2.0)
    b = _run(m, x).hidden_states[3].cpu().numpy()
    w2 = dict(w)
    w2["encoder.layer.0.lambda_1"] = w["encoder.layer.0.lambda_1"] * 2.0
    ref, _ = oracle.vit_forward(cfg, w2, x)
    assert rel_l2(b, ref[-1]) < 5e-6
    assert rel_l2(a, b) > 1e-4


def test_error_behaviour():
    cfg = cfgs.vit_micro()
    m, _ = _model(cfg, 7)
    with pytest.raises(ValueError, match="channel dimension"):
        m(torch.zeros(1, 4, 64, 64, device=DEV))
    with pytest.raises(RuntimeError, match="no CPU"):
        m(torch.zeros(1, 3, 64, 64))
    m.train()
    out = m(torch.zeros(1, 3, 64, 64, device=DEV))             # round 3: the f32 build trains (bf16 operands, fp32 master weights)
    assert out.hidden_states[3].requires_grad
    out = m(torch.zeros(1, 3, 96, 96, device=DEV))             # ... also at a grid that resamples the position table (round 3)
    assert out.hidden_states[3].requires_grad and tuple(out.hidden_states[3].shape) == (1, 37, cfg.hidden_size)
    for p in m.parameters():
        p.requires_grad_(False)
    assert m(torch.zeros(1, 3, 64, 64, device=DEV)).hidden_states[3] is not None   # frozen backbone in train mode is fine


def test_half_precision_inputs_are_widened_and_outputs_follow_the_input_dtype():
    """The reference's trainer feeds ``.half()`` images (ref src/layoutdit/training/trainer.py:155): the module widens
    them to fp32 with the library's own kernel (ldit_cast_f16_f32) and hands the hidden states back as fp16."""
    cfg = cfgs.vit_micro()
    m, _ = _model(cfg, 7)
    x = torch.from_numpy(synth.synth_images(2, 64, 64, seed=3)).to(DEV)
    with torch.no_grad():
        ref = m(x.half().float()).hidden_states[3]
        out = m(x.half()).hidden_states[3]
        assert out.dtype == torch.float16
        assert torch.equal(out, ref.half())
        with pytest.raises(ValueError, match="float32 or float16"):       # the two dtypes the reference feeds (fp32 eval, fp16 train)
            m(x.to(torch.bfloat16))


def test_backbone_feature_maps_vs_golden(golden_dir):
    """DiTBackbone.forward -> {p2..p5} (ref src/layoutdit/modeling/dit_backbone.py:38-62)."""
    g0 = np.load(os.path.join(golden_dir, "g0_micro.npz"))
    g5 = np.load(os.path.join(golden_dir, "g5_maps.npz"))
    bb = DiTBackbone(config=cfgs.vit_micro())
    bb.dit.load_numpy(synth.synth_weights(cfgs.vit_micro(), int(g0["seeds"][0])))
    bb = bb.to(DEV).eval()
    with torch.no_grad():
        feats = bb(torch.from_numpy(g0["x"]).to(DEV))
    assert list(feats) == ["p2", "p3", "p4", "p5"]
    for k in feats:
        assert tuple(feats[k].shape) == g5[f"micro_{k}"].shape
        assert max_rel(feats[k].cpu().numpy(), g5[f"micro_{k}"]) < 2e-5, k
    g1 = np.load(os.path.join(golden_dir, "g1_tiny.npz"))
    bb = DiTBackbone(config=cfgs.vit_tiny())
    bb.dit.load_numpy(synth.synth_weights(cfgs.vit_tiny(), int(g1["seeds"][0])))
    bb = bb.to(DEV).eval()
    x = synth.synth_images(2, 224, 224, seed=int(g1["seeds"][1]))
    with torch.no_grad():
        feats = bb(torch.from_numpy(x).to(DEV))
    for k in feats:
        a = feats[k].contiguous().cpu().numpy()
        assert list(a.shape) == list(g5[f"tiny_{k}_shape"])
        assert rel_l2(a.reshape(-1)[::13], g5[f"tiny_{k}_sample"]) < 2e-5, k


def test_forward_is_hipgraph_capturable_and_replays_bit_exact():
    """include/ldit.h promises enqueue-only entry points (no allocation, no sync): capture one forward into a HIP graph on
    a side stream, replay it on new inputs written into the captured input buffer, and compare with eager runs."""
    cfg = cfgs.vit_tiny()
    m, _ = _model(cfg, 1)
    xs = [torch.from_numpy(synth.synth_images(2, 224, 224, seed=900 + i)).to(DEV) for i in range(3)]
    with torch.no_grad():
        eager = [[h.clone() for h in m(x).hidden_states if h is not None] for x in xs]   # also warms attributes / packing
        static_x = xs[0].clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m(static_x)                                                                 # warm-up on the capture stream
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = [h for h in m(static_x).hidden_states if h is not None]
        for x, ref in zip(xs, eager):
            static_x.copy_(x)
            graph.replay()
            torch.cuda.synchronize()
            for a, b in zip(static_out, ref):
                assert torch.equal(a, b)


def test_concurrent_streams_do_not_interfere():
    """Two modules on two streams at once: the library keeps no global mutable state (one workspace per module)."""
    cfg = cfgs.vit_tiny()
    m1, _ = _model(cfg, 1)
    m2, _ = _model(cfg, 2)
    x1 = torch.from_numpy(synth.synth_images(2, 224, 224, seed=11)).to(DEV)
    x2 = torch.from_numpy(synth.synth_images(3, 224, 224, seed=12)).to(DEV)
    with torch.no_grad():
        r1 = m1(x1).hidden_states[12].clone()
        r2 = m2(x2).hidden_states[12].clone()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        outs = []
        for _ in range(5):
            with torch.cuda.stream(s1):
                a = m1(x1).hidden_states[12]
            with torch.cuda.stream(s2):
                b = m2(x2).hidden_states[12]
            outs.append((a, b))
        torch.cuda.synchronize()
    for a, b in outs:
        assert torch.equal(a, r1) and torch.equal(b, r2)
