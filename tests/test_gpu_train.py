"""Train step on the GPU (BASELINE.json configs[2]), through the C ABI.

Per kernel: attention forward-with-LSE / backward, LayerNorm backward, the training epilogues of the bf16 GEMM (fp32
output, split-K slabs, GELU derivative, pre-activation copy, per-row stochastic-depth factor), fused AdamW - each against a
float64 torch reference on the SAME bf16-rounded inputs.
Whole path: every parameter gradient of ``hs = self.dit(x).hidden_states`` for upstream gradients at the taps, against the
float64 autograd oracle (oracle/vit_oracle_torch.py: train_reference, itself pinned to HF BeitModel's gradients in
tests/test_oracle_grad_golden.py) and directly against the committed HF gradient goldens g6 / g7 - eval-mode arithmetic and
train mode with the stochastic-depth factors HF drew.

Tolerance: the backward runs on bf16 operands with fp32 accumulation, like the forward (bf16 gate 2e-2 on
activations).  Gradients are gated per tensor at relative-L2 <= 3e-2 of the tensor's own norm (measured ~3e-3 .. 1e-2)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import ctypes as C                                        # noqa: E402

from layoutdit_amd import _lib, config as cfgs, synth, training   # noqa: E402
from layoutdit_amd.modeling import DiTEncoder              # noqa: E402
from oracle import oracle                                  # noqa: E402
from oracle.vit_oracle_torch import train_reference        # noqa: E402
from tests.golden.make_golden_grad import sample_stride, upstream         # noqa: E402
from tests.util import rel_l2                              # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16
GRAD_TOL = 3e-2


def _rand(seed, *shape, scale=1.0):
    n = int(np.prod(shape))
    return (scale * synth.normal(seed, 9, n)).astype(np.float32).reshape(shape)


def _bf(a):
    return torch.from_numpy(a).to(BF)


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ---- attention ----------------------------------------------------------------------------------------------------------
def _attention_ref(q, k, v, do, H):
    """float64 reference on bf16-rounded inputs: o, log2-domain lse, dq, dk, dv."""
    q, k, v, do = (t.to(torch.float64).requires_grad_(t is not do) for t in (q, k, v, do))
    B, N, HD = q.shape
    D = HD // H
    qh, kh, vh = (t.view(B, N, H, D).transpose(1, 2) for t in (q, k, v))
    s = (qh @ kh.transpose(-1, -2)) * D ** -0.5
    lse2 = torch.logsumexp(s, dim=-1) / np.log(2.0)
    o = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, N, HD)
    (o * do).sum().backward()
    return o.detach(), lse2.detach(), q.grad, k.grad, v.grad


@pytest.mark.parametrize("B,N,H", [(2, 197, 3), (1, 256, 2), (3, 17, 2), (2, 33, 1), (1, 257, 2), (2, 300, 1), (1, 1025, 2), (1, 512, 1)])
def test_attention_forward_lse_and_backward(B, N, H):
    lib = _lib.load()
    D, HD = 64, H * 64
    qkv = _bf(_rand(7, B, N, 3 * HD)).to(DEV)
    do = _bf(_rand(8, B, N, HD, scale=0.5)).to(DEV)
    q, k, v = qkv[..., :HD], qkv[..., HD:2 * HD], qkv[..., 2 * HD:]
    o = torch.empty((B, N, HD), dtype=BF, device=DEV)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=DEV)
    _lib.check(lib.ldit_attention_fwd_lse_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, N, H, D,
                                               3 * HD, 3 * HD, 3 * HD, HD, D ** -0.5, _stream()))
    ro, rl, rdq, rdk, rdv = _attention_ref(q.cpu().float(), k.cpu().float(), v.cpu().float(), do.cpu().float(), H)
    assert rel_l2(o.float().cpu().numpy(), ro.numpy()) < 6e-3
    np.testing.assert_allclose(lse.cpu().numpy(), rl.numpy(), atol=2e-3)
    dqkv = torch.full((B, N, 3 * HD), float("nan"), dtype=BF, device=DEV)
    _lib.check(lib.ldit_attention_bwd_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
                                           dqkv.data_ptr(), dqkv[..., HD:].data_ptr(), dqkv[..., 2 * HD:].data_ptr(), B, N, H, D,
                                           3 * HD, HD, HD, 3 * HD, D ** -0.5, _stream()))
    torch.cuda.synchronize()
    got = dqkv.float().cpu().numpy()
    assert np.isfinite(got).all()                       # every row of dq | dk | dv was written
    for name, a, r in (("dq", got[..., :HD], rdq), ("dk", got[..., HD:2 * HD], rdk), ("dv", got[..., 2 * HD:], rdv)):
        assert rel_l2(a, r.numpy()) < 1.5e-2, name


def test_attention_backward_long_sequences_are_reproducible():
    """N > 256 takes the blocked form (a workgroup per 256-row block and role; every sum inside one wave): two runs agree bit
    for bit, and every row of dq | dk | dv is written."""
    lib = _lib.load()
    B, N, H, D = 2, 600, 2, 64
    HD = H * D
    qkv = _bf(_rand(7, B, N, 3 * HD)).to(DEV)
    do = _bf(_rand(8, B, N, HD, scale=0.5)).to(DEV)
    q, k, v = qkv[..., :HD], qkv[..., HD:2 * HD], qkv[..., 2 * HD:]
    o = torch.empty((B, N, HD), dtype=BF, device=DEV)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=DEV)
    _lib.check(lib.ldit_attention_fwd_lse_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, N, H, D,
                                               3 * HD, 3 * HD, 3 * HD, HD, D ** -0.5, _stream()))
    outs = []
    for _ in range(2):
        dqkv = torch.full((B, N, 3 * HD), float("nan"), dtype=BF, device=DEV)
        _lib.check(lib.ldit_attention_bwd_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
                                               dqkv.data_ptr(), dqkv[..., HD:].data_ptr(), dqkv[..., 2 * HD:].data_ptr(), B, N, H, D,
                                               3 * HD, HD, HD, 3 * HD, D ** -0.5, _stream()))
        outs.append(dqkv)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(outs[0].float()).all()) and torch.equal(outs[0], outs[1])


# ---- LayerNorm backward -----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,Cc", [(394, 768), (5, 192), (3000, 1024), (68, 128), (9000, 768), (40, 2048)])
def test_layernorm_backward(rows, Cc):
    lib = _lib.load()
    x = _rand(11, rows, Cc) * 2.0 + 0.5
    x[0] *= 50.0
    g, dy, dh0 = 1.0 + 0.1 * _rand(12, Cc), _rand(13, rows, Cc), _rand(14, rows, Cc)
    xt = torch.from_numpy(x).double().requires_grad_(True)
    gt = torch.from_numpy(g).double().requires_grad_(True)
    bt = torch.zeros(Cc, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.layer_norm(xt, (Cc,), gt, bt, 1e-12)
    (y * torch.from_numpy(dy).double()).sum().backward()
    dh = torch.from_numpy(dh0).to(DEV)
    dg, db = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV)
    need = lib.ldit_layernorm_bwd_scratch_bytes(rows, Cc)
    scratch = torch.empty(need, dtype=torch.uint8, device=DEV)
    dyd, xd, gd = torch.from_numpy(dy).to(DEV), torch.from_numpy(x).to(DEV), torch.from_numpy(g).to(DEV)
    _lib.check(lib.ldit_layernorm_bwd_f32(dyd.data_ptr(), xd.data_ptr(), gd.data_ptr(), dh.data_ptr(), rows, Cc, 1e-12,
                                          dg.data_ptr(), db.data_ptr(), scratch.data_ptr(), need, _stream()))
    torch.cuda.synchronize()
    assert rel_l2(dh.cpu().numpy(), dh0.astype(np.float64) + xt.grad.numpy()) < 2e-5
    assert rel_l2(dg.cpu().numpy(), gt.grad.numpy()) < 2e-5
    assert rel_l2(db.cpu().numpy(), bt.grad.numpy()) < 2e-5


# ---- the bf16 GEMM's training epilogues ----------------------------------------------------------------------------------------
def _ex(x, w, epi, M, N, K, bias=None, lam=None, R=None, Y=None, Ypre=None, rowscale=None, aux=None, splits=1):
    lib = _lib.load()
    p = lambda t: None if t is None else t.data_ptr()       # noqa: E731
    _lib.check(lib.ldit_linear_bf16_ex(p(x), K, p(w), p(bias), p(Y), N, M, N, K, epi, p(lam), p(R), None, p(Ypre), p(rowscale),
                                       p(aux), N, splits, _stream()))


@pytest.mark.parametrize("M,N,K", [(394, 768, 256), (34, 512, 128), (1040, 384, 192), (300, 200, 64)])
def test_linear_bf16_training_epilogues(M, N, K):
    x, w, b = _bf(_rand(20, M, K)), _bf(_rand(21, N, K, scale=0.05)), torch.from_numpy(_rand(22, N, scale=0.1))
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    ref = x.double() @ w.double().T + b.double()
    # fp32 output
    y = torch.empty((M, N), device=DEV)
    _ex(xd, wd, _lib.EPI_F32, M, N, K, bias=bd, Y=y)
    assert rel_l2(y.cpu().numpy(), ref.numpy()) < 1e-5
    # GELU forward with the pre-activation copy
    g, pre = torch.empty((M, N), dtype=BF, device=DEV), torch.empty((M, N), dtype=BF, device=DEV)
    _ex(xd, wd, _lib.EPI_BIAS_GELU, M, N, K, bias=bd, Y=g, Ypre=pre)
    rt = ref.clone().requires_grad_(True)
    torch.nn.functional.gelu(rt).sum().backward()
    assert rel_l2(pre.float().cpu().numpy(), rt.grad.numpy()) < 3e-3          # the saved GELU derivative
    assert rel_l2(g.float().cpu().numpy(), torch.nn.functional.gelu(ref).numpy()) < 3e-3
    # LayerScale + residual with per-row stochastic-depth factors and the pre-LayerScale copy
    lam = torch.from_numpy(np.abs(_rand(23, N)) * 0.3 + 0.05)
    r = torch.from_numpy(_rand(24, M, N))
    rs = torch.from_numpy((np.arange(M) % 3 != 0).astype(np.float32) / 0.9)
    h, z = r.clone().to(DEV), torch.empty((M, N), dtype=BF, device=DEV)
    _ex(xd, wd, _lib.EPI_SCALE_RESID, M, N, K, bias=bd, lam=lam.to(DEV), R=h, Y=h, Ypre=z, rowscale=rs.to(DEV))
    want = r.double() + rs.double()[:, None] * lam.double() * ref
    assert rel_l2(h.cpu().numpy(), want.numpy()) < 1e-5
    assert rel_l2(z.float().cpu().numpy(), ref.numpy()) < 3e-3
    # dgrad times the saved GELU derivative
    a = _bf(_rand(25, M, N, scale=0.5) + 0.5)
    d = torch.empty((M, N), dtype=BF, device=DEV)
    _ex(xd, wd, _lib.EPI_GELU_BWD, M, N, K, Y=d, aux=a.to(DEV))
    assert rel_l2(d.float().cpu().numpy(), ((ref - b.double()) * a.double()).numpy()) < 3e-3


@pytest.mark.parametrize("M,N,K,splits", [(768, 512, 1344, 7), (128, 768, 128, 2), (384, 256, 12608, 16)])
def test_linear_bf16_split_k_slabs(M, N, K, splits):
    """wgrad shape: few output tiles, long K - K is split over workgroups into fp32 slabs, summed in a fixed order."""
    lib = _lib.load()
    x, w = _bf(_rand(30, M, K, scale=0.3)), _bf(_rand(31, N, K, scale=0.3))
    slabs = torch.full((splits, M, N), float("nan"), device=DEV)
    _ex(x.to(DEV), w.to(DEV), _lib.EPI_F32, M, N, K, Y=slabs, splits=splits)
    out = torch.empty((M, N), device=DEV)
    _lib.check(lib.ldit_reduce_slabs_f32(slabs.data_ptr(), out.data_ptr(), M * N, splits, _stream()))
    assert rel_l2(out.cpu().numpy(), (x.double() @ w.double().T).numpy()) < 1e-5
    out2 = torch.empty((M, N), device=DEV)
    _ex(x.to(DEV), w.to(DEV), _lib.EPI_F32, M, N, K, Y=slabs, splits=splits)
    _lib.check(lib.ldit_reduce_slabs_f32(slabs.data_ptr(), out2.data_ptr(), M * N, splits, _stream()))
    assert torch.equal(out, out2)                        # no atomics: bit-reproducible
    rc = lib.ldit_linear_bf16_ex(16, 64, 16, None, 16, 64, 64, 64, 64, _lib.EPI_BIAS, None, None, None, None, None, None, 0, 2, None)
    assert rc == _lib.LDIT_EINVAL                        # split-K needs the fp32 slab epilogue


# ---- AdamW -----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_fused_adamw_matches_torch(wd):
    lib = _lib.load()
    n = 4 * 1000 + 4
    p0, grads = _rand(40, n), [_rand(41 + i, n, scale=0.1) for i in range(3)]
    ref = torch.nn.Parameter(torch.from_numpy(p0.copy()).double())
    opt = torch.optim.AdamW([ref], lr=1e-2, weight_decay=wd, betas=(0.9, 0.999), eps=1e-8)
    p = torch.from_numpy(p0.copy()).to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for i, g in enumerate(grads):
        ref.grad = torch.from_numpy(g).double() * 0.5
        opt.step()
        gd = torch.from_numpy(g).to(DEV)
        mirror = torch.empty(n, dtype=BF, device=DEV)
        _lib.check(lib.ldit_adamw_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-2, 0.9, 0.999, 1e-8, wd,
                                       i + 1, 0.5, mirror.data_ptr() if i == 2 else None, _stream()))
    assert rel_l2(p.cpu().numpy(), ref.detach().numpy()) < 1e-6
    assert torch.equal(mirror, p.to(BF))                    # the bf16 mirror of the updated parameters, same pass
    assert lib.ldit_adamw_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-2, 0.9, 0.999, 1e-8, wd, 0, 1.0,
                               None, _stream()) == _lib.LDIT_EINVAL


# ---- whole path -----------------------------------------------------------------------------------------------------------------------
_KEYMAP = {"ln1_w": "layernorm_before.weight", "ln1_b": "layernorm_before.bias", "wq": "attention.attention.query.weight",
           "wk": "attention.attention.key.weight", "wv": "attention.attention.value.weight",
           "bq": "attention.attention.query.bias", "bv": "attention.attention.value.bias",
           "wo": "attention.output.dense.weight", "bo": "attention.output.dense.bias", "lam1": "lambda_1",
           "ln2_w": "layernorm_after.weight", "ln2_b": "layernorm_after.bias", "w1": "intermediate.dense.weight",
           "b1": "intermediate.dense.bias", "w2": "output.dense.weight", "b2": "output.dense.bias", "lam2": "lambda_2"}
_EMB = {"patch_w": "embeddings.patch_embeddings.projection.weight", "patch_b": "embeddings.patch_embeddings.projection.bias",
        "cls": "embeddings.cls_token", "pos": "embeddings.position_embeddings"}


def _hf_name(short: str) -> str:
    if short in _EMB:
        return _EMB[short]
    l, k = short.split(".")
    return f"encoder.layer.{l}.{_KEYMAP[k]}"


def _grads_through_autograd(cfg, w, x, dtaps, scales, monkeypatch):
    m = DiTEncoder(cfg, compute_dtype="bf16").load_numpy(w).to(DEV).train()
    if scales is not None:
        monkeypatch.setattr(training, "sample_drop_scales", lambda *a, **k: torch.from_numpy(scales).to(DEV).contiguous())
    else:
        m.config.drop_path_rate = 0.0
    out = m(torch.from_numpy(x).to(DEV))
    loss = sum((out.hidden_states[t] * torch.from_numpy(d).to(DEV)).sum() for t, d in zip(cfg.taps, dtaps))
    loss.backward()
    torch.cuda.synchronize()
    st = m._flat_state
    grads = {_hf_name(name): p.grad.detach().cpu().numpy() for name, p, _, _ in st.named}
    taps = [out.hidden_states[t].detach().cpu().numpy() for t in cfg.taps]
    assert m.embeddings.mask_token.grad is None and m.pooler.layernorm.weight.grad is None      # inert parameters
    return taps, grads, m


@pytest.mark.parametrize("name,geom", [("g6_grad_micro.npz", "micro"), ("g7_grad_tiny.npz", "tiny"), ("g8_grad_base.npz", "base")])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_parameter_gradients_vs_oracle_and_hf_golden(golden_dir, monkeypatch, name, geom, mode):
    g = np.load(os.path.join(golden_dir, name))
    cfg = cfgs.GEOMETRIES[geom]()
    B, size = int(g["geometry"][6]), int(g["geometry"][7])
    wseed, xseed, gseed = (int(v) for v in g["seeds"])
    w = synth.synth_weights(cfg, wseed)
    x = synth.synth_images(B, size, size, seed=xseed, kind="uniform" if size < 224 else "doc")
    dtaps = upstream(cfg, B, cfg.tokens(size, size), gseed)
    scales = None if mode == "eval" else g["train_drop_scales"]
    taps, grads, _ = _grads_through_autograd(cfg, w, x, dtaps, scales, monkeypatch)
    ref_taps, ref = train_reference(cfg, w, x, dtaps, drop_scales=scales)
    for a, r in zip(taps, ref_taps):
        assert rel_l2(a, r) < 2e-2
    stride = int(g["stride"][0])
    worst = {}
    for k, r in ref.items():
        if k.endswith("mask_token"):
            continue
        got = grads[k]
        assert got.shape == r.shape, k
        e = rel_l2(got, r)
        worst[k.split(".")[-2] + "." + k.split(".")[-1]] = max(worst.get(k.split(".")[-2] + "." + k.split(".")[-1], 0.0), e)
        assert e < GRAD_TOL, (k, e)
        hf = g[f"{mode}_grad/{k}"]                       # HF BeitModel's own gradient, strided sample
        assert rel_l2(got.reshape(-1)[::sample_stride(g, got.size)], hf) < GRAD_TOL, (k, "vs HF golden")
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/grad_parity_{geom}_{mode}.txt", "w") as f:
        for k, e in sorted(worst.items(), key=lambda kv: -kv[1]):
            f.write(f"{k:40s} {e:.3e}\n")


def test_config2_full_size_train_step_vs_hf_golden_and_properties(golden_dir):
    """BASELINE configs[2] at its OWN geometry and batch: ViT-B/16 224x224 bs=64 bf16, forward + backward + fused AdamW
    (ref trainer.py:148-187 on self.dit).  An oracle run of 64 images is minutes of CPU time, so the full batch is pinned
    through properties of the domain:
      * linearity of the loss in the upstream gradients: with d loss / d taps zero for images 2..63, every parameter gradient
        of the 64-image step is the gradient of the 2-image problem of tests/golden/g8_grad_base.npz - compared directly with
        HF BeitModel's gradients (eval arithmetic; 62 more images ride through every GEMM, attention and reduction);
      * batch invariance: taps of images 0..1 equal the bs=2 run bit for bit;
      * the backward is bit-reproducible (no atomics) with generic upstream gradients on all 64 images;
      * one fused AdamW step equals torch.optim.AdamW on the whole flat block, and its bf16 mirror is the rounded block."""
    g = np.load(os.path.join(golden_dir, "g8_grad_base.npz"))
    cfg = cfgs.vit_base()
    wseed, xseed, gseed = (int(v) for v in g["seeds"])
    B = 64
    w = synth.synth_weights(cfg, wseed)
    x = torch.from_numpy(synth.synth_images(B, 224, 224, seed=xseed, kind="doc")).to(DEV)
    N, Cc, L = cfg.tokens(224, 224), cfg.hidden_size, cfg.num_hidden_layers
    m = DiTEncoder(cfg, compute_dtype="bf16").load_numpy(w).to(DEV).train()
    m.config.drop_path_rate = 0.0
    st = training.flat_state(m, 224, 224)
    st.repack()
    saved = st.new_saved(B)
    # -- 1. linearity: upstream gradients on images 0..1 only = the golden's own problem
    d2 = upstream(cfg, 2, N, gseed)
    dt = []
    for d in d2:
        t = torch.zeros((B, N, Cc), device=DEV)
        t[:2] = torch.from_numpy(d).to(DEV)
        dt.append(t)
    taps64 = st.forward(x, cfg.taps, None, saved)
    st.backward(x, cfg.taps, dt, None, saved, L, 0)
    torch.cuda.synchronize()
    stride = int(g["stride"][0])
    worst = 0.0
    for name, p, off, shape in st.named:
        k = _hf_name(name)
        got = st.grad_view(off, shape).cpu().numpy()
        e = rel_l2(got.reshape(-1)[::sample_stride(g, got.size)], g[f"eval_grad/{k}"])
        worst = max(worst, e)
        assert e < GRAD_TOL, (k, e, "bs=64 step vs HF golden of its first two images")
    # -- 1b. the same two pages as the LAST two images of the batch (images reversed, upstream gradients on images 63, 62): every
    #        reduction over tokens - weight gradients, bias column sums out of the GEMM epilogues, LayerNorm partial rows - must pick
    #        up its last row tiles and partial rows as it does its first (round 4: the fc1 bias gradient once summed fewer partial
    #        rows than the 256-row dgrad tiles had written; with gradients on images 0..1 only nothing noticed)
    dt = []
    for d in d2:
        t = torch.zeros((B, N, Cc), device=DEV)
        t[B - 2:] = torch.from_numpy(d).to(DEV).flip(0)
        dt.append(t)
    xr = x.flip(0).contiguous()
    st.forward(xr, cfg.taps, None, saved)
    st.backward(xr, cfg.taps, dt, None, saved, L, 0)
    torch.cuda.synchronize()
    for name, p, off, shape in st.named:
        k = _hf_name(name)
        got = st.grad_view(off, shape).cpu().numpy()
        e = rel_l2(got.reshape(-1)[::sample_stride(g, got.size)], g[f"eval_grad/{k}"])
        assert e < GRAD_TOL, (k, e, "bs=64 step, the golden's two pages as images 63, 62")
    # -- 2. batch invariance of the training forward
    saved2 = st.new_saved(2)
    taps2 = st.forward(x[:2].contiguous(), cfg.taps, None, saved2)
    for a, b, t in zip(taps64, taps2, cfg.taps):
        assert torch.equal(a[:2], b), t
        assert rel_l2(b.cpu().numpy().reshape(-1)[::max(stride, 7)], g[f"eval_tap{t}_sample"]) < 2e-2
    del saved2
    # -- 3. generic upstream gradients on every image, stochastic depth on: rerun-bit-equal backward
    gen = torch.Generator(device=DEV)
    gen.manual_seed(5)
    dt = [torch.randn((B, N, Cc), device=DEV, generator=gen) / (B * N * Cc) ** 0.5 for _ in cfg.taps]
    drop = training.sample_drop_scales(L, B, 0.1, DEV, gen)
    assert bool((drop == 0).any())
    step = training.TrainStep(m, lr=1e-4, weight_decay=0.0, dtaps=dt, drop_path_rate=0.1)
    st.forward(x, cfg.taps, drop, saved)
    st.backward(x, cfg.taps, dt, drop, saved, L, 0)
    g1 = st.grads.clone()
    assert bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0
    p0 = st.params.clone()
    # -- 4. the fused step (same inputs): same gradients, AdamW == torch.optim.AdamW, mirror == rounded parameters
    step.step(x, drop_scales=drop)
    torch.cuda.synchronize()
    assert torch.equal(st.grads, g1)
    ref = torch.nn.Parameter(p0.clone())
    ref.grad = g1.clone()
    torch.optim.AdamW([ref], lr=1e-4, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8).step()
    diff = (st.params - ref.detach()).abs()
    assert float(diff.max()) <= 2.0 ** -22 * max(1.0, float(p0.abs().max()))        # <= 2 ulp of the largest parameter
    assert float((st.params - p0).abs().max()) > 5e-5                                # a first Adam step moves ~lr
    mirror = st.packed.view(BF)[: st.numel]
    assert torch.equal(mirror, st.params.to(BF))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/grad_parity_base_bs64.txt", "w") as f:
        f.write(f"worst per-tensor rel-L2 vs HF golden (strided): {worst:.3e}\n")


def test_key_bias_slot_and_reproducibility(monkeypatch):
    """The fused bias gradient's key third is exactly zero (BEiT has no key bias, TF:306) and two backward passes over the
    same inputs give bit-identical gradients (no atomics anywhere in the backward)."""
    cfg = cfgs.vit_micro()
    w = synth.synth_weights(cfg, 3)
    x = synth.synth_images(4, 64, 64, seed=5, kind="uniform")
    dtaps = upstream(cfg, 4, cfg.tokens(64, 64), 9)
    _, g1, m = _grads_through_autograd(cfg, w, x, dtaps, None, monkeypatch)
    st = m._flat_state
    Cc = cfg.hidden_size
    flat1 = st.grads.clone()
    for l in range(cfg.num_hidden_layers):
        off = st.offsets[4 + 14 * l + 3]
        assert bool((st.grads[off + Cc: off + 2 * Cc] == 0).all())
    for p in m.parameters():
        p.grad = None
    out = m(torch.from_numpy(x).to(DEV))
    sum((out.hidden_states[t] * torch.from_numpy(d).to(DEV)).sum() for t, d in zip(cfg.taps, dtaps)).backward()
    torch.cuda.synchronize()
    assert torch.equal(st.grads, flat1)


def test_reference_style_loop_with_torch_adamw_and_fused_step_agree(monkeypatch):
    """Drop-in: the reference's loop shape - loss.backward() then torch.optim.AdamW(lr 1e-4, wd 0).step()
    (ref trainer.py:62-68,169-180) - on DiTEncoder in train mode, against TrainStep's fused AdamW on the same gradients."""
    cfg = cfgs.vit_micro()
    cfg.drop_path_rate = 0.0
    # distinct taps: with the micro geometry's duplicated tap (d/3 = d/2 = 1) autograd hands over ONE summed upstream gradient
    # where the fused step adds two - a 1-ulp reassociation that Adam's sign-like first steps amplify on near-zero gradients
    cfg.taps = [1, 2, 3]
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
        opt.step()
        fused.step(x)
        if it == 0:
            # same gradients bit for bit -> the two optimizers differ only in the rounding of lr / (1 - beta1^t)
            assert torch.equal(a._flat_state.grads, fused.state.grads)
            assert float((a._flat_state.params - fused.state.params).abs().max()) < 2.5e-7      # <= 2 ulp of |p| < 2
    torch.cuda.synchronize()
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        # three steps: the 1-ulp parameter differences of step 1 perturb later gradients, and Adam's sign-like early steps
        # amplify that on near-zero gradient entries (cls / pos): agreement to 1e-3 of the tensor, not bit level
        assert rel_l2(sb[k].cpu().numpy(), sa[k].cpu().numpy()) < 1e-3, k
    moved = rel_l2(sa["encoder.layer.0.intermediate.dense.weight"].cpu().numpy(), w["encoder.layer.0.intermediate.dense.weight"])
    assert moved > 1e-3                                    # the parameters really were updated
    # the inference path sees the updated weights (its packed copy is invalidated by the step)
    b.eval()
    a.eval()
    with torch.no_grad():
        ha, hb = a(x).hidden_states, b(x).hidden_states
    for t in cfg.taps:
        assert rel_l2(hb[t].cpu().numpy(), ha[t].cpu().numpy()) < 2e-3
    with torch.no_grad():
        stale = DiTEncoder(cfg, compute_dtype="bf16").load_numpy(w).to(DEV).eval()(x).hidden_states
    assert rel_l2(hb[cfg.taps[-1]].cpu().numpy(), stale[cfg.taps[-1]].cpu().numpy()) > 1e-2      # ... not the initial ones


def test_train_mode_without_grad_applies_stochastic_depth(monkeypatch):
    cfg = cfgs.vit_micro()
    w = synth.synth_weights(cfg, 3)
    x = synth.synth_images(4, 64, 64, seed=5, kind="uniform")
    scales = np.ones((cfg.num_hidden_layers, 2, 4), dtype=np.float32)
    scales[1, 0, 2] = 0.0
    scales[2, 1, 0] = 1.0 / 0.9
    monkeypatch.setattr(training, "sample_drop_scales", lambda *a, **k: torch.from_numpy(scales).to(DEV).contiguous())
    m = DiTEncoder(cfg, compute_dtype="bf16").load_numpy(w).to(DEV).train()
    with torch.no_grad():
        hs = m(torch.from_numpy(x).to(DEV)).hidden_states
    ref, _ = train_reference(cfg, w, x, [np.zeros((4, 17, 128), np.float32)] * 4, drop_scales=scales)
    for t, r in zip(cfg.taps, ref):
        assert rel_l2(hs[t].cpu().numpy(), r) < 2e-2
    # the f32 build trains (and applies stochastic depth) on the same bf16-operand kernels; the fp8 build refuses
    m32 = DiTEncoder(cfg).load_numpy(w).to(DEV).train()
    with torch.no_grad():
        hs32 = m32(torch.from_numpy(x).to(DEV)).hidden_states
    for t in cfg.taps:
        assert torch.equal(hs32[t], hs[t])
    m8 = DiTEncoder(cfg, compute_dtype="fp8").load_numpy(w).to(DEV).train()
    with pytest.raises(NotImplementedError, match="inference only|fp8"):
        m8(torch.from_numpy(x).to(DEV))
    # with a drop-path rate of 0 train == eval arithmetic: the inference kernels of the build run (any grid)
    x96 = torch.from_numpy(synth.synth_images(2, 96, 96, seed=5, kind="uniform")).to(DEV)
    m32.config.drop_path_rate = 0.0
    with torch.no_grad():
        a = m32(x96).hidden_states[cfg.taps[-1]]
        b = m32.eval()(x96).hidden_states[cfg.taps[-1]]
    assert torch.equal(a, b)


def test_f32_build_trains_on_the_bf16_kernels_like_the_reference_cpu_branch(monkeypatch):
    """`DiTEncoder(cfg)` (compute_dtype 'f32') + .train() + loss.backward() (ref trainer.py:171-172,176-178): mixed precision
    on the bf16 train kernels, every parameter gradient inside the bf16 gate of the float64 oracle; a GradScaler-style scaled
    loss (ref trainer.py:177-180) gives the same gradients times the scale."""
    cfg = cfgs.vit_micro()
    w = synth.synth_weights(cfg, 3)
    x = synth.synth_images(4, 64, 64, seed=5, kind="uniform")
    dtaps = upstream(cfg, 4, cfg.tokens(64, 64), 9)
    m = DiTEncoder(cfg).load_numpy(w).to(DEV).train()
    m.config.drop_path_rate = 0.0
    out = m(torch.from_numpy(x).to(DEV))
    loss = sum((out.hidden_states[t] * torch.from_numpy(d).to(DEV)).sum() for t, d in zip(cfg.taps, dtaps))
    (loss * 1024.0).backward()
    _, ref = train_reference(cfg, w, x, dtaps, drop_scales=None)
    st = m._flat_state
    for name, p, _, _ in st.named:
        k = _hf_name(name)
        assert rel_l2(p.grad.cpu().numpy() / 1024.0, ref[k]) < GRAD_TOL, k
    with pytest.raises(RuntimeError, match="second time"):
        loss.backward()
    # eval mode of the same module is still the exact-fp32 path
    m.eval()
    with torch.no_grad():
        h = m(torch.from_numpy(x).to(DEV)).hidden_states[cfg.taps[-1]]
    ref_taps, _ = train_reference(cfg, w, x, dtaps, drop_scales=None)
    assert rel_l2(h.cpu().numpy(), ref_taps[-1]) < 2e-5


@pytest.mark.parametrize("size,B", [(96, 3), (272, 2)])
def test_training_at_other_grids_and_beyond_256_tokens(size, B):
    """Round 3: the train path is no longer tied to the position table's own grid or to 256 tokens (ViT-L/16 at 512 x 512 = 1025
    tokens trains).  Micro geometry at 96 x 96 (grid 6 x 6: the position table is resampled bicubically, TF:113-151, and its
    gradient travels back through the resample's adjoint) and at 272 x 272 (290 tokens: the blocked attention backward):
    every parameter gradient, embeddings.position_embeddings included, against the float64 autograd oracle."""
    cfg = cfgs.vit_micro()
    cfg.drop_path_rate = 0.0
    w = synth.synth_weights(cfg, 3)
    x = synth.synth_images(B, size, size, seed=5, kind="uniform")
    N = cfg.tokens(size, size)
    dtaps = upstream(cfg, B, N, 9)
    m = DiTEncoder(cfg, compute_dtype="bf16").load_numpy(w).to(DEV).train()
    out = m(torch.from_numpy(x).to(DEV))
    sum((out.hidden_states[t] * torch.from_numpy(d).to(DEV)).sum() for t, d in zip(cfg.taps, dtaps)).backward()
    torch.cuda.synchronize()
    ref_taps, ref = train_reference(cfg, w, x, dtaps)
    for t, r in zip(cfg.taps, ref_taps):
        assert rel_l2(out.hidden_states[t].detach().cpu().numpy(), r) < 2e-2
    got = {k: p.grad for k, p in m.state_dict(keep_vars=True).items() if p.grad is not None}
    assert tuple(got["embeddings.position_embeddings"].shape) == (1, 17, cfg.hidden_size)
    for k, r in ref.items():
        if k.endswith("key.bias"):
            continue
        assert k in got, k
        assert rel_l2(got[k].cpu().numpy(), r) < GRAD_TOL, (k, rel_l2(got[k].cpu().numpy(), r))
    with pytest.raises(NotImplementedError, match="position table's own"):
        training.TrainStep(m, lr=1e-3, img_size=(size, size))          # the fused step updates the flat block: native grid only


def test_vit_large_512_trains():
    """ViT-L/16 at 512 x 512 (BASELINE configs[3]'s geometry, 1025 tokens, resampled 32 x 32 position grid) through
    loss.backward(): sampled parameter gradients against the float64 autograd oracle (one image: the CPU oracle run is the
    long pole of this test), bit-reproducible."""
    cfg = cfgs.vit_large()
    cfg.drop_path_rate = 0.0
    w = synth.synth_weights(cfg, 3)
    x = synth.synth_images(1, 512, 512, seed=1234)
    N = cfg.tokens(512, 512)
    assert N == 1025
    dtaps = upstream(cfg, 1, N, 11)
    m = DiTEncoder(cfg, compute_dtype="bf16").load_numpy(w).to(DEV).train()
    xd = torch.from_numpy(x).to(DEV)
    dd = [torch.from_numpy(d).to(DEV) for d in dtaps]
    out = m(xd)
    sum((out.hidden_states[t] * d).sum() for t, d in zip(cfg.taps, dd)).backward()
    torch.cuda.synchronize()
    first = m._flat_state.grads.clone()
    got = {k: p.grad.clone() for k, p in m.state_dict(keep_vars=True).items() if p.grad is not None}
    for p in m.parameters():
        p.grad = None
    out = m(xd)
    sum((out.hidden_states[t] * d).sum() for t, d in zip(cfg.taps, dd)).backward()
    torch.cuda.synchronize()
    assert torch.equal(m._flat_state.grads, first)
    _, ref = train_reference(cfg, w, x, dtaps)
    for k in ("embeddings.position_embeddings", "embeddings.patch_embeddings.projection.weight", "encoder.layer.0.attention.attention.query.weight",
              "encoder.layer.11.attention.attention.key.weight", "encoder.layer.23.intermediate.dense.weight", "encoder.layer.5.lambda_1",
              "encoder.layer.17.layernorm_after.bias", "encoder.layer.23.attention.output.dense.bias"):
        assert rel_l2(got[k].cpu().numpy(), ref[k]) < GRAD_TOL, (k, rel_l2(got[k].cpu().numpy(), ref[k]))


def test_caches_follow_parameter_updates_that_bypass_version_counters():
    """the fused AdamW updates parameters through raw pointers (no tensor version moves).  After one fused step
    at 224-style native size, an eval forward at ANOTHER grid must use a position table resampled from the UPDATED
    embeddings; `.data` writes need FlatState.mark_dirty() to reach the bf16 mirror."""
    cfg = cfgs.vit_micro()
    cfg.drop_path_rate = 0.0
    w = synth.synth_weights(cfg, 3)
    x = torch.from_numpy(synth.synth_images(4, 64, 64, seed=5, kind="uniform")).to(DEV)
    x96 = torch.from_numpy(synth.synth_images(2, 96, 96, seed=6, kind="uniform")).to(DEV)
    m = DiTEncoder(cfg, compute_dtype="bf16").load_numpy(w).to(DEV)
    with torch.no_grad():
        before = m.eval()(x96).hidden_states[cfg.taps[-1]].clone()      # fills the resampled-table cache
    step = training.TrainStep(m.train(), lr=5e-2, weight_decay=0.0, drop_path_rate=0.0, img_size=(64, 64))
    step.step(x)
    torch.cuda.synchronize()
    fresh = DiTEncoder(cfg, compute_dtype="bf16").to(DEV)
    fresh.load_state_dict(m.state_dict())
    with torch.no_grad():
        got = m.eval()(x96).hidden_states[cfg.taps[-1]]
        want = fresh.eval()(x96).hidden_states[cfg.taps[-1]]
    assert torch.equal(got, want)
    assert rel_l2(got.cpu().numpy(), before.cpu().numpy()) > 1e-3          # the step really moved the output
    # .data writes bypass the version counter: mark_parameters_changed() is the documented hook for every cache
    m.train()
    a = m(x).hidden_states[cfg.taps[-1]].detach().clone()                 # autograd path: reads the flat state's bf16 mirror
    m.encoder.layer[0].intermediate.dense.weight.data.mul_(1.5)
    m.mark_parameters_changed()
    b = m(x).hidden_states[cfg.taps[-1]].detach()
    assert not torch.equal(a, b)
    fresh = DiTEncoder(cfg, compute_dtype="bf16").to(DEV)
    fresh.load_state_dict(m.state_dict())
    with torch.no_grad():
        assert torch.equal(m.eval()(x96).hidden_states[cfg.taps[-1]], fresh.eval()(x96).hidden_states[cfg.taps[-1]])


# ---- reduction-major GEMMs (dgrad / wgrad without transposed copies) --------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,epi", [(394, 768, 256, "f32"), (1040, 384, 192, "bf16"), (300, 200, 64, "gelu"), (12608, 768, 3072, "f32"),
                                       (68, 512, 128, "gelu")])
def test_dgrad_reads_the_weight_as_stored(M, N, K, epi):
    """dX[M, N] = dY[M, K] . W[K, N] with W in nn.Linear's own [reduction, output] layout (ds_read_b64_tr_b16 gathers it)."""
    lib = _lib.load()
    dy, w = _bf(_rand(60, M, K, scale=0.5)), _bf(_rand(61, K, N, scale=0.05))
    zeros = torch.zeros(64, device=DEV)
    ref = dy.double() @ w.double()
    code = {"f32": _lib.EPI_F32, "bf16": _lib.EPI_BIAS, "gelu": _lib.EPI_GELU_BWD}[epi]
    out = torch.full((M, N), float("nan"), dtype=torch.float32 if epi == "f32" else BF, device=DEV)
    aux = _bf(_rand(62, M, N, scale=0.5) + 0.5).to(DEV) if epi == "gelu" else None
    dyd, wd = dy.to(DEV), w.to(DEV)
    _lib.check(lib.ldit_linear_bf16_tr(dyd.data_ptr(), K, 0, wd.data_ptr(), N, out.data_ptr(), N, M, N, K, code,
                                       None if aux is None else aux.data_ptr(), 1, zeros.data_ptr(), _stream()))
    if aux is not None:
        ref = ref * aux.cpu().double()
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < (1e-5 if epi == "f32" else 3e-3)


@pytest.mark.parametrize("T,Nout,Kout,splits", [(12608, 768, 3072, 3), (68, 128, 512, 2), (197, 384, 128, 1), (1000, 256, 768, 8),
                                               (12544, 768, 768, 8)])
def test_wgrad_reads_both_operands_token_major(T, Nout, Kout, splits):
    """dW[Nout, Kout] = dY[T, Nout]^T . X[T, Kout]: both operands reduction-major, tokens not a multiple of 64 (zero page),
    K split over workgroups into slabs summed in a fixed order."""
    lib = _lib.load()
    dy, x = _bf(_rand(70, T, Nout, scale=0.3)), _bf(_rand(71, T, Kout, scale=0.3))
    zeros = torch.zeros(64, device=DEV)
    dyd, xd = dy.to(DEV), x.to(DEV)
    slabs = torch.full((splits, Nout, Kout), float("nan"), device=DEV)
    _lib.check(lib.ldit_linear_bf16_tr(dyd.data_ptr(), Nout, 1, xd.data_ptr(), Kout, slabs.data_ptr(), Kout, Nout, Kout, T, _lib.EPI_F32,
                                       None, splits, zeros.data_ptr(), _stream()))
    out = slabs[0] if splits == 1 else torch.empty((Nout, Kout), device=DEV)
    if splits > 1:
        _lib.check(lib.ldit_reduce_slabs_f32(slabs.data_ptr(), out.data_ptr(), Nout * Kout, splits, _stream()))
    assert rel_l2(out.cpu().numpy(), (dy.double().T @ x.double()).numpy()) < 1e-5


@pytest.mark.parametrize("form,tiles", [("dgrad", (2, 3, 4, 5)), ("wgrad", (2, 3, 6))])
def test_every_reduction_major_tile_gives_the_same_bits(form, tiles):
    """Every tile of gemm_bf16_tr (128 x 128, 256 x 256, the 192- / 320-row dgrad tiles, the 256 x 128 wgrad tile) sums a dot product
    in the same order: forcing each gives the default choice's output bit for bit, ragged row and column tiles included."""
    lib = _lib.load()
    zeros = torch.zeros(64, device=DEV)
    outs = {}
    try:
        for tile in (None,) + tuple(tiles):
            _lib.set_switch("LDIT_GEMM_BF16_TR_TILE", tile)
            if form == "dgrad":
                M, N, K = 1300, 520, 256
                dy, w = _bf(_rand(80, M, K, scale=0.5)).to(DEV), _bf(_rand(81, K, N, scale=0.05)).to(DEV)
                aux = _bf(_rand(82, M, N, scale=0.5) + 0.5).to(DEV)
                got = []
                for code, dt in ((_lib.EPI_F32, torch.float32), (_lib.EPI_BIAS, BF), (_lib.EPI_GELU_BWD, BF)):
                    out = torch.full((M, N), float("nan"), dtype=dt, device=DEV)
                    _lib.check(lib.ldit_linear_bf16_tr(dy.data_ptr(), K, 0, w.data_ptr(), N, out.data_ptr(), N, M, N, K, code,
                                                       aux.data_ptr() if code == _lib.EPI_GELU_BWD else None, 1, zeros.data_ptr(), _stream()))
                    got.append(out)
                if tile is None:
                    assert rel_l2(got[0].cpu().numpy(), (dy.double() @ w.double()).cpu().numpy()) < 1e-5
            else:
                T, Nout, Kout, splits = 1000, 264, 520, 2
                dy, x = _bf(_rand(83, T, Nout, scale=0.3)).to(DEV), _bf(_rand(84, T, Kout, scale=0.3)).to(DEV)
                slabs = torch.full((splits, Nout, Kout), float("nan"), device=DEV)
                _lib.check(lib.ldit_linear_bf16_tr(dy.data_ptr(), Nout, 1, x.data_ptr(), Kout, slabs.data_ptr(), Kout, Nout, Kout, T,
                                                   _lib.EPI_F32, None, splits, zeros.data_ptr(), _stream()))
                got = [slabs]
                if tile is None:
                    assert rel_l2(slabs.sum(0).cpu().numpy(), (dy.double().T @ x.double()).cpu().numpy()) < 1e-5
            torch.cuda.synchronize()
            outs[tile] = got
    finally:
        _lib.set_switch("LDIT_GEMM_BF16_TR_TILE", None)
    for tile in tiles:
        for a, b in zip(outs[None], outs[tile]):
            assert torch.equal(a, b), (form, tile)
