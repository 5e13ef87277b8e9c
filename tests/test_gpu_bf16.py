"""bf16 path, per-kernel parity on the GPU (through the C ABI).  Inputs are rounded to bf16 FIRST and the oracle is run
on the rounded values, so the only differences left are fp32 accumulation order (fp32 outputs: rel-L2 <= 1e-5) and the
final bf16 rounding of the result (bf16 outputs: one ulp = 2^-8 relative -> rel-L2 <= 3e-3)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from layoutdit_amd import _lib, ops, synth          # noqa: E402
from oracle import oracle                           # noqa: E402
from tests.util import max_rel, rel_l2              # noqa: E402

DEV = "cuda:0"


def _rand(seed, *shape, scale=1.0):
    n = int(np.prod(shape))
    return (scale * synth.normal(seed, 9, n)).astype(np.float32).reshape(shape)


def _bf16_round(a):
    return torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()


@pytest.fixture(autouse=True)
def _cleanup():
    _lib.load()
    yield
    _lib.set_switch("LDIT_GEMM_BF16_TILE", None)


def test_cast_matches_torch_round_to_nearest_even():
    x = _rand(1, 1000, 37, scale=3.0)
    x.reshape(-1)[:4] = [1.00390625, 1.01171875, -0.0, 3.3895314e38]      # ties and the largest finite bf16 neighbourhood
    got = ops.cast_bf16(torch.from_numpy(x).to(DEV)).cpu()
    assert torch.equal(got, torch.from_numpy(x).to(torch.bfloat16))


@pytest.mark.parametrize("tile", ["auto", "1", "2", "3", "4", "5"])
@pytest.mark.parametrize("M,N,K", [(37, 50, 64), (394, 576, 192), (1025, 1024, 1024), (513, 3072, 768), (300, 768, 4096)])
def test_linear_bf16_bias(M, N, K, tile):
    if tile != "auto":
        _lib.set_switch("LDIT_GEMM_BF16_TILE", tile)
    x, w, b = _bf16_round(_rand(1, M, K)), _bf16_round(_rand(2, N, K, scale=0.05)), _rand(3, N, scale=0.1)
    xd, wd = torch.from_numpy(x).to(DEV).to(torch.bfloat16), torch.from_numpy(w).to(DEV).to(torch.bfloat16)
    y = ops.linear_bf16(xd, wd, torch.from_numpy(b).to(DEV)).float().cpu().numpy()
    ref = oracle.linear(x, w, b)
    assert rel_l2(y, ref) < 3e-3
    assert max_rel(y, ref) < 1.6e-2        # <= one bf16 ulp (2^-8) of max(|ref|, 1), plus accumulation noise


@pytest.mark.parametrize("tile", ["1", "2", "3", "4", "5"])
def test_linear_bf16_gelu_and_residual(tile):
    _lib.set_switch("LDIT_GEMM_BF16_TILE", tile)
    M, N, K = 394, 320, 128
    x, w, b = _bf16_round(_rand(4, M, K)), _bf16_round(_rand(5, N, K, scale=0.2)), _rand(6, N)
    xd, wd, bd = (torch.from_numpy(x).to(DEV).to(torch.bfloat16), torch.from_numpy(w).to(DEV).to(torch.bfloat16),
                  torch.from_numpy(b).to(DEV))
    y = ops.linear_bf16(xd, wd, bd, epilogue=_lib.EPI_BIAS_GELU).float().cpu().numpy()
    assert rel_l2(y, oracle.gelu(oracle.linear(x, w, b))) < 3e-3
    lam, r = np.abs(_rand(7, N)) * 0.3 + 0.05, _rand(8, M, N)
    h = torch.from_numpy(r).to(DEV)
    tap = torch.full((M, N), float("nan"), device=DEV)
    out = ops.linear_bf16(xd, wd, bd, epilogue=_lib.EPI_SCALE_RESID, lam=torch.from_numpy(lam).to(DEV), residual=h, out=h,
                          out2=tap)
    ref = (r.astype(np.float64) + lam.astype(np.float64) * oracle.linear(x, w, b).astype(np.float64)).astype(np.float32)
    assert out.data_ptr() == h.data_ptr() and out.dtype == torch.float32
    assert rel_l2(h.cpu().numpy(), ref) < 1e-5              # fp32 result of exact bf16 products: only summation order differs
    np.testing.assert_array_equal(tap.cpu().numpy(), h.cpu().numpy())


@pytest.mark.parametrize("B,N,H", [(1, 197, 3), (2, 33, 2), (1, 256, 1), (1, 257, 2), (1, 1025, 3)])
def test_attention_bf16(B, N, H):
    """bf16 operands (rounded first), fp32 softmax; P is rounded to bf16 before the second product (2^-9 relative per
    term, averaged over the row) and the output is rounded to bf16: gate rel-L2 <= 6e-3."""
    D = 64
    q, k, v = (_bf16_round(_rand(20 + i, B, N, H * D)) for i in range(3))
    dev = [torch.from_numpy(a).to(DEV).to(torch.bfloat16) for a in (q, k, v)]
    o = ops.attention_bf16(*dev, heads=H).float().cpu().numpy()
    ref = oracle.attention(q, k, v, heads=H)
    assert rel_l2(o, ref) < 6e-3
    assert np.isfinite(o).all()


@pytest.mark.parametrize("N", [197, 130, 256, 257, 100])
def test_attention_bf16_workgroup_width_does_not_change_a_bit(N):
    """Five to eight query tiles (N = 129 .. 256) run as one eight-wave workgroup per (image, head), everything else on four-wave
    workgroups (attention_bf16.hip, launch_attn).  A query tile's arithmetic does not depend on its workgroup: both widths forced,
    both entry modes, bit-equal - so batch and shape invariance of the bf16 / fp8 builds and of the train step are untouched."""
    B, H, D = 2, 3, 64
    q, k, v = (torch.from_numpy(_bf16_round(_rand(220 + i, B, N, H * D))).to(DEV).to(torch.bfloat16) for i in range(3))
    got = {}
    for nw in ("4", "8", None):
        _lib.set_switch("LDIT_ATTN_BF16_NW", nw)
        got[nw] = (ops.attention_bf16(q, k, v, heads=H), ops.attention_bf16(q, k, v, heads=H, prescaled=True))
    _lib.set_switch("LDIT_ATTN_BF16_NW", None)
    for mode in (0, 1):
        assert torch.equal(got["4"][mode], got["8"][mode]) and torch.equal(got["4"][mode], got[None][mode])


@pytest.mark.parametrize("B,N,H", [(1, 197, 3), (2, 33, 2), (1, 64, 1), (1, 257, 2), (1, 1025, 3), (2, 128, 2)])
def test_attention_bf16_prescaled_queries(B, N, H):
    """The packed inference path hands the kernel q' = (scale log2 e) q (folded into W_q at pack time) and scale = 0: scores are
    exp2-domain exponents, the running maximum is subtracted by an extra MFMA step (two bf16 terms).  Against the float64
    softmax of the ROUNDED q' (what the kernel sees), every regime of the maximum logic forced by the data (rule 26):
      * a dominant key in the LAST chunk (the maximum jumps by far more than the deferred-rescale threshold 2^8),
      * all scores of the FIRST chunk very negative (the first chunk re-bases on its own maximum; nothing may underflow to l = 0),
      * scores that creep up chunk by chunk (many small growths below the threshold, then one above)."""
    D = 64
    c = D ** -0.5 * np.log2(np.e)
    q, k, v = (_rand(120 + i, B, N, H * D) for i in range(3))
    k[:, N - 3] *= 5.0                                         # dominant key in the last chunk
    if N > 64:
        k[:, :64] = -np.abs(k[:, :64]) * 1.5                   # first chunk: strongly negative against a positive query block
        q[:, ::2] = np.abs(q[:, ::2]) * 1.2
    k *= np.linspace(0.6, 1.6, N, dtype=np.float32)[None, :, None]     # creeping growth
    qp = _bf16_round((q * c).astype(np.float32))
    k, v = _bf16_round(k), _bf16_round(v)
    dev = [torch.from_numpy(a).to(DEV).to(torch.bfloat16) for a in (qp, k, v)]
    o = ops.attention_bf16(*dev, heads=H, prescaled=True).float().cpu().numpy()
    # float64 reference: softmax of the natural-log scores ln(2) * (q' . k)
    qh, kh, vh = (torch.from_numpy(a).double().view(B, N, H, D).transpose(1, 2) for a in (qp, k, v))
    ref = (torch.softmax(np.log(2.0) * (qh @ kh.transpose(-1, -2)), -1) @ vh).transpose(1, 2).reshape(B, N, H * D).numpy()
    assert np.isfinite(o).all()
    assert rel_l2(o, ref) < 6e-3
    # and the plain entry (scale applied in the kernel) agrees with it on the unscaled queries' own rounding
    o2 = ops.attention_bf16(torch.from_numpy(_bf16_round(q)).to(DEV).to(torch.bfloat16), dev[1], dev[2], heads=H).float().cpu().numpy()
    assert rel_l2(o2, o) < 1.5e-2


def test_attention_bf16_on_fused_views_and_large_logits():
    B, N, H, D = 2, 300, 2, 64
    qkv = _bf16_round(_rand(30, B, N, 3 * H * D))
    qkv[0, 290, H * D:2 * H * D] *= 6.0           # a dominant key in the last chunk: forces the online rescale
    t = torch.from_numpy(qkv).to(DEV).to(torch.bfloat16)
    qkv = t.float().cpu().numpy()
    C_ = H * D
    o = ops.attention_bf16(t[..., :C_], t[..., C_:2 * C_], t[..., 2 * C_:], heads=H).float().cpu().numpy()
    ref = oracle.attention(qkv[..., :C_], qkv[..., C_:2 * C_], qkv[..., 2 * C_:], heads=H)
    assert rel_l2(o, ref) < 6e-3


@pytest.mark.parametrize("geom,size,B", [("tiny", 224, 2), ("base", 224, 2)])
def test_forward_bf16_vs_fp32_oracle(geom, size, B):
    """Whole path in bf16 against the fp32 oracle: SURVEY.md 8(d) gate rel-L2 <= 2e-2 per tap (measured ~3e-3).
    The ViT-L 512x512 case (configs[3]) is pinned to the HF golden in tests/test_gpu_lowp_pinning.py."""
    from layoutdit_amd import config as cfgs
    from layoutdit_amd.modeling import DiTEncoder
    cfg = cfgs.GEOMETRIES[geom]()
    w = synth.synth_weights(cfg, 1)
    m = DiTEncoder(cfg, compute_dtype="bf16").load_numpy(w).to(DEV).eval()
    x = synth.synth_images(B, size, size, seed=1234)
    with torch.no_grad():
        out = m(torch.from_numpy(x).to(DEV))
    ref, _ = oracle.vit_forward(cfg, w, x)
    for t, r in zip(cfg.taps, ref):
        h = out.hidden_states[t]
        assert h.dtype == torch.float32
        assert rel_l2(h.cpu().numpy(), r) < 2e-2, t


def test_linear_bf16_ragged_tail_is_peeled():
    """M = 4 x 256 + 16 takes the two-launch path (main part on 256-row tiles + a peeled 16-row tail)."""
    M, N, K = 1040, 512, 256
    x, w, b = _bf16_round(_rand(40, M, K)), _bf16_round(_rand(41, N, K, scale=0.05)), _rand(42, N, scale=0.1)
    xd, wd = torch.from_numpy(x).to(DEV).to(torch.bfloat16), torch.from_numpy(w).to(DEV).to(torch.bfloat16)
    y = ops.linear_bf16(xd, wd, torch.from_numpy(b).to(DEV)).float().cpu().numpy()
    assert rel_l2(y, oracle.linear(x, w, b)) < 3e-3
    lam, r = np.abs(_rand(43, N)) * 0.3 + 0.05, _rand(44, M, N)
    h = torch.from_numpy(r).to(DEV)
    tap = torch.zeros((M, N), device=DEV)
    ops.linear_bf16(xd, wd, torch.from_numpy(b).to(DEV), epilogue=_lib.EPI_SCALE_RESID, lam=torch.from_numpy(lam).to(DEV),
                    residual=h, out=h, out2=tap)
    ref = (r.astype(np.float64) + lam.astype(np.float64) * oracle.linear(x, w, b).astype(np.float64)).astype(np.float32)
    assert rel_l2(h.cpu().numpy(), ref) < 1e-5
    np.testing.assert_array_equal(tap.cpu().numpy(), h.cpu().numpy())


@pytest.mark.parametrize("M,N,K", [(16, 1024, 1024), (1, 128, 128), (64, 320, 4096), (37, 200, 256)])
def test_linear_bf16_tail(M, N, K):
    """<= 64 rows: gemm_bf16_tail (one wave per 32 x 32 tile, all of K in the tile kernels' order)."""
    x, w, b = _bf16_round(_rand(50, M, K)), _bf16_round(_rand(51, N, K, scale=0.05)), _rand(52, N, scale=0.1)
    xd, wd, bd = (torch.from_numpy(x).to(DEV).to(torch.bfloat16), torch.from_numpy(w).to(DEV).to(torch.bfloat16),
                  torch.from_numpy(b).to(DEV))
    y = ops.linear_bf16(xd, wd, bd, epilogue=_lib.EPI_BIAS_GELU).float().cpu().numpy()
    assert rel_l2(y, oracle.gelu(oracle.linear(x, w, b))) < 3e-3
    lam, r = np.abs(_rand(53, N)) * 0.3 + 0.05, _rand(54, M, N)
    h = torch.from_numpy(r).to(DEV)
    ops.linear_bf16(xd, wd, bd, epilogue=_lib.EPI_SCALE_RESID, lam=torch.from_numpy(lam).to(DEV), residual=h, out=h)
    ref = (r.astype(np.float64) + lam.astype(np.float64) * oracle.linear(x, w, b).astype(np.float64)).astype(np.float32)
    assert rel_l2(h.cpu().numpy(), ref) < 1e-5


@pytest.mark.parametrize("M,N,K,rows", [(1040, 1024, 1024, 16), (600, 320, 4096, 64), (528, 200, 192, 37)])
@pytest.mark.parametrize("tile", ["auto", "2", "3", "4", "5"])
def test_linear_bf16_tail_rows_have_the_bits_of_tile_rows(M, N, K, rows, tile):
    """Batch invariance at the kernel level: the last `rows` rows computed ALONE (tail kernel) equal, bit for bit, the same
    rows inside the big problem - whichever tile height serves them there (128 / 256 / 192 / 320 rows, or the peeled tail) and
    for every epilogue.  A DP shard boundary or a permutation moves token rows between exactly these two situations."""
    x, w, b = _bf16_round(_rand(60, M, K)), _bf16_round(_rand(61, N, K, scale=0.05)), _rand(62, N, scale=0.1)
    xd, wd, bd = (torch.from_numpy(x).to(DEV).to(torch.bfloat16), torch.from_numpy(w).to(DEV).to(torch.bfloat16),
                  torch.from_numpy(b).to(DEV))
    lam, r = torch.from_numpy(np.abs(_rand(63, N)) * 0.3 + 0.05).to(DEV), torch.from_numpy(_rand(64, M, N)).to(DEV)
    xt = xd[M - rows:].contiguous()
    for epi in (_lib.EPI_BIAS, _lib.EPI_BIAS_GELU, _lib.EPI_SCALE_RESID):
        kw = dict(lam=lam, residual=r) if epi == _lib.EPI_SCALE_RESID else {}
        kt = dict(lam=lam, residual=r[M - rows:].contiguous()) if epi == _lib.EPI_SCALE_RESID else {}
        if tile != "auto":
            _lib.set_switch("LDIT_GEMM_BF16_TILE", tile)
        big = ops.linear_bf16(xd, wd, bd, epilogue=epi, **kw)
        _lib.set_switch("LDIT_GEMM_BF16_TILE", None)
        small = ops.linear_bf16(xt, wd, bd, epilogue=epi, **kt)
        assert torch.equal(big[M - rows:], small), (epi, tile)


def test_peeled_tail_rides_in_the_main_launch_with_the_bits_of_its_own_launch():
    """M = 64 x 256 + 16 rows (ViT-L/16 512 x 512 at bs=16) and N = 1024: the 16 rows past the last full tile row would cost a second
    round of the machine, so they are peeled - since round 4 as extra workgroups of the SAME launch (one 32 x 16 tile each), before
    that as a launch of their own (LDIT_GEMM_BF16_TAIL_LAUNCH=1 keeps it).  Both give the rows the bits they get when they are
    computed alone, for every epilogue, and leave the main rows untouched."""
    M, N, K, rows = 64 * 256 + 16, 1024, 192, 16
    x, w, b = _bf16_round(_rand(70, M, K)), _bf16_round(_rand(71, N, K, scale=0.05)), _rand(72, N, scale=0.1)
    xd, wd, bd = (torch.from_numpy(x).to(DEV).to(torch.bfloat16), torch.from_numpy(w).to(DEV).to(torch.bfloat16),
                  torch.from_numpy(b).to(DEV))
    lam, r = torch.from_numpy(np.abs(_rand(73, N)) * 0.3 + 0.05).to(DEV), torch.from_numpy(_rand(74, M, N)).to(DEV)
    xt = xd[M - rows:].contiguous()
    try:
        for epi in (_lib.EPI_BIAS, _lib.EPI_BIAS_GELU, _lib.EPI_SCALE_RESID):
            kw = dict(lam=lam, residual=r) if epi == _lib.EPI_SCALE_RESID else {}
            kt = dict(lam=lam, residual=r[M - rows:].contiguous()) if epi == _lib.EPI_SCALE_RESID else {}
            _lib.set_switch("LDIT_GEMM_BF16_TAIL_LAUNCH", None)
            riding = ops.linear_bf16(xd, wd, bd, epilogue=epi, **kw)
            small = ops.linear_bf16(xt, wd, bd, epilogue=epi, **kt)
            _lib.set_switch("LDIT_GEMM_BF16_TAIL_LAUNCH", "1")
            apart = ops.linear_bf16(xd, wd, bd, epilogue=epi, **kw)
            assert torch.equal(riding, apart), epi
            assert torch.equal(riding[M - rows:], small), epi
            ref = torch.from_numpy(x[:64]).double() @ torch.from_numpy(w).double().T + torch.from_numpy(b).double()
            if epi == _lib.EPI_BIAS:
                assert rel_l2(riding[:64].float().cpu().numpy(), ref.numpy()) < 1e-2
    finally:
        _lib.set_switch("LDIT_GEMM_BF16_TAIL_LAUNCH", None)


def test_forward_bf16_is_batch_invariant():
    """An image's taps do not depend on the batch it rides in: every dot product has a fixed k-order and the epilogues use
    the same explicit fmas on the slab path (interior tiles) and on the direct path (ragged tiles), so bs=64 (M = 12608,
    256 x 256 tiles) and bs=2 (M = 394, 128 x 128 tiles, ragged) agree bit for bit."""
    from layoutdit_amd import config as cfgs
    from layoutdit_amd.modeling import DiTEncoder
    cfg = cfgs.vit_base()
    m = DiTEncoder(cfg, compute_dtype="bf16").load_numpy(synth.synth_weights(cfg, 0)).to(DEV).eval()
    x = torch.from_numpy(synth.synth_images(64, 224, 224, seed=1234)).to(DEV)
    with torch.no_grad():
        big = m(x).hidden_states
        small = m(x[40:42]).hidden_states
    for t in cfg.taps:
        assert torch.equal(big[t][40:42], small[t]), t


def test_forward_bf16_vs_cpu_bf16_autocast_of_the_oracle():
    """besides the fp32 oracle, compare with the oracle run in bf16 on the CPU - here the torch-ops
    restatement under ``torch.autocast("cpu", bfloat16)``, i.e. what the reference's BeitModel computes under the autocast
    its trainer uses (linears and attention in bf16, LayerNorm and the LayerScale residual adds in fp32).  Two bf16
    pipelines with different rounding points: gate at 1e-2 rel-L2 per tap."""
    from layoutdit_amd import config as cfgs
    from layoutdit_amd.modeling import DiTEncoder
    from oracle.vit_oracle_torch import TorchOracle
    cfg = cfgs.vit_tiny()
    w = synth.synth_weights(cfg, 1)
    x = synth.synth_images(2, 224, 224, seed=1234)
    m = DiTEncoder(cfg, compute_dtype="bf16").load_numpy(w).to(DEV).eval()
    with torch.no_grad():
        out = m(torch.from_numpy(x).to(DEV))
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ref = TorchOracle(cfg, w).forward(x)
    for t, r in zip(cfg.taps, ref):
        assert rel_l2(out.hidden_states[t].cpu().numpy(), r.float().numpy()) < 1e-2, t
