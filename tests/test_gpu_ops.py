"""Per-kernel parity on the GPU: each HIP kernel, called through the C ABI (ctypes, include/ldit.h), against the CPU
oracle on the same seeded inputs and against the golden per-op vectors.

Tolerances (fp32, written per test): the north-star gate is 1e-3 relative; the kernels are exact-fp32 MFMA so the
gates here are 100x tighter (relative-L2 <= 1e-5, element-wise <= 1e-5 * max(|ref|, 1))."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from layoutdit_amd import _lib, ops, synth          # noqa: E402
from oracle import oracle                           # noqa: E402
from tests.golden.kat_inputs import attn_inputs, ln_inputs  # noqa: E402
from tests.util import max_rel, rel_l2              # noqa: E402

DEV = "cuda:0"


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def _rand(seed, *shape, scale=1.0):
    n = int(np.prod(shape))
    return (scale * synth.normal(seed, 7, n)).astype(np.float32).reshape(shape)


@pytest.fixture(autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    _lib.load()
    yield
    _lib.set_switch("LDIT_GEMM_TILE", None)


# ---------------------------------------------------------------------------------------------------------- GEMM
GEMM_SHAPES = [
    (37, 50, 96),        # ragged in M and N, smaller than any tile
    (394, 576, 192),     # ViT-Tiny qkv at bs=2
    (34, 64, 64),        # micro
    (1000, 768, 768),    # o_proj-like
    (513, 3072, 768),    # fc1-like, ragged M
    (321, 768, 3072),    # fc2-like, one row past a 320-row tile
]


@pytest.mark.parametrize("tile", ["auto", "0", "1", "2", "3", "4", "5", "6", "7"])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_linear_bias(M, N, K, tile):
    if tile != "auto":
        _lib.set_switch("LDIT_GEMM_TILE", tile)
    x, w, b = _rand(1, M, K), _rand(2, N, K, scale=0.05), _rand(3, N, scale=0.1)
    y = ops.linear(_dev(x), _dev(w), _dev(b)).cpu().numpy()
    ref = oracle.linear(x, w, b)
    assert rel_l2(y, ref) < 1e-5
    # a k-ordered fp32 fma chain carries ~1e-7 * sum|a*b| of rounding (cdna guide, FP32-input MFMA): K = 3072 of
    # O(0.05) products needs the looser element-wise gate, still 20x inside the 1e-3 north-star tolerance
    assert max_rel(y, ref) < (5e-5 if K >= 1024 else 1e-5)


def test_linear_no_bias_matches_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_ops.npz"))
    y = ops.linear(_dev(g["lin_x"]), _dev(g["lin_w"]), _dev(g["lin_b"])).cpu().numpy()
    assert rel_l2(y, g["lin_y"]) < 1e-6
    x, w = _rand(4, 130, 64), _rand(5, 70, 64)
    assert rel_l2(ops.linear(_dev(x), _dev(w)).cpu().numpy(), oracle.linear(x, w)) < 1e-5


def test_linear_identity_asymmetric():
    """A = I against an asymmetric W catches a transposed or row-permuted C/D register map."""
    K = 128
    w = (np.arange(96 * K, dtype=np.float32).reshape(96, K) % 251) - 100.0
    eye = np.eye(K, dtype=np.float32)
    y = ops.linear(_dev(eye), _dev(w)).cpu().numpy()
    np.testing.assert_array_equal(y, w.T)


@pytest.mark.parametrize("tile", ["0", "1", "2", "3", "4", "5", "6", "7"])
def test_linear_gelu_epilogue(tile):
    _lib.set_switch("LDIT_GEMM_TILE", tile)
    M, N, K = 333, 320, 96
    x, w, b = _rand(6, M, K), _rand(7, N, K, scale=0.2), _rand(8, N)
    y = ops.linear(_dev(x), _dev(w), _dev(b), epilogue=_lib.EPI_BIAS_GELU).cpu().numpy()
    ref = oracle.gelu(oracle.linear(x, w, b))
    np.testing.assert_allclose(y, ref, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("tile", ["0", "1", "2", "3", "4", "5", "6", "7"])
def test_linear_scale_residual_inplace_and_tap(tile):
    """h <- h + lam * (x W^T + b), updated IN PLACE (R aliases Y) with a second copy to the tap buffer."""
    _lib.set_switch("LDIT_GEMM_TILE", tile)
    M, N, K = 197 * 2, 192, 768
    x, w, b = _rand(9, M, K), _rand(10, N, K, scale=0.05), _rand(11, N, scale=0.1)
    lam, r = np.abs(_rand(12, N)) * 0.3 + 0.05, _rand(13, M, N)
    h = _dev(r)
    tap = torch.full((M, N), float("nan"), device=DEV)
    out = ops.linear(_dev(x), _dev(w), _dev(b), epilogue=_lib.EPI_SCALE_RESID, lam=_dev(lam), residual=h, out=h, out2=tap)
    ref = (r.astype(np.float64) + lam.astype(np.float64) * oracle.linear(x, w, b).astype(np.float64)).astype(np.float32)
    assert out.data_ptr() == h.data_ptr()
    assert rel_l2(h.cpu().numpy(), ref) < 1e-6
    np.testing.assert_array_equal(tap.cpu().numpy(), h.cpu().numpy())


@pytest.mark.parametrize("M,N,K", [(197, 768, 768), (197, 3072, 768), (197, 768, 3072), (394, 2304, 768), (37, 50, 96), (5, 40, 64)])
def test_thin_tiling_is_bit_identical_to_the_big_tilings(M, N, K):
    """The serving-size kernel (32x32 tiles of 16x16x4 MFMAs, LDIT_GEMM_TILE=4; the default while the 64x64 tiling
    would be at most 192 workgroups) keeps the k order
    of the big tilings: every epilogue's output must be BIT-equal to the 64x64 tiling's (2) and to the panel tilings' (3: 304
    rows; 5, 6, 7: the 144 / 80 / 48-row panels of the mid-size batches), which is what keeps a row's value independent of the
    batch it rides in."""
    x, w, b = _rand(21, M, K), _rand(22, N, K, scale=0.05), _rand(23, N, scale=0.1)
    lam, r = np.abs(_rand(24, N)) * 0.3 + 0.05, _rand(25, M, N)
    outs = {}
    for tile in ("4", "2", "3", "5", "6", "7"):
        _lib.set_switch("LDIT_GEMM_TILE", tile)
        y0 = ops.linear(_dev(x), _dev(w), _dev(b))
        y1 = ops.linear(_dev(x), _dev(w), _dev(b), epilogue=_lib.EPI_BIAS_GELU)
        h = _dev(r)
        ops.linear(_dev(x), _dev(w), _dev(b), epilogue=_lib.EPI_SCALE_RESID, lam=_dev(lam), residual=h, out=h)
        y3 = ops.linear(_dev(x), _dev(w))
        outs[tile] = [t.cpu().numpy() for t in (y0, y1, h, y3)]
    for other in ("2", "3", "5", "6", "7"):
        for got, want in zip(outs["4"], outs[other]):
            np.testing.assert_array_equal(got, want)
    _lib.set_switch("LDIT_GEMM_TILE", None)
    auto = ops.linear(_dev(x), _dev(w), _dev(b)).cpu().numpy()          # whichever tiling the default picks, the bits are the same
    np.testing.assert_array_equal(auto, outs["4"][0])


@pytest.mark.parametrize("M,N,K,tile", [(64 * 197, 768, 768, "3"), (20000, 384, 96, "3"), (9000, 640, 64, "6"), (30000, 256, 160, "7"), (5000, 136, 96, "5")])
def test_persistent_panel_loop_is_bit_identical_to_one_tile_per_workgroup(M, N, K, tile):
    """LDIT_GEMM_PERSIST=1 (round 4, measured equal and therefore not the default): the fp32 panel GEMM as 256 (or 512) workgroups
    that walk their tiles, the next tile's first k-tile prefetched under the epilogue behind a counted vmcnt.  More tiles than
    resident workgroups on every case here (several tiles per workgroup, ragged last row panel, ragged columns, odd and even
    k-tile counts, every panel height, every epilogue): the bits must be those of the ordinary launch."""
    x, w, b = _rand(41, M, K), _rand(42, N, K, scale=0.05), _rand(43, N, scale=0.1)
    lam, r = np.abs(_rand(44, N)) * 0.3 + 0.05, _rand(45, M, N)
    _lib.set_switch("LDIT_GEMM_TILE", tile)
    outs = {}
    for mode in (None, "1"):
        _lib.set_switch("LDIT_GEMM_PERSIST", mode)
        y0 = ops.linear(_dev(x), _dev(w), _dev(b))
        y1 = ops.linear(_dev(x), _dev(w), _dev(b), epilogue=_lib.EPI_BIAS_GELU)
        h, tap = _dev(r), torch.full((M, N), float("nan"), device=DEV)
        ops.linear(_dev(x), _dev(w), _dev(b), epilogue=_lib.EPI_SCALE_RESID, lam=_dev(lam), residual=h, out=h, out2=tap)
        outs[mode] = [t.cpu().numpy() for t in (y0, y1, h, tap)]
    _lib.set_switch("LDIT_GEMM_PERSIST", None)
    _lib.set_switch("LDIT_GEMM_TILE", None)
    for got, want in zip(outs["1"], outs[None]):
        np.testing.assert_array_equal(got, want)
    assert rel_l2(outs["1"][0], oracle.linear(x, w, b)) < 1e-6


@pytest.mark.parametrize("tile", ["auto", "0", "1", "2", "3", "4", "6"])
def test_linear_row_strides_through_the_c_abi(tile):
    """lda > K and ldy > N (operands that are column slices of wider buffers: the fused q|k|v tensor, an output written into
    a slice of a concatenated map) on every fp32 tiling; the columns outside the slice must stay untouched."""
    if tile != "auto":
        _lib.set_switch("LDIT_GEMM_TILE", tile)
    lib = _lib.load()
    M, N, K, lda, ldy = 211, 96, 64, 160, 224
    xw = _rand(31, M, lda)
    w, b = _rand(32, N, K, scale=0.1), _rand(33, N, scale=0.2)
    lam, r = np.abs(_rand(34, N)) + 0.05, _rand(35, M, ldy)
    x_t, w_t, b_t, lam_t = _dev(xw), _dev(w), _dev(b), _dev(lam)
    for epi in (_lib.EPI_BIAS, _lib.EPI_BIAS_GELU, _lib.EPI_SCALE_RESID):
        y_t = _dev(r)                                    # doubles as the residual (in place) for EPI_SCALE_RESID
        xs, ys = x_t[:, 32:32 + K], y_t[:, 64:64 + N]     # 16-byte aligned column slices
        rc = lib.ldit_linear_f32(xs.data_ptr(), lda, w_t.data_ptr(), b_t.data_ptr(), ys.data_ptr(), ldy, M, N, K, epi,
                                 lam_t.data_ptr() if epi == _lib.EPI_SCALE_RESID else None,
                                 ys.data_ptr() if epi == _lib.EPI_SCALE_RESID else None, None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.ldit_last_error().decode()
        got = y_t.cpu().numpy()
        ref = oracle.linear(xw[:, 32:32 + K], w, b)
        if epi == _lib.EPI_BIAS_GELU:
            ref = oracle.gelu(ref)
        if epi == _lib.EPI_SCALE_RESID:
            ref = (r[:, 64:64 + N].astype(np.float64) + lam.astype(np.float64) * ref.astype(np.float64)).astype(np.float32)
        assert rel_l2(got[:, 64:64 + N], ref) < 1e-5, (tile, epi)
        np.testing.assert_array_equal(got[:, :64], r[:, :64])
        np.testing.assert_array_equal(got[:, 64 + N:], r[:, 64 + N:])


def test_linear_rejects_bad_arguments():
    x, w = _dev(_rand(1, 8, 40)), _dev(_rand(2, 8, 40))
    with pytest.raises(_lib.LditError) as e:        # K not a multiple of 32
        ops.linear(x, w)
    assert e.value.code == _lib.LDIT_EUNSUPPORTED
    with pytest.raises(ValueError):
        ops.linear(torch.zeros(4, 32), torch.zeros(4, 32))   # CPU tensors: no fallback


# ------------------------------------------------------------------------------------------------------ LayerNorm
def test_layernorm_golden_rows(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_ops.npz"))
    x, gam, bet = ln_inputs()
    y = ops.layernorm(_dev(x), _dev(gam), _dev(bet), eps=1e-12).cpu().numpy()
    ref64 = oracle.layernorm(x, gam, bet, eps=1e-12)
    tight = [r for r in range(16) if r not in (1, 3)]     # rows 1 and 3 are ill-conditioned in fp32 by construction
    assert max_rel(y[tight], g["ln_y"][tight]) < 2e-5
    assert max_rel(y[tight], ref64[tight]) < 2e-5
    assert rel_l2(y[1], ref64[1]) < 5e-2
    assert np.all(np.isfinite(y))


@pytest.mark.parametrize("rows,C", [(1, 64), (5, 192), (394, 768), (1025, 1024), (3, 2048)])
def test_layernorm_shapes(rows, C):
    x = _rand(20, rows, C, scale=3.0) + 1.5
    gam, bet = 1.0 + 0.1 * _rand(21, C), 0.1 * _rand(22, C)
    y = ops.layernorm(_dev(x), _dev(gam), _dev(bet), eps=1e-12).cpu().numpy()
    assert max_rel(y, oracle.layernorm(x, gam, bet, eps=1e-12)) < 1e-5


# ------------------------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("n_tok", [197, 1025])
def test_attention_golden(golden_dir, n_tok):
    g = np.load(os.path.join(golden_dir, "g4_ops.npz"))
    q, k, v = attn_inputs(n_tok)
    o = ops.attention(_dev(q), _dev(k), _dev(v), heads=3).cpu().numpy()
    assert rel_l2(o, g[f"attn{n_tok}_o"]) < 1e-5
    assert max_rel(o, g[f"attn{n_tok}_o"]) < 1e-5
    assert rel_l2(o, oracle.attention(q, k, v, heads=3)) < 1e-5


@pytest.mark.parametrize("B,N,H", [(1, 1, 1), (2, 17, 2), (1, 32, 1), (1, 33, 3), (3, 224, 2), (2, 225, 1), (1, 449, 2)])
def test_attention_ragged_lengths(B, N, H):
    """Token counts on and around the 32-key tile and the 224-key LDS chunk boundaries."""
    D = 64
    q, k, v = _rand(30, B, N, H * D), _rand(31, B, N, H * D), _rand(32, B, N, H * D)
    o = ops.attention(_dev(q), _dev(k), _dev(v), heads=H).cpu().numpy()
    assert rel_l2(o, oracle.attention(q, k, v, heads=H)) < 1e-5


def test_attention_on_fused_qkv_views():
    """The forward feeds column slices of the fused [M, 3C] projection (row stride 3C)."""
    B, N, H, D = 2, 197, 3, 64
    qkv = _rand(33, B, N, 3 * H * D)
    t = _dev(qkv)
    C_ = H * D
    o = ops.attention(t[..., :C_], t[..., C_:2 * C_], t[..., 2 * C_:], heads=H).cpu().numpy()
    ref = oracle.attention(qkv[..., :C_], qkv[..., C_:2 * C_], qkv[..., 2 * C_:], heads=H)
    assert rel_l2(o, ref) < 1e-5


def test_attention_large_logits_force_rescale():
    """A dominant key placed in the LAST chunk makes the running max jump at the final rescale (N = 449: 3 chunks)."""
    B, N, H, D = 1, 449, 1, 64
    q, k, v = _rand(34, B, N, D), _rand(35, B, N, D), _rand(36, B, N, D)
    k[0, 440] = 6.0 * q[0, 7]
    q[0, 7] *= 3.0
    o = ops.attention(_dev(q), _dev(k), _dev(v), heads=H).cpu().numpy()
    assert max_rel(o, oracle.attention(q, k, v, heads=H)) < 1e-5


# ------------------------------------------------------------------------------------------------- embedding / maps
@pytest.mark.parametrize("B,H,W,C", [(2, 64, 64, 64), (2, 224, 224, 192), (1, 96, 160, 128)])
def test_embed(B, H, W, C):
    x = synth.synth_images(B, H, W, seed=5, kind="uniform")
    pw, pb = _rand(40, C, 3, 16, 16, scale=0.02), _rand(41, C, scale=0.02)
    T = (H // 16) * (W // 16) + 1
    cls, pos = _rand(42, C, scale=0.02), _rand(43, T, C, scale=0.02)
    out = ops.embed(_dev(x), _dev(pw), _dev(pb), _dev(cls), _dev(pos), 16).cpu().numpy()
    ref = np.empty((B, T, C), np.float32)
    oracle.lib().oracle_embed(x.ctypes.data, pw.ctypes.data, pb.ctypes.data, cls.ctypes.data, pos.ctypes.data, B, 3, H, W,
                              16, C, ref.ctypes.data)
    assert rel_l2(out, ref) < 1e-5
    assert max_rel(out, ref) < 1e-5


@pytest.mark.parametrize("B,H,W,C", [(2, 64, 64, 64), (3, 224, 224, 192), (1, 96, 160, 128), (5, 224, 224, 768), (2, 512, 512, 256)])
def test_embed_bf16_is_the_oracle_embedding_of_the_bf16_rounded_operands(B, H, W, C):
    """The patch embedding of the bf16 / fp8 builds and of the train step (ldit_embed_bf16): bf16 im2col + bf16 MFMA GEMM with the
    row-remapping epilogue.  Products of bf16 values are exact in fp32, so against the oracle (TF:81-90, TF:153-176) fed the SAME
    bf16-rounded pixels and projection only the accumulation order differs: 1e-5; against the unrounded operands: the bf16 gate.
    Geometries cover whole 256-row tiles through the slab epilogue (5 x 196, 2 x 1024 rows), ragged last tiles and sub-tile
    problems (direct epilogue), and N below one 256-column tile."""
    x = synth.synth_images(B, H, W, seed=5, kind="uniform")
    pw, pb = _rand(40, C, 3, 16, 16, scale=0.02), _rand(41, C, scale=0.02)
    T = (H // 16) * (W // 16) + 1
    cls, pos = _rand(42, C, scale=0.02), _rand(43, T, C, scale=0.02)
    pw16 = _dev(pw).to(torch.bfloat16).contiguous()
    out = ops.embed_bf16(_dev(x), pw16.reshape(C, -1), _dev(pb), _dev(cls), _dev(pos), 16).cpu().numpy()
    xr = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    pwr = pw16.to(torch.float32).cpu().numpy()
    ref = np.empty((B, T, C), np.float32)
    oracle.lib().oracle_embed(xr.ctypes.data, pwr.ctypes.data, pb.ctypes.data, cls.ctypes.data, pos.ctypes.data, B, 3, H, W, 16, C,
                              ref.ctypes.data)
    assert rel_l2(out, ref) < 1e-5 and max_rel(out, ref) < 1e-5
    full = np.empty((B, T, C), np.float32)
    oracle.lib().oracle_embed(x.ctypes.data, pw.ctypes.data, pb.ctypes.data, cls.ctypes.data, pos.ctypes.data, B, 3, H, W, 16, C,
                              full.ctypes.data)
    assert rel_l2(out, full) < 2e-2
    again = ops.embed_bf16(_dev(x), pw16.reshape(C, -1), _dev(pb), _dev(cls), _dev(pos), 16).cpu().numpy()
    assert np.array_equal(out, again)


def test_tap_to_map_golden(golden_dir):
    g5 = np.load(os.path.join(golden_dir, "g5_maps.npz"))
    g0 = np.load(os.path.join(golden_dir, "g0_micro.npz"))
    for i, (idx, scale) in enumerate(zip([1, 1, 2, 3], [4.0, 2.0, 1.0, 0.5]), start=2):
        m = ops.tap_to_map(_dev(g0["hidden"][idx]), 4, 4, scale).cpu().numpy()
        assert m.shape == g5[f"micro_p{i}"].shape
        assert max_rel(m, g5[f"micro_p{i}"]) < 1e-6


@pytest.mark.parametrize("scale", [4.0, 2.0, 1.0, 0.5])
def test_tap_to_map_vs_oracle(scale):
    tap = _rand(50, 2, 14 * 14 + 1, 192)
    m = ops.tap_to_map(_dev(tap), 14, 14, scale).cpu().numpy()
    assert max_rel(m, oracle.tap_to_map(tap, 14, 14, scale)) < 1e-6


# ------------------------------------------------------------------------------------------------ input transform
def test_preprocess_variable_sizes_vs_oracle_and_torch():
    """Detector input transform (ref src/layoutdit/modeling/model.py:50-54): normalise with mean = std = 0.5 and resize
    bilinearly to 224 x 224.  torchvision's source is absent, so parity is against F.interpolate semantics (torch, fp32)
    and the double-precision restatement in the oracle; the fp32 source-coordinate arithmetic alone moves an output by
    ~1e-5 on noisy images, hence 1e-4 vs the oracle and 2e-5 vs torch."""
    import torch.nn.functional as F
    sizes = [(300, 211), (224, 224), (97, 640), (512, 512), (33, 17)]
    imgs = [synth.uniform01(50 + i, 1, 3 * h * w).reshape(3, h, w).astype(np.float32) for i, (h, w) in enumerate(sizes)]
    out = ops.preprocess([_dev(a) for a in imgs]).cpu().numpy()
    assert out.shape == (len(sizes), 3, 224, 224)
    for i, a in enumerate(imgs):
        ref = oracle.preprocess(a)
        assert np.abs(out[i] - ref).max() < 1e-4, sizes[i]
        t = F.interpolate(((torch.from_numpy(a) - 0.5) / 0.5)[None], size=(224, 224), mode="bilinear", align_corners=False)[0]
        assert np.abs(out[i] - t.numpy()).max() < 2e-5, sizes[i]


def test_preprocess_many_images_and_bad_arguments():
    imgs = [_dev(synth.uniform01(80 + i, 1, 3 * 40 * 56).reshape(3, 40, 56).astype(np.float32)) for i in range(70)]
    out = ops.preprocess(imgs, size=64)              # more than one descriptor batch (48 per launch)
    one = ops.preprocess(imgs[69:70], size=64)
    np.testing.assert_array_equal(out[69].cpu().numpy(), one[0].cpu().numpy())
    with pytest.raises(ValueError):
        ops.preprocess([])
    with pytest.raises(ValueError):
        ops.preprocess([torch.zeros(3, 8, 8)])       # CPU tensor
