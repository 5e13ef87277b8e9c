"""Seeded shape fuzzing of the GEMM and attention kernels on the GPU (through the C ABI): random ragged M / N around
every tile boundary, every tiling forced in turn, every epilogue; parity against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from layoutdit_amd import _lib, ops, synth          # noqa: E402
from oracle import oracle                           # noqa: E402
from tests.util import rel_l2                       # noqa: E402

DEV = "cuda:0"


def _rand(seed, *shape, scale=1.0):
    return (scale * synth.normal(seed, 3, int(np.prod(shape)))).astype(np.float32).reshape(shape)


def _cases(n, seed):
    u = synth.uniform01(seed, 77, 4 * n).reshape(n, 4)
    edges_m = [1, 15, 16, 17, 31, 33, 63, 64, 65, 127, 129, 303, 304, 305, 319, 321, 607, 609, 1000]
    edges_n = [4, 8, 31, 32, 33, 64, 96, 127, 128, 129, 200, 256, 257, 384, 768]
    out = []
    for a, b, c, d in u:
        M = edges_m[int(a * len(edges_m))] if d < 0.7 else 1 + int(a * 900)
        N = edges_n[int(b * len(edges_n))] if d < 0.7 else 4 * (1 + int(b * 150))
        K = 32 * (1 + int(c * 12))
        out.append((M, N, K))
    return out


@pytest.fixture(autouse=True)
def _cleanup():
    _lib.load()
    yield
    _lib.set_switch("LDIT_GEMM_TILE", None)
    _lib.set_switch("LDIT_GEMM_BF16_TILE", None)
    _lib.set_switch("LDIT_GEMM_FP8_TILE", None)


@pytest.mark.parametrize("tile", ["auto", "0", "1", "2", "3", "4", "5", "6", "7"])
def test_fuzz_linear_f32(tile):
    if tile != "auto":
        _lib.set_switch("LDIT_GEMM_TILE", tile)
    for idx, (M, N, K) in enumerate(_cases(14, 100 + len(tile))):
        epi = idx % 3
        x, w, b = _rand(idx, M, K), _rand(idx + 50, N, K, scale=0.1), _rand(idx + 90, N, scale=0.2)
        lam, r = np.abs(_rand(idx + 7, N)) + 0.05, _rand(idx + 8, M, N)
        kw = {}
        if epi == _lib.EPI_SCALE_RESID:
            kw = dict(lam=torch.from_numpy(lam).to(DEV), residual=torch.from_numpy(r).to(DEV))
        y = ops.linear(torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV), torch.from_numpy(b).to(DEV), epilogue=epi,
                       **kw).cpu().numpy()
        ref = oracle.linear(x, w, b)
        if epi == _lib.EPI_BIAS_GELU:
            ref = oracle.gelu(ref)
        if epi == _lib.EPI_SCALE_RESID:
            ref = (r.astype(np.float64) + lam.astype(np.float64) * ref.astype(np.float64)).astype(np.float32)
        assert rel_l2(y, ref) < 1e-5, (tile, M, N, K, epi)
        assert np.isfinite(y).all()


@pytest.mark.parametrize("tile", ["auto", "1", "2", "3", "4", "5"])
def test_fuzz_linear_bf16(tile):
    if tile != "auto":
        _lib.set_switch("LDIT_GEMM_BF16_TILE", tile)
    for idx, (M, N, K) in enumerate(_cases(10, 200 + len(tile))):
        K = 64 * ((K + 63) // 64)
        epi = idx % 3
        x = torch.from_numpy(_rand(idx, M, K)).to(torch.bfloat16)
        w = torch.from_numpy(_rand(idx + 50, N, K, scale=0.1)).to(torch.bfloat16)
        b, lam, r = _rand(idx + 90, N, scale=0.2), np.abs(_rand(idx + 7, N)) + 0.05, _rand(idx + 8, M, N)
        kw = {}
        if epi == _lib.EPI_SCALE_RESID:
            kw = dict(lam=torch.from_numpy(lam).to(DEV), residual=torch.from_numpy(r).to(DEV))
        y = ops.linear_bf16(x.to(DEV), w.to(DEV), torch.from_numpy(b).to(DEV), epilogue=epi, **kw).float().cpu().numpy()
        ref = oracle.linear(x.float().numpy(), w.float().numpy(), b)
        if epi == _lib.EPI_BIAS_GELU:
            ref = oracle.gelu(ref)
        if epi == _lib.EPI_SCALE_RESID:
            ref = (r.astype(np.float64) + lam.astype(np.float64) * ref.astype(np.float64)).astype(np.float32)
        assert rel_l2(y, ref) < (1e-5 if epi == _lib.EPI_SCALE_RESID else 4e-3), (tile, M, N, K, epi)


@pytest.mark.parametrize("tile", ["auto", "0", "1", "2", "3", "4"])
def test_fuzz_linear_fp8(tile):
    """fp8 GEMM on the same ragged shapes; reference = oracle on the dequantised codes (gates: tests/test_gpu_fp8.py)."""
    if tile != "auto":
        _lib.set_switch("LDIT_GEMM_FP8_TILE", tile)
    F8 = torch.float8_e4m3fn
    for idx, (M, N, K) in enumerate(_cases(10, 500 + len(tile))):
        K = 128 * ((K + 127) // 128)
        epi = (0, 2)[idx % 2]                       # bias -> bf16, scale+residual -> fp32 (GELU -> fp8 has its own test)
        x, w = _rand(idx, M, K), _rand(idx + 50, N, K, scale=0.1)
        sx, sw = float(np.abs(x).max()) / 448.0, float(np.abs(w).max()) / 448.0
        xq = (torch.from_numpy(x) * torch.tensor(np.float32(1.0 / sx))).clamp(-448.0, 448.0).to(F8)
        wq = (torch.from_numpy(w) * torch.tensor(np.float32(1.0 / sw))).clamp(-448.0, 448.0).to(F8)
        b, lam, r = _rand(idx + 90, N, scale=0.2), np.abs(_rand(idx + 7, N)) + 0.05, _rand(idx + 8, M, N)
        kw = {}
        if epi == _lib.EPI_SCALE_RESID:
            kw = dict(lam=torch.from_numpy(lam).to(DEV), residual=torch.from_numpy(r).to(DEV))
        y = ops.linear_fp8(xq.to(DEV), wq.to(DEV), sx * sw, torch.from_numpy(b).to(DEV), epilogue=epi, **kw).float().cpu().numpy()
        ref = oracle.linear(xq.float().numpy(), wq.float().numpy()) * np.float32(sx * sw) + b
        if epi == _lib.EPI_SCALE_RESID:
            ref = (r.astype(np.float64) + lam.astype(np.float64) * ref.astype(np.float64)).astype(np.float32)
        assert rel_l2(y, ref) < (1e-4 if epi == _lib.EPI_SCALE_RESID else 4e-3), (tile, M, N, K, epi)
        assert np.isfinite(y).all()


def test_fuzz_attention_lengths():
    u = synth.uniform01(300, 5, 24).reshape(12, 2)
    for idx, (a, b) in enumerate(u):
        N, H, B = 1 + int(a * 700), 1 + int(b * 3), 1 + idx % 2
        q, k, v = (_rand(400 + 3 * idx + i, B, N, H * 64) for i in range(3))
        o = ops.attention(*(torch.from_numpy(t).to(DEV) for t in (q, k, v)), heads=H).cpu().numpy()
        ref = oracle.attention(q, k, v, heads=H)
        assert rel_l2(o, ref) < 1e-5, (B, N, H)
        qb, kb, vb = (torch.from_numpy(t).to(torch.bfloat16) for t in (q, k, v))
        ob = ops.attention_bf16(qb.to(DEV), kb.to(DEV), vb.to(DEV), heads=H).float().cpu().numpy()
        refb = oracle.attention(qb.float().numpy(), kb.float().numpy(), vb.float().numpy(), heads=H)
        assert rel_l2(ob, refb) < 8e-3, (B, N, H)
