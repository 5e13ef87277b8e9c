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
    os.environ.pop("LDIT_GEMM_BF16_TILE", None)


def test_cast_matches_torch_round_to_nearest_even():
    x = _rand(1, 1000, 37, scale=3.0)
    x.reshape(-1)[:4] = [1.00390625, 1.01171875, -0.0, 3.3895314e38]      # ties and the largest finite bf16 neighbourhood
    got = ops.cast_bf16(torch.from_numpy(x).to(DEV)).cpu()
    assert torch.equal(got, torch.from_numpy(x).to(torch.bfloat16))


@pytest.mark.parametrize("tile", ["auto", "0", "1", "2"])
@pytest.mark.parametrize("M,N,K", [(37, 50, 64), (394, 576, 192), (1025, 1024, 1024), (513, 3072, 768), (300, 768, 4096)])
def test_linear_bf16_bias(M, N, K, tile):
    if tile != "auto":
        os.environ["LDIT_GEMM_BF16_TILE"] = tile
    x, w, b = _bf16_round(_rand(1, M, K)), _bf16_round(_rand(2, N, K, scale=0.05)), _rand(3, N, scale=0.1)
    xd, wd = torch.from_numpy(x).to(DEV).to(torch.bfloat16), torch.from_numpy(w).to(DEV).to(torch.bfloat16)
    y = ops.linear_bf16(xd, wd, torch.from_numpy(b).to(DEV)).float().cpu().numpy()
    ref = oracle.linear(x, w, b)
    assert rel_l2(y, ref) < 3e-3
    assert max_rel(y, ref) < 1.6e-2        # <= one bf16 ulp (2^-8) of max(|ref|, 1), plus accumulation noise


@pytest.mark.parametrize("tile", ["0", "1", "2"])
def test_linear_bf16_gelu_and_residual(tile):
    os.environ["LDIT_GEMM_BF16_TILE"] = tile
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
