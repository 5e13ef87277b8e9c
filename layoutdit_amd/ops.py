"""Torch-tensor front ends of the single-kernel C entry points (``include/ldit.h``).

PyTorch is plumbing here: it owns device memory and the current HIP stream; all arithmetic is in ``libldit_hip.so``.
Every function requires CUDA(=HIP) fp32 contiguous tensors and raises otherwise - nothing silently runs in eager.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib


def _device(*tensors) -> torch.device:
    """The one device every tensor argument lives on (None entries skipped); mixed devices are an error."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError(f"tensor arguments live on different devices ({dev} and {t.device})")
    if dev is None or dev.type != "cuda":
        raise ValueError("expected tensors on the GPU (libldit_hip has no CPU path)")
    return dev


def _launch(dev: torch.device, fn, *args) -> None:
    """Call one C entry point with the stream of the TENSORS' device (not of whatever device is current), with that
    device made current for the launch: the library enqueues on the stream it is handed and HIP launches on the
    current device, so both must be the operands' own."""
    with torch.cuda.device(dev):
        _lib.check(fn(*args, torch.cuda.current_stream(dev).cuda_stream))


def _req(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name}: expected a tensor on the GPU (libldit_hip has no CPU path)")
    if t.dtype != torch.float32:
        raise ValueError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")
    return t


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = _lib.EPI_BIAS,
           lam: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``epilogue(x @ weight.T + bias)`` on the fp32 MFMA GEMM.  ``x``: [M, K], ``weight``: [N, K]."""
    lib = _lib.load()
    x, weight = _req(x, "x"), _req(weight, "weight")
    M, K = x.shape
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError(f"weight {tuple(weight.shape)} does not match x {tuple(x.shape)}")
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=torch.float32)
    for t, n in ((bias, "bias"), (lam, "lam"), (residual, "residual"), (out, "out"), (out2, "out2")):
        if t is not None:
            _req(t, n)
    _launch(_device(x, weight, bias, lam, residual, out, out2), lib.ldit_linear_f32, _ptr(x), K, _ptr(weight), _ptr(bias), _ptr(out), N, M, N, K, epilogue, _ptr(lam),
                                   _ptr(residual), _ptr(out2))
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    lib = _lib.load()
    x, gamma, beta = _req(x, "x"), _req(gamma, "gamma"), _req(beta, "beta")
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    _launch(_device(x, gamma, beta), lib.ldit_layernorm_f32, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), rows, C, eps)
    return y


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: Optional[float] = None) -> torch.Tensor:
    """``q, k, v``: [B, N, H*D] token-major (may be column slices of one fused tensor: last-dim stride 1)."""
    lib = _lib.load()
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        if not t.is_cuda or t.dtype != torch.float32 or t.stride(-1) != 1 or t.stride(0) != t.shape[1] * t.stride(1):
            raise ValueError(f"{n}: expected GPU float32 [B,N,H*D] with unit last stride and dense batch stride")
    B, N, HD = q.shape
    D = HD // heads
    o = torch.empty((B, N, HD), device=q.device, dtype=torch.float32)
    _launch(_device(q, k, v), lib.ldit_attention_f32, _ptr(q), _ptr(k), _ptr(v), _ptr(o), B, N, heads, D, q.stride(1), k.stride(1),
                                      v.stride(1), HD, float(D ** -0.5 if scale is None else scale))
    return o


def embed(x: torch.Tensor, patch_w: torch.Tensor, patch_b: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor,
          patch: int) -> torch.Tensor:
    lib = _lib.load()
    x, patch_w, patch_b = _req(x, "x"), _req(patch_w, "patch_w"), _req(patch_b, "patch_b")
    cls, pos = _req(cls, "cls"), _req(pos, "pos")
    B, in_ch, H, W = x.shape
    Cc = patch_w.shape[0]
    T = (H // patch) * (W // patch) + 1
    out = torch.empty((B, T, Cc), device=x.device, dtype=torch.float32)
    _launch(_device(x, patch_w, patch_b, cls, pos), lib.ldit_embed_f32, _ptr(x), _ptr(patch_w), _ptr(patch_b), _ptr(cls), _ptr(pos), _ptr(out), B, in_ch, H, W,
                                  patch, Cc)
    return out


def embed_bf16(x: torch.Tensor, patch_w_bf16: torch.Tensor, patch_b: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor,
               patch: int) -> torch.Tensor:
    """The patch embedding of the bf16 / fp8 builds and of the train step: bf16 im2col of the batch + bf16 MFMA GEMM, fp32 out."""
    lib = _lib.load()
    x, patch_b, cls, pos = _req(x, "x"), _req(patch_b, "patch_b"), _req(cls, "cls"), _req(pos, "pos")
    if patch_w_bf16.dtype != torch.bfloat16 or not patch_w_bf16.is_contiguous():
        raise ValueError("patch_w_bf16 must be a contiguous bfloat16 tensor")
    B, in_ch, H, W = x.shape
    Cc = patch_w_bf16.shape[0]
    P = (H // patch) * (W // patch)
    out = torch.empty((B, P + 1, Cc), device=x.device, dtype=torch.float32)
    scratch = torch.empty((B * P, in_ch * patch * patch), device=x.device, dtype=torch.bfloat16)
    _launch(_device(x, patch_w_bf16, patch_b, cls, pos), lib.ldit_embed_bf16, _ptr(x), _ptr(patch_w_bf16), _ptr(patch_b), _ptr(cls),
            _ptr(pos), _ptr(out), _ptr(scratch), B, in_ch, H, W, patch, Cc)
    return out


def tap_to_map(tap: torch.Tensor, gh: int, gw: int, scale: float) -> torch.Tensor:
    lib = _lib.load()
    tap = _req(tap, "tap")
    B, T, Cc = tap.shape
    if T != gh * gw + 1:
        raise ValueError(f"tap has {T} tokens, grid {gh}x{gw} needs {gh * gw + 1}")
    out = torch.empty((B, Cc, int(gh * scale), int(gw * scale)), device=tap.device, dtype=torch.float32)
    _launch(_device(tap), lib.ldit_tap_to_map_f32, _ptr(tap), _ptr(out), B, gh, gw, Cc, float(scale))
    return out


def _req16(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float16 or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous float16 tensor on the GPU")
    return t


def widen_f16(x: torch.Tensor) -> torch.Tensor:
    """fp16 -> fp32 on the library's conversion kernel (fp16 pixel batches, ref trainer.py:153-155)."""
    lib = _lib.load()
    x = _req16(x, "x")
    out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    _launch(_device(x), lib.ldit_cast_f16_f32, _ptr(x), _ptr(out), x.numel())
    return out


def narrow_f16(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> fp16 (round to nearest even) on the library's conversion kernel."""
    lib = _lib.load()
    x = _req(x, "x")
    out = torch.empty(x.shape, device=x.device, dtype=torch.float16)
    _launch(_device(x), lib.ldit_cast_f32_f16, _ptr(x), _ptr(out), x.numel())
    return out


class _TapToMapFn(torch.autograd.Function):
    """tap_to_map with its adjoint (ldit_tap_to_map_bwd_f32), so DiTBackbone.forward stays differentiable in train mode."""

    @staticmethod
    def forward(ctx, tap, gh, gw, scale):
        ctx.geom = (gh, gw, float(scale))
        return tap_to_map(tap.detach(), gh, gw, scale)

    @staticmethod
    def backward(ctx, dmap):
        gh, gw, scale = ctx.geom
        lib = _lib.load()
        dmap = _req(dmap.contiguous(), "dmap")
        B, Cc = dmap.shape[0], dmap.shape[1]
        dtap = torch.empty((B, gh * gw + 1, Cc), device=dmap.device, dtype=torch.float32)
        _launch(_device(dmap), lib.ldit_tap_to_map_bwd_f32, _ptr(dmap), _ptr(dtap), B, gh, gw, Cc, scale)
        return dtap, None, None, None


def tap_to_map_autograd(tap: torch.Tensor, gh: int, gw: int, scale: float) -> torch.Tensor:
    return _TapToMapFn.apply(tap, gh, gw, scale)


def preprocess(images: Sequence[torch.Tensor], size=224, mean: float = 0.5, std: float = 0.5) -> torch.Tensor:
    """The detector's input transform in one kernel (ref src/layoutdit/modeling/model.py:50-54: fixed_size 224,
    mean = std = 0.5): list of ``[3, h, w]`` images in [0, 1] (all fp32 or all fp16) -> normalised, bilinearly resized
    fp32 ``[B, 3, size, size]``."""
    lib = _lib.load()
    imgs, ptrs, hs, ws, half, ch = image_list_args(images)
    B = len(imgs)
    out_h, out_w = (size, size) if isinstance(size, int) else (int(size[0]), int(size[1]))
    out = torch.empty((B, ch, out_h, out_w), device=imgs[0].device, dtype=torch.float32)
    if half:
        _launch(_device(*imgs), lib.ldit_preprocess_f16, ptrs, hs, ws, B, ch, mean, std, out_h, out_w, _ptr(out))
    else:
        _launch(_device(*imgs), lib.ldit_preprocess_f32, ptrs, hs, ws, B, ch, mean, std, out_h, out_w, _ptr(out))
    return out


def image_list_args(images: Sequence[torch.Tensor]):
    """``(checked tensors, device pointers, heights, widths, half_in, channels)`` of a ragged list of ``[C, h, w]`` images (all fp32
    or all fp16, same C, one GPU) - the HOST arrays the image-list entries of the C ABI take."""
    if len(images) == 0:
        raise ValueError("empty image list")
    half = images[0].dtype == torch.float16
    imgs = [(_req16 if half else _req)(t, f"images[{i}]") for i, t in enumerate(images)]
    ch = imgs[0].shape[0]
    if any(t.dim() != 3 or t.shape[0] != ch for t in imgs):
        raise ValueError("every image must be [C, h, w] with the same C")
    B = len(imgs)
    ptrs = (C.c_void_p * B)(*[t.data_ptr() for t in imgs])
    hs = (C.c_int32 * B)(*[t.shape[1] for t in imgs])
    ws = (C.c_int32 * B)(*[t.shape[2] for t in imgs])
    return imgs, ptrs, hs, ws, half, ch


def embed_bf16_images(images: Sequence[torch.Tensor], patch_w_bf16: torch.Tensor, patch_b: torch.Tensor, cls: torch.Tensor,
                      pos: torch.Tensor, patch: int, size=224, mean: float = 0.5, std: float = 0.5) -> torch.Tensor:
    """``embed_bf16(preprocess(images, size, mean, std), ...)`` with the input transform evaluated inside the im2col pass (SURVEY.md
    8(f)-2; ref model.py:50-54): same bits, no fp32 batch in between."""
    lib = _lib.load()
    imgs, ptrs, hs, ws, half, ch = image_list_args(images)
    patch_b, cls, pos = _req(patch_b, "patch_b"), _req(cls, "cls"), _req(pos, "pos")
    if patch_w_bf16.dtype != torch.bfloat16 or not patch_w_bf16.is_contiguous():
        raise ValueError("patch_w_bf16 must be a contiguous bfloat16 tensor")
    out_h, out_w = (size, size) if isinstance(size, int) else (int(size[0]), int(size[1]))
    B, Cc = len(imgs), patch_w_bf16.shape[0]
    P = (out_h // patch) * (out_w // patch)
    dev = imgs[0].device
    out = torch.empty((B, P + 1, Cc), device=dev, dtype=torch.float32)
    scratch = torch.empty((B * P, ch * patch * patch), device=dev, dtype=torch.bfloat16)
    _launch(_device(*imgs, patch_w_bf16, patch_b, cls, pos), lib.ldit_embed_bf16_images, ptrs, hs, ws, int(half), mean, std,
            _ptr(patch_w_bf16), _ptr(patch_b), _ptr(cls), _ptr(pos), _ptr(out), _ptr(scratch), B, ch, out_h, out_w, patch, Cc)
    return out


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> bf16 (round to nearest even) on the library's conversion kernel."""
    lib = _lib.load()
    x = _req(x, "x")
    out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    _launch(_device(x), lib.ldit_cast_f32_bf16, _ptr(x), _ptr(out), x.numel())
    return out


def linear_bf16(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = _lib.EPI_BIAS,
                lam: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 MFMA GEMM: ``x`` [M, K] bf16, ``weight`` [N, K] bf16, fp32 accumulation; bf16 result for the bias / GELU
    epilogues, fp32 (in place on ``residual`` if ``out is residual``) for the scale+residual epilogue."""
    lib = _lib.load()
    for t, n in ((x, "x"), (weight, "weight")):
        if not t.is_cuda or t.dtype != torch.bfloat16 or not t.is_contiguous():
            raise ValueError(f"{n}: expected a contiguous bfloat16 GPU tensor")
    M, K = x.shape
    N = weight.shape[0]
    if out is None:
        out = torch.empty((M, N), device=x.device,
                          dtype=torch.float32 if epilogue == _lib.EPI_SCALE_RESID else torch.bfloat16)
    for t, n in ((bias, "bias"), (lam, "lam"), (residual, "residual"), (out2, "out2")):
        if t is not None:
            _req(t, n)
    _launch(_device(x, weight, bias, lam, residual, out, out2), lib.ldit_linear_bf16, _ptr(x), K, _ptr(weight), _ptr(bias), _ptr(out), N, M, N, K, epilogue, _ptr(lam),
                                    _ptr(residual), _ptr(out2))
    return out


def split_planes(x: torch.Tensor, planes: int) -> torch.Tensor:
    """fp32 [rows, cols] -> bf16 [rows, planes * cols]: p0 = bf16(x), p1 = bf16(x - p0), (p2 = bf16(x - p0 - p1)) side by side."""
    lib = _lib.load()
    x = _req(x, "x")
    rows, cols = x.shape
    out = torch.empty((rows, planes * cols), device=x.device, dtype=torch.bfloat16)
    _launch(_device(x), lib.ldit_split_f32_planes, _ptr(x), cols, _ptr(out), rows, cols, planes)
    return out


def attention_planes(qkv_planes: torch.Tensor, heads: int, planes: int) -> torch.Tensor:
    """The attention of the split-fp32 builds: ``qkv_planes`` bf16 [B, N, planes * 3C] = the planes of the fused q|k|v rows (q
    pre-multiplied by scale * log2 e); returns the planes of the fp32 result, bf16 [B, N, planes * C]."""
    lib = _lib.load()
    if not qkv_planes.is_cuda or qkv_planes.dtype != torch.bfloat16 or not qkv_planes.is_contiguous():
        raise ValueError("qkv_planes: expected a contiguous bfloat16 GPU tensor")
    B, N, W6 = qkv_planes.shape
    Cc = W6 // (3 * planes)
    out = torch.empty((B, N, planes * Cc), device=qkv_planes.device, dtype=torch.bfloat16)
    base = qkv_planes.data_ptr()
    _launch(_device(qkv_planes), lib.ldit_attention_planes, base, base + 2 * Cc, base + 4 * Cc, _ptr(out), B, N, heads, Cc // heads,
            W6, 3 * Cc, planes * Cc, planes)
    return out


def layernorm_planes(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, planes: int) -> torch.Tensor:
    lib = _lib.load()
    x, gamma, beta = _req(x, "x"), _req(gamma, "gamma"), _req(beta, "beta")
    rows, Cc = x.shape
    out = torch.empty((rows, planes * Cc), device=x.device, dtype=torch.bfloat16)
    _launch(_device(x, gamma, beta), lib.ldit_layernorm_f32_planes, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(out), rows, Cc, float(eps), planes)
    return out


def linear_planes(xp: torch.Tensor, wp: torch.Tensor, planes: int, bias: Optional[torch.Tensor] = None, epilogue: int = _lib.EPI_BIAS,
                  lam: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The GEMM of the split-fp32 builds: ``xp`` [M, planes*K], ``wp`` [N, planes*K] bf16 planes of fp32 operands (``split_planes``);
    3 (planes = 2) or 6 (planes = 3) plane products on the bf16 MFMA, fp32 accumulation.  fp32 result (bias / scale+residual),
    or the planes of erf-GELU(.) as bf16 [M, planes*N] (bias+GELU)."""
    lib = _lib.load()
    for t, n in ((xp, "xp"), (wp, "wp")):
        if not t.is_cuda or t.dtype != torch.bfloat16 or not t.is_contiguous():
            raise ValueError(f"{n}: expected a contiguous bfloat16 GPU tensor")
    M, K = xp.shape[0], xp.shape[1] // planes
    N = wp.shape[0]
    if out is None:
        out = (torch.empty((M, planes * N), device=xp.device, dtype=torch.bfloat16) if epilogue == _lib.EPI_BIAS_GELU
               else torch.empty((M, N), device=xp.device, dtype=torch.float32))
    for t, n in ((bias, "bias"), (lam, "lam"), (residual, "residual"), (out2, "out2")):
        if t is not None:
            _req(t, n)
    _launch(_device(xp, wp, bias, lam, residual, out, out2), lib.ldit_linear_planes, _ptr(xp), planes * K, _ptr(wp), _ptr(bias), _ptr(out),
            out.shape[1], M, N, K, epilogue, _ptr(lam), _ptr(residual), _ptr(out2), planes)
    return out


FP8_MAX = 448.0          # largest finite e4m3 magnitude


def amax(x: torch.Tensor) -> torch.Tensor:
    """max |x| as a one-element fp32 GPU tensor (no host sync)."""
    lib = _lib.load()
    x = _req(x, "x")
    out = torch.empty(1, device=x.device, dtype=torch.float32)
    _launch(_device(x), lib.ldit_amax_f32, _ptr(x), x.numel(), _ptr(out))
    return out


def quant_fp8(x: torch.Tensor, scale: float) -> torch.Tensor:
    """fp32 -> fp8 e4m3 codes of ``x / scale`` (round to nearest even, saturating at +-448)."""
    lib = _lib.load()
    x = _req(x, "x")
    out = torch.empty(x.shape, device=x.device, dtype=torch.float8_e4m3fn)
    _launch(_device(x), lib.ldit_quant_f32_fp8, _ptr(x), _ptr(out), x.numel(), 1.0 / float(scale))
    return out


def quant_rows_fp8(w: torch.Tensor):
    """Per-output-channel weight quantisation: ``(codes [N, K] fp8 e4m3, scales [N] fp32)`` with
    ``w[n, :] ~= scales[n] * codes[n, :]``."""
    lib = _lib.load()
    w = _req(w, "w")
    N, K = w.shape
    codes = torch.empty((N, K), device=w.device, dtype=torch.float8_e4m3fn)
    scales = torch.empty(N, device=w.device, dtype=torch.float32)
    _launch(_device(w), lib.ldit_quant_rows_f32_fp8, _ptr(w), _ptr(codes), _ptr(scales), N, K)
    return codes, scales


def linear_fp8(x: torch.Tensor, weight: torch.Tensor, ab_scale: float, bias: Optional[torch.Tensor] = None,
               epilogue: int = _lib.EPI_BIAS, lam: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None, out_scale: float = 1.0,
               w_scales: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp8 MFMA GEMM: ``x`` [M, K], ``weight`` [N, K] fp8 e4m3 codes; the accumulator is multiplied by ``ab_scale`` (=
    scale_x * scale_w per tensor) or by ``ab_scale * w_scales[n]`` (``ab_scale`` = scale_x, ``w_scales`` per output
    channel); bf16 result for the bias epilogue, fp8 codes of ``gelu(.) / out_scale`` for the GELU epilogue, fp32 for the
    scale+residual epilogue."""
    lib = _lib.load()
    for t, n in ((x, "x"), (weight, "weight")):
        if not t.is_cuda or t.dtype != torch.float8_e4m3fn or not t.is_contiguous():
            raise ValueError(f"{n}: expected a contiguous float8_e4m3fn GPU tensor")
    M, K = x.shape
    N = weight.shape[0]
    if out is None:
        dt = {_lib.EPI_BIAS: torch.bfloat16, _lib.EPI_BIAS_GELU: torch.float8_e4m3fn, _lib.EPI_SCALE_RESID: torch.float32}[epilogue]
        out = torch.empty((M, N), device=x.device, dtype=dt)
    for t, n in ((bias, "bias"), (lam, "lam"), (residual, "residual"), (out2, "out2"), (w_scales, "w_scales")):
        if t is not None:
            _req(t, n)
    _launch(_device(x, weight, bias, lam, residual, out, out2, w_scales), lib.ldit_linear_fp8, _ptr(x), K, _ptr(weight), _ptr(bias), _ptr(out), N, M, N, K, epilogue, _ptr(lam),
                                   _ptr(residual), _ptr(out2), float(ab_scale), 1.0 / float(out_scale), _ptr(w_scales))
    return out


def attention_bf16(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: Optional[float] = None,
                   prescaled: bool = False) -> torch.Tensor:
    """bf16 fused attention; ``q, k, v``: bf16 [B, N, H*D] token-major (column slices of a fused tensor are fine)."""
    lib = _lib.load()
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        if not t.is_cuda or t.dtype != torch.bfloat16 or t.stride(-1) != 1 or t.stride(0) != t.shape[1] * t.stride(1):
            raise ValueError(f"{n}: expected GPU bfloat16 [B,N,H*D] with unit last stride and dense batch stride")
    B, N, HD = q.shape
    D = HD // heads
    o = torch.empty((B, N, HD), device=q.device, dtype=torch.bfloat16)
    # prescaled: q already carries scale * log2(e) (the packed inference path); the library is told so with scale = 0
    _launch(_device(q, k, v), lib.ldit_attention_bf16, _ptr(q), _ptr(k), _ptr(v), _ptr(o), B, N, heads, D, q.stride(1), k.stride(1),
                                       v.stride(1), HD, 0.0 if prescaled else float(D ** -0.5 if scale is None else scale))
    return o


def fpn_merge(lat: torch.Tensor, gh: int, gw: int, scale: float, top: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One level of the FPN top-down pathway, NHWC: ``bilinear_scale(lat tokens) + nearest(top)``.
    ``lat``: [B, 1 + gh*gw, Ch] (lateral 1x1 convolution of a tap's tokens); ``top``: [B, th, tw, Ch] NHWC or None."""
    lib = _lib.load()
    lat = _req(lat, "lat")
    B, T, Ch = lat.shape
    if T != gh * gw + 1:
        raise ValueError(f"lat has {T} tokens, grid {gh}x{gw} needs {gh * gw + 1}")
    if top is not None:
        top = _req(top, "top")
    out = torch.empty((B, int(gh * scale), int(gw * scale), Ch), device=lat.device, dtype=torch.float32)
    th, tw = (0, 0) if top is None else (top.shape[1], top.shape[2])
    _launch(_device(lat, top), lib.ldit_fpn_merge_f32, _ptr(lat), _ptr(top), _ptr(out), B, gh, gw, Ch, float(scale), th, tw)
    return out


_ZEROS = {}


def _zero_page(device: torch.device) -> torch.Tensor:
    z = _ZEROS.get(device)
    if z is None:
        z = torch.zeros(64, dtype=torch.float32, device=device)
        _ZEROS[device] = z
    return z


def conv3x3_nhwc(x: torch.Tensor, weight_ohwi: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """3x3 / padding 1 convolution of an NHWC map as an implicit-im2col fp32 MFMA GEMM.  ``x``: [B, H, W, Cin];
    ``weight_ohwi``: [Cout, 3, 3, Cin] (= ``conv.weight.permute(0, 2, 3, 1)``); returns NHWC [B, H, W, Cout]."""
    lib = _lib.load()
    x, weight_ohwi = _req(x, "x"), _req(weight_ohwi, "weight_ohwi")
    B, H, W, Cin = x.shape
    Cout = weight_ohwi.shape[0]
    if tuple(weight_ohwi.shape) != (Cout, 3, 3, Cin):
        raise ValueError(f"weight {tuple(weight_ohwi.shape)} is not [Cout, 3, 3, {Cin}]")
    if bias is not None:
        _req(bias, "bias")
    y = torch.empty((B, H, W, Cout), device=x.device, dtype=torch.float32)
    _launch(_device(x, weight_ohwi, bias), lib.ldit_conv3x3_nhwc_f32, _ptr(x), _ptr(weight_ohwi), _ptr(bias), _ptr(y), B, H, W, Cin,
            Cout, _ptr(_zero_page(x.device)))
    return y


# ---- FPN backward building blocks (include/ldit.h "FPN backward") ------------------------------------------------------------
def fpn_merge_bwd(d_inner: torch.Tensor, gh: int, gw: int, scale: float, d_top: Optional[torch.Tensor] = None,
                  want_lat: bool = True) -> Optional[torch.Tensor]:
    """Adjoint of :func:`fpn_merge`.  ``d_inner``: [B, gh*s, gw*s, Ch] NHWC.  Returns ``d_lat`` [B, 1+gh*gw, Ch] (CLS row 0)
    and ACCUMULATES the nearest-upsample adjoint into ``d_top`` [B, th, tw, Ch] (in place) when given."""
    lib = _lib.load()
    d_inner = _req(d_inner, "d_inner")
    B, _, _, Ch = d_inner.shape
    if d_top is not None:
        _req(d_top, "d_top")
    d_lat = torch.empty((B, gh * gw + 1, Ch), device=d_inner.device, dtype=torch.float32) if want_lat else None
    th, tw = (0, 0) if d_top is None else (d_top.shape[1], d_top.shape[2])
    _launch(_device(d_inner, d_top), lib.ldit_fpn_merge_bwd_f32, _ptr(d_inner), _ptr(d_lat), _ptr(d_top), B, gh, gw, Ch, float(scale),
            th, tw)
    return d_lat


def pad_nhwc_bf16(x: torch.Tensor, slack_rows: int = 0) -> torch.Tensor:
    """bf16 zero-padded copy of an fp32 NHWC map as a 2-D operand: returns ``[slack + B*(H+2)*(W+2) + slack, C]`` bf16 whose
    middle rows are the padded pixels in (b, y, x) order and whose ``slack_rows`` leading / trailing rows are zero (room for
    the constant row offsets of the 3x3 taps, see ``layoutdit_amd/csrc/fpn_bwd.hip``)."""
    lib = _lib.load()
    x = _req(x, "x")
    B, H, W, Cc = x.shape
    rows = B * (H + 2) * (W + 2)
    buf = torch.zeros((rows + 2 * slack_rows, Cc), device=x.device, dtype=torch.bfloat16)
    _launch(_device(x), lib.ldit_pad_nhwc_f32_bf16, _ptr(x), buf[slack_rows:].data_ptr(), B, H, W, Cc)
    return buf


def colsum(x: torch.Tensor) -> torch.Tensor:
    """``x.sum(0)`` of a contiguous fp32 [M, N] matrix (two stages, fixed order)."""
    lib = _lib.load()
    x = _req(x, "x")
    M, N = x.shape
    out = torch.empty(N, device=x.device, dtype=torch.float32)
    need = ((M + 511) // 512) * N * 4                  # = ldit_colsum_scratch_bytes(M, N): one partial row per 512 rows
    scratch = torch.empty(max(need, 16), device=x.device, dtype=torch.uint8)
    _launch(_device(x), lib.ldit_colsum_f32, _ptr(x), M, N, N, _ptr(out), _ptr(scratch), need)
    return out


def colamax(x: torch.Tensor) -> torch.Tensor:
    """``x.abs().amax(0)`` of a contiguous fp32 [M, N] matrix (per-channel activation ranges, fp8 calibration)."""
    lib = _lib.load()
    x = _req(x, "x")
    M, N = x.shape
    out = torch.empty(N, device=x.device, dtype=torch.float32)
    need = ((M + 511) // 512) * N * 4
    scratch = torch.empty(max(need, 16), device=x.device, dtype=torch.uint8)
    _launch(_device(x), lib.ldit_colamax_f32, _ptr(x), M, N, N, _ptr(out), _ptr(scratch), need)
    return out


def _splits_for(K: int, M: int, N: int) -> int:
    """K-splits of a wgrad GEMM with an [M, N] output - the rule of csrc/api_train.hip pick_splits: enough 128 x 128 tiles for the
    machine (~512 / output tiles, at most 8), every split non-empty."""
    nk = (K + 63) // 64
    t128 = ((M + 127) // 128) * ((N + 127) // 128)
    s = max(1, min(nk, 8, 512 // max(t128, 1)))
    while s > 1 and ((nk + s - 1) // s) * (s - 1) >= nk:
        s -= 1
    return s


def wgrad_bf16(a: torch.Tensor, w: torch.Tensor, K: int, w_row_offset: int = 0) -> torch.Tensor:
    """``a[:K].T @ w[w_row_offset : w_row_offset + K]`` with both operands reduction-major bf16 (gemm_bf16_tr.hip, wgrad form):
    ``a`` [>= K, M], ``w`` [>= w_row_offset + K, N] contiguous.  fp32 [M, N]; K is split over workgroups into slabs that are
    summed in a fixed order."""
    lib = _lib.load()
    for t, n in ((a, "a"), (w, "w")):
        if not t.is_cuda or t.dtype != torch.bfloat16 or not t.is_contiguous() or t.dim() != 2:
            raise ValueError(f"{n}: expected a contiguous 2-D bfloat16 GPU tensor")
    if a.shape[0] < K or w_row_offset < 0 or w.shape[0] < w_row_offset + K:
        raise ValueError("wgrad_bf16: K rows are not available in both operands")
    M, N = a.shape[1], w.shape[1]
    dev = _device(a, w)
    splits = _splits_for(K, M, N)
    slabs = torch.empty((splits, M, N), device=dev, dtype=torch.float32)
    _launch(_device(a, w), lib.ldit_linear_bf16_tr, a.data_ptr(), M, 1, w[w_row_offset:].data_ptr(), N, slabs.data_ptr(), N, M, N, K,
            _lib.EPI_F32, None, splits, _zero_page(dev).data_ptr())
    if splits == 1:
        return slabs[0]
    out = torch.empty((M, N), device=dev, dtype=torch.float32)
    _launch(_device(slabs), lib.ldit_reduce_slabs_f32, slabs.data_ptr(), out.data_ptr(), M * N, splits)
    return out
