"""``DiTEncoder``: the ``nn.Module`` that stands where LayoutDiT puts HuggingFace's ``BeitModel``.

Drop-in surface (everything the reference touches on ``DiTBackbone.dit``):

* ``module(x).hidden_states[idx]`` for ``idx`` in ``[d/3, d/2, 2d/3, d]``   ref ``src/layoutdit/modeling/dit_backbone.py:47,50-52``
* ``module.config.num_hidden_layers`` / ``.hidden_size``                     ref ``dit_backbone.py:33-36``
* ``state_dict()`` / ``load_state_dict(strict=False)`` with BEiT key names   ref ``src/layoutdit/modeling/model.py:65-70,110-116``
* ``.to(device)``, ``.eval()``, ``.train()``, ``parameters()``               ref ``main.py:34``, ``trainer.py:38,65``

The arithmetic is NOT here: ``forward`` hands device pointers to ``ldit_vit_forward`` in ``libldit_hip.so``
(``include/ldit.h``) on the current HIP stream.  There is no eager fallback; a CPU tensor is an error.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from ..config import DiTConfig
from .keys import to_v4


def resample_position_table(pe: torch.Tensor, g0: int, gh: int, gw: int) -> torch.Tensor:
    """``[1, 1 + g0*g0, C]`` position table -> ``[1 + gh*gw, C]`` for another grid: BEiT's ``interpolate_pos_encoding``
    (TF:models/beit/modeling_beit.py:113-151; bicubic, align_corners=False, CLS row kept).  Plain torch ops on a
    parameter-sized tensor: host-side plumbing done once per grid, and differentiable - the training path chains its
    adjoint behind the library's gradient of the resampled table."""
    Cc = pe.shape[-1]
    patch = pe[:, 1:].reshape(1, g0, g0, Cc).permute(0, 3, 1, 2)
    patch = F.interpolate(patch, size=(gh, gw), mode="bicubic", align_corners=False)
    patch = patch.permute(0, 2, 3, 1).reshape(1, gh * gw, Cc)
    return torch.cat((pe[:, :1], patch), dim=1)[0].contiguous()


class _Affine(nn.Module):
    """weight (+ optional bias) holder; the attribute path gives the tensor its BEiT key name."""

    def __init__(self, *shape: int, bias: bool = True, bias_shape: Optional[Tuple[int, ...]] = None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*shape))
        if bias:
            self.bias = nn.Parameter(torch.empty(*(bias_shape or (shape[0],))))
        else:
            self.register_parameter("bias", None)


_DTYPES = {"f32": _lib.DTYPE_F32, "bf16": _lib.DTYPE_BF16, "fp8": _lib.DTYPE_FP8, "f32x3": _lib.DTYPE_F32X3, "f32x6": _lib.DTYPE_F32X6}


class _Holder(nn.Module):
    pass


@dataclass
class DiTEncoderOutput:
    """What ``BeitModel.forward`` returns, reduced to what LayoutDiT reads.  ``hidden_states`` has ``L+1`` slots;
    slots the caller did not ask for are ``None`` (the reference only indexes ``layer_idxs``)."""
    hidden_states: Tuple[Optional[torch.Tensor], ...]
    last_hidden_state: Optional[torch.Tensor] = None
    pooler_output: None = None   # BeitPooler is dead work for LayoutDiT (never read at dit_backbone.py:47)


class DiTEncoder(nn.Module):
    def __init__(self, config: Optional[DiTConfig] = None, compute_dtype: str = "f32"):
        """``compute_dtype``: arithmetic of the INFERENCE forward - ``"f32"`` (exact-fp32 MFMA, the parity path), ``"bf16"``
        (bf16 GEMM / attention operands with fp32 accumulation, residual stream, LayerNorm and softmax; parameters and
        returned taps stay fp32) or ``"fp8"`` (the four GEMMs of a layer on fp8 e4m3 operands with per-tensor scales,
        attention on bf16; needs one ``calibrate_fp8(sample_batch)`` call before the first forward).
        ``"f32x3"`` / ``"f32x6"``: the fp32 forward with every GEMM operand held as two / three bf16 planes and every product
        formed from three / six plane products on the bf16 MFMA (16x the fp32 matrix rate) with fp32 accumulation; LayerNorm,
        softmax, erf-GELU and the residual stream exactly as ``"f32"``.  ``"f32x6"`` has the error of fp32 arithmetic (6-8e-7 vs
        float64, under the ``"f32"`` build's own) at 1.4x its speed, ``"f32x3"`` 4.4-5.6e-6 - inside the fp32 build's own parity
        gates, 180x inside the reference's 1e-3 - at 2.4-2.5x (include/ldit.h, LDIT_F32X3).

        TRAINING (``.train()`` + ``loss.backward()``, ref trainer.py:168-180) is mixed precision for the ``"f32"`` and
        ``"bf16"`` builds alike: bf16 MFMA operands, fp32 accumulation, fp32 residual stream / LayerNorm / softmax /
        gradients / master parameters - what the reference's CUDA branch does with fp16 autocast + GradScaler
        (trainer.py:168,177-180; bf16 needs no loss scaling, a scaled loss passes through unharmed) and within the
        bf16 gate of its fp32 CPU branch (trainer.py:171-172; gradients rel-L2 <= 3e-2 per tensor, measured <= 1e-2:
        tests/test_gpu_train.py).  The ``"fp8"`` build is inference only.

        ACTIVATION: HF's ``hidden_act="gelu"`` is the exact erf-GELU.  The ``"f32"`` / ``"f32x3"`` / ``"f32x6"`` inference forwards
        evaluate it (< 1 ulp); the bf16 / fp8 inference forwards and EVERY training forward evaluate its logistic form
        ``v * sigmoid(1.5958 v + 0.07136 v^3)`` (|deviation| <= 4.8e-4, below the bf16 rounding that follows) and the backward
        differentiates that same function.  An ``"f32"`` model therefore trains through a (slightly) different activation, on bf16
        operands, than its eval forward computes - inside the stated gates, documented here and in INTEGRATION.md."""
        super().__init__()
        if compute_dtype not in _DTYPES:
            raise ValueError(f"compute_dtype {compute_dtype!r}: expected 'f32', 'f32x3', 'f32x6', 'bf16' or 'fp8'")
        self.compute_dtype = compute_dtype
        self.config = config or DiTConfig()
        cfg = self.config
        Cc, Fm, p, ch = cfg.hidden_size, cfg.intermediate_size, cfg.patch_size, cfg.num_channels

        emb = _Holder()
        emb.cls_token = nn.Parameter(torch.zeros(1, 1, Cc))
        emb.mask_token = nn.Parameter(torch.zeros(1, 1, Cc))           # inert: kept for key compatibility
        emb.position_embeddings = nn.Parameter(torch.zeros(1, cfg.num_patches + 1, Cc))
        emb.patch_embeddings = _Holder()
        emb.patch_embeddings.projection = _Affine(Cc, ch, p, p)
        self.embeddings = emb

        self.encoder = _Holder()
        layers = []
        for _ in range(cfg.num_hidden_layers):
            blk = _Holder()
            blk.lambda_1 = nn.Parameter(torch.full((Cc,), cfg.layer_scale_init_value))
            blk.lambda_2 = nn.Parameter(torch.full((Cc,), cfg.layer_scale_init_value))
            blk.layernorm_before = _Affine(Cc)
            blk.layernorm_after = _Affine(Cc)
            blk.attention = _Holder()
            blk.attention.attention = _Holder()
            blk.attention.attention.query = _Affine(Cc, Cc)
            blk.attention.attention.key = _Affine(Cc, Cc, bias=False)   # TF:models/beit/modeling_beit.py:306
            blk.attention.attention.value = _Affine(Cc, Cc)
            blk.attention.output = _Holder()
            blk.attention.output.dense = _Affine(Cc, Cc)
            blk.intermediate = _Holder()
            blk.intermediate.dense = _Affine(Fm, Cc)
            blk.output = _Holder()
            blk.output.dense = _Affine(Cc, Fm)
            layers.append(blk)
        self.encoder.layer = nn.ModuleList(layers)
        self.pooler = _Holder()
        self.pooler.layernorm = _Affine(Cc)                              # inert: kept for key compatibility
        self.reset_parameters()

        # fp8 activation scales [L, 4] (order: enum ldit_fp8_act); zeros = not calibrated.  Not part of state_dict():
        # the checkpoint stays key-compatible with BeitModel.
        self.register_buffer("fp8_act_scales", torch.zeros(cfg.num_hidden_layers, _lib.FP8_A_COUNT), persistent=False)
        # fp8: per-channel smoothing factors of the two LayerNorm outputs of every layer ([L, 2, C]; ones = off).  The fold is a
        # pack-time transform (SmoothQuant): LN gamma / beta are divided by s, the matching input columns of the next GEMM's
        # weight multiplied by s - the product is unchanged, the fp8 operands are flatter (see calibrate_fp8).
        self.register_buffer("fp8_smooth", torch.ones(cfg.num_hidden_layers, 2, Cc), persistent=False)
        self._packed: Optional[torch.Tensor] = None
        self._packed_key = None
        self._workspace: Optional[torch.Tensor] = None
        self._pos_cache: Dict[Tuple[int, int], Tuple[object, torch.Tensor]] = {}

    # ---- parameters ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def reset_parameters(self) -> None:
        """BEiT's own init (TF:models/beit/modeling_beit.py:467-482): N(0, 0.02) matrices, zero biases, LN = (1, 0)."""
        for name, p in self.named_parameters():
            if "lambda_" in name:
                p.fill_(self.config.layer_scale_init_value)
            elif "layernorm" in name:
                p.fill_(1.0) if name.endswith("weight") else p.zero_()
            elif name.endswith("bias") or "token" in name or "position" in name:
                p.zero_()
            else:
                p.normal_(0.0, 0.02)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """Accepts transformers-4.49 keys, 5.x keys and the detector-prefixed keys the reference saves."""
        return super().load_state_dict({to_v4(k): v for k, v in state_dict.items()}, strict=strict, assign=assign)

    @torch.no_grad()
    def load_numpy(self, weights) -> "DiTEncoder":
        sd = {k: torch.from_numpy(v.copy()) for k, v in weights.items()}
        self.load_state_dict(sd, strict=True)
        return self

    def mark_parameters_changed(self) -> None:
        """Invalidate every cached copy of the parameters (the packed inference block, resampled position tables, the bf16
        mirror of the training state).  The caches are keyed on autograd's version counters, which writes through
        ``p.data`` (``p.data.copy_()`` / ``p.data.mul_()``, EMA-style code) and foreign kernels do not move: call this
        after such a write.  ``load_state_dict``, ``p.copy_()`` and optimizers need no call."""
        self._packed_key = None
        self._pos_cache.clear()
        st = getattr(self, "_flat_state", None)
        if st is not None:
            st.mark_dirty()

    # ---- packing / scratch -------------------------------------------------------------------------------------------
    def _lcfg(self, img_h: int, img_w: int, taps: Sequence[int]) -> _lib.LditCfg:
        cfg = self.config
        c = _lib.LditCfg(hidden=cfg.hidden_size, layers=cfg.num_hidden_layers, heads=cfg.num_attention_heads,
                         mlp=cfg.intermediate_size, patch=cfg.patch_size, in_ch=cfg.num_channels, img_h=img_h,
                         img_w=img_w, n_taps=len(taps), ln_eps=cfg.layer_norm_eps,
                         dtype=_DTYPES[self.compute_dtype], flags=0)
        for i, t in enumerate(taps):
            c.taps[i] = t
        return c

    def _position_table(self, gh: int, gw: int) -> torch.Tensor:
        """[1+gh*gw, C] table for this grid: the parameter itself, or its bicubic resample
        (TF:models/beit/modeling_beit.py:113-151) - host-side plumbing done once per grid and parameter version."""
        pe = self.embeddings.position_embeddings
        g0 = self.config.image_size // self.config.patch_size
        if gh == g0 and gw == g0:
            return pe.detach().reshape(-1, pe.shape[-1])
        key = (pe.data_ptr(), pe._version, str(pe.device))
        hit = self._pos_cache.get((gh, gw))
        if hit is not None and hit[0] == key:
            return hit[1]
        with torch.no_grad():
            table = resample_position_table(pe, g0, gh, gw)
        self._pos_cache[(gh, gw)] = (key, table)
        return table

    def _pack(self, lcfg: _lib.LditCfg, pos: torch.Tensor, device: torch.device) -> torch.Tensor:
        params = [p for p in self.parameters()]
        key = (str(device), lcfg.img_h, lcfg.img_w, lcfg.dtype, pos.data_ptr(),
               tuple((p.data_ptr(), p._version) for p in params), self.fp8_act_scales._version, self.fp8_smooth._version)
        if self._packed is not None and self._packed_key == key:
            return self._packed
        lib = _lib.load()
        for p in params:
            if p.device != device or p.dtype != torch.float32:
                raise ValueError("DiTEncoder parameters must be float32 on the input's GPU (call .to(device))")
        nbytes = lib.ldit_packed_bytes(C.byref(lcfg))
        if nbytes == 0:
            raise _lib.LditError(_lib.LDIT_EUNSUPPORTED, lib.ldit_last_error().decode())
        packed = torch.empty(nbytes, dtype=torch.uint8, device=device)
        L = self.config.num_hidden_layers
        layers = (_lib.LditLayerWeights * max(L, 1))()
        keep = []

        def ptr(t: torch.Tensor) -> int:
            t = t.detach()
            if not t.is_contiguous():
                t = t.contiguous()
            keep.append(t)
            return t.data_ptr()

        smooth = lcfg.dtype == _lib.DTYPE_FP8 and bool((self.fp8_smooth != 1.0).any())
        for i, blk in enumerate(self.encoder.layer):
            a = blk.attention.attention
            lw = layers[i]
            ln1_w, ln1_b, wq, wk, wv = blk.layernorm_before.weight, blk.layernorm_before.bias, a.query.weight, a.key.weight, a.value.weight
            ln2_w, ln2_b, w1 = blk.layernorm_after.weight, blk.layernorm_after.bias, blk.intermediate.dense.weight
            if smooth:
                # pack-time fold on temporaries (parameter-sized host-side plumbing; the parameters themselves are untouched)
                s1, s2 = self.fp8_smooth[i, 0].to(device), self.fp8_smooth[i, 1].to(device)
                ln1_w, ln1_b, ln2_w, ln2_b = ln1_w.detach() / s1, ln1_b.detach() / s1, ln2_w.detach() / s2, ln2_b.detach() / s2
                wq, wk, wv, w1 = wq.detach() * s1, wk.detach() * s1, wv.detach() * s1, w1.detach() * s2
            lw.ln1_w, lw.ln1_b = ptr(ln1_w), ptr(ln1_b)
            lw.wq, lw.bq, lw.wk = ptr(wq), ptr(a.query.bias), ptr(wk)
            lw.wv, lw.bv = ptr(wv), ptr(a.value.bias)
            lw.wo, lw.bo = ptr(blk.attention.output.dense.weight), ptr(blk.attention.output.dense.bias)
            lw.lam1 = ptr(blk.lambda_1)
            lw.ln2_w, lw.ln2_b = ptr(ln2_w), ptr(ln2_b)
            lw.w1, lw.b1 = ptr(w1), ptr(blk.intermediate.dense.bias)
            lw.w2, lw.b2 = ptr(blk.output.dense.weight), ptr(blk.output.dense.bias)
            lw.lam2 = ptr(blk.lambda_2)
        proj = self.embeddings.patch_embeddings.projection
        w = _lib.LditWeights(patch_w=ptr(proj.weight), patch_b=ptr(proj.bias), cls=ptr(self.embeddings.cls_token),
                             pos=ptr(pos), layer=layers)
        _lib.check(lib.ldit_pack_weights(C.byref(lcfg), C.byref(w), packed.data_ptr(), nbytes,
                                         torch.cuda.current_stream(device).cuda_stream))
        if lcfg.dtype == _lib.DTYPE_FP8:
            sc = self.fp8_act_scales.detach().to("cpu", torch.float32).contiguous()
            if not bool((sc > 0).all()):
                raise RuntimeError("compute_dtype='fp8': activation scales are not calibrated - call "
                                   "calibrate_fp8(sample_pixel_values) once (or assign fp8_act_scales) before the forward")
            arr = (C.c_float * sc.numel())(*sc.reshape(-1).tolist())
            _lib.check(lib.ldit_set_fp8_act_scales(C.byref(lcfg), packed.data_ptr(), nbytes, arr,
                                                   torch.cuda.current_stream(device).cuda_stream))
        self._packed, self._packed_key = packed, key
        return packed

    @torch.no_grad()
    def calibrate_fp8(self, pixel_values: torch.Tensor, margin: float = 1.0, smooth_alpha: float = 0.0) -> torch.Tensor:
        """Measure the four per-layer activation ranges of the fp8 build on a sample batch and store
        ``scale = margin * amax / 448`` in ``fp8_act_scales``.  Host-side sequencing of the library's own fp32 kernels
        (``layoutdit_amd.ops``), layer by layer, so the statistics are those of the exact path.

        ``smooth_alpha > 0`` (0.5 is SmoothQuant's usual value; default 0 = off) also measures the per-CHANNEL range of the two LayerNorm outputs and stores the
        SmoothQuant factors ``s_c = amax_c(y)^alpha / amax_c(|W[:, c]|)^(1 - alpha)`` (normalised to geometric mean 1) in
        ``fp8_smooth``: pretrained BEiT / DiT checkpoints carry a few LayerNorm channels tens of times larger than the rest,
        which a per-tensor fp8 scale would pay for with the resolution of all others.  Dividing those channels by ``s_c`` in
        the LayerNorm's affine and multiplying the next GEMM's input columns by ``s_c`` leaves ``y W^T`` unchanged; the
        weight rows are re-quantised per output channel at pack time as always.  Off by default: on the synthetic outlier stress
        of tests/test_gpu_lowp_pinning.py it moves the error by -8 ... +1 % only (e4m3's exponent already absorbs a 60x channel;
        the stress loses its accuracy in the saturated attention logits) - it is there for checkpoints whose LayerNorm
        outliers exceed what e4m3's range covers."""
        from .. import ops
        cfg = self.config
        if not pixel_values.is_cuda:
            raise RuntimeError("calibrate_fp8 runs on the GPU kernels: move the sample batch to a HIP device")
        x = pixel_values.detach().to(torch.float32).contiguous()
        B, _, H, W = x.shape
        p, Cc, Hh = cfg.patch_size, cfg.hidden_size, cfg.num_attention_heads
        pos = self._position_table(H // p, W // p).contiguous()
        proj = self.embeddings.patch_embeddings.projection
        h = ops.embed(x, proj.weight.detach().contiguous(), proj.bias.detach(), self.embeddings.cls_token.detach().reshape(-1),
                      pos, p)
        T = h.shape[1]
        h = h.reshape(B * T, Cc)
        amax = []
        smooth = torch.ones_like(self.fp8_smooth)

        def factors(y2d: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
            """per-channel s and the smoothed activation's per-tensor amax"""
            ya = ops.colamax(y2d).clamp_min(1e-12)                                  # [C] range of each LayerNorm channel
            wa = ops.colamax(w).clamp_min(1e-12)                                    # [C] range of the weight's input column
            s_ = ya.pow(smooth_alpha) / wa.pow(1.0 - smooth_alpha)
            s_ = s_ / torch.exp(torch.log(s_).mean())                               # geometric mean 1: keeps the overall scale
            return s_, (ya / s_).amax().reshape(1)

        for li, blk in enumerate(self.encoder.layer):
            a = blk.attention.attention
            y = ops.layernorm(h, blk.layernorm_before.weight.detach(), blk.layernorm_before.bias.detach(), cfg.layer_norm_eps)
            wqkv = torch.cat([a.query.weight, a.key.weight, a.value.weight]).detach().contiguous()
            bqkv = torch.cat([a.query.bias, torch.zeros_like(a.query.bias), a.value.bias]).detach().contiguous()
            qkv = ops.linear(y, wqkv, bqkv).reshape(B, T, 3 * Cc)
            o = ops.attention(qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:], Hh).reshape(B * T, Cc)
            h = ops.linear(o, blk.attention.output.dense.weight.detach(), blk.attention.output.dense.bias.detach(),
                           epilogue=_lib.EPI_SCALE_RESID, lam=blk.lambda_1.detach(), residual=h, out=h)
            y2 = ops.layernorm(h, blk.layernorm_after.weight.detach(), blk.layernorm_after.bias.detach(), cfg.layer_norm_eps)
            g = ops.linear(y2, blk.intermediate.dense.weight.detach(), blk.intermediate.dense.bias.detach(),
                           epilogue=_lib.EPI_BIAS_GELU)
            h = ops.linear(g, blk.output.dense.weight.detach(), blk.output.dense.bias.detach(),
                           epilogue=_lib.EPI_SCALE_RESID, lam=blk.lambda_2.detach(), residual=h, out=h)
            if smooth_alpha > 0.0:
                smooth[li, 0], ay = factors(y, wqkv)
                smooth[li, 1], ay2 = factors(y2, blk.intermediate.dense.weight.detach().contiguous())
            else:
                ay, ay2 = ops.amax(y), ops.amax(y2)
            amax.append(torch.cat([ay, ops.amax(o), ay2, ops.amax(g)]))
        scales = torch.stack(amax).clamp_min(1e-20) * (float(margin) / ops.FP8_MAX)
        self.fp8_act_scales.copy_(scales)
        self.fp8_smooth.copy_(smooth)
        return self.fp8_act_scales

    def _scratch(self, lcfg: _lib.LditCfg, batch: int, device: torch.device) -> torch.Tensor:
        need = _lib.load().ldit_workspace_bytes(C.byref(lcfg), batch)
        if need == 0:
            raise _lib.LditError(_lib.LDIT_EUNSUPPORTED, _lib.load().ldit_last_error().decode())
        ws = self._workspace
        if ws is None or ws.device != device or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=device)
            self._workspace = ws
        return ws

    # ---- train mode ------------------------------------------------------------------------------------------------
    def _forward_train(self, pixel_values: torch.Tensor, taps, wants_grad: bool) -> DiTEncoderOutput:
        """``model.train()`` semantics of HF BeitModel (ref trainer.py:38,206): per-sample stochastic depth on both residual
        branches (TF:360-378,432-434,440-442) and - when parameters require grad - participation in autograd through
        ``layoutdit_amd.training`` (C ABI: ldit_vit_forward_train / ldit_vit_backward).  bf16 build only."""
        from .. import training
        cfg = self.config
        if self.compute_dtype == "fp8":
            raise NotImplementedError("the fp8 build is inference only: train a DiTEncoder(..., compute_dtype='bf16' or 'f32') "
                                      "(both train on bf16 MFMA operands with fp32 master parameters), or call .eval()")
        x = self._pixels_f32(pixel_values)
        L = cfg.num_hidden_layers
        drop = None
        if cfg.drop_path_rate > 0.0 and L > 1:
            drop = training.sample_drop_scales(L, x.shape[0], cfg.drop_path_rate, x.device)
        if wants_grad:
            outs = training.encoder_forward_autograd(self, x, taps, drop)
        else:
            st = training.flat_state(self, x.shape[2], x.shape[3])
            st.repack()
            with torch.no_grad():
                outs = st.forward(x, taps, drop, st.saved_nograd(x.shape[0]))
        hidden: List[Optional[torch.Tensor]] = [None] * (L + 1)
        for t, o in zip(taps, outs):
            hidden[t] = o
        if pixel_values.dtype != torch.float32 and not wants_grad:
            hidden = self._like_input(hidden, pixel_values.dtype)
        return DiTEncoderOutput(hidden_states=tuple(hidden), last_hidden_state=hidden[L])

    def _train_forward_applies(self, H: int, W: int) -> bool:
        """Train mode WITHOUT gradients (``model.train()`` under ``torch.no_grad()``): HF still applies stochastic depth
        (TF:360-378), which only the training forward implements.  With a drop-path rate of 0 (or one layer) train and
        eval arithmetic coincide and the inference kernels of ``compute_dtype`` run; otherwise the training forward runs
        (any grid since round 3); the fp8 build refuses rather than silently computing without stochastic depth."""
        cfg = self.config
        if cfg.drop_path_rate <= 0.0 or cfg.num_hidden_layers < 2:
            return False
        if self.compute_dtype == "fp8":
            raise NotImplementedError("train mode with stochastic depth on the fp8 (inference-only) build: call .eval()")
        return True

    # ---- forward -------------------------------------------------------------------------------------------------
    def forward(self, pixel_values: torch.Tensor, taps: Optional[Sequence[int]] = None,
                _timing: Optional[dict] = None) -> DiTEncoderOutput:
        cfg = self.config
        if pixel_values.dim() != 4:
            raise ValueError(f"pixel_values must be [B, C, H, W], got {tuple(pixel_values.shape)}")
        if pixel_values.shape[1] != cfg.num_channels:
            # same condition and wording as TF:models/beit/modeling_beit.py:84-89
            raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in the "
                             f"configuration. Expected {cfg.num_channels} but got {pixel_values.shape[1]}.")
        if not pixel_values.is_cuda:
            raise RuntimeError("DiTEncoder runs only on the GPU through libldit_hip.so; move the input and the module "
                               "to a HIP device (there is no CPU / eager fallback)")
        B, _, H, W = pixel_values.shape
        p = cfg.patch_size
        if H % p or W % p:
            raise ValueError(f"input {H}x{W} is not a multiple of the patch size {p}")
        taps = list(cfg.taps if taps is None else taps)
        if len(taps) > _lib.LDIT_MAX_TAPS:
            raise ValueError(f"at most {_lib.LDIT_MAX_TAPS} hidden states per call")
        device = pixel_values.device
        wants_grad = torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters())
        if self.training and (wants_grad or self._train_forward_applies(H, W)):
            return self._forward_train(pixel_values, taps, wants_grad)
        x = self._pixels_f32(pixel_values)
        with torch.no_grad(), torch.cuda.device(device):
            lib = _lib.load()
            gh, gw = H // p, W // p
            lcfg = self._lcfg(H, W, taps)
            pos = self._position_table(gh, gw)
            packed = self._pack(lcfg, pos, device)
            ws = self._scratch(lcfg, B, device)
            T = gh * gw + 1
            outs = [torch.empty((B, T, cfg.hidden_size), dtype=torch.float32, device=device) for _ in taps]
            tap_ptrs = (C.c_void_p * max(len(outs), 1))(*[o.data_ptr() for o in outs])
            stream = torch.cuda.current_stream(device).cuda_stream
            if _timing is None:
                _lib.check(lib.ldit_vit_forward(C.byref(lcfg), packed.data_ptr(), x.data_ptr(), B, tap_ptrs, ws.data_ptr(),
                                                ws.numel(), stream))
            else:
                ms = (C.c_double * _lib.K_COUNT)()
                cnt = (C.c_int64 * _lib.K_COUNT)()
                _lib.check(lib.ldit_vit_forward_timed(C.byref(lcfg), packed.data_ptr(), x.data_ptr(), B, tap_ptrs,
                                                      ws.data_ptr(), ws.numel(), stream, ms, cnt))
                for i, name in enumerate(_lib.KERNEL_FAMILIES):
                    _timing[name + "_ms"] = _timing.get(name + "_ms", 0.0) + ms[i]
                    _timing[name + "_launches"] = _timing.get(name + "_launches", 0) + cnt[i]
        hidden: List[Optional[torch.Tensor]] = [None] * (cfg.num_hidden_layers + 1)
        for t, o in zip(taps, outs):
            hidden[t] = o
        hidden = self._like_input(hidden, pixel_values.dtype)
        return DiTEncoderOutput(hidden_states=tuple(hidden), last_hidden_state=hidden[cfg.num_hidden_layers])

    def forward_image_list(self, images: Sequence[torch.Tensor], size: Optional[Tuple[int, int]] = None, mean: float = 0.5,
                           std: float = 0.5, taps: Optional[Sequence[int]] = None) -> DiTEncoderOutput:
        """Inference forward fed by the detector's image list BEFORE its input transform (ref model.py:50-54: torchvision's
        ``GeneralizedRCNNTransform`` with ``fixed_size``, ``image_mean = image_std = 0.5``): a ragged list of ``[3, h, w]`` tensors in
        [0, 1], fp32 or fp16.  Equal - bit for bit - to ``self(DetectorInputTransform(...)(images).tensors)``, but the bf16 / fp8 /
        split-fp32 builds never materialise the resized fp32 batch: the pass that writes the patch-embedding operand evaluates the
        normalise + bilinear resize itself (C ABI ``ldit_vit_forward_images``).  Eval mode only (the training
        forward keeps the two-step path: its backward re-reads the resized pixels)."""
        from .. import ops
        cfg = self.config
        if self.training:
            raise RuntimeError("forward_image_list is the inference entry; in train mode feed DetectorInputTransform's batch to forward()")
        H, W = size or (cfg.image_size, cfg.image_size)
        p = cfg.patch_size
        if H % p or W % p:
            raise ValueError(f"target size {H}x{W} is not a multiple of the patch size {p}")
        imgs, ptrs, hs, ws_, half, ch = ops.image_list_args(images)
        if ch != cfg.num_channels:
            raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in the "
                             f"configuration. Expected {cfg.num_channels} but got {ch}.")
        taps = list(cfg.taps if taps is None else taps)
        device = imgs[0].device
        B = len(imgs)
        with torch.no_grad(), torch.cuda.device(device):
            lib = _lib.load()
            gh, gw = H // p, W // p
            lcfg = self._lcfg(H, W, taps)
            packed = self._pack(lcfg, self._position_table(gh, gw), device)
            ws = self._scratch(lcfg, B, device)
            T = gh * gw + 1
            outs = [torch.empty((B, T, cfg.hidden_size), dtype=torch.float32, device=device) for _ in taps]
            tap_ptrs = (C.c_void_p * max(len(outs), 1))(*[o.data_ptr() for o in outs])
            _lib.check(lib.ldit_vit_forward_images(C.byref(lcfg), packed.data_ptr(), ptrs, hs, ws_, int(half), float(mean), float(std), B,
                                                   tap_ptrs, ws.data_ptr(), ws.numel(), torch.cuda.current_stream(device).cuda_stream))
        hidden: List[Optional[torch.Tensor]] = [None] * (cfg.num_hidden_layers + 1)
        for t, o in zip(taps, outs):
            hidden[t] = o
        hidden = self._like_input(hidden, torch.float16 if half else torch.float32)
        return DiTEncoderOutput(hidden_states=tuple(hidden), last_hidden_state=hidden[cfg.num_hidden_layers])

    @staticmethod
    def _pixels_f32(pixel_values: torch.Tensor) -> torch.Tensor:
        """fp32 NCHW contiguous pixels for the kernels.  fp16 batches (the reference's trainer feeds ``.half()`` images under
        fp16 autocast, ref trainer.py:153-155,168) are widened by the library (ldit_cast_f16_f32), not by a framework cast."""
        x = pixel_values.detach()
        if x.dtype == torch.float32:
            return x.contiguous()
        if x.dtype == torch.float16:
            from .. import ops
            return ops.widen_f16(x.contiguous())
        raise ValueError(f"pixel_values must be float32 or float16, got {x.dtype}")

    @staticmethod
    def _like_input(hidden, dtype):
        if dtype == torch.float32:
            return hidden
        from .. import ops
        return [None if h is None else ops.narrow_f16(h) for h in hidden]
