"""Train step of the encoder (BASELINE.json configs[2]): host-side mirror of what the reference's loop
does to ``self.dit`` - ``loss.backward()`` through ``hidden_states`` and ``AdamW.step()``
(ref ``src/layoutdit/training/trainer.py:62-68,148-187``) - on the library's C ABI
(``ldit_vit_forward_train`` / ``ldit_vit_backward`` / ``ldit_adamw_step`` / ``ldit_pack_train``, ``include/ldit.h``).

Three pieces:

* :class:`FlatState` - the encoder's parameters re-homed as views of ONE flat fp32 block in the library's layout
  (``ldit_flat_param_layout``), with a same-shaped flat gradient block; what DDP / apex call "flattening".
* :func:`encoder_forward_autograd` - a ``torch.autograd.Function`` around the C ABI, so the reference's own trainer
  (``loss.backward()`` + ``torch.optim.AdamW``) works unchanged with ``backbone.dit = DiTEncoder(...)`` in train mode.
* :class:`TrainStep` - the fused step ``bench.py --config 2`` measures: training forward, backward in per-layer stages with
  the gradient all-reduce of each finished layer (one bucket per layer, RCCL over xGMI) overlapped with the rest of the
  backward, fused AdamW on the flat block, re-pack of the bf16 operand copies.

PyTorch is plumbing here (device memory, streams, ``torch.distributed``, the stochastic-depth coin flips); there is no
eager fallback - every FLOP of forward, backward and update runs in ``libldit_hip.so``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _lib
from .dp import Rank

_LAYER_SLOTS = ("ln1_w", "ln1_b", "wqkv", "bqkv", "wo", "bo", "lam1", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2", "lam2")


def drop_path_rates(num_layers: int, rate: float) -> List[float]:
    """Stochastic-depth rate per layer, ``rate * i / (L - 1)`` (TF:models/beit/modeling_beit.py:499-502)."""
    return [rate * i / max(num_layers - 1, 1) for i in range(num_layers)]


def sample_drop_scales(num_layers: int, batch: int, rate: float, device, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """``[L, 2, B]`` per-sample factors of the two residual branches of every layer: ``floor(keep + U(0,1)) / keep``
    exactly as ``BeitDropPath`` draws them (TF:360-378); layers with rate 0 get ones."""
    keep = 1.0 - torch.tensor(drop_path_rates(num_layers, rate), dtype=torch.float32, device=device).view(-1, 1, 1)
    u = torch.rand((num_layers, 2, batch), dtype=torch.float32, device=device, generator=generator)
    return (torch.floor(u + keep) / keep).contiguous()


class FlatState:
    """The encoder's trainable parameters as views of one flat fp32 block (and their gradients as views of another)."""

    def __init__(self, encoder, img_h: int, img_w: int):
        cfg = encoder.config
        g0 = cfg.image_size // cfg.patch_size
        self.grid = (img_h // cfg.patch_size, img_w // cfg.patch_size)
        # At the position table's own grid the embeddings.position_embeddings parameter is a view of the flat block like every
        # other parameter.  At another grid (round 3) the block's `pos` slot holds the table RESAMPLED for that grid (TF:113-151,
        # bicubic) - a derived tensor, refreshed from the parameter when it changes; the library differentiates with respect to
        # the slot, and the resample's adjoint (torch autograd on the parameter-sized op) carries that gradient to the parameter.
        self.native = self.grid == (g0, g0)
        self.encoder_ref = encoder
        lib = _lib.load()
        self.lcfg = encoder._lcfg(img_h, img_w, cfg.taps)
        self.lcfg.dtype = _lib.DTYPE_BF16
        L, Cc = cfg.num_hidden_layers, cfg.hidden_size
        n_off = 4 + 14 * L + 1
        offs = (C.c_int64 * n_off)()
        _lib.check(lib.ldit_flat_param_layout(C.byref(self.lcfg), offs, n_off))
        self.offsets = list(offs)
        self.numel = self.offsets[-1]
        self.device = next(encoder.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("training runs only on the GPU through libldit_hip.so: move the module to a HIP device first")
        self.params = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self.layer_start = [self.offsets[4 + 14 * l] for l in range(L)] + [self.numel]
        self.named: List[Tuple[str, torch.nn.Parameter, int, Tuple[int, ...]]] = []      # (name, parameter, offset, shape)

        def bind(p: torch.nn.Parameter, off: int, name: str):
            self.named.append((name, p, off, tuple(p.shape)))

        emb = encoder.embeddings
        bind(emb.patch_embeddings.projection.weight, self.offsets[0], "patch_w")
        bind(emb.patch_embeddings.projection.bias, self.offsets[1], "patch_b")
        bind(emb.cls_token, self.offsets[2], "cls")
        self.pos_param = emb.position_embeddings
        self.pos_off, self.pos_numel = self.offsets[3], (self.grid[0] * self.grid[1] + 1) * Cc
        self._pos_synced = None
        if self.native:
            bind(emb.position_embeddings, self.offsets[3], "pos")
        for l, blk in enumerate(encoder.encoder.layer):
            o = dict(zip(_LAYER_SLOTS, self.offsets[4 + 14 * l: 4 + 14 * (l + 1)]))
            a = blk.attention.attention
            bind(blk.layernorm_before.weight, o["ln1_w"], f"{l}.ln1_w")
            bind(blk.layernorm_before.bias, o["ln1_b"], f"{l}.ln1_b")
            bind(a.query.weight, o["wqkv"], f"{l}.wq")
            bind(a.key.weight, o["wqkv"] + Cc * Cc, f"{l}.wk")
            bind(a.value.weight, o["wqkv"] + 2 * Cc * Cc, f"{l}.wv")
            bind(a.query.bias, o["bqkv"], f"{l}.bq")
            bind(a.value.bias, o["bqkv"] + 2 * Cc, f"{l}.bv")
            bind(blk.attention.output.dense.weight, o["wo"], f"{l}.wo")
            bind(blk.attention.output.dense.bias, o["bo"], f"{l}.bo")
            bind(blk.lambda_1, o["lam1"], f"{l}.lam1")
            bind(blk.layernorm_after.weight, o["ln2_w"], f"{l}.ln2_w")
            bind(blk.layernorm_after.bias, o["ln2_b"], f"{l}.ln2_b")
            bind(blk.intermediate.dense.weight, o["w1"], f"{l}.w1")
            bind(blk.intermediate.dense.bias, o["b1"], f"{l}.b1")
            bind(blk.output.dense.weight, o["w2"], f"{l}.w2")
            bind(blk.output.dense.bias, o["b2"], f"{l}.b2")
            bind(blk.lambda_2, o["lam2"], f"{l}.lam2")
        with torch.no_grad():
            for _, p, off, shape in self.named:
                if p.dtype != torch.float32 or p.device != self.device:
                    raise ValueError("DiTEncoder parameters must be float32 on one GPU")
                view = self.params[off: off + p.numel()].view(shape)
                view.copy_(p.data)
                p.data = view
        # scratch of the library, sized once per batch size
        self.packed = torch.empty(lib.ldit_train_mirror_bytes(C.byref(self.lcfg)), dtype=torch.uint8, device=self.device)   # bf16 mirror
        self._packed_version = None
        self._dirty = True
        self._ws: Dict[int, torch.Tensor] = {}
        self._saved_nograd: Optional[Tuple[int, torch.Tensor]] = None

    # ---- bookkeeping ---------------------------------------------------------------------------------------------------
    def intact(self) -> bool:
        """False once something (``.to()``, ``load_state_dict(assign=True)``) re-homed a parameter."""
        base = self.params.data_ptr()
        return all(p.data_ptr() == base + 4 * off for _, p, off, _ in self.named)

    def grad_view(self, off: int, shape) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= s
        return self.grads[off: off + n].view(shape)

    def bucket(self, stage: int) -> torch.Tensor:
        """Gradient slice finished by backward stage ``stage`` (layer ``stage`` for ``stage >= 1``, embeddings for 0)."""
        if stage == 0:
            return self.grads[: self.layer_start[0]]
        return self.grads[self.layer_start[stage - 1]: self.layer_start[stage]]

    def version(self):
        return tuple(p._version for _, p, _, _ in self.named) + (self.pos_param._version,)

    def sync_pos(self) -> None:
        """Non-native grid: (re)fill the block's `pos` slot with the bicubic resample of the position parameter."""
        if self.native:
            return
        key = (self.pos_param.data_ptr(), self.pos_param._version)
        if self._pos_synced != key or self._dirty:
            from .modeling.dit_encoder import resample_position_table
            g0 = self.encoder_ref.config.image_size // self.encoder_ref.config.patch_size
            with torch.no_grad():
                table = resample_position_table(self.pos_param.detach(), g0, self.grid[0], self.grid[1])
                self.params[self.pos_off: self.pos_off + self.pos_numel].copy_(table.reshape(-1))
            self._pos_synced = key

    def pos_param_grad(self, flat_grads: torch.Tensor) -> torch.Tensor:
        """Gradient of the position PARAMETER from the gradient of the resampled table (non-native grid): the adjoint of the
        bicubic resample, by torch autograd on the parameter-sized op (host-side plumbing, like the resample itself)."""
        from .modeling.dit_encoder import resample_position_table
        g0 = self.encoder_ref.config.image_size // self.encoder_ref.config.patch_size
        d_table = flat_grads[self.pos_off: self.pos_off + self.pos_numel].view(-1, self.pos_param.shape[-1])
        with torch.enable_grad():
            pe = self.pos_param.detach().requires_grad_(True)
            table = resample_position_table(pe, g0, self.grid[0], self.grid[1])
            (g,) = torch.autograd.grad(table, pe, d_table)
        return g

    def mark_dirty(self) -> None:
        """Tell the state that the flat block was written behind autograd's version counters (``p.data.copy_()``,
        ``p.data.mul_()``, EMA-style code, a foreign kernel): the next :meth:`repack` rebuilds the bf16 mirror.  Writes through
        the parameters themselves (``p.copy_()``, optimizers) bump ``p._version`` and need no call."""
        self._dirty = True

    def repack(self, force: bool = False) -> None:
        """flat fp32 parameters -> their bf16 mirror (one pass), when a parameter's version changed, after
        :meth:`mark_dirty`, or with ``force=True``.  ``.data`` writes bypass the version counter: call ``mark_dirty()``."""
        v = self.version()
        if force or self._dirty or v != self._packed_version:
            self.sync_pos()
            with torch.cuda.device(self.device):
                _lib.check(_lib.load().ldit_pack_train(C.byref(self.lcfg), self.params.data_ptr(), self.packed.data_ptr(),
                                                       self.packed.numel(), torch.cuda.current_stream(self.device).cuda_stream))
            self._packed_version = v
            self._dirty = False

    def saved_nograd(self, batch: int) -> torch.Tensor:
        """One reusable activation block per batch size for forwards that never run a backward (train mode under no_grad)."""
        if self._saved_nograd is None or self._saved_nograd[0] != batch:
            self._saved_nograd = None
            self._saved_nograd = (batch, self.new_saved(batch))
        return self._saved_nograd[1]

    def workspace(self, batch: int) -> torch.Tensor:
        ws = self._ws.get(batch)
        if ws is None:
            need = _lib.load().ldit_train_workspace_bytes(C.byref(self.lcfg), batch)
            if need == 0:
                raise _lib.LditError(_lib.LDIT_EUNSUPPORTED, _lib.load().ldit_last_error().decode())
            self._ws = {batch: torch.empty(need, dtype=torch.uint8, device=self.device)}
            ws = self._ws[batch]
        return ws

    def new_saved(self, batch: int) -> torch.Tensor:
        need = _lib.load().ldit_train_saved_bytes(C.byref(self.lcfg), batch)
        if need == 0:
            raise _lib.LditError(_lib.LDIT_EUNSUPPORTED, _lib.load().ldit_last_error().decode())
        return torch.empty(need, dtype=torch.uint8, device=self.device)

    # ---- the two C calls --------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, taps: Sequence[int], drop_scales: Optional[torch.Tensor], saved: torch.Tensor,
                timing: Optional[dict] = None) -> List[torch.Tensor]:
        lib = _lib.load()
        B, T, Cc = x.shape[0], self.lcfg.img_h // self.lcfg.patch * (self.lcfg.img_w // self.lcfg.patch) + 1, self.lcfg.hidden
        lcfg = self._cfg_with_taps(taps)
        outs = [torch.empty((B, T, Cc), dtype=torch.float32, device=self.device) for _ in taps]
        ptrs = (C.c_void_p * max(len(outs), 1))(*[o.data_ptr() for o in outs])
        ms, cnt = self._probe(timing)
        with torch.cuda.device(self.device):
            _lib.check(lib.ldit_vit_forward_train(C.byref(lcfg), self.packed.data_ptr(), self.params.data_ptr(), x.data_ptr(), B, ptrs,
                                                  None if drop_scales is None else drop_scales.data_ptr(), saved.data_ptr(),
                                                  saved.numel(), torch.cuda.current_stream(self.device).cuda_stream, ms, cnt))
        self._collect(timing, ms, cnt)
        return outs

    def backward(self, x: torch.Tensor, taps: Sequence[int], dtaps: Sequence[Optional[torch.Tensor]],
                 drop_scales: Optional[torch.Tensor], saved: torch.Tensor, stage_hi: int, stage_lo: int,
                 timing: Optional[dict] = None, grads: Optional[torch.Tensor] = None) -> None:
        """``grads``: flat fp32 block the gradients are written to (default: this state's own block, which the next backward
        overwrites; the autograd path hands in a fresh block per backward so that what it returns stays valid)."""
        lib = _lib.load()
        B = x.shape[0]
        grads = self.grads if grads is None else grads
        if grads.numel() != self.numel or grads.dtype != torch.float32 or grads.device != self.device or not grads.is_contiguous():
            raise ValueError("grads: expected a contiguous fp32 block of the flat parameter layout on the state's device")
        lcfg = self._cfg_with_taps(taps)
        ptrs = (C.c_void_p * max(len(taps), 1))(*[None if d is None else d.data_ptr() for d in dtaps])
        ws = self.workspace(B)
        ms, cnt = self._probe(timing)
        with torch.cuda.device(self.device):
            _lib.check(lib.ldit_vit_backward(C.byref(lcfg), self.params.data_ptr(), self.packed.data_ptr(), x.data_ptr(), B, ptrs,
                                             None if drop_scales is None else drop_scales.data_ptr(), saved.data_ptr(),
                                             saved.numel(), grads.data_ptr(), grads.numel() * 4, ws.data_ptr(),
                                             ws.numel(), stage_hi, stage_lo,
                                             torch.cuda.current_stream(self.device).cuda_stream, ms, cnt))
        self._collect(timing, ms, cnt)

    def _cfg_with_taps(self, taps: Sequence[int]) -> _lib.LditCfg:
        c = _lib.LditCfg.from_buffer_copy(self.lcfg)
        c.n_taps = len(taps)
        for i, t in enumerate(taps):
            c.taps[i] = t
        return c

    @staticmethod
    def _probe(timing):
        if timing is None:
            return None, None
        return (C.c_double * _lib.K_COUNT)(), (C.c_int64 * _lib.K_COUNT)()

    @staticmethod
    def _collect(timing, ms, cnt):
        if timing is not None:
            for i, name in enumerate(_lib.KERNEL_FAMILIES):
                timing[name + "_ms"] = timing.get(name + "_ms", 0.0) + ms[i]
                timing[name + "_launches"] = timing.get(name + "_launches", 0) + cnt[i]


def flat_state(encoder, img_h: int, img_w: int) -> FlatState:
    """The encoder's :class:`FlatState` for this input size, (re)built when missing or when a parameter was re-homed."""
    st = getattr(encoder, "_flat_state", None)
    if st is None or not st.intact() or (st.lcfg.img_h, st.lcfg.img_w) != (img_h, img_w):
        st = FlatState(encoder, img_h, img_w)
        encoder._flat_state = st
    return st


class _EncoderFn(torch.autograd.Function):
    """``hidden_states[taps] = encoder(x)`` with gradients for every encoder parameter ("must participate
    in autograd").  ``params`` are passed so that autograd records the dependency; their values are read from the flat block."""

    @staticmethod
    def forward(ctx, state: FlatState, x: torch.Tensor, taps: Tuple[int, ...], drop_scales: Optional[torch.Tensor], *params):
        state.repack()
        saved = state.new_saved(x.shape[0])
        outs = state.forward(x, taps, drop_scales, saved)
        ctx.state, ctx.taps, ctx.drop, ctx.saved_acts = state, taps, drop_scales, saved
        ctx.save_for_backward(x)               # the pixels: needed by the patch-embedding wgrad
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dtaps):
        st: FlatState = ctx.state
        if ctx.saved_acts is None:
            raise RuntimeError("DiTEncoder: trying to backward through the encoder a second time - its saved activations "
                               "(several GB) are freed after the first backward and retain_graph=True is not supported; "
                               "run the forward again")
        d = [None if g is None else g.contiguous().to(torch.float32) for g in dtaps]
        L = st.lcfg.layers
        (x,) = ctx.saved_tensors
        need = ctx.needs_input_grad[4:]
        n_extra = 0 if st.native else 1
        # The per-parameter gradients handed to autograd are views of ONE flat block that this backward owns: the library writes
        # it directly (every named slot is written by stages L..0), so nothing is copied.  It must not be the state's own block -
        # AccumulateGrad may keep the views as p.grad, and the next backward would overwrite them (rounds 2-3 cloned 343 MB here).
        flat = torch.empty_like(st.grads) if any(need) else None
        st.backward(x, ctx.taps, d, ctx.drop, ctx.saved_acts, L, 0, grads=flat)
        ctx.saved_acts = None
        if flat is None:
            return (None,) * (4 + len(st.named) + n_extra)
        st.grads = flat             # "the gradients of the last backward" stays where TrainStep and the tests look for it
        grads = tuple(flat[off: off + _numel(shape)].view(shape) if n else None
                      for (_, _, off, shape), n in zip(st.named, need))
        if not st.native:           # the position parameter rides behind the named ones: gradient through the resample's adjoint
            grads += (st.pos_param_grad(flat) if need[len(st.named)] else None,)
        return (None, None, None, None) + grads


def _numel(shape) -> int:
    n = 1
    for v in shape:
        n *= v
    return n


def encoder_forward_autograd(encoder, x: torch.Tensor, taps: Sequence[int], drop_scales: Optional[torch.Tensor] = None):
    """Differentiable encoder forward; returns the list of tapped hidden states (fp32 ``[B, 1+P, C]``)."""
    st = flat_state(encoder, x.shape[2], x.shape[3])
    params = [p for _, p, _, _ in st.named] + ([] if st.native else [st.pos_param])
    return list(_EncoderFn.apply(st, x, tuple(taps), drop_scales, *params))


class TrainStep:
    """Fused train step on the flat state: forward, staged backward with overlapped per-layer gradient all-reduce,
    AdamW, re-pack.  ``dtaps`` stands in for the detector head: fixed synthetic upstream gradients, one per tap
    (the head - FPN / RPN / RoI, torchvision - is outside this repository's scope)."""

    def __init__(self, encoder, rank: Optional[Rank] = None, lr: float = 1e-4, weight_decay: float = 0.0,
                 betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, seed: int = 0,
                 dtaps: Optional[Sequence[torch.Tensor]] = None, drop_path_rate: Optional[float] = None,
                 img_size: Optional[Tuple[int, int]] = None, force_comm: bool = False):
        cfg = encoder.config
        if encoder.compute_dtype == "fp8":
            raise NotImplementedError("the fp8 build is inference only; the train step runs the 'bf16' and 'f32' builds on bf16 "
                                      "MFMA operands with fp32 master parameters (BASELINE configs[2])")
        h, w = img_size or (cfg.image_size, cfg.image_size)
        self.encoder, self.rank = encoder, rank
        self.state = flat_state(encoder, h, w)
        if not self.state.native:
            raise NotImplementedError("TrainStep (fused AdamW on the flat block) needs the input grid to be the position table's own: "
                                      "at another grid the block's position slot is a derived (resampled) tensor, not a parameter; "
                                      "use the autograd path (loss.backward() + torch.optim) there")
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.exp_avg = torch.zeros_like(self.state.params)
        self.exp_avg_sq = torch.zeros_like(self.state.params)
        self.steps = 0
        self.taps = list(cfg.taps)
        self.drop_path_rate = cfg.drop_path_rate if drop_path_rate is None else drop_path_rate
        self.gen = torch.Generator(device=self.state.device)
        self.gen.manual_seed(seed + (rank.rank if rank else 0))
        self._dtaps = None if dtaps is None else [d.contiguous() for d in dtaps]
        self._dtaps_seed = seed
        self._saved = None
        self.world = rank.world if rank else 1
        # force_comm: run the staged backward + per-layer bucket all-reduce even with one rank (rehearses the RCCL path on a
        # one-GPU box: a one-rank all-reduce is the identity)
        self.comm = self.world > 1 or (force_comm and rank is not None and dist.is_initialized())

    @property
    def flat_params(self) -> torch.Tensor:
        return self.state.params

    def _upstream(self, x: torch.Tensor) -> List[torch.Tensor]:
        if self._dtaps is None or self._dtaps[0].shape[0] != x.shape[0]:
            cfg = self.encoder.config
            T = cfg.tokens(x.shape[2], x.shape[3])
            g = torch.Generator(device=x.device)
            g.manual_seed(1000 + self._dtaps_seed)
            n = x.shape[0] * T * cfg.hidden_size
            self._dtaps = [torch.randn((x.shape[0], T, cfg.hidden_size), device=x.device, generator=g) / n ** 0.5
                           for _ in self.taps]
        return self._dtaps

    def _all_reduce(self, t: torch.Tensor):
        """Sum over ranks.  RCCL (backend nccl): asynchronous on RCCL's own stream, ordered after the kernels already enqueued
        on the current stream - the remaining backward overlaps it.  gloo (CPU tests): through a host copy, synchronous."""
        if self.rank.backend == "nccl":
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        t.copy_(host)
        return None

    @torch.no_grad()
    def step(self, x: torch.Tensor, _timing: Optional[dict] = None, drop_scales: Optional[torch.Tensor] = None,
             _trace: Optional[list] = None) -> List[torch.Tensor]:
        """``_trace`` (tests): receives ``(what, stage, torch.cuda.Event)`` tuples recorded on the compute stream in host order -
        ``"backward_done"`` after a stage's backward kernels, ``"allreduce_issued"`` after its bucket was handed to RCCL,
        ``"waited"`` after the last ``wait()``, ``"adamw_done"`` - so the overlap structure can be asserted, not just claimed."""
        st = self.state

        def mark(what: str, stage):
            if _trace is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream(st.device))
                _trace.append((what, stage, ev))

        if not st.intact():
            raise RuntimeError("the encoder's parameters were re-homed after TrainStep was built (.to() / assign): rebuild it")
        B = x.shape[0]
        L = st.lcfg.layers
        if drop_scales is None and self.drop_path_rate > 0.0:
            drop_scales = sample_drop_scales(L, B, self.drop_path_rate, st.device, self.gen)
        st.repack()
        if self._saved is None or self._saved[0] != B:
            self._saved = (B, st.new_saved(B))
        saved = self._saved[1]
        dt = self._upstream(x)
        taps = st.forward(x, self.taps, drop_scales, saved, _timing)
        works = []
        if self.comm:
            for stage in range(L, -1, -1):                       # one bucket per layer, reduced while the next layers run
                st.backward(x, self.taps, dt, drop_scales, saved, stage, stage, _timing)
                mark("backward_done", stage)
                works.append(self._all_reduce(st.bucket(stage)))
                mark("allreduce_issued", stage)
            for w in works:
                if w is not None:
                    w.wait()
            mark("waited", None)
        else:
            st.backward(x, self.taps, dt, drop_scales, saved, L, 0, _timing)
            mark("backward_done", 0)
        self.steps += 1
        with torch.cuda.device(st.device):
            _lib.check(_lib.load().ldit_adamw_step(st.params.data_ptr(), st.grads.data_ptr(), self.exp_avg.data_ptr(),
                                                   self.exp_avg_sq.data_ptr(), st.numel, self.lr, self.betas[0], self.betas[1],
                                                   self.eps, self.wd, self.steps, 1.0 / self.world, st.packed.data_ptr(),
                                                   torch.cuda.current_stream(st.device).cuda_stream))
        mark("adamw_done", None)
        st._packed_version = st.version()     # the update refreshed the bf16 mirror itself
        st._dirty = False
        # the update went through raw pointers: no tensor version moved, so every cache keyed on versions is stale now
        self.encoder._packed_key = None       # the eval path's packed copy (DiTEncoder._pack)
        self.encoder._pos_cache.clear()       # bicubic resamples of the position table for other grids (_position_table)
        return taps
