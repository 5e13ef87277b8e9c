#!/usr/bin/env python3
"""Real-weights parity hook (last bullet): given a LOCAL checkpoint of the DiT / BEiT encoder, report how every
build of the MI355X path (``f32``, ``f32x6``, ``f32x3``, ``bf16``, ``fp8``) compares with the CPU oracle on synthetic document pages.

    python scripts/parity_checkpoint.py --checkpoint /path/to/pytorch_model.bin [--heads 12] [--batch 2] [--size 224] [--builds f32,bf16]

Pretrained BEiT-family checkpoints carry activation outliers that the synthetic weights of the test suite do not reproduce, so the
low-precision gates validated there (bf16 2e-2, fp8 1e-1 + cos >= 0.995, f32x3 inside the fp32 gates) are optimistic until this has
been run on the checkpoint that will be served.  Nothing is downloaded: ``microsoft/dit-base`` (ref
src/layoutdit/modeling/dit_backbone.py:25-31) must already be a file on disk.

* The file is read with loaders that execute nothing from it: ``torch.load(..., weights_only=True)`` or safetensors.
* Keys: transformers 4.49 (what the reference pins), 5.x, and the detector-prefixed spellings the reference's own saver writes
  (ref src/layoutdit/modeling/model.py:90-121) - ``layoutdit_amd.modeling.keys``.
* The geometry (hidden size, layers, MLP width, patch size, position-table grid) is read off the tensor shapes; the head count is
  not recoverable from shapes (``--heads``, default hidden / 64 as in every BEiT / DiT release).
* The oracle is test infrastructure (this file lives under tests/ for that reason; scripts/parity_checkpoint.py only launches it):
  ``oracle/vit_oracle.c`` with double accumulation.  The GPU builds run through the C ABI as everywhere else.

Per build and tap: relative L2, worst element relative to max(|ref|, 1), cosine; a PASS / FAIL against the build's gate
(f32 / f32x6 / f32x3: the north-star's 1e-3 in both norms; bf16 2e-2 rel-L2; fp8 1e-1 rel-L2 and cosine >= 0.995).  Exit status 1
if any requested build fails its gate."""
from __future__ import annotations

import argparse
import json
import os
import sys
from typing import Dict, Tuple

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from layoutdit_amd import synth                                  # noqa: E402
from layoutdit_amd.config import DiTConfig                       # noqa: E402
from layoutdit_amd.modeling.keys import to_v4                    # noqa: E402

GATES = {"f32": (1e-3, 1e-3, None), "f32x6": (1e-3, 1e-3, None), "f32x3": (1e-3, 1e-3, None),
         "bf16": (2e-2, None, None), "fp8": (1e-1, None, 0.995)}       # (rel-L2, worst element, cosine)


def read_state_dict(path: str) -> Dict[str, np.ndarray]:
    """fp32 numpy arrays keyed by transformers-4.49 BEiT names.  Never unpickles arbitrary objects."""
    if path.endswith(".safetensors"):
        from safetensors.numpy import load_file
        raw = dict(load_file(path))
    else:
        import torch
        obj = torch.load(path, map_location="cpu", weights_only=True)
        for wrap in ("state_dict", "model", "module"):                      # common containers around the tensor dict
            if isinstance(obj, dict) and wrap in obj and isinstance(obj[wrap], dict):
                obj = obj[wrap]
        raw = {k: v.detach().to(torch.float32).numpy() for k, v in obj.items() if hasattr(v, "detach")}
    out = {}
    for k, v in raw.items():
        out[to_v4(k)] = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
    return out


def derive_config(sd: Dict[str, np.ndarray], heads: int = 0) -> DiTConfig:
    """The encoder geometry from the tensor shapes of a (4.49-keyed) state dict."""
    try:
        pw = sd["embeddings.patch_embeddings.projection.weight"]
        pos = sd["embeddings.position_embeddings"]
    except KeyError as e:
        raise SystemExit(f"not a BEiT / DiT encoder checkpoint: {e.args[0]} is missing (keys seen: {sorted(sd)[:5]} ...)")
    hidden, in_ch, patch = int(pw.shape[0]), int(pw.shape[1]), int(pw.shape[2])
    layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layer."))
    mlp = int(sd["encoder.layer.0.intermediate.dense.weight"].shape[0])
    g0 = int(round((pos.shape[1] - 1) ** 0.5))
    if g0 * g0 + 1 != pos.shape[1]:
        raise SystemExit(f"position table of {pos.shape[1]} rows is not 1 + a square grid")
    heads = heads or max(1, hidden // 64)
    if hidden % heads:
        raise SystemExit(f"--heads {heads} does not divide the hidden size {hidden}")
    return DiTConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=mlp,
                     patch_size=patch, image_size=g0 * patch, num_channels=in_ch)


def load_checkpoint(path: str, heads: int = 0) -> Tuple[DiTConfig, Dict[str, np.ndarray], Dict[str, list]]:
    """-> (config, the tensors the path reads, {"missing": [...], "unexpected": [...]}).  The inert tensors (mask token, pooler
    LayerNorm) may be absent; relative-position-bias tables mean the checkpoint is a BEiT variant this path does not implement."""
    sd = read_state_dict(path)
    cfg = derive_config(sd, heads)
    shapes = synth.param_shapes(cfg)
    inert = {"embeddings.mask_token", "pooler.layernorm.weight", "pooler.layernorm.bias"}
    missing = [k for k in shapes if k not in sd and k not in inert]
    unexpected = [k for k in sd if k not in shapes]
    if any("relative_position" in k for k in unexpected):
        raise SystemExit("the checkpoint carries relative-position-bias tables: not a DiT (absolute position embedding) encoder")
    if missing:
        raise SystemExit(f"checkpoint lacks {len(missing)} tensors of the path, e.g. {missing[:3]}")
    weights = {}
    for k, shape in shapes.items():
        if k in sd:
            if tuple(sd[k].shape) != tuple(shape):
                raise SystemExit(f"{k}: shape {tuple(sd[k].shape)} in the file, {tuple(shape)} expected")
            weights[k] = sd[k]
        else:
            weights[k] = np.zeros(shape, np.float32)             # inert, never read by the forward
    return cfg, weights, {"missing": [k for k in shapes if k not in sd], "unexpected": unexpected}


def compare(got: np.ndarray, ref: np.ndarray) -> Dict[str, float]:
    g, r = got.astype(np.float64).ravel(), ref.astype(np.float64).ravel()
    d = g - r
    return {"rel_l2": float(np.linalg.norm(d) / max(np.linalg.norm(r), 1e-300)),
            "worst_elem": float(np.max(np.abs(d) / np.maximum(np.abs(r), 1.0))),
            "cosine": float(np.dot(g, r) / max(np.linalg.norm(g) * np.linalg.norm(r), 1e-300))}


def verdict(build: str, rows) -> bool:
    l2, elem, cos = GATES[build]
    ok = all(r["rel_l2"] <= l2 for r in rows)
    if elem is not None:
        ok = ok and all(r["worst_elem"] <= elem for r in rows)
    if cos is not None:
        ok = ok and all(r["cosine"] >= cos for r in rows)
    return ok


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--checkpoint", required=True, help="local file: torch state_dict (.bin / .pth / .pt) or .safetensors")
    ap.add_argument("--heads", type=int, default=0, help="attention heads (default hidden / 64)")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=0, help="square input size (default: the position table's own grid)")
    ap.add_argument("--builds", default="f32,f32x6,f32x3,bf16,fp8")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--json", default=None, help="also write the report to this file")
    args = ap.parse_args(argv)

    cfg, weights, keys = load_checkpoint(args.checkpoint, args.heads)
    size = args.size or cfg.image_size
    print(f"checkpoint {args.checkpoint}: hidden {cfg.hidden_size}, {cfg.num_hidden_layers} layers, {cfg.num_attention_heads} heads, "
          f"mlp {cfg.intermediate_size}, patch {cfg.patch_size}, table grid {cfg.image_size // cfg.patch_size}; taps {cfg.taps}; "
          f"{len(keys['unexpected'])} tensors ignored" + (f" (e.g. {keys['unexpected'][:2]})" if keys["unexpected"] else ""))
    x = synth.synth_images(args.batch, size, size, seed=args.seed)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("parity_checkpoint needs the GPU: layoutdit_amd has no CPU path (the oracle is only the checker)")
    from layoutdit_amd.modeling import DiTEncoder
    from layoutdit_amd.modeling.dit_encoder import resample_position_table
    from oracle import oracle

    pos = None
    g0 = cfg.image_size // cfg.patch_size
    if size // cfg.patch_size != g0:
        pos = resample_position_table(torch.from_numpy(weights["embeddings.position_embeddings"]), g0, size // cfg.patch_size,
                                      size // cfg.patch_size).numpy()
    ref, _ = oracle.vit_forward(cfg, weights, x, pos=pos)
    report = {"checkpoint": os.path.abspath(args.checkpoint), "geometry": {"hidden": cfg.hidden_size, "layers": cfg.num_hidden_layers,
              "heads": cfg.num_attention_heads, "mlp": cfg.intermediate_size, "patch": cfg.patch_size}, "input": [args.batch, 3, size, size],
              "taps": list(cfg.taps), "builds": {}}
    xd = torch.from_numpy(x).to("cuda:0")
    failed = []
    for build in [b for b in args.builds.split(",") if b]:
        try:
            m = DiTEncoder(cfg, compute_dtype=build).load_numpy(weights).to("cuda:0").eval()
            with torch.no_grad():
                if build == "fp8":          # activation ranges from ANOTHER batch of pages, as a deployment would
                    m.calibrate_fp8(torch.from_numpy(synth.synth_images(args.batch, size, size, seed=args.seed + 1)).to("cuda:0"))
                out = m(xd).hidden_states
            torch.cuda.synchronize()
        except Exception as e:              # noqa: BLE001 - a geometry a build does not cover is a finding, not a crash
            print(f"{build:6s}  not available for this geometry: {e}")
            report["builds"][build] = {"unavailable": str(e)}
            continue
        rows = [compare(out[t].cpu().numpy(), r) for t, r in zip(cfg.taps, ref)]
        ok = verdict(build, rows)
        report["builds"][build] = {"per_tap": rows, "pass": ok, "gate": dict(zip(("rel_l2", "worst_elem", "cosine"), GATES[build]))}
        for t, r in zip(cfg.taps, rows):
            print(f"{build:6s} tap {t:2d}: rel-L2 {r['rel_l2']:.3e}   worst element {r['worst_elem']:.3e}   cosine {r['cosine']:.6f}")
        print(f"{build:6s} {'PASS' if ok else 'FAIL'} against gate {report['builds'][build]['gate']}")
        if not ok:
            failed.append(build)
        del m
    if args.json:
        with open(args.json, "w") as f:
            json.dump(report, f, indent=1)
    return 1 if failed else 0


if __name__ == "__main__":
    raise SystemExit(main())
