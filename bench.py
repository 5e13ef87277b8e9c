#!/usr/bin/env python3
"""Headline benchmark: images/sec of the ViT-B/16 224x224 bs=64 fp32 encoder forward (BASELINE.json metric/config 2)
on N MI355X GPUs, one process per GPU, batch-sharded replicas (weak scaling: 64 images per GPU, no collective on the
data path).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one forward of the hot path (``self.dit(x).hidden_states`` -> taps 4/6/8/12) over one 64-image synthetic
batch already resident in HBM.  Rank 0 prints ONE JSON line.  Extra objects:
  roofline     - the dominant kernel family (the fp32 MFMA GEMM): algorithmic FLOPs / HIP-event time per launch,
                 measured live in a second K-step pass with events on the launch stream, against the 157.3 TFLOP/s
                 fp32-matrix peak of gfx950 (MI355X_MICROARCH.md).
  cpu_baseline - (N = 1 only) the same forward on the host cores: the torch-ops restatement in oracle/ (the ATen CPU
                 kernels the reference's HF BeitModel path executes), on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

from layoutdit_amd import config as cfgs, dp, synth           # noqa: E402
from layoutdit_amd.modeling import DiTEncoder                 # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3    # gfx950 dense fp32 matrix peak (spec; 155 measured), MI355X_MICROARCH.md
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "fp8": 5000.0}   # dense MFMA peaks (bf16: ~2.5 PF dense, never the 2:1-sparse figure)
PER_GPU_BATCH = 64


def gemm_flops_per_image(cfg, size: int) -> int:
    P = (size // cfg.patch_size) ** 2
    N = P + 1
    C, Fm, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    return 2 * (P * 3 * cfg.patch_size ** 2 * C + L * N * (4 * C * C + 2 * C * Fm))


def pmc_traffic(args):
    """HBM bytes per GEMM launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside this
    process).  Only valid for the workload the counters were collected on; otherwise null."""
    if (args.model, args.size, args.batch, args.dtype) != ("base", 224, 64, "f32"):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_final_traffic.json")) as f:
            return round(json.load(f)["gemm_hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def host_cores() -> int:
    """Cores this process may actually use: min(affinity mask, cgroup CPU quota).  The GPU box exposes 256 hardware
    threads but grants a 16-CPU quota per GPU; running 256 threads against that quota thrashes (0.5 img/s)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(cfg, weights, x_np, sample: int):
    """Time the torch-ops restatement on the host cores over `sample` images of the same batch."""
    from oracle.vit_oracle_torch import TorchOracle
    cores = host_cores()
    torch.set_num_threads(cores)
    ora = TorchOracle(cfg, weights)
    xs = torch.from_numpy(x_np[:sample])
    ora.forward(xs[:1])                      # page in / thread pool warm-up (not timed)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        ora.forward(xs)
        times.append(time.perf_counter() - t0)
    best = min(times)
    return {"value": round(sample / best, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{sample} of the 64 images, ViT-B/16 224x224 fp32, torch CPU ops (oracle/vit_oracle_torch.py), "
                      f"{cores} threads (cgroup quota), best of 3 runs, {best:.2f} s per run"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="base", choices=sorted(cfgs.GEOMETRIES))
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "fp8"],
                    help="f32 = BASELINE configs[1] (headline); bf16 with --model large --size 512 --batch 16 = configs[3]; "
                         "fp8 with --batch 32 = configs[4] (per-GPU share of bs=256 over 8 GPUs)")
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="images per GPU")
    ap.add_argument("--cpu-sample", type=int, default=64, help="images timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-roofline-pass", action="store_true")
    args = ap.parse_args()

    r = dp.init()
    if r.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={r.world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: layoutdit_amd has no CPU path")
    dev = torch.device("cuda", r.local_rank)
    torch.cuda.set_device(dev)

    cfg = cfgs.GEOMETRIES[args.model]()
    weights = synth.synth_weights(cfg, seed=0)
    model = DiTEncoder(cfg, compute_dtype=args.dtype).load_numpy(weights).to(dev).eval()
    lo, hi = dp.shard_range(args.batch * r.world, r.rank, r.world)       # weak scaling: args.batch images per rank
    x_np = synth.synth_images(hi - lo, args.size, args.size, seed=1234, first_index=lo)
    x = torch.from_numpy(x_np).to(dev)                                    # resident in HBM before the timed region
    if args.dtype == "fp8":
        model.calibrate_fp8(x)                                            # per-tensor activation scales, untimed set-up

    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            out = model(x)
        torch.cuda.synchronize(dev)
        dp.barrier(r)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = model(x)
        torch.cuda.synchronize(dev)
        dp.barrier(r)
        t1 = time.perf_counter()
    elapsed = dp.max_over_ranks(r, t1 - t0)
    assert all(torch.isfinite(h).all() for h in out.hidden_states if h is not None)

    # second pass: per-kernel HIP events on the launch stream (same inputs, same K steps)
    timing: dict = {}
    if not args.no_roofline_pass:
        with torch.no_grad():
            for _ in range(args.steps):
                model(x, _timing=timing)
        torch.cuda.synchronize(dev)

    if r.is_main:
        images = args.batch * r.world * args.steps
        ms_per_step = 1e3 * elapsed / args.steps
        line = {
            "metric": "images/sec ViT-B/16 224px bs=64 fwd"
            if (args.model, args.size, args.batch, args.dtype) == ("base", 224, 64, "f32")
            else f"images/sec ViT-{args.model}/16 {args.size}px bs={args.batch} {args.dtype} fwd",
            "value": round(images / elapsed, 2), "unit": "images/sec", "n_gpus": r.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"ViT-{args.model}/16 {args.size}x{args.size} bs={args.batch} {args.dtype} forward, taps "
                                   f"{cfg.taps}" + (" (BASELINE.json configs[1])"
                                                    if (args.model, args.size, args.batch, args.dtype) == ("base", 224, 64, "f32")
                                                    else ""),
                       "images_per_gpu": args.batch, "global_batch": args.batch * r.world,
                       "parallelism": f"dp{r.world}: batch-sharded replicas, no data-path collective",
                       "weights": "synthetic seed 0", "images": "synthetic doc-like pages, seed 1234"},
        }
        flops_img = cfg.flops_per_image(args.size, args.size)
        line["model_tflops"] = round(flops_img * args.batch * r.world / (ms_per_step * 1e-3) / 1e12, 2)
        peak = PEAK_TFLOPS[args.dtype]
        line["model_mfma_roofline_frac"] = round(line["model_tflops"] / (peak * r.world), 4)
        if timing:
            n = int(timing["gemm_launches"])
            gemm_flops = gemm_flops_per_image(cfg, args.size) * args.batch * args.steps
            achieved = gemm_flops / (timing["gemm_ms"] * 1e-3) / 1e12
            line["roofline"] = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                                "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                                "traffic": pmc_traffic(args),
                                "kernel": "fp32 MFMA GEMM family (patch-embed, qkv, o_proj, fc1, fc2)" if args.dtype == "f32"
                                else f"{args.dtype} MFMA GEMM family (qkv, o_proj, fc1, fc2; patch-embed stays fp32)",
                                "launches": n, "avg_launch_ms": round(timing["gemm_ms"] / n, 5),
                                "flops_per_launch": gemm_flops // n}
            line["kernel_ms_per_step"] = {k[:-3]: round(v / args.steps, 4) for k, v in timing.items() if k.endswith("_ms")}
        if r.world == 1 and args.cpu_sample > 0 and args.model == "base" and args.dtype == "f32":
            line["cpu_baseline"] = cpu_baseline(cfg, weights, x_np, min(args.cpu_sample, args.batch))
        print(json.dumps(line), flush=True)
    dp.finalize(r)


if __name__ == "__main__":
    main()
