#!/usr/bin/env python3
"""Benchmarks of the hot path on N MI355X GPUs, one process per GPU, batch-sharded replicas (weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1,2,3,4}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`--config` selects a BASELINE.json `configs[i]` workload (default 1 = the headline metric):
    1  ViT-B/16 224x224 bs=64/GPU fp32 forward                       images/sec (BASELINE.json `metric`)
    2  ViT-B/16 224x224 bs=64/GPU bf16 train step: forward + backward + gradient all-reduce (RCCL, N > 1) + fused AdamW
    3  ViT-L/16 512x512 bs=16/GPU bf16 forward (N = 1025 tokens)
    4  ViT-B/16 224x224 bs=32/GPU fp8 forward (bs=256 over 8 GPUs)
    5  (not a BASELINE config) config 1's workload on the split-fp32 build "f32x3" (every GEMM = three bf16-plane products on the
       bf16 MFMA, fp32 everything else)
    6  (not a BASELINE config) config 1's workload on "f32x6" (six plane products: the error of fp32 arithmetic)
(`--model/--size/--batch/--dtype` still override single fields for experiments.)

With `--gpus N > 1` and no WORLD_SIZE in the environment the script launches itself: the parent process (which never
touches the GPU) starts N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's JSON line and
exits with the worst child status.  Under torch.distributed.run it simply joins the job it was started in.

A step = one pass of the hot path over one synthetic batch already resident in HBM.  K steps are timed between
barrier + synchronize brackets (max over ranks); rank 0 prints ONE JSON line.  Extra objects:
  step_ms      - median / p10 / p90 of the per-step durations from HIP events recorded on the launch stream inside the
                 same timed region
  roofline     - the dominant kernel family (the MFMA GEMMs): algorithmic FLOPs / HIP-event time per launch, measured live
                 in a second K-step pass with events on the launch stream, against the dense MFMA peak of the dtype;
                 `traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r0N_traffic.json,
                 labelled with the commit they were collected on) or null
  split_fp32   - (N = 1, config 1 only) the SAME workload on the two split-fp32 builds, measured in the same process right after the
                 headline: images/sec, ms per step, and the relative-L2 distance of every tap from the headline (fp32 MFMA) build's
                 taps.  Reported beside the headline, never as `value`: the headline stays the exact-fp32 MFMA path
  cpu_baseline - (N = 1, config 1 only) the same forward on the host cores: the torch-ops restatement in oracle/ (value)
                 and the plain-C OpenMP restatement beside it, best + median of 5, CPU model and core count
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "fp8": 5000.0,   # dense MFMA peaks, MI355X_MICROARCH.md (never the 2:1-sparse figures)
               "f32x3": 2500.0 / 3, "f32x6": 2500.0 / 6}        # split-fp32 builds: 3 / 6 bf16 MFMA plane products per fp32 product
CONFIGS = {
    1: dict(model="base", size=224, batch=64, dtype="f32", mode="forward"),
    2: dict(model="base", size=224, batch=64, dtype="bf16", mode="train"),
    3: dict(model="large", size=512, batch=16, dtype="bf16", mode="forward"),
    4: dict(model="base", size=224, batch=32, dtype="fp8", mode="forward"),
    5: dict(model="base", size=224, batch=64, dtype="f32x3", mode="forward"),
    6: dict(model="base", size=224, batch=64, dtype="f32x6", mode="forward"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS),
                    help="1..4 = BASELINE.json configs[i]; 5 / 6 = configs[1]'s workload on the f32x3 / f32x6 split-fp32 builds")
    ap.add_argument("--model", default=None, choices=["micro", "tiny", "base", "large"])
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16", "fp8", "f32x3", "f32x6"])
    ap.add_argument("--batch", type=int, default=None, help="images per GPU")
    ap.add_argument("--mode", default=None, choices=["forward", "train"])
    ap.add_argument("--cpu-sample", type=int, default=64, help="images timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-roofline-pass", action="store_true")
    ap.add_argument("--no-split-fp32", action="store_true", help="skip the split-fp32 builds beside the headline line")
    args = ap.parse_args(argv)
    for k, v in CONFIGS[args.config].items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    return args


# ---- self-launch -------------------------------------------------------------------------------------------------------
def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_children(n: int) -> int:
    """Parent of an N-rank job.  Touches no GPU API (a process that has initialised the GPU must never exec, and this one
    only ever starts fresh interpreters): N children with the rendezvous in their environment.  Every child is polled; the
    first non-zero exit terminates the others (a rank that died in start-up would otherwise leave rank 0 in the rendezvous
    until its timeout) and becomes the exit status.  stderr of every rank goes to bench_rank<k>.err under LDIT_BENCH_LOGDIR
    (default: the system temp dir); rank 0's stdout (the JSON line) is relayed."""
    import tempfile
    port = free_port()
    logdir = os.environ.get("LDIT_BENCH_LOGDIR") or tempfile.mkdtemp(prefix="ldit_bench_")
    os.makedirs(logdir, exist_ok=True)
    procs, logs = [], []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # this pool's driver only supports dmabuf IPC; an operator may override
        err = open(os.path.join(logdir, f"bench_rank{rank}.err"), "w")
        logs.append(err)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, stderr=err, text=True))
    import threading
    out_chunks = []
    reader = threading.Thread(target=lambda: out_chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    worst, live = 0, set(range(n))
    while live:
        for k in sorted(live):
            rc = procs[k].poll()
            if rc is None:
                continue
            live.discard(k)
            if rc != 0 and worst == 0:
                worst = rc
                sys.stderr.write(f"bench.py: rank {k} exited with status {rc} (log: {logdir}/bench_rank{k}.err); "
                                 f"stopping the other ranks\n")
                for j in live:
                    procs[j].terminate()          # our own fresh children, by handle - never by pattern
        if live:
            time.sleep(0.05)
    reader.join(timeout=5)
    for err in logs:
        err.close()
    if worst != 0:
        for k in range(n):
            try:
                tail = open(os.path.join(logdir, f"bench_rank{k}.err")).read()[-2000:]
            except OSError:
                tail = ""
            if tail.strip():
                sys.stderr.write(f"---- rank {k} stderr (tail) ----\n{tail}\n")
    sys.stdout.write("".join(out_chunks))
    sys.stdout.flush()
    return worst


# ---- helpers -------------------------------------------------------------------------------------------------------------
def gemm_flops_per_image(cfg, size: int) -> int:
    P = (size // cfg.patch_size) ** 2
    N = P + 1
    C, Fm, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    return 2 * (P * 3 * cfg.patch_size ** 2 * C + L * N * (4 * C * C + 2 * C * Fm))


def pmc_traffic(args):
    """HBM bytes per GEMM launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside this
    process), with the commit they were measured on.  Only for the exact workload they were collected on; else null."""
    key = f"{args.mode}:{args.model}:{args.size}:{args.batch}:{args.dtype}"
    for name in ("r04_traffic.json", "r03_traffic.json"):       # newest committed PMC passes first
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f).get(key)
            if rec:
                return round(rec["gemm_hbm_bytes_per_launch"]), f"profiles/{name}[{key}] profiled@{rec['commit']}"
        except (OSError, KeyError, ValueError):
            pass
    return None, None


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


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, weights, x_np, sample: int):
    """The same forward on the host cores, two ways, each 1 warm-up + 5 timed runs, best and median:
      * the torch-ops restatement (oracle/vit_oracle_torch.py: ATen CPU kernels = what the reference's CPU path executes),
        over `sample` images of the batch - the reported `value`;
      * the plain-C restatement with OpenMP (oracle/libvit_oracle_f32.so), over a quarter of that sample (it is the slower
        of the two; bounded so that the default bench run stays within minutes)."""
    import statistics

    import torch
    from oracle import oracle as c_oracle
    from oracle.vit_oracle_torch import TorchOracle
    cores = host_cores()
    torch.set_num_threads(cores)
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    ora = TorchOracle(cfg, weights)
    xs = torch.from_numpy(x_np[:sample])
    ora.forward(xs[:1])                      # page in / thread pool warm-up (not timed)
    ora.forward(xs)                          # the warm-up run
    t_torch = []
    for _ in range(5):
        t0 = time.perf_counter()
        ora.forward(xs)
        t_torch.append(time.perf_counter() - t0)
    n_c = max(1, sample // 4)
    xc = x_np[:n_c]
    c_oracle.vit_forward(cfg, weights, xc[:1], f32acc=True)
    c_oracle.vit_forward(cfg, weights, xc, f32acc=True)
    t_c = []
    for _ in range(5):
        t0 = time.perf_counter()
        c_oracle.vit_forward(cfg, weights, xc, f32acc=True)
        t_c.append(time.perf_counter() - t0)
    best, med = min(t_torch), statistics.median(t_torch)
    return {"value": round(sample / best, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "median": round(sample / med, 3), "cpu_model": cpu_model(),
            "c_openmp": {"value": round(n_c / min(t_c), 3), "median": round(n_c / statistics.median(t_c), 3), "unit": "images/sec",
                         "cores": cores, "sample": f"{n_c} images, oracle/libvit_oracle_f32.so (plain C + OpenMP, float accumulation), "
                                                   f"1 warm-up + 5 timed runs"},
            "sample": f"{sample} of the 64 images, ViT-B/16 224x224 fp32, torch CPU ops (oracle/vit_oracle_torch.py), "
                      f"{cores} threads (cgroup quota), 1 warm-up + 5 timed runs: value = best ({best:.2f} s per run), "
                      f"median {med:.2f} s"}


def split_fp32_lines(cfg, weights, x, ref_out, steps, warmup, dev):
    """Config 1 on the split-fp32 builds (include/ldit.h LDIT_F32X3 / LDIT_F32X6), timed with HIP events over `steps` forwards."""
    import torch
    from layoutdit_amd.modeling import DiTEncoder
    res = {}
    for dt, products in (("f32x6", 6), ("f32x3", 3)):
        m = DiTEncoder(cfg, compute_dtype=dt).load_numpy(weights).to(dev).eval()
        with torch.no_grad():
            for _ in range(max(warmup, 1)):
                out = m(x)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                out = m(x)
            e1.record()
            torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / steps
        errs = [float((out.hidden_states[t] - ref_out.hidden_states[t]).norm() / ref_out.hidden_states[t].norm()) for t in cfg.taps]
        tf = cfg.flops_per_image(x.shape[2], x.shape[3]) * x.shape[0] / (ms * 1e-3) / 1e12
        res[dt] = {"value": round(x.shape[0] / (ms * 1e-3), 2), "unit": "images/sec", "ms_per_step": round(ms, 4),
                   "plane_products_per_fp32_product": products, "model_tflops_fp32_equivalent": round(tf, 2),
                   "bf16_mfma_roofline_frac": round(tf * products / 2500.0, 4),
                   "rel_l2_vs_headline_taps": [float(f"{e:.3e}") for e in errs]}
        del m
    res["note"] = ("same workload, same process; GEMM operands held as 3 (2) bf16 planes of the fp32 value, products on "
                   "the bf16 MFMA (GEMMs: v_mfma_f32_16x16x32_bf16) with fp32 accumulation (the two products of the attention included); LayerNorm / softmax / "
                   "erf-GELU / residual fp32 as in the headline. Error vs the float64 oracle (tests/test_gpu_split_fp32.py, ViT-B): "
                   "f32x6 6.1-8.4e-7, headline fp32 MFMA build 7.2-9.3e-7, f32x3 4.4-5.6e-6")
    return res


def percentiles(ms):
    import numpy as np
    a = np.sort(np.asarray(ms, dtype=np.float64))
    return {"median": round(float(np.median(a)), 4), "p10": round(float(np.percentile(a, 10)), 4),
            "p90": round(float(np.percentile(a, 90)), 4), "source": "HIP events on the launch stream, one pair per step"}


def per_rank_report(rows):
    """rows[k] = [images/s, GEMM-family roofline fraction (nan when the roofline pass was skipped), ms per step] of rank k, each from
    that rank's OWN clock and HIP events -> the "MFMA util % at 1/2/4/8 GPU" half of BASELINE.json's metric: min / mean / max over
    the ranks and the per-rank lists, so a slow rank shows (the headline `value` only sees it through the max-over-ranks time)."""
    import math

    def stats(col):
        v = [row[col] for row in rows if not math.isnan(row[col])]
        if not v:
            return None
        return {"min": round(min(v), 4), "mean": round(sum(v) / len(v), 4), "max": round(max(v), 4), "per_rank": [round(x, 4) for x in v]}
    return {"images_per_sec": stats(0), "gemm_mfma_roofline_frac": stats(1), "ms_per_step": stats(2), "ranks": len(rows),
            "source": "each rank's own wall clock between the barriers and its own HIP-event GEMM time"}


# ---- one rank ----------------------------------------------------------------------------------------------------------------
def run_rank(args) -> None:
    import torch
    from layoutdit_amd import config as cfgs, dp, synth
    from layoutdit_amd.modeling import DiTEncoder

    if os.environ.get("LDIT_BENCH_DRYRUN") == "1":
        # launcher / rendezvous rehearsal without a GPU (tests/test_dp_gloo.py): join over gloo, reduce, report
        if os.environ.get("LDIT_BENCH_DRYRUN_FAIL_RANK") == os.environ.get("RANK", "0"):
            sys.stderr.write("dry run: this rank fails in start-up on request\n")
            raise SystemExit(3)
        r = dp.init(backend="gloo")
        dp.barrier(r)
        worst = dp.max_over_ranks(r, float(r.rank + 1))
        # the per-rank report of the real run, on made-up numbers: rank k "measured" 100 (k + 1) images/s at MFMA fraction 0.1 (k + 1)
        per = per_rank_report(dp.gather_over_ranks(r, [100.0 * (r.rank + 1), 0.1 * (r.rank + 1), 5.0 + r.rank]))
        if r.is_main:
            print(json.dumps({"dryrun": True, "n_gpus": r.world, "max_over_ranks": worst, "gpus_arg": args.gpus, "per_rank": per}), flush=True)
        dp.finalize(r)
        return
    r = dp.init()
    if r.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={r.world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: layoutdit_amd has no CPU path")
    dev = torch.device("cuda", r.local_rank)
    torch.cuda.set_device(dev)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(r.world)))
    placement = dp.pin_to_gpu_numa(r.local_rank, local_world)           # one process per GPU, on its NUMA node

    cfg = cfgs.GEOMETRIES[args.model]()
    weights = synth.synth_weights(cfg, seed=0)
    train = args.mode == "train"
    model = DiTEncoder(cfg, compute_dtype=args.dtype).load_numpy(weights).to(dev)
    lo, hi = dp.shard_range(args.batch * r.world, r.rank, r.world)       # weak scaling: args.batch images per rank
    x_np = synth.synth_images(hi - lo, args.size, args.size, seed=1234, first_index=lo)
    x = torch.from_numpy(x_np).to(dev)                                    # resident in HBM before the timed region
    if train:
        from layoutdit_amd.training import TrainStep
        model.train()
        stepper = TrainStep(model, r, lr=1e-4, weight_decay=0.0, seed=4321)   # trainer.py:62-68: AdamW(lr 1e-4, wd 0)
        run = lambda: stepper.step(x)                                          # noqa: E731
    else:
        model.eval()
        if args.dtype == "fp8":
            model.calibrate_fp8(x)                                        # per-tensor activation scales, untimed set-up
        run = lambda: model(x)                                            # noqa: E731

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    with torch.set_grad_enabled(False):
        for _ in range(max(args.warmup, 1)):
            out = run()
        torch.cuda.synchronize(dev)
        dp.barrier(r)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(args.steps):
            out = run()
            ev[i + 1].record()
        torch.cuda.synchronize(dev)
        dp.barrier(r)
        t1 = time.perf_counter()
    elapsed = dp.max_over_ranks(r, t1 - t0)
    step_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    if train:
        assert bool(torch.isfinite(stepper.flat_params).all())
    else:
        assert all(torch.isfinite(h).all() for h in out.hidden_states if h is not None)

    # second pass: per-kernel HIP events on the launch stream (same inputs, same K steps)
    timing: dict = {}
    if not args.no_roofline_pass:
        with torch.no_grad():
            for _ in range(args.steps):
                if train:
                    stepper.step(x, _timing=timing)
                else:
                    model(x, _timing=timing)
        torch.cuda.synchronize(dev)

    # every rank: its own throughput (own clock) and its own GEMM-family roofline fraction (own HIP events), gathered to rank 0
    mult_r = 3 if train else 1
    own_frac = float("nan")
    if timing and timing.get("gemm_ms"):
        own_frac = (gemm_flops_per_image(cfg, args.size) * mult_r * args.batch * args.steps / (timing["gemm_ms"] * 1e-3) / 1e12
                    / PEAK_TFLOPS[args.dtype])
    per_rank = dp.gather_over_ranks(r, [args.batch * args.steps / (t1 - t0), own_frac, 1e3 * (t1 - t0) / args.steps])

    if r.is_main:
        headline = (args.model, args.size, args.batch, args.dtype, args.mode) == ("base", 224, 64, "f32", "forward")
        images = args.batch * r.world * args.steps
        ms_per_step = 1e3 * elapsed / args.steps
        what = "fwd" if not train else "train step (fwd+bwd+AdamW)"
        line = {
            "metric": "images/sec ViT-B/16 224px bs=64 fwd" if headline
            else f"images/sec ViT-{args.model}/16 {args.size}px bs={args.batch} {args.dtype} {what}",
            "value": round(images / elapsed, 2), "unit": "images/sec", "n_gpus": r.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"ViT-{args.model}/16 {args.size}x{args.size} bs={args.batch} {args.dtype} "
                                   f"{'forward' if not train else 'train step'}, taps {cfg.taps} "
                                   + (f"(BASELINE.json configs[{args.config}])" if args.config <= 4 else
                                      f"(BASELINE.json configs[1] workload on the {args.dtype} split-fp32 build - not a BASELINE config)"),
                       "images_per_gpu": args.batch, "global_batch": args.batch * r.world,
                       "parallelism": f"dp{r.world}: batch-sharded replicas, "
                                      + ("bucketed gradient all-reduce (one bucket per layer) overlapped with backward"
                                         if train else "no data-path collective"),
                       "weights": "synthetic seed 0", "images": "synthetic doc-like pages, seed 1234",
                       "rank0_placement": placement},
            "step_ms": percentiles(step_ms),
        }
        mult = 3 if train else 1                       # train step ~ 3x the forward's matmul FLOPs
        flops_img = cfg.flops_per_image(args.size, args.size) * mult
        line["model_tflops"] = round(flops_img * args.batch * r.world / (ms_per_step * 1e-3) / 1e12, 2)
        peak = PEAK_TFLOPS[args.dtype]
        line["model_mfma_roofline_frac"] = round(line["model_tflops"] / (peak * r.world), 4)
        if timing and timing.get("gemm_launches"):
            n = int(timing["gemm_launches"])
            gemm_flops = gemm_flops_per_image(cfg, args.size) * mult * args.batch * args.steps
            achieved = gemm_flops / (timing["gemm_ms"] * 1e-3) / 1e12
            traffic, source = pmc_traffic(args)
            line["roofline"] = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                                "traffic_source": source,
                                "kernel": "fp32 MFMA GEMM family (patch-embed, qkv, o_proj, fc1, fc2)" if args.dtype == "f32"
                                else f"bf16 MFMA GEMM family on split fp32 operands ({args.dtype}: peak = 2.5 PF / plane products per fp32 "
                                     f"product; patch-embed on plane products too)" if args.dtype.startswith("f32x") else f"{args.dtype} MFMA GEMM family (qkv, o_proj, fc1, fc2"
                                     + (" + their dgrad / wgrad" if train else "") + "; patch-embed = bf16 im2col pass + bf16 MFMA GEMM)",
                                "launches": n, "avg_launch_ms": round(timing["gemm_ms"] / n, 5),
                                "flops_per_launch": gemm_flops // n}
            line["kernel_ms_per_step"] = {k[:-3]: round(v / args.steps, 4) for k, v in timing.items() if k.endswith("_ms")}
        line["per_rank"] = per_rank_report(per_rank)
        if r.world == 1 and headline and not args.no_split_fp32:
            line["split_fp32"] = split_fp32_lines(cfg, weights, x, out, args.steps, args.warmup, dev)
        if r.world == 1 and args.cpu_sample > 0 and headline:
            line["cpu_baseline"] = cpu_baseline(cfg, weights, x_np, min(args.cpu_sample, args.batch))
        print(json.dumps(line), flush=True)
    dp.finalize(r)


def main() -> None:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_children(args.gpus))
    run_rank(args)


if __name__ == "__main__":
    main()
