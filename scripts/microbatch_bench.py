#!/usr/bin/env python3
"""Does splitting the batch into independent micro-batches on separate HIP streams fill the kernel-boundary bubbles?

    python scripts/microbatch_bench.py --config 1|3|4 [--streams 2] [--steps 20]

One DiTEncoder per stream (same parameters, its own workspace), each fed a contiguous slice of the batch; fork / join by
events on the caller's stream.  Images are independent and every kernel is batch invariant, so the taps are bit-identical
to the one-stream forward (checked here).  Prints ms per batch, interleaved A/B.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CONFIGS  # noqa: E402
from layoutdit_amd import config as cfgs, synth  # noqa: E402
from layoutdit_amd.modeling import DiTEncoder  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    c = CONFIGS[a.config]
    dev = torch.device("cuda", 0)
    cfg = cfgs.GEOMETRIES[c["model"]]()
    w = synth.synth_weights(cfg, seed=0)
    x = torch.from_numpy(synth.synth_images(c["batch"], c["size"], c["size"], seed=1234)).to(dev)
    models = [DiTEncoder(cfg, compute_dtype=c["dtype"]).load_numpy(w).to(dev).eval() for _ in range(a.streams + 1)]
    if c["dtype"] == "fp8":
        models[0].calibrate_fp8(x)
        for m in models[1:]:
            m.fp8_act_scales.copy_(models[0].fp8_act_scales)
    streams = [torch.cuda.Stream(dev) for _ in range(a.streams)]
    B = c["batch"]
    cuts = [B * i // a.streams for i in range(a.streams + 1)]

    def one():
        return models[0](x)

    def multi():
        cur = torch.cuda.current_stream(dev)
        fork = torch.cuda.Event()
        fork.record(cur)
        outs = []
        for i, s in enumerate(streams):
            s.wait_event(fork)
            with torch.cuda.stream(s):
                outs.append(models[1 + i](x[cuts[i]:cuts[i + 1]]))
            e = torch.cuda.Event()
            e.record(s)
            cur.wait_event(e)
        return outs

    with torch.no_grad():
        ref = one()
        got = multi()
        torch.cuda.synchronize()
        for t in cfg.taps:
            cat = torch.cat([o.hidden_states[t] for o in got])
            assert torch.equal(cat, ref.hidden_states[t]), f"tap {t} differs"
        print("taps bit-identical to the one-stream forward")

        def timeit(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.steps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / a.steps

        for r in range(a.rounds):
            t1 = timeit(one)
            t2 = timeit(multi)
            print(f"config {a.config} round {r}: one stream {t1:.3f} ms   {a.streams} streams {t2:.3f} ms   ({100 * (t1 - t2) / t1:+.1f} %)",
                  flush=True)


if __name__ == "__main__":
    main()
