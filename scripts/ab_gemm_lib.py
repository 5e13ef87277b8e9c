#!/usr/bin/env python3
"""Same-process, interleaved A/B of the bf16 GEMM of two builds of the library (cdna guide 5.4 rule 24): the shipped
``layoutdit_amd/libldit_hip.so`` against an alternative .so given by path (built HERE with extra -D flags; ``csrc/build/`` travels to
the GPU box), both called through the C ABI on the same random bf16 operands, with torch's ``mm`` (hipBLASLt) as the known-good
reference beside them.  Every result is checked against a float64 product of the bf16 operands on a row sample first.

    python scripts/ab_gemm_lib.py layoutdit_amd/csrc/build/libldit_alt.so [shapes=vitl|vitb|x3|square|all] [EPI=bias|resid|gelu]
"""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdit_amd import _lib  # noqa: E402

SHAPES = {
    "vitl": [("ViT-L qkv", 16384, 3072, 1024), ("ViT-L o_proj", 16384, 1024, 1024), ("ViT-L fc1", 16384, 4096, 1024), ("ViT-L fc2", 16384, 1024, 4096)],
    "vitb": [("ViT-B qkv", 12608, 2304, 768), ("ViT-B o_proj", 12608, 768, 768), ("ViT-B fc1", 12608, 3072, 768), ("ViT-B fc2", 12608, 768, 3072)],
    "x3": [("x3 qkv K'=2304", 12608, 2304, 2304), ("x3 fc2 K'=9216", 12608, 768, 9216)],
    "square": [("square", 4096, 4096, 4096), ("square", 8192, 8192, 8192)],
}


def bind(path):
    lib = C.CDLL(path)
    for name in ("ldit_linear_bf16", "ldit_last_error", "ldit_abi_version"):
        res, args = _lib.SIGNATURES[name]
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    assert lib.ldit_abi_version() == _lib.LDIT_ABI_VERSION
    return lib


def main():
    alt_paths = [a for a in sys.argv[1:] if a.endswith(".so")]
    rest = [a for a in sys.argv[1:] if not a.endswith(".so")]
    which = rest[0] if rest else "all"
    epi_name = os.environ.get("EPI", "bias")
    epi = {"bias": _lib.EPI_BIAS, "resid": _lib.EPI_SCALE_RESID, "gelu": _lib.EPI_BIAS_GELU}[epi_name]
    libs = [("shipped", bind(_lib.LIB_PATH))] + [(os.path.basename(q)[len("libldit_"):-3], bind(q)) for q in alt_paths]
    shapes = sum(SHAPES.values(), []) if which == "all" else SHAPES[which]
    stream = torch.cuda.current_stream().cuda_stream

    def t(fn, n=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    for name, m, n, k in shapes:
        x = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16)
        w = (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16)
        b = torch.randn(n, device="cuda")
        lam = torch.rand(n, device="cuda")
        r0 = torch.randn(m, n, device="cuda")
        out_dt = torch.float32 if epi == _lib.EPI_SCALE_RESID else torch.bfloat16
        outs = {}

        def call(lib, out):
            rc = lib.ldit_linear_bf16(x.data_ptr(), k, w.data_ptr(), b.data_ptr(), out.data_ptr(), n, m, n, k, epi,
                                      lam.data_ptr() if epi == _lib.EPI_SCALE_RESID else None,
                                      r0.data_ptr() if epi == _lib.EPI_SCALE_RESID else None, None, stream)
            if rc != 0:
                raise RuntimeError(lib.ldit_last_error().decode())

        rows = torch.arange(0, m, max(1, m // 97), device="cuda")
        ref = x[rows].double() @ w.double().t() + b.double()
        if epi == _lib.EPI_SCALE_RESID:
            ref = r0[rows].double() + lam.double() * ref
        for tag, lib in libs:
            out = torch.zeros(m, n, device="cuda", dtype=out_dt)
            call(lib, out)
            torch.cuda.synchronize()
            outs[tag] = out
            if epi != _lib.EPI_BIAS_GELU:
                err = float((out[rows].double() - ref).norm() / ref.norm())
                assert err < (1e-5 if out_dt == torch.float32 else 6e-3), (name, tag, err)
        same = {tag: bool(torch.equal(outs["shipped"], outs[tag])) for tag, _ in libs[1:]}
        scratch = torch.empty(m, n, device="cuda", dtype=out_dt)
        vend = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        times = {tag: [] for tag, _ in libs}
        times["vendor"] = []
        for _ in range(5):
            for tag, lib in libs:
                times[tag].append(t(lambda: call(lib, scratch)))
            times["vendor"].append(t(lambda: torch.mm(x, w.t(), out=vend)))
        med = {q: statistics.median(v_) for q, v_ in times.items()}
        fl = 2.0 * m * n * k
        v = med["vendor"]
        txt = " | ".join(f"{tag} {med[tag]:7.1f} us {fl / med[tag] / 1e6:6.0f} TF ({v / med[tag]:.2f} of vendor{'' if tag == 'shipped' else ', bits ' + ('=' if same[tag] else 'DIFFER')})"
                         for tag, _ in libs)
        print(f"{name:16s} M={m:6d} N={n:5d} K={k:5d} [{epi_name}]: {txt} | vendor {v:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
