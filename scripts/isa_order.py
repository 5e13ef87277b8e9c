#!/usr/bin/env python3
"""Op order of a kernel's code, one character per instruction (hipcc --cuda-device-only -S output):
M MFMA, r ds_read_b128 / b64, t ds_read_b64_tr_b16, w ds_write, D LDS-DMA piece, L other global load, S global store, . other VALU,
[..] s_waitcnt, BAR s_barrier, BR a branch, SCR scratch access.  The view that showed gemm_bf16_tr's reads sunk behind its MFMAs and the
epilogues' per-pass store drains (profiles/r04_tr_pinned_order_ab.txt, r04_epilogue_prefetch_ab.txt); tests/test_isa_audit.py keeps both fixed.
    python scripts/isa_order.py file.s <kernel name substring> [short|long]      (which MFMA loop when the compiler cloned it)"""
import re
import sys


def kernels(asm_text):
    """{mangled name: [instruction lines]} of a device-assembly file."""
    out = {}
    for f in re.split(r"\n(?=_Z\w+:)", asm_text)[1:]:
        name = f.split(":")[0]
        lines = [re.sub(r"\s+", " ", l.split(";")[0]).strip() for l in f.split("s_endpgm")[0].splitlines()]
        out[name] = [l for l in lines if l and (not l.startswith(".") or l.startswith(".L"))]
    return out


def symbol(x, valu=True):
    op = x.split()[0]
    if op.startswith("v_mfma"): return "M"
    if op.startswith("ds_read_b64_tr"): return "t"
    if op.startswith("ds_read"): return "r"
    if op.startswith("ds_write"): return "w"
    if op.startswith("global_load_lds"): return "D"
    if op.startswith(("global_load", "buffer_load")): return "L"
    if op.startswith(("global_store", "buffer_store")): return "S"
    if op.startswith("s_waitcnt"): return " [" + x.split(" ", 1)[1].replace("lgkmcnt", "l").replace("vmcnt", "v") + "] "
    if op.startswith("s_barrier"): return " BAR "
    if op.startswith(("s_cbranch", "s_branch")): return " BR "
    if op.startswith("scratch"): return " SCR "
    return "." if (valu and op.startswith("v_")) else ""


def mfma_loop(lines, pick="long"):
    """Instruction lines of the MFMA loop of a kernel (the longest / shortest backward-branch body with >= 4 MFMAs), or None."""
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    best = None
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\S*\s+(\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = [x for x in lines[labels[m.group(1)]:i + 1] if not x.endswith(":")]
            if sum("v_mfma" in x for x in body) >= 4 and (best is None or (len(body) < len(best) if pick == "short" else len(body) > len(best))):
                best = body
    return best


def order(lines, valu=True):
    return "".join(symbol(x, valu) for x in lines if not x.endswith(":"))


def main():
    s = open(sys.argv[1]).read(); key = sys.argv[2]; pick = sys.argv[3] if len(sys.argv) > 3 else "long"
    for name, lines in kernels(s).items():
        if key not in name: continue
        body = mfma_loop(lines, pick)
        if body is None: continue
        print(name[-70:]); print(order(body))


if __name__ == "__main__":
    main()
