#!/usr/bin/env python3
"""Op order of the innermost MFMA loop of a kernel, one character per instruction (hipcc --cuda-device-only -S output):
M MFMA, r ds_read_b128 / b64, t ds_read_b64_tr_b16, w ds_write, D LDS-DMA piece, g other global load, . other VALU, [..] s_waitcnt,
BAR s_barrier, SCR scratch access.  The view that showed gemm_bf16_tr's reads sunk behind its MFMAs (profiles/r04_tr_pinned_order_ab.txt).
    python scripts/isa_order.py file.s <kernel name substring> [short|long]      (which loop when the compiler cloned it)"""
import re, sys
s = open(sys.argv[1]).read(); key = sys.argv[2]; pick = sys.argv[3] if len(sys.argv) > 3 else "long"
for f in re.split(r"\n(?=_Z\w+:)", s)[1:]:
    name = f.split(":")[0]
    if key not in name: continue
    lines = [re.sub(r"\s+", " ", l.split(";")[0]).strip() for l in f.split("s_endpgm")[0].splitlines()]
    lines = [l for l in lines if l and not l.startswith(".") or l.startswith(".L")]
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    best = None
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\S*\s+(\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = lines[labels[m.group(1)]:i + 1]
            nm = sum("v_mfma" in x for x in body)
            if nm >= 4 and (best is None or (len(body) < len(best) if pick == "short" else len(body) > len(best))): best = body
    if best is None: continue
    out = []
    for x in best:
        op = x.split()[0]
        if op.startswith("v_mfma"): t = "M"
        elif op.startswith("ds_read_b64_tr"): t = "t"
        elif op.startswith("ds_read"): t = "r"
        elif op.startswith("ds_write"): t = "w"
        elif op.startswith("global_load_lds"): t = "D"
        elif op.startswith(("global_load", "buffer_load")): t = "g"
        elif op.startswith("s_waitcnt"): t = " [" + x.split(" ", 1)[1].replace("lgkmcnt", "l").replace("vmcnt", "v") + "] "
        elif op.startswith("s_barrier"): t = " BAR "
        elif op.startswith("scratch"): t = " SCR "
        else: t = "." if op.startswith("v_") else ""
        out.append(t)
    print(name[-70:]); print("".join(out))
