#!/usr/bin/env python3
"""Instruction mix of the innermost MFMA loop of every kernel in a device-assembly file (hipcc --cuda-device-only -S):
python scripts/isa_loops.py file.s [name filter] [mfma mnemonic prefix]"""
import re, sys
from collections import Counter
path = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""; mf = sys.argv[3] if len(sys.argv) > 3 else "v_mfma"
cur = None; ker = {}
for l in open(path):
    t = re.sub(r"\s+", " ", l.split(";")[0]).strip()
    if not t: continue
    if t.endswith(":") and t.startswith("_Z"): cur = t[:-1]; ker[cur] = []; continue
    if cur is not None and not t.startswith("."): ker[cur].append(t)
    elif cur is not None and t.startswith(".L") and t.endswith(":"): ker[cur].append(t)
for name, ins in ker.items():
    if filt not in name: continue
    labels = {t[:-1]: i for i, t in enumerate(ins) if t.endswith(":")}
    best = None
    for i, t in enumerate(ins):
        m = re.search(r"s_cbranch\S*\s+(\S+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = [x for x in ins[labels[m.group(1)]:i + 1] if not x.endswith(":")]
            c = Counter(x.split()[0] for x in body)
            n = sum(v for k, v in c.items() if k.startswith(mf))
            if n >= 8 and (best is None or len(body) < len(best[0])): best = (body, c, n)
    if best:
        body, c, n = best
        w = Counter(x for x in body if x.startswith("s_waitcnt"))
        print(name[-60:], "| len", len(body), "mfma", n, "dma", c.get("global_load_lds_dwordx4", 0), "ds_read_b128", c.get("ds_read_b128", 0),
              "tr", c.get("ds_read_b64_tr_b16", 0), "valu", sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma")),
              "salu", sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith(("s_waitcnt", "s_barrier", "s_nop"))), "| waits", dict(w))
