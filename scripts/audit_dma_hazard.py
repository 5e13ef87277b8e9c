#!/usr/bin/env python3
"""ISA audit of the inline-asm LDS-DMA pieces (gemm_bf16*.hip, gemm_fp8.hip, attention_*.hip): a scalar base or an M0 value that comes
fresh out of a v_readfirstlane must not be read by `global_load_lds_dwordx4 voff, s[base]` within five wait states (CDNA guide 5.7
item 2: hipcc pads nothing inside an asm string).  Compiles each file to device assembly and reports every DMA whose scalar base was
written by a v_readfirstlane (or any VALU) less than six instructions earlier.  CPU only (hipcc cross-compiles)."""
import re, subprocess, sys, os, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or ["gemm_bf16.hip", "gemm_bf16_tr.hip", "gemm_fp8.hip", "attention_bf16.hip", "attention_planes.hip"]
total = suspect = 0
for f in files:
    with tempfile.TemporaryDirectory() as d:
        s = os.path.join(d, "x.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{root}/include", "-Wno-unused-function",
                        "--cuda-device-only", "-S", os.path.join(root, "layoutdit_amd", "csrc", f), "-o", s], check=True,
                       stderr=subprocess.DEVNULL)
        kernel = "?"
        instr = []
        for line in open(s):
            t = line.strip()
            if t.endswith(":") and t.startswith("_Z"):
                kernel, instr = t[:-1], []
                continue
            if not t or t[0] in ".;/" or t.endswith(":"):
                continue
            instr.append(t)
            if t.startswith("global_load_lds_dwordx4"):
                m = re.search(r"s\[(\d+):(\d+)\]", t)
                if not m:
                    continue
                total += 1
                regs = {int(m.group(1)), int(m.group(2))}
                for back in range(2, 7):
                    if len(instr) < back:
                        break
                    p = instr[-back]
                    w = re.match(r"v_readfirstlane_b32 s(\d+)", p)
                    if w and int(w.group(1)) in regs:
                        suspect += 1
                        print(f"{f}: {kernel[:60]}: `{p}` {back - 1} instructions before `{t}`")
print(f"{total} scalar-base LDS-DMA instructions audited, {suspect} within five wait states of a v_readfirstlane of their base")
sys.exit(1 if suspect else 0)
