#!/usr/bin/env python3
"""ISA audit of the inline-asm LDS-DMA pieces (gemm_bf16*.hip, gemm_fp8.hip, attention_*.hip): a scalar base or an M0 value that comes
fresh out of a v_readfirstlane must not be read by `global_load_lds_dwordx4 voff, s[base]` within five wait states (CDNA guide 5.7
item 2: hipcc pads nothing inside an asm string).  Compiles each file to device assembly and reports every DMA whose scalar base was
written by a vector-unit instruction (v_readfirstlane / v_readlane, v_cmp*, the carry-out of v_add_co ...) less than six
instructions earlier.  CPU only (hipcc cross-compiles)."""
import re, subprocess, sys, os, tempfile


def sgprs_written_by_vector_op(p: str) -> set:
    """SGPR numbers that the vector-unit instruction `p` (one line of device assembly) writes: v_readfirstlane / v_readlane results,
    v_cmp* masks, the carry-out pair of v_add_co / v_addc_co / v_sub_co / v_subb_co / v_mad_u64_u32.  Empty for anything else."""
    if not p.startswith("v_") or " " not in p:
        return set()
    ops_ = [o.strip() for o in p.split(None, 1)[1].split(",")]
    written = set()
    for o in ops_[:2]:                       # at most two destinations (value, carry-out)
        one = re.fullmatch(r"s(\d+)", o)
        rng = re.fullmatch(r"s\[(\d+):(\d+)\]", o)
        if one:
            written.add(int(one.group(1)))
        elif rng:
            written.update(range(int(rng.group(1)), int(rng.group(2)) + 1))
        elif o.startswith("v"):
            if not p.startswith(("v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_subrev_co", "v_subbrev_co", "v_mad_u64", "v_mad_i64")):
                break                        # a VGPR destination with no carry-out form: nothing scalar is written
        else:
            break
    return written


def main() -> int:
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
                        if sgprs_written_by_vector_op(p) & regs:
                            suspect += 1
                            print(f"{f}: {kernel[:60]}: `{p}` {back - 1} instructions before `{t}`")
    print(f"{total} scalar-base LDS-DMA instructions audited, {suspect} within five wait states of a vector-unit write of their base")
    return 1 if suspect else 0


if __name__ == "__main__":
    sys.exit(main())
