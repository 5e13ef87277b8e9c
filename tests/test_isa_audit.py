"""ISA-level audit of the inline-asm LDS-DMA pieces (CPU: hipcc cross-compiles to gfx950 assembly).  The GEMM and attention kernels issue
their 1-KB LDS-DMA pieces as `global_load_lds_dwordx4 voff, s[base]` from inside asm statements, where hipcc pads no hazards: a scalar
base fresh out of a v_readfirstlane needs five wait states before the DMA reads it (CDNA guide 5.7 item 2).  The kernels form their
bases by scalar arithmetic far from any v_readfirstlane; this test keeps it that way - a passing GPU test is no evidence for a hazard."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc (cross-compiles without a GPU)")
def test_no_lds_dma_reads_a_scalar_base_fresh_from_readfirstlane():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "audit_dma_hazard.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 within five wait states" in r.stdout
    n = int(r.stdout.strip().splitlines()[-1].split()[0])
    assert n > 500          # the audit really saw the kernels' DMA instructions


def test_the_audit_sees_every_vector_unit_write_of_a_scalar_register():
    """the audit must not only know v_readfirstlane - v_readlane, compare masks and carry-outs write SGPRs from the
    vector unit too and need the same wait states before a DMA reads them as its base."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("audit_dma_hazard", os.path.join(ROOT, "scripts", "audit_dma_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    w = mod.sgprs_written_by_vector_op
    assert w("v_readfirstlane_b32 s20, v5") == {20}
    assert w("v_readlane_b32 s7, v3, 12") == {7}
    assert w("v_cmp_lt_u32_e64 s[26:27], v1, v2") == {26, 27}
    assert w("v_add_co_u32_e64 v4, s[8:9], v1, v2") == {8, 9}
    assert w("v_addc_co_u32_e64 v4, s[10:11], v1, v2, s[8:9]") == {10, 11}
    assert w("v_mad_u64_u32 v[2:3], s[4:5], v0, v1, v[6:7]") == {4, 5}
    assert w("v_add_u32_e32 v4, s8, v1") == set()                 # s8 is a SOURCE here
    assert w("v_cndmask_b32_e64 v1, v2, v3, s[8:9]") == set()
    assert w("s_add_u32 s8, s8, s10") == set() and w("v_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]") == set()


def _device_asm(src, tmp_path):
    out = os.path.join(str(tmp_path), os.path.basename(src) + ".s")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-Wno-unused-function",
                        "-S", "--cuda-device-only", "-o", out, os.path.join(ROOT, "layoutdit_amd", "csrc", src)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


def _isa_order():
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_order", os.path.join(ROOT, "scripts", "isa_order.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc (cross-compiles without a GPU)")
def test_reduction_major_gemm_issues_a_steps_reads_in_front_of_its_mfmas(tmp_path):
    """Round 4: hipcc had sunk the (asm) LDS reads of gemm_bf16_tr's k-loop behind each step's MFMAs, right in front of the hand-written
    wait - every step exposed a full LDS latency (profiles/r04_tr_pinned_order_ab.txt).  The order is pinned with sched_barrier now; this
    keeps it pinned: between two waits of the loop every LDS read stands in front of every MFMA."""
    iso = _isa_order()
    ks = iso.kernels(_device_asm("gemm_bf16_tr.hip", tmp_path))
    seen = 0
    for name, lines in ks.items():
        if "gemm_bf16_tr" not in name:
            continue
        body = iso.mfma_loop(lines, "short")
        assert body is not None, name
        o = iso.order(body, valu=False)
        assert o.count("M") >= 16 and (o.count("t") + o.count("r")) >= 16, (name, o)
        for seg in [x for x in __import__("re").split(r"\[[^\]]*\]|BAR", o) if "M" in x and ("t" in x or "r" in x)]:
            first_m = seg.index("M")
            assert "t" not in seg[first_m:] and "r" not in seg[first_m:], (name, seg, o)
        seen += 1
    assert seen >= 10


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc (cross-compiles without a GPU)")
def test_residual_epilogues_do_not_wait_for_their_own_stores(tmp_path):
    """Round 4: vmcnt counts stores and retires in order - an epilogue that fetches, waits, computes and stores pass by pass sits out the
    acknowledgement of the previous pass's stores every time (profiles/r04_epilogue_prefetch_ab.txt).  The fp32 panel GEMM fetches every
    residual quad of its wave tile before the first store: >= 70 stores in a row without a wait.  The bf16 slab epilogue fetches per slab:
    behind a slab's first store there is no vmcnt wait until the next slab's accumulators enter LDS."""
    iso = _isa_order()
    import re
    lines = [l for n, l in iso.kernels(_device_asm("gemm_panel_f32.hip", tmp_path)).items() if "gemm_panel_f32ILi9ELb1ELi2ELi0ELb0" in n]
    assert len(lines) == 1
    tail = lines[0][max(i for i, l in enumerate(lines[0]) if "v_mfma" in l):]
    o = iso.order(tail, valu=False)
    assert max(len(x) for x in re.findall(r"S+", o)) >= 70, o[:600]
    lines = [l for n, l in iso.kernels(_device_asm("gemm_bf16.hip", tmp_path)).items() if "gemm_bf16_mfmaILi2ELi4ELi3ELi2ELi2E" in n]
    assert len(lines) == 1
    tail = lines[0][max(i for i, l in enumerate(lines[0]) if l.startswith("s_barrier")):]
    o = iso.order(tail, valu=False)
    slabs = re.split(r"w{8}", o)
    assert len(slabs) >= 4, o[:600]                       # three slabs of the 192-row tile's waves
    for slab in slabs[1:3]:                               # (the last one runs into the direct-store paths of the ragged tiles)
        assert "S" in slab, slab
        assert "[v(" not in slab[slab.index("S"):], slab
