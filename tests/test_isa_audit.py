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
