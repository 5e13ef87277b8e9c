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
