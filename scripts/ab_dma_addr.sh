#!/bin/bash
# Same-box A/B of the LDS-DMA addressing in the bf16 GEMM (profiles/r03_dma_scalar_base_ab.txt): the shipped library (pieces issued from
# a scalar base + per-lane constant offset) against a build with -DLDIT_BF16_VADDR_DMA (per-lane 64-bit addresses through the builtin,
# rounds 1-2).  Build the alternative HERE (hipcc cross-compiles; layoutdit_amd/csrc/build/ travels to the GPU box) with
#   cd layoutdit_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DLDIT_BF16_VADDR_DMA -c gemm_bf16.hip -o build/gemm_bf16_vaddr.o \
#     && hipcc --offload-arch=gfx950 -shared -fPIC -o build/libldit_vaddr.so $(ls build/*.o | grep -v 'gemm_bf16.o\|gemm_bf16_vaddr.o') build/gemm_bf16_vaddr.o
# then run this script through gpurun.  The shipped library is never overwritten: the alternative is loaded by path in its own process.
set -e
test -f layoutdit_amd/csrc/build/libldit_vaddr.so || { echo "build layoutdit_amd/csrc/build/libldit_vaddr.so first (see the header of this script)"; exit 1; }
for i in 1 2; do
echo "== sbase (shipped) =="; timeout -k 10 300 python scripts/gemm_vendor_ref.py 2>&1 | grep -v amdgpu.ids | cut -c1-78
echo "== vaddr (rounds 1-2) =="; timeout -k 10 300 python -c "
import sys; sys.argv=['x']
import layoutdit_amd._lib as L; L.LIB_PATH='layoutdit_amd/csrc/build/libldit_vaddr.so'
import runpy; runpy.run_path('scripts/gemm_vendor_ref.py', run_name='__main__')" 2>&1 | grep -v amdgpu.ids | cut -c1-78
done
