#!/bin/bash
# Build an alternative libldit (for same-process A/B runs, scripts/ab_gemm_lib.py): one source recompiled with extra -D flags,
# linked with the shipped objects.  Run HERE (hipcc cross-compiles); layoutdit_amd/csrc/build/ travels to the GPU box.
#   scripts/build_alt.sh <name> <source.hip> [-DFLAG ...]   ->  layoutdit_amd/csrc/build/libldit_<name>.so
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../layoutdit_amd/csrc"
make -j8 >/dev/null
base=${src%.hip}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wall -Wno-unused-function "$@" -c "$src" -o "build/alt_${base}_${name}.o"
hipcc --offload-arch=gfx950 -shared -fPIC -o "build/libldit_${name}.so" $(ls build/*.o | grep -v "build/${base}.o\|build/alt_\|gemm_bf16_m16\|gemm_bf16_st\|gemm_bf16_vaddr") "build/alt_${base}_${name}.o"
ls -la "build/libldit_${name}.so"
