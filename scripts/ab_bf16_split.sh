#!/bin/bash
# Same-box A/B of the bf16 GEMM's one-loader-per-SIMD DMA issue (production) against every-wave-issues (build with
# -DLDIT_BF16_NO_SPLIT into csrc/build/libldit_hip_nosplit.so).  Swaps the library file between bench runs.
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd $ROOT
cp layoutdit_amd/libldit_hip.so /tmp/split.so
# whatever ends this script (interrupt, error, normal exit), the production library is put back
trap 'cp /tmp/split.so "$ROOT/layoutdit_amd/libldit_hip.so"' EXIT INT TERM
for round in 1 2 3; do
  for v in split nosplit; do
    if [ $v = split ]; then cp /tmp/split.so layoutdit_amd/libldit_hip.so; else cp layoutdit_amd/csrc/build/libldit_hip_nosplit.so layoutdit_amd/libldit_hip.so; fi
    for c in 3 2; do
      echo -n "$v config $c: "; python bench.py --config $c --cpu-sample 0 --no-roofline-pass 2>/dev/null | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'
    done
  done
done
cp /tmp/split.so layoutdit_amd/libldit_hip.so
