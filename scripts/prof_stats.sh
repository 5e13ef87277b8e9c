#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py for several configs in one gpurun call (kernel-time breakdowns while tuning):
#   bash scripts/prof_stats.sh <tag> <config> [<config> ...]   -> gpurun_out/<tag>_cfg<N>_kernel_stats.csv
set -o pipefail
TAG=$1; shift
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out
for C in "$@"; do
  rm -rf $ROOT/gpurun_out/prof_${TAG}_$C
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_$C -o s -- python3 $ROOT/bench.py --config $C --steps 5 --warmup 2 --cpu-sample 0 --no-roofline-pass --no-split-fp32 > $ROOT/gpurun_out/prof_${TAG}_$C.log 2>&1
  echo "cfg $C rc=$?"
  cp "$(find $ROOT/gpurun_out/prof_${TAG}_$C -name '*kernel_stats.csv' | head -1)" $ROOT/gpurun_out/${TAG}_cfg${C}_kernel_stats.csv
  rm -rf $ROOT/gpurun_out/prof_${TAG}_$C
  tail -1 $ROOT/gpurun_out/prof_${TAG}_$C.log | cut -c1-200
done
