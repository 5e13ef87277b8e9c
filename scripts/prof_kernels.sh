#!/bin/bash
# Per-kernel time of the headline bench under rocprofv3 (run on the GPU box via gpurun).  Writes under gpurun_out/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r1}
mkdir -p gpurun_out
echo "nproc=$(nproc) cpu.max=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null) affinity=$(python3 -c 'import os; print(len(os.sched_getaffinity(0)))')"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-roofline-pass > gpurun_out/prof_bench_$TAG.log 2>&1
echo "rocprofv3 rc=$?"
find gpurun_out/prof_$TAG -name '*stats*' | head
