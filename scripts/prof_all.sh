#!/bin/bash
# Everything the profiles/ directory is built from, in one gpurun call: bench line, rocprofv3 kernel stats, PMC passes.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r01}
mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; tail -c 600 gpurun_out/bench_$TAG.json
bash scripts/prof_kernels.sh $TAG
bash scripts/prof_pmc.sh $TAG
