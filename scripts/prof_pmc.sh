#!/bin/bash
# PMC counter passes for the headline bench (each --pmc set in its own run, kernel-trace/stats NOT combined).
# Usage on the GPU box: bash scripts/prof_pmc.sh <tag>
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r1}
mkdir -p gpurun_out
run() {  # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_${TAG}_$name -o $name -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-roofline-pass > gpurun_out/pmc_${TAG}_$name.log 2>&1
  echo "pmc pass $name rc=$?"
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
run tcc TCC_HIT_sum TCC_MISS_sum
find gpurun_out -name '*counter_collection.csv' | head
