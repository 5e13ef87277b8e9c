#!/bin/bash
# Everything profiles/ holds for ONE bench configuration, in one gpurun call:
#   bash scripts/prof_config.sh <tag> [bench.py arguments, e.g. --config 3]
# -> gpurun_out/<tag>_bench.json            the bench line
#    gpurun_out/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command (5 steps)
#    gpurun_out/pmc_<tag>_{sq,fetch,write,lds}/  PMC passes, each counter set in its own run (never combined with traces)
set -o pipefail
TAG=$1; shift
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out
python3 $ROOT/bench.py "$@" > $ROOT/gpurun_out/${TAG}_bench.json 2> $ROOT/gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
cut -c1-260 $ROOT/gpurun_out/${TAG}_bench.json
rm -rf $ROOT/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -o $TAG -- python3 $ROOT/bench.py "$@" --steps 5 --warmup 2 --cpu-sample 0 --no-roofline-pass --no-split-fp32 > $ROOT/gpurun_out/prof_$TAG.log 2>&1
echo "kernel-trace rc=$?"
cp "$(find $ROOT/gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1)" $ROOT/gpurun_out/${TAG}_kernel_stats.csv
run() {  # name counters...
  local name=$1; shift
  rm -rf $ROOT/gpurun_out/pmc_${TAG}_$name
  rocprofv3 --pmc "$@" --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_$name -o $name -- python3 $ROOT/bench.py "${BARGS[@]}" --steps 2 --warmup 1 --cpu-sample 0 --no-roofline-pass --no-split-fp32 > $ROOT/gpurun_out/pmc_${TAG}_$name.log 2>&1
  echo "pmc pass $name rc=$?"
}
BARGS=("$@")
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
# keep the merge-back small: the per-dispatch traces are not needed, only the counter tables
find $ROOT/gpurun_out/prof_$TAG -name '*kernel_trace.csv' -delete
