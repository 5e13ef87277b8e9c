#!/bin/bash
# PMC counters for the bf16 GEMM microbench (GPU box)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export M=16384
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_bf16_sq -o sq -- python3 scripts/gemm_bf16_bench.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM --output-format csv -d gpurun_out/pmc_bf16_lds -o lds -- python3 scripts/gemm_bf16_bench.py > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_bf16_tcc -o tcc -- python3 scripts/gemm_bf16_bench.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_bf16_fetch -o fetch -- python3 scripts/gemm_bf16_bench.py > /dev/null 2>&1
python3 scripts/pmc_summary.py gpurun_out/pmc_bf16_sq/sq_counter_collection.csv | cut -c1-300
python3 scripts/pmc_summary.py gpurun_out/pmc_bf16_lds/lds_counter_collection.csv | cut -c1-300
python3 scripts/pmc_summary.py gpurun_out/pmc_bf16_tcc/tcc_counter_collection.csv gpurun_out/pmc_bf16_fetch/fetch_counter_collection.csv | cut -c1-300
