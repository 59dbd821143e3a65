#!/bin/bash
# Several rocprofv3 --pmc passes of bench.py (one counter group per pass), each summarised for the k_rb kernels.
# usage (GPU box, repo root): tools/pmc_diag.sh <tag> <bench.py args ...>
set -e
tag=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
"$REPO/tools/pmc_run.sh" "${tag}_sq1" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "$@"
"$REPO/tools/pmc_run.sh" "${tag}_sq2" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "$@"
"$REPO/tools/pmc_run.sh" "${tag}_ta" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "$@"
"$REPO/tools/pmc_run.sh" "${tag}_tcc" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "$@"
