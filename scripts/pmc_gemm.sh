#!/bin/bash
# usage: pmc_gemm.sh <tag> <variant> [n] [dtype] [impl]   -- collects PMC passes for one GEMM kernel
# (impl = ours | torch: the vendor library on the same operands, for a counter-by-counter diff)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; variant=$2; n=${3:-8192}; dtype=${4:-bf16}; impl=${5:-ours}
out=gpurun_out/prof/$tag
mkdir -p "$out"
run() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- \
    python3 bench/prof_gemm.py --n "$n" --variant "$variant" --dtype "$dtype" --impl "$impl" --iters 3 > "$out/$name.log" 2>&1 || return 1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE && \
run sq2 SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum && \
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum && \
run sq3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 TCC_EA0_RDREQ_sum TCC_REQ_sum
