#!/bin/bash
# usage: pmc_gemm_f32.sh <tag> <impl> [n]  -- kernel trace + PMC passes for the fp32 GEMM
# (impl = ours | torch: the vendor library on the same operands)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; impl=$2; n=${3:-8192}
out=gpurun_out/prof/$tag
mkdir -p "$out"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- \
  python3 bench/prof_gemm.py --n "$n" --dtype f32 --impl "$impl" --iters 3 > "$out/trace.log" 2>&1 || exit 1
run() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- \
    python3 bench/prof_gemm.py --n "$n" --dtype f32 --impl "$impl" --iters 3 > "$out/$name.log" 2>&1 || return 1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE && \
run sq3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES
