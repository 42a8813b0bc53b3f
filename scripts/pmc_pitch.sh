#!/bin/bash
# usage: pmc_pitch.sh <n> <pad>...  -- cycles and L2 counters of the shipped bf16 kernel and of torch.matmul with
# operands that are views of buffers <pad> elements wider than K
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
n=$1; shift
for pad in "$@"; do
  for impl in ours torch; do
    out=gpurun_out/prof/pitch_${impl}_$pad
    rm -rf "$out"; mkdir -p "$out"
    timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum SQ_WAIT_ANY SQ_BUSY_CYCLES TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum \
      --output-format csv -d "$out" -- \
      python3 bench/prof_gemm.py --n "$n" --pad "$pad" --variant 0 --dtype bf16 --impl "$impl" --iters 3 > "$out.log" 2>&1 || exit 1
    echo "== $impl pad $pad"
    python3 scripts/pmc_summary.py "$out" | grep -A 11 -i "gemm_nt\|Cijk"
  done
done
