#!/bin/bash
# usage: pmc_cycles.sh <n> <variant>...   -- one PMC pass per launcher variant ("torch" = the vendor
# library): cycles the GPU was active, matrix-pipe busy cycles, wave cycles and the wait counters,
# which give a kernel's matrix-pipe utilisation independent of the clock it ran at
# (DTF_PMC_K=<k>: an n x n x k product)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
n=$1; shift
for v in "$@"; do
  out=gpurun_out/prof/cycles_$v
  rm -rf "$out"; mkdir -p "$out"
  impl=ours; var=$v
  if [ "$v" = torch ]; then impl=torch; var=0; fi
  timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU \
    --output-format csv -d "$out" -- \
    python3 bench/prof_gemm.py --n "$n" --k "${DTF_PMC_K:-0}" --variant "$var" --dtype bf16 --impl "$impl" --iters 3 > "$out.log" 2>&1 || exit 1
  echo "== variant $v"
  python3 scripts/pmc_summary.py "$out" | grep -A 12 -i "gemm\|Cijk"
done
