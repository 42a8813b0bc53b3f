#!/bin/bash
# usage: pmc_l2.sh <tag> <variant> [n] : L2 hit/miss + fabric reads of one bf16 GEMM variant
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; variant=$2; n=${3:-16384}
out=gpurun_out/prof/$tag
mkdir -p "$out"
timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d "$out/l2" -- \
  python3 bench/prof_gemm.py --n "$n" --variant "$variant" --dtype bf16 --impl ours --iters 3 > "$out/l2.log" 2>&1 || exit 1
python scripts/pmc_summary.py "$out" gemm
