#!/bin/bash
# Rehearse bench.py's multi-rank path with N ranks that all use GPU 0 (gloo control plane):
# exercises strategy probing, the peer-memory candidate and result reporting on one GPU.
# Every rank's stdout goes to $OUT/rank<r>.out (rank 0's must hold exactly one JSON line).
n=${1:-2}; shift
OUT=${OUT:-gpurun_out/same_device}
mkdir -p "$OUT"
port=$((20000 + RANDOM % 20000))
pids=()
for r in $(seq 0 $((n - 1))); do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=$n MASTER_ADDR=127.0.0.1 MASTER_PORT=$port DTF_COMM_BACKEND=gloo \
    OMP_NUM_THREADS=4 timeout -k 10 300 python bench.py --gpus $n "$@" \
    > "$OUT/rank$r.out" 2> "$OUT/rank$r.err" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
cat "$OUT/rank0.out"
grep -h "candidate\|note:" "$OUT/rank0.err" | head -40
exit $rc
