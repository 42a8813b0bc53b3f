#!/bin/bash
# Rehearse bench.py's multi-rank path with N ranks that all use GPU 0 (gloo control plane):
# exercises strategy probing, the peer-memory candidate and result reporting on one GPU.
n=${1:-2}; shift
port=$((20000 + RANDOM % 20000))
pids=()
for r in $(seq 0 $((n - 1))); do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=$n MASTER_ADDR=127.0.0.1 MASTER_PORT=$port DTF_COMM_BACKEND=gloo \
    OMP_NUM_THREADS=4 timeout -k 10 300 python bench.py --gpus $n "$@" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
exit $rc
