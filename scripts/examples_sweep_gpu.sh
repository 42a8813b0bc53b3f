#!/bin/bash
# One-GPU sweep of the example programs over strategies / options: prints the last AvgTime line
# and the closing lines of each run (anomalies in per-step time show up here, not in unit tests).
run() {
  echo "=== $*"
  timeout -k 10 150 python -m distributed_tensorflow_amd.launch --num_workers 1 --protocol rccl --timeout 140 "$@" > /tmp/sweep_out.txt 2>&1
  rc=$?
  grep -E "AvgTime" /tmp/sweep_out.txt | tail -1
  grep -E "Test-Accuracy|GFLOPS|Weight-Error|Total Time|Global Step|Traceback|Error" /tmp/sweep_out.txt | tail -4
  echo "rc=$rc"
  [ $rc -eq 0 ] || { tail -20 /tmp/sweep_out.txt; exit 1; }
}
run --strategy allreduce_sync examples/tf_distributed.py -- --training_epochs 2
run --strategy reduce_scatter_sync examples/tf_distributed.py -- --training_epochs 2
run --strategy allreduce_sync examples/tf_distributed.py -- --training_epochs 2 --capture_collectives
run --strategy allreduce_sync examples/tf_distributed.py -- --training_epochs 1 --nodevice_data
run --strategy allreduce_sync examples/tf_distributed_worker.py -- --training_epochs 1
run --strategy allreduce_sync examples/tf_distributed_1000Matrix.py -- --training_epochs 2
run --strategy allreduce_sync examples/tf_distributed_1000Matrix.py -- --training_epochs 1 --matrix_size 4096 --steps_per_epoch 200
run --strategy reduce_scatter_sync examples/linear_regression.py -- --max_steps 1000
run --strategy allreduce_sync examples/linear_regression.py -- --max_steps 1000 --capture_collectives
run --strategy allreduce_sync examples/linear_regression.py -- --max_steps 500 --nodevice_data
run --strategy allreduce_sync examples/mnist_monitored.py -- --last_step 1000 --optimizer adam
