#!/bin/bash
# The costs the MNIST example prints over 600 steps under reduce_scatter_sync, eight runs each
# with the batch read through the step ring and with the feed gather (DTF_NO_RING_READS=1): how
# far single-batch costs scatter around the learning trend (first print 9.2-11.7, last 6.9-8.9),
# which is why tests/test_gpu_examples.py compares pairs of prints.
for i in 1 2 3 4 5 6 7 8; do
  for env in "" "DTF_NO_RING_READS=1"; do
    out=$(env $env timeout -k 10 100 python -m distributed_tensorflow_amd.launch --num_workers 1 --protocol rccl --strategy reduce_scatter_sync --timeout 90 examples/tf_distributed.py -- --max_steps 600 --logs_path /tmp/lg$i --data_dir /tmp/nodata 2>&1)
    echo "[$env] rc=$? $(echo "$out" | grep -o 'Cost: [0-9.]*' | tr '\n' ' ')"
  done
done
