#!/usr/bin/env python3
"""A few hundred captured MNIST training steps (for rocprofv3 kernel traces)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench"))
import bench_mnist  # noqa: E402
r = bench_mnist.bench_mnist("cuda", 300, 20, True, True, device_data=True)
print(r["ms_per_step"])
