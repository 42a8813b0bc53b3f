#!/usr/bin/env python3
"""The reference's ``tf_distributed_worker.py`` variant of the MNIST job: identical model and
loop, but the default role is ``worker`` (``tf_distributed_worker.py:13``), the default cluster
has 7 workers (``:9``) and it trains for 100 epochs (``:22``). In the reference this is a
copy-paste of ``tf_distributed.py``; here it reuses that script's ``main`` with the three
defaults changed, so the two always describe the same cluster when given the same flags
(the reference's two scripts hard-code *different* clusters and cannot be mixed)."""
from __future__ import print_function

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import distributed_tensorflow_amd as tf  # noqa: E402
import tf_distributed as base  # noqa: E402

DEFAULT_JOB = "worker"
DEFAULT_NUM_WORKERS = 7
DEFAULT_EPOCHS = 100


def main(argv=None):
    return base.main(argv, default_num_workers=DEFAULT_NUM_WORKERS)


if __name__ == "__main__":
    base.define_flags(default_job=DEFAULT_JOB, default_epochs=DEFAULT_EPOCHS)
    tf.app.run(main=main)
