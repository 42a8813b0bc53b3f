"""``train`` namespace: cluster definition, servers, placement, optimizers, supervision and
checkpointing -- the ``tf.train.*`` symbols the reference scripts use
(``tf_distributed.py:11,18,34,75,92``)."""
from .cluster import ClusterDef, ClusterSpec, task_from_env  # noqa: F401
from .device_setter import (GreedyLoadBalancingStrategy, byte_size_load_fn,  # noqa: F401
                            replica_device_setter)
from .optimizer import GradientDescentOptimizer, Optimizer  # noqa: F401
