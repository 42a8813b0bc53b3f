"""``nn`` namespace (``tf.nn.sigmoid`` / ``tf.nn.softmax`` at ``tf_distributed.py:63,65``)."""
from .framework.ops import (bias_add, dropout, l2_loss, log_softmax, relu, sigmoid, softmax,  # noqa: F401
                            softmax_cross_entropy_with_logits,
                            sparse_softmax_cross_entropy_with_logits, tanh)
