"""``nn`` namespace (``tf.nn.sigmoid`` / ``tf.nn.softmax`` at ``tf_distributed.py:63,65``)."""
from .framework.ops import (bias_add, dropout, l2_loss, log_softmax, relu, relu6,  # noqa: F401
                            sigmoid, sigmoid_cross_entropy_with_logits, softmax,
                            softmax_cross_entropy_with_logits, softplus,
                            sparse_softmax_cross_entropy_with_logits, tanh, xw_plus_b)
