"""MI355X-native distributed tensor runtime with a TF1-style ``tf.train`` facade.

``import distributed_tensorflow_amd as tf`` gives the surface the reference scripts program
against (``tf_distributed.py``, ``tf_distributed_worker.py``, ``tf_distributed_1000Matrix.py``):
``tf.train.ClusterSpec/Server/replica_device_setter/Supervisor/GradientDescentOptimizer``,
``tf.app.flags``, ``tf.device``, ``tf.name_scope``, ``tf.Variable``, ``tf.placeholder``, the
math ops, ``tf.summary`` and ``tf.Session`` -- executed by hand-written gfx950 HIP kernels on
PyTorch-ROCm tensors, with RCCL collectives in place of the gRPC parameter-server transport.
"""
from . import app, summary, train  # noqa: F401
from .app import flags  # noqa: F401
from .client.session import (ConfigProto, GPUOptions, InteractiveSession, RunMetadata,  # noqa: F401
                             RunOptions, Session, get_default_session)
from .client.variable_store import FailedPreconditionError  # noqa: F401
from .framework import dtypes  # noqa: F401
from .framework.device import DeviceSpec  # noqa: F401
from .framework.dtypes import (DType, as_dtype, bfloat16, float16, float32,  # noqa: F401
                               float64, float8_e4m3fn, int32, int64, string)
from .framework.dtypes import bool_ as bool  # noqa: F401,A001
from .framework.gradients import gradients  # noqa: F401
from .framework.graph import (Graph, GraphKeys, Operation, Tensor,  # noqa: F401
                              add_to_collection, colocate_with, control_dependencies, device,
                              get_collection, get_default_graph, name_scope,
                              reset_default_graph, set_random_seed)
from .framework.ops import (abs, add, add_n, argmax, axpby, cast, constant,  # noqa: F401,A001
                            convert_to_tensor, divide, equal, exp, fill, greater, group,
                            identity, less, log, matmul, maximum, minimum, multiply, negative,
                            no_op, not_equal, ones, ones_like, placeholder, pow,
                            random_normal, random_uniform, reciprocal, reduce_max,
                            reduce_mean, reduce_sum, reshape, scale, sigmoid, sign, sqrt, square,
                            stop_gradient, subtract, tanh, transpose, truncated_normal, zeros,
                            zeros_like)
from .framework.variables import (Variable, constant_initializer, get_variable,  # noqa: F401
                                  global_variables, global_variables_initializer,
                                  glorot_uniform_initializer, initialize_all_variables,
                                  is_variable_initialized, local_variables,
                                  local_variables_initializer, ones_initializer,
                                  random_normal_initializer, random_uniform_initializer,
                                  report_uninitialized_variables, trainable_variables,
                                  truncated_normal_initializer, variables_initializer,
                                  zeros_initializer)
from . import layers, losses, nn  # noqa: F401,E402
from .framework.ops import (clip_by_global_norm, clip_by_norm, clip_by_value,  # noqa: F401,E402
                            argmin, cos, expand_dims, floor, global_norm, greater_equal,
                            less_equal, one_hot, reduce_min, sin, squeeze, where)

__version__ = "0.1.0"
