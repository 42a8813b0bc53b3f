"""``train`` namespace: cluster definition, servers, placement, optimizers, supervision and
checkpointing -- the ``tf.train.*`` symbols the reference scripts use
(``tf_distributed.py:11,18,34,75,92``)."""
from .cluster import ClusterDef, ClusterSpec, task_from_env  # noqa: F401
from .device_setter import (GreedyLoadBalancingStrategy, byte_size_load_fn,  # noqa: F401
                            replica_device_setter)
from .learning_rate_decay import (cosine_decay, exponential_decay,  # noqa: F401
                                  inverse_time_decay, natural_exp_decay,
                                  piecewise_constant, polynomial_decay)
from .moving_averages import ExponentialMovingAverage  # noqa: F401
from .optimizer import (AdadeltaOptimizer, AdagradOptimizer, AdamOptimizer,  # noqa: F401
                        GradientDescentOptimizer, MomentumOptimizer, Optimizer,
                        RMSPropOptimizer, SyncReplicasOptimizer)
from .saver import (CheckpointReader, NewCheckpointReader, Saver,  # noqa: F401
                    checkpoint_exists, get_checkpoint_state, latest_checkpoint,
                    list_variables, load_variable)
from .server import Server, ServerDef  # noqa: F401
from .supervisor import Coordinator, SessionManager, Supervisor  # noqa: F401
from .training_util import (create_global_step, get_global_step,  # noqa: F401
                            get_or_create_global_step)
from .monitored_session import (CheckpointSaverHook, ChiefSessionCreator, FinalOpsHook,  # noqa: F401,E402
                                LoggingTensorHook, MonitoredSession, MonitoredTrainingSession,
                                NanLossDuringTrainingError, NanTensorHook, Scaffold,
                                SessionRunArgs, SessionRunContext, SessionRunHook,
                                SessionRunValues, StepCounterHook, StopAtStepHook,
                                SummarySaverHook, WorkerSessionCreator)
