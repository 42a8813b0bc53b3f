"""``tf.train.MonitoredTrainingSession`` and its hooks: the hook-driven successor of the
``Supervisor`` flow the reference uses (``tf_distributed.py:91-101``: chief initialises, the
others wait, a managed session runs the training loop).

The session protocol is the Supervisor's (``SessionManager``): the chief restores the latest
checkpoint or runs the init op and publishes the state (broadcast under the collective
strategies, plain assignment to the parameter server otherwise), every other task waits for it.
Around every ``run`` the hooks may add fetches of their own (``before_run``) and see their values
(``after_run``); they ride on the SAME ``Session.run`` as the caller's fetches -- one captured
step, no second launch -- and a hook asks for the end of training through
``run_context.request_stop()``.
"""
from __future__ import annotations

import os
import time
from typing import Any, Callable, Dict, List, Optional, Sequence

import numpy as np

from .. import summary as summary_lib
from ..client.session import ConfigProto, Session
from ..framework.graph import GraphKeys, Operation, Tensor, get_default_graph
from ..framework.variables import (global_variables, global_variables_initializer,
                                   local_variables_initializer, report_uninitialized_variables)
from .saver import Saver
from .supervisor import SessionManager
from .training_util import get_global_step


# ---------------------------------------------------------------------------------------
# hook protocol
# ---------------------------------------------------------------------------------------
class SessionRunArgs:
    def __init__(self, fetches=None, feed_dict: Optional[Dict[Any, Any]] = None):
        self.fetches = fetches
        self.feed_dict = feed_dict


class SessionRunValues:
    def __init__(self, results):
        self.results = results


class SessionRunContext:
    def __init__(self, original_args: SessionRunArgs, session: Session):
        self.original_args = original_args
        self.session = session
        self._stop = False

    @property
    def stop_requested(self) -> bool:
        return self._stop

    def request_stop(self) -> None:
        self._stop = True


class SessionRunHook:
    def begin(self) -> None:
        """Called once, before the session exists: the graph may still be extended."""

    def after_create_session(self, session: Session, coord=None) -> None:
        pass

    def before_run(self, run_context: SessionRunContext) -> Optional[SessionRunArgs]:
        return None

    def after_run(self, run_context: SessionRunContext, run_values: SessionRunValues) -> None:
        pass

    def end(self, session: Session) -> None:
        pass


def _global_step_tensor() -> Tensor:
    step = get_global_step()
    if step is None:
        raise RuntimeError("this hook needs a global_step variable "
                           "(tf.train.get_or_create_global_step())")
    return step._as_graph_element()


class StopAtStepHook(SessionRunHook):
    """Stop after ``num_steps`` more steps, or when ``global_step`` reaches ``last_step``."""

    def __init__(self, num_steps: Optional[int] = None, last_step: Optional[int] = None):
        if (num_steps is None) == (last_step is None):
            raise ValueError("exactly one of num_steps and last_step must be given")
        self._num_steps, self._last_step = num_steps, last_step

    def begin(self):
        self._step = _global_step_tensor()

    def after_create_session(self, session, coord=None):
        if self._last_step is None:
            self._last_step = int(session.run(self._step)) + self._num_steps

    def before_run(self, run_context):
        return SessionRunArgs(self._step)

    def after_run(self, run_context, run_values):
        if int(run_values.results) >= self._last_step:
            run_context.request_stop()


class _EveryN:
    """Fires every ``every_steps`` steps or ``every_secs`` seconds (and at the first call)."""

    def __init__(self, every_secs: Optional[float], every_steps: Optional[int]):
        if (every_secs is None) == (every_steps is None):
            raise ValueError("exactly one of the seconds and steps intervals must be given")
        self._secs, self._steps = every_secs, every_steps
        self._last_time: Optional[float] = None
        self._last_step: Optional[int] = None

    def should_fire(self, step: int) -> bool:
        if self._last_step is None:
            return True
        if step == self._last_step:
            return False
        if self._secs is not None:
            return time.monotonic() - self._last_time >= self._secs
        return step >= self._last_step + self._steps

    def fired(self, step: int):
        elapsed = None if self._last_time is None else time.monotonic() - self._last_time
        steps = None if self._last_step is None else step - self._last_step
        self._last_time, self._last_step = time.monotonic(), step
        return elapsed, steps


class CheckpointSaverHook(SessionRunHook):
    """Saves a V2 checkpoint every ``save_secs`` / ``save_steps`` and when training ends."""

    def __init__(self, checkpoint_dir: str, save_secs: Optional[float] = None,
                 save_steps: Optional[int] = None, saver: Optional[Saver] = None,
                 checkpoint_basename: str = "model.ckpt", scaffold: Optional["Scaffold"] = None):
        self._path = os.path.join(checkpoint_dir, checkpoint_basename)
        self._timer = _EveryN(save_secs, save_steps)
        self._saver, self._scaffold = saver, scaffold
        self.saved_steps: List[int] = []

    def begin(self):
        self._step = _global_step_tensor()
        if self._saver is None and (self._scaffold is None or self._scaffold.saver is None
                                    and self._scaffold._finalized):
            self._saver = Saver()

    def _save(self, session, step: int) -> None:
        sharded = getattr(getattr(session, "_runtime", None), "has_sharded_storage", None)
        if sharded is not None and sharded():
            # reading a variable that is stored as per-rank blocks is a collective; this hook
            # runs on the chief only and would wait for peers that never call it
            if not getattr(self, "_warned", False):
                import warnings
                warnings.warn("variables are stored sharded: CheckpointSaverHook is off; call "
                              "saver.save(...) on every worker")
                self._warned = True
            return
        if self._saver is None:               # the scaffold was finalized after begin()
            self._saver = self._scaffold.saver
        os.makedirs(os.path.dirname(self._path), exist_ok=True)
        self._saver.save(session, self._path, global_step=step)
        self.saved_steps.append(step)

    def after_create_session(self, session, coord=None):
        step = int(session.run(self._step))
        self._save(session, step)
        self._timer.fired(step)

    def before_run(self, run_context):
        return SessionRunArgs(self._step)

    def after_run(self, run_context, run_values):
        step = int(run_values.results)
        if self._timer.should_fire(step):
            self._timer.fired(step)
            self._save(run_context.session, step)

    def end(self, session):
        step = int(session.run(self._step))
        if not self.saved_steps or self.saved_steps[-1] != step:
            self._save(session, step)


class StepCounterHook(SessionRunHook):
    """``global_step/sec`` every ``every_n_steps`` steps (logged, and written as a scalar summary
    when an output directory or writer is given)."""

    def __init__(self, every_n_steps: Optional[int] = 100, every_n_secs: Optional[float] = None,
                 output_dir: Optional[str] = None, summary_writer=None):
        self._timer = _EveryN(every_n_secs, every_n_steps if every_n_secs is None else None)
        self._output_dir, self._writer = output_dir, summary_writer
        self.rates: List[float] = []

    def begin(self):
        self._step = _global_step_tensor()
        if self._writer is None and self._output_dir:
            self._writer = summary_lib.FileWriter(self._output_dir)

    def before_run(self, run_context):
        return SessionRunArgs(self._step)

    def after_run(self, run_context, run_values):
        step = int(run_values.results)
        if self._timer.should_fire(step):
            elapsed, steps = self._timer.fired(step)
            if elapsed and steps:
                rate = steps / elapsed
                self.rates.append(rate)
                if self._writer is not None:
                    self._writer.add_summary(summary_lib.encode_scalar_summary(
                        "global_step/sec", rate), step)

    def end(self, session):
        if self._writer is not None:
            self._writer.flush()


class LoggingTensorHook(SessionRunHook):
    """Prints the given tensors every ``every_n_iter`` local steps (or ``every_n_secs``)."""

    def __init__(self, tensors, every_n_iter: Optional[int] = None,
                 every_n_secs: Optional[float] = None, at_end: bool = False,
                 formatter: Optional[Callable[[Dict[str, Any]], str]] = None,
                 print_fn: Callable[[str], None] = print):
        if not isinstance(tensors, dict):
            tensors = {getattr(t, "name", str(t)): t for t in tensors}
        self._tensors = dict(tensors)
        self._timer = None if (every_n_iter is None and every_n_secs is None) else \
            _EveryN(every_n_secs, every_n_iter)
        if self._timer is None and not at_end:
            raise ValueError("either at_end or one of every_n_iter / every_n_secs is required")
        self._at_end, self._formatter, self._print = at_end, formatter, print_fn
        self._iter = 0
        self.lines: List[str] = []

    def begin(self):
        g = get_default_graph()
        self._tensors = {k: g.as_graph_element(v) for k, v in self._tensors.items()}

    def _emit(self, values: Dict[str, Any]) -> None:
        line = self._formatter(values) if self._formatter else ", ".join(
            "%s = %s" % (k, np.asarray(v)) for k, v in values.items())
        self.lines.append(line)
        self._print(line)

    def before_run(self, run_context):
        self._fire = self._timer is not None and self._timer.should_fire(self._iter)
        return SessionRunArgs(self._tensors) if self._fire else None

    def after_run(self, run_context, run_values):
        if self._fire:
            self._timer.fired(self._iter)
            self._emit(run_values.results)
        self._iter += 1

    def end(self, session):
        if self._at_end:
            self._emit(session.run(self._tensors))


class NanLossDuringTrainingError(RuntimeError):
    def __str__(self) -> str:
        return "NaN loss during training."


class NanTensorHook(SessionRunHook):
    """Stops (or raises ``NanLossDuringTrainingError``) when the loss is not a number."""

    def __init__(self, loss_tensor: Tensor, fail_on_nan_loss: bool = True):
        self._loss, self._fail = loss_tensor, fail_on_nan_loss

    def before_run(self, run_context):
        return SessionRunArgs(self._loss)

    def after_run(self, run_context, run_values):
        if np.isnan(np.asarray(run_values.results)).any():
            if self._fail:
                raise NanLossDuringTrainingError()
            run_context.request_stop()


class SummarySaverHook(SessionRunHook):
    """Evaluates the summary op every ``save_steps`` / ``save_secs`` -- as part of the training
    step's own ``run`` -- and appends it to the event file."""

    def __init__(self, save_steps: Optional[int] = None, save_secs: Optional[float] = None,
                 output_dir: Optional[str] = None, summary_writer=None, scaffold=None,
                 summary_op=None):
        self._timer = _EveryN(save_secs, save_steps)
        self._output_dir, self._writer = output_dir, summary_writer
        self._scaffold, self._summary_op = scaffold, summary_op
        self.written_steps: List[int] = []

    def begin(self):
        self._step = _global_step_tensor()
        if self._summary_op is None and self._scaffold is not None:
            self._summary_op = self._scaffold.summary_op
        if self._summary_op is None:
            self._summary_op = summary_lib.merge_all()
        if self._writer is None and self._output_dir:
            self._writer = summary_lib.FileWriter(self._output_dir)
        self._next_step: Optional[int] = None

    def before_run(self, run_context):
        self._fire = self._summary_op is not None and self._writer is not None and (
            self._next_step is None or self._timer.should_fire(self._next_step))
        fetches = {"step": self._step}
        if self._fire:
            fetches["summary"] = self._summary_op
        return SessionRunArgs(fetches)

    def after_run(self, run_context, run_values):
        step = int(run_values.results["step"])
        if self._fire:
            self._timer.fired(step)
            self._writer.add_summary(run_values.results["summary"], step)
            self.written_steps.append(step)
        self._next_step = step + 1

    def end(self, session):
        if self._writer is not None:
            self._writer.flush()


class FinalOpsHook(SessionRunHook):
    """Runs ``final_ops`` once when training ends; the values are left in ``final_ops_values``."""

    def __init__(self, final_ops, final_ops_feed_dict=None):
        self._ops, self._feed = final_ops, final_ops_feed_dict
        self.final_ops_values = None

    def end(self, session):
        if self._ops is not None:
            self.final_ops_values = session.run(self._ops, feed_dict=self._feed)


# ---------------------------------------------------------------------------------------
# session creation
# ---------------------------------------------------------------------------------------
class Scaffold:
    """The ops a training session needs besides the model: initialisation, readiness test,
    saver and summaries; anything not given is created by ``finalize``."""

    def __init__(self, init_op: Optional[Operation] = None, init_feed_dict=None,
                 init_fn: Optional[Callable] = None, ready_op=None, local_init_op=None,
                 summary_op=None, saver: Optional[Saver] = None):
        self.init_op, self.init_feed_dict, self.ready_op = init_op, init_feed_dict, ready_op
        self.local_init_op, self.summary_op, self.saver = local_init_op, summary_op, saver
        self.init_fn = (lambda sess: init_fn(self, sess)) if init_fn else None
        self._finalized = False

    def finalize(self) -> "Scaffold":
        if self._finalized:
            return self
        g = get_default_graph()
        if self.init_op is None:
            self.init_op = global_variables_initializer()
        if self.ready_op is None:
            self.ready_op = report_uninitialized_variables()
        if self.local_init_op is None:
            self.local_init_op = local_variables_initializer()
        if self.summary_op is None:
            self.summary_op = summary_lib.merge_all()
        if self.saver is None:
            savers = g.get_collection(GraphKeys.SAVERS)
            self.saver = savers[0] if savers else (Saver() if global_variables() else None)
        g.finalize()
        self._finalized = True
        return self


class ChiefSessionCreator:
    def __init__(self, scaffold: Optional[Scaffold] = None, master: str = "",
                 config: Optional[ConfigProto] = None, checkpoint_dir: Optional[str] = None):
        self._scaffold = scaffold or Scaffold()
        self._master, self._config, self._checkpoint_dir = master, config, checkpoint_dir

    @property
    def scaffold(self) -> Scaffold:
        return self._scaffold

    def create_session(self) -> Session:
        s = self._scaffold.finalize()
        manager = SessionManager(local_init_op=s.local_init_op, ready_op=s.ready_op,
                                 graph=get_default_graph())
        return manager.prepare_session(self._master, init_op=s.init_op, saver=s.saver,
                                       checkpoint_dir=self._checkpoint_dir, config=self._config,
                                       init_feed_dict=s.init_feed_dict, init_fn=s.init_fn)


class WorkerSessionCreator:
    def __init__(self, scaffold: Optional[Scaffold] = None, master: str = "",
                 config: Optional[ConfigProto] = None, max_wait_secs: float = 30 * 60,
                 recovery_wait_secs: float = 1.0):
        self._scaffold = scaffold or Scaffold()
        self._master, self._config = master, config
        self._max_wait_secs, self._recovery_wait_secs = max_wait_secs, recovery_wait_secs

    @property
    def scaffold(self) -> Scaffold:
        return self._scaffold

    def create_session(self) -> Session:
        s = self._scaffold.finalize()
        manager = SessionManager(local_init_op=s.local_init_op, ready_op=s.ready_op,
                                 graph=get_default_graph(),
                                 recovery_wait_secs=self._recovery_wait_secs)
        return manager.wait_for_session(self._master, config=self._config,
                                        max_wait_secs=self._max_wait_secs)


# ---------------------------------------------------------------------------------------
# the monitored session
# ---------------------------------------------------------------------------------------
class MonitoredSession:
    """A session whose ``run`` calls pass through hooks, and which knows when to stop::

        with tf.train.MonitoredTrainingSession(master=server.target, is_chief=..., hooks=[
                tf.train.StopAtStepHook(last_step=10000)]) as sess:
            while not sess.should_stop():
                sess.run(train_op, feed_dict=...)
    """

    def __init__(self, session_creator=None, hooks: Optional[Sequence[SessionRunHook]] = None):
        self._hooks = list(hooks or [])
        for h in self._hooks:
            h.begin()
        creator = session_creator or ChiefSessionCreator()
        self._sess: Optional[Session] = creator.create_session()
        self._stop = False
        for h in self._hooks:
            h.after_create_session(self._sess, None)

    @property
    def graph(self):
        return self._sess.graph if self._sess is not None else None

    def raw_session(self) -> Optional[Session]:
        return self._sess

    def should_stop(self) -> bool:
        return self._stop or self._sess is None

    def run(self, fetches, feed_dict=None, options=None, run_metadata=None):
        if self._sess is None:
            raise RuntimeError("run() on a closed MonitoredSession")
        context = SessionRunContext(SessionRunArgs(fetches, feed_dict), self._sess)
        feeds = dict(feed_dict or {})
        requests: Dict[int, Any] = {}
        for i, h in enumerate(self._hooks):
            req = h.before_run(context)
            if req is None:
                continue
            if req.fetches is not None:
                requests[i] = req.fetches
            for k, v in (req.feed_dict or {}).items():
                if k in feeds:
                    raise RuntimeError("a hook feeds %r, which the caller (or another hook) "
                                       "feeds as well" % (k,))
                feeds[k] = v
        kwargs = {}
        if options is not None or run_metadata is not None:
            kwargs = {"options": options, "run_metadata": run_metadata}
        out = self._sess.run({"caller": fetches, "hooks": requests}, feed_dict=feeds or None,
                             **kwargs)
        for i, h in enumerate(self._hooks):
            h.after_run(context, SessionRunValues(out["hooks"].get(i)))
        if context.stop_requested:
            self._stop = True
        return out["caller"]

    def close(self) -> None:
        if self._sess is None:
            return
        sess, self._sess = self._sess, None
        try:
            for h in self._hooks:
                h.end(sess)
        finally:
            sess.close()

    def __enter__(self) -> "MonitoredSession":
        return self

    def __exit__(self, exc_type, exc, tb) -> bool:
        if exc_type is None:
            self.close()
        else:                     # training failed: no final checkpoint of a broken state
            sess, self._sess = self._sess, None
            if sess is not None:
                sess.close()
        return False


def MonitoredTrainingSession(master: str = "", is_chief: bool = True,
                             checkpoint_dir: Optional[str] = None,
                             scaffold: Optional[Scaffold] = None,
                             hooks: Optional[Sequence[SessionRunHook]] = None,
                             chief_only_hooks: Optional[Sequence[SessionRunHook]] = None,
                             save_checkpoint_secs: Optional[float] = 600,
                             save_summaries_steps: Optional[int] = 100,
                             save_summaries_secs: Optional[float] = None,
                             config: Optional[ConfigProto] = None,
                             stop_grace_period_secs: float = 120,
                             log_step_count_steps: Optional[int] = 100,
                             max_wait_secs: float = 7200,
                             save_checkpoint_steps: Optional[int] = None) -> MonitoredSession:
    """The chief restores or initialises the model, saves checkpoints and summaries under
    ``checkpoint_dir`` and counts steps per second; every other task waits for the chief's
    state. ``hooks`` run on every task, ``chief_only_hooks`` on the chief."""
    scaffold = scaffold or Scaffold()
    all_hooks = list(hooks or [])
    if not is_chief:
        return MonitoredSession(WorkerSessionCreator(scaffold, master, config, max_wait_secs),
                                all_hooks)
    all_hooks += list(chief_only_hooks or [])
    if checkpoint_dir:
        if log_step_count_steps:
            all_hooks.append(StepCounterHook(every_n_steps=log_step_count_steps,
                                             output_dir=checkpoint_dir))
        if save_summaries_steps or save_summaries_secs:
            all_hooks.append(SummarySaverHook(
                save_steps=save_summaries_steps if not save_summaries_secs else None,
                save_secs=save_summaries_secs, output_dir=checkpoint_dir, scaffold=scaffold))
        if save_checkpoint_secs or save_checkpoint_steps:
            all_hooks.append(CheckpointSaverHook(
                checkpoint_dir, save_secs=save_checkpoint_secs if not save_checkpoint_steps
                else None, save_steps=save_checkpoint_steps, scaffold=scaffold))
    return MonitoredSession(ChiefSessionCreator(scaffold, master, config, checkpoint_dir),
                            all_hooks)
