"""Supervisor / SessionManager / Coordinator: the chief / non-chief start-up protocol and the
background services around a training session.

Reference usage (``tf_distributed.py:92,96,130``)::

    sv = tf.train.Supervisor(is_chief=(task_index == 0), global_step=global_step, init_op=init_op)
    with sv.prepare_or_wait_for_session(server.target) as sess:
        ...
    sv.stop()

Protocol:

* the **chief** (worker 0) creates the session, restores the latest checkpoint from ``logdir``
  if there is one, otherwise runs ``init_op``, then checks ``ready_op`` (no uninitialized
  variables);
* a **non-chief** creates a session and polls ``ready_op`` every ``recovery_wait_secs`` until
  the chief has initialised the shared state.

How "shared state" becomes visible depends on the runtime: with a parameter server the state
*is* shared (it lives in the ps), so polling it is the whole protocol; with the collective
runtimes the chief's values are broadcast to every rank and a barrier replaces the poll.

With ``logdir`` set the chief also runs the standard services: a checkpoint thread
(``save_model_secs``), a summary thread (``save_summaries_secs``) and a step-rate thread;
``stop()`` requests them to stop, joins them and closes the summary writer. ``managed_session``
adds the retry-on-failure loop the reference lacks: if the ps becomes unreachable the session
is rebuilt and, on the chief, the last checkpoint restored.
"""
from __future__ import annotations

import contextlib
import os
import threading
import time
from typing import Any, Callable, List, Optional

import numpy as np

from .. import summary as summary_lib
from ..client.session import ConfigProto, Session
from ..client.variable_store import FailedPreconditionError
from ..framework import ops as F
from ..framework.graph import Graph, GraphKeys, Operation, Tensor, get_default_graph
from ..framework.variables import (global_variables, global_variables_initializer,
                                   local_variables_initializer, report_uninitialized_variables)
from .saver import Saver, latest_checkpoint


class Coordinator:
    """Cooperative stop flag + thread join (what Supervisor's services run under)."""

    def __init__(self):
        self._stop = threading.Event()
        self._threads: List[threading.Thread] = []
        self._exc: Optional[BaseException] = None

    def request_stop(self, ex: Optional[BaseException] = None) -> None:
        if ex is not None and self._exc is None:
            self._exc = ex
        self._stop.set()

    def should_stop(self) -> bool:
        return self._stop.is_set()

    def wait_for_stop(self, timeout: Optional[float] = None) -> bool:
        return self._stop.wait(timeout)

    def clear_stop(self) -> None:
        self._stop.clear()
        self._exc = None

    def register_thread(self, thread: threading.Thread) -> None:
        self._threads.append(thread)

    def join(self, threads: Optional[List[threading.Thread]] = None,
             stop_grace_period_secs: float = 120.0) -> None:
        deadline = time.monotonic() + stop_grace_period_secs
        for t in (threads if threads is not None else self._threads):
            t.join(max(0.0, deadline - time.monotonic()))
        stragglers = [t.name for t in (threads or self._threads) if t.is_alive()]
        self._threads = [t for t in self._threads if t.is_alive()]
        if self._exc is not None:
            exc, self._exc = self._exc, None
            raise exc
        if stragglers:
            raise RuntimeError("Coordinator stopped with threads still running: %s"
                               % " ".join(stragglers))

    @contextlib.contextmanager
    def stop_on_exception(self):
        try:
            yield
        except BaseException as e:  # noqa: BLE001
            self.request_stop(e)


class SessionManager:
    def __init__(self, local_init_op: Optional[Operation] = None, ready_op: Optional[Tensor] = None,
                 graph: Optional[Graph] = None, recovery_wait_secs: float = 30.0):
        self._local_init_op = local_init_op
        self._ready_op = ready_op
        self._graph = graph or get_default_graph()
        self._recovery_wait_secs = recovery_wait_secs

    def _model_ready(self, sess: Session):
        if self._ready_op is None:
            return True, None
        try:
            missing = sess.run(self._ready_op)
        except FailedPreconditionError as e:
            return False, str(e)
        if len(missing) == 0:
            return True, None
        return False, "Variables not initialized: " + ", ".join(
            m.decode() if isinstance(m, bytes) else str(m) for m in missing)

    def prepare_session(self, master: str, init_op: Optional[Operation] = None,
                        saver: Optional[Saver] = None, checkpoint_dir: Optional[str] = None,
                        config: Optional[ConfigProto] = None, init_feed_dict=None,
                        init_fn: Optional[Callable] = None) -> Session:
        """Chief path: create, restore-or-init, verify ready."""
        sess = Session(master, graph=self._graph, config=config)
        restored = False
        if saver is not None and checkpoint_dir:
            ckpt = latest_checkpoint(checkpoint_dir)
            if ckpt:
                saver.restore(sess, ckpt)
                restored = True
        if not restored:
            if init_op is None and init_fn is None:
                ready, msg = self._model_ready(sess)
                if not ready:
                    raise RuntimeError("Model is not initialized and no init_op or init_fn or "
                                       "checkpoint was given: %s" % msg)
            if init_op is not None:
                sess.run(init_op, feed_dict=init_feed_dict)
            if init_fn is not None:
                init_fn(sess)
        if self._local_init_op is not None:
            sess.run(self._local_init_op)
        self._publish_initial_state(sess)
        ready, msg = self._model_ready(sess)
        if not ready:
            raise RuntimeError("Init operations did not make model ready: %s" % msg)
        return sess

    def wait_for_session(self, master: str, config: Optional[ConfigProto] = None,
                         max_wait_secs: float = float("inf")) -> Session:
        """Non-chief path: poll until the chief has initialised the model."""
        deadline = time.monotonic() + max_wait_secs
        sess = Session(master, graph=self._graph, config=config)
        runtime = sess._runtime
        if runtime is not None and getattr(runtime, "strategy", "") in (
                "allreduce_sync", "reduce_scatter_sync"):
            # collective runtimes: receive the chief's broadcast instead of polling
            self._publish_initial_state(sess)
            if self._local_init_op is not None:
                sess.run(self._local_init_op)
            return sess
        while True:
            ready, msg = self._model_ready(sess)
            if ready:
                if self._local_init_op is not None:
                    sess.run(self._local_init_op)
                return sess
            if time.monotonic() + self._recovery_wait_secs > deadline:
                sess.close()
                raise TimeoutError("Session was not ready after waiting %s secs: %s"
                                   % (max_wait_secs, msg))
            time.sleep(self._recovery_wait_secs)

    def _publish_initial_state(self, sess: Session) -> None:
        runtime = sess._runtime
        if runtime is not None:
            sess._shard_plan()      # every rank knows the variable layouts before the exchange
            # trainable variables first: the flat arena a collective runtime builds in this order
            # then holds every gradient in its leading span, which is all its all-reduce moves
            variables = self._graph.get_collection(GraphKeys.GLOBAL_VARIABLES)
            names = [v.op.name for v in variables if v.trainable] + \
                    [v.op.name for v in variables if not v.trainable]
            runtime.after_init(sess, names)


class Supervisor:
    USE_DEFAULT = object()

    def __init__(self, graph: Optional[Graph] = None, ready_op=USE_DEFAULT, is_chief: bool = True,
                 init_op=USE_DEFAULT, init_feed_dict=None, local_init_op=USE_DEFAULT,
                 logdir: Optional[str] = None, summary_op=USE_DEFAULT, saver=USE_DEFAULT,
                 global_step=USE_DEFAULT, save_summaries_secs: float = 120.0,
                 save_model_secs: float = 600.0, recovery_wait_secs: float = 30.0,
                 stop_grace_secs: float = 120.0, checkpoint_basename: str = "model.ckpt",
                 session_manager: Optional[SessionManager] = None, summary_writer=USE_DEFAULT,
                 init_fn: Optional[Callable] = None):
        self._graph = graph or get_default_graph()
        self._is_chief = is_chief
        self._logdir = logdir
        self._save_summaries_secs = save_summaries_secs
        self._save_model_secs = save_model_secs
        self._recovery_wait_secs = recovery_wait_secs
        self._stop_grace_secs = stop_grace_secs
        self._init_feed_dict = init_feed_dict
        self._init_fn = init_fn
        self._coord = Coordinator()
        self._save_lock = threading.Lock()
        self._save_path = os.path.join(logdir, checkpoint_basename) if logdir else None
        self._summary_writer = None
        self._started_threads: List[threading.Thread] = []
        with self._graph.as_default():
            self._ready_op = report_uninitialized_variables() \
                if ready_op is Supervisor.USE_DEFAULT else ready_op
            self._init_op = global_variables_initializer() \
                if init_op is Supervisor.USE_DEFAULT else init_op
            self._local_init_op = local_variables_initializer() \
                if local_init_op is Supervisor.USE_DEFAULT else local_init_op
            if saver is Supervisor.USE_DEFAULT:
                savers = self._graph.get_collection(GraphKeys.SAVERS)
                saver = savers[0] if savers else (Saver() if global_variables() else None)
            self._saver = saver
            if summary_op is Supervisor.USE_DEFAULT:
                summary_op = summary_lib.merge_all()
            self._summary_op = summary_op
            if global_step is Supervisor.USE_DEFAULT:
                steps = self._graph.get_collection(GraphKeys.GLOBAL_STEP)
                global_step = steps[0] if steps else None
            self._global_step = global_step
        self._summary_writer_arg = summary_writer
        self._session_manager = session_manager or SessionManager(
            local_init_op=self._local_init_op, ready_op=self._ready_op, graph=self._graph,
            recovery_wait_secs=recovery_wait_secs)
        self._graph.finalize()     # like TF: the graph is frozen once supervised

    # -- accessors -----------------------------------------------------------------------
    @property
    def is_chief(self) -> bool:
        return self._is_chief

    @property
    def coord(self) -> Coordinator:
        return self._coord

    @property
    def saver(self) -> Optional[Saver]:
        return self._saver

    @property
    def global_step(self):
        return self._global_step

    @property
    def init_op(self):
        return self._init_op

    @property
    def ready_op(self):
        return self._ready_op

    @property
    def summary_op(self):
        return self._summary_op

    @property
    def summary_writer(self):
        return self._summary_writer

    @property
    def save_path(self) -> Optional[str]:
        return self._save_path

    @property
    def session_manager(self) -> SessionManager:
        return self._session_manager

    # -- session acquisition ---------------------------------------------------------------
    def prepare_or_wait_for_session(self, master: str = "", config: Optional[ConfigProto] = None,
                                    wait_for_checkpoint: bool = False,
                                    max_wait_secs: float = 7200.0,
                                    start_standard_services: bool = True) -> Session:
        self._coord.clear_stop()
        if self._is_chief:
            sess = self._session_manager.prepare_session(
                master, init_op=self._init_op, saver=self._saver, checkpoint_dir=self._logdir,
                config=config, init_feed_dict=self._init_feed_dict, init_fn=self._init_fn)
            if start_standard_services:
                self.start_standard_services(sess)
        else:
            sess = self._session_manager.wait_for_session(master, config=config,
                                                          max_wait_secs=max_wait_secs)
        return sess

    def start_standard_services(self, sess: Session) -> List[threading.Thread]:
        """Chief only, and only with a ``logdir``: checkpoint / summary / step-rate threads."""
        if not self._is_chief or not self._logdir:
            return []
        os.makedirs(self._logdir, exist_ok=True)
        if self._summary_writer is None and self._summary_writer_arg is Supervisor.USE_DEFAULT:
            self._summary_writer = summary_lib.FileWriter(self._logdir)
        elif self._summary_writer_arg is not Supervisor.USE_DEFAULT:
            self._summary_writer = self._summary_writer_arg
        threads = []
        sharded = getattr(getattr(sess, "_runtime", None), "has_sharded_storage", None)
        if sharded is not None and sharded() and self._save_model_secs:
            # reading a variable that is stored as per-rank blocks is a collective: the chief's
            # background thread cannot do it alone. Call Saver.save from every worker instead.
            import warnings
            warnings.warn("variables are stored sharded: the chief-only checkpoint service is "
                          "off; call saver.save(...) on every worker")
        elif self._saver is not None and self._save_model_secs:
            threads.append(_Looper(self._coord, self._save_model_secs, "sv-checkpoint",
                                   lambda: self._checkpoint(sess)))
            # final checkpoint: taken when the session closes (the reference's
            # `with sv.prepare_or_wait_for_session(...) as sess:` closes it before sv.stop())
            sess.add_close_hook(self._final_checkpoint)
        if self._summary_op is not None and self._save_summaries_secs and self._summary_writer:
            threads.append(_Looper(self._coord, self._save_summaries_secs, "sv-summary",
                                   lambda: self._write_summary(sess)))
        if self._global_step is not None and self._save_summaries_secs and self._summary_writer:
            threads.append(_StepRate(self, sess))
        for t in threads:
            t.start()
            self._coord.register_thread(t)
        self._started_threads.extend(threads)
        return threads

    def _step_value(self, sess: Session) -> Optional[int]:
        if self._global_step is None:
            return None
        return int(np.asarray(sess.run(self._global_step)).item())

    def _checkpoint(self, sess: Session) -> None:
        if self._saver is not None and self._save_path and not sess._closed:
            with self._save_lock:
                self._saver.save(sess, self._save_path, global_step=self._step_value(sess))

    def _final_checkpoint(self, sess: Session) -> None:
        try:
            self._checkpoint(sess)
        except Exception:  # noqa: BLE001 -- the cluster may already be gone at shutdown
            pass

    def _write_summary(self, sess: Session) -> None:
        # summaries that need fed placeholders cannot be computed by a background thread;
        # they are skipped rather than failing the service (the training loop writes them)
        try:
            value = sess.run(self._summary_op)
        except RuntimeError:
            return
        self._summary_writer.add_summary(value, self._step_value(sess))

    def summary_computed(self, sess: Session, summary, global_step=None) -> None:
        if self._summary_writer is None:
            raise RuntimeError("Writing a summary requires a summary writer (set logdir).")
        if global_step is None and self._global_step is not None:
            global_step = self._step_value(sess)
        self._summary_writer.add_summary(summary, global_step)

    # -- shutdown --------------------------------------------------------------------------
    def request_stop(self, ex: Optional[BaseException] = None) -> None:
        self._coord.request_stop(ex)

    def should_stop(self) -> bool:
        return self._coord.should_stop()

    def stop(self, threads=None, close_summary_writer: bool = True) -> None:
        self._coord.request_stop()
        try:
            self._coord.join(threads, stop_grace_period_secs=self._stop_grace_secs)
        finally:
            self._started_threads = []
            if close_summary_writer and self._summary_writer is not None:
                self._summary_writer.close()
                self._summary_writer = None

    # -- managed session with recovery -----------------------------------------------------
    @contextlib.contextmanager
    def managed_session(self, master: str = "", config: Optional[ConfigProto] = None,
                        start_standard_services: bool = True, close_summary_writer: bool = True):
        sess = self.prepare_or_wait_for_session(master, config,
                                                start_standard_services=start_standard_services)
        try:
            yield sess
        except Exception as e:  # noqa: BLE001
            self.request_stop(e)
        finally:
            try:
                self.stop(close_summary_writer=close_summary_writer)
            finally:
                sess.close()

    def run_with_recovery(self, master: str, step_fn: Callable[[Session], bool],
                          config: Optional[ConfigProto] = None, max_restarts: int = 3) -> int:
        """Call ``step_fn(sess)`` until it returns False; when the cluster becomes unavailable
        rebuild the session (chief: restore the last checkpoint) and continue. Returns the
        number of recoveries performed."""
        from ..parallel.comm import CommError
        from ..parallel.ps_transport import UnavailableError
        restarts = 0
        while True:
            sess = self.prepare_or_wait_for_session(master, config)
            try:
                while step_fn(sess):
                    pass
                return restarts
            except (UnavailableError, CommError, FailedPreconditionError):
                restarts += 1
                if restarts > max_restarts:
                    raise
                time.sleep(min(self._recovery_wait_secs, 1.0))
            finally:
                try:
                    self.stop()
                except (RuntimeError, UnavailableError, CommError, FailedPreconditionError):
                    pass            # a service thread hit the same outage
                sess.close()


class _Looper(threading.Thread):
    def __init__(self, coord: Coordinator, interval: float, name: str, fn: Callable[[], None]):
        super().__init__(name=name, daemon=True)
        self._coord, self._interval, self._fn = coord, interval, fn

    def run(self) -> None:
        from ..client.session import SessionClosedError
        with self._coord.stop_on_exception():
            while not self._coord.wait_for_stop(self._interval):
                try:
                    self._fn()
                except SessionClosedError:
                    return          # the job closed its session while this round was under way


class _StepRate(threading.Thread):
    """Writes ``global_step/sec`` like the step-counter service of a supervised job."""

    def __init__(self, sv: Supervisor, sess: Session):
        super().__init__(name="sv-steprate", daemon=True)
        self._sv, self._sess = sv, sess

    def run(self) -> None:
        from ..client.session import SessionClosedError
        sv = self._sv
        with sv.coord.stop_on_exception():
            try:
                self._loop()
            except SessionClosedError:
                return              # the job closed its session while this round was under way

    def _loop(self) -> None:
        sv = self._sv
        last_t, last_step = time.time(), sv._step_value(self._sess)
        while not sv.coord.wait_for_stop(sv._save_summaries_secs):
            now, step = time.time(), sv._step_value(self._sess)
            rate = (step - last_step) / max(now - last_t, 1e-9)
            if sv.summary_writer is not None:
                sv.summary_writer.add_summary(
                    summary_lib.encode_scalar_summary("global_step/sec", rate), step)
            last_t, last_step = now, step
