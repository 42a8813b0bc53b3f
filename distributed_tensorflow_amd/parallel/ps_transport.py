"""Parameter-server data plane for the CPU plumbing configuration (1 ps + N workers on
localhost or a LAN, no GPU required).

In the reference each worker step makes >= 3 gRPC round trips to the ps: ``RecvTensor`` for
W1/W2/b1/b2 (318 KB), the gradient push (318 KB) that the ps applies with
``ApplyGradientDescent``, and the ``global_step`` read (``tf_distributed.py:111``; placement
at ``:34-36``). This module keeps those semantics -- variables live only in the ps task,
workers update them asynchronously and without locks -- but batches each direction into ONE
framed message per ps task per step:

    step start : PULL  [names]                 -> values         (variable read)
    step end   : APPLY [(name, lr, grad)...]   -> ack            (gradient push + SGD on the ps)
                 + [(name, delta)...]                           (counter increments)

Wire format: ``b"DTF1" | u8 op | u32 header_len | u64 payload_len | header JSON | payload``;
the payload is the raw little-endian tensor bytes, back to back, described by the header.
The server is a thread-per-connection TCP server; one connection per worker, kept open.

``PsService`` is what ``Server(job_name="ps")`` runs; ``RemoteVariableStore`` is the worker's
view of one ps task and plugs into the session like a local store.
"""
from __future__ import annotations

import ctypes
import json
import socket
import struct
import threading
import time
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ..client.variable_store import FailedPreconditionError, LocalVariableStore

_MAGIC = b"DTF1"
_HDR = struct.Struct("<4sBIQ")
OP_PULL, OP_ASSIGN, OP_APPLY, OP_UNINIT, OP_SHUTDOWN, OP_NAMES, OP_PING, OP_DONE = 1, 2, 3, 4, 5, 6, 7, 8
OP_OK, OP_ERR = 100, 101

_NP = {"float32": np.float32, "float64": np.float64, "int32": np.int32, "int64": np.int64,
       "bool": np.bool_, "float16": np.float16}


class UnavailableError(ConnectionError):
    """The ps task cannot be reached (not started yet, crashed, or shut down)."""


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray(n)
    view = memoryview(buf)
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError("peer closed the connection")
        got += k
    return bytes(buf)


def _send_msg(sock: socket.socket, op: int, header: Dict[str, Any],
              arrays: Sequence[np.ndarray] = ()) -> None:
    hdr = json.dumps(header).encode()
    payload = [np.ascontiguousarray(a) for a in arrays]
    total = sum(a.nbytes for a in payload)
    sock.sendall(_HDR.pack(_MAGIC, op, len(hdr), total) + hdr)
    for a in payload:
        if a.nbytes:
            sock.sendall(memoryview(a).cast("B"))


def _recv_msg(sock: socket.socket) -> Tuple[int, Dict[str, Any], bytes]:
    magic, op, hlen, plen = _HDR.unpack(_recv_exact(sock, _HDR.size))
    if magic != _MAGIC:
        raise ConnectionError("bad frame magic %r" % (magic,))
    header = json.loads(_recv_exact(sock, hlen)) if hlen else {}
    payload = _recv_exact(sock, plen) if plen else b""
    return op, header, payload


def _describe(arrays: Sequence[np.ndarray]) -> List[Dict[str, Any]]:
    return [{"dtype": str(a.dtype), "shape": list(a.shape)} for a in arrays]


def _split_payload(specs: Sequence[Dict[str, Any]], payload: bytes) -> List[np.ndarray]:
    out, off = [], 0
    for spec in specs:
        dt = np.dtype(_NP[spec["dtype"]])
        count = int(np.prod(spec["shape"])) if spec["shape"] else 1
        nbytes = count * dt.itemsize
        out.append(np.frombuffer(payload, dtype=dt, count=count, offset=off)
                   .reshape(spec["shape"]).copy())
        off += nbytes
    return out


def _to_numpy(t: Any) -> np.ndarray:
    if isinstance(t, torch.Tensor):
        t = t.detach()
        if t.dtype == torch.bfloat16:
            t = t.float()
        return t.cpu().numpy()
    return np.asarray(t)


# ---------------------------------------------------------------------------------------
# server side
# ---------------------------------------------------------------------------------------
class PsService:
    """Variable host of one ``/job:ps/task:K``. State lives in a ``LocalVariableStore`` on the
    CPU; updates from different workers interleave without locking (asynchronous SGD)."""

    def __init__(self, host: str, port: int, store: Optional[LocalVariableStore] = None,
                 expected_workers: int = 0):
        self.store = store or LocalVariableStore(torch.device("cpu"))
        self.expected_workers = expected_workers   # DONE reports that end the service (0 = off)
        self._done_workers: set = set()
        self._conns: List[socket.socket] = []
        self._sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self._sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self._sock.bind((host if host not in ("localhost",) else "127.0.0.1", port))
        self._sock.listen(64)
        self.port = self._sock.getsockname()[1]
        self._stop = threading.Event()
        self._threads: List[threading.Thread] = []
        self._accept_thread = threading.Thread(target=self._accept_loop, daemon=True,
                                               name="ps-accept:%d" % self.port)
        self.requests_served = 0
        self.bytes_in = 0
        self.bytes_out = 0

    def start(self) -> None:
        self._accept_thread.start()

    def _accept_loop(self) -> None:
        self._sock.settimeout(0.25)
        while not self._stop.is_set():
            try:
                conn, _ = self._sock.accept()
            except socket.timeout:
                continue
            except OSError:
                break
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self._conns.append(conn)
            t = threading.Thread(target=self._serve, args=(conn,), daemon=True)
            t.start()
            self._threads.append(t)
        try:
            self._sock.close()
        except OSError:
            pass

    def _serve(self, conn: socket.socket) -> None:
        try:
            while not self._stop.is_set():
                try:
                    op, header, payload = _recv_msg(conn)
                except (ConnectionError, OSError):
                    return
                self.requests_served += 1
                self.bytes_in += len(payload)
                try:
                    try:
                        reply_header, arrays = self._handle(op, header, payload)
                        self.bytes_out += sum(a.nbytes for a in arrays)
                        _send_msg(conn, OP_OK, reply_header, arrays)
                    except (ConnectionError, OSError):
                        raise
                    except Exception as e:  # noqa: BLE001 -- report to the client, keep serving
                        _send_msg(conn, OP_ERR, {"type": type(e).__name__, "message": str(e)})
                except (ConnectionError, OSError):
                    return          # the peer (or stop()) closed the connection mid-reply
                if op == OP_SHUTDOWN:
                    return
        finally:
            conn.close()

    def _handle(self, op: int, header: Dict[str, Any], payload: bytes):
        store = self.store
        if op == OP_PULL:
            arrays = [_to_numpy(store.read(n)) for n in header["names"]]
            return {"specs": _describe(arrays)}, arrays
        if op == OP_ASSIGN:
            for name, arr in zip(header["names"], _split_payload(header["specs"], payload)):
                store.assign(name, torch.from_numpy(arr))
            return {}, []
        if op == OP_APPLY:
            arrays = _split_payload(header["specs"], payload)
            k = 0
            for name, lr in header.get("sgd", []):
                store.apply_sgd(name, lr, torch.from_numpy(arrays[k]))
                k += 1
            for name in header.get("add", []):
                store.assign_add(name, torch.from_numpy(arrays[k]))
                k += 1
            want = header.get("return", [])
            out = [_to_numpy(store.read(n)) for n in want]
            return {"specs": _describe(out)}, out
        if op == OP_UNINIT:
            return {"names": store.uninitialized(header["names"])}, []
        if op == OP_NAMES:
            return {"names": store.names()}, []
        if op == OP_PING:
            return {"updates": store.update_count}, []
        if op == OP_SHUTDOWN:
            self._stop.set()
            return {}, []
        if op == OP_DONE:
            self._done_workers.add(header.get("worker"))
            if self.expected_workers and len(self._done_workers) >= self.expected_workers:
                self._stop.set()
            return {"done": len(self._done_workers)}, []
        raise ValueError("unknown ps op %d" % op)

    def stop(self) -> None:
        """Stop accepting and serving; unblocks and joins the connection threads so the process
        can exit cleanly."""
        self._stop.set()
        for conn in list(self._conns):
            try:
                conn.shutdown(socket.SHUT_RDWR)
            except OSError:
                pass
        me = threading.current_thread()
        for t in list(self._threads) + [self._accept_thread]:
            if t is not me and t.is_alive():
                t.join(2.0)

    def wait(self, timeout: Optional[float] = None) -> bool:
        """Block until ``stop()`` / a remote SHUTDOWN; True when stopped."""
        return self._stop.wait(timeout)

    @property
    def stopped(self) -> bool:
        return self._stop.is_set()


class _NativeStoreView:
    """Read access, for the hosting process, to the variables a ``NativePsService`` holds
    (writes from this process go through the wire like anybody else's)."""

    def __init__(self, service: "NativePsService"):
        self._svc = service
        self.device = torch.device("cpu")

    @property
    def update_count(self) -> int:
        return int(self._svc._lib.dtf_ps_updates(self._svc._handle)) if self._svc._handle else 0

    def names(self) -> List[str]:
        svc = self._svc
        need = svc._lib.dtf_ps_names(svc._handle, None, 0)
        if need <= 0:
            return []
        buf = ctypes.create_string_buffer(need + 1)
        svc._lib.dtf_ps_names(svc._handle, buf, need + 1)
        return sorted(n for n in buf.value.decode().split("\n") if n)

    def is_initialized(self, name: str) -> bool:
        svc = self._svc
        return svc._lib.dtf_ps_read(svc._handle, name.encode(), None, 0, None, 0) >= 0

    def uninitialized(self, names: Iterable[str]) -> List[str]:
        return [n for n in names if not self.is_initialized(n)]

    def read(self, name: str) -> torch.Tensor:
        svc = self._svc
        meta = ctypes.create_string_buffer(256)
        size = svc._lib.dtf_ps_read(svc._handle, name.encode(), None, 0, meta, 256)
        if size < 0:
            raise FailedPreconditionError("Attempting to use uninitialized value %s" % name)
        dtype, _, dims = meta.value.decode().partition(":")
        shape = [int(d) for d in dims.split(",") if d]
        raw = ctypes.create_string_buffer(max(1, size))
        svc._lib.dtf_ps_read(svc._handle, name.encode(), raw, size, None, 0)
        arr = np.frombuffer(raw.raw[:size], dtype=np.dtype(_NP[dtype])).reshape(shape).copy()
        return torch.from_numpy(arr)

    def local_tensor(self, name: str) -> None:
        return None

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {n: self.read(n) for n in self.names()}

    def _client(self) -> "RemoteVariableStore":
        if self._svc._loopback is None:
            self._svc._loopback = RemoteVariableStore("127.0.0.1:%d" % self._svc.port)
        return self._svc._loopback

    def assign(self, name: str, value: torch.Tensor) -> torch.Tensor:
        return self._client().assign(name, value)


class NativePsService:
    """The ps task served by the C++ runtime (``csrc/runtime/ps_server.cc``): same wire protocol
    and the same Python-side interface as ``PsService``, with the connection threads, the
    variable table and the SGD application outside the interpreter."""

    def __init__(self, host: str, port: int, expected_workers: int = 0):
        from ..utils import native
        lib = native.load()
        if lib is None or not hasattr(lib, "dtf_ps_start"):
            raise RuntimeError("native host library with the ps server is not built")
        self._lib = lib
        self._host, self._req_port = host, port
        self.expected_workers = expected_workers
        self._handle = lib.dtf_ps_start((host or "127.0.0.1").encode(), int(port),
                                        int(expected_workers))
        if not self._handle:
            raise OSError(lib.dtf_ps_last_error().decode())
        self.port = lib.dtf_ps_port(self._handle)
        self._loopback: Optional["RemoteVariableStore"] = None
        self._stopped_flag = False
        self.store = _NativeStoreView(self)

    def start(self) -> None:
        pass                       # serving since construction

    @property
    def requests_served(self) -> int:
        return int(self._lib.dtf_ps_requests(self._handle)) if self._handle else 0

    def wait(self, timeout: Optional[float] = None) -> bool:
        if not self._handle:
            return True
        ms = -1 if timeout is None else int(timeout * 1000)
        return bool(self._lib.dtf_ps_wait(self._handle, ms))

    @property
    def stopped(self) -> bool:
        return self._stopped_flag or not self._handle or \
            bool(self._lib.dtf_ps_stopped(self._handle))

    def stop(self) -> None:
        if self._loopback is not None:
            self._loopback.close()
            self._loopback = None
        handle, self._handle = self._handle, None
        self._stopped_flag = True
        if handle:
            self._lib.dtf_ps_stop(handle)


def make_ps_service(host: str, port: int, expected_workers: int = 0):
    """The C++ ps task when the host library is built (``DTF_PS_NATIVE=0`` forces the Python
    one), else the Python implementation."""
    import os
    if os.environ.get("DTF_PS_NATIVE", "1") != "0":
        try:
            return NativePsService(host, port, expected_workers)
        except RuntimeError:
            pass
    svc = PsService(host, port, expected_workers=expected_workers)
    return svc


# ---------------------------------------------------------------------------------------
# worker side
# ---------------------------------------------------------------------------------------
class PsClient:
    def __init__(self, address: str, connect_timeout_s: float = 60.0, retry_s: float = 0.2):
        host, _, port = address.rpartition(":")
        self.address = address
        self._host = "127.0.0.1" if host in ("localhost", "") else host
        self._port = int(port)
        self._timeout = connect_timeout_s
        self._retry = retry_s
        self._sock: Optional[socket.socket] = None
        self._lock = threading.Lock()

    def _connect(self) -> socket.socket:
        deadline = time.monotonic() + self._timeout
        last: Optional[Exception] = None
        while time.monotonic() < deadline:
            try:
                s = socket.create_connection((self._host, self._port), timeout=5.0)
                s.settimeout(None)
                s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                return s
            except OSError as e:
                last = e
                time.sleep(self._retry)
        raise UnavailableError("cannot reach ps task at %s: %s" % (self.address, last))

    def call(self, op: int, header: Dict[str, Any], arrays: Sequence[np.ndarray] = ()):
        with self._lock:
            if self._sock is None:
                self._sock = self._connect()
            try:
                _send_msg(self._sock, op, header, arrays)
                rop, rheader, payload = _recv_msg(self._sock)
            except (ConnectionError, OSError) as e:
                try:
                    self._sock.close()
                finally:
                    self._sock = None
                raise UnavailableError("ps task at %s went away: %s" % (self.address, e))
        if rop == OP_ERR:
            if rheader.get("type") == "FailedPreconditionError":
                raise FailedPreconditionError(rheader.get("message", ""))
            raise RuntimeError("ps task %s: %s: %s" % (self.address, rheader.get("type"),
                                                       rheader.get("message")))
        return rheader, payload

    def close(self) -> None:
        with self._lock:
            if self._sock is not None:
                try:
                    self._sock.close()
                finally:
                    self._sock = None


class RemoteVariableStore:
    """A worker's view of the variables hosted by one ps task.

    Reads within a step come from the snapshot pulled at ``begin_step`` (one PULL for all the
    variables the step touches). Updates are queued and shipped by ``end_step`` in one APPLY;
    the ps applies them on arrival, unsynchronised with other workers."""

    def __init__(self, address: str, device: torch.device = torch.device("cpu"),
                 connect_timeout_s: float = 60.0):
        self.client = PsClient(address, connect_timeout_s)
        self.device = torch.device(device)
        # step state is per thread: the training loop and the Supervisor's service threads
        # (checkpoint, summaries) run steps through the same store concurrently
        self._tls = threading.local()
        self.update_count = 0
        self.bytes_pulled = 0
        self.bytes_pushed = 0

    @property
    def _cache(self) -> Dict[str, torch.Tensor]:
        try:
            return self._tls.cache
        except AttributeError:
            self._tls.cache = {}
            return self._tls.cache

    @property
    def _sgd(self) -> List[Tuple[str, float, np.ndarray]]:
        try:
            return self._tls.sgd
        except AttributeError:
            self._tls.sgd = []
            return self._tls.sgd

    @_sgd.setter
    def _sgd(self, value) -> None:
        self._tls.sgd = value

    @property
    def _add(self) -> List[Tuple[str, np.ndarray]]:
        try:
            return self._tls.add
        except AttributeError:
            self._tls.add = []
            return self._tls.add

    @_add.setter
    def _add(self, value) -> None:
        self._tls.add = value

    # -- step bracket --------------------------------------------------------------------
    def begin_step(self, names: Sequence[str]) -> None:
        self._cache.clear()
        if names:
            self._pull(list(names))

    def end_step(self) -> None:
        self._cache.clear()
        if not self._sgd and not self._add:
            return
        arrays = [g for _, _, g in self._sgd] + [d for _, d in self._add]
        header = {"sgd": [(n, lr) for n, lr, _ in self._sgd], "add": [n for n, _ in self._add],
                  "specs": _describe(arrays)}
        self._sgd, self._add = [], []
        self.bytes_pushed += sum(a.nbytes for a in arrays)
        self.client.call(OP_APPLY, header, arrays)

    def _pull(self, names: List[str]) -> None:
        header, payload = self.client.call(OP_PULL, {"names": names})
        self.bytes_pulled += len(payload)
        for name, arr in zip(names, _split_payload(header["specs"], payload)):
            self._cache[name] = torch.from_numpy(arr).to(self.device)

    # -- store interface -----------------------------------------------------------------
    def read(self, name: str) -> torch.Tensor:
        if name not in self._cache:
            self._pull([name])
        return self._cache[name]

    def local_tensor(self, name: str) -> None:
        return None       # state is remote: in-place GPU updates do not apply

    def is_initialized(self, name: str) -> bool:
        return not self.uninitialized([name])

    def uninitialized(self, names: Iterable[str]) -> List[str]:
        header, _ = self.client.call(OP_UNINIT, {"names": list(names)})
        return header["names"]

    def names(self) -> List[str]:
        header, _ = self.client.call(OP_NAMES, {})
        return header["names"]

    def assign(self, name: str, value: torch.Tensor) -> torch.Tensor:
        arr = _to_numpy(value)
        self.client.call(OP_ASSIGN, {"names": [name], "specs": _describe([arr])}, [arr])
        self._cache[name] = value.detach().to(self.device)
        return self._cache[name]

    def assign_many(self, values: Dict[str, torch.Tensor]) -> None:
        names = list(values)
        arrays = [_to_numpy(values[n]) for n in names]
        self.client.call(OP_ASSIGN, {"names": names, "specs": _describe(arrays)}, arrays)

    def assign_add(self, name: str, delta) -> torch.Tensor:
        cur = self.read(name)
        d = _to_numpy(delta).astype(_to_numpy(cur).dtype, copy=False)
        self._add.append((name, d))
        return cur + torch.as_tensor(d).to(cur.device)

    def apply_sgd(self, name: str, lr: float, grad: torch.Tensor) -> torch.Tensor:
        cur = self.read(name)
        self._sgd.append((name, float(lr), _to_numpy(grad)))
        self.update_count += 1
        return cur - float(lr) * grad.to(cur.device)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        names = self.names()
        header, payload = self.client.call(OP_PULL, {"names": names})
        return {n: torch.from_numpy(a) for n, a in
                zip(names, _split_payload(header["specs"], payload))}

    def ping(self) -> int:
        header, _ = self.client.call(OP_PING, {})
        return int(header["updates"])

    def report_done(self, worker: int) -> int:
        """Tell the ps this worker has finished; the ps leaves ``join()`` once every worker of
        the cluster has reported. Returns how many have reported so far."""
        header, _ = self.client.call(OP_DONE, {"worker": int(worker)})
        return int(header["done"])

    def shutdown_server(self) -> None:
        try:
            self.client.call(OP_SHUTDOWN, {})
        except UnavailableError:
            pass
        self.client.close()

    def close(self) -> None:
        self.client.close()
