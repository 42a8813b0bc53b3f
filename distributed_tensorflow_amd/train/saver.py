"""Saver: checkpoints in the Saver-V2 "tensor bundle" layout.

The reference never saves -- its ``Supervisor`` is built without ``logdir``
(``tf_distributed.py:92``), so the default ``Saver`` it creates is inert and training always
restarts from the seed-1 initial values. The facade still provides the real thing so that
``Supervisor(logdir=...)`` / ``Saver.save`` / ``Saver.restore`` work and recovery after a
failure has something to restore from.

A checkpoint with prefix ``P`` is

* ``P.index``                  -- SSTable (``utils.sstable``): key ``""`` ->
  ``BundleHeaderProto{num_shards, endianness=LITTLE, version{producer=1}}``; key ``<variable
  name>`` -> ``BundleEntryProto{dtype, shape, shard_id, offset, size, crc32c}``;
* ``P.data-0000k-of-0000N``    -- raw little-endian tensor bytes back to back, in key order;
* ``checkpoint``               -- text proto naming the latest and all kept prefixes.

Keys are the variable op names (``global_step``, ``weights/Variable``, ...), which is what
makes checkpoints portable between the strategies: parameter-server state, replicated state and
the owner-sharded flat arena all save to and restore from the same keys. With ``sharded=True``
each store (each ps task) writes its own data shard, so nothing is gathered to one place.
"""
from __future__ import annotations

import os
import re
import struct
import time
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ..framework import dtypes
from ..framework.graph import GraphKeys, get_default_graph
from ..framework.variables import Variable, global_variables
from ..utils import crc32c as _crc
from ..utils import protowire as pw
from ..utils import sstable


# ---------------------------------------------------------------------------------------
# protobuf pieces
# ---------------------------------------------------------------------------------------
def _encode_header(num_shards: int) -> bytes:
    # endianness LITTLE = 0 (default, omitted); version = VersionDef{producer = 1}
    return pw.f_varint(1, num_shards) + pw.f_message(3, pw.f_varint(1, 1))


def _encode_shape(shape: Sequence[int]) -> bytes:
    return b"".join(pw.f_message(2, pw.f_varint(1, int(d))) for d in shape)


def _encode_entry(dtype_enum: int, shape: Sequence[int], shard_id: int, offset: int, size: int,
                  crc: int) -> bytes:
    out = pw.f_varint(1, dtype_enum) + pw.f_message(2, _encode_shape(shape))
    if shard_id:
        out += pw.f_varint(3, shard_id)
    if offset:
        out += pw.f_varint(4, offset)
    out += pw.f_varint(5, size) + pw.f_fixed32(6, crc)
    return out


def _decode_entry(data: bytes) -> Dict[str, Any]:
    f = pw.parse(data)
    shape = []
    for dim in pw.parse(f[2][0]).get(2, []) if 2 in f else []:
        shape.append(pw.as_signed64(pw.parse(dim).get(1, [0])[0]))
    return {"dtype": int(f.get(1, [0])[0]), "shape": shape, "shard_id": int(f.get(3, [0])[0]),
            "offset": int(f.get(4, [0])[0]), "size": int(f.get(5, [0])[0]),
            "crc32c": int(f.get(6, [0])[0])}


def _data_path(prefix: str, shard: int, num_shards: int) -> str:
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def _tensor_bytes(t: torch.Tensor) -> Tuple[bytes, dtypes.DType]:
    t = t.detach().cpu().contiguous()
    dt = dtypes.as_dtype(t.dtype)
    if t.dtype == torch.bfloat16:
        raw = t.view(torch.int16).numpy().tobytes()
    elif t.dtype == torch.float8_e4m3fn:
        raw = t.view(torch.uint8).numpy().tobytes()
    else:
        raw = t.numpy().tobytes()
    return raw, dt


def _tensor_from_bytes(raw: bytes, dt: dtypes.DType, shape: Sequence[int]) -> torch.Tensor:
    if dt.torch == torch.bfloat16:
        arr = np.frombuffer(raw, dtype=np.int16).reshape(shape).copy()
        return torch.from_numpy(arr).view(torch.bfloat16)
    if dt.torch == torch.float8_e4m3fn:
        arr = np.frombuffer(raw, dtype=np.uint8).reshape(shape).copy()
        return torch.from_numpy(arr).view(torch.float8_e4m3fn)
    arr = np.frombuffer(raw, dtype=np.dtype(dt.numpy)).reshape(shape).copy()
    return torch.from_numpy(arr)


# ---------------------------------------------------------------------------------------
# bundle writer / reader
# ---------------------------------------------------------------------------------------
def write_bundle(prefix: str, shards: Sequence[Dict[str, torch.Tensor]]) -> None:
    """Write ``prefix.index`` + one data file per element of ``shards`` (name -> tensor)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)) or ".", exist_ok=True)
    num_shards = max(1, len(shards))
    entries: Dict[str, bytes] = {}
    for shard_id in range(num_shards):
        tensors = shards[shard_id] if shard_id < len(shards) else {}
        tmp = _data_path(prefix, shard_id, num_shards) + ".tmp"
        offset = 0
        with open(tmp, "wb") as fh:
            for name in sorted(tensors):
                if name in entries:
                    raise ValueError("variable %r appears in two shards" % name)
                raw, dt = _tensor_bytes(tensors[name])
                fh.write(raw)
                entries[name] = _encode_entry(dt.enum, tuple(tensors[name].shape), shard_id,
                                              offset, len(raw), _crc.masked_crc32c(raw))
                offset += len(raw)
        os.replace(tmp, _data_path(prefix, shard_id, num_shards))
    table = [(b"", _encode_header(num_shards))]
    table += [(name.encode(), entries[name]) for name in sorted(entries, key=lambda s: s.encode())]
    tmp_index = prefix + ".index.tmp"
    sstable.write_table(tmp_index, table)
    os.replace(tmp_index, prefix + ".index")


def read_bundle_index(prefix: str) -> Tuple[int, Dict[str, Dict[str, Any]]]:
    rows = sstable.read_table(prefix + ".index")
    if not rows or rows[0][0] != b"":
        raise IOError("%s.index has no bundle header" % prefix)
    num_shards = int(pw.parse(rows[0][1]).get(1, [1])[0])
    return num_shards, {k.decode(): _decode_entry(v) for k, v in rows[1:]}


def read_bundle(prefix: str, names: Optional[Sequence[str]] = None) -> Dict[str, torch.Tensor]:
    num_shards, index = read_bundle_index(prefix)
    want = list(index) if names is None else list(names)
    out: Dict[str, torch.Tensor] = {}
    handles: Dict[int, Any] = {}
    try:
        for name in want:
            if name not in index:
                raise KeyError("Key %s not found in checkpoint %s" % (name, prefix))
            e = index[name]
            fh = handles.get(e["shard_id"])
            if fh is None:
                fh = handles[e["shard_id"]] = open(_data_path(prefix, e["shard_id"], num_shards),
                                                   "rb")
            fh.seek(e["offset"])
            raw = fh.read(e["size"])
            if len(raw) != e["size"] or _crc.masked_crc32c(raw) != e["crc32c"]:
                raise IOError("checksum mismatch for %s in %s" % (name, prefix))
            out[name] = _tensor_from_bytes(raw, dtypes.from_enum(e["dtype"]), e["shape"])
    finally:
        for fh in handles.values():
            fh.close()
    return out


class CheckpointReader:
    """``tf.train.NewCheckpointReader``: look into a checkpoint without a graph or a session.
    ``prefix`` is a checkpoint prefix or a directory (then its latest checkpoint)."""

    def __init__(self, prefix: str):
        if os.path.isdir(prefix):
            latest = latest_checkpoint(prefix)
            if latest is None:
                raise ValueError("no checkpoint found in directory %s" % prefix)
            prefix = latest
        if not checkpoint_exists(prefix):
            raise ValueError("not a checkpoint: %s" % prefix)
        self._prefix = prefix
        self._num_shards, self._index = read_bundle_index(prefix)

    def has_tensor(self, name: str) -> bool:
        return name in self._index

    def get_variable_to_shape_map(self) -> Dict[str, List[int]]:
        return {k: list(e["shape"]) for k, e in self._index.items()}

    def get_variable_to_dtype_map(self) -> Dict[str, dtypes.DType]:
        return {k: dtypes.from_enum(e["dtype"]) for k, e in self._index.items()}

    def get_tensor(self, name: str) -> np.ndarray:
        """The stored value as a numpy array (bfloat16 and fp8 tensors come back as float32:
        numpy has neither type)."""
        t = read_bundle(self._prefix, [name])[name]
        if t.dtype in (torch.bfloat16, torch.float8_e4m3fn):
            t = t.float()
        return t.numpy()

    def debug_string(self) -> str:
        dt = self.get_variable_to_dtype_map()
        return "".join("%s (%s) %s\n" % (k, dt[k].name, list(e["shape"]))
                       for k, e in sorted(self._index.items()))


def NewCheckpointReader(prefix: str) -> CheckpointReader:  # noqa: N802 -- TensorFlow's name
    return CheckpointReader(prefix)


def list_variables(ckpt_dir_or_file: str) -> List[Tuple[str, List[int]]]:
    """``[(name, shape)]`` of a checkpoint, sorted by name."""
    return sorted(CheckpointReader(ckpt_dir_or_file).get_variable_to_shape_map().items())


def load_variable(ckpt_dir_or_file: str, name: str) -> np.ndarray:
    if name.endswith(":0"):
        name = name[:-2]
    return CheckpointReader(ckpt_dir_or_file).get_tensor(name)


# ---------------------------------------------------------------------------------------
# "checkpoint" state file
# ---------------------------------------------------------------------------------------
def update_checkpoint_state(save_dir: str, latest: str, all_paths: Sequence[str]) -> None:
    lines = ['model_checkpoint_path: "%s"' % os.path.basename(latest)]
    lines += ['all_model_checkpoint_paths: "%s"' % os.path.basename(p) for p in all_paths]
    tmp = os.path.join(save_dir, "checkpoint.tmp")
    with open(tmp, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    os.replace(tmp, os.path.join(save_dir, "checkpoint"))


class CheckpointState(dict):
    """The ``checkpoint`` state file: ``model_checkpoint_path`` and
    ``all_model_checkpoint_paths``, readable as attributes (TensorFlow's proto) or as keys."""

    def __getattr__(self, name: str):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)


def get_checkpoint_state(checkpoint_dir: str) -> Optional[CheckpointState]:
    path = os.path.join(checkpoint_dir, "checkpoint")
    if not os.path.exists(path):
        return None
    latest, all_paths = None, []
    with open(path) as fh:
        for line in fh:
            m = re.match(r'\s*(\w+):\s*"(.*)"\s*$', line)
            if not m:
                continue
            full = m.group(2) if os.path.isabs(m.group(2)) \
                else os.path.join(checkpoint_dir, m.group(2))
            if m.group(1) == "model_checkpoint_path":
                latest = full
            elif m.group(1) == "all_model_checkpoint_paths":
                all_paths.append(full)
    if latest is None:
        return None
    return CheckpointState(model_checkpoint_path=latest, all_model_checkpoint_paths=all_paths)


def latest_checkpoint(checkpoint_dir: str) -> Optional[str]:
    state = get_checkpoint_state(checkpoint_dir)
    if state and os.path.exists(state["model_checkpoint_path"] + ".index"):
        return state["model_checkpoint_path"]
    return None


def checkpoint_exists(prefix: str) -> bool:
    return os.path.exists(prefix + ".index")


# ---------------------------------------------------------------------------------------
# Saver
# ---------------------------------------------------------------------------------------
class Saver:
    def __init__(self, var_list: Optional[Sequence[Variable]] = None, max_to_keep: int = 5,
                 sharded: bool = False, name: Optional[str] = None):
        # ``var_list`` may be a dict {name in the checkpoint: variable} (e.g.
        # ``ExponentialMovingAverage.variables_to_restore()``): the variable is then written and
        # looked up under that name instead of its own
        self._names: Dict[Variable, str] = {}
        if isinstance(var_list, dict):
            self._names = {v: str(k) for k, v in var_list.items()}
            var_list = [var_list[k] for k in sorted(var_list)]
        self._var_list = list(var_list) if var_list is not None else None
        self._max_to_keep = max_to_keep
        self._sharded = sharded
        self._kept: List[str] = []
        # the graph is bound at construction: save() may run on a service thread whose
        # default-graph stack is not the one the model was built under
        self._graph = get_default_graph()
        self._graph.add_to_collection(GraphKeys.SAVERS, self)

    def _variables(self) -> List[Variable]:
        if self._var_list is not None:
            return self._var_list
        return self._graph.get_collection(GraphKeys.GLOBAL_VARIABLES)

    def _key(self, v: Variable) -> str:
        return self._names.get(v, v.op.name)

    @property
    def last_checkpoints(self) -> List[str]:
        return list(self._kept)

    def save(self, sess, save_path: str, global_step=None, write_state: bool = True) -> str:
        """Write the current value of every variable; returns the checkpoint prefix."""
        if global_step is not None:
            if not isinstance(global_step, (int, np.integer)):
                global_step = int(np.asarray(sess.run(global_step)).item())
            prefix = "%s-%d" % (save_path, int(global_step))
        else:
            prefix = save_path
        groups: Dict[int, Tuple[Any, List[Variable]]] = {}
        for v in self._variables():
            store = sess._store_for(v.op)
            groups.setdefault(id(store), (store, []))[1].append(v)
        shards: List[Dict[str, torch.Tensor]] = []
        merged: Dict[str, torch.Tensor] = {}
        for store, variables in groups.values():
            begin = getattr(store, "begin_step", None)
            if begin is not None:
                begin([v.op.name for v in variables])       # one pull per ps task
            # a variable stored as per-rank blocks (sharded products) is all-gathered: saving
            # such a graph is a collective, every worker calls save()
            whole = getattr(getattr(sess, "_runtime", None), "read_variable_full", None)
            values = {self._key(v): (whole(v.op.name) if whole is not None
                                  and store is getattr(sess._runtime, "local_store", None)
                                  else store.read(v.op.name)) for v in variables}
            if self._sharded:
                shards.append(values)
            else:
                merged.update(values)
        write_bundle(prefix, shards if self._sharded else [merged])
        if write_state:
            self._kept = [p for p in self._kept if p != prefix] + [prefix]
            while self._max_to_keep and len(self._kept) > self._max_to_keep:
                self._delete(self._kept.pop(0))
            update_checkpoint_state(os.path.dirname(os.path.abspath(prefix)), prefix, self._kept)
        return prefix

    @staticmethod
    def _delete(prefix: str) -> None:
        directory = os.path.dirname(os.path.abspath(prefix))
        base = os.path.basename(prefix)
        for fname in os.listdir(directory):
            if fname == base + ".index" or fname.startswith(base + ".data-"):
                try:
                    os.remove(os.path.join(directory, fname))
                except OSError:
                    pass

    def restore(self, sess, save_path: str) -> None:
        """Assign every variable from the checkpoint (variables need not be initialized)."""
        if not checkpoint_exists(save_path):
            raise ValueError("The passed save_path is not a valid checkpoint: %s" % save_path)
        variables = self._variables()
        values = read_bundle(save_path, [self._key(v) for v in variables])
        for v in variables:
            t = values[self._key(v)]
            if tuple(t.shape) != tuple(v.shape or ()):
                raise ValueError("shape mismatch for %s: checkpoint %s, variable %s"
                                 % (v.op.name, tuple(t.shape), v.shape))
            store = sess._store_for(v.op)
            assign_whole = getattr(getattr(sess, "_runtime", None), "assign_variable_full", None)
            if assign_whole is not None and store is getattr(sess._runtime, "local_store", None):
                assign_whole(v.op.name, t)       # keeps the local block of a sharded variable
            else:
                store.assign(v.op.name, t.to(getattr(store, "device", "cpu")))
