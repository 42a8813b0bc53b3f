"""Loader for the host-side native runtime library ``libdtf_host.so``.

The library (sources under ``csrc/io`` and ``csrc/runtime``) holds the pieces of the runtime
that are C++ in a production framework and need no GPU: CRC-32C, the TFRecord event-file
writer, the checkpoint bundle (SSTable index + data shard) writer/reader and the TCP
parameter-server data plane. It is built in-tree by ``build_native.py``; when it has not been
built, callers fall back to the pure-Python implementations (same bytes, slower).
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional

_LIB: Optional[ctypes.CDLL] = None
_TRIED = False
_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libdtf_host.so")


_LOAD_LOCK = threading.Lock()


def load() -> Optional[ctypes.CDLL]:
    if _TRIED:
        return _LIB
    with _LOAD_LOCK:             # (threads of one process may ask at the same time)
        return _load_locked()


def _load_locked() -> Optional[ctypes.CDLL]:
    global _LIB, _TRIED
    if _TRIED:
        return _LIB
    if os.environ.get("DTF_DISABLE_NATIVE_HOST") == "1" or not os.path.exists(HOST_LIB_PATH):
        _TRIED = True
        return None
    lib = ctypes.CDLL(HOST_LIB_PATH)
    c = ctypes
    lib.dtf_crc32c_extend.restype = c.c_uint32
    lib.dtf_crc32c_extend.argtypes = [c.c_uint32, c.c_char_p, c.c_size_t]
    lib.dtf_crc32c_hw.restype = c.c_int
    # events writer
    lib.dtf_events_open.restype = c.c_void_p
    lib.dtf_events_open.argtypes = [c.c_char_p]
    lib.dtf_events_write.restype = c.c_int
    lib.dtf_events_write.argtypes = [c.c_void_p, c.c_char_p, c.c_size_t]
    lib.dtf_events_flush.restype = c.c_int
    lib.dtf_events_flush.argtypes = [c.c_void_p]
    lib.dtf_events_close.restype = c.c_int
    lib.dtf_events_close.argtypes = [c.c_void_p]
    # sstable (checkpoint index) builder / reader
    lib.dtf_table_builder_new.restype = c.c_void_p
    lib.dtf_table_builder_new.argtypes = [c.c_char_p]
    lib.dtf_table_builder_add.restype = c.c_int
    lib.dtf_table_builder_add.argtypes = [c.c_void_p, c.c_char_p, c.c_size_t, c.c_char_p,
                                          c.c_size_t]
    lib.dtf_table_builder_finish.restype = c.c_int
    lib.dtf_table_builder_finish.argtypes = [c.c_void_p]
    lib.dtf_table_open.restype = c.c_void_p
    lib.dtf_table_open.argtypes = [c.c_char_p]
    lib.dtf_table_count.restype = c.c_int64
    lib.dtf_table_count.argtypes = [c.c_void_p]
    lib.dtf_table_entry.restype = c.c_int
    lib.dtf_table_entry.argtypes = [c.c_void_p, c.c_int64, c.POINTER(c.c_void_p),
                                    c.POINTER(c.c_size_t), c.POINTER(c.c_void_p),
                                    c.POINTER(c.c_size_t)]
    lib.dtf_table_close.restype = None
    lib.dtf_table_close.argtypes = [c.c_void_p]
    lib.dtf_last_error.restype = c.c_char_p
    # parameter-server task
    if hasattr(lib, "dtf_ps_start"):
        lib.dtf_ps_start.restype = c.c_void_p
        lib.dtf_ps_start.argtypes = [c.c_char_p, c.c_int, c.c_int]
        lib.dtf_ps_last_error.restype = c.c_char_p
        lib.dtf_ps_port.restype = c.c_int
        lib.dtf_ps_port.argtypes = [c.c_void_p]
        lib.dtf_ps_stopped.restype = c.c_int
        lib.dtf_ps_stopped.argtypes = [c.c_void_p]
        lib.dtf_ps_requests.restype = c.c_long
        lib.dtf_ps_requests.argtypes = [c.c_void_p]
        lib.dtf_ps_updates.restype = c.c_long
        lib.dtf_ps_updates.argtypes = [c.c_void_p]
        lib.dtf_ps_wait.restype = c.c_int
        lib.dtf_ps_wait.argtypes = [c.c_void_p, c.c_long]
        lib.dtf_ps_stop.restype = None
        lib.dtf_ps_stop.argtypes = [c.c_void_p]
        lib.dtf_ps_read.restype = c.c_long
        lib.dtf_ps_read.argtypes = [c.c_void_p, c.c_char_p, c.c_void_p, c.c_long, c.c_char_p,
                                    c.c_long]
        lib.dtf_ps_names.restype = c.c_long
        lib.dtf_ps_names.argtypes = [c.c_void_p, c.c_char_p, c.c_long]
    # control-plane rendezvous + native RCCL communicator
    if hasattr(lib, "dtf_rdzv_create"):
        lib.dtf_rdzv_create.restype = c.c_void_p
        lib.dtf_rdzv_create.argtypes = [c.c_char_p, c.c_int, c.c_int, c.c_int, c.c_double,
                                        c.POINTER(c.c_int)]
        lib.dtf_rdzv_all_gather.restype = c.c_int
        lib.dtf_rdzv_all_gather.argtypes = [c.c_void_p, c.c_char_p, c.c_long, c.c_char_p]
        lib.dtf_rdzv_broadcast.restype = c.c_int
        lib.dtf_rdzv_broadcast.argtypes = [c.c_void_p, c.c_char_p, c.c_long, c.c_int]
        lib.dtf_rdzv_barrier.restype = c.c_int
        lib.dtf_rdzv_barrier.argtypes = [c.c_void_p]
        lib.dtf_rdzv_set_timeout.restype = None
        lib.dtf_rdzv_set_timeout.argtypes = [c.c_void_p, c.c_double]
        lib.dtf_rdzv_destroy.restype = None
        lib.dtf_rdzv_destroy.argtypes = [c.c_void_p]
    if hasattr(lib, "dtf_rccl_available"):
        P, I, LG = c.c_void_p, c.c_int, c.c_long
        lib.dtf_rccl_available.restype = I
        lib.dtf_rccl_version.restype = I
        lib.dtf_rccl_has_split.restype = I
        lib.dtf_rccl_error_string.restype = c.c_char_p
        lib.dtf_rccl_error_string.argtypes = [I]
        lib.dtf_rccl_unique_id.argtypes = [c.c_char_p]
        lib.dtf_rccl_init.argtypes = [c.POINTER(P), I, c.c_char_p, I]
        lib.dtf_rccl_destroy.argtypes = [P, I]
        lib.dtf_rccl_async_error.argtypes = [P, c.POINTER(I)]
        lib.dtf_rccl_split.argtypes = [P, I, I, c.POINTER(P)]
        lib.dtf_rccl_all_reduce.argtypes = [P, P, P, LG, I, I, P]
        lib.dtf_rccl_reduce_scatter.argtypes = [P, P, P, LG, I, I, P]
        lib.dtf_rccl_all_gather.argtypes = [P, P, P, LG, I, P]
        lib.dtf_rccl_broadcast.argtypes = [P, P, LG, I, I, P]
        for name in ("unique_id", "init", "destroy", "async_error", "split", "all_reduce",
                     "reduce_scatter", "all_gather", "broadcast"):
            getattr(lib, "dtf_rccl_" + name).restype = I
    _LIB = lib
    _TRIED = True                # only now: a concurrent caller must not see "tried, nothing"
    return _LIB


def available() -> bool:
    return load() is not None
