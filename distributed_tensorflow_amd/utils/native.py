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
from typing import Optional

_LIB: Optional[ctypes.CDLL] = None
_TRIED = False
_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libdtf_host.so")


def load() -> Optional[ctypes.CDLL]:
    global _LIB, _TRIED
    if _TRIED:
        return _LIB
    _TRIED = True
    if os.environ.get("DTF_DISABLE_NATIVE_HOST") == "1" or not os.path.exists(HOST_LIB_PATH):
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
    _LIB = lib
    return _LIB


def available() -> bool:
    return load() is not None
