"""SSTable (LevelDB table format) writer / reader -- the container of Saver-V2 ``.index`` files.

Layout: data blocks of prefix-compressed ``(key, value)`` entries with a restart array (every
16 entries), each followed by a 1-byte compression type (0 = none) and the masked CRC-32C of
block + type; an (empty) metaindex block; an index block mapping a separator key to each data
block's ``(offset, size)`` handle; a 48-byte footer -- the two block handles, padding, and the
magic ``0xdb4775248b80fb57``. Keys are added in increasing bytewise order.

``write_table`` / ``read_table`` use the C++ implementation in ``libdtf_host.so`` when it is
built and this pure-Python one otherwise; both produce identical bytes.
"""
from __future__ import annotations

import ctypes
import struct
from typing import Iterable, List, Tuple

from . import crc32c as _crc
from . import native
from .protowire import decode_varint, encode_varint

RESTART_INTERVAL = 16
BLOCK_SIZE = 4096
TABLE_MAGIC = 0xDB4775248B80FB57


class _BlockBuilder:
    def __init__(self):
        self.buf = bytearray()
        self.restarts = [0]
        self.last_key = b""
        self.counter = 0

    def add(self, key: bytes, value: bytes) -> None:
        shared = 0
        if self.counter < RESTART_INTERVAL:
            m = min(len(self.last_key), len(key))
            while shared < m and self.last_key[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.counter = 0
        self.buf += encode_varint(shared) + encode_varint(len(key) - shared) \
            + encode_varint(len(value)) + key[shared:] + value
        self.last_key = key
        self.counter += 1

    def empty(self) -> bool:
        return not self.buf

    def estimate(self) -> int:
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self) -> bytes:
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) \
            + struct.pack("<I", len(self.restarts))


def _shortest_separator(a: bytes, b: bytes) -> bytes:
    m = min(len(a), len(b))
    d = 0
    while d < m and a[d] == b[d]:
        d += 1
    if d < m and a[d] < 0xFF and a[d] + 1 < b[d]:
        return a[:d] + bytes([a[d] + 1])
    return a


def _short_successor(a: bytes) -> bytes:
    for i, byte in enumerate(a):
        if byte != 0xFF:
            return a[:i] + bytes([byte + 1])
    return a


def _block_with_trailer(contents: bytes) -> bytes:
    crc = _crc.crc32c(contents + b"\x00")
    return contents + b"\x00" + struct.pack("<I", _crc.mask(crc))


def build_table_py(entries: Iterable[Tuple[bytes, bytes]]) -> bytes:
    out = bytearray()
    data, index = _BlockBuilder(), _BlockBuilder()
    last_key = b""
    pending = None      # (offset, size) of the last flushed data block
    first = True

    def flush():
        nonlocal data, pending
        if data.empty():
            return
        contents = data.finish()
        pending = (len(out), len(contents))
        out.extend(_block_with_trailer(contents))
        data = _BlockBuilder()

    for key, value in entries:
        if not first and not last_key < key:
            raise ValueError("table keys must be added in increasing order")
        first = False
        if pending is not None:
            index.add(_shortest_separator(last_key, key),
                      encode_varint(pending[0]) + encode_varint(pending[1]))
            pending = None
        data.add(key, value)
        last_key = key
        if data.estimate() >= BLOCK_SIZE:
            flush()
    flush()
    if pending is not None:
        index.add(_short_successor(last_key), encode_varint(pending[0]) + encode_varint(pending[1]))
    meta = _BlockBuilder().finish()
    meta_off, meta_size = len(out), len(meta)
    out.extend(_block_with_trailer(meta))
    idx = index.finish()
    idx_off, idx_size = len(out), len(idx)
    out.extend(_block_with_trailer(idx))
    footer = encode_varint(meta_off) + encode_varint(meta_size) + encode_varint(idx_off) \
        + encode_varint(idx_size)
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    return bytes(out)


def _read_block(buf: bytes, off: int, size: int) -> bytes:
    if off + size + 5 > len(buf):
        raise IOError("block handle outside the file")
    contents = buf[off:off + size]
    (stored,) = struct.unpack_from("<I", buf, off + size + 1)
    if _crc.mask(_crc.crc32c(buf[off:off + size + 1])) != stored:
        raise IOError("block checksum mismatch")
    if buf[off + size] != 0:
        raise IOError("compressed table blocks are not supported")
    return contents


def _parse_block(block: bytes) -> List[Tuple[bytes, bytes]]:
    (num_restarts,) = struct.unpack_from("<I", block, len(block) - 4)
    end = len(block) - 4 - 4 * num_restarts
    pos = 0
    key = b""
    out = []
    while pos < end:
        shared, pos = decode_varint(block, pos)
        non_shared, pos = decode_varint(block, pos)
        vlen, pos = decode_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def parse_table_py(buf: bytes) -> List[Tuple[bytes, bytes]]:
    if len(buf) < 48:
        raise IOError("file too short to be a table")
    (magic,) = struct.unpack_from("<Q", buf, len(buf) - 8)
    if magic != TABLE_MAGIC:
        raise IOError("bad table magic")
    pos = len(buf) - 48
    _meta_off, pos = decode_varint(buf, pos)
    _meta_size, pos = decode_varint(buf, pos)
    idx_off, pos = decode_varint(buf, pos)
    idx_size, pos = decode_varint(buf, pos)
    entries: List[Tuple[bytes, bytes]] = []
    for _sep, handle in _parse_block(_read_block(buf, idx_off, idx_size)):
        off, p = decode_varint(handle, 0)
        size, _ = decode_varint(handle, p)
        entries.extend(_parse_block(_read_block(buf, off, size)))
    return entries


def write_table(path: str, entries: Iterable[Tuple[bytes, bytes]]) -> None:
    lib = native.load()
    if lib is None:
        with open(path, "wb") as fh:
            fh.write(build_table_py(entries))
        return
    handle = lib.dtf_table_builder_new(path.encode())
    if not handle:
        raise IOError(lib.dtf_last_error().decode())
    for key, value in entries:
        if lib.dtf_table_builder_add(handle, key, len(key), value, len(value)) != 0:
            err = lib.dtf_last_error().decode()
            lib.dtf_table_builder_finish(handle)
            raise ValueError(err)
    if lib.dtf_table_builder_finish(handle) != 0:
        raise IOError(lib.dtf_last_error().decode())


def read_table(path: str) -> List[Tuple[bytes, bytes]]:
    lib = native.load()
    if lib is None:
        with open(path, "rb") as fh:
            return parse_table_py(fh.read())
    handle = lib.dtf_table_open(path.encode())
    if not handle:
        raise IOError("%s: %s" % (path, lib.dtf_last_error().decode()))
    try:
        out = []
        kp, vp = ctypes.c_void_p(), ctypes.c_void_p()
        kl, vl = ctypes.c_size_t(), ctypes.c_size_t()
        for i in range(lib.dtf_table_count(handle)):
            lib.dtf_table_entry(handle, i, ctypes.byref(kp), ctypes.byref(kl), ctypes.byref(vp),
                                ctypes.byref(vl))
            out.append((ctypes.string_at(kp.value, kl.value) if kl.value else b"",
                        ctypes.string_at(vp.value, vl.value) if vl.value else b""))
        return out
    finally:
        lib.dtf_table_close(handle)
