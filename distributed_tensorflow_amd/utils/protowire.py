"""Minimal protobuf wire-format encoder/decoder.

The event-file, summary, checkpoint-bundle and cluster-definition records this framework
writes are protobuf messages, but they are tiny and fixed, so they are hand-encoded here
instead of depending on generated ``_pb2`` modules. Only the wire types those messages use
are supported: varint (0), 64-bit (1), length-delimited (2) and 32-bit (5).
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, List, Tuple, Union

VARINT, FIXED64, BYTES, FIXED32 = 0, 1, 2, 5


def encode_varint(value: int) -> bytes:
    if value < 0:
        value += 1 << 64  # two's complement, as protobuf encodes negative int32/int64
    out = bytearray()
    while True:
        low = value & 0x7F
        value >>= 7
        if value:
            out.append(low | 0x80)
        else:
            out.append(low)
            return bytes(out)


def decode_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = 0
    result = 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        byte = buf[pos]
        pos += 1
        result |= (byte & 0x7F) << shift
        if not byte & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _tag(field: int, wire: int) -> bytes:
    return encode_varint((field << 3) | wire)


def f_varint(field: int, value: int) -> bytes:
    return _tag(field, VARINT) + encode_varint(int(value))


def f_bool(field: int, value: bool) -> bytes:
    return f_varint(field, 1 if value else 0)


def f_bytes(field: int, value: Union[bytes, str]) -> bytes:
    if isinstance(value, str):
        value = value.encode("utf-8")
    return _tag(field, BYTES) + encode_varint(len(value)) + value


def f_message(field: int, payload: bytes) -> bytes:
    return f_bytes(field, payload)


def f_double(field: int, value: float) -> bytes:
    return _tag(field, FIXED64) + struct.pack("<d", value)


def f_float(field: int, value: float) -> bytes:
    return _tag(field, FIXED32) + struct.pack("<f", value)


def f_fixed32(field: int, value: int) -> bytes:
    return _tag(field, FIXED32) + struct.pack("<I", value & 0xFFFFFFFF)


def f_fixed64(field: int, value: int) -> bytes:
    return _tag(field, FIXED64) + struct.pack("<Q", value & 0xFFFFFFFFFFFFFFFF)


def iter_fields(buf: bytes) -> Iterator[Tuple[int, int, Union[int, bytes]]]:
    """Yield ``(field_number, wire_type, value)``; value is int for varint/fixed, bytes else."""
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = decode_varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == VARINT:
            value, pos = decode_varint(buf, pos)
        elif wire == FIXED64:
            value = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wire == FIXED32:
            value = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wire == BYTES:
            length, pos = decode_varint(buf, pos)
            if pos + length > n:
                raise ValueError("truncated length-delimited field")
            value = bytes(buf[pos:pos + length])
            pos += length
        else:
            raise ValueError("unsupported wire type %d" % wire)
        yield field, wire, value


def parse(buf: bytes) -> Dict[int, List[Union[int, bytes]]]:
    out: Dict[int, List[Union[int, bytes]]] = {}
    for field, _wire, value in iter_fields(buf):
        out.setdefault(field, []).append(value)
    return out


def as_float(bits: int) -> float:
    return struct.unpack("<f", struct.pack("<I", bits))[0]


def as_double(bits: int) -> float:
    return struct.unpack("<d", struct.pack("<Q", bits))[0]


def as_signed64(value: int) -> int:
    return value - (1 << 64) if value >= (1 << 63) else value
