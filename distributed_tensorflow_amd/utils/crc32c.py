"""CRC-32C (Castagnoli) and the "masked" variant used by TFRecord framing and checkpoint
index blocks. The native library (``csrc/io/crc32c.cc``, SSE4.2 ``crc32`` instruction with a
slice-by-8 table fallback) is used when it has been built; this table-driven Python version is
the fallback and the test oracle for it.
"""
from __future__ import annotations

_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ _POLY if _c & 1 else _c >> 1
    _TABLE.append(_c)
_MASK_DELTA = 0xA282EAD8


def crc32c_py(data: bytes, crc: int = 0) -> int:
    crc ^= 0xFFFFFFFF
    table = _TABLE
    for b in data:
        crc = table[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def crc32c(data: bytes, crc: int = 0) -> int:
    from . import native
    lib = native.load()
    if lib is not None:
        return lib.dtf_crc32c_extend(crc, bytes(data), len(data))
    return crc32c_py(data, crc)


def mask(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xFFFFFFFF


def unmask(masked: int) -> int:
    rot = (masked - _MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    return mask(crc32c(data))
