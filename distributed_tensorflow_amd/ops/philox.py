"""Counter-based Philox4x32-10 random streams (host implementation).

Variable initial values (reference ``tf_distributed.py:51-52``,
``tf_distributed_1000Matrix.py:43-44``) come from a counter-based generator so that any rank
can produce any slice of a tensor without communication: element ``e`` of a stream keyed by
``(graph_seed, op_seed)`` depends only on ``e``. The HIP kernel ``philox_normal_fill``
implements the same function; this numpy version is the CPU path and the test oracle.

Layout: block ``b = e // 4`` has counter ``(b_lo, b_hi, 0, 0)`` and key
``(graph_seed, op_seed)``; its four 32-bit outputs give four uniforms
``u = (x >> 8) * 2^-24 + 2^-25`` in (0, 1); normals are Box-Muller pairs
``(u0, u1) -> lanes 0, 1`` and ``(u2, u3) -> lanes 2, 3``.
"""
from __future__ import annotations

import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_TWO_PI = np.float32(6.283185307179586)


def philox4x32(block_index: np.ndarray, key0: int, key1: int) -> np.ndarray:
    """Return uint32 array [n, 4] of Philox4x32-10 outputs for the given block indices."""
    idx = block_index.astype(np.uint64)
    c0 = idx & _MASK
    c1 = idx >> np.uint64(32)
    c2 = np.zeros_like(c0)
    c3 = np.zeros_like(c0)
    k0 = key0 & 0xFFFFFFFF
    k1 = key1 & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0)
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=1).astype(np.uint32)


def _uniform01(bits: np.ndarray) -> np.ndarray:
    return ((bits >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
            + np.float32(2.0 ** -25))


def uniform(n: int, seed: int, op_seed: int, offset: int = 0) -> np.ndarray:
    """``n`` float32 uniforms in (0,1): elements ``offset .. offset+n`` of the stream."""
    first, last = offset // 4, (offset + n + 3) // 4
    bits = philox4x32(np.arange(first, max(last, first + 1), dtype=np.uint64), seed, op_seed)
    flat = _uniform01(bits).reshape(-1)
    start = offset - first * 4
    return flat[start:start + n]


def normal(n: int, seed: int, op_seed: int, offset: int = 0) -> np.ndarray:
    """``n`` float32 standard normals: elements ``offset .. offset+n`` of the stream."""
    first, last = offset // 4, (offset + n + 3) // 4
    bits = philox4x32(np.arange(first, max(last, first + 1), dtype=np.uint64), seed, op_seed)
    u = _uniform01(bits)
    out = np.empty_like(u)
    for a in (0, 2):
        r = np.sqrt(np.float32(-2.0) * np.log(u[:, a]), dtype=np.float32)
        theta = _TWO_PI * u[:, a + 1]
        out[:, a] = r * np.cos(theta, dtype=np.float32)
        out[:, a + 1] = r * np.sin(theta, dtype=np.float32)
    flat = out.reshape(-1)
    start = offset - first * 4
    return flat[start:start + n]


def window(rows: int, cols: int, seed: int, op_seed: int, row0: int, col0: int,
           global_cols: int, normal_dist: bool = True) -> np.ndarray:
    """Host version of the 2-D window generator: element (i, j) of the result is stream
    element ``(row0 + i) * global_cols + col0 + j`` (N(0,1) or U(0,1))."""
    idx = ((np.arange(rows, dtype=np.uint64)[:, None] + np.uint64(row0)) * np.uint64(global_cols)
           + np.arange(cols, dtype=np.uint64)[None, :] + np.uint64(col0)).reshape(-1)
    bits = philox4x32(idx >> np.uint64(2), seed, op_seed)
    u = _uniform01(bits)
    lane = (idx & np.uint64(3)).astype(np.int64)
    sel = np.arange(idx.size)
    if not normal_dist:
        return u[sel, lane].reshape(rows, cols)
    pair = (lane >> 1) * 2
    ua, ub = u[sel, pair], u[sel, pair + 1]
    rad = np.sqrt(np.float32(-2.0) * np.log(ua), dtype=np.float32)
    theta = _TWO_PI * ub
    out = np.where(lane & 1, rad * np.sin(theta, dtype=np.float32),
                   rad * np.cos(theta, dtype=np.float32))
    return out.astype(np.float32).reshape(rows, cols)
