"""Element types of graph tensors and their torch / numpy / checkpoint encodings.

The reference is fp32 throughout (``tf_distributed.py:44,46``); bf16 and OCP fp8 (e4m3fn, the
gfx950 matrix-core format -- not the MI300 ``fnuz`` variant) are added for the sharded-GEMM
workloads. ``enum`` values are the checkpoint ``DataType`` codes so Saver V2 index entries can
be written without a translation table.
"""
from __future__ import annotations

from typing import Any

import numpy as np
import torch


class DType:
    __slots__ = ("name", "torch", "numpy", "enum", "size", "is_floating")

    def __init__(self, name: str, torch_dtype, numpy_dtype, enum: int, size: int,
                 is_floating: bool):
        self.name = name
        self.torch = torch_dtype
        self.numpy = numpy_dtype
        self.enum = enum
        self.size = size
        self.is_floating = is_floating

    @property
    def as_numpy_dtype(self):
        return self.numpy

    def __repr__(self) -> str:
        return "dtf.%s" % self.name

    def __eq__(self, other) -> bool:
        try:
            return as_dtype(other).name == self.name
        except TypeError:
            return False

    def __hash__(self) -> int:
        return hash(self.name)


float32 = DType("float32", torch.float32, np.float32, 1, 4, True)
float64 = DType("float64", torch.float64, np.float64, 2, 8, True)
int32 = DType("int32", torch.int32, np.int32, 3, 4, False)
int64 = DType("int64", torch.int64, np.int64, 9, 8, False)
bool_ = DType("bool", torch.bool, np.bool_, 10, 1, False)
bfloat16 = DType("bfloat16", torch.bfloat16, None, 14, 2, True)
float16 = DType("float16", torch.float16, np.float16, 19, 2, True)
float8_e4m3fn = DType("float8_e4m3fn", torch.float8_e4m3fn, None, 24, 1, True)
string = DType("string", None, object, 7, 0, False)

_ALL = [float32, float64, int32, int64, bool_, bfloat16, float16, float8_e4m3fn, string]
_BY_NAME = {d.name: d for d in _ALL}
_BY_TORCH = {d.torch: d for d in _ALL if d.torch is not None}
_BY_ENUM = {d.enum: d for d in _ALL}


def as_dtype(value: Any) -> DType:
    if isinstance(value, DType):
        return value
    if isinstance(value, str) and value in _BY_NAME:
        return _BY_NAME[value]
    if isinstance(value, torch.dtype) and value in _BY_TORCH:
        return _BY_TORCH[value]
    if isinstance(value, int) and not isinstance(value, bool) and value in _BY_ENUM:
        return _BY_ENUM[value]
    if value is float:
        return float32
    if value is int:
        return int32
    if value is bool:
        return bool_
    try:
        np_dtype = np.dtype(value)
    except TypeError:
        raise TypeError("cannot convert %r to a dtype" % (value,))
    for d in _ALL:
        if d.numpy is not None and d.numpy is not object and np.dtype(d.numpy) == np_dtype:
            return d
    raise TypeError("unsupported dtype %r" % (value,))


def from_enum(code: int) -> DType:
    return _BY_ENUM[code]
