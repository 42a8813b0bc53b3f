"""ctypes binding of ``libdtf_hip.so`` -- the hand-written gfx950 kernel library.

Each wrapper takes torch tensors that live on the GPU, checks on the host that their shapes,
dtypes, contiguity and alignment match what the kernel and its launch grid assume (a kernel
that indexes out of bounds can take the whole machine down, so nothing is launched on a
mismatch), and enqueues the kernel on torch's *current* stream. Launchers never allocate or
synchronise, so calls made while torch is capturing a HIP graph are captured.

The library is built in-tree by ``build_native.py``. When a GPU is present and the library is
missing, importing this module's users fails loudly rather than falling back to torch ops.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import threading
from typing import Any, List, Optional, Sequence, Tuple

import torch

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_HERE, "lib", "libdtf_hip.so")

ACT_NONE, ACT_SIGMOID, ACT_RELU, ACT_TANH = 0, 1, 2, 3
ACTIVATIONS = {"none": ACT_NONE, "sigmoid": ACT_SIGMOID, "relu": ACT_RELU, "tanh": ACT_TANH}

# op codes (must match csrc/kernels/elementwise.hip and reduce.hip)
BINARY = {"add": 0, "sub": 1, "mul": 2, "div": 3, "max": 4, "min": 5, "pow": 6,
          "sigmoid_grad": 7, "tanh_grad": 8, "relu_grad": 9}
COMPARE = {"eq": 0, "ne": 1, "lt": 2, "gt": 3, "le": 4, "ge": 5}
UNARY = {"neg": 0, "log": 1, "exp": 2, "sqrt": 3, "square": 4, "recip": 5, "abs": 6,
         "sigmoid": 7, "tanh": 8, "relu": 9, "floor": 10, "cos": 11, "sin": 12,
         "softplus": 13}
ROW_SUM, ROW_MEAN, ROW_MAX = 0, 1, 2

_lib: Optional[ctypes.CDLL] = None


class HipKernelError(RuntimeError):
    pass


def library_present() -> bool:
    return os.path.exists(LIB_PATH)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not library_present():
            raise HipKernelError(
                "native kernel library %s is missing: run `python build_native.py` "
                "(refusing to fall back to torch kernels on a GPU)" % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


class _CallRecorder:
    """Stands in for the kernel library while a session captures a step: every call goes through
    to the real library and is noted as (function, arguments) when it was made by the capturing
    thread. A short step is later re-issued from that list instead of replayed as a HIP graph."""

    def __init__(self, real: ctypes.CDLL):
        self._real = real
        self._thread = threading.get_ident()
        self.calls: List[Tuple[Any, tuple]] = []

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if not callable(fn):
            return fn

        def call(*args):
            rc = fn(*args)
            if threading.get_ident() == self._thread:
                self.calls.append((fn, args))
            return rc
        return call


@contextlib.contextmanager
def record_calls():
    """``with record_calls() as rec:`` -- ``rec.calls`` lists the library calls this thread makes
    inside the block, in order."""
    global _lib
    real = lib()
    if isinstance(real, _CallRecorder):            # nested: keep the outer recorder
        yield real
        return
    rec = _CallRecorder(real)
    _lib = rec
    try:
        yield rec
    finally:
        _lib = real


def _declare(L: ctypes.CDLL) -> None:
    c = ctypes
    P, I, LG, F, U = c.c_void_p, c.c_int, c.c_long, c.c_float, c.c_uint
    sigs = {
        "dtf_gemm_f32": [P, P, P, P, I, I, I, LG, LG, LG, I, I, I, P, I, I, P],
        "dtf_gemm_f32_ex": [P, P, P, P, I, I, I, LG, LG, LG, I, I, I, P, I, I, I, I, P],
        "dtf_pad_2d_f32": [P, P, LG, LG, LG, LG, LG, P],
        "dtf_pad_2d_b16": [P, P, LG, LG, LG, LG, LG, P],
        "dtf_pad_2d_b8": [P, P, LG, LG, LG, LG, LG, P],
        "dtf_gemm_f32_skinny_ring": [P, P, P, I, I, I, P, P, P, I, I, I, LG, LG, LG, I, I, I, P],
        "dtf_mlp_head_train_ring": [P, P, P, P, P, P, P, P, P, P, P, P, LG, LG, LG, I, P, P, P, I,
                                    I, P],
        "dtf_gemm_f32_splits": [I, I, I, P],
        "dtf_gemm_f32_tile": [I, I, I],
        "dtf_gemm_bf16_nt": [P, P, P, I, I, I, LG, LG, LG, F, F, I, I, P],
        "dtf_gemm_bf16_splits": [I, I, I],
        "dtf_gemm_bf16_nt_splitk": [P, P, P, P, I, I, I, LG, LG, LG, F, F, I, I, P],
        "dtf_gemm_bf16_generic": [P, P, P, I, I, I, LG, LG, LG, LG, LG, F, F, I, P],
        "dtf_gemm_fp8_nt": [P, P, P, I, I, I, LG, LG, LG, F, F, I, P],
        "dtf_gemm_fp8_splits": [I, I, I],
        "dtf_gemm_last_kernel": [],
        "dtf_gemm_fp8_nt_splitk": [P, P, P, P, I, I, I, LG, LG, LG, F, F, I, I, P],
        "dtf_gemm_fp8_generic": [P, P, P, I, I, I, LG, LG, LG, LG, LG, F, F, I, P],
        "dtf_gemm_fp8_mx_nt": [P, P, P, P, P, I, I, I, LG, LG, LG, LG, LG, F, F, I, P],
        "dtf_quantize_mx_fp8": [P, I, LG, LG, LG, P, LG, P, P, I, P],
        "dtf_cast_to_fp8": [P, I, P, LG, F, P],
        "dtf_transpose": [P, P, LG, LG, LG, LG, I, P],
        "dtf_cast_from_fp8": [P, P, I, LG, F, P],
        "dtf_binary_f32": [I, P, P, P, LG, LG, LG, P],
        "dtf_unary_f32": [I, P, P, LG, P],
        "dtf_compare_f32": [I, P, P, P, LG, LG, LG, P],
        "dtf_equal_i64": [P, P, P, LG, P],
        "dtf_axpby_f32": [F, P, F, P, P, LG, P],
        "dtf_axpby_bf16": [F, P, F, P, P, LG, P],
        "dtf_cast_f32_to_bf16": [P, P, LG, P],
        "dtf_cast_bf16_to_f32": [P, P, LG, P],
        "dtf_cast_bool_to_f32": [P, P, LG, P],
        "dtf_fill_f32": [P, F, LG, P],
        "dtf_row_scale_f32": [P, P, P, LG, LG, P],
        "dtf_reduce_rows_f32": [I, P, P, LG, LG, P],
        "dtf_argmax_rows_f32": [P, P, LG, LG, P],
        "dtf_peer_push": [P, P, I, LG, I, P],
        "dtf_reduce_all_workspace": [LG],
        "dtf_reduce_all_f32": [P, P, LG, F, P, P],
        "dtf_reduce_cols_f32": [P, P, LG, LG, P],
        "dtf_sigmoid_grad_bias_f32": [P, P, P, P, LG, LG, P],
        "dtf_softmax_xent_grad_f32": [P, P, P, P, P, LG, LG, P],
        "dtf_xent_grad_narrow_f32": [P, P, P, P, P, LG, LG, P],
        "dtf_mlp_head_supported": [LG, LG, LG],
        "dtf_mlp_head_train": [P, P, P, P, P, P, P, P, P, P, P, P, P, LG, LG, LG, I, P],
        "dtf_mlp_head_workspace": [LG, LG, LG],
        "dtf_argmax_accuracy_f32": [P, P, P, P, LG, LG, P],
        "dtf_argmax_accuracy_workspace": [LG],
        "dtf_softmax_rows_f32": [P, P, P, LG, LG, P],
        "dtf_softmax_xent_workspace": [LG],
        "dtf_softmax_xent_f32": [P, P, P, P, P, P, LG, LG, P],
        "dtf_softmax_grad_f32": [P, P, P, LG, LG, P],
        "dtf_multi_sgd_f32": [I, P, P, P, P, I, P, P, P],
        "dtf_dropout_f32": [P, P, P, LG, U, U, P, F, P, P],
        "dtf_bump_counter_u64": [P, LG, P],
        "dtf_gemm_bf16_nt_mirrored": [P, P, P, P, I, I, I, I, LG, LG, LG, F, I, P],
        "dtf_multi_optim_f32": [I, I, P, P, P, P, P, P, I, P, P, P, P],
        "dtf_affine_scalars_f32": [I, P, P, P, P],
        "dtf_philox_normal": [P, LG, U, U, LG, F, F, I, P],
        "dtf_philox_uniform": [P, LG, U, U, LG, F, F, I, P],
        "dtf_reduce_partials": [I, P, P, LG, F, I, P],
        "dtf_pack_f32": [I, P, P, P, P],
        "dtf_pack_f32_signal": [I, P, P, P, P, P, P, P],
        "dtf_gather_feeds": [P, I, P, P, P],
        "dtf_pack_f32_ring": [I, P, P, P, P, P, P, I, I, P],
        "dtf_gather_feeds_ring": [P, P, P, I, I, P, P, P],
        "dtf_mlp_step_supported": [LG, LG, LG, LG],
        "dtf_mlp_step_train": [P, P],
        "dtf_philox_fill_2d": [P, LG, LG, LG, U, U, LG, LG, LG, F, F, I, I, P],
        "dtf_copy_2d": [P, P, LG, LG, LG, LG, P],
        "dtf_multi_copy": [I, P, P, P, P],
        "dtf_pad_rows": [P, P, LG, LG, LG, LG, LG, P],
        "dtf_assign_add_f32": [P, P, LG, LG, F, P],
        "dtf_sign_f32": [P, P, LG, P],
        "dtf_mse_f32": [P, P, P, LG, P, P],
        "dtf_mse_grad_f32": [P, P, P, P, LG, P],
        "dtf_mse_step_supported": [LG, LG, I],
        "dtf_mse_step_f32": [P, P, P, F, P, P, P, LG, LG, I, P, P, P, P, P, P, I, I, I, P],
        "dtf_select_f32": [P, P, P, P, LG, LG, LG, LG, P],
        "dtf_one_hot_f32": [P, I, P, LG, LG, F, F, P],
        "dtf_broadcast_mid_f32": [P, P, LG, LG, LG, F, P],
        "dtf_log_softmax_grad_f32": [P, P, P, LG, LG, P],
        "dtf_absmax_scale": [P, I, LG, F, P, P, P],
        "dtf_cast_to_fp8_dev_scale": [P, I, P, LG, P, P],
        "dtf_truncated_normal_f32": [P, LG, U, U, F, F, P],
        "dtf_cast_generic": [P, I, P, I, LG, P],
        "dtf_flags_alloc": [LG, c.POINTER(c.c_void_p), c.c_char_p],
        "dtf_flags_open": [c.c_char_p, c.POINTER(c.c_void_p)],
        "dtf_flags_close": [P, I],
        "dtf_peer_signal": [c.POINTER(c.c_void_p), I, I, P],
        "dtf_peer_wait": [c.POINTER(c.c_void_p), I, I, LG, P, P],
    }
    for name, argtypes in sigs.items():
        fn = getattr(L, name)
        fn.argtypes = argtypes
        fn.restype = c.c_int
    L.dtf_absmax_workspace.argtypes = [LG]
    L.dtf_absmax_workspace.restype = LG
    L.dtf_mx_scale_bytes.argtypes = [LG, LG]
    L.dtf_mx_scale_bytes.restype = LG


# ---------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------
def _broadcast_shape(a, b) -> Tuple[int, ...]:
    """NumPy broadcasting of two shapes (``torch.broadcast_shapes`` imports the symbolic-shape
    machinery on first use: a quarter of a second of start-up for a three-line rule)."""
    a, b = tuple(a), tuple(b)
    n = max(len(a), len(b))
    a, b = (1,) * (n - len(a)) + a, (1,) * (n - len(b)) + b
    out = []
    for x, y in zip(a, b):
        if x != y and x != 1 and y != 1:
            raise HipKernelError("shapes %s and %s do not broadcast" % (a, b))
        out.append(y if x == 1 else x)
    return tuple(out)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _check(code: int, what: str) -> None:
    if code != 0:
        if code > 0:
            raise HipKernelError("%s: HIP launch error %d" % (what, code))
        raise HipKernelError("%s: rejected arguments (code %d)" % (what, code))


def _req(t: torch.Tensor, dtype: torch.dtype, what: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise HipKernelError("%s: expected a GPU tensor" % what)
    if t.dtype != dtype:
        raise HipKernelError("%s: expected dtype %s, got %s" % (what, dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _req_rows(t: torch.Tensor, dtype: torch.dtype, what: str) -> torch.Tensor:
    """Rank-2 GPU tensor whose rows are contiguous (any row pitch): a row or column block of a
    larger matrix is passed to the kernel as it is, with its pitch as leading dimension."""
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise HipKernelError("%s: expected a GPU tensor" % what)
    if t.dtype != dtype:
        raise HipKernelError("%s: expected dtype %s, got %s" % (what, dtype, t.dtype))
    if t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] <= 1) \
            and (t.stride(0) >= t.shape[1] or t.shape[0] <= 1):
        return t
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _aligned(t: torch.Tensor) -> torch.Tensor:
    """Kernels with 16-byte vector paths need a 16-byte aligned base."""
    return t if t.data_ptr() % 16 == 0 else t.clone()


# ---------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------
def gemm_f32(a: torch.Tensor, b: torch.Tensor, trans_a: bool = False, trans_b: bool = False,
             bias: Optional[torch.Tensor] = None, activation: int = ACT_NONE,
             out: Optional[torch.Tensor] = None, splits: Optional[int] = None,
             tile: int = 0, plan_mnk: Optional[Sequence[int]] = None) -> torch.Tensor:
    """``act(op(a) @ op(b) + bias)`` on the fp32 MFMA kernels. ``a`` is [M,K] ([K,M] when
    ``trans_a``), ``b`` is [K,N] ([N,K] when ``trans_b``). ``tile``: 0 = chosen from the shape;
    16 / 32 (small-product kernels), 64, 128 or 256 force a block tile (128 and 256 run the
    slot-scheduled kernel when the operands are 16-byte aligned with contiguous dimensions that
    are multiples of 4, the general kernel otherwise); 129 = 128x128 tiles on the general kernel
    whatever the operands, for measurements. ``plan_mnk``: choose tile and K split
    as for a product of THAT shape (with the same K): a row or column block of a larger product
    then comes out bit-identical to the same block of the whole result."""
    if getattr(a, "dim", lambda: 0)() != 2 or getattr(b, "dim", lambda: 0)() != 2:
        raise HipKernelError("gemm_f32: operands must be rank 2")
    a = _req_rows(a, torch.float32, "gemm_f32 a")
    b = _req_rows(b, torch.float32, "gemm_f32 b")
    m, k = (a.shape[1], a.shape[0]) if trans_a else a.shape
    k2, n = (b.shape[1], b.shape[0]) if trans_b else b.shape
    if k != k2:
        raise HipKernelError("gemm_f32: inner dimensions differ (%s vs %s)"
                             % (tuple(a.shape), tuple(b.shape)))
    if bias is not None:
        bias = _req(bias, torch.float32, "gemm_f32 bias")
        if bias.numel() != n:
            raise HipKernelError("gemm_f32: bias has %d elements, N=%d" % (bias.numel(), n))
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    elif out.shape != (m, n) or out.dtype != torch.float32 or out.device != a.device \
            or not (out.stride(1) == 1 or n <= 1) or (m > 1 and out.stride(0) < n):
        raise HipKernelError("gemm_f32: bad out tensor")
    if m == 0 or n == 0:
        return out
    if k == 0:
        return out.zero_()
    L = lib()
    lda = a.stride(0) if a.shape[0] > 1 else a.shape[1]
    ldb = b.stride(0) if b.shape[0] > 1 else b.shape[1]
    ldc = out.stride(0) if m > 1 else n
    partial = None
    # tile and K split come from the product's shape (or the plan's), never from how the operands
    # happen to lie in memory: the padded copies below leave both as they are
    if plan_mnk is not None:
        pm, pn, pk = (int(v) for v in plan_mnk)
        if pk != k:
            raise HipKernelError("gemm_f32: plan_mnk must have the product's own K")
        tile = L.dtf_gemm_f32_tile(pm, pn, pk)
        splits = L.dtf_gemm_f32_splits(pm, pn, pk, None)
    elif not tile:
        tile = L.dtf_gemm_f32_tile(m, n, k)
    if splits is None:
        splits = L.dtf_gemm_f32_splits(m, n, k, None)
    splits = max(1, min(int(splits), 64))
    a_ptr, b_ptr, kk, m_load, n_load = a.data_ptr(), b.data_ptr(), k, m, n
    if tile not in (16, 32) and m * n * k >= _F32_PAD_MIN_WORK:
        # The kernels' fast loads want 16-byte aligned rows and a K that is a multiple of 4 for
        # K-contiguous operands; an operand that is contiguous along M / N may have any M / N as
        # long as its rows can be read up to the next multiple of 4. Operands that do not
        # qualify (an odd K, an odd row pitch, a view that starts off the grid) are copied once
        # into zero-padded aligned images -- (M + N) K elements moved against M N K products;
        # element-wise predicated loads cost 2-3x the whole product's time.
        a_kc, b_kc = not trans_a, trans_b
        k4 = (k + 3) // 4 * 4
        pad_k = k4 != k and (a_kc or b_kc)
        images = []
        for t, kc, ld, extent in ((a, a_kc, lda, m), (b, b_kc, ldb, n)):
            e4 = (extent + 3) // 4 * 4
            aligned = t.data_ptr() % 16 == 0 and ld % 4 == 0
            if kc:
                ok = aligned and not pad_k
            else:
                ok = aligned and not pad_k and _rows_readable_to(t, e4)
            if ok:
                images.append((t.data_ptr(), ld, extent if kc else e4))
                continue
            rows, cols = (extent, k) if kc else (k, extent)
            drows, dld = (extent, k4 if pad_k else k) if kc else (k4 if pad_k else k, e4)
            if kc and dld % 4:
                dld = (dld + 3) // 4 * 4          # (neither operand pads K: keep the pitch on the grid)
            img = torch.empty((drows, dld), dtype=torch.float32, device=t.device)
            _check(L.dtf_pad_2d_f32(t.data_ptr(), img.data_ptr(), rows, cols, ld, drows, dld,
                                    _stream()), "gemm_f32 (aligned image)")
            images.append((img.data_ptr(), dld, extent if kc else e4, img))
        (a_ptr, lda, m_load), (b_ptr, ldb, n_load) = images[0][:3], images[1][:3]
        if pad_k:
            kk = k4
    if splits > 1:
        # split-K workspace: one plane per split (caching allocator; graph-capturable)
        partial = torch.empty(splits * m * n, dtype=torch.float32, device=a.device)
    _check(L.dtf_gemm_f32_ex(a_ptr, b_ptr, out.data_ptr(), _ptr(bias), m, n, kk,
                             lda, ldb, ldc, int(trans_a), int(trans_b),
                             int(activation), _ptr(partial), splits, int(tile), int(m_load),
                             int(n_load), _stream()),
           "gemm_f32")
    return out


def gemm_f32_ring_ok(a_like: torch.Tensor, b: torch.Tensor, trans_a: bool = False,
                     trans_b: bool = False) -> bool:
    """Whether ``gemm_f32_ring`` takes this product: fp32 rank-2 operands with contiguous rows
    and a shape the small-product kernel serves."""
    if getattr(a_like, "dim", lambda: 0)() != 2 or getattr(b, "dim", lambda: 0)() != 2:
        return False
    if a_like.dtype != torch.float32 or b.dtype != torch.float32 or not a_like.is_contiguous() \
            or not b.is_contiguous() or a_like.device.type != "cuda":
        return False
    m, k = (a_like.shape[1], a_like.shape[0]) if trans_a else a_like.shape
    k2, n = (b.shape[1], b.shape[0]) if trans_b else b.shape
    return k == k2 and min(m, n, k) > 0 and lib().dtf_gemm_f32_tile(m, n, k) == 16


def gemm_f32_ring(ring: "RingArgs", a_index: int, a_like: torch.Tensor, b: torch.Tensor,
                  trans_a: bool = False, trans_b: bool = False,
                  bias: Optional[torch.Tensor] = None, activation: int = ACT_NONE,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``gemm_f32`` inside a captured step whose A operand is a fed tensor: A is read through
    entry ``a_index`` of the step ring's address table (``a_like``: the captured stand-in for
    the fed tensor, for shape and pitch only), so no gather launch copies the batch first.
    Only products the small-product kernel takes (``gemm_f32_ring_ok``)."""
    if not gemm_f32_ring_ok(a_like, b, trans_a, trans_b):
        raise HipKernelError("gemm_f32_ring: not a small contiguous fp32 product")
    m, k = (a_like.shape[1], a_like.shape[0]) if trans_a else a_like.shape
    n = b.shape[0] if trans_b else b.shape[1]
    if bias is not None:
        bias = _req(bias, torch.float32, "gemm_f32_ring bias")
        if bias.numel() != n:
            raise HipKernelError("gemm_f32_ring: bias has %d elements, N=%d" % (bias.numel(), n))
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=b.device)
    elif out.shape != (m, n) or out.dtype != torch.float32 or not out.is_contiguous():
        raise HipKernelError("gemm_f32_ring: bad out tensor")
    _check(lib().dtf_gemm_f32_skinny_ring(
        ring.table.data_ptr(), ring.ctr.data_ptr(), ring.out.data_ptr(), ring.slots,
        int(a_index), ring.take_first(), b.data_ptr(), out.data_ptr(), _ptr(bias), m, n, k,
        a_like.shape[1], b.shape[1], n, int(trans_a), int(trans_b), int(activation), _stream()),
        "gemm_f32_ring")
    return out


# products smaller than this run on whatever loads their operands allow (a copy would cost more
# than it saves)
_F32_PAD_MIN_WORK = 64 * 64 * 64


def _rows_readable_to(t: torch.Tensor, cols: int) -> bool:
    """Whether every row of the rank-2 row-contiguous tensor, the last one included, can be read
    up to column ``cols`` (>= its width) without leaving its storage or its own row pitch."""
    if cols <= t.shape[1]:
        return True
    rows = t.shape[0]
    ld = t.stride(0) if rows > 1 else t.shape[1]
    if cols > ld:                      # (one row: its pitch is its width)
        return False
    last = t.storage_offset() + (rows - 1) * ld + cols
    return last <= t.untyped_storage().nbytes() // t.element_size()


# bf16 products smaller than this run on whatever kernel their operands allow
_BF16_PAD_MIN_WORK = 1 << 20


def gemm_bf16_nt(a: torch.Tensor, bt: torch.Tensor, out_dtype: torch.dtype = torch.bfloat16,
                 alpha: float = 1.0, out: Optional[torch.Tensor] = None,
                 variant: int = 0, beta: float = 0.0,
                 splits: Optional[int] = None) -> torch.Tensor:
    """``alpha * a @ bt.T (+ beta * out)`` with a [M,K], bt [N,K] bf16 (both K-contiguous); fp32
    accumulate; result bf16 or fp32. Uses the 256x256x64 LDS-DMA kernel when the shape
    qualifies (operands with another K, an odd pitch or an offset start, and results with an
    odd pitch, get there through padded copies when the product is large enough to care) and
    the predicated generic kernel otherwise. ``splits``: K splits for grids with
    few output tiles (default: the library's plan; 1 = never split). ``variant``: 0 = the
    library's choice of kernel by grid (256/192/128-row ping-pong tiles, or the kernels with
    request-issuing waves on 64x64, 128x128 or 128x256 tiles); another value forces one kernel
    (4 ping-pong, 5 / 11 the two forms of the 128x128 tile, 12 64x64 tiles, 13 128x256 tiles, 3 /
    7 / 8 / 10 the four-wave schedules: see ``csrc/kernels/gemm_bf16.hip``)."""
    a = _req(a, torch.bfloat16, "gemm_bf16_nt a")
    bt = _req(bt, torch.bfloat16, "gemm_bf16_nt bt")
    if a.dim() != 2 or bt.dim() != 2 or a.shape[1] != bt.shape[1]:
        raise HipKernelError("gemm_bf16_nt: need a [M,K] and bt [N,K], got %s and %s"
                             % (tuple(a.shape), tuple(bt.shape)))
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise HipKernelError("gemm_bf16_nt: out dtype must be bf16 or fp32")
    m, k = a.shape
    n = bt.shape[0]
    if beta != 0.0 and out is None:
        raise HipKernelError("gemm_bf16_nt: beta needs an out tensor to accumulate into")
    if out is None:
        out = torch.empty((m, n), dtype=out_dtype, device=a.device)
    elif out.shape != (m, n) or out.dtype != out_dtype or out.stride(1) != 1:
        raise HipKernelError("gemm_bf16_nt: bad out tensor")
    if m == 0 or n == 0:
        return out
    if k == 0:
        return out.mul_(beta) if beta != 0.0 else out.zero_()
    out_f32 = int(out_dtype == torch.float32)
    if splits is None:
        splits = lib().dtf_gemm_bf16_splits(m, n, k) if variant == 0 else 1
    if splits > 1:
        ws = torch.empty(splits * m * n, dtype=torch.float32, device=a.device)
        code = lib().dtf_gemm_bf16_nt_splitk(a.data_ptr(), bt.data_ptr(), out.data_ptr(),
                                             ws.data_ptr(), m, n, k, a.stride(0), bt.stride(0),
                                             out.stride(0), float(alpha), float(beta), out_f32,
                                             int(splits), _stream())
        if code == 0:
            return out
        if code != -2:
            _check(code, "gemm_bf16_nt (split-K)")
    # The fast kernel wants K % 64 == 0 (the reference's 1000 and 784 are not), 16-byte aligned
    # rows with pitches that are multiples of 8 elements, and an out whose pitch is a multiple of
    # 4 elements on an 8/16-byte grid. A product large enough to care gets there by copies:
    # operands into zero-padded aligned images -- (M + N) K elements moved once against M N K
    # products --, the result through a temporary with a padded pitch. (The predicated kernel
    # alone took 1172 us for 8192 x 8192 x 1000; images + fast kernel take 142. Running only the
    # K % 64 tail on the predicated kernel and accumulating the rest on top costs two more passes
    # over C and measured 3x slower than the images.)
    if m * n * k >= _BF16_PAD_MIN_WORK and k > 64:
        def operand_ok(t):
            return t.data_ptr() % 16 == 0 and t.stride(0) % 8 == 0 and k % 64 == 0 and \
                t.stride(0) * 2 * 256 < (1 << 32)
        out_ok = out.stride(0) % 4 == 0 and out.data_ptr() % (16 if out_f32 else 8) == 0
        if not (operand_ok(a) and operand_ok(bt) and out_ok) and (out_ok or beta == 0.0):
            kp = (k + 63) // 64 * 64
            images = []
            for t in (a, bt):
                if operand_ok(t):
                    images.append(t)
                    continue
                img = torch.empty((t.shape[0], kp), dtype=torch.bfloat16, device=t.device)
                if t.data_ptr() % 16 == 0 and t.stride(0) % 8 == 0 and k % 8 == 0:
                    code = lib().dtf_pad_rows(t.data_ptr(), img.data_ptr(), t.shape[0], 2 * k,
                                              2 * kp, 2 * t.stride(0), 2 * kp, _stream())
                else:
                    code = lib().dtf_pad_2d_b16(t.data_ptr(), img.data_ptr(), t.shape[0], k,
                                                t.stride(0), t.shape[0], kp, _stream())
                _check(code, "gemm_bf16_nt (aligned image)")
                images.append(img)
            if out_ok:
                return gemm_bf16_nt(images[0], images[1], out_dtype, alpha=alpha, out=out,
                                    variant=variant, beta=beta)
            n4 = (n + 3) // 4 * 4
            tmp = torch.empty((m, n4), dtype=out_dtype, device=a.device)
            gemm_bf16_nt(images[0], images[1], out_dtype, alpha=alpha, out=tmp[:, :n],
                         variant=variant)
            return copy_2d(tmp[:, :n], out)
    code = lib().dtf_gemm_bf16_nt(a.data_ptr(), bt.data_ptr(), out.data_ptr(), m, n, k,
                                  a.stride(0), bt.stride(0), out.stride(0), float(alpha),
                                  float(beta), out_f32, int(variant), _stream())
    if code == -2:    # shape/alignment outside the fast kernel's contract
        code = lib().dtf_gemm_bf16_generic(a.data_ptr(), bt.data_ptr(), out.data_ptr(), m, n, k,
                                           a.stride(0), 1, 1, bt.stride(0), out.stride(0),
                                           float(alpha), float(beta), out_f32, _stream())
    _check(code, "gemm_bf16_nt")
    return out


def gemm_bf16_nt_mirrored(a: torch.Tensor, bt: torch.Tensor, out: torch.Tensor,
                          mirror_table: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """``out = alpha * a @ bt.T`` as :func:`gemm_bf16_nt`, with every result vector ALSO stored,
    by the product's own epilogue, into further copies of ``out``: ``mirror_table`` is a
    one-dimensional int64 tensor in GPU memory holding their addresses (1 to 7; each laid out like
    ``out`` -- same pitch, 16-byte aligned -- typically the same rows of the peers' result buffers
    mapped through HIP IPC). Fast-kernel shapes only (K % 64 == 0, aligned operands); raises
    otherwise: the caller keeps the copy-after-product path for those."""
    a = _req(a, torch.bfloat16, "gemm_bf16_nt_mirrored a")
    bt = _req(bt, torch.bfloat16, "gemm_bf16_nt_mirrored bt")
    if a.dim() != 2 or bt.dim() != 2 or a.shape[1] != bt.shape[1]:
        raise HipKernelError("gemm_bf16_nt_mirrored: need a [M,K] and bt [N,K]")
    m, k = a.shape
    n = bt.shape[0]
    if out.dtype not in (torch.bfloat16, torch.float32) or out.shape != (m, n) \
            or out.stride(1) != 1 or out.device != a.device:
        raise HipKernelError("gemm_bf16_nt_mirrored: bad out tensor")
    if mirror_table.dtype != torch.int64 or mirror_table.dim() != 1 \
            or not 1 <= mirror_table.numel() <= 7 or mirror_table.device != a.device \
            or not mirror_table.is_contiguous():
        raise HipKernelError("gemm_bf16_nt_mirrored: the table must hold 1 to 7 int64 addresses "
                             "in GPU memory")
    _check(lib().dtf_gemm_bf16_nt_mirrored(
        a.data_ptr(), bt.data_ptr(), out.data_ptr(), mirror_table.data_ptr(),
        mirror_table.numel(), m, n, k, a.stride(0), bt.stride(0), out.stride(0), float(alpha),
        int(out.dtype == torch.float32), _stream()), "gemm_bf16_nt_mirrored")
    return out


FP8 = torch.float8_e4m3fn
FP8_MAX = 448.0


def gemm_fp8_nt(a: torch.Tensor, bt: torch.Tensor, out_dtype: torch.dtype = torch.bfloat16,
                alpha: float = 1.0, out: Optional[torch.Tensor] = None,
                force_generic: bool = False, beta: float = 0.0,
                splits: Optional[int] = None) -> torch.Tensor:
    """``alpha * a @ bt.T (+ beta * out)`` with a [M,K], bt [N,K] in OCP fp8 e4m3 (both K-contiguous), fp32
    accumulate, bf16 or fp32 result. ``alpha`` is where the product of the operands'
    dequantisation scales goes. Shapes the block-scaled-MFMA kernels (256x256x128 tiles, or
    128x128x128 with request-issuing waves when every such tile has a CU) cannot take run
    on the predicated generic kernel. ``splits``: K splits for grids with few output tiles
    (default: the library's plan; 1 = never split)."""
    a = _req(a, FP8, "gemm_fp8_nt a")
    bt = _req(bt, FP8, "gemm_fp8_nt bt")
    if a.dim() != 2 or bt.dim() != 2 or a.shape[1] != bt.shape[1]:
        raise HipKernelError("gemm_fp8_nt: need a [M,K] and bt [N,K], got %s and %s"
                             % (tuple(a.shape), tuple(bt.shape)))
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise HipKernelError("gemm_fp8_nt: out dtype must be bf16 or fp32")
    m, k = a.shape
    n = bt.shape[0]
    if beta != 0.0 and out is None:
        raise HipKernelError("gemm_fp8_nt: beta needs an out tensor to accumulate into")
    if out is None:
        out = torch.empty((m, n), dtype=out_dtype, device=a.device)
    elif out.shape != (m, n) or out.dtype != out_dtype or out.stride(1) != 1:
        raise HipKernelError("gemm_fp8_nt: bad out tensor")
    if m == 0 or n == 0:
        return out
    if k == 0:
        return out.mul_(beta) if beta != 0.0 else out.zero_()
    out_f32 = int(out_dtype == torch.float32)
    if not force_generic and m * n * k >= _BF16_PAD_MIN_WORK and k > 128:
        # as for bf16: operands off the fast kernel's grid (K % 128, 16-byte rows) are copied into
        # zero-padded aligned images, a result with an odd pitch goes through a padded temporary
        def operand_ok(t):
            return t.data_ptr() % 16 == 0 and t.stride(0) % 16 == 0 and k % 128 == 0 and \
                t.stride(0) * 256 < (1 << 32)
        out_ok = out.stride(0) % 4 == 0 and out.data_ptr() % (16 if out_f32 else 8) == 0
        if not (operand_ok(a) and operand_ok(bt) and out_ok) and (out_ok or beta == 0.0):
            kp = (k + 127) // 128 * 128
            images = []
            for t in (a, bt):
                if operand_ok(t):
                    images.append(t)
                    continue
                img = torch.empty((t.shape[0], kp), dtype=FP8, device=t.device)
                _check(lib().dtf_pad_2d_b8(t.data_ptr(), img.data_ptr(), t.shape[0], k,
                                           t.stride(0), t.shape[0], kp, _stream()),
                       "gemm_fp8_nt (aligned image)")
                images.append(img)
            if out_ok:
                return gemm_fp8_nt(images[0], images[1], out_dtype, alpha=alpha, out=out,
                                   beta=beta, splits=splits)
            n4 = (n + 3) // 4 * 4
            tmp = torch.empty((m, n4), dtype=out_dtype, device=a.device)
            gemm_fp8_nt(images[0], images[1], out_dtype, alpha=alpha, out=tmp[:, :n],
                        splits=splits)
            return copy_2d(tmp[:, :n], out)
    if splits is None:
        splits = 1 if force_generic else lib().dtf_gemm_fp8_splits(m, n, k)
    if splits > 1 and not force_generic:
        ws = torch.empty(splits * m * n, dtype=torch.float32, device=a.device)
        code = lib().dtf_gemm_fp8_nt_splitk(a.data_ptr(), bt.data_ptr(), out.data_ptr(),
                                            ws.data_ptr(), m, n, k, a.stride(0), bt.stride(0),
                                            out.stride(0), float(alpha), float(beta), out_f32,
                                            int(splits), _stream())
        if code == 0:
            return out
        if code != -2:
            _check(code, "gemm_fp8_nt (split-K)")
    code = -2 if force_generic else lib().dtf_gemm_fp8_nt(
        a.data_ptr(), bt.data_ptr(), out.data_ptr(), m, n, k, a.stride(0), bt.stride(0),
        out.stride(0), float(alpha), float(beta), out_f32, _stream())
    if code == -2:
        code = lib().dtf_gemm_fp8_generic(a.data_ptr(), bt.data_ptr(), out.data_ptr(), m, n, k,
                                          a.stride(0), 1, 1, bt.stride(0), out.stride(0),
                                          float(alpha), float(beta), out_f32, _stream())
    _check(code, "gemm_fp8_nt")
    return out


class MxFp8:
    """A matrix in block-scaled fp8 (OCP MX: e4m3 elements, one power-of-two E8M0 scale per 32
    consecutive elements of a row). ``q`` [R, K] fp8; ``scales`` the scale bytes of the matrix it
    was quantised as (``row_tiles`` 256-row tiles) in the order the GEMM reads them for an operand
    of this ``role`` (0: left operand, 1: the K-contiguous right operand), ``scale_offset`` the
    byte at which this (block of the) matrix starts in them; ``scales_rm`` (optional) the scale
    bytes as a plain [R, K/32] matrix. ``mx[r0:r1]`` is the block of rows r0..r1 (multiples of
    256) sharing the parent's storage."""

    __slots__ = ("q", "scales", "scales_rm", "role", "scale_offset", "row_tiles")

    def __init__(self, q, scales, scales_rm, role, scale_offset=0, row_tiles=None):
        self.q, self.scales, self.scales_rm, self.role = q, scales, scales_rm, role
        self.scale_offset = int(scale_offset)
        self.row_tiles = int(row_tiles) if row_tiles is not None else (q.shape[0] + 255) // 256

    @property
    def shape(self):
        return self.q.shape

    @property
    def dtype(self):
        return self.q.dtype

    def __getitem__(self, key) -> "MxFp8":
        if not isinstance(key, slice) or key.step not in (None, 1):
            raise HipKernelError("MxFp8: only row slices mx[r0:r1] are supported")
        r0, r1, _ = key.indices(self.q.shape[0])
        if r0 % 256 or (r1 % 256 and r1 != self.q.shape[0]) or r1 <= r0:
            raise HipKernelError("MxFp8: row blocks start and end on multiples of 256")
        per_tile = 2048 if self.role == 0 else 4096
        return MxFp8(self.q[r0:r1], self.scales,
                     None if self.scales_rm is None else self.scales_rm[r0:r1], self.role,
                     self.scale_offset + (r0 // 256) * per_tile, self.row_tiles)

    def dequantize(self) -> torch.Tensor:
        """fp32 values (needs ``scales_rm``); a torch reference for tests, not a kernel."""
        if self.scales_rm is None:
            raise HipKernelError("MxFp8.dequantize: quantise with want_rowmajor=True")
        r, k = self.q.shape
        f = torch.pow(2.0, self.scales_rm.to(torch.float32) - 127.0)
        return (self.q.to(torch.float32).view(r, k // 32, 32) * f.view(r, k // 32, 1)).view(r, k)


def quantize_mx_fp8(x: torch.Tensor, role: int, want_rowmajor: bool = False) -> MxFp8:
    """Quantise the rows of ``x`` [R, K] (fp32 or bf16, K % 128 == 0) to block-scaled fp8: per
    32 consecutive k the smallest power-of-two scale under which the block fits e4m3 without
    saturating. ``role`` 0 for the left operand of ``gemm_fp8_mx_nt``, 1 for the right one."""
    if not isinstance(x, torch.Tensor) or x.device.type != "cuda" or x.dim() != 2 \
            or x.dtype not in (torch.float32, torch.bfloat16):
        raise HipKernelError("quantize_mx_fp8: need a rank-2 fp32 or bf16 GPU tensor")
    if x.stride(1) != 1:
        x = x.contiguous()
    r, k = x.shape
    if r == 0 or k == 0 or k % 128:
        raise HipKernelError("quantize_mx_fp8: K must be a positive multiple of 128, got %s"
                             % (tuple(x.shape),))
    if role not in (0, 1):
        raise HipKernelError("quantize_mx_fp8: role is 0 (left operand) or 1 (right operand)")
    q = torch.empty((r, k), dtype=FP8, device=x.device)
    nbytes = int(lib().dtf_mx_scale_bytes(r, k))
    # rows past R in the last 256-row tile: unit scales
    scales = torch.full((nbytes,), 127, dtype=torch.uint8, device=x.device)
    rm = torch.empty((r, k // 32), dtype=torch.uint8, device=x.device) if want_rowmajor else None
    _check(lib().dtf_quantize_mx_fp8(x.data_ptr(), int(x.dtype == torch.bfloat16), r, k,
                                     x.stride(0), q.data_ptr(), k, scales.data_ptr(), _ptr(rm),
                                     int(role), _stream()), "quantize_mx_fp8")
    return MxFp8(q, scales, rm, role)


def gemm_fp8_mx_nt(a: MxFp8, bt: MxFp8, out_dtype: torch.dtype = torch.bfloat16,
                   alpha: float = 1.0, out: Optional[torch.Tensor] = None,
                   beta: float = 0.0) -> torch.Tensor:
    """``alpha * A @ Bt.T (+ beta * out)`` of two block-scaled fp8 matrices (``a`` quantised with
    role 0, ``bt`` [N, K] with role 1): the scales are applied by the matrix instruction
    (``v_mfma_scale_f32_16x16x128_f8f6f4``), fp32 accumulate, bf16 or fp32 result. M and N must
    be multiples of 256, K of 128."""
    if not isinstance(a, MxFp8) or not isinstance(bt, MxFp8) or a.role != 0 or bt.role != 1:
        raise HipKernelError("gemm_fp8_mx_nt: operands must be MxFp8 with roles 0 and 1")
    m, k = a.q.shape
    n = bt.q.shape[0]
    if bt.q.shape[1] != k:
        raise HipKernelError("gemm_fp8_mx_nt: inner dimensions differ")
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise HipKernelError("gemm_fp8_mx_nt: out dtype must be bf16 or fp32")
    if m % 256 or n % 256 or k % 128:
        raise HipKernelError("gemm_fp8_mx_nt: M and N must be multiples of 256 and K of 128 "
                             "(got %d x %d x %d)" % (m, n, k))
    if beta != 0.0 and out is None:
        raise HipKernelError("gemm_fp8_mx_nt: beta needs an out tensor to accumulate into")
    if out is None:
        out = torch.empty((m, n), dtype=out_dtype, device=a.q.device)
    elif out.shape != (m, n) or out.dtype != out_dtype or out.stride(1) != 1:
        raise HipKernelError("gemm_fp8_mx_nt: bad out tensor")
    _check(lib().dtf_gemm_fp8_mx_nt(a.q.data_ptr(), bt.q.data_ptr(), out.data_ptr(),
                                    a.scales.data_ptr() + a.scale_offset,
                                    bt.scales.data_ptr() + bt.scale_offset, m, n, k,
                                    a.q.stride(0), bt.q.stride(0), out.stride(0), a.row_tiles,
                                    bt.row_tiles, float(alpha), float(beta),
                                    int(out_dtype == torch.float32), _stream()),
           "gemm_fp8_mx_nt")
    return out


def quantize_fp8(x: torch.Tensor, scale: Optional[float] = None):
    """Per-tensor fp8 quantisation: returns ``(q, inv_scale)`` with ``q = e4m3(x * scale)``
    (saturating at +-448) and ``x ~= q * inv_scale``. ``scale`` defaults to ``448 / max|x|``."""
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise HipKernelError("quantize_fp8: source must be fp32 or bf16")
    x = _aligned(_req(x, x.dtype, "quantize_fp8 x"))
    if scale is None:
        # abs-max on the device (fixed-order kernel), one 12-byte read-back for the Python float
        amax = float(absmax_scale(x)[0].item()) if x.numel() else 0.0
        scale = FP8_MAX / amax if amax > 0 else 1.0
    out = torch.empty(x.shape, dtype=FP8, device=x.device)
    _check(lib().dtf_cast_to_fp8(x.data_ptr(), int(x.dtype == torch.bfloat16), out.data_ptr(),
                                 x.numel(), float(scale), _stream()), "quantize_fp8")
    return out, 1.0 / float(scale)


def dequantize_fp8(q: torch.Tensor, inv_scale: float = 1.0,
                   dtype: torch.dtype = torch.float32) -> torch.Tensor:
    q = _req(q, FP8, "dequantize_fp8 q")
    if dtype not in (torch.float32, torch.bfloat16):
        raise HipKernelError("dequantize_fp8: target must be fp32 or bf16")
    out = torch.empty(q.shape, dtype=dtype, device=q.device)
    _check(lib().dtf_cast_from_fp8(q.data_ptr(), out.data_ptr(), int(dtype == torch.bfloat16),
                                   q.numel(), float(inv_scale), _stream()), "dequantize_fp8")
    return out


def transpose2d(x: torch.Tensor) -> torch.Tensor:
    """Materialised transpose of a rank-2 tensor of 1-, 2- or 4-byte elements."""
    if not isinstance(x, torch.Tensor) or x.device.type != "cuda" or x.dim() != 2:
        raise HipKernelError("transpose2d: expected a rank-2 GPU tensor")
    if x.element_size() not in (1, 2, 4):
        raise HipKernelError("transpose2d: unsupported element size %d" % x.element_size())
    if x.stride(1) != 1:
        x = x.contiguous()
    rows, cols = x.shape
    out = torch.empty((cols, rows), dtype=x.dtype, device=x.device)
    _check(lib().dtf_transpose(x.data_ptr(), out.data_ptr(), rows, cols, x.stride(0),
                               max(1, rows), x.element_size(), _stream()), "transpose2d")
    return out


def gemm_bf16(a: torch.Tensor, b: torch.Tensor, trans_a: bool = False, trans_b: bool = False,
              out_dtype: Optional[torch.dtype] = None, alpha: float = 1.0,
              force_generic: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """General bf16 matmul ``op(a) @ op(b)``. The NT case (``trans_b`` with K-contiguous
    operands) is the fast kernel's native layout; for the other layouts a large product
    re-lays the offending operand out with one tiled transpose pass (memory-bound: ~3 % of the
    GEMM at 16384^3) and runs the same kernel, a small one runs the generic strided kernel."""
    a = _req(a, torch.bfloat16, "gemm_bf16 a")
    b = _req(b, torch.bfloat16, "gemm_bf16 b")
    if out_dtype is None:
        out_dtype = out.dtype if out is not None else torch.bfloat16
    if out is not None and (out.dtype != out_dtype or not out.is_contiguous()):
        raise HipKernelError("gemm_bf16: bad out tensor")
    if not trans_a and trans_b and not force_generic:
        return gemm_bf16_nt(a, b, out_dtype, alpha, out=out)
    if not force_generic and a.dim() == 2 and b.dim() == 2:
        m_, k_ = (a.shape[1], a.shape[0]) if trans_a else a.shape
        n_ = b.shape[0] if trans_b else b.shape[1]
        if k_ % 64 == 0 and 2.0 * m_ * n_ * k_ >= 2.0 * 512 ** 3:
            a_kc = transpose2d(a) if trans_a else a          # [M, K]
            bt_kc = b if trans_b else transpose2d(b)         # [N, K]
            return gemm_bf16_nt(a_kc, bt_kc, out_dtype, alpha, out=out)
    m, k = (a.shape[1], a.shape[0]) if trans_a else a.shape
    k2, n = (b.shape[1], b.shape[0]) if trans_b else b.shape
    if k != k2:
        raise HipKernelError("gemm_bf16: inner dimensions differ")
    if out is None:
        out = torch.empty((m, n), dtype=out_dtype, device=a.device)
    elif tuple(out.shape) != (m, n):
        raise HipKernelError("gemm_bf16: bad out tensor")
    if m == 0 or n == 0:
        return out
    a_rs, a_cs = (1, a.stride(0)) if trans_a else (a.stride(0), 1)
    b_rs, b_cs = (1, b.stride(0)) if trans_b else (b.stride(0), 1)
    _check(lib().dtf_gemm_bf16_generic(a.data_ptr(), b.data_ptr(), out.data_ptr(), m, n, k, a_rs,
                                       a_cs, b_rs, b_cs, out.stride(0), float(alpha), 0.0,
                                       int(out_dtype == torch.float32), _stream()), "gemm_bf16")
    return out


# ---------------------------------------------------------------------------------------
# elementwise
# ---------------------------------------------------------------------------------------
def _bcast_numel(out_shape: torch.Size, t: torch.Tensor, what: str) -> int:
    """``t`` may equal the output shape, be a trailing suffix of it, or be a scalar."""
    if tuple(t.shape) == tuple(out_shape):
        return t.numel()
    squeezed = [d for d in t.shape]
    while squeezed and squeezed[0] == 1:
        squeezed.pop(0)
    if tuple(squeezed) == tuple(out_shape[len(out_shape) - len(squeezed):]):
        return t.numel()
    raise HipKernelError("%s: broadcast of %s into %s is not a trailing-suffix broadcast"
                         % (what, tuple(t.shape), tuple(out_shape)))


def binary(op: str, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    a = _req(a, torch.float32, "binary a")
    b = _req(b, torch.float32, "binary b")
    out_shape = _broadcast_shape(a.shape, b.shape)
    an = _bcast_numel(out_shape, a, "binary a")
    bn = _bcast_numel(out_shape, b, "binary b")
    out = torch.empty(out_shape, dtype=torch.float32, device=a.device)
    _check(lib().dtf_binary_f32(BINARY[op], a.data_ptr(), b.data_ptr(), out.data_ptr(),
                                out.numel(), an, bn, _stream()), "binary " + op)
    return out


def compare(op: str, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    if a.dtype == torch.int64 and b.dtype == torch.int64 and op == "eq" and a.shape == b.shape:
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty(a.shape, dtype=torch.bool, device=a.device)
        _check(lib().dtf_equal_i64(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(),
                                   _stream()), "equal_i64")
        return out
    a = _req(a, torch.float32, "compare a")
    b = _req(b, torch.float32, "compare b")
    out_shape = _broadcast_shape(a.shape, b.shape)
    an = _bcast_numel(out_shape, a, "compare a")
    bn = _bcast_numel(out_shape, b, "compare b")
    out = torch.empty(out_shape, dtype=torch.bool, device=a.device)
    _check(lib().dtf_compare_f32(COMPARE[op], a.data_ptr(), b.data_ptr(), out.data_ptr(),
                                 out.numel(), an, bn, _stream()), "compare " + op)
    return out


def select(cond: torch.Tensor, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``cond ? a : b`` element-wise (fp32 values, bool condition); an operand whose element
    count divides the result's trailing layout -- a scalar, a row -- is repeated."""
    if cond.dtype != torch.bool or cond.device.type != "cuda":
        raise HipKernelError("select: cond must be a bool GPU tensor")
    a = _req(a, torch.float32, "select a")
    b = _req(b, torch.float32, "select b")
    out_shape = _broadcast_shape(_broadcast_shape(cond.shape, a.shape), b.shape)
    cond = cond.contiguous()
    cn = _bcast_numel(out_shape, cond, "select cond")
    an = _bcast_numel(out_shape, a, "select a")
    bn = _bcast_numel(out_shape, b, "select b")
    out = torch.empty(out_shape, dtype=torch.float32, device=a.device)
    _check(lib().dtf_select_f32(cond.data_ptr(), a.data_ptr(), b.data_ptr(), out.data_ptr(),
                                out.numel(), cn, an, bn, _stream()), "select")
    return out


def unary(op: str, x: torch.Tensor) -> torch.Tensor:
    x = _aligned(_req(x, torch.float32, "unary x"))
    out = torch.empty_like(x)
    _check(lib().dtf_unary_f32(UNARY[op], x.data_ptr(), out.data_ptr(), x.numel(), _stream()),
           "unary " + op)
    return out


def axpby(alpha: float, x: torch.Tensor, beta: float = 0.0, y: Optional[torch.Tensor] = None,
          out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``alpha*x + beta*y`` (fp32 or bf16; ``y`` optional)."""
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise HipKernelError("axpby: dtype must be fp32 or bf16")
    x = _aligned(_req(x, x.dtype, "axpby x"))
    if y is not None:
        y = _aligned(_req(y, x.dtype, "axpby y"))
        if y.shape != x.shape:
            raise HipKernelError("axpby: shapes differ")
    if out is None:
        out = torch.empty_like(x)
    elif out.shape != x.shape or out.dtype != x.dtype or not out.is_contiguous():
        raise HipKernelError("axpby: bad out tensor")
    fn = lib().dtf_axpby_f32 if x.dtype == torch.float32 else lib().dtf_axpby_bf16
    _check(fn(float(alpha), x.data_ptr(), float(beta), _ptr(y), out.data_ptr(), x.numel(),
              _stream()), "axpby")
    return out


_CAST_KINDS = {torch.float32: 0, torch.bfloat16: 1, torch.int32: 2, torch.int64: 3, torch.bool: 4,
               torch.float64: 5, torch.float8_e4m3fn: 6}


def cast(x: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if x.dtype == dtype and out is None:
        return x
    x = x.contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=dtype, device=x.device)
    elif out.shape != x.shape or out.dtype != dtype or not out.is_contiguous():
        raise HipKernelError("cast: bad out tensor")
    n = x.numel()
    if x.dtype == torch.float32 and dtype == torch.bfloat16:
        _check(lib().dtf_cast_f32_to_bf16(_aligned(x).data_ptr(), out.data_ptr(), n, _stream()),
               "cast f32->bf16")
    elif x.dtype == torch.bfloat16 and dtype == torch.float32:
        _check(lib().dtf_cast_bf16_to_f32(x.data_ptr(), out.data_ptr(), n, _stream()),
               "cast bf16->f32")
    elif x.dtype == torch.bool and dtype == torch.float32:
        _check(lib().dtf_cast_bool_to_f32(x.data_ptr(), out.data_ptr(), n, _stream()),
               "cast bool->f32")
    elif x.dtype in _CAST_KINDS and dtype in _CAST_KINDS:
        _check(lib().dtf_cast_generic(x.data_ptr(), _CAST_KINDS[x.dtype], out.data_ptr(),
                                      _CAST_KINDS[dtype], n, _stream()),
               "cast %s->%s" % (x.dtype, dtype))
    else:
        raise HipKernelError("cast %s -> %s has no HIP kernel" % (x.dtype, dtype))
    return out


def fill(shape: Sequence[int], value: float, device) -> torch.Tensor:
    out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    _check(lib().dtf_fill_f32(out.data_ptr(), float(value), out.numel(), _stream()), "fill")
    return out


def row_scale(x: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    x = _req(x, torch.float32, "row_scale x")
    s = _req(s, torch.float32, "row_scale s")
    if x.dim() != 2 or s.numel() != x.shape[0]:
        raise HipKernelError("row_scale: need x [R,C] and s [R]")
    out = torch.empty_like(x)
    _check(lib().dtf_row_scale_f32(x.data_ptr(), s.data_ptr(), out.data_ptr(), x.shape[0],
                                   x.shape[1], _stream()), "row_scale")
    return out


# ---------------------------------------------------------------------------------------
# reductions / softmax / loss
# ---------------------------------------------------------------------------------------
def _as_rows(x: torch.Tensor) -> torch.Tensor:
    return x.reshape(-1, x.shape[-1]) if x.dim() != 2 else x


def reduce_rows(x: torch.Tensor, op: int) -> torch.Tensor:
    x = _req(x, torch.float32, "reduce_rows x")
    if x.dim() != 2:
        raise HipKernelError("reduce_rows: rank-2 input required")
    out = torch.empty((x.shape[0],), dtype=torch.float32, device=x.device)
    _check(lib().dtf_reduce_rows_f32(op, x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1],
                                     _stream()), "reduce_rows")
    return out


def argmax_rows(x: torch.Tensor) -> torch.Tensor:
    x = _req(x, torch.float32, "argmax_rows x")
    if x.dim() != 2 or x.shape[1] == 0:
        raise HipKernelError("argmax_rows: rank-2 input with >=1 column required")
    out = torch.empty((x.shape[0],), dtype=torch.int64, device=x.device)
    _check(lib().dtf_argmax_rows_f32(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1],
                                     _stream()), "argmax_rows")
    return out


def reduce_all(x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """``scale * sum(x)`` as a 0-d tensor; fixed-order (bitwise reproducible) fold."""
    x = _req(x, torch.float32, "reduce_all x")
    n = x.numel()
    out = torch.empty((), dtype=torch.float32, device=x.device)
    if n == 0:
        return out.zero_()
    ws = torch.empty((lib().dtf_reduce_all_workspace(n),), dtype=torch.float32, device=x.device)
    _check(lib().dtf_reduce_all_f32(x.data_ptr(), out.data_ptr(), n, float(scale), ws.data_ptr(),
                                    _stream()), "reduce_all")
    return out


def _out_vec(out: Optional[torch.Tensor], n: int, device, what: str) -> torch.Tensor:
    """``out`` (a caller-placed contiguous fp32 buffer of n elements, e.g. a gradient's slot in
    the flat gradient arena) or a fresh vector."""
    if out is None:
        return torch.empty((n,), dtype=torch.float32, device=device)
    if out.dtype != torch.float32 or out.numel() != n or not out.is_contiguous() \
            or out.device.type != "cuda":
        raise HipKernelError("%s: bad out tensor" % what)
    return out


def reduce_cols(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = _req(x, torch.float32, "reduce_cols x")
    if x.dim() != 2:
        raise HipKernelError("reduce_cols: rank-2 input required")
    out = _out_vec(out, x.shape[1], x.device, "reduce_cols")
    _check(lib().dtf_reduce_cols_f32(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1],
                                     _stream()), "reduce_cols")
    return out


def sigmoid_grad_bias(da: torch.Tensor, a: torch.Tensor, want_bias: bool = True,
                      db_out: Optional[torch.Tensor] = None):
    da = _req(da, torch.float32, "sigmoid_grad_bias da")
    a = _req(a, torch.float32, "sigmoid_grad_bias a")
    if da.dim() != 2 or da.shape != a.shape:
        raise HipKernelError("sigmoid_grad_bias: need matching rank-2 inputs")
    dz = torch.empty_like(da)
    db = _out_vec(db_out, da.shape[1], da.device, "sigmoid_grad_bias") if want_bias else None
    _check(lib().dtf_sigmoid_grad_bias_f32(da.data_ptr(), a.data_ptr(), dz.data_ptr(), _ptr(db),
                                           da.shape[0], da.shape[1], _stream()),
           "sigmoid_grad_bias")
    return dz, db


def softmax_rows(x: torch.Tensor, log: bool = False) -> torch.Tensor:
    x = _req(x, torch.float32, "softmax x")
    x2 = _as_rows(x)
    out = torch.empty_like(x2)
    _check(lib().dtf_softmax_rows_f32(x2.data_ptr(), None if log else out.data_ptr(),
                                      out.data_ptr() if log else None, x2.shape[0], x2.shape[1],
                                      _stream()), "softmax_rows")
    return out.view(x.shape)


def softmax_grad(y: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    y = _req(y, torch.float32, "softmax_grad y")
    dy = _req(dy, torch.float32, "softmax_grad dy")
    if y.shape != dy.shape:
        raise HipKernelError("softmax_grad: shapes differ")
    y2, dy2 = _as_rows(y), _as_rows(dy)
    dx = torch.empty_like(y2)
    _check(lib().dtf_softmax_grad_f32(y2.data_ptr(), dy2.data_ptr(), dx.data_ptr(), y2.shape[0],
                                      y2.shape[1], _stream()), "softmax_grad")
    return dx.view(y.shape)


def softmax_xent(logits: torch.Tensor, labels: torch.Tensor, want_softmax: bool = True,
                 want_rows: bool = False, want_mean: bool = True):
    """Fused softmax + cross-entropy. Returns ``(mean_loss, softmax, row_loss)`` with ``None``
    for the parts not requested."""
    logits = _req(logits, torch.float32, "softmax_xent logits")
    labels = _req(labels, torch.float32, "softmax_xent labels")
    if logits.dim() != 2 or logits.shape != labels.shape:
        raise HipKernelError("softmax_xent: need matching rank-2 logits/labels")
    r, c = logits.shape
    dev = logits.device
    y = torch.empty_like(logits) if want_softmax else None
    rows = torch.empty((r,), dtype=torch.float32, device=dev) if want_rows else None
    mean = torch.empty((), dtype=torch.float32, device=dev) if want_mean else None
    ws = torch.empty((lib().dtf_softmax_xent_workspace(r),), dtype=torch.float32, device=dev) \
        if want_mean else None
    _check(lib().dtf_softmax_xent_f32(logits.data_ptr(), labels.data_ptr(), _ptr(y), _ptr(rows),
                                      _ptr(mean), _ptr(ws), r, c, _stream()), "softmax_xent")
    return mean, y, rows


def softmax_xent_grad(y: torch.Tensor, labels: torch.Tensor, dloss: torch.Tensor,
                      want_bias: bool = True, db_out: Optional[torch.Tensor] = None):
    """``dz = (y*rowsum(labels) - labels) * dloss / R`` and its column sum (C <= 64)."""
    y = _req(y, torch.float32, "softmax_xent_grad y")
    labels = _req(labels, torch.float32, "softmax_xent_grad labels")
    dloss = _req(dloss, torch.float32, "softmax_xent_grad dloss")
    if y.dim() != 2 or y.shape != labels.shape or dloss.numel() != 1:
        raise HipKernelError("softmax_xent_grad: bad shapes")
    r, c = y.shape
    if c > 64:
        dz = binary("sub", binary("mul", y, reduce_rows(labels, ROW_SUM).unsqueeze(1)
                                  .expand_as(y).contiguous()), labels)
        dz = binary("mul", dz, axpby(1.0 / r, dloss.reshape(1)))
        return dz, (reduce_cols(dz, out=db_out) if want_bias else None)
    dz = torch.empty_like(y)
    db = _out_vec(db_out, c, y.device, "softmax_xent_grad") if want_bias else None
    fn = lib().dtf_xent_grad_narrow_f32 if c <= 32 else lib().dtf_softmax_xent_grad_f32
    _check(fn(y.data_ptr(), labels.data_ptr(), dloss.data_ptr(), dz.data_ptr(), _ptr(db), r, c,
              _stream()), "softmax_xent_grad")
    return dz, db


def mlp_head_supported(b: int, h: int, c: int) -> bool:
    return bool(lib().dtf_mlp_head_supported(b, h, c))


def mlp_head_train(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, labels: torch.Tensor,
                   dloss: torch.Tensor, sigmoid_input: bool, want_accuracy: bool,
                   dw_out: Optional[torch.Tensor] = None, db_out: Optional[torch.Tensor] = None,
                   dbp_out: Optional[torch.Tensor] = None, ring: Optional["RingArgs"] = None,
                   labels_index: int = -1):
    """Fused forward + backward of a small softmax head (B <= 128, H <= 128, C <= 16).
    With ``ring`` the labels are read through entry ``labels_index`` of the step ring's address
    table (``labels`` then only gives their shape).
    Returns ``(loss, y, dw, db, da_or_dz_prev, db_prev_or_None, accuracy_or_None)``. The three
    parameter gradients may be placed by the caller (``*_out``: contiguous fp32 buffers of the
    right size, e.g. their slots in a flat gradient arena)."""
    a = _req(a, torch.float32, "mlp_head a")
    w = _req(w, torch.float32, "mlp_head w")
    bias = _req(bias, torch.float32, "mlp_head bias")
    labels = _req(labels, torch.float32, "mlp_head labels")
    dloss = _req(dloss, torch.float32, "mlp_head dloss")
    if a.dim() != 2 or w.dim() != 2 or a.shape[1] != w.shape[0] or bias.numel() != w.shape[1] \
            or tuple(labels.shape) != (a.shape[0], w.shape[1]) or dloss.numel() != 1:
        raise HipKernelError("mlp_head_train: inconsistent shapes a%s w%s bias%s labels%s"
                             % (tuple(a.shape), tuple(w.shape), tuple(bias.shape),
                                tuple(labels.shape)))
    b, h = a.shape
    c = w.shape[1]
    dev = a.device
    ws_n = int(lib().dtf_mlp_head_workspace(b, h, c))
    buf = torch.empty(2 + b * c + h * c + c + b * h + h + ws_n, dtype=torch.float32, device=dev)
    off = 0

    def take(n, shape):
        nonlocal off
        t = buf[off:off + n].view(shape)
        off += n
        return t

    loss, acc = take(1, ()), take(1, ())
    y, dw, db = take(b * c, (b, c)), take(h * c, (h, c)), take(c, (c,))
    da, dbp = take(b * h, (b, h)), take(h, (h,))
    if dw_out is not None:
        dw = _out_vec(dw_out.view(-1), h * c, dev, "mlp_head dw").view(h, c)
    if db_out is not None:
        db = _out_vec(db_out, c, dev, "mlp_head db")
    if dbp_out is not None:
        dbp = _out_vec(dbp_out, h, dev, "mlp_head dbp")
    ws = take(ws_n, (ws_n,))
    if ring is not None:
        if ring.take_first():
            raise HipKernelError("mlp_head_train: cannot be the first ring kernel of a step")
        _check(lib().dtf_mlp_head_train_ring(
            a.data_ptr(), w.data_ptr(), bias.data_ptr(), dloss.data_ptr(), loss.data_ptr(),
            y.data_ptr(), dw.data_ptr(), db.data_ptr(), da.data_ptr(),
            dbp.data_ptr() if sigmoid_input else None,
            acc.data_ptr() if want_accuracy else None, ws.data_ptr(), b, h, c,
            int(sigmoid_input), ring.table.data_ptr(), ring.ctr.data_ptr(), ring.out.data_ptr(),
            ring.slots, int(labels_index), _stream()), "mlp_head_train (ring)")
        return loss, y, dw, db, da, (dbp if sigmoid_input else None), \
            (acc if want_accuracy else None)
    _check(lib().dtf_mlp_head_train(a.data_ptr(), w.data_ptr(), bias.data_ptr(),
                                    labels.data_ptr(), dloss.data_ptr(), loss.data_ptr(),
                                    y.data_ptr(), dw.data_ptr(), db.data_ptr(), da.data_ptr(),
                                    dbp.data_ptr() if sigmoid_input else None,
                                    acc.data_ptr() if want_accuracy else None, ws.data_ptr(),
                                    b, h, c, int(sigmoid_input), _stream()), "mlp_head_train")
    return loss, y, dw, db, da, (dbp if sigmoid_input else None), (acc if want_accuracy else None)


def argmax_accuracy(y: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    y = _req(y, torch.float32, "argmax_accuracy y")
    labels = _req(labels, torch.float32, "argmax_accuracy labels")
    if y.dim() != 2 or y.shape != labels.shape or y.shape[1] == 0:
        raise HipKernelError("argmax_accuracy: need matching rank-2 inputs")
    ws_n = int(lib().dtf_argmax_accuracy_workspace(y.shape[0]))
    buf = torch.empty(1 + ws_n, dtype=torch.float32, device=y.device)
    out = buf[0].view(())
    _check(lib().dtf_argmax_accuracy_f32(y.data_ptr(), labels.data_ptr(), out.data_ptr(),
                                         buf[1:].data_ptr(), y.shape[0], y.shape[1], _stream()),
           "argmax_accuracy")
    return out


# ---------------------------------------------------------------------------------------
# optimizer / rng / partial sums
# ---------------------------------------------------------------------------------------
def multi_sgd(params: Sequence[torch.Tensor], grads: Sequence[torch.Tensor],
              lrs: Sequence[float], counters: Sequence[torch.Tensor] = (),
              deltas: Sequence[float] = ()) -> None:
    """In place ``p -= lr * g`` for every pair and ``counter += delta`` -- one launch per 16
    tensors."""
    if len(params) != len(grads) or len(params) != len(lrs) or len(counters) != len(deltas):
        raise HipKernelError("multi_sgd: argument lists differ in length")
    for p, g in zip(params, grads):
        if p.dtype != torch.float32 or g.dtype != torch.float32 or not p.is_contiguous() \
                or p.numel() != g.numel() or p.device.type != "cuda":
            raise HipKernelError("multi_sgd: params/grads must be matching contiguous fp32 GPU "
                                 "tensors")
    for c in counters:
        if c.dtype != torch.float32 or c.numel() != 1 or c.device.type != "cuda":
            raise HipKernelError("multi_sgd: counters must be fp32 GPU scalars")
    if len(counters) > 4:
        raise HipKernelError("multi_sgd: at most 4 counters")
    grads = [g if g.is_contiguous() else g.contiguous() for g in grads]
    c = ctypes
    first = True
    for start in range(0, max(len(params), 1), 16):
        ps, gs, ls = params[start:start + 16], grads[start:start + 16], lrs[start:start + 16]
        n = len(ps)
        w_arr = (c.c_void_p * max(n, 1))(*[p.data_ptr() for p in ps])
        g_arr = (c.c_void_p * max(n, 1))(*[g.data_ptr() for g in gs])
        n_arr = (c.c_long * max(n, 1))(*[p.numel() for p in ps])
        lr_arr = (c.c_float * max(n, 1))(*[float(x) for x in ls])
        nc = len(counters) if first else 0
        c_arr = (c.c_void_p * max(nc, 1))(*[t.data_ptr() for t in counters][:nc])
        d_arr = (c.c_float * max(nc, 1))(*[float(d) for d in deltas][:nc])
        _check(lib().dtf_multi_sgd_f32(n, w_arr, g_arr, n_arr, lr_arr, nc, c_arr, d_arr,
                                       _stream()), "multi_sgd")
        first = False


OPTIM_RULES = {"momentum": 0, "nesterov": 1, "adagrad": 2, "rmsprop": 3, "adam": 4, "sgd": 5,
               "adadelta": 6, "ema": 7}


def multi_optim(rule: str, params: Sequence[torch.Tensor], grads: Sequence[torch.Tensor],
                slots0: Sequence[torch.Tensor], slots1: Sequence[torch.Tensor] = (),
                hyper: Sequence[float] = (), powers: Sequence[torch.Tensor] = (),
                lr: Optional[torch.Tensor] = None) -> None:
    """In-place update of every ``params[i]`` (and its state ``slots0[i]`` / ``slots1[i]``) from
    ``grads[i]`` by one of the slot-carrying rules -- one launch per 16 tensors. ``hyper``:
    momentum / nesterov (lr, mu); adagrad (lr,); rmsprop (lr, rho, mu, eps); adam (lr, b1, b2,
    eps) with ``powers`` = the device scalars holding b1^t and b2^t (read, not advanced: see
    ``affine_scalars``); adadelta (lr, rho, eps); sgd (lr,) takes no slots; ema (rate,) takes
    no slots either: ``params[i] -= rate * (params[i] - grads[i])``, the moving average
    ``params[i]`` of the tensor ``grads[i]``. ``lr``: a device scalar that replaces the
    constant learning rate (a schedule computed in the step's graph), read when the kernel runs."""
    code = OPTIM_RULES.get(rule)
    if code is None:
        raise HipKernelError("multi_optim: unknown rule %r" % (rule,))
    two = rule in ("rmsprop", "adam", "adadelta")
    none = rule in ("sgd", "ema")
    if len(params) != len(grads) or (not none and len(params) != len(slots0)) or \
            (two and len(slots1) != len(params)):
        raise HipKernelError("multi_optim: argument lists differ in length")
    if lr is not None and (lr.dtype != torch.float32 or lr.numel() != 1
                           or lr.device.type != "cuda"):
        raise HipKernelError("multi_optim: lr must be an fp32 GPU scalar")
    if rule == "adam" and len(powers) != 2:
        raise HipKernelError("multi_optim: adam needs the two power scalars")
    want = {"momentum": 2, "nesterov": 2, "adagrad": 1, "rmsprop": 4, "adam": 4, "sgd": 1,
            "adadelta": 3, "ema": 1}[rule]
    if len(hyper) != want:
        raise HipKernelError("multi_optim: %s takes %d constants" % (rule, want))
    grads = [g if g.is_contiguous() else g.contiguous() for g in grads]
    for group in zip(params, grads, *([] if none else [slots0]), *([slots1] if two else [])):
        p = group[0]
        for t in group:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != p.numel() \
                    or t.device.type != "cuda":
                raise HipKernelError("multi_optim: params, grads and slots must be matching "
                                     "contiguous fp32 GPU tensors")
    for t in powers:
        if t.dtype != torch.float32 or t.numel() != 1 or t.device.type != "cuda":
            raise HipKernelError("multi_optim: powers must be fp32 GPU scalars")
    c = ctypes
    hyper = [float(x) for x in hyper]
    if rule == "rmsprop":
        hyper.append(1.0 - hyper[1])              # complements rounded once, from double
    elif rule == "adam":
        hyper += [1.0 - hyper[1], 1.0 - hyper[2]]
    elif rule == "adadelta":
        hyper.append(1.0 - hyper[1])
    h_arr = (c.c_float * max(len(hyper), 1))(*hyper)
    pw = [t.data_ptr() for t in powers] + [None, None]
    for start in range(0, len(params), 16):
        sl = slice(start, start + 16)
        ps, gs = params[sl], grads[sl]
        n = len(ps)
        w_arr = (c.c_void_p * n)(*[t.data_ptr() for t in ps])
        g_arr = (c.c_void_p * n)(*[t.data_ptr() for t in gs])
        a_arr = None if none else (c.c_void_p * n)(*[t.data_ptr() for t in slots0[sl]])
        b_arr = (c.c_void_p * n)(*[t.data_ptr() for t in slots1[sl]]) if two else None
        n_arr = (c.c_long * n)(*[t.numel() for t in ps])
        _check(lib().dtf_multi_optim_f32(code, n, w_arr, g_arr, a_arr, b_arr, n_arr, h_arr,
                                         len(hyper), pw[0], pw[1],
                                         None if lr is None else lr.data_ptr(), _stream()),
               "multi_optim")


def affine_scalars(scalars: Sequence[torch.Tensor], factors: Sequence[float],
                   increments: Sequence[float]) -> None:
    """``scalars[i] = scalars[i] * factors[i] + increments[i]`` for up to 8 distinct fp32 device
    scalars in one one-wave launch (Adam's running powers, step counters)."""
    if len(scalars) != len(factors) or len(scalars) != len(increments) or len(scalars) > 8:
        raise HipKernelError("affine_scalars: up to 8 scalars with one factor and increment each")
    for t in scalars:
        if t.dtype != torch.float32 or t.numel() != 1 or t.device.type != "cuda":
            raise HipKernelError("affine_scalars: fp32 GPU scalars")
    if not scalars:
        return
    c = ctypes
    p_arr = (c.c_void_p * len(scalars))(*[t.data_ptr() for t in scalars])
    f_arr = (c.c_float * len(scalars))(*[float(f) for f in factors])
    d_arr = (c.c_float * len(scalars))(*[float(d) for d in increments])
    _check(lib().dtf_affine_scalars_f32(len(scalars), p_arr, f_arr, d_arr, _stream()),
           "affine_scalars")


def dropout(x: torch.Tensor, keep_prob: float, seed: int, op_seed: int, round_counter: torch.Tensor,
            keep_tensor: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """``(x * mask, mask)`` with ``mask = (u < keep) / keep`` from round ``round_counter[0]`` of
    the ``(seed, op_seed)`` Philox stream. ``round_counter`` is a one-element int64 device tensor
    (see :func:`bump_counter`; the caller bumps it after the call): an integer, because an fp32
    counter advanced by one sticks at 2^24 and would repeat one mask from then on.
    ``keep_tensor``: an fp32 device scalar that replaces ``keep_prob`` when the kernel runs."""
    x = _req(x, torch.float32, "dropout x")
    for t, what, dt in ((round_counter, "round counter", torch.int64),
                        (keep_tensor, "keep_prob", torch.float32)):
        if t is not None and (t.dtype != dt or t.numel() != 1 or t.device.type != "cuda"):
            raise HipKernelError("dropout: %s must be a one-element %s GPU tensor" % (what, dt))
    out, mask = torch.empty_like(x), torch.empty_like(x)
    _check(lib().dtf_dropout_f32(x.data_ptr(), out.data_ptr(), mask.data_ptr(), x.numel(),
                                 seed & 0xFFFFFFFF, op_seed & 0xFFFFFFFF,
                                 round_counter.data_ptr(), float(keep_prob),
                                 None if keep_tensor is None else keep_tensor.data_ptr(),
                                 _stream()), "dropout")
    return out, mask


def bump_counter(counter: torch.Tensor, by: int = 1) -> torch.Tensor:
    """``counter[0] += by`` for a one-element int64 device tensor, in one stream-ordered launch
    (capturable: every replay of a HIP graph advances the counter again)."""
    if counter.dtype != torch.int64 or counter.numel() != 1 or counter.device.type != "cuda":
        raise HipKernelError("bump_counter: the counter must be a one-element int64 GPU tensor")
    _check(lib().dtf_bump_counter_u64(counter.data_ptr(), int(by), _stream()), "bump_counter")
    return counter


def philox_normal(shape: Sequence[int], seed: int, op_seed: int, mean: float = 0.0,
                  stddev: float = 1.0, dtype: torch.dtype = torch.float32, device=None,
                  offset: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if dtype not in (torch.float32, torch.bfloat16):
        raise HipKernelError("philox_normal: dtype must be fp32 or bf16")
    if out is None:
        out = torch.empty(tuple(shape), dtype=dtype, device=device)
    _check(lib().dtf_philox_normal(out.data_ptr(), out.numel(), seed & 0xFFFFFFFF,
                                   op_seed & 0xFFFFFFFF, int(offset), float(mean), float(stddev),
                                   int(dtype == torch.bfloat16), _stream()), "philox_normal")
    return out


def philox_uniform(shape: Sequence[int], seed: int, op_seed: int, minval: float = 0.0,
                   maxval: float = 1.0, dtype: torch.dtype = torch.float32, device=None,
                   offset: int = 0) -> torch.Tensor:
    if dtype not in (torch.float32, torch.bfloat16):
        raise HipKernelError("philox_uniform: dtype must be fp32 or bf16")
    out = torch.empty(tuple(shape), dtype=dtype, device=device)
    _check(lib().dtf_philox_uniform(out.data_ptr(), out.numel(), seed & 0xFFFFFFFF,
                                    op_seed & 0xFFFFFFFF, int(offset), float(minval),
                                    float(maxval), int(dtype == torch.bfloat16), _stream()),
           "philox_uniform")
    return out


def reduce_partials(parts: Sequence[torch.Tensor], scale: float = 1.0,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``scale * sum(parts)`` for up to 8 same-shape fp32 / bf16 buffers (fp32 accumulate)."""
    if not 1 <= len(parts) <= 8:
        raise HipKernelError("reduce_partials: 1..8 buffers")
    dt = parts[0].dtype
    if dt not in (torch.float32, torch.bfloat16):
        raise HipKernelError("reduce_partials: dtype must be fp32 or bf16")
    parts = [_aligned(_req(p, dt, "reduce_partials part")) for p in parts]
    if any(p.shape != parts[0].shape for p in parts):
        raise HipKernelError("reduce_partials: shapes differ")
    if out is None:
        out = torch.empty_like(parts[0])
    elif out.shape != parts[0].shape or out.dtype != dt or not out.is_contiguous():
        raise HipKernelError("reduce_partials: bad out tensor")
    arr = (ctypes.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
    _check(lib().dtf_reduce_partials(len(parts), arr, out.data_ptr(), out.numel(), float(scale),
                                     int(dt == torch.bfloat16), _stream()), "reduce_partials")
    return out


def pack_f32(parts: Sequence[torch.Tensor], out: torch.Tensor, signal=None,
             adds: Optional[Sequence[float]] = None) -> List[int]:
    """Gather up to 16 small fp32 tensors into the head of ``out`` (a GPU tensor or a pinned
    host tensor) with one launch; returns the offset of each part. ``signal = (seq, index)``:
    after the gather the kernel copies the current value of the one-word pinned tensor ``seq``
    into ``out[index]`` behind a system-scope release, so a host polling that word knows the
    packed values are complete when it sees the number it stored in ``seq`` before the launch
    (or before the replay of a graph that captured it). ``adds[i]`` is added to every element
    of part i (0 by default): a step counter can be fetched ahead of the launch that bumps it."""
    if not 1 <= len(parts) <= 16:
        raise HipKernelError("pack_f32: 1..16 parts")
    parts = [_req(p, torch.float32, "pack_f32 part") for p in parts]
    total = sum(p.numel() for p in parts)
    # ``out`` lives in GPU memory or in pinned host memory: pinned memory is mapped into the
    # GPU's address space, so the kernel can store the packed words straight into the buffer the
    # host reads after the stream synchronisation -- no separate device->host copy
    if out.dtype != torch.float32 or out.numel() < total or not out.is_contiguous() \
            or not (out.device.type == "cuda" or out.is_pinned()):
        raise HipKernelError("pack_f32: bad out tensor")
    src = (ctypes.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
    sizes = (ctypes.c_int * len(parts))(*[p.numel() for p in parts])
    seq_src = seq_dst = None
    if signal is not None:
        seq, index = signal
        if seq.dtype != torch.float32 or seq.numel() != 1 or not (seq.is_pinned() or seq.is_cuda) \
                or not total <= index < out.numel():
            raise HipKernelError("pack_f32: bad signal")
        seq_src, seq_dst = seq.data_ptr(), out.data_ptr() + 4 * int(index)
    add_arr = None
    if adds is not None:
        if len(adds) != len(parts):
            raise HipKernelError("pack_f32: one addend per part")
        add_arr = (ctypes.c_float * len(parts))(*[float(a) for a in adds])
    _check(lib().dtf_pack_f32_signal(len(parts), src, sizes, add_arr, out.data_ptr(), seq_src,
                                     seq_dst, _stream()), "pack_f32")
    offsets, off = [], 0
    for p in parts:
        offsets.append(off)
        off += p.numel()
    return offsets


def peer_push(src: torch.Tensor, dsts: Sequence[torch.Tensor], blocks_per_peer: int = 8) -> None:
    """Copy the contiguous GPU tensor ``src`` into each of ``dsts`` (1..8 contiguous tensors of
    the same dtype and size, typically peer-mapped buffers of other GPUs) with one kernel on the
    current stream; ``blocks_per_peer`` workgroups serve each destination."""
    dsts = list(dsts)
    if not 1 <= len(dsts) <= 8:
        raise HipKernelError("peer_push: 1..8 destinations")
    if not (isinstance(src, torch.Tensor) and src.device.type == "cuda" and src.is_contiguous()):
        raise HipKernelError("peer_push: src must be a contiguous GPU tensor")
    nbytes = src.numel() * src.element_size()
    for d in dsts:
        if not (isinstance(d, torch.Tensor) and d.device.type == "cuda" and d.is_contiguous()
                and d.dtype == src.dtype and d.numel() == src.numel()):
            raise HipKernelError("peer_push: destinations must be contiguous GPU tensors of the "
                                 "source's dtype and size")
    if src.data_ptr() % 16 or any(d.data_ptr() % 16 for d in dsts):
        raise HipKernelError("peer_push: buffers must be 16-byte aligned")
    ptrs = (ctypes.c_void_p * len(dsts))(*[d.data_ptr() for d in dsts])
    _check(lib().dtf_peer_push(src.data_ptr(), ptrs, len(dsts), nbytes, int(blocks_per_peer),
                               _stream()), "peer_push")


def gather_feeds(table: torch.Tensor, dsts: Sequence[torch.Tensor]) -> None:
    """Launch the kernel that copies ``len(dsts)`` (<= 8) inputs into the contiguous GPU tensors
    ``dsts`` from the source ADDRESSES stored in ``table`` (int64, pinned host or GPU memory,
    one per destination). The table is read when the kernel runs: a captured launch picks up
    whatever addresses the host stored before the replay. Sources must hold at least as many
    bytes as their destination and stay alive until the kernel has run."""
    dsts = list(dsts)
    if not 1 <= len(dsts) <= 8:
        raise HipKernelError("gather_feeds: 1..8 destinations")
    if table.dtype != torch.int64 or table.numel() < len(dsts) or not table.is_contiguous() \
            or not (table.device.type == "cuda" or table.is_pinned()):
        raise HipKernelError("gather_feeds: table must be int64 in pinned host or GPU memory")
    for d in dsts:
        if not (isinstance(d, torch.Tensor) and d.device.type == "cuda" and d.is_contiguous()
                and d.numel() > 0 and d.data_ptr() % 16 == 0):
            raise HipKernelError("gather_feeds: destinations must be non-empty contiguous "
                                 "16-byte aligned GPU tensors")
    n = len(dsts)
    dst = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts])
    nbytes = (ctypes.c_long * n)(*[d.numel() * d.element_size() for d in dsts])
    _check(lib().dtf_gather_feeds(table.data_ptr(), n, dst, nbytes, _stream()), "gather_feeds")


RING_FEEDS, RING_WORDS = 8, 256      # step_ring.h: inputs and result words per ring slot
RING_CTR_WORDS = 2 + 2 * RING_FEEDS  # step_ring.h: two counters + the replay's input addresses


class RingArgs:
    """The device-visible parts of a step ring (``csrc/kernels/step_ring.h``): pinned address
    table ``[slots, 8]`` int64, device counters and address copies ``[2 + 16]`` int32, pinned results ``[slots, 256]``
    fp32. ``first_taken`` records, while a step is being captured, whether an earlier kernel of
    the step already publishes the replay number."""

    def __init__(self, table: torch.Tensor, ctr: torch.Tensor, out: torch.Tensor, slots: int):
        if slots < 1 or slots & (slots - 1):
            raise HipKernelError("step ring: the number of slots must be a power of two")
        if table.dtype != torch.int64 or not table.is_pinned() or table.numel() != slots * RING_FEEDS \
                or out.dtype != torch.float32 or not out.is_pinned() \
                or out.numel() != slots * RING_WORDS or ctr.dtype != torch.int32 \
                or not ctr.is_cuda or ctr.numel() != RING_CTR_WORDS:
            raise HipKernelError("step ring: bad buffers")
        self.table, self.ctr, self.out, self.slots = table, ctr, out, slots
        self.first_taken = False

    def take_first(self) -> int:
        first = not self.first_taken
        self.first_taken = True
        return int(first)


def pack_f32_ring(parts: Sequence[torch.Tensor], ring: RingArgs,
                  adds: Optional[Sequence[float]] = None) -> List[int]:
    """``pack_f32`` as the last ring kernel of a captured step: the words go to the replay's
    slot of ``ring.out``, the slot is stamped and the next replay opened. ``parts`` may be empty."""
    if len(parts) > 16:
        raise HipKernelError("pack_f32_ring: at most 16 parts")
    parts = [_req(p, torch.float32, "pack_f32 part") for p in parts]
    if sum(p.numel() for p in parts) > RING_WORDS - 1:
        raise HipKernelError("pack_f32_ring: more than %d words" % (RING_WORDS - 1))
    n = max(len(parts), 1)
    src = (ctypes.c_void_p * n)(*[p.data_ptr() for p in parts])
    sizes = (ctypes.c_int * n)(*[p.numel() for p in parts])
    add_arr = None
    if adds is not None:
        if len(adds) != len(parts):
            raise HipKernelError("pack_f32_ring: one addend per part")
        add_arr = (ctypes.c_float * n)(*[float(a) for a in adds])
    _check(lib().dtf_pack_f32_ring(len(parts), src, sizes, add_arr, ring.table.data_ptr(),
                                   ring.ctr.data_ptr(), ring.out.data_ptr(), ring.slots,
                                   ring.take_first(), _stream()), "pack_f32_ring")
    offsets, off = [], 0
    for p in parts:
        offsets.append(off)
        off += p.numel()
    return offsets


def gather_feeds_ring(ring: RingArgs, dsts: Sequence[torch.Tensor]) -> None:
    """``gather_feeds`` as the first kernel of a ring step: input i is copied from the address in
    entry i of the replay's slot of ``ring.table``."""
    dsts = list(dsts)
    if not 1 <= len(dsts) <= RING_FEEDS:
        raise HipKernelError("gather_feeds_ring: 1..8 destinations")
    for d in dsts:
        if not (isinstance(d, torch.Tensor) and d.device.type == "cuda" and d.is_contiguous()
                and d.numel() > 0 and d.data_ptr() % 16 == 0):
            raise HipKernelError("gather_feeds_ring: destinations must be non-empty contiguous "
                                 "16-byte aligned GPU tensors")
    if ring.first_taken:
        raise HipKernelError("gather_feeds_ring: must be the first ring kernel of the step")
    ring.first_taken = True
    n = len(dsts)
    dst = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts])
    nbytes = (ctypes.c_long * n)(*[d.numel() * d.element_size() for d in dsts])
    _check(lib().dtf_gather_feeds_ring(ring.table.data_ptr(), ring.ctr.data_ptr(),
                                       ring.out.data_ptr(), ring.slots, n, dst, nbytes, _stream()),
           "gather_feeds_ring")


class _MlpStep(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("labels", ctypes.c_void_p), ("w1", ctypes.c_void_p),
                ("b1", ctypes.c_void_p), ("w2", ctypes.c_void_p), ("b2", ctypes.c_void_p),
                ("B", ctypes.c_long), ("D", ctypes.c_long), ("H", ctypes.c_long),
                ("C", ctypes.c_long),
                ("lr_w1", ctypes.c_float), ("lr_b1", ctypes.c_float), ("lr_w2", ctypes.c_float),
                ("lr_b2", ctypes.c_float), ("dloss", ctypes.c_float),
                ("loss", ctypes.c_void_p), ("accuracy", ctypes.c_void_p), ("y", ctypes.c_void_p),
                ("a", ctypes.c_void_p), ("dz1", ctypes.c_void_p),
                ("head_ws", ctypes.c_void_p),
                ("ncounters", ctypes.c_int), ("counter", ctypes.c_void_p * 4),
                ("delta", ctypes.c_float * 4),
                ("npack", ctypes.c_int), ("pack_src", ctypes.c_void_p * 16),
                ("pack_size", ctypes.c_int * 16), ("pack_dst", ctypes.c_void_p),
                ("seq_src", ctypes.c_void_p), ("seq_dst", ctypes.c_void_p),
                ("ring_table", ctypes.c_void_p), ("ring_ctr", ctypes.c_void_p),
                ("ring_out", ctypes.c_void_p), ("ring", ctypes.c_int),
                ("x_index", ctypes.c_int), ("labels_index", ctypes.c_int),
                ("ring_first", ctypes.c_int), ("ring_last", ctypes.c_int),
                ("host_seq", ctypes.c_long),
                ("g_w1", ctypes.c_void_p), ("g_b1", ctypes.c_void_p), ("g_w2", ctypes.c_void_p),
                ("g_b2", ctypes.c_void_p), ("pack_add", ctypes.c_float * 16)]


def mlp_step_supported(b: int, d: int, h: int, c: int) -> bool:
    return bool(lib().dtf_mlp_step_supported(b, d, h, c))


def mlp_step_scratch(b: int, d: int, h: int, c: int, device) -> dict:
    """Buffers one two-layer training step needs besides the model: results (loss, accuracy,
    softmax) and scratch (hidden activations, dz1, the head's per-block shares). Allocate once and pass to every ``mlp_step_train`` call of that shape."""
    L = lib()
    ws_n = int(L.dtf_mlp_head_workspace(b, h, c))
    buf = torch.zeros(2 + b * c + 2 * b * h + ws_n, dtype=torch.float32, device=device)
    off = 0
    out = {}
    for name, n, shape in (("loss", 1, ()), ("accuracy", 1, ()), ("y", b * c, (b, c)),
                           ("a", b * h, (b, h)), ("dz1", b * h, (b, h)),
                           ("head_ws", ws_n, (ws_n,))):
        out[name] = buf[off:off + n].view(shape)
        off += n
    return out


def mlp_step_train(x: Optional[torch.Tensor], labels: Optional[torch.Tensor], w1: torch.Tensor,
                   b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, lrs: Sequence[float],
                   scratch: dict, dims: Sequence[int], dloss: float = 1.0,
                   want_accuracy: bool = True, want_softmax: bool = False,
                   counters: Sequence[torch.Tensor] = (), deltas: Sequence[float] = (),
                   pack: Sequence[torch.Tensor] = (), pack_out: Optional[torch.Tensor] = None,
                   signal=None, ring: Optional[RingArgs] = None, x_index: int = -1,
                   labels_index: int = -1, ring_last: bool = True,
                   grads: Optional[Sequence[torch.Tensor]] = None,
                   pack_adds: Optional[Sequence[float]] = None):
    """One SGD step of ``softmax(sigmoid(x w1 + b1) w2 + b2)`` under cross-entropy in three
    launches (``csrc/kernels/mlp_step.hip``): parameters are updated in place
    (``lrs = (lr_w1, lr_b1, lr_w2, lr_b2)``), ``counters[j] += deltas[j]``, loss / accuracy /
    softmax land in ``scratch``. ``dims = (B, D, H, C)``. ``pack``: small fp32 tensors gathered
    at the end of the step into ``pack_out`` (pinned host or GPU memory; ``signal = (seq, index)``
    echoes the host's step number behind them as in ``pack_f32``) or, with a ``ring``, into the
    replay's slot of the ring. With a ring, ``x_index`` / ``labels_index`` >= 0 name the table
    entries the batch is read through (``x`` / ``labels`` may then be ``None``), and
    ``ring_last`` says that this step's last launch is the last ring kernel of the captured step
    (it gathers ``pack`` into the ring, stamps the slot and opens the next replay).

    ``grads = (g_w1, g_b1, g_w2, g_b2)``: gradient mode -- the same three launches leave the four
    gradients in these tensors (shaped like the parameters; e.g. the slots of a flat gradient
    arena) and write neither parameters nor counters: the caller's collective and update launch
    follow. ``pack_adds[i]`` is added to every word of packed part i (a step counter fetched
    ahead of the launch that bumps it).

    Returns the argument block of the call: ``lib().dtf_mlp_step_train(ctypes.byref(block),
    stream)`` issues the same step again (every tensor named here must still be alive)."""
    b, d, h, c = (int(v) for v in dims)
    if not mlp_step_supported(b, d, h, c):
        raise HipKernelError("mlp_step_train: unsupported shape B=%d D=%d H=%d C=%d" % (b, d, h, c))

    def need(t, shape, what):
        t = _req(t, torch.float32, "mlp_step " + what)
        if tuple(t.shape) != tuple(shape) or not t.is_contiguous():
            raise HipKernelError("mlp_step_train: %s must be contiguous %s, got %s"
                                 % (what, tuple(shape), tuple(t.shape)))
        return t

    w1, b1 = need(w1, (d, h), "w1"), need(b1, (h,), "b1")
    w2, b2 = need(w2, (h, c), "w2"), need(b2, (c,), "b2")
    a = _MlpStep()
    a.host_seq = -1
    if ring is None or x_index < 0:
        a.x = need(x, (b, d), "x").data_ptr()
        x_index = -1
    if ring is None or labels_index < 0:
        a.labels = need(labels, (b, c), "labels").data_ptr()
        labels_index = -1
    a.w1, a.b1, a.w2, a.b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
    a.B, a.D, a.H, a.C = b, d, h, c
    a.lr_w1, a.lr_b1, a.lr_w2, a.lr_b2 = (float(v) for v in lrs)
    a.dloss = float(dloss)
    if tuple(scratch["a"].shape) != (b, h) or tuple(scratch["y"].shape) != (b, c) \
            or scratch["a"].device != w1.device:
        raise HipKernelError("mlp_step_train: scratch was made for another shape or device")
    a.loss = scratch["loss"].data_ptr()
    a.accuracy = scratch["accuracy"].data_ptr() if want_accuracy else None
    a.y = scratch["y"].data_ptr() if want_softmax else None
    a.a, a.dz1 = scratch["a"].data_ptr(), scratch["dz1"].data_ptr()
    a.head_ws = scratch["head_ws"].data_ptr()
    if grads is not None:
        if len(grads) != 4 or counters:
            raise HipKernelError("mlp_step_train: gradient mode takes four destinations and no "
                                 "counters")
        g_w1, g_b1 = need(grads[0], (d, h), "g_w1"), need(grads[1], (h,), "g_b1")
        g_w2, g_b2 = need(grads[2], (h, c), "g_w2"), need(grads[3], (c,), "g_b2")
        a.g_w1, a.g_b1 = g_w1.data_ptr(), g_b1.data_ptr()
        a.g_w2, a.g_b2 = g_w2.data_ptr(), g_b2.data_ptr()
    if pack_adds is not None and len(pack_adds) != len(pack):
        raise HipKernelError("mlp_step_train: one addend per packed part")
    if len(counters) != len(deltas) or len(counters) > 4:
        raise HipKernelError("mlp_step_train: at most 4 counters, one delta each")
    a.ncounters = len(counters)
    for j, (t, dl) in enumerate(zip(counters, deltas)):
        if t.dtype != torch.float32 or t.numel() != 1 or not t.is_cuda:
            raise HipKernelError("mlp_step_train: counters must be fp32 GPU scalars")
        a.counter[j] = t.data_ptr()
        a.delta[j] = float(dl)
    if len(pack) > 16:
        raise HipKernelError("mlp_step_train: at most 16 packed parts")
    total = 0
    a.npack = len(pack)
    for i, t in enumerate(pack):
        t = _req(t, torch.float32, "mlp_step packed part")
        a.pack_src[i] = t.data_ptr()
        a.pack_size[i] = t.numel()
        a.pack_add[i] = float(pack_adds[i]) if pack_adds is not None else 0.0
        total += t.numel()
    if total > RING_WORDS - 1:
        raise HipKernelError("mlp_step_train: more than %d packed words" % (RING_WORDS - 1))
    if ring is not None:
        a.ring_table, a.ring_ctr = ring.table.data_ptr(), ring.ctr.data_ptr()
        a.ring_out, a.ring = ring.out.data_ptr(), ring.slots
        a.x_index, a.labels_index = x_index, labels_index
        a.ring_first = ring.take_first()
        a.ring_last = int(bool(ring_last))
        if pack and not ring_last:
            raise HipKernelError("mlp_step_train: a ring gather needs ring_last")
    else:
        a.x_index = a.labels_index = -1
        if pack:
            if pack_out is None or pack_out.dtype != torch.float32 or pack_out.numel() < total \
                    or not (pack_out.is_cuda or pack_out.is_pinned()):
                raise HipKernelError("mlp_step_train: bad pack_out")
            a.pack_dst = pack_out.data_ptr()
            if signal is not None:
                seq, index = signal
                if seq.numel() != 1 or seq.dtype != torch.float32 or not total <= index < pack_out.numel():
                    raise HipKernelError("mlp_step_train: bad signal")
                a.seq_src, a.seq_dst = seq.data_ptr(), pack_out.data_ptr() + 4 * int(index)
    _check(lib().dtf_mlp_step_train(ctypes.byref(a), _stream()), "mlp_step_train")
    return a


def philox_window(rows: int, cols: int, seed: int, op_seed: int, row0: int, col0: int,
                  global_cols: int, dtype: torch.dtype = torch.float32, device=None,
                  normal: bool = True, a: float = 0.0, b: float = 1.0) -> torch.Tensor:
    """The ``[rows, cols]`` window at ``(row0, col0)`` of a global random matrix with
    ``global_cols`` columns whose element ``(i, j)`` is stream element ``i*global_cols + j``:
    a rank's shard of a global matrix, generated locally. Values are ``a + b * g`` with ``g``
    N(0,1) (``normal``) or U(0,1)."""
    kinds = {torch.float32: 0, torch.bfloat16: 1, torch.float8_e4m3fn: 2}
    if dtype not in kinds:
        raise HipKernelError("philox_window: dtype must be fp32, bf16 or fp8 e4m3")
    if row0 < 0 or col0 < 0 or col0 + cols > global_cols:
        raise HipKernelError("philox_window: window outside the global matrix")
    out = torch.empty((rows, cols), dtype=dtype, device=device)
    _check(lib().dtf_philox_fill_2d(out.data_ptr(), rows, cols, out.stride(0) if rows else cols,
                                    seed & 0xFFFFFFFF, op_seed & 0xFFFFFFFF, row0, col0,
                                    global_cols, float(a), float(b), int(normal),
                                    kinds[dtype], _stream()), "philox_window")
    return out


# ---------------------------------------------------------------------------------------
# data movement / bookkeeping (csrc/kernels/misc_ops.hip)
# ---------------------------------------------------------------------------------------
def copy_2d(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """``dst[...] = src`` for two rank-2 GPU views of the same shape and dtype whose rows are
    contiguous (any row pitch): one launch, no temporary."""
    for t, what in ((src, "src"), (dst, "dst")):
        if not (isinstance(t, torch.Tensor) and t.device.type == "cuda" and t.dim() == 2):
            raise HipKernelError("copy_2d: %s must be a rank-2 GPU tensor" % what)
        if t.shape[1] > 1 and t.stride(1) != 1:
            raise HipKernelError("copy_2d: %s rows must be contiguous" % what)
    if src.shape != dst.shape or src.dtype != dst.dtype:
        raise HipKernelError("copy_2d: shape / dtype mismatch")
    rows, cols = src.shape
    es = src.element_size()
    sp = (src.stride(0) if rows > 1 else cols) * es
    dp = (dst.stride(0) if rows > 1 else cols) * es
    _check(lib().dtf_copy_2d(src.data_ptr(), dst.data_ptr(), rows, cols * es, sp, dp, _stream()),
           "copy_2d")
    return dst


def multi_copy(srcs: Sequence[torch.Tensor], dsts: Sequence[torch.Tensor]) -> None:
    """``dsts[i][...] = srcs[i]`` for pairs of contiguous GPU tensors of equal size in bytes --
    one launch per 16 pairs."""
    if len(srcs) != len(dsts):
        raise HipKernelError("multi_copy: as many sources as destinations")
    for a, b in zip(srcs, dsts):
        if not (a.device.type == "cuda" and b.device.type == "cuda" and a.is_contiguous()
                and b.is_contiguous()
                and a.numel() * a.element_size() == b.numel() * b.element_size()):
            raise HipKernelError("multi_copy: pairs must be contiguous GPU tensors of equal bytes")
    c = ctypes
    for start in range(0, len(srcs), 16):
        ss, ds = srcs[start:start + 16], dsts[start:start + 16]
        n = len(ss)
        src = (c.c_void_p * n)(*[t.data_ptr() if t.numel() else None for t in ss])
        dst = (c.c_void_p * n)(*[t.data_ptr() if t.numel() else None for t in ds])
        nbytes = (c.c_long * n)(*[t.numel() * t.element_size() for t in ss])
        _check(lib().dtf_multi_copy(n, src, dst, nbytes, _stream()), "multi_copy")


def pack_into(parts: Sequence[torch.Tensor], out: torch.Tensor) -> List[torch.Tensor]:
    """Copy every tensor of ``parts`` (contiguous GPU tensors of ``out``'s dtype) one behind the
    other into the flat GPU tensor ``out`` -- one launch per 16 parts, however large they are --
    and return the views of ``out`` that now hold them, shaped like the parts."""
    if out.device.type != "cuda" or not out.is_contiguous() or out.dim() != 1:
        raise HipKernelError("pack_into: out must be a flat contiguous GPU tensor")
    if sum(int(p.numel()) for p in parts) > out.numel():
        raise HipKernelError("pack_into: %d elements do not fit into %d"
                             % (sum(int(p.numel()) for p in parts), out.numel()))
    views, off = [], 0
    for p in parts:
        if not (isinstance(p, torch.Tensor) and p.device.type == "cuda" and p.is_contiguous()
                and p.dtype == out.dtype):
            raise HipKernelError("pack_into: parts must be contiguous GPU tensors of out's dtype")
        views.append(out[off:off + p.numel()].view(p.shape))
        off += p.numel()
    if off > out.numel():
        raise HipKernelError("pack_into: %d elements do not fit into %d" % (off, out.numel()))
    es = out.element_size()
    c = ctypes
    for start in range(0, len(parts), 16):
        ps, vs = parts[start:start + 16], views[start:start + 16]
        n = len(ps)
        src = (c.c_void_p * n)(*[p.data_ptr() if p.numel() else None for p in ps])
        dst = (c.c_void_p * n)(*[v.data_ptr() if v.numel() else None for v in vs])
        nbytes = (c.c_long * n)(*[p.numel() * es for p in ps])
        _check(lib().dtf_multi_copy(n, src, dst, nbytes, _stream()), "pack_into")
    return views


def assign_add(var: torch.Tensor, delta: torch.Tensor, sign: float = 1.0) -> torch.Tensor:
    """``var += sign * delta`` in place (delta the same shape, a trailing suffix or a scalar)."""
    var_c = _req(var, torch.float32, "assign_add var")
    if var_c is not var:
        raise HipKernelError("assign_add: the variable must be contiguous")
    delta = _req(delta, torch.float32, "assign_add delta")
    dn = _bcast_numel(var.shape, delta, "assign_add delta") if delta.numel() != 1 else 1
    _check(lib().dtf_assign_add_f32(var.data_ptr(), delta.data_ptr(), var.numel(), dn,
                                    float(sign), _stream()), "assign_add")
    return var


def one_hot(indices: torch.Tensor, depth: int, on_value: float = 1.0,
            off_value: float = 0.0) -> torch.Tensor:
    """fp32 ``[..., depth]`` with ``on_value`` where the last axis equals the index."""
    if indices.dtype not in (torch.int32, torch.int64) or indices.device.type != "cuda":
        raise HipKernelError("one_hot: indices must be an int32 / int64 GPU tensor")
    indices = indices.contiguous()
    out = torch.empty(tuple(indices.shape) + (int(depth),), dtype=torch.float32,
                      device=indices.device)
    _check(lib().dtf_one_hot_f32(indices.data_ptr(), int(indices.dtype == torch.int64),
                                 out.data_ptr(), indices.numel(), int(depth), float(on_value),
                                 float(off_value), _stream()), "one_hot")
    return out


def mse(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``mean((a - b) ** 2)`` over all elements as a 0-d tensor: the difference and its square
    are never written to memory (two launches: partial sums, fold)."""
    a = _req(a, torch.float32, "mse a")
    b = _req(b, torch.float32, "mse b")
    if a.shape != b.shape or a.numel() == 0:
        raise HipKernelError("mse: operands must have the same non-empty shape")
    n = a.numel()
    buf = torch.empty(1 + int(lib().dtf_reduce_all_workspace(n)), dtype=torch.float32,
                      device=a.device)
    _check(lib().dtf_mse_f32(a.data_ptr(), b.data_ptr(), buf.data_ptr(), n, buf[1:].data_ptr(),
                             _stream()), "mse")
    return buf[0].view(())


def mse_grad(a: torch.Tensor, b: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """Gradient of ``mse(a, b)`` with respect to ``a``: ``(2 g / n) (a - b)``, ``g`` an fp32
    device scalar."""
    a = _req(a, torch.float32, "mse_grad a")
    b = _req(b, torch.float32, "mse_grad b")
    g = _req(g, torch.float32, "mse_grad g")
    if a.shape != b.shape or a.numel() == 0 or g.numel() != 1:
        raise HipKernelError("mse_grad: same-shape operands and a scalar gradient")
    out = torch.empty_like(a)
    _check(lib().dtf_mse_grad_f32(a.data_ptr(), b.data_ptr(), g.data_ptr(), out.data_ptr(),
                                  a.numel(), _stream()), "mse_grad")
    return out


def mse_step_supported(rows: int, cols: int, want_dbias: bool) -> bool:
    return bool(lib().dtf_mse_step_supported(int(rows), int(cols), int(bool(want_dbias))))


def mse_step(pred: torch.Tensor, labels: torch.Tensor, g=1.0, want_dbias: bool = False,
             out: Optional[Sequence[torch.Tensor]] = None, ring: Optional["RingArgs"] = None,
             pack: Optional[Sequence[torch.Tensor]] = None,
             pack_adds: Optional[Sequence[float]] = None, labels_index: int = -1):
    """``loss = mean((pred - labels) ** 2)``, ``dpred = (2 g / n) (pred - labels)`` and (rank-2
    operands, ``want_dbias``) ``dbias = dpred.sum(0)`` in ONE launch (``csrc/kernels/mse_step.hip``;
    bit-identical to ``mse`` + ``mse_grad`` + ``reduce_cols``). ``g``: a number or an fp32 device
    scalar. ``out = (loss, dpred[, dbias])``: destinations to reuse. With a ``ring`` and ``pack``
    (a list, possibly empty) the launch also gathers those small fp32 tensors (+ ``pack_adds``)
    into the replay's slot of the step ring, stamps it and opens the next replay; with a ``ring``
    and ``labels_index >= 0`` the labels are read where the replay's address table says they lie
    (``labels`` then only gives the shape).
    Returns ``(loss, dpred, dbias or None)``."""
    pred = _req(pred, torch.float32, "mse_step pred")
    labels = _req(labels, torch.float32, "mse_step labels")
    if pred.shape != labels.shape or pred.numel() == 0 or not pred.is_contiguous() \
            or not labels.is_contiguous():
        raise HipKernelError("mse_step: contiguous operands of the same non-empty shape")
    if want_dbias and pred.dim() != 2:
        raise HipKernelError("mse_step: the bias gradient needs rank-2 operands")
    rows, cols = (pred.shape if pred.dim() == 2 else (pred.numel(), 1))
    if not mse_step_supported(rows, cols, want_dbias):
        raise HipKernelError("mse_step: shape %s is outside the fused launch" % (tuple(pred.shape),))
    if out is not None:
        loss, dpred = out[0], out[1]
        dbias = out[2] if want_dbias else None
    else:
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(pred)
        dbias = torch.empty((cols,), dtype=torch.float32, device=pred.device) if want_dbias \
            else None
    g_dev, g_const = None, 0.0
    if isinstance(g, torch.Tensor):
        g_dev = _req(g, torch.float32, "mse_step g").data_ptr()
    else:
        g_const = float(g)
    c = ctypes
    npack, src, sizes, add_arr = -1, None, None, None
    ring_ptrs = (None, None, None, 0, 0)
    if ring is not None:
        parts = [_req(t, torch.float32, "mse_step packed part") for t in (pack or ())] \
            if pack is not None else None
        if parts is not None:
            if len(parts) > 16 or sum(t.numel() for t in parts) > RING_WORDS - 1:
                raise HipKernelError("mse_step: too many packed words")
            npack = len(parts)
            m = max(npack, 1)
            src = (c.c_void_p * m)(*[t.data_ptr() for t in parts])
            sizes = (c.c_int * m)(*[t.numel() for t in parts])
            if pack_adds is not None:
                if len(pack_adds) != npack:
                    raise HipKernelError("mse_step: one addend per packed part")
                add_arr = (c.c_float * m)(*[float(a) for a in pack_adds])
        if parts is not None or labels_index >= 0:
            ring_ptrs = (ring.table.data_ptr(), ring.ctr.data_ptr(), ring.out.data_ptr(),
                         ring.slots, ring.take_first())
    elif pack is not None or labels_index >= 0:
        raise HipKernelError("mse_step: a gather / a ring read needs the step ring")
    _check(lib().dtf_mse_step_f32(pred.data_ptr(), labels.data_ptr(), g_dev, g_const,
                                  loss.data_ptr(), dpred.data_ptr(), _ptr(dbias), rows, cols,
                                  npack, src, sizes, add_arr, *ring_ptrs,
                                  int(labels_index) if ring is not None else -1, _stream()),
           "mse_step")
    return loss, dpred, dbias


def sign(x: torch.Tensor) -> torch.Tensor:
    x = _req(x, torch.float32, "sign x")
    out = torch.empty_like(x)
    _check(lib().dtf_sign_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "sign")
    return out


def broadcast_mid(g: torch.Tensor, outer: int, mid: int, inner: int, scale: float,
                  shape: Sequence[int]) -> torch.Tensor:
    """``out[o, m, i] = g[o, i] * scale`` reshaped to ``shape`` (gradient of a reduction over the
    middle extent; bias-style broadcast when ``outer == 1``)."""
    g = _req(g, torch.float32, "broadcast_mid g")
    if g.numel() != outer * inner:
        raise HipKernelError("broadcast_mid: g has %d elements, expected %d"
                             % (g.numel(), outer * inner))
    out = torch.empty(tuple(shape), dtype=torch.float32, device=g.device)
    if out.numel() != outer * mid * inner:
        raise HipKernelError("broadcast_mid: shape does not match the extents")
    _check(lib().dtf_broadcast_mid_f32(g.data_ptr(), out.data_ptr(), outer, mid, inner,
                                       float(scale), _stream()), "broadcast_mid")
    return out


def log_softmax_grad(y: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    y = _req(y, torch.float32, "log_softmax_grad y")
    dy = _req(dy, torch.float32, "log_softmax_grad dy")
    if y.shape != dy.shape or y.dim() < 1:
        raise HipKernelError("log_softmax_grad: shapes differ")
    y2, dy2 = _as_rows(y), _as_rows(dy)
    dx = torch.empty_like(y2)
    _check(lib().dtf_log_softmax_grad_f32(y2.data_ptr(), dy2.data_ptr(), dx.data_ptr(),
                                          y2.shape[0], y2.shape[1], _stream()),
           "log_softmax_grad")
    return dx.view(y.shape)


def absmax_scale(x: torch.Tensor, target: float = FP8_MAX) -> torch.Tensor:
    """3-element GPU tensor ``[max|x|, target / max|x|, max|x| / target]`` (scale 1 for an
    all-zero tensor), computed on the device in a fixed order: the quantisation scale never
    visits the host."""
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise HipKernelError("absmax_scale: source must be fp32 or bf16")
    x = _req(x, x.dtype, "absmax_scale x")
    n = x.numel()
    ws = torch.empty((max(1, int(lib().dtf_absmax_workspace(max(n, 1)))),), dtype=torch.float32,
                     device=x.device)
    out = torch.empty((3,), dtype=torch.float32, device=x.device)
    _check(lib().dtf_absmax_scale(x.data_ptr(), int(x.dtype == torch.bfloat16), n, float(target),
                                  ws.data_ptr(), out.data_ptr(), _stream()), "absmax_scale")
    return out


def quantize_fp8_device(x: torch.Tensor):
    """Per-tensor fp8 quantisation with the scale found and applied on the device: returns
    ``(q, stats)`` with ``stats = [max|x|, scale, 1/scale]`` in GPU memory. Two launches, no
    synchronisation (``quantize_fp8`` returns the inverse scale as a Python float and therefore
    reads it back)."""
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise HipKernelError("quantize_fp8: source must be fp32 or bf16")
    x = _aligned(_req(x, x.dtype, "quantize_fp8 x"))
    stats = absmax_scale(x)
    out = torch.empty(x.shape, dtype=FP8, device=x.device)
    _check(lib().dtf_cast_to_fp8_dev_scale(x.data_ptr(), int(x.dtype == torch.bfloat16),
                                           out.data_ptr(), x.numel(), stats[1:].data_ptr(),
                                           _stream()), "quantize_fp8")
    return out, stats


def truncated_normal(shape: Sequence[int], seed: int, op_seed: int, mean: float = 0.0,
                     stddev: float = 1.0, device=None) -> torch.Tensor:
    """N(mean, stddev) restricted to two standard deviations, element e drawn by inverse CDF from
    uniform e of the Philox stream keyed by (seed, op_seed)."""
    out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    _check(lib().dtf_truncated_normal_f32(out.data_ptr(), out.numel(), seed & 0xFFFFFFFF,
                                          op_seed & 0xFFFFFFFF, float(mean), float(stddev),
                                          _stream()), "truncated_normal")
    return out


# ---------------------------------------------------------------------------------------
# device-side completion flags of the peer-memory collectives (csrc/kernels/peer_sync.hip)
# ---------------------------------------------------------------------------------------
def flags_alloc(nbytes: int):
    """This rank's flag page: ``nbytes`` of zeroed fine-grained device memory on the current
    device. Returns ``(address, ipc_handle_bytes)``."""
    ptr = ctypes.c_void_p()
    handle = ctypes.create_string_buffer(64)
    code = lib().dtf_flags_alloc(int(nbytes), ctypes.byref(ptr), handle)
    if code != 0:
        raise HipKernelError("flags_alloc: code %d" % code)
    return ptr.value, handle.raw


def flags_open(handle: bytes) -> int:
    """Map a peer's flag page from the 64-byte handle its ``flags_alloc`` returned."""
    if len(handle) != 64:
        raise HipKernelError("flags_open: a handle is 64 bytes")
    ptr = ctypes.c_void_p()
    code = lib().dtf_flags_open(ctypes.create_string_buffer(handle, 64), ctypes.byref(ptr))
    if code != 0:
        raise HipKernelError("flags_open: code %d" % code)
    return ptr.value


def flags_close(address: int, owned: bool) -> None:
    lib().dtf_flags_close(address, int(bool(owned)))


def peer_signal(flag_addresses: Sequence[int], value: int) -> None:
    """On the current stream: store ``value`` into each flag (system-scope release behind
    everything the stream did before)."""
    arr = (ctypes.c_void_p * len(flag_addresses))(*[int(a) for a in flag_addresses])
    _check(lib().dtf_peer_signal(arr, len(flag_addresses), int(value), _stream()), "peer_signal")


def peer_wait(flag_addresses: Sequence[int], value: int, timeout_s: float,
              status: torch.Tensor) -> None:
    """On the current stream: wait until every flag has reached ``value``; a flag still behind
    after ``timeout_s`` puts its position + 1 into the pinned int32 ``status``."""
    if status.dtype != torch.int32 or not status.is_pinned():
        raise HipKernelError("peer_wait: status must be a pinned int32 tensor")
    arr = (ctypes.c_void_p * len(flag_addresses))(*[int(a) for a in flag_addresses])
    _check(lib().dtf_peer_wait(arr, len(flag_addresses), int(value), int(timeout_s * 1e6),
                               status.data_ptr(), _stream()), "peer_wait")
