"""Host-side mirror of the reference's `types` module and `CompressionType`.

reference: src/types.rs (Apply :25-29, MatVec/MatMat/ConjMatMat :40-101, RelDiff
:162-196, error enum :11-21) and src/lib.rs:82-87 (CompressionType).  All
arithmetic happens in the HIP library behind the C ABI.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import (CompressionError, HipRuntimeError, LayoutError, LinalgError, PivotedQRError,  # noqa: F401
                   RustyCompressionError)


@dataclass(frozen=True)
class CompressionType:
    """`enum CompressionType { ADAPTIVE(f64), RANK(usize) }` (src/lib.rs:82-87)."""

    kind: str
    value: float

    @staticmethod
    def ADAPTIVE(tol: float) -> "CompressionType":
        return CompressionType("ADAPTIVE", float(tol))

    @staticmethod
    def RANK(rank: int) -> "CompressionType":
        return CompressionType("RANK", int(rank))


def as_device(x, dtype=None) -> torch.Tensor:
    """Borrow `x` (numpy array, CPU or CUDA tensor) as a CUDA tensor; no copy if it already is one."""
    if isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise HipRuntimeError("no HIP device: the engine has no CPU path")
        t = t.cuda()
    return t


def as_index(x) -> torch.Tensor:
    t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x).astype(np.int64))
    t = t.to(torch.int64)
    if not t.is_cuda:
        t = t.cuda()
    return t.contiguous()


def empty(rows: int, cols: int, like: torch.Tensor) -> torch.Tensor:
    """Owned C-order (row-major) output array, like the reference's `Array2::zeros`."""
    return torch.empty((rows, cols), dtype=like.dtype, device=like.device)


def matmat(op, x) -> torch.Tensor:
    """`MatMat::matmat` for dense matrices (src/types.rs:58-71): one GEMM, not a gemv loop."""
    a = as_device(op)
    x = as_device(x, a.dtype)
    assert a.shape[1] == x.shape[0]
    y = empty(a.shape[0], x.shape[1], a)
    _lib.default_context().call(f"rc_matmat_{_lib.suffix(a.dtype)}", _lib.mat(a), _lib.mat(x), _lib.mat(y))
    return y


def conj_matmat(op, x) -> torch.Tensor:
    """`ConjMatMat::conj_matmat` (src/types.rs:88-101): A^H X."""
    a = as_device(op)
    x = as_device(x, a.dtype)
    assert a.shape[0] == x.shape[0]
    y = empty(a.shape[1], x.shape[1], a)
    _lib.default_context().call(f"rc_conj_matmat_{_lib.suffix(a.dtype)}", _lib.mat(a), _lib.mat(x), _lib.mat(y))
    return y


def dot(a, b) -> torch.Tensor:
    """ndarray `.dot` of two matrices (or matrix . vector) through rc_gemm."""
    a = as_device(a)
    b = as_device(b, a.dtype)
    vec = b.dim() == 1
    b2 = b.unsqueeze(1) if vec else b
    assert a.shape[1] == b2.shape[0], "shape mismatch in dot"
    out = empty(a.shape[0], b2.shape[1], a)
    one, zero = _lib.scalar_arg(a.dtype, 1.0), _lib.scalar_arg(a.dtype, 0.0)
    _lib.default_context().call(f"rc_gemm_{_lib.suffix(a.dtype)}", ctypes.c_int32(0), ctypes.c_int32(0), one, _lib.mat(a), _lib.mat(b2), zero, _lib.mat(out))
    return out[:, 0] if vec else out


def rel_diff_fro(first, second) -> float:
    """`RelDiff::rel_diff_fro` (src/types.rs:182-188)."""
    b = as_device(second)
    a = as_device(first, b.dtype)
    out = _lib.real_out(b.dtype)
    _lib.default_context().call(f"rc_rel_diff_fro_{_lib.suffix(b.dtype)}", _lib.mat(a), _lib.mat(b), ctypes.byref(out))
    return float(out.value)


def rel_diff_l2(first, second) -> float:
    """`RelDiff::rel_diff_l2` (src/types.rs:190-196): vectors as n x 1 matrices."""
    b = as_device(second)
    a = as_device(first, b.dtype)
    return rel_diff_fro(a.reshape(-1, 1), b.reshape(-1, 1))
