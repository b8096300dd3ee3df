"""Operators behind the reference's `MatVec` / `ConjMatVec` / `MatMat` / `ConjMatMat` traits (src/types.rs:40-101).

The reference's range finders and `compute_from_range_estimate` are implemented for ANY operator
(`impl<Op: MatMat<A = $scalar>> SampleRange for Op`, src/random_sampling.rs:102, :130, :222; src/qr.rs:311-323,
src/svd.rs:171-183), not only for dense arrays.  Here an operator is any object with

    nrows(), ncols(), dtype                      MatVec::nrows / ncols (src/types.rs:44-48), the scalar type `A`
    matmat(x) -> A x         (ncols x s -> nrows x s device tensor)        MatMat::matmat          src/types.rs:58-71
    conj_matmat(x) -> A^H x  (nrows x s -> ncols x s device tensor)        ConjMatMat::conj_matmat src/types.rs:88-101

(`matmat_into(x, y)` / `conj_matmat_into(x, y)`, if present, write the product straight into the library's buffer).
The products run on the GPU -- through this library's own calls or the host's torch code on the current stream -- and the
library calls them back through the `rc_operator` table of the C ABI (include/rusty_compression_amd.h): the samplers, the
pivoted QR, the SVD and everything else stay inside the HIP library.  No CPU path.
"""
from __future__ import annotations

import ctypes
import traceback
from typing import Optional

import torch

from . import _lib

_PRODUCT_FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, _lib.rc_matrix, _lib.rc_matrix)


class rc_operator(ctypes.Structure):
    _fields_ = [
        ("rows", ctypes.c_int64),
        ("cols", ctypes.c_int64),
        ("matmat", _PRODUCT_FN),
        ("conj_matmat", _PRODUCT_FN),
        ("user", ctypes.c_void_p),
    ]


_TYPESTR = {torch.float64: "<f8", torch.float32: "<f4", torch.complex128: "<c16", torch.complex64: "<c8"}


class _DeviceView:
    """A strided device view handed over by the library (rc_matrix) as a zero-copy torch tensor (CUDA array interface)."""

    def __init__(self, m: _lib.rc_matrix, dtype: torch.dtype):
        es = torch.empty(0, dtype=dtype).element_size()
        self.__cuda_array_interface__ = {
            "shape": (int(m.rows), int(m.cols)),
            "typestr": _TYPESTR[dtype],
            "data": (int(m.data or 0), False),
            "strides": (int(m.row_stride) * es, int(m.col_stride) * es),
            "version": 3,
        }


def view_of(m: _lib.rc_matrix, dtype: torch.dtype) -> torch.Tensor:
    if m.rows == 0 or m.cols == 0 or not m.data:
        return torch.empty((int(m.rows), int(m.cols)), dtype=dtype, device="cuda")
    return torch.as_tensor(_DeviceView(m, dtype), device="cuda")


class Operator:
    """Base class / protocol of an operator (see the module text).  Subclasses implement the two products."""

    dtype: torch.dtype = torch.float64

    def nrows(self) -> int:
        raise NotImplementedError

    def ncols(self) -> int:
        raise NotImplementedError

    def matmat(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def conj_matmat(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    @property
    def shape(self):
        return (self.nrows(), self.ncols())


class DenseOperator(Operator):
    """A dense device matrix BEHIND the callback table: each product is the library's own rc_matmat / rc_conj_matmat on the views
    the library hands over, so the *_op_* entry points reproduce their dense twins bit for bit in f64 (tests/test_gpu_parity.py)."""

    def __init__(self, a):
        from .types import as_device

        self.a = as_device(a)
        self.dtype = self.a.dtype
        self.calls = {"matmat": 0, "conj_matmat": 0}

    def nrows(self):
        return self.a.shape[0]

    def ncols(self):
        return self.a.shape[1]

    def _raw(self, name, x: _lib.rc_matrix, y: _lib.rc_matrix, ctx_h):
        return getattr(_lib.lib(), f"{name}_{_lib.suffix(self.dtype)}")(ctypes.c_void_p(ctx_h), _lib.mat(self.a), x, y)

    def matmat_raw(self, ctx_h, x, y):
        self.calls["matmat"] += 1
        return self._raw("rc_matmat", x, y, ctx_h)

    def conj_matmat_raw(self, ctx_h, x, y):
        self.calls["conj_matmat"] += 1
        return self._raw("rc_conj_matmat", x, y, ctx_h)

    def matmat(self, x):
        from .types import matmat

        return matmat(self.a, x)

    def conj_matmat(self, x):
        from .types import conj_matmat

        return conj_matmat(self.a, x)


class LowRankOperator(Operator):
    """A = U V^H given by its factors and never formed: A x = U (V^H x), A^H x = V (U^H x), two skinny GEMMs of the library each."""

    def __init__(self, u, v):
        from .types import as_device

        self.u = as_device(u)
        self.v = as_device(v, self.u.dtype)
        assert self.u.shape[1] == self.v.shape[1], "U (m x r) and V (n x r) must share the inner extent"
        self.dtype = self.u.dtype

    def nrows(self):
        return self.u.shape[0]

    def ncols(self):
        return self.v.shape[0]

    def matmat(self, x):
        from .types import conj_matmat, dot

        return dot(self.u, conj_matmat(self.v, x))      # U (V^H x)

    def conj_matmat(self, x):
        from .types import conj_matmat, dot

        return dot(self.v, conj_matmat(self.u, x))      # V (U^H x)


def is_operator(op) -> bool:
    return not isinstance(op, torch.Tensor) and all(hasattr(op, f) for f in ("nrows", "ncols", "matmat"))


class OperatorTable:
    """The rc_operator of a Python operator: keeps the ctypes callbacks alive and carries a callback's exception back to the caller."""

    def __init__(self, op):
        self.op = op
        self.dtype = getattr(op, "dtype", torch.float64)
        self.error: Optional[BaseException] = None

        def product(name):
            raw = getattr(op, name + "_raw", None)
            into = getattr(op, name + "_into", None)
            plain = getattr(op, name, None)
            if raw is None and into is None and plain is None:
                return _PRODUCT_FN()  # NULL: e.g. an operator that is only MatMat

            def fn(_user, ctx_h, x, y):
                try:
                    if raw is not None:
                        return int(raw(ctx_h, x, y))
                    xv, yv = view_of(x, self.dtype), view_of(y, self.dtype)
                    if into is not None:
                        into(xv, yv)
                    else:
                        res = plain(xv)
                        assert tuple(res.shape) == tuple(yv.shape), f"{name} returned {tuple(res.shape)}, expected {tuple(yv.shape)}"
                        yv.copy_(res)
                    return _lib.RC_OK
                except _lib.RustyCompressionError as e:   # the library's own errors keep their status
                    self.error = e
                    return next((code for code, exc in _lib._STATUS_EXC.items() if type(e) is exc), _lib.RC_RUNTIME_ERROR)
                except BaseException as e:  # a callback must not unwind into C
                    self.error = e
                    traceback.print_exc()
                    return _lib.RC_RUNTIME_ERROR

            return _PRODUCT_FN(fn)

        self._cb = (product("matmat"), product("conj_matmat"))
        self.table = rc_operator(int(op.nrows()), int(op.ncols()), self._cb[0], self._cb[1], None)

    def byref(self):
        return ctypes.byref(self.table)

    def call(self, ctx, name: str, *args):
        """ctx.call with the callback's own exception re-raised (chained) when it was a callback that failed."""
        self.error = None
        try:
            ctx.call(name, *args)
        except _lib.RustyCompressionError as outer:
            if self.error is not None and not isinstance(self.error, _lib.RustyCompressionError):
                raise self.error from outer
            raise
