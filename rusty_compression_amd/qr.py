"""Mirror of the reference's `qr` and `pivoted_qr` modules (src/qr.rs, src/pivoted_qr.rs).

`QR` / `LQ` own device-resident factors (torch CUDA tensors, C order like the
reference's `Array2`); every method is one call into the C ABI."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import torch

from . import _lib
from .types import CompressionType, as_device, as_index, empty


def _ctx():
    return _lib.default_context()


def pivoted_qr(arr, rank=None):
    """`PivotedQR::pivoted_qr` (src/pivoted_qr.rs:25-31, :81-183). Returns (q, r, ind).

    rank=None is the full factorization (k = min(m, n)); an integer stops after
    that many Householder steps (truncated mode, see rc_pivoted_qr)."""
    a = as_device(arr)
    assert a.dim() == 2
    m, n = a.shape
    k = min(m, n) if rank is None else min(int(rank), m, n)
    q, r = empty(m, k, a), empty(k, n, a)
    ind = torch.empty(n, dtype=torch.int64, device=a.device)
    _ctx().call(f"rc_pivoted_qr_{_lib.suffix(a.dtype)}", _lib.mat(a), _lib.mat(q), _lib.mat(r), _lib.i64p(ind))
    return q, r, ind


def pivoted_lq(arr, rank=None):
    """`PivotedQR::pivoted_lq` (src/pivoted_qr.rs:32-41). Returns (l, q, ind)."""
    a = as_device(arr)
    m, n = a.shape
    k = min(m, n) if rank is None else min(int(rank), m, n)
    l, q = empty(m, k, a), empty(k, n, a)
    ind = torch.empty(m, dtype=torch.int64, device=a.device)
    _ctx().call(f"rc_pivoted_lq_{_lib.suffix(a.dtype)}", _lib.mat(a), _lib.mat(l), _lib.mat(q), _lib.i64p(ind))
    return l, q, ind


def _rank_by_tolerance(tri: torch.Tensor, tol: float) -> int:
    assert (tol < 1.0) and (0.0 <= tol), "Require 0 <= tol < 1.0"
    rank = ctypes.c_int64(-1)
    _ctx().call(f"rc_rank_by_tolerance_{_lib.suffix(tri.dtype)}", _lib.mat(tri), ctypes.c_double(tol), ctypes.byref(rank))
    return int(rank.value)


@dataclass
class QR:
    """`struct QR` (src/qr.rs:31-40): A P = Q R, ind[j] = column of A at position j."""

    q: torch.Tensor
    r: torch.Tensor
    ind: torch.Tensor

    # -- QRTraits (src/qr.rs:141-238) -----------------------------------------
    def nrows(self) -> int:
        return self.q.shape[0]

    def ncols(self) -> int:
        return self.r.shape[1]

    def rank(self) -> int:
        return self.q.shape[1]

    def get_q(self):
        return self.q

    def get_r(self):
        return self.r

    def get_ind(self):
        return self.ind

    @staticmethod
    def compute_from(arr) -> "QR":
        """src/qr.rs:251-253"""
        return QR(*pivoted_qr(arr))

    def to_mat(self) -> torch.Tensor:
        """src/qr.rs:160-166"""
        out = empty(self.nrows(), self.ncols(), self.q)
        _ctx().call(f"rc_qr_to_mat_{_lib.suffix(self.q.dtype)}", _lib.mat(self.q), _lib.mat(self.r), _lib.i64p(self.ind), _lib.mat(out))
        return out

    def compress_qr_rank(self, max_rank: int) -> "QR":
        """src/qr.rs:169-184 (owned copies; `ind` stays full length)."""
        max_rank = min(int(max_rank), self.q.shape[1])
        return QR(self.q[:, :max_rank].contiguous(), self.r[:max_rank, :].contiguous(), self.ind.clone())

    def compress_qr_tolerance(self, tol: float) -> "QR":
        """src/qr.rs:187-200"""
        return self.compress_qr_rank(_rank_by_tolerance(self.r, tol))

    def compress(self, compression_type: CompressionType) -> "QR":
        """src/qr.rs:203-208"""
        if compression_type.kind == "ADAPTIVE":
            return self.compress_qr_tolerance(compression_type.value)
        return self.compress_qr_rank(int(compression_type.value))

    def column_id(self):
        """src/qr.rs:270-309"""
        from .col_interp_decomp import ColumnID

        c = empty(self.nrows(), self.rank(), self.q)
        z = empty(self.rank(), self.ncols(), self.q)
        _ctx().call(f"rc_qr_column_id_{_lib.suffix(self.q.dtype)}", _lib.mat(self.q), _lib.mat(self.r), _lib.i64p(self.ind), _lib.mat(c), _lib.mat(z))
        return ColumnID(c, z, self.ind.clone())

    @staticmethod
    def compute_from_range_estimate(range_, op) -> "QR":
        """src/qr.rs:311-323 (`op`: a dense matrix or an operator with conj_matmat, operator.py)"""
        from .operator import OperatorTable, is_operator

        if is_operator(op):
            tab = OperatorTable(op)
            rg = as_device(range_, tab.dtype)
            m, n = op.nrows(), op.ncols()
            k = min(rg.shape[1], n)
            q, r = empty(m, k, rg), empty(k, n, rg)
            ind = torch.empty(n, dtype=torch.int64, device=rg.device)
            tab.call(_ctx(), f"rc_qr_from_range_estimate_op_{_lib.suffix(tab.dtype)}", _lib.mat(rg), tab.byref(), _lib.mat(q), _lib.mat(r), _lib.i64p(ind))
            return QR(q, r, ind)
        a = as_device(op)
        rg = as_device(range_, a.dtype)
        m, n = a.shape
        k = min(rg.shape[1], n)
        q, r = empty(m, k, a), empty(k, n, a)
        ind = torch.empty(n, dtype=torch.int64, device=a.device)
        _ctx().call(f"rc_qr_from_range_estimate_{_lib.suffix(a.dtype)}", _lib.mat(rg), _lib.mat(a), _lib.mat(q), _lib.mat(r), _lib.i64p(ind))
        return QR(q, r, ind)


@dataclass
class LQ:
    """`struct LQ` (src/qr.rs:42-51): P A = L Q, ind[j] = row of A at position j."""

    l: torch.Tensor
    q: torch.Tensor
    ind: torch.Tensor

    # -- LQTraits (src/qr.rs:54-139) ------------------------------------------
    def nrows(self) -> int:
        return self.l.shape[0]

    def ncols(self) -> int:
        return self.q.shape[1]

    def rank(self) -> int:
        return self.q.shape[0]

    def get_q(self):
        return self.q

    def get_l(self):
        return self.l

    def get_ind(self):
        return self.ind

    @staticmethod
    def compute_from(arr) -> "LQ":
        """src/qr.rs:354-362"""
        return LQ(*pivoted_lq(arr))

    def to_mat(self) -> torch.Tensor:
        """src/qr.rs:73-77"""
        out = empty(self.nrows(), self.ncols(), self.q)
        _ctx().call(f"rc_lq_to_mat_{_lib.suffix(self.q.dtype)}", _lib.mat(self.l), _lib.mat(self.q), _lib.i64p(self.ind), _lib.mat(out))
        return out

    def compress_lq_rank(self, max_rank: int) -> "LQ":
        """src/qr.rs:80-96"""
        max_rank = min(int(max_rank), self.q.shape[0])
        return LQ(self.l[:, :max_rank].contiguous(), self.q[:max_rank, :].contiguous(), self.ind.clone())

    def compress_lq_tolerance(self, tol: float) -> "LQ":
        """src/qr.rs:99-112"""
        return self.compress_lq_rank(_rank_by_tolerance(self.l, tol))

    def compress(self, compression_type: CompressionType) -> "LQ":
        """src/qr.rs:114-119"""
        if compression_type.kind == "ADAPTIVE":
            return self.compress_lq_tolerance(compression_type.value)
        return self.compress_lq_rank(int(compression_type.value))

    def row_id(self):
        """src/qr.rs:363-403"""
        from .row_interp_decomp import RowID

        x = empty(self.nrows(), self.rank(), self.q)
        r = empty(self.rank(), self.ncols(), self.q)
        _ctx().call(f"rc_lq_row_id_{_lib.suffix(self.q.dtype)}", _lib.mat(self.l), _lib.mat(self.q), _lib.i64p(self.ind), _lib.mat(x), _lib.mat(r))
        return RowID(x, r, self.ind.clone())
