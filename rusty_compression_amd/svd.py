"""Mirror of the reference's `svd` and `compute_svd` modules (src/svd.rs, src/compute_svd.rs)."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import torch

from . import _lib
from .types import CompressionType, as_device, empty


def _ctx():
    return _lib.default_context()


def _sptr(s: torch.Tensor):
    return ctypes.c_void_p(s.data_ptr())


def compute_svd(arr):
    """`ComputeSVD::compute_svd` (src/compute_svd.rs:18-27): thin SVD. Returns (u, s, vt)."""
    a = as_device(arr)
    m, n = a.shape
    r = min(m, n)
    u, vt = empty(m, r, a), empty(r, n, a)
    s = torch.empty(r, dtype=_lib.real_dtype(a.dtype), device=a.device)
    _ctx().call(f"rc_compute_svd_{_lib.suffix(a.dtype)}", _lib.mat(a), _lib.mat(u), _sptr(s), _lib.mat(vt))
    return u, s, vt


@dataclass
class SVD:
    """`struct SVD` (src/svd.rs:13-20)."""

    u: torch.Tensor
    s: torch.Tensor
    vt: torch.Tensor

    # -- SVDTraits (src/svd.rs:23-122) ----------------------------------------
    def nrows(self) -> int:
        return self.u.shape[0]

    def ncols(self) -> int:
        return self.vt.shape[1]

    def rank(self) -> int:
        return self.u.shape[1]

    def get_u(self):
        return self.u

    def get_s(self):
        return self.s

    def get_vt(self):
        return self.vt

    @staticmethod
    def compute_from(arr) -> "SVD":
        """src/svd.rs:165-169"""
        return SVD(*compute_svd(arr))

    def to_mat(self) -> torch.Tensor:
        """src/svd.rs:42-54"""
        out = empty(self.nrows(), self.ncols(), self.u)
        s = self.s.to(_lib.real_dtype(self.u.dtype)).contiguous()
        _ctx().call(f"rc_svd_to_mat_{_lib.suffix(self.u.dtype)}", _lib.mat(self.u), _sptr(s), _lib.mat(self.vt), _lib.mat(out))
        return out

    def compress_svd_rank(self, max_rank: int) -> "SVD":
        """src/svd.rs:68-84"""
        max_rank = min(int(max_rank), self.s.shape[0])
        return SVD(self.u[:, :max_rank].contiguous(), self.s[:max_rank].clone(), self.vt[:max_rank, :].contiguous())

    def compress_svd_tolerance(self, tol: float) -> "SVD":
        """src/svd.rs:87-101"""
        assert (tol < 1.0) and (0.0 <= tol), "Require 0 <= tol < 1.0"
        rank = ctypes.c_int64(-1)
        s = self.s.contiguous()
        _ctx().call(f"rc_svd_rank_by_tolerance_{_lib.suffix(s.dtype)}", _sptr(s), ctypes.c_int64(s.numel()), ctypes.c_double(tol), ctypes.byref(rank))
        return self.compress_svd_rank(int(rank.value))

    def compress(self, compression_type: CompressionType) -> "SVD":
        """src/svd.rs:60-65"""
        if compression_type.kind == "ADAPTIVE":
            return self.compress_svd_tolerance(compression_type.value)
        return self.compress_svd_rank(int(compression_type.value))

    def to_qr(self):
        """src/svd.rs:150-163 (consumes `self` in the reference)."""
        from .qr import QR

        r_, n = self.vt.shape
        k = min(r_, n)
        q, r = empty(self.nrows(), k, self.u), empty(k, n, self.u)
        ind = torch.empty(n, dtype=torch.int64, device=self.u.device)
        s = self.s.to(_lib.real_dtype(self.u.dtype)).contiguous()
        _ctx().call(f"rc_svd_to_qr_{_lib.suffix(self.u.dtype)}", _lib.mat(self.u), _sptr(s), _lib.mat(self.vt), _lib.mat(q), _lib.mat(r), _lib.i64p(ind))
        return QR(q, r, ind)

    @staticmethod
    def compute_from_range_estimate(range_, op) -> "SVD":
        """src/svd.rs:171-183 (`op`: a dense matrix or an operator with conj_matmat, operator.py)"""
        from .operator import OperatorTable, is_operator

        if is_operator(op):
            tab = OperatorTable(op)
            rg = as_device(range_, tab.dtype)
            m, n = op.nrows(), op.ncols()
            r = min(rg.shape[1], n)
            u, vt = empty(m, r, rg), empty(r, n, rg)
            s = torch.empty(r, dtype=_lib.real_dtype(tab.dtype), device=rg.device)
            tab.call(_ctx(), f"rc_svd_from_range_estimate_op_{_lib.suffix(tab.dtype)}", _lib.mat(rg), tab.byref(), _lib.mat(u), _sptr(s), _lib.mat(vt))
            return SVD(u, s, vt)
        a = as_device(op)
        rg = as_device(range_, a.dtype)
        m, n = a.shape
        r = min(rg.shape[1], n)
        u, vt = empty(m, r, a), empty(r, n, a)
        s = torch.empty(r, dtype=_lib.real_dtype(a.dtype), device=a.device)
        _ctx().call(f"rc_svd_from_range_estimate_{_lib.suffix(a.dtype)}", _lib.mat(rg), _lib.mat(a), _lib.mat(u), _sptr(s), _lib.mat(vt))
        return SVD(u, s, vt)
