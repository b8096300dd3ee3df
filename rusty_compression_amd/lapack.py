"""The three LAPACK-granularity entry points: what the reference's own code calls per factorization --
`?geqp3` (`$qrf` at src/pivoted_qr.rs:139-172), `?orgqr` / `?ungqr` (`lax::Lapack::q`, src/pivoted_qr.rs:104-108) and the upper
triangular solve of the IDs (`solve_triangular`, src/qr.rs:298, :392) -- for a maintainer who swaps only those calls."""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .types import as_device, empty


def _ptr(t: torch.Tensor):
    return ctypes.c_void_p(t.data_ptr() if t.numel() else 0)


def geqp3(arr, kmax=None):
    """`?geqp3` in LAPACK's output format.  Returns (a, jpvt, tau): a = the factorization of A P with its columns in pivoted
    order (R on / above the diagonal, Householder vectors below it), jpvt 0-based, tau of length kmax (default min(m, n))."""
    a = as_device(arr).clone()
    m, n = a.shape
    k = min(m, n) if kmax is None else int(kmax)
    jpvt = torch.empty(n, dtype=torch.int64, device=a.device)
    tau = torch.empty(max(k, 0), dtype=a.dtype, device=a.device)
    _lib.default_context().call(f"rc_geqp3_{_lib.suffix(a.dtype)}", _lib.mat(a), ctypes.c_int64(k), _lib.i64p(jpvt), _ptr(tau))
    return a, jpvt, tau


def orgqr(a, tau, k=None):
    """`?orgqr` / `?ungqr`: Q (m x k) = H_0 ... H_{k-1} [I; 0] from the first k columns of a `geqp3` result."""
    a = as_device(a)
    tau = as_device(tau, a.dtype)
    k = int(tau.numel()) if k is None else int(k)
    q = empty(a.shape[0], max(k, 0), a)
    _lib.default_context().call(f"rc_orgqr_{_lib.suffix(a.dtype)}", _lib.mat(a), _ptr(tau), ctypes.c_int64(k), _lib.mat(q))
    return q


def trsm_upper(t, b):
    """Solves T X = B for X (T upper triangular k x k); returns X, `b` is not modified."""
    t = as_device(t)
    x = as_device(b, t.dtype).clone()
    _lib.default_context().call(f"rc_trsm_upper_{_lib.suffix(t.dtype)}", _lib.mat(t), _lib.mat(x))
    return x
