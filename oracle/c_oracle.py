"""ctypes loader for the C restatement of the oracle (oracle/rc_oracle.c).

CPU ORACLE -- test infrastructure only.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this.  The C file restates the
LAPACK routines the reference calls (file:line citations in rc_oracle_impl.h)
without linking LAPACK, so it is an independent check of the algorithm the HIP
kernels implement; tests/test_oracle_cpu.py pins it against SciPy's LAPACK
(oracle/ref_lapack.py) and the reference's permutation known answers.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librc_oracle.so")


def build(force: bool = False) -> str:
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(
        os.path.getmtime(os.path.join(_HERE, f)) for f in ("rc_oracle.c", "rc_oracle_impl.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _suf(dtype):
    return {np.dtype(np.float64): "d", np.dtype(np.float32): "s"}[np.dtype(dtype)]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_i64 = ctypes.c_int64


def geqp3(a: np.ndarray, kmax=None, pivot=True):
    """Householder QRCP with ?laqp2 semantics. Returns (factored F-order copy, jpvt0, tau)."""
    m, n = a.shape
    w = np.asfortranarray(a.copy())
    k = min(m, n) if kmax is None else min(kmax, m, n)
    jpvt = np.zeros(n, dtype=np.int64)
    tau = np.zeros(max(min(m, n), 1), dtype=a.dtype)
    fn = getattr(lib(), f"rco_geqp3_{_suf(a.dtype)}")
    fn.restype = ctypes.c_int
    rc = fn(_i64(m), _i64(n), _p(w), _i64(m), _p(jpvt), _p(tau), _i64(k), ctypes.c_int(1 if pivot else 0))
    assert rc == 0
    return w, jpvt, tau[: min(m, n)]


def orgqr(w: np.ndarray, tau: np.ndarray, nq: int, k: int) -> np.ndarray:
    m = w.shape[0]
    q = np.asfortranarray(w[:, :nq].copy())
    fn = getattr(lib(), f"rco_orgqr_{_suf(w.dtype)}")
    fn.restype = ctypes.c_int
    rc = fn(_i64(m), _i64(nq), _i64(k), _p(q), _i64(m), _p(np.ascontiguousarray(tau)))
    assert rc == 0
    return q


def pivoted_qr(a: np.ndarray, kmax=None):
    """Restates /root/reference/src/pivoted_qr.rs:81-119 without LAPACK. Returns (q, r, ind)."""
    m, n = a.shape
    k = min(m, n) if kmax is None else min(kmax, m, n)
    w, jpvt, tau = geqp3(a, kmax=k)
    r = np.triu(w[:k, :])
    q = orgqr(w, tau, k, k)
    return np.ascontiguousarray(q), np.ascontiguousarray(r), jpvt


def trtrs_upper(r: np.ndarray, b: np.ndarray) -> np.ndarray:
    k = r.shape[0]
    rr = np.asfortranarray(r)
    x = np.asfortranarray(b.copy().reshape(k, -1))
    fn = getattr(lib(), f"rco_trtrs_upper_{_suf(r.dtype)}")
    fn.restype = ctypes.c_int
    rc = fn(_i64(k), _i64(x.shape[1]), _p(rr), _i64(k), _p(x), _i64(k))
    assert rc == 0
    return np.ascontiguousarray(x.reshape(b.shape))


def svd_thin(a: np.ndarray):
    """Thin SVD via Householder QR/LQ + one-sided Jacobi (the algorithm the HIP path uses)."""
    m, n = a.shape
    r = min(m, n)
    aa = np.asfortranarray(a)
    u = np.zeros((m, r), dtype=a.dtype, order="F")
    s = np.zeros(r, dtype=a.dtype)
    vt = np.zeros((r, n), dtype=a.dtype, order="F")
    fn = getattr(lib(), f"rco_svd_thin_{_suf(a.dtype)}")
    fn.restype = ctypes.c_int
    sweeps = fn(_i64(m), _i64(n), _p(aa), _i64(m), _p(u), _i64(m), _p(s), _p(vt), _i64(r))
    assert sweeps >= 0
    return np.ascontiguousarray(u), s, np.ascontiguousarray(vt)
