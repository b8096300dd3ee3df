"""CPU ORACLE (test infrastructure, never shipped, never on the product path).

Restates the hot path of rusty-compression (reference @ /root/reference, crate
v0.1.1) function by function, issuing the SAME LAPACK routines the reference
reaches through its third-party crates:

    lapack::{d,s}geqp3      <- src/pivoted_qr.rs:139-150, :161-172, :187-190
    lax::Lapack::q (= ?orgqr) <- src/pivoted_qr.rs:104-108
    ndarray_linalg svddc_into(JobSvd::Some) (= ?gesdd) <- src/compute_svd.rs:19
    SolveTriangular (= ?trtrs) <- src/qr.rs:290-301, :384-395
    ndarray::dot (BLAS gemv / gemm) <- src/types.rs:118-120, :128-132

The arithmetic of the reference lives in those crates (ndarray 0.15.*,
ndarray-linalg 0.16.*, lapack 0.*, lax 0.*; no Cargo.lock => unpinned) and in
whatever LAPACK backend the user links (dev build: openblas-system,
Cargo.toml:30).  Here the backend is SciPy's bundled OpenBLAS LAPACK
(scipy.linalg.lapack).  PARITY PINNING: the reference holds no golden vectors
except the permutation known-answer tests (src/permutation.rs:192-239), which
this file reproduces exactly (tests/test_oracle_cpu.py); every other reference
test is a property check, restated in tests/.  The sketch stage
(src/random_sampling.rs) has zero reference tests: parity for it is pinned by
this oracle's goldens only ("parity unpinned by the reference").

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import numpy as np
from scipy.linalg import lapack as _lapack

# ----------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------


def _lp(name: str, dtype) -> Callable:
    """The LAPACK routine the reference reaches for this scalar type (src/pivoted_qr.rs:187-190: s/d/c/z geqp3;
    ?orgqr is ?ungqr for the complex types)."""
    dt = np.dtype(dtype)
    pre = {np.dtype(np.float64): "d", np.dtype(np.float32): "s", np.dtype(np.complex128): "z", np.dtype(np.complex64): "c"}[dt]
    if dt.kind == "c" and name == "orgqr":
        name = "ungqr"
    return getattr(_lapack, pre + name)


def _h(x: np.ndarray) -> np.ndarray:
    """Conjugate transpose (`.t().map(|item| item.conj())` in the reference); the plain transpose for real types."""
    return x.conj().T


class CompressionError(Exception):
    """src/types.rs:15-16 `CompressionError`."""


class PivotedQRError(Exception):
    """src/types.rs:19-20 `PivotedQRError`."""


class LinalgError(Exception):
    """src/types.rs:13-14 `LinalgError`."""


# ----------------------------------------------------------------------------
# permutation.rs
# ----------------------------------------------------------------------------


def invert_permutation_vector(perm: np.ndarray) -> np.ndarray:
    """src/permutation.rs:28-38: inverse[perm[i]] = i."""
    perm = np.asarray(perm, dtype=np.int64)
    inv = np.zeros(perm.shape[0], dtype=np.int64)
    for i, e in enumerate(perm):
        inv[e] = i
    return inv


def apply_permutation_matrix(mat: np.ndarray, index_array: np.ndarray, mode: str) -> np.ndarray:
    """src/permutation.rs:84-144. mode in {COL, ROW, COLINV, ROWINV}."""
    m, n = mat.shape
    idx = np.asarray(index_array, dtype=np.int64)
    out = np.zeros((m, n), dtype=mat.dtype)
    if mode == "COL":
        assert idx.shape[0] == n, "Length of index array and number of columns differ."
        for i in range(n):
            out[:, i] = mat[:, idx[i]]
    elif mode == "ROW":
        assert idx.shape[0] == m, "Length of index array and number of rows differ."
        for i in range(m):
            out[i, :] = mat[idx[i], :]
    elif mode == "COLINV":
        assert idx.shape[0] == n, "Length of index array and number of columns differ."
        inv = invert_permutation_vector(idx)
        for i in range(n):
            out[:, i] = mat[:, inv[i]]
    elif mode == "ROWINV":
        assert idx.shape[0] == m, "Length of index array and number of rows differ."
        inv = invert_permutation_vector(idx)
        for i in range(m):
            out[i, :] = mat[inv[i], :]
    else:
        raise ValueError(mode)
    return out


def apply_permutation_vector(vec: np.ndarray, index_array: np.ndarray, mode: str) -> np.ndarray:
    """src/permutation.rs:153-183. mode in {NOINV, INV}."""
    n = vec.shape[0]
    idx = np.asarray(index_array, dtype=np.int64)
    assert idx.shape[0] == n
    out = np.zeros(n, dtype=vec.dtype)
    if mode == "INV":
        inv = invert_permutation_vector(idx)
        for i in range(n):
            out[i] = vec[inv[i]]
    elif mode == "NOINV":
        for i in range(n):
            out[i] = vec[idx[i]]
    else:
        raise ValueError(mode)
    return out


# ----------------------------------------------------------------------------
# types.rs: operator products and RelDiff
# ----------------------------------------------------------------------------


def matmat(a: np.ndarray, x: np.ndarray, faithful: bool = False) -> np.ndarray:
    """src/types.rs:60-70 (+ blanket impl :145): Y[:, j] = A.dot(X[:, j]).

    faithful=True reproduces the reference's call shape (one gemv per column);
    faithful=False is the single-GEMM form (differs by summation order only).
    """
    if not faithful:
        return a @ x
    out = np.zeros((a.shape[0], x.shape[1]), dtype=a.dtype)
    for j in range(x.shape[1]):
        out[:, j] = a.dot(x[:, j])
    return out


def conj_matmat(a: np.ndarray, x: np.ndarray, faithful: bool = False) -> np.ndarray:
    """src/types.rs:90-100 (+ :128-132, :146): out[:, j] = conj(conj(x_j) . A) = A^H x_j."""
    if not faithful:
        return _h(a) @ x
    out = np.zeros((a.shape[1], x.shape[1]), dtype=a.dtype)
    for j in range(x.shape[1]):
        out[:, j] = np.conj(np.conj(x[:, j]).dot(a))
    return out


def rel_diff_fro(first: np.ndarray, second: np.ndarray) -> float:
    """src/types.rs:182-188."""
    return float(np.linalg.norm(first - second, "fro") / np.linalg.norm(second, "fro"))


def rel_diff_l2(first: np.ndarray, second: np.ndarray) -> float:
    """src/types.rs:190-196."""
    return float(np.linalg.norm(first - second) / np.linalg.norm(second))


# ----------------------------------------------------------------------------
# random_matrix.rs (test-input generators; the Gaussian stream itself cannot
# match rand_distr's ziggurat sample-for-sample: Omega is always passed
# explicitly in parity tests)
# ----------------------------------------------------------------------------


def random_gaussian(shape: Tuple[int, int], rng: np.random.Generator, dtype=np.float64) -> np.ndarray:
    """src/random_matrix.rs:120-125 (real), :136-143 (complex: real and imaginary part each N(0,1), drawn in that
    order per element): drawn as f64, row-major fill, cast."""
    if np.dtype(dtype).kind == "c":
        g = rng.standard_normal(tuple(shape) + (2,))
        return (g[..., 0] + 1j * g[..., 1]).astype(dtype)
    return rng.standard_normal(shape).astype(dtype)


def random_orthogonal_matrix(shape, rng, dtype=np.float64) -> np.ndarray:
    """src/random_matrix.rs:35-56: U of the thin SVD (gesdd) of a Gaussian."""
    m, n = shape
    swap = n > m
    if swap:
        m, n = n, m
    g = random_gaussian((m, n), rng, dtype)
    u, _, _ = compute_svd(g)
    return _h(u).copy() if swap else u


def random_approximate_low_rank_matrix(shape, sigma_max, sigma_min, rng, dtype=np.float64) -> np.ndarray:
    """src/random_matrix.rs:70-93: U diag(geomspace(sigma_min, sigma_max)) Vt (ascending!)."""
    assert sigma_min < sigma_max and sigma_min > 0.0
    m, n = shape
    r = min(m, n)
    u = random_orthogonal_matrix((m, r), rng, dtype)
    vt = random_orthogonal_matrix((r, n), rng, dtype)
    s = np.geomspace(sigma_min, sigma_max, r).astype(np.empty(0, dtype).real.dtype)
    return (u @ (np.diag(s) @ vt)).astype(dtype)


# ----------------------------------------------------------------------------
# pivoted_qr.rs
# ----------------------------------------------------------------------------


def pivoted_qr(arr: np.ndarray):
    """src/pivoted_qr.rs:25-31, :81-119, :121-183.

    F-order working copy; ?geqp3 with jpvt zero-initialised (all columns free,
    :132); jpvt-1 (:177); R = upper triangle of rows 0..k (:100-102);
    Q = first k columns after ?orgqr (:104-114). Returns (q, r, ind).
    """
    arr = np.asarray(arr)
    m, n = arr.shape
    k = min(m, n)
    a = np.asfortranarray(arr.copy())
    geqp3 = _lp("geqp3", arr.dtype)
    qr_, jpvt, tau, work, info = geqp3(a, overwrite_a=1)
    if info != 0:
        raise PivotedQRError(info)
    ind = jpvt.astype(np.int64) - 1
    r = np.triu(qr_[:k, :]).astype(arr.dtype)
    orgqr = _lp("orgqr", arr.dtype)
    # lax::Lapack::q on an m x n F-layout buffer forms min(m, n) columns
    q_, work, info = orgqr(qr_[:, :k].copy(order="F"), tau)
    if info != 0:
        raise PivotedQRError(info)
    q = np.ascontiguousarray(q_[:, :k])
    return q, np.ascontiguousarray(r), ind


def geqp3_raw(arr: np.ndarray):
    """The `$qrf` call alone (src/pivoted_qr.rs:139-150, :161-172): returns (qr, jpvt0, tau) exactly as ?geqp3 leaves them
    (qr = R on / above the diagonal and the Householder vectors below it, columns in pivoted order; jpvt0 = jpvt - 1, :177)."""
    arr = np.asarray(arr)
    qr_, jpvt, tau, work, info = _lp("geqp3", arr.dtype)(np.asfortranarray(arr.copy()), overwrite_a=1)
    if info != 0:
        raise PivotedQRError(info)
    return qr_, jpvt.astype(np.int64) - 1, tau


def orgqr_raw(qr_: np.ndarray, tau: np.ndarray) -> np.ndarray:
    """`lax::Lapack::q` alone (src/pivoted_qr.rs:104-108): Q = H_0 ... H_{k-1} [I; 0], k = len(tau) columns."""
    k = len(tau)
    q_, work, info = _lp("orgqr", qr_.dtype)(np.asfortranarray(qr_[:, :k].copy()), tau)
    if info != 0:
        raise PivotedQRError(info)
    return np.ascontiguousarray(q_[:, :k])


def trtrs_upper(t: np.ndarray, b: np.ndarray) -> np.ndarray:
    """`solve_triangular(UPLO::Upper, Diag::NonUnit, ..)` (src/qr.rs:298, :392) for all right-hand sides at once."""
    x, info = _lp("trtrs", t.dtype)(np.asfortranarray(t), np.asfortranarray(b), lower=0, trans=0, unitdiag=0)
    if info != 0:
        raise LinalgError(info)
    return np.ascontiguousarray(x)


def pivoted_lq(arr: np.ndarray):
    """src/pivoted_qr.rs:32-41: pivoted QR of arr^H, transposed back. Returns (l, q, ind)."""
    q, r, ind = pivoted_qr(np.ascontiguousarray(_h(arr)))
    return np.ascontiguousarray(_h(r)), np.ascontiguousarray(_h(q)), ind


# ----------------------------------------------------------------------------
# compute_svd.rs
# ----------------------------------------------------------------------------


def compute_svd(arr: np.ndarray):
    """src/compute_svd.rs:18-27: ?gesdd, JobSvd::Some (thin). Returns (u, s, vt)."""
    arr = np.asarray(arr)
    gesdd = _lp("gesdd", arr.dtype)
    u, s, vt, info = gesdd(np.asfortranarray(arr.copy()), compute_uv=1, full_matrices=0)
    if info != 0:
        raise LinalgError(info)
    return np.ascontiguousarray(u), s.copy(), np.ascontiguousarray(vt)


# ----------------------------------------------------------------------------
# qr.rs / svd.rs / *_interp_decomp.rs containers
# ----------------------------------------------------------------------------


@dataclass
class QR:
    q: np.ndarray
    r: np.ndarray
    ind: np.ndarray

    # src/qr.rs:145-157
    def nrows(self):
        return self.q.shape[0]

    def ncols(self):
        return self.r.shape[1]

    def rank(self):
        return self.q.shape[1]

    @staticmethod
    def compute_from(arr) -> "QR":
        """src/qr.rs:251-253."""
        return QR(*pivoted_qr(arr))

    def to_mat(self) -> np.ndarray:
        """src/qr.rs:160-166: Q . (R with COLINV permutation)."""
        return self.q @ apply_permutation_matrix(self.r, self.ind, "COLINV")

    def compress_qr_rank(self, max_rank: int) -> "QR":
        """src/qr.rs:169-184 (ind kept full length)."""
        max_rank = min(max_rank, self.q.shape[1])
        return QR(self.q[:, :max_rank].copy(), self.r[:max_rank, :].copy(), self.ind.copy())

    def compress_qr_tolerance(self, tol: float) -> "QR":
        """src/qr.rs:187-200: first i with |r_ii / r_00| < tol, else CompressionError."""
        assert 0.0 <= tol < 1.0, "Require 0 <= tol < 1.0"
        d = np.diag(self.r)
        for i, item in enumerate(d):
            if abs(item / self.r[0, 0]) < tol:
                return self.compress_qr_rank(i)
        raise CompressionError()

    def compress(self, kind: str, value) -> "QR":
        """src/qr.rs:203-208. kind in {RANK, ADAPTIVE}."""
        if kind == "ADAPTIVE":
            return self.compress_qr_tolerance(float(value))
        return self.compress_qr_rank(int(value))

    def column_id(self) -> "ColumnID":
        """src/qr.rs:270-309."""
        rank = self.rank()
        ncols = self.ncols()
        dtype = self.q.dtype
        if rank == ncols:
            return ColumnID(
                self.q @ self.r,
                apply_permutation_matrix(np.eye(rank, dtype=dtype), self.ind, "COLINV"),
                self.ind.copy(),
            )
        z = np.zeros((rank, ncols), dtype=dtype)
        z[:, :rank] = np.eye(rank, dtype=dtype)
        first_part = self.r[:, :rank].copy()
        c = self.q @ first_part
        trtrs = _lp("trtrs", dtype)
        for index in range(ncols - rank):
            col = self.r[:, rank + index].copy()
            x, info = trtrs(first_part, col, lower=0, trans=0, unitdiag=0)
            if info != 0:
                raise LinalgError(info)
            z[:, rank + index] = x.reshape(-1)
        return ColumnID(c, apply_permutation_matrix(z, self.ind, "COLINV"), self.ind.copy())

    @staticmethod
    def compute_from_range_estimate(range_: np.ndarray, op: np.ndarray, faithful: bool = False) -> "QR":
        """src/qr.rs:311-323."""
        b = np.ascontiguousarray(_h(conj_matmat(op, range_, faithful)))
        qr = QR.compute_from(b)
        return QR(range_ @ qr.q, qr.r.copy(), qr.ind.copy())


@dataclass
class LQ:
    l: np.ndarray
    q: np.ndarray
    ind: np.ndarray

    # src/qr.rs:58-70
    def nrows(self):
        return self.l.shape[0]

    def ncols(self):
        return self.q.shape[1]

    def rank(self):
        return self.q.shape[0]

    @staticmethod
    def compute_from(arr) -> "LQ":
        """src/qr.rs:354-362."""
        q, r, ind = pivoted_qr(np.ascontiguousarray(_h(np.asarray(arr))))
        return LQ(np.ascontiguousarray(_h(r)), np.ascontiguousarray(_h(q)), ind)

    def to_mat(self) -> np.ndarray:
        """src/qr.rs:73-77."""
        return apply_permutation_matrix(self.l, self.ind, "ROWINV") @ self.q

    def compress_lq_rank(self, max_rank: int) -> "LQ":
        """src/qr.rs:80-96."""
        max_rank = min(max_rank, self.q.shape[0])
        return LQ(self.l[:, :max_rank].copy(), self.q[:max_rank, :].copy(), self.ind.copy())

    def compress_lq_tolerance(self, tol: float) -> "LQ":
        """src/qr.rs:99-112."""
        assert 0.0 <= tol < 1.0, "Require 0 <= tol < 1.0"
        d = np.diag(self.l)
        for i, item in enumerate(d):
            if abs(item / self.l[0, 0]) < tol:
                return self.compress_lq_rank(i)
        raise CompressionError()

    def compress(self, kind: str, value) -> "LQ":
        """src/qr.rs:114-119."""
        if kind == "ADAPTIVE":
            return self.compress_lq_tolerance(float(value))
        return self.compress_lq_rank(int(value))

    def row_id(self) -> "RowID":
        """src/qr.rs:363-403."""
        rank = self.rank()
        nrows = self.nrows()
        dtype = self.q.dtype
        if rank == nrows:
            return RowID(
                apply_permutation_matrix(np.eye(rank, dtype=dtype), self.ind, "ROWINV"),
                self.l @ self.q,
                self.ind.copy(),
            )
        x = np.zeros((nrows, rank), dtype=dtype)
        x[:rank, :] = np.eye(rank, dtype=dtype)
        first_part = self.l[:rank, :].copy()
        r = first_part @ self.q
        first_part_t = np.ascontiguousarray(first_part.T)
        trtrs = _lp("trtrs", dtype)
        for index in range(nrows - rank):
            row = self.l[rank + index, :].copy()
            sol, info = trtrs(first_part_t, row, lower=0, trans=0, unitdiag=0)
            if info != 0:
                raise LinalgError(info)
            x[rank + index, :] = sol.reshape(-1)
        return RowID(apply_permutation_matrix(x, self.ind, "ROWINV"), r, self.ind.copy())


@dataclass
class SVD:
    u: np.ndarray
    s: np.ndarray
    vt: np.ndarray

    def nrows(self):
        return self.u.shape[0]

    def ncols(self):
        return self.vt.shape[1]

    def rank(self):
        return self.u.shape[1]

    @staticmethod
    def compute_from(arr) -> "SVD":
        """src/svd.rs:165-169."""
        return SVD(*compute_svd(arr))

    def to_mat(self) -> np.ndarray:
        """src/svd.rs:42-54."""
        return self.u @ (self.s[:, None].astype(self.vt.dtype) * self.vt)

    def compress_svd_rank(self, max_rank: int) -> "SVD":
        """src/svd.rs:68-84."""
        max_rank = min(max_rank, self.s.shape[0])
        return SVD(self.u[:, :max_rank].copy(), self.s[:max_rank].copy(), self.vt[:max_rank, :].copy())

    def compress_svd_tolerance(self, tol: float) -> "SVD":
        """src/svd.rs:87-101."""
        assert 0.0 <= tol < 1.0, "Require 0 <= tol < 1.0"
        first = self.s[0]
        for i, item in enumerate(self.s):
            if float(item / first) < tol:
                return self.compress_svd_rank(i)
        raise CompressionError()

    def compress(self, kind: str, value) -> "SVD":
        """src/svd.rs:60-65."""
        if kind == "ADAPTIVE":
            return self.compress_svd_tolerance(float(value))
        return self.compress_svd_rank(int(value))

    def to_qr(self) -> QR:
        """src/svd.rs:150-163: QRCP of diag(S) Vt, Q = U Q_b."""
        svt = self.s[:, None].astype(self.vt.dtype) * self.vt
        qr = QR.compute_from(svt)
        qr.q = self.u @ qr.q
        return qr

    @staticmethod
    def compute_from_range_estimate(range_: np.ndarray, op: np.ndarray, faithful: bool = False) -> "SVD":
        """src/svd.rs:171-183."""
        b = np.ascontiguousarray(_h(conj_matmat(op, range_, faithful)))
        svd = SVD.compute_from(b)
        return SVD(range_ @ svd.u, svd.s.copy(), svd.vt.copy())


@dataclass
class ColumnID:
    c: np.ndarray
    z: np.ndarray
    col_ind: np.ndarray

    def rank(self):
        return self.c.shape[1]

    def to_mat(self):
        """src/col_interp_decomp.rs:63-65."""
        return self.c @ self.z

    def two_sided_id(self) -> "TwoSidedID":
        """src/col_interp_decomp.rs:116-125."""
        row_id = LQ.compute_from(self.c).row_id()
        return TwoSidedID(row_id.x.copy(), row_id.r.copy(), self.z.copy(), row_id.row_ind.copy(), self.col_ind.copy())

    def dot(self, rhs):
        """src/col_interp_decomp.rs:134-154."""
        return self.c @ (self.z @ rhs)


@dataclass
class RowID:
    x: np.ndarray
    r: np.ndarray
    row_ind: np.ndarray

    def rank(self):
        return self.r.shape[0]

    def to_mat(self):
        """src/row_interp_decomp.rs:65-67."""
        return self.x @ self.r

    def two_sided_id(self) -> "TwoSidedID":
        """src/row_interp_decomp.rs:120-130."""
        col_id = QR.compute_from(self.r).column_id()
        return TwoSidedID(self.x.copy(), col_id.c.copy(), col_id.z.copy(), self.row_ind.copy(), col_id.col_ind.copy())

    def dot(self, rhs):
        """src/row_interp_decomp.rs:134-154."""
        return self.x @ (self.r @ rhs)


@dataclass
class TwoSidedID:
    """src/two_sided_interp_decomp.rs:19-30 (field order c, x, r, row_ind, col_ind)."""

    c: np.ndarray
    x: np.ndarray
    r: np.ndarray
    row_ind: np.ndarray
    col_ind: np.ndarray

    def rank(self):
        return self.c.shape[1]

    def to_mat(self):
        """src/two_sided_interp_decomp.rs:62-64."""
        return self.c @ (self.x @ self.r)

    def dot(self, rhs):
        """src/two_sided_interp_decomp.rs:154-171."""
        return self.c @ (self.x @ (self.r @ rhs))


# ----------------------------------------------------------------------------
# random_sampling.rs
# ----------------------------------------------------------------------------

OmegaSource = Callable[[Tuple[int, int]], np.ndarray]


def sample_range_by_rank(op: np.ndarray, k: int, p: int, omega_source: OmegaSource, faithful: bool = False) -> np.ndarray:
    """src/random_sampling.rs:103-118. omega_source((n, k+p)) supplies Omega."""
    n = op.shape[1]
    omega = omega_source((n, k + p))
    basis = matmat(op, omega, faithful)
    qr = QR.compute_from(basis).compress("RANK", k)
    return qr.q.copy()


def sample_range_power_iteration(op, k, p, it_count, omega_source: OmegaSource, faithful: bool = False, fixed: bool = False) -> np.ndarray:
    """src/random_sampling.rs:131-160, including the shadowing quirk: every
    iteration restarts from the outer op_omega (:145) and only the last
    iteration's product is kept (:150-153).  fixed=True is the algorithm the
    reference documents (the product is carried from iteration to iteration)."""
    n = op.shape[1]
    omega = omega_source((n, k + p))
    op_omega = matmat(op, omega, faithful)
    res = op_omega.copy()
    for index in range(it_count):
        q = QR.compute_from(op_omega).q
        w = QR.compute_from(conj_matmat(op, q, faithful)).q
        inner = matmat(op, w, faithful)  # shadows, outer op_omega untouched
        if fixed:
            op_omega = inner
        if index == it_count - 1:
            res = inner.copy()
    return QR.compute_from(res).compress("RANK", k).q.copy()


def max_col_norm(mat: np.ndarray) -> float:
    """src/random_sampling.rs:184-191."""
    mx = np.empty(0, mat.dtype).real.dtype.type(0)
    for j in range(mat.shape[1]):
        mx = max(mx, np.linalg.norm(mat[:, j]))
    return mx


def sample_range_adaptive(op, rel_tol: float, sample_size: int, omega_source: OmegaSource,
                          faithful: bool = False, max_iter: Optional[int] = None):
    """src/random_sampling.rs:223-274. Returns (q, residuals)."""
    dtype = op.dtype
    real = np.empty(0, dtype).real.dtype.type
    tol_factor = real(10.0 * math.sqrt(2.0 / math.pi))
    m_rows, n = op.shape
    rel_tol_r = real(rel_tol)
    omega = omega_source((n, sample_size))
    op_omega = matmat(op, omega, faithful)
    operator_norm = real(max_col_norm(op_omega)) * tol_factor
    max_norm = operator_norm
    q = np.zeros((m_rows, 0), dtype=dtype)
    b = np.zeros((0, n), dtype=dtype)
    residuals: List[Tuple[int, float]] = []
    it = 0
    while max_norm / operator_norm >= rel_tol_r:
        if q.shape[1] > 0:
            op_omega = op_omega - q @ (_h(q) @ op_omega)
        qr = QR.compute_from(op_omega)
        b = np.concatenate([b, np.ascontiguousarray(_h(conj_matmat(op, qr.q, faithful)))], axis=0)
        q = np.concatenate([q, qr.q], axis=1)
        omega = omega_source((n, sample_size))
        op_omega = matmat(op, omega, faithful) - q @ (b @ omega)
        max_norm = real(max_col_norm(op_omega)) * tol_factor
        residuals.append((q.shape[1], float(max_norm / operator_norm)))
        it += 1
        if max_iter is not None and it >= max_iter:
            break
    return q, residuals


# ----------------------------------------------------------------------------
# reference-shaped end-to-end pipelines used by bench.py's cpu_baseline leg
# ----------------------------------------------------------------------------


def rsvd_id_reference_shape(a: np.ndarray, omega: np.ndarray, k: int, faithful: bool = True):
    """cfg3 'rSVD+ID' as the reference executes it: sample_range_by_rank ->
    SVD::compute_from_range_estimate -> QR::compute_from_range_estimate ->
    column_id."""
    p = omega.shape[1] - k
    q = sample_range_by_rank(a, k, p, lambda shp: omega, faithful)
    svd = SVD.compute_from_range_estimate(q, a, faithful)
    qr = QR.compute_from_range_estimate(q, a, faithful)
    cid = qr.column_id()
    return q, svd, qr, cid
