"""GPU parity tests of the LAPACK-granularity entry points: rc_geqp3_* / rc_orgqr_* / rc_trsm_upper_* against the
routines the reference itself calls at those places -- ?geqp3 (`$qrf`, src/pivoted_qr.rs:139-172), ?orgqr / ?ungqr (`lax::Lapack::q`,
src/pivoted_qr.rs:104-108), ?trtrs (`solve_triangular`, src/qr.rs:298, :392) -- issued by the oracle (oracle/ref_lapack.py, SciPy
LAPACK; the reference holds no fixtures for them: parity pinned by the oracle only).

Tolerances: tests/helpers.py TOL (f64 / c64 factors <= 1e-10, f32 / c32 <= 1e-4 relative Frobenius); pivots exact on the prefix the
data determines; reflectors and tau compared entry for entry on that prefix (both sides use LAPACK's ?larfg sign convention).
"""
import numpy as np
import pytest

import rusty_compression_amd as rc
from oracle import ref_lapack as o
from tests.helpers import TOL, agreed_pivot_prefix, is_permutation, npy, rel

pytestmark = pytest.mark.gpu
FACTOR = {np.dtype(np.float64): 1e-10, np.dtype(np.complex128): 1e-10, np.dtype(np.float32): 1e-4, np.dtype(np.complex64): 1e-4}
# A[:, jpvt] = Q R and Q^H Q = I of our own factors (backward error; the c32 Householder chain accumulates ~k eps over k steps)
RECON = {np.dtype(np.float64): 1e-12, np.dtype(np.complex128): 1e-12, np.dtype(np.float32): 1e-5, np.dtype(np.complex64): 1e-4}
REAL = {np.dtype(np.float64): np.float64, np.dtype(np.complex128): np.float64, np.dtype(np.float32): np.float32, np.dtype(np.complex64): np.float32}
# tall (Householder chain / blocked panels), square, short-wide (the cooperative register-resident kernel), tiny
SHAPES = [(300, 60), (200, 200), (96, 4096), (64, 700), (5, 3), (1, 7)]
DTYPES = [np.float64, np.float32, np.complex128, np.complex64]


def _mat(dtype, shape, seed):
    smin = 1e-3 if REAL[np.dtype(dtype)] == np.float32 else 1e-6   # spectrum 1 .. smin: pivots well separated in the working precision
    return o.random_approximate_low_rank_matrix(shape, 1.0, smin, np.random.default_rng(seed), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", SHAPES)
def test_geqp3_and_orgqr_match_lapack_output_format(dtype, shape):
    dt = np.dtype(dtype)
    a = _mat(dtype, shape, 11 + shape[0])
    m, n = shape
    k = min(m, n)
    f, jp, tau = rc.geqp3(a)
    f, jp, tau = npy(f), npy(jp), npy(tau)
    fo, jpo, tauo = o.geqp3_raw(a)
    assert is_permutation(jp, n) and tau.shape == (k,)
    r, ro = np.triu(f[:k]), np.triu(fo[:k])
    ns = agreed_pivot_prefix(jp, r, jpo, ro, REAL[dt])
    assert ns >= min(k, 3), f"only {ns} pivots agree"
    tol = FACTOR[dt]
    # R, the reflectors and tau, entry for entry, on the columns both sides factored in the same order
    assert rel(r[:ns, :ns], ro[:ns, :ns]) <= tol
    assert rel(np.abs(np.diag(r))[:ns], np.abs(np.diag(ro))[:ns]) <= tol
    assert rel(np.tril(f[:, :ns], -1), np.tril(fo[:, :ns], -1)) <= 50 * tol   # v = x / (alpha - beta): one division more than R
    assert rel(tau[:ns], tauo[:ns]) <= 50 * tol
    # the factorization itself: A[:, jpvt] = Q R with Q from OUR reflectors through OUR ?orgqr
    q = npy(rc.orgqr(f, tau))
    assert q.shape == (m, k)
    wide = np.complex128 if dt.kind == "c" else np.float64
    assert rel(q.astype(wide) @ r.astype(wide), a[:, jp]) <= RECON[dt]
    assert rel(q.conj().T.astype(wide) @ q.astype(wide), np.eye(k)) <= RECON[dt]
    # ?orgqr alone: LAPACK's reflectors in, LAPACK's Q out
    assert rel(npy(rc.orgqr(fo, tauo)), o.orgqr_raw(fo, tauo)) <= tol
    # fewer columns than reflectors stored in a: the first kk reflectors only
    kk = max(1, k // 2)
    assert rel(npy(rc.orgqr(fo, tauo[:kk])), o.orgqr_raw(fo, tauo[:kk])) <= tol


@pytest.mark.parametrize("shape", [(8192, 133), (128, 8192)])
def test_geqp3_at_the_headline_shapes_of_cfg3(shape):
    """The two pivoted QRs of a cfg3 compression at their full sizes (the sketch Y: 8192 x 133, the projection B: 128 x 8192; f64)
    through the LAPACK seam: pivots on the determined prefix, R and tau against dgeqp3, Q against dorgqr."""
    dt = np.dtype(np.float64)
    a = _mat(np.float64, shape, 77)
    m, n = shape
    k = min(m, n)
    f, jp, tau = (npy(t) for t in rc.geqp3(a))
    fo, jpo, tauo = o.geqp3_raw(a)
    r, ro = np.triu(f[:k]), np.triu(fo[:k])
    ns = agreed_pivot_prefix(jp, r, jpo, ro, np.float64)
    assert is_permutation(jp, n) and ns >= k // 2, f"only {ns} of {k} pivots agree"
    assert rel(r[:ns, :ns], ro[:ns, :ns]) <= FACTOR[dt] and rel(tau[:ns], tauo[:ns]) <= 50 * FACTOR[dt]
    q = npy(rc.orgqr(f, tau))
    assert rel(q @ r, a[:, jp]) <= RECON[dt] and rel(q.T @ q, np.eye(k)) <= RECON[dt]
    assert rel(npy(rc.orgqr(fo, tauo)), o.orgqr_raw(fo, tauo)) <= FACTOR[dt]


@pytest.mark.parametrize("dtype", DTYPES)
def test_geqp3_truncated_and_strided_views(dtype):
    dt = np.dtype(dtype)
    a = _mat(dtype, (180, 140), 5)
    fo, jpo, tauo = o.geqp3_raw(a)
    f, jp, tau = rc.geqp3(a, kmax=25)
    f, jp, tau = npy(f), npy(jp), npy(tau)
    assert tau.shape == (25,) and is_permutation(jp, 140)
    assert np.array_equal(jp[:25], jpo[:25])                     # the first kmax steps are ?geqp3's first kmax steps
    assert rel(np.triu(f[:25]), np.triu(fo[:25])[:, np.argsort(jpo)][:, jp]) <= FACTOR[dt]   # R[:kmax] in OUR column order
    assert rel(np.tril(f[:, :25], -1), np.tril(fo[:, :25], -1)) <= 50 * FACTOR[dt]
    # a column-major (transposed) device view in place, through the C ABI directly
    import ctypes
    import torch
    from rusty_compression_amd import _lib
    at = torch.from_numpy(np.ascontiguousarray(a.T)).cuda().t()   # 180 x 140 with unit ROW stride
    jpv = torch.empty(140, dtype=torch.int64, device="cuda")
    tv = torch.empty(140, dtype=at.dtype, device="cuda")
    _lib.default_context().call(f"rc_geqp3_{_lib.suffix(at.dtype)}", _lib.mat(at), ctypes.c_int64(140), _lib.i64p(jpv), ctypes.c_void_p(tv.data_ptr()))
    ns = agreed_pivot_prefix(npy(jpv), np.triu(npy(at)[:140]), jpo, np.triu(fo[:140]), REAL[dt])
    assert ns >= 25 and rel(np.triu(npy(at)[:ns, :ns]), np.triu(fo[:ns, :ns])) <= FACTOR[dt]


@pytest.mark.parametrize("dtype", DTYPES)
def test_trsm_upper_matches_trtrs(dtype):
    dt = np.dtype(dtype)
    rng = np.random.default_rng(3)
    for k, nrhs in ((64, 1000), (37, 5), (1, 3), (130, 257)):
        t = np.triu(o.random_gaussian((k, k), rng, dtype)) + 4 * np.eye(k, dtype=dtype) * np.sqrt(k)
        b = o.random_gaussian((k, nrhs), rng, dtype)
        x = npy(rc.trsm_upper(t, b))
        assert rel(x, o.trtrs_upper(t, b)) <= FACTOR[dt]
    # the ID's use of it (src/qr.rs:290-301): Z = R11^-1 R12 of a pivoted QR
    a = _mat(dtype, (90, 150), 8)
    q, r, ind = o.pivoted_qr(a)
    z = npy(rc.trsm_upper(r[:40, :40], r[:40, 40:]))
    assert rel(z, o.trtrs_upper(np.ascontiguousarray(r[:40, :40]), np.ascontiguousarray(r[:40, 40:]))) <= FACTOR[dt] * 10


def test_lapack_level_argument_errors():
    a = _mat(np.float64, (20, 10), 1)
    with pytest.raises(AssertionError):   # RC_INVALID_ARGUMENT: the class of the reference's asserts / panics
        rc.geqp3(a, kmax=11)
    f, jp, tau = rc.geqp3(a)
    with pytest.raises(AssertionError):   # RC_INVALID_ARGUMENT: the class of the reference's asserts / panics
        rc.orgqr(f, tau, k=11)
    with pytest.raises(AssertionError):   # RC_INVALID_ARGUMENT: the class of the reference's asserts / panics
        rc.trsm_upper(np.eye(4), np.ones((5, 2)))
    f0, jp0, tau0 = rc.geqp3(a, kmax=0)
    assert np.array_equal(npy(jp0), np.arange(10)) and rel(npy(f0), a) == 0.0
