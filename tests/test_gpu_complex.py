"""GPU parity tests of the complex instantiations (c32 / c64): every call goes through the
`_c32` / `_c64` entry points of the C ABI and is compared with the oracle (oracle/ref_lapack.py issuing
zgeqp3 / zungqr / zgesdd / ztrtrs -- the routines the reference reaches for complex scalars, src/pivoted_qr.rs:189-190)
and with the reference's own unit tests for the complex types (the c32 / c64 half of its 89 tests, restated with seeds).

Tolerances: c64 factors <= 1e-10 relative Frobenius, singular values <= 1e-12; c32 1e-4 / 1e-5 (tests/helpers.py TOL).
Permutation indices bit-exact on the prefix the data determines.
"""
import numpy as np
import pytest
import torch

import rusty_compression_amd as rc
from oracle import ref_lapack as o
from tests.helpers import agreed_pivot_prefix, is_permutation, npy, rel, stable_prefix

pytestmark = pytest.mark.gpu
CT = rc.CompressionType
TOLC = {np.dtype(np.complex128): dict(factor=1e-10, sval=1e-12, orth=1e-12, recon=1e-12, real=np.float64),
        np.dtype(np.complex64): dict(factor=1e-4, sval=1e-5, orth=2e-5, recon=2e-5, real=np.float32)}
CASES = [(np.complex128, (100, 50)), (np.complex64, (100, 50)), (np.complex128, (50, 100)), (np.complex64, (50, 100))]


def _mat(dtype, shape, smin, seed):
    return o.random_approximate_low_rank_matrix(shape, 1.0, smin, np.random.default_rng(seed), dtype)


@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
def test_complex_gemm_all_ops_and_conj_matmat(dtype):
    rng = np.random.default_rng(0)
    tol = 1e-13 if dtype == np.complex128 else 3e-6
    for (m, k, n) in ((37, 29, 41), (128, 300, 65), (5, 1, 9), (1, 64, 1)):
        a = o.random_gaussian((m, k), rng, dtype)
        b = o.random_gaussian((k, n), rng, dtype)
        ref = a.astype(np.complex128) @ b.astype(np.complex128)
        assert rel(npy(rc.dot(a, b)), ref) <= tol
        assert rel(npy(rc.matmat(a, b)), ref) <= tol
        y = o.random_gaussian((m, n), rng, dtype)
        assert rel(npy(rc.conj_matmat(a, y)), a.conj().T.astype(np.complex128) @ y.astype(np.complex128)) <= tol  # A^H X (src/types.rs:128-132)
        # strided / transposed device views
        ta = torch.from_numpy(np.ascontiguousarray(a.T)).cuda().t()
        assert rel(npy(rc.dot(ta, b)), ref) <= tol
    v = o.random_gaussian((29, 1), rng, dtype)[:, 0]
    a = o.random_gaussian((37, 29), rng, dtype)
    assert rel(npy(rc.dot(a, v)), a.astype(np.complex128) @ v.astype(np.complex128)) <= tol


def test_complex_gaussian_stream_order_and_permutations():
    from oracle import philox as ph

    # element (i, j): re = normal 2 (offset + i cols + j), im = the next one (src/random_matrix.rs:136-143)
    g = npy(rc.random_gaussian((7, 5), rc.Rng(3, 11), torch.complex128))
    z = ph.normals(3, 2 * 11, 2 * 35).reshape(7, 5, 2)
    assert (np.abs(g.real - z[..., 0]) <= 4 * np.spacing(np.abs(z[..., 0]))).all()
    assert (np.abs(g.imag - z[..., 1]) <= 4 * np.spacing(np.abs(z[..., 1]))).all()
    g32 = npy(rc.random_gaussian((7, 5), rc.Rng(3, 11), torch.complex64))
    assert np.array_equal(g32, g.astype(np.complex64))
    rng = np.random.default_rng(2)
    a = o.random_gaussian((9, 6), rng, np.complex128)
    pc, pr = rng.permutation(6), rng.permutation(9)
    for mode, p in (("COL", pc), ("COLINV", pc), ("ROW", pr), ("ROWINV", pr)):
        assert np.array_equal(npy(rc.apply_permutation(a, p, rc.MatrixPermutationMode[mode])), o.apply_permutation_matrix(a, p, mode))
    assert abs(rc.rel_diff_fro(a * (1 + 1e-3), a) - 1e-3) < 1e-12
    assert abs(rc.max_col_norm(a) - o.max_col_norm(a)) <= 1e-13 * o.max_col_norm(a)


@pytest.mark.parametrize("dtype,shape", CASES)
def test_complex_pivoted_qr_lq_match_zgeqp3(dtype, shape):
    """src/pivoted_qr.rs:296-316 (c64 / c32 rows) + oracle parity: identical pivots, R and Q to tolerance."""
    t = TOLC[np.dtype(dtype)]
    a = _mat(dtype, shape, 1e-5, 100 + shape[0])
    k = min(shape)
    q, r, ind = (npy(x) for x in rc.pivoted_qr(a))
    oq, orr, oind = o.pivoted_qr(a)
    assert is_permutation(ind, shape[1])
    ns = agreed_pivot_prefix(ind[:k], r, oind[:k], orr, t["real"])
    assert ns == min(k, stable_prefix(orr, t["real"]))
    assert np.abs(q.conj().T @ q - np.eye(k)).max() <= t["orth"] * 10           # reference: 1e-6
    assert rel(q @ r, a[:, ind]) <= t["recon"] * 10
    assert np.abs(np.diag(r).imag).max() <= 1e-6 * np.abs(r[0, 0])               # ?geqp3: real diagonal
    assert rel(r[:ns], orr[:ns]) <= t["factor"] * 3
    lead = min(ns, int((np.abs(np.diag(orr)) > np.abs(orr[0, 0]) * 1e-2).sum()))
    assert rel(q[:, :lead], oq[:, :lead]) <= t["factor"] * 10
    l, ql, indl = (npy(x) for x in rc.pivoted_lq(a))
    ol, oql, oindl = o.pivoted_lq(a)
    nl = agreed_pivot_prefix(indl[:k], l.conj().T, oindl[:k], ol.conj().T, t["real"])
    assert nl == min(k, stable_prefix(ol.conj().T, t["real"]))
    assert np.abs(ql @ ql.conj().T - np.eye(k)).max() <= t["orth"] * 10
    assert rel(l @ ql, a[indl, :]) <= t["recon"] * 10
    assert rel(l[:, :nl], ol[:, :nl]) <= t["factor"] * 3


@pytest.mark.parametrize("dtype,shape", CASES)
def test_complex_svd_matches_zgesdd_and_reference_properties(dtype, shape):
    """src/svd.rs:289-320 (complex rows): SVD -> QR -> matrix, RANK(20), ADAPTIVE(1e-4); S against ?gesdd."""
    t = TOLC[np.dtype(dtype)]
    a = _mat(dtype, shape, 1e-10, 200 + shape[0])
    svd = rc.SVD.compute_from(a)
    u, s, vt = npy(svd.u), npy(svd.s), npy(svd.vt)
    so = o.compute_svd(a)[1]
    r = min(shape)
    assert s.dtype == t["real"] and np.all(s[:-1] >= s[1:])
    assert np.abs(s - so).max() <= t["sval"] * so[0]
    assert rel((u * s) @ vt, a) <= t["recon"] * 10
    lead = int((so > so[0] * (1e-6 if dtype == np.complex128 else 1e-2)).sum())
    assert np.abs(u[:, :lead].conj().T @ u[:, :lead] - np.eye(lead)).max() <= t["orth"] * 10
    assert np.abs(vt[:lead] @ vt[:lead].conj().T - np.eye(lead)).max() <= t["orth"] * 10
    assert r == s.shape[0]
    assert rc.rel_diff_fro(svd.to_qr().to_mat(), a) < (1e-12 if dtype == np.complex128 else 1e-5)     # test_svd_to_qr_*
    c = svd.compress(CT.RANK(20))
    assert c.u.shape[1] == 20 and c.vt.shape[0] == 20 and rc.rel_diff_fro(c.to_mat(), a) < 1e-4       # test_svd_compression_by_rank_*
    assert rc.rel_diff_fro(svd.compress(CT.ADAPTIVE(1e-4)).to_mat(), a) < 1e-4                       # test_svd_compression_by_tol_*
    assert svd.compress(CT.ADAPTIVE(1e-4)).rank() == o.SVD.compute_from(a).compress("ADAPTIVE", 1e-4).rank()


@pytest.mark.parametrize("dtype,shape", CASES)
def test_complex_qr_compression_and_ids_reference_properties(dtype, shape):
    """src/qr.rs:573-615, src/col_interp_decomp.rs:232-241, src/row_interp_decomp.rs:226-235 (complex rows), and the
    factors against the oracle's."""
    t = TOLC[np.dtype(dtype)]
    tol = 1e-4
    m, n = shape
    a = _mat(dtype, shape, 1e-10, 300 + m)
    full = rc.QR.compute_from(a)
    qr30 = full.compress(CT.RANK(30))
    assert qr30.q.shape[1] == 30 and qr30.r.shape[0] == 30 and rc.rel_diff_fro(qr30.to_mat(), a) < 1e-4   # by_rank
    qr = full.compress(CT.ADAPTIVE(tol))
    assert rc.rel_diff_fro(qr.to_mat(), a) < 5 * tol and qr.rank() < min(m, n)                            # by_tol
    oqr = o.QR.compute_from(a).compress("ADAPTIVE", tol)
    assert qr.rank() == oqr.rank()
    cid, ocid = qr.column_id(), oqr.column_id()
    assert rc.rel_diff_fro(cid.to_mat(), a) < 5 * tol
    perm = npy(cid.col_ind)
    for i in range(qr.rank()):                                                                           # col_id: C columns are A columns
        assert np.linalg.norm(a[:, perm[i]] - npy(cid.c)[:, i]) / np.linalg.norm(npy(cid.c)[:, i]) < tol
    assert np.array_equal(perm[:qr.rank()], ocid.col_ind[:qr.rank()])
    assert rel(npy(cid.c), ocid.c) <= t["factor"] * 10 and rel(npy(cid.z), ocid.z) <= t["factor"] * 100
    lq = rc.LQ.compute_from(a).compress(CT.ADAPTIVE(tol))
    rid = lq.row_id()
    orid = o.LQ.compute_from(a).compress("ADAPTIVE", tol).row_id()
    assert rc.rel_diff_fro(rid.to_mat(), a) < 5 * tol
    rperm = npy(rid.row_ind)
    for i in range(lq.rank()):                                                                           # row_id: R rows are A rows
        assert np.linalg.norm(a[rperm[i], :] - npy(rid.r)[i, :]) / np.linalg.norm(npy(rid.r)[i, :]) < tol
    assert rel(npy(rid.x), orid.x) <= t["factor"] * 100 and rel(npy(rid.r), orid.r) <= t["factor"] * 10
    for ts, rank in ((cid.two_sided_id(), qr.rank()), (rid.two_sided_id(), lq.rank())):                   # two-sided from col / row ID
        assert rc.rel_diff_fro(ts.to_mat(), a) < 5 * tol
        x = npy(ts.x)
        assert x.shape == (rank, rank)
        sub = a[np.ix_(npy(ts.row_ind)[:rank], npy(ts.col_ind)[:rank])]
        assert (np.abs(x - sub) < 10 * tol * np.abs(sub)).all()
    # Apply (src/col_interp_decomp.rs:134-154)
    rhs = o.random_gaussian((n, 3), np.random.default_rng(1), dtype)
    assert rel(npy(cid.dot(rhs)), npy(cid.to_mat()).astype(np.complex128) @ rhs) <= t["recon"] * 100


@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
def test_complex_samplers_match_the_oracle_with_explicit_omega(dtype):
    """sample_range_by_rank / power iteration (with the reference's single surviving step) / adaptive, SVD and QR from the
    range estimate (src/random_sampling.rs:103-274, src/svd.rs:171-183, src/qr.rs:311-323) for complex scalars."""
    t = TOLC[np.dtype(dtype)]
    rng = np.random.default_rng(9)
    a = o.random_approximate_low_rank_matrix((300, 200), 1.0, 1e-8 if dtype == np.complex128 else 1e-4, rng, dtype)
    k, p = 24, 5
    om = o.random_gaussian((200, k + p), rng, dtype)
    q = npy(rc.sample_range_by_rank(a, k, p, om))
    oq = o.sample_range_by_rank(a, k, p, lambda s: om)
    assert rel(q, oq) <= t["factor"] * 10 and np.abs(q.conj().T @ q - np.eye(k)).max() <= t["orth"] * 10
    qp = npy(rc.sample_range_power_iteration(a, k, p, 2, om))
    oqp = o.sample_range_power_iteration(a, k, p, 2, lambda s: om)
    assert rel(qp, oqp) <= t["factor"] * 100
    svd = rc.SVD.compute_from_range_estimate(q, a)
    osvd = o.SVD.compute_from_range_estimate(oq, a)
    assert np.abs(npy(svd.s) - osvd.s).max() <= t["sval"] * 10 * osvd.s[0]
    assert rel(npy(svd.to_mat()), osvd.to_mat()) <= t["factor"] * 10
    qr = rc.QR.compute_from_range_estimate(q, a)
    oqr = o.QR.compute_from_range_estimate(oq, a)
    ns = min(k, stable_prefix(oqr.r, t["real"]))
    assert np.array_equal(npy(qr.ind)[:ns], oqr.ind[:ns])
    assert rel(npy(qr.r), oqr.r) <= t["factor"] * 10 and rel(npy(qr.to_mat()), oqr.to_mat()) <= t["factor"] * 10
    # adaptive range finder with explicit Omega blocks
    b = o.random_approximate_low_rank_matrix((200, 120), 1.0, 1e-8 if dtype == np.complex128 else 1e-4, rng, dtype)
    s = 5
    omegas = o.random_gaussian((120, s * 30), rng, dtype)
    cnt = [0]

    def src(shape):
        blk = omegas[:, cnt[0] * s:(cnt[0] + 1) * s]
        cnt[0] += 1
        return blk

    rel_tol = 1e-5 if dtype == np.complex128 else 1e-3
    oqa, ores = o.sample_range_adaptive(b, rel_tol, s, src)
    qa, res = rc.sample_range_adaptive(b, rel_tol, s, omegas[:, : s * cnt[0]])
    assert qa.shape[1] == oqa.shape[1] and [r_ for r_, _ in res] == [r_ for r_, _ in ores]
    assert np.allclose([e for _, e in res], [e for _, e in ores], rtol=1e-6 if dtype == np.complex128 else 5e-2)
    qan = npy(qa)
    assert np.abs(qan.conj().T @ qan - np.eye(qan.shape[1])).max() <= t["orth"] * 100
    assert rel(qan @ (qan.conj().T @ b), oqa @ (oqa.conj().T @ b)) <= t["factor"] * 100


def test_complex_rank_k_column_id_and_error_behaviour():
    rng = np.random.default_rng(4)
    a = o.random_approximate_low_rank_matrix((160, 120), 1.0, 1e-10, rng, np.complex128)
    from rusty_compression_amd import batch

    c, z, ind = batch.column_id_rank(torch.from_numpy(a).cuda(), 25)
    ocid = o.QR.compute_from(a).compress("RANK", 25).column_id()
    assert np.array_equal(npy(ind)[:25], ocid.col_ind[:25])
    assert rel(npy(c), ocid.c) <= 1e-10 and rel(npy(c) @ npy(z), ocid.c @ ocid.z) <= 1e-9
    with pytest.raises(rc.CompressionError):   # src/qr.rs:196-199
        rc.QR.compute_from(np.eye(6, dtype=np.complex128)).compress(CT.ADAPTIVE(1e-3))
    with pytest.raises(AssertionError):
        rc.apply_permutation(a, np.arange(3), rc.MatrixPermutationMode.COL)


@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
def test_complex_fused_rsvd_id_batch_and_rank_by_tolerance(dtype):
    """rc_rsvd_id_c*, rc_batch_column_id_c*, rc_svd_rank_by_tolerance_c*: the fused call equals the separate complex calls (same
    Omega), the batch equals the one-matrix calls bit for bit, the rank rule is the real one."""
    import ctypes

    from rusty_compression_amd import _lib, batch

    t = TOLC[np.dtype(dtype)]
    suf = _lib.suffix(torch.complex128 if dtype == np.complex128 else torch.complex64)
    rng = np.random.default_rng(21)
    m, n, k, p = 260, 180, 20, 5
    a = torch.from_numpy(o.random_approximate_low_rank_matrix((m, n), 1.0, 1e-8 if dtype == np.complex128 else 1e-4, rng, dtype)).cuda()
    om = torch.from_numpy(o.random_gaussian((n, k + p), rng, dtype)).cuda()
    rdt = torch.float64 if dtype == np.complex128 else torch.float32
    e = lambda r, c: torch.empty((r, c), dtype=a.dtype, device="cuda")
    rq, u, vt, qq, qr_, c_, z_ = e(m, k), e(m, k), e(k, n), e(m, k), e(k, n), e(m, k), e(k, n)
    sv = torch.empty(k, dtype=rdt, device="cuda")
    ind = torch.empty(n, dtype=torch.int64, device="cuda")
    out = _lib.rc_rsvd_id_out(_lib.mat(rq), _lib.mat(u), ctypes.c_void_p(sv.data_ptr()), _lib.mat(vt), _lib.mat(qq), _lib.mat(qr_),
                              ctypes.c_void_p(ind.data_ptr()), _lib.mat(c_), _lib.mat(z_))
    ctx = _lib.default_context()
    ctx.call(f"rc_rsvd_id_{suf}", _lib.mat(a), ctypes.c_int64(k), ctypes.c_int64(p), _lib.mat(om), ctypes.c_uint64(0), ctypes.byref(out))
    q1 = rc.sample_range_by_rank(a, k, p, om)
    svd1 = rc.SVD.compute_from_range_estimate(q1, a)
    qr1 = rc.QR.compute_from_range_estimate(q1, a)
    cid1 = qr1.column_id()
    an = npy(a)
    assert rel(npy(rq), npy(q1)) <= t["factor"]
    assert np.abs(npy(sv) - npy(svd1.s)).max() <= t["sval"] * 10 * float(svd1.s[0])
    assert rel((npy(u) * npy(sv)) @ npy(vt), npy(svd1.to_mat())) <= t["factor"] * 10
    assert np.array_equal(npy(ind)[:k], npy(qr1.ind)[:k]) and rel(npy(qr_), npy(qr1.r)) <= t["factor"] * 10
    assert rel(npy(qq) @ npy(qr_), npy(qr1.q) @ npy(qr1.r)) <= t["factor"] * 10
    assert rel(npy(c_) @ npy(z_), npy(cid1.c) @ npy(cid1.z)) <= t["factor"] * 100
    assert abs(rel(npy(c_) @ npy(z_), an) - rel(npy(cid1.c) @ npy(cid1.z), an)) <= t["factor"] * 10
    # rank rule on the (real) singular values
    rk = ctypes.c_int64(-1)
    ctx.call(f"rc_svd_rank_by_tolerance_{suf}", ctypes.c_void_p(sv.data_ptr()), ctypes.c_int64(k), ctypes.c_double(0.5), ctypes.byref(rk))
    rk_real = ctypes.c_int64(-2)
    ctx.call(f"rc_svd_rank_by_tolerance_{'f64' if dtype == np.complex128 else 'f32'}", ctypes.c_void_p(sv.data_ptr()), ctypes.c_int64(k), ctypes.c_double(0.5), ctypes.byref(rk_real))
    assert rk.value == rk_real.value and 1 <= rk.value <= k
    # batch of three complex matrices on two contexts == the one-matrix calls
    mats = [torch.from_numpy(o.random_approximate_low_rank_matrix((96, 80), 1.0, 1e-6 if dtype == np.complex128 else 1e-3, rng, dtype)).cuda() for _ in range(3)]
    kk = 12
    es = mats[0].element_size()
    per = ((96 * kk + kk * 80) * es + 7) // 8 * 8 + 80 * 8
    packed = torch.empty(3 * per, dtype=torch.uint8, device="cuda")
    c2 = _lib.Context(0, None)
    try:
        ctxs = (ctypes.c_void_p * 2)(ctx._h, c2._h)
        marr = (_lib.rc_matrix * 3)(*[_lib.mat(x) for x in mats])
        fn = getattr(_lib.lib(), f"rc_batch_column_id_{suf}")
        fn.restype = ctypes.c_int32
        st = fn(ctxs, ctypes.c_int32(2), marr, ctypes.c_int32(3), ctypes.c_int64(kk), ctypes.c_void_p(packed.data_ptr()))
        assert st == 0, f"rc_batch_column_id_{suf} returned {st}"
    finally:
        c2.close()
    for i, x in enumerate(mats):
        blk = packed[i * per:(i + 1) * per]
        cc = blk[: 96 * kk * es].view(x.dtype).reshape(96, kk)
        zz = blk[96 * kk * es:(96 * kk + kk * 80) * es].view(x.dtype).reshape(kk, 80)
        ii = blk[per - 80 * 8:].view(torch.int64)
        c1, z1, i1 = batch.column_id_rank(x, kk)
        assert torch.equal(ii, i1) and torch.equal(cc, c1) and torch.equal(zz, z1)
