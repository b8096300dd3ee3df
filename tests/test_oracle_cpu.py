"""CPU tests: pin the oracle.

1. against the ONLY exact fixtures the reference holds for this path -- the
   known answers of its permutation tests (/root/reference/src/permutation.rs:192-239);
2. against the committed golden vectors (tests/golden/, generated from SciPy's
   LAPACK = the routines the reference calls, see tests/golden/make_golden.py);
3. the LAPACK-free C restatement (oracle/rc_oracle.c) against the same vectors;
4. the reference's own property tests (src/pivoted_qr.rs:193-317, src/qr.rs:418-616,
   src/svd.rs:193-321, src/col_interp_decomp.rs:163-242) restated on the oracle.
"""
import numpy as np
import pytest

from oracle import c_oracle, ref_lapack as o
from tests.helpers import TOL, golden, is_permutation, rel, sign_normalise, stable_prefix

QRCP_FILES = [f"qrcp_{t}_{s}_{g}.npz" for t in ("f64", "f32") for s in ("thin", "thick") for g in ("s5", "s10")]


def test_permutation_known_answers_of_the_reference():
    g = golden("perm_known.npz")
    for mode in ("COL", "COLINV", "ROW", "ROWINV"):
        assert np.array_equal(o.apply_permutation_matrix(g["mat"], g["perm"], mode), g[mode])
    assert np.array_equal(o.apply_permutation_vector(g["vec"], g["perm"], "NOINV"), g["NOINV"])
    assert np.array_equal(o.apply_permutation_vector(g["vec"], g["perm"], "INV"), g["INV"])
    assert np.array_equal(o.invert_permutation_vector(g["perm"]), np.array([1, 2, 0]))


def test_permutation_length_mismatch_asserts():
    with pytest.raises(AssertionError):
        o.apply_permutation_matrix(np.eye(3), np.array([0, 1]), "COL")
    with pytest.raises(AssertionError):
        o.apply_permutation_matrix(np.eye(3), np.array([0, 1]), "ROWINV")


@pytest.mark.parametrize("name", QRCP_FILES)
def test_lapack_oracle_reproduces_golden_qrcp(name):
    g = golden(name)
    a = g["a"]
    tol = TOL[a.dtype]
    q, r, ind = o.pivoted_qr(a)
    assert np.array_equal(ind, g["ind"])
    assert rel(r, g["r"]) <= tol["factor"]
    l, ql, indl = o.pivoted_lq(a)
    assert np.array_equal(indl, g["indl"])
    assert rel(l, g["l"]) <= tol["factor"]
    _, s, _ = o.compute_svd(a)
    assert np.abs(s - g["s"]).max() / g["s"][0] <= tol["sval"]


@pytest.mark.parametrize("name", QRCP_FILES)
def test_c_restatement_matches_golden_qrcp(name):
    """The LAPACK-free restatement picks the same pivots and signs as ?geqp3/?orgqr."""
    g = golden(name)
    a = g["a"]
    f64 = a.dtype == np.float64
    q, r, ind = c_oracle.pivoted_qr(a)
    ns = stable_prefix(g["r"], a.dtype)
    assert ns >= 20
    assert np.array_equal(ind[:ns], g["ind"][:ns]) and is_permutation(ind, a.shape[1])
    assert rel(r[:ns, :ns], g["r"][:ns, :ns]) <= (1e-13 if f64 else 5e-6)
    assert rel(o.apply_permutation_matrix(r[:ns], ind, "COLINV"), o.apply_permutation_matrix(g["r"][:ns], g["ind"], "COLINV")) <= (1e-13 if f64 else 5e-6)
    # Q's trailing columns are conditioned like 1/sigma_min: compare through the product
    assert rel(q @ r, a[:, ind]) <= (1e-14 if f64 else 5e-6)
    assert np.abs(q.T @ q - np.eye(q.shape[1])).max() <= (1e-13 if f64 else 5e-6)
    if name.endswith("s5.npz"):
        assert rel(q, g["q"]) <= (1e-9 if f64 else 5e-2)
    u, s, vt = c_oracle.svd_thin(a)
    assert np.abs(s - g["s"]).max() / g["s"][0] <= (1e-13 if f64 else 1e-5)
    assert rel(u @ np.diag(s) @ vt, a) <= (1e-13 if f64 else 1e-5)


def test_c_restatement_truncated_and_trtrs():
    rng = np.random.default_rng(5)
    a = o.random_approximate_low_rank_matrix((120, 80), 1.0, 1e-6, rng)
    q, r, ind = o.pivoted_qr(a)
    qt, rt, indt = c_oracle.pivoted_qr(a, kmax=25)
    assert np.array_equal(indt[:25], ind[:25]) and is_permutation(indt, 80)
    assert rel(qt, q[:, :25]) <= 1e-10
    # R rows agree once both are brought back to the original column order
    assert rel(o.apply_permutation_matrix(rt, indt, "COLINV"), o.apply_permutation_matrix(r[:25], ind, "COLINV")) <= 1e-12
    x = c_oracle.trtrs_upper(r[:25, :25], r[:25, 25:])
    assert rel(r[:25, :25] @ x, r[:25, 25:]) <= 1e-12


def test_golden_pipeline_is_reproducible():
    g = golden("cfg1_sketch_rsvd.npz")
    a, omega, k, p = g["a"], g["omega"], int(g["k"]), int(g["p"])
    q = o.sample_range_by_rank(a, k, p, lambda s: omega)
    assert rel(q, g["q_sample"]) <= 1e-10
    svd = o.SVD.compute_from_range_estimate(q, a)
    assert np.abs(svd.s - g["s"]).max() / g["s"][0] <= 1e-12
    u, vt = sign_normalise(svd.u, svd.vt)
    assert rel(u, g["u"]) <= 1e-8 and rel(vt, g["vt"]) <= 1e-8
    qr = o.QR.compute_from_range_estimate(q, a)
    assert np.array_equal(qr.ind, g["qr_ind"])
    # matvec-loop form (the reference's call shape) differs from the GEMM form by rounding only
    q2 = o.sample_range_by_rank(a, k, p, lambda s: omega, faithful=True)
    assert rel(q2, q) <= 1e-9


def test_golden_adaptive_is_reproducible():
    g = golden("adaptive_500x200.npz")
    omegas = g["omegas"]
    cnt = [0]

    def src(shape):
        blk = omegas[:, cnt[0] * 5:(cnt[0] + 1) * 5]
        cnt[0] += 1
        return blk

    q, res = o.sample_range_adaptive(g["a"], 1e-5, 5, src)
    assert [r for r, _ in res] == g["hist_rank"].tolist()
    assert np.allclose([e for _, e in res], g["hist_res"], rtol=1e-6)
    assert rel(q, g["q"]) <= 1e-8


# ---- the reference's property tests, restated on the oracle ------------------------
@pytest.mark.parametrize("dtype,shape", [(np.float64, (100, 50)), (np.float32, (100, 50)), (np.float64, (50, 100)), (np.float32, (50, 100))])
def test_reference_properties_hold_for_the_oracle(dtype, shape):
    rng = np.random.default_rng(11)
    a = o.random_approximate_low_rank_matrix(shape, 1.0, 1e-10, rng, dtype)
    # src/pivoted_qr.rs:225-242
    q, r, ind = o.pivoted_qr(a)
    assert np.abs(q.T @ q - np.eye(q.shape[1])).max() < 1e-6
    prod = q @ r
    for j in range(a.shape[1]):
        assert np.linalg.norm(prod[:, j] - a[:, ind[j]]) / np.linalg.norm(a[:, ind[j]]) < 1e-6 or np.linalg.norm(a[:, ind[j]]) < 1e-6
    # src/qr.rs:432-450 (RANK(30), 1e-4), :466-483 (ADAPTIVE(1e-4), 5 tol)
    qr = o.QR.compute_from(a)
    c = qr.compress("RANK", 30)
    assert c.q.shape[1] == 30 and c.r.shape[0] == 30 and o.rel_diff_fro(c.to_mat(), a) < 1e-4
    t = qr.compress("ADAPTIVE", 1e-4)
    assert o.rel_diff_fro(t.to_mat(), a) < 5e-4 and t.rank() < min(shape)
    # src/qr.rs:497-524 column ID, :538-564 row ID
    cid = t.column_id()
    assert o.rel_diff_fro(cid.to_mat(), a) < 5e-4
    ap = o.apply_permutation_matrix(a, cid.col_ind, "COL")
    for i in range(t.rank()):
        assert o.rel_diff_l2(ap[:, i], cid.c[:, i]) < 1e-4
    lq = o.LQ.compute_from(a).compress("ADAPTIVE", 1e-4)
    rid = lq.row_id()
    assert o.rel_diff_fro(rid.to_mat(), a) < 5e-4
    # src/col_interp_decomp.rs:198-223
    ts = cid.two_sided_id()
    assert o.rel_diff_fro(ts.to_mat(), a) < 5e-4
    mp = o.apply_permutation_matrix(o.apply_permutation_matrix(a, ts.row_ind, "ROW"), ts.col_ind, "COL")
    k = t.rank()
    assert ts.x.shape == (k, k)
    assert np.all(np.abs(ts.x - mp[:k, :k]) < 10 * 1e-4 * np.abs(mp[:k, :k]))
    # src/svd.rs:214-223, :246-253, :277-281
    svd = o.SVD.compute_from(a)
    assert o.rel_diff_fro(svd.to_qr().to_mat(), a) < (1e-12 if dtype == np.float64 else 1e-5)
    assert o.rel_diff_fro(svd.compress("RANK", 20).to_mat(), a) < 1e-4
    assert o.rel_diff_fro(svd.compress("ADAPTIVE", 1e-4).to_mat(), a) < 1e-4


def test_tolerance_not_reachable_is_a_compression_error():
    a = np.eye(6)
    with pytest.raises(o.CompressionError):
        o.QR.compute_from(a).compress("ADAPTIVE", 1e-3)
    with pytest.raises(AssertionError):
        o.QR.compute_from(a).compress("ADAPTIVE", 1.5)


# ---------------------------------------------------------------- Gaussian stream (a1): oracle/philox.py
def test_philox_restatement_reproduces_random123_known_answers_and_golden():
    """The integer generator behind rc_random_gaussian_* (ABI text: include/rusty_compression_amd.h, random_matrix.rs
    section), restated in numpy, against Random123's published philox4x32-10 vectors and the committed golden."""
    from oracle import philox as ph

    g = golden("philox_stream.npz")
    for ctr, key, want in zip(g["kat_ctr"], g["kat_key"], g["kat_out"]):
        assert np.array_equal(ph.philox4x32_10(ctr[None, :], tuple(int(x) for x in key))[0], want)
    for i, (seed, off) in enumerate(g["pairs"]):
        assert np.array_equal(ph.words(int(seed), int(off), 4096), g[f"words_{i}"])
        assert np.array_equal(ph.normals(int(seed), int(off), 4096), g[f"normals_{i}"])
    # stream continuity and row-major draw order of the matrix form
    a = ph.random_gaussian((10, 7), 7, 0)
    b = ph.random_gaussian((5, 7), 7, 70)
    assert np.array_equal(np.vstack([a, b]), ph.random_gaussian((15, 7), 7, 0))
    assert np.array_equal(ph.random_gaussian((10, 7), 7, 0, np.float32), a.astype(np.float32))


def test_philox_normals_are_normal():
    from scipy import stats

    from oracle import philox as ph

    z = ph.normals(11, 0, 400000)
    assert stats.kstest(z, "norm").pvalue > 1e-3
    # chi-square on 64 equiprobable bins
    edges = stats.norm.ppf(np.linspace(0, 1, 65)[1:-1])
    cnt = np.bincount(np.searchsorted(edges, z), minlength=64)
    assert stats.chisquare(cnt).pvalue > 1e-3
