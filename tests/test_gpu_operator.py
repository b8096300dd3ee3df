"""GPU tests of the operator-callback entry points (rc_operator, rc_*_op_*; include/rusty_compression_amd.h).

The reference's range finders and `compute_from_range_estimate` are generic over the operator
(`impl<Op: MatMat<A = $scalar>> SampleRange for Op`, /root/reference/src/random_sampling.rs:102, :130, :222;
src/qr.rs:311-323, src/svd.rs:171-183; trait contract src/types.rs:40-101).  Two kinds of check:

  (a) a DENSE matrix behind callbacks (each product = the library's own rc_matmat / rc_conj_matmat on the views the
      library hands over) reproduces the dense entry point BIT FOR BIT in f64 (to rounding in f32: the f32 products
      pick their tiles by operand layout);
  (b) a factored operator U V^T that is never materialised on the device against the SciPy-LAPACK oracle given the
      same operator (the oracle is handed the product U V^T formed on the host).
The reference has no tests of its range finders: parity here is pinned by the oracle only.
"""
import ctypes

import numpy as np
import pytest
import torch

import rusty_compression_amd as rc
from oracle import ref_lapack as o
from rusty_compression_amd import _lib
from rusty_compression_amd.operator import DenseOperator, LowRankOperator, Operator, OperatorTable
from tests.helpers import TOL, npy, rel

pytestmark = pytest.mark.gpu


def _recipe(m, n, dtype, seed=3, smin=1e-8):
    rng = np.random.default_rng(seed)
    return o.random_approximate_low_rank_matrix((m, n), 1.0, smin, rng).astype(dtype)


def _same(x, y, dtype):
    if dtype == np.float64:
        return torch.equal(x, y)
    return rel(npy(x), npy(y)) <= 5e-5


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_dense_matrix_behind_callbacks_reproduces_the_dense_entry_points(dtype):
    m, n, k, p = 700, 520, 40, 6
    a = torch.from_numpy(_recipe(m, n, dtype)).cuda()
    op = DenseOperator(a)
    rng = np.random.default_rng(11)
    omega = rng.standard_normal((n, k + p)).astype(dtype)

    q_d = rc.sample_range_by_rank(a, k, p, omega)
    q_o = rc.sample_range_by_rank(op, k, p, omega)
    assert op.calls == {"matmat": 1, "conj_matmat": 0}
    assert _same(q_d, q_o, dtype), "sample_range_by_rank"

    qp_d = rc.sample_range_power_iteration(a, k, p, 2, omega)
    qp_o = rc.sample_range_power_iteration(op, k, p, 2, omega)
    assert _same(qp_d, qp_o, dtype), "sample_range_power_iteration"
    assert op.calls["conj_matmat"] == 1 and op.calls["matmat"] == 3   # one surviving power step: A Omega, A^H Q0, A W

    s_d = rc.SVD.compute_from_range_estimate(q_d, a)
    s_o = rc.SVD.compute_from_range_estimate(q_d, op)
    assert _same(s_d.s, s_o.s, dtype) and _same(s_d.u, s_o.u, dtype) and _same(s_d.vt, s_o.vt, dtype), "svd_from_range_estimate"
    r_d = rc.QR.compute_from_range_estimate(q_d, a)
    r_o = rc.QR.compute_from_range_estimate(q_d, op)
    assert torch.equal(r_d.ind, r_o.ind) and _same(r_d.r, r_o.r, dtype) and _same(r_d.q, r_o.q, dtype), "qr_from_range_estimate"

    omegas = rng.standard_normal((n, 8 * 80)).astype(dtype)
    tol = 1e-5 if dtype == np.float64 else 1e-3
    qa_d, h_d = rc.sample_range_adaptive(a, tol, 8, omegas)
    qa_o, h_o = rc.sample_range_adaptive(op, tol, 8, omegas)
    assert [r for r, _ in h_d] == [r for r, _ in h_o] and qa_d.shape == qa_o.shape
    if dtype == np.float64:
        assert h_d == h_o and torch.equal(qa_d, qa_o), "sample_range_adaptive"
    else:
        assert rel(npy(qa_o) @ (npy(qa_o).T @ npy(a)), npy(qa_d) @ (npy(qa_d).T @ npy(a))) <= 1e-4


def test_fused_rsvd_id_over_an_operator_equals_the_dense_fused_call():
    m, n, k, p = 1024, 768, 32, 5
    a = torch.from_numpy(_recipe(m, n, np.float64, seed=5)).cuda()
    omega = torch.from_numpy(np.random.default_rng(2).standard_normal((n, k + p))).cuda()

    def outputs():
        mk = lambda r, c: torch.empty((r, c), dtype=torch.float64, device="cuda")  # noqa: E731
        b = dict(range_q=mk(m, k), u=mk(m, k), s=torch.empty(k, dtype=torch.float64, device="cuda"), vt=mk(k, n), qr_q=mk(m, k), qr_r=mk(k, n),
                 qr_ind=torch.empty(n, dtype=torch.int64, device="cuda"), id_c=mk(m, k), id_z=mk(k, n))
        out = _lib.rc_rsvd_id_out(_lib.mat(b["range_q"]), _lib.mat(b["u"]), ctypes.c_void_p(b["s"].data_ptr()), _lib.mat(b["vt"]), _lib.mat(b["qr_q"]),
                                  _lib.mat(b["qr_r"]), ctypes.c_void_p(b["qr_ind"].data_ptr()), _lib.mat(b["id_c"]), _lib.mat(b["id_z"]))
        return b, out

    ctx = _lib.default_context()
    bd, od = outputs()
    ctx.call("rc_rsvd_id_f64", _lib.mat(a), ctypes.c_int64(k), ctypes.c_int64(p), _lib.mat(omega), ctypes.c_uint64(0), ctypes.byref(od))
    bo, oo = outputs()
    tab = OperatorTable(DenseOperator(a))
    tab.call(ctx, "rc_rsvd_id_op_f64", tab.byref(), ctypes.c_int64(k), ctypes.c_int64(p), _lib.mat(omega), ctypes.c_uint64(0), ctypes.byref(oo))
    ctx.synchronize()
    for name in bd:
        assert torch.equal(bd[name], bo[name]), name
    # ... and the fused call agrees with the oracle on the factors it returns (same checks as __graft_entry__.smoke)
    an, om = npy(a), npy(omega)
    oq = o.sample_range_by_rank(an, k, p, lambda s: om)
    osvd = o.SVD.compute_from_range_estimate(oq, an)
    oqr = o.QR.compute_from_range_estimate(oq, an)
    assert rel(npy(bo["range_q"]), oq) <= TOL[np.dtype(np.float64)]["factor"]
    assert np.abs(npy(bo["s"]) - osvd.s).max() / osvd.s[0] <= TOL[np.dtype(np.float64)]["sval"]
    assert np.array_equal(npy(bo["qr_ind"])[:k], oqr.ind[:k]) and rel(npy(bo["qr_r"]), oqr.r) <= TOL[np.dtype(np.float64)]["factor"]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_factored_operator_never_materialised_matches_the_oracle(dtype):
    """A = U diag(sigma) V^T held as two skinny factors (LowRankOperator: every product is two skinny GEMMs); the oracle gets the
    same operator as the host product of the factors."""
    m, n, r, k, p = 2048, 1536, 60, 24, 8
    rng = np.random.default_rng(21)
    u = np.linalg.qr(rng.standard_normal((m, r)))[0] * np.geomspace(1.0, 1e-6 if dtype == np.float64 else 1e-3, r)
    v = np.linalg.qr(rng.standard_normal((n, r)))[0]
    u, v = u.astype(dtype), v.astype(dtype)
    a_host = (u.astype(np.float64) @ v.astype(np.float64).T).astype(dtype)   # for the oracle only
    op = LowRankOperator(u, v)
    assert op.shape == (m, n)
    omega = rng.standard_normal((n, k + p)).astype(dtype)
    tol = TOL[np.dtype(dtype)]

    q = rc.sample_range_by_rank(op, k, p, omega)
    oq = o.sample_range_by_rank(a_host, k, p, lambda s: omega)
    assert rel(npy(q), oq) <= tol["factor"] * 10, rel(npy(q), oq)

    svd = rc.SVD.compute_from_range_estimate(q, op)
    osvd = o.SVD.compute_from_range_estimate(npy(q), a_host)
    assert np.abs(npy(svd.s) - osvd.s).max() / osvd.s[0] <= tol["sval"] * 10
    assert rel(npy(svd.to_mat()), osvd.to_mat()) <= tol["factor"] * 10

    qr = rc.QR.compute_from_range_estimate(q, op)
    oqr = o.QR.compute_from_range_estimate(npy(q), a_host)
    assert np.array_equal(npy(qr.ind)[:8], oqr.ind[:8])   # leading pivots are determined by the data (steep spectrum)
    assert rel(npy(qr.to_mat()), oqr.to_mat()) <= tol["factor"] * 10
    cid = qr.column_id()
    assert rel(npy(cid.to_mat()), oqr.column_id().to_mat()) <= tol["factor"] * 100

    # adaptive sampling of the same operator: same rank history as the oracle, same projector
    s = 10
    omegas = rng.standard_normal((n, s * 12)).astype(dtype)
    blocks = iter(range(12))
    rel_tol = 1e-4 if dtype == np.float64 else 1e-2
    qa, hist = rc.sample_range_adaptive(op, rel_tol, s, omegas)
    oqa, ohist = o.sample_range_adaptive(a_host, rel_tol, s, lambda shape: omegas[:, (i := next(blocks)) * s:(i + 1) * s])
    assert [x for x, _ in hist] == [x for x, _ in ohist]
    pa, opa = npy(qa) @ (npy(qa).T.astype(np.float64) @ a_host), oqa @ (oqa.T.astype(np.float64) @ a_host)
    assert rel(pa, opa) <= (1e-6 if dtype == np.float64 else 5e-3)


class _MatMatOnly(Operator):
    """An operator that is only `MatMat` (no conj_matmat): enough for SampleRange (src/random_sampling.rs:102), not for the others."""

    conj_matmat = None

    def __init__(self, a):
        self.a = a
        self.dtype = a.dtype

    def nrows(self):
        return self.a.shape[0]

    def ncols(self):
        return self.a.shape[1]

    def matmat(self, x):
        return rc.matmat(self.a, x)


def test_operator_with_matmat_only_and_error_propagation():
    a = torch.from_numpy(_recipe(300, 200, np.float64)).cuda()
    omega = np.random.default_rng(0).standard_normal((200, 20))
    op = _MatMatOnly(a)
    q = rc.sample_range_by_rank(op, 15, 5, omega)
    assert rel(npy(q), npy(rc.sample_range_by_rank(a, 15, 5, omega))) <= 1e-12   # (the torch-level product is copied into y: same values)
    with pytest.raises(AssertionError, match="conj_matmat"):   # RC_INVALID_ARGUMENT: the reference would not compile (trait bound)
        rc.QR.compute_from_range_estimate(q, op)

    class Boom(_MatMatOnly):
        def matmat(self, x):
            raise ValueError("the host's operator failed")

    with pytest.raises(ValueError, match="the host's operator failed"):
        rc.sample_range_by_rank(Boom(a), 15, 5, omega)

    # a callback that returns a status: the entry point returns that status and names the product
    tab = OperatorTable(DenseOperator(a))
    status_fn = type(tab.table.matmat)(lambda user, ctx, x, y: _lib.RC_LINALG_ERROR)
    tab.table.matmat = status_fn
    ctx = _lib.default_context()
    qbuf = torch.empty((300, 15), dtype=torch.float64, device="cuda")
    st = _lib.lib().rc_sample_range_by_rank_op_f64(ctx._h, tab.byref(), ctypes.c_int64(15), ctypes.c_int64(5), _lib.mat(torch.from_numpy(omega).cuda()),
                                                   ctypes.c_uint64(0), _lib.mat(qbuf))
    assert st == _lib.RC_LINALG_ERROR
    assert b"matmat" in _lib.lib().rc_last_error_message(ctx._h)
    # the context is usable afterwards (the nested-call depth unwound)
    assert rel(npy(rc.sample_range_by_rank(a, 15, 5, omega)), npy(q)) <= 1e-12


def test_operator_callbacks_are_rejected_inside_a_graph_capture():
    a = torch.from_numpy(_recipe(256, 192, np.float64)).cuda()
    st_ = torch.cuda.Stream()
    with torch.cuda.stream(st_):
        ctx = _lib.Context(torch.cuda.current_device(), st_.cuda_stream)
        tab = OperatorTable(DenseOperator(a))
        omega = torch.from_numpy(np.random.default_rng(0).standard_normal((192, 16))).cuda()
        q = torch.empty((256, 12), dtype=torch.float64, device="cuda")
        args = (tab.byref(), ctypes.c_int64(12), ctypes.c_int64(4), _lib.mat(omega), ctypes.c_uint64(0), _lib.mat(q))
        ctx.call("rc_sample_range_by_rank_op_f64", *args)   # eager: fine
        ctx.synchronize()
        ctx.check(_lib.lib().rc_graph_begin_capture(ctx._h))
        st = _lib.lib().rc_sample_range_by_rank_op_f64(ctx._h, *args)
        graph = ctypes.c_void_p(None)
        _lib.lib().rc_graph_end_capture(ctx._h, ctypes.byref(graph))
        if graph.value:
            _lib.lib().rc_graph_destroy(ctx._h, graph)
        assert st == _lib.RC_INVALID_ARGUMENT
        ctx.close()


def test_get_stream_returns_the_context_stream():
    s = torch.cuda.Stream()
    ctx = _lib.Context(torch.cuda.current_device(), s.cuda_stream)
    out = ctypes.c_void_p()
    assert _lib.lib().rc_get_stream(ctx._h, ctypes.byref(out)) == 0 and (out.value or 0) == s.cuda_stream
    ctx.close()


@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
def test_complex_operators_behind_callbacks(dtype):
    """rc_*_op_c64 / _c32 (the reference instantiates its operator-generic range finders for c32 / c64 too: src/random_sampling.rs:123-126,
    :165-168, :277-280).  A dense complex matrix behind callbacks against the dense entry points (to rounding: the projection
    B = Q^H A is the conjugate transpose of the callback's A^H Q, another product than the dense path's), and a factored operator
    U V^H never materialised against the oracle on the host product."""
    real = np.float64 if dtype == np.complex128 else np.float32
    tol = TOL[np.dtype(real)]
    rng = np.random.default_rng(8)
    m, n, r, k, p = 900, 640, 40, 20, 6
    u = np.linalg.qr(rng.standard_normal((m, r)) + 1j * rng.standard_normal((m, r)))[0] * np.geomspace(1.0, 1e-6 if real == np.float64 else 1e-3, r)
    v = np.linalg.qr(rng.standard_normal((n, r)) + 1j * rng.standard_normal((n, r)))[0]
    u, v = u.astype(dtype), v.astype(dtype)
    a_host = (u.astype(np.complex128) @ v.astype(np.complex128).conj().T).astype(dtype)
    a = torch.from_numpy(a_host).cuda()
    omega = (rng.standard_normal((n, k + p)) + 1j * rng.standard_normal((n, k + p))).astype(dtype)

    dense_op, lr_op = DenseOperator(a), LowRankOperator(u, v)
    q_d = rc.sample_range_by_rank(a, k, p, omega)
    q_o = rc.sample_range_by_rank(dense_op, k, p, omega)
    assert torch.equal(q_d, q_o)                      # the sketch is the same product on the same views
    q_l = rc.sample_range_by_rank(lr_op, k, p, omega)
    oq = o.sample_range_by_rank(a_host, k, p, lambda s: omega)
    assert rel(npy(q_l) @ (npy(q_l).conj().T @ a_host), oq @ (oq.conj().T @ a_host)) <= 100 * tol["factor"]

    s_d = rc.SVD.compute_from_range_estimate(q_d, a)
    s_o = rc.SVD.compute_from_range_estimate(q_d, dense_op)
    s_l = rc.SVD.compute_from_range_estimate(q_d, lr_op)
    osvd = o.SVD.compute_from_range_estimate(npy(q_d), a_host)
    for sv in (s_d, s_o, s_l):
        assert np.abs(npy(sv.s) - osvd.s).max() / osvd.s[0] <= 10 * tol["sval"]
        assert rel(npy(sv.to_mat()), osvd.to_mat()) <= 10 * tol["factor"]
    r_o = rc.QR.compute_from_range_estimate(q_d, dense_op)
    r_l = rc.QR.compute_from_range_estimate(q_d, lr_op)
    oqr = o.QR.compute_from_range_estimate(npy(q_d), a_host)
    for qr in (r_o, r_l):
        assert np.array_equal(npy(qr.ind)[:6], oqr.ind[:6])
        assert rel(npy(qr.to_mat()), oqr.to_mat()) <= 10 * tol["factor"]
    qp = rc.sample_range_power_iteration(lr_op, k, p, 1, omega)
    oqp = o.sample_range_power_iteration(a_host, k, p, 1, lambda s: omega)
    assert rel(npy(qp) @ (npy(qp).conj().T @ a_host), oqp @ (oqp.conj().T @ a_host)) <= 100 * tol["factor"]
    s_ = 8
    omegas = (rng.standard_normal((n, s_ * 12)) + 1j * rng.standard_normal((n, s_ * 12))).astype(dtype)
    rel_tol = 1e-4 if real == np.float64 else 1e-2
    qa, hist = rc.sample_range_adaptive(lr_op, rel_tol, s_, omegas)
    blocks = iter(range(12))
    oqa, ohist = o.sample_range_adaptive(a_host, rel_tol, s_, lambda shape: omegas[:, (i := next(blocks)) * s_:(i + 1) * s_])
    assert [x for x, _ in hist] == [x for x, _ in ohist]
