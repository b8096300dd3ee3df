"""Generates the golden vectors under tests/golden/ from the CPU oracle
(oracle/ref_lapack.py = SciPy's LAPACK, the routines the reference itself calls).

Run in the build container:  python tests/golden/make_golden.py
The fixtures are DATA (seeded inputs + expected outputs).  The only values taken
from the reference are the known answers of its permutation tests
(/root/reference/src/permutation.rs:192-239), stored as numbers in perm_known.npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_lapack as o  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def sign_normalise(u, vt):
    """Fix the per-pair sign ambiguity of an SVD: largest-|.| entry of each u_j positive."""
    u = u.copy(); vt = vt.copy()
    for j in range(u.shape[1]):
        i = int(np.argmax(np.abs(u[:, j])))
        if u[i, j] < 0:
            u[:, j] *= -1
            vt[j, :] *= -1
    return u, vt


def main():
    # 1. permutation known answers (values of the reference's own tests)
    mat = np.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0], [7.0, 8.0, 9.0]])
    save(
        "perm_known.npz",
        mat=mat,
        perm=np.array([2, 0, 1], dtype=np.int64),
        COL=np.array([[3.0, 1.0, 2.0], [6.0, 4.0, 5.0], [9.0, 7.0, 8.0]]),
        COLINV=np.array([[2.0, 3.0, 1.0], [5.0, 6.0, 4.0], [8.0, 9.0, 7.0]]),
        ROW=np.array([[7.0, 8.0, 9.0], [1.0, 2.0, 3.0], [4.0, 5.0, 6.0]]),
        ROWINV=np.array([[4.0, 5.0, 6.0], [7.0, 8.0, 9.0], [1.0, 2.0, 3.0]]),
        vec=np.array([1.0, 2.0, 3.0]),
        NOINV=np.array([3.0, 1.0, 2.0]),
        INV=np.array([2.0, 3.0, 1.0]),
    )

    # 2. pivoted QR / LQ, thin and thick, f64 and f32 (reference test shapes, src/pivoted_qr.rs:296-315)
    for dt, tag in ((np.float64, "f64"), (np.float32, "f32")):
        for shp, shape_tag in (((100, 50), "thin"), ((50, 100), "thick")):
            for smin, stag in ((1e-5, "s5"), (1e-10, "s10")):
                seed = {"f64": 1, "f32": 2}[tag] * 100 + {"thin": 1, "thick": 2}[shape_tag] * 10 + {"s5": 1, "s10": 2}[stag]
                rng = np.random.default_rng(seed)
                a = o.random_approximate_low_rank_matrix(shp, 1.0, smin, rng, dt)
                q, r, ind = o.pivoted_qr(a)
                l, ql, indl = o.pivoted_lq(a)
                u, s, vt = o.compute_svd(a)
                un, vtn = sign_normalise(u, vt)
                save(f"qrcp_{tag}_{shape_tag}_{stag}.npz", a=a, q=q, r=r, ind=ind, l=l, ql=ql, indl=indl, s=s, u=un, vt=vtn)

    # 3. cfg1-sized pipeline (512 x 256 f64, k = 32, p = 5), explicit Omega
    rng = np.random.default_rng(31)
    a = o.random_approximate_low_rank_matrix((512, 256), 1.0, 1e-10, rng)
    k, p = 32, 5
    omega = rng.standard_normal((256, k + p))
    qs = o.sample_range_by_rank(a, k, p, lambda s: omega)
    qp = o.sample_range_power_iteration(a, k, p, 2, lambda s: omega)
    svd = o.SVD.compute_from_range_estimate(qs, a)
    un, vtn = sign_normalise(svd.u, svd.vt)
    qr = o.QR.compute_from_range_estimate(qs, a)
    cid = qr.column_id()
    save("cfg1_sketch_rsvd.npz", a=a, omega=omega, k=np.int64(k), p=np.int64(p), q_sample=qs, q_power=qp, s=svd.s, u=un, vt=vtn,
         qr_q=qr.q, qr_r=qr.r, qr_ind=qr.ind, id_c=cid.c, id_z=cid.z)

    # 4. interpolative decompositions from the full matrix (examples/interpolative_decomposition.rs sequence)
    full = o.QR.compute_from(a)
    out = dict(a_seed=np.int64(31))
    for kind, val, tag in (("RANK", 32, "rank32"), ("ADAPTIVE", 1e-4, "tol1e4")):
        qrc = full.compress(kind, val)
        cid = qrc.column_id()
        ts = cid.two_sided_id()
        lqc = o.LQ.compute_from(a).compress(kind, val)
        rid = lqc.row_id()
        ts2 = rid.two_sided_id()
        out.update({
            f"{tag}_rank": np.int64(qrc.rank()), f"{tag}_ind": qrc.ind, f"{tag}_c": cid.c, f"{tag}_z": cid.z,
            f"{tag}_ts_c": ts.c, f"{tag}_ts_x": ts.x, f"{tag}_ts_r": ts.r, f"{tag}_ts_row_ind": ts.row_ind, f"{tag}_ts_col_ind": ts.col_ind,
            f"{tag}_lq_rank": np.int64(lqc.rank()), f"{tag}_lq_ind": lqc.ind, f"{tag}_rid_x": rid.x, f"{tag}_rid_r": rid.r,
            f"{tag}_ts2_c": ts2.c, f"{tag}_ts2_x": ts2.x, f"{tag}_ts2_r": ts2.r, f"{tag}_ts2_row_ind": ts2.row_ind, f"{tag}_ts2_col_ind": ts2.col_ind,
        })
    save("cfg1_id.npz", **out)

    # 5. adaptive sampling (examples/adaptive_sampling.rs: 500 x 200, tol 1e-5, sample_size 5), explicit Omega blocks
    rng = np.random.default_rng(52)
    b = o.random_approximate_low_rank_matrix((500, 200), 1.0, 1e-10, rng)
    s = 5
    omegas = rng.standard_normal((200, s * 48))
    cnt = [0]

    def src(shape):
        blk = omegas[:, cnt[0] * s:(cnt[0] + 1) * s]
        cnt[0] += 1
        return blk

    q, res = o.sample_range_adaptive(b, 1e-5, s, src)
    qr = o.QR.compute_from_range_estimate(q, b)
    save("adaptive_500x200.npz", a=b, omegas=omegas[:, : s * cnt[0]], q=q, hist_rank=np.array([r for r, _ in res], dtype=np.int64),
         hist_res=np.array([e for _, e in res]), rel_err=np.float64(o.rel_diff_fro(qr.to_mat(), b)))


if __name__ == "__main__":
    main()
