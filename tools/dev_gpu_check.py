"""Developer smoke script (not a test): runs each C-ABI entry point once on the GPU and
prints its deviation from the CPU oracle. Usage: python tools/dev_gpu_check.py"""
import sys, os, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rusty_compression_amd as rc
from oracle import ref_lapack as o

def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

def npy(t): return t.detach().cpu().numpy()

def section(name, fn):
    t0 = time.time()
    try:
        fn()
        print(f"[ok ] {name} ({time.time()-t0:.2f}s)", flush=True)
    except Exception as e:
        print(f"[ERR] {name}: {type(e).__name__}: {e}", flush=True)
        traceback.print_exc()

rng = np.random.default_rng(0)

def t_gemm():
    for dt, tol in ((np.float64, 1e-13), (np.float32, 2e-6)):
        for (m, k, n) in ((100, 50, 37), (256, 512, 133), (128, 1000, 300), (33, 7, 5), (300, 260, 69), (1, 64, 1), (513, 129, 257)):
            a = rng.standard_normal((m, k)).astype(dt); b = rng.standard_normal((k, n)).astype(dt)
            ref = a.astype(np.float64) @ b.astype(np.float64)
            for la in ("C", "F"):
                for lb in ("C", "F"):
                    ta = torch.from_numpy(np.asarray(a, order=la)).cuda(); tb = torch.from_numpy(np.asarray(b, order=lb)).cuda()
                    if la == "F": ta = torch.from_numpy(np.ascontiguousarray(a.T)).cuda().t()
                    if lb == "F": tb = torch.from_numpy(np.ascontiguousarray(b.T)).cuda().t()
                    c = npy(rc.dot(ta, tb))
                    e = rel(c, ref)
                    flag = "" if e < tol else "  <-- BAD"
                    print(f"   gemm {dt.__name__} {m}x{k}x{n} A:{la} B:{lb} err {e:.2e}{flag}")
    a = rng.standard_normal((700, 300)); x = rng.standard_normal((700, 40))
    print("   conj_matmat", rel(npy(rc.conj_matmat(a, x)), a.T @ x))

def t_perm():
    mat = np.array([[1., 2, 3], [4, 5, 6], [7, 8, 9]]); perm = np.array([2, 0, 1])
    for mode in ("COL", "COLINV", "ROW", "ROWINV"):
        got = npy(rc.apply_permutation(mat, perm, rc.MatrixPermutationMode[mode]))
        print("   perm", mode, np.array_equal(got, o.apply_permutation_matrix(mat, perm, mode)))
    v = np.array([1., 2, 3])
    print("   vperm", npy(rc.apply_permutation(v, perm, rc.VectorPermutationMode.NOINV)), npy(rc.apply_permutation(v, perm, rc.VectorPermutationMode.INV)))

def t_gauss():
    g = npy(rc.random_gaussian((2000, 133), rc.Rng(7)))
    print("   gauss mean %.4f std %.4f" % (g.mean(), g.std()))
    r1 = rc.Rng(7); a = npy(rc.random_gaussian((10, 7), r1)); b = npy(rc.random_gaussian((5, 7), r1))
    full = npy(rc.random_gaussian((15, 7), rc.Rng(7)))
    print("   gauss stream consistency", np.array_equal(np.vstack([a, b]), full))

def t_qr():
    for dt, rt, qt in ((np.float64, 1e-12, 1e-9), (np.float32, 1e-4, 5e-2)):
        for shp in ((100, 50), (50, 100), (512, 69), (64, 512), (300, 300)):
            a = o.random_approximate_low_rank_matrix(shp, 1.0, 1e-5, rng, dt)
            q, r, ind = o.pivoted_qr(a)
            gq, gr, gi = rc.pivoted_qr(a)
            gq, gr, gi = npy(gq), npy(gr), npy(gi)
            same = np.array_equal(gi, ind)
            print(f"   qr {dt.__name__} {shp} ind_eq {same} firstdiff {np.nonzero(gi!=ind)[0][:1]} R {rel(gr, r):.2e} Q {rel(gq, q):.2e} orth {np.abs(gq.T@gq-np.eye(gq.shape[1])).max():.2e} recon {rel(gq@gr, a[:, gi]):.2e}")
        a = o.random_approximate_low_rank_matrix((200, 120), 1.0, 1e-5, rng, dt)
        l, q, ind = o.pivoted_lq(a); gl, gq, gi = rc.pivoted_lq(a)
        print(f"   lq {dt.__name__} ind_eq {np.array_equal(npy(gi), ind)} L {rel(npy(gl), l):.2e} Q {rel(npy(gq), q):.2e}")
        q, r, ind = o.pivoted_qr(a); gq, gr, gi = rc.pivoted_qr(a, rank=30)
        print(f"   qr trunc ind30_eq {np.array_equal(npy(gi)[:30], ind[:30])} perm_valid {sorted(npy(gi).tolist())==list(range(120))} R {rel(npy(gr), r[:30]):.2e} Q {rel(npy(gq), q[:, :30]):.2e}")

def t_svd():
    for dt in (np.float64, np.float32):
        for shp in ((100, 50), (50, 100), (128, 600), (77, 77)):
            a = o.random_approximate_low_rank_matrix(shp, 1.0, 1e-10, rng, dt)
            u, s, vt = o.compute_svd(a); gu, gs, gvt = (npy(t) for t in rc.compute_svd(a))
            print(f"   svd {dt.__name__} {shp} S {np.abs(gs-s).max()/s[0]:.2e} recon {rel(gu@np.diag(gs)@gvt, a):.2e} orthU {np.abs(gu.T@gu-np.eye(len(s))).max():.2e} orthV {np.abs(gvt@gvt.T-np.eye(len(s))).max():.2e}")

def t_id():
    for dt in (np.float64, np.float32):
        for shp in ((100, 50), (50, 100)):
            a = o.random_approximate_low_rank_matrix(shp, 1.0, 1e-10, rng, dt)
            oq = o.QR.compute_from(a).compress("ADAPTIVE", 1e-4); gq = rc.QR.compute_from(a).compress(rc.CompressionType.ADAPTIVE(1e-4))
            oc = oq.column_id(); gc = gq.column_id()
            ot = oc.two_sided_id(); gt = gc.two_sided_id()
            print(f"   id {dt.__name__} {shp} rank {oq.rank()} {gq.rank()} C {rel(npy(gc.c), oc.c):.2e} Z {rel(npy(gc.z), oc.z):.2e} tomat {rel(npy(gq.to_mat()), oq.to_mat()):.2e} ts.c {rel(npy(gt.c), ot.c):.2e} ts.x {rel(npy(gt.x), ot.x):.2e} rowind {np.array_equal(npy(gt.row_ind), ot.row_ind)} err {rc.rel_diff_fro(gt.to_mat(), a):.2e}")
            ol = o.LQ.compute_from(a).compress("ADAPTIVE", 1e-4); gl = rc.LQ.compute_from(a).compress(rc.CompressionType.ADAPTIVE(1e-4))
            orid = ol.row_id(); grid = gl.row_id(); ots = orid.two_sided_id(); gts = grid.two_sided_id()
            print(f"      rowid rank {ol.rank()} {gl.rank()} X {rel(npy(grid.x), orid.x):.2e} R {rel(npy(grid.r), orid.r):.2e} tomat {rel(npy(gl.to_mat()), ol.to_mat()):.2e} ts.x {rel(npy(gts.x), ots.x):.2e} ts.r {rel(npy(gts.r), ots.r):.2e} colind {np.array_equal(npy(gts.col_ind), ots.col_ind)}")
            osv = o.SVD.compute_from(a); gsv = rc.SVD.compute_from(a)
            print(f"      svd.to_qr {rc.rel_diff_fro(gsv.to_qr().to_mat(), a):.2e} svd rank20 {rc.rel_diff_fro(gsv.compress(rc.CompressionType.RANK(20)).to_mat(), a):.2e} adaptive rank {gsv.compress(rc.CompressionType.ADAPTIVE(1e-4)).rank()} vs {osv.compress('ADAPTIVE',1e-4).rank()}")

def t_sampling():
    a = o.random_approximate_low_rank_matrix((600, 400), 1.0, 1e-10, rng)
    om = rng.standard_normal((400, 37))
    oq = o.sample_range_by_rank(a, 32, 5, lambda s: om); gq = npy(rc.sample_range_by_rank(a, 32, 5, om))
    print(f"   sample_by_rank Q {rel(gq, oq):.2e}")
    oq = o.sample_range_power_iteration(a, 32, 5, 2, lambda s: om); gq = npy(rc.sample_range_power_iteration(a, 32, 5, 2, om))
    print(f"   power Q {rel(gq, oq):.2e}")
    rg = oq
    osv = o.SVD.compute_from_range_estimate(rg, a); gsv = rc.SVD.compute_from_range_estimate(rg, a)
    print(f"   svd_from_range S {np.abs(npy(gsv.s)-osv.s).max()/osv.s[0]:.2e} recon {rel(npy(gsv.to_mat()), osv.to_mat()):.2e}")
    oqr = o.QR.compute_from_range_estimate(rg, a); gqr = rc.QR.compute_from_range_estimate(rg, a)
    print(f"   qr_from_range ind {np.array_equal(npy(gqr.ind), oqr.ind)} R {rel(npy(gqr.r), oqr.r):.2e} Q {rel(npy(gqr.q), oqr.q):.2e}")
    b = o.random_approximate_low_rank_matrix((500, 200), 1.0, 1e-10, rng)
    oms = rng.standard_normal((200, 5 * 60)); cnt = [0]
    def src(shape):
        blk = oms[:, cnt[0] * 5:(cnt[0] + 1) * 5]; cnt[0] += 1; return blk
    oq, ores = o.sample_range_adaptive(b, 1e-5, 5, src)
    gq, gres = rc.sample_range_adaptive(b, 1e-5, 5, oms)
    print(f"   adaptive rank {oq.shape[1]} {gq.shape[1]} hist {len(ores)} {len(gres)} lastres {ores[-1]} {gres[-1]} Q {rel(npy(gq), oq) if gq.shape==oq.shape else 'shape'}")
    gq2, gres2 = rc.sample_range_adaptive(b, 1e-5, 5, rc.Rng(3))
    print(f"   adaptive(device rng) rank {gq2.shape[1]} err {rc.rel_diff_fro(rc.QR.compute_from_range_estimate(gq2, b).to_mat(), b):.2e}")
    print("   max_col_norm", rc.max_col_norm(b), o.max_col_norm(b))

def t_big():
    torch.manual_seed(0)
    for (n, k) in ((4096, 64), (8192, 128)):
        a = rc.random_gaussian((n, n), rc.Rng(n))
        om = rc.random_gaussian((n, k + 5), rc.Rng(1))
        torch.cuda.synchronize(); t0 = time.time()
        q = rc.sample_range_by_rank(a, k, 5, om)
        torch.cuda.synchronize(); t1 = time.time()
        svd = rc.SVD.compute_from_range_estimate(q, a)
        torch.cuda.synchronize(); t2 = time.time()
        qr = rc.QR.compute_from_range_estimate(q, a)
        cid = qr.column_id()
        torch.cuda.synchronize(); t3 = time.time()
        qq = q.cpu().numpy()
        print(f"   big {n} k={k}: sample {t1-t0:.3f}s svd {t2-t1:.3f}s qr+id {t3-t2:.3f}s orth {np.abs(qq.T@qq-np.eye(k)).max():.2e}")
        # compare against oracle at this size (GEMM form)
        an = a.cpu().numpy(); omn = om.cpu().numpy()
        t0 = time.time(); oq = o.sample_range_by_rank(an, k, 5, lambda s: omn); t1 = time.time()
        print(f"      oracle sample {t1-t0:.2f}s Q {rel(qq, oq):.2e}")
        osv = o.SVD.compute_from_range_estimate(oq, an)
        print(f"      S {np.abs(npy(svd.s)-osv.s).max()/osv.s[0]:.2e} USVt {rel(npy(svd.to_mat()), osv.to_mat()):.2e}")

which = sys.argv[1:] or ["gemm", "perm", "gauss", "qr", "svd", "id", "sampling", "big"]
for name in which:
    section(name, globals()["t_" + name])
