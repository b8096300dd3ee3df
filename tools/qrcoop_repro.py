"""Developer tool: pivoted QR through the blocked path, cooperative panels on / off (two processes), outputs compared."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def child():
    import torch
    import rusty_compression_amd as rc
    m, n = int(os.environ.get("M", 300)), int(os.environ.get("N", 700))
    dt = torch.float64 if os.environ.get("DT", "f64") == "f64" else torch.float32
    g1 = rc.random_gaussian((m, m), rc.Rng(41)).to(dt)
    g2 = rc.random_gaussian((m, n), rc.Rng(42)).to(dt)
    sig = torch.logspace(0, -6, m, dtype=dt, device="cuda")
    b = rc.dot(g1, sig[:, None] * g2)
    qr = rc.QR.compute_from(b)
    torch.cuda.synchronize()
    err = rc.rel_diff_fro(qr.to_mat(), b)
    np.savez(os.environ["OUT"], q=qr.q.cpu().numpy(), r=qr.r.cpu().numpy(), ind=qr.ind.cpu().numpy(), b=b.cpu().numpy())
    print("coop", os.environ.get("RC_QRCP_COOP"), "err", err, flush=True)

if os.environ.get("OUT"):
    child()
else:
    for coop in ("0", "1"):
        env = dict(os.environ, RC_QRCP_COOP=coop, OUT=f"/tmp/qrc{coop}.npz")
        subprocess.run([sys.executable, __file__], env=env, check=False)
    a, b = np.load("/tmp/qrc0.npz"), np.load("/tmp/qrc1.npz")
    print("ind equal:", np.array_equal(a["ind"], b["ind"]), "first diff", (np.nonzero(a["ind"] != b["ind"])[0][:5]))
    d = np.abs(a["r"] - b["r"])
    k = a["r"].shape[0]
    rowerr = d.max(axis=1)
    print("R row errors (first 40):", np.array2string(rowerr[:40], precision=2))
    print("R max diff", d.max(), "at", np.unravel_index(d.argmax(), d.shape))
    print("diag0", np.diag(a["r"])[:8], "\ndiag1", np.diag(b["r"])[:8])
    i0, i1 = a["ind"], b["ind"]
    print("perm valid:", len(set(i0.tolist())) == len(i0), len(set(i1.tolist())) == len(i1), "dups in coop:", len(i1) - len(set(i1.tolist())))
    d0, d1 = np.abs(np.diag(a["r"])), np.abs(np.diag(b["r"]))
    bad = np.nonzero(np.abs(d0 - d1) > 1e-8 * d0.max())[0]
    print("diag first mismatch:", bad[:5], "of", len(d0))
    print("ind0[40:60]", i0[40:60]); print("ind1[40:60]", i1[40:60])
    # reconstruct check of each: || B P - Q R ||
    for nm, z in (("classic", a), ("coop", b)):
        bp = z["b"][:, z["ind"]]
        res = bp - z["q"] @ z["r"]
        colerr = np.linalg.norm(res, axis=0)
        worst = np.argsort(-colerr)[:8]
        print(nm, "resid", np.linalg.norm(res) / np.linalg.norm(bp), "worst permuted cols", worst, colerr[worst])
    # which columns (in permuted order) are wrong in row 0..3
    for row in range(3):
        bad = np.nonzero(d[row] > 1e-10 * np.abs(a["r"]).max())[0]
        print("row", row, "bad cols:", len(bad), bad[:10])
