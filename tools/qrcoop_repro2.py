"""Developer tool: truncated pivoted QR (one or two panels) through the blocked path, cooperative panels on / off."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def child():
    import torch
    import rusty_compression_amd as rc
    from rusty_compression_amd.qr import pivoted_qr
    m, n, k = int(os.environ.get("M", 200)), int(os.environ.get("N", 520)), int(os.environ.get("K", 10))
    g1 = rc.random_gaussian((m, m), rc.Rng(41)); g2 = rc.random_gaussian((m, n), rc.Rng(42))
    sig = torch.logspace(0, -6, m, dtype=torch.float64, device="cuda")
    b = rc.dot(g1, sig[:, None] * g2)
    q, r, ind = pivoted_qr(b, k)
    torch.cuda.synchronize()
    np.savez(os.environ["OUT"], q=q.cpu().numpy(), r=r.cpu().numpy(), ind=ind.cpu().numpy(), b=b.cpu().numpy())

if os.environ.get("OUT"):
    child()
else:
    for coop in ("0", "1"):
        env = dict(os.environ, RC_QRCP_COOP=coop, OUT=f"/tmp/qrd{coop}.npz", RC_QRCP_DEBUG="1")
        subprocess.run([sys.executable, __file__], env=env, check=False)
    a, b = np.load("/tmp/qrd0.npz"), np.load("/tmp/qrd1.npz")
    k = a["r"].shape[0]
    print("shapes", a["q"].shape, a["r"].shape, "ind[:k] equal", np.array_equal(a["ind"][:k], b["ind"][:k]))
    print("perm valid", len(set(a["ind"].tolist())) == len(a["ind"]), len(set(b["ind"].tolist())) == len(b["ind"]))
    dif = np.nonzero(a["ind"] != b["ind"])[0]
    print("ind differs at", dif[:10], "classic", a["ind"][dif[:10]], "coop", b["ind"][dif[:10]])
    for nm, z in (("classic", a), ("coop", b)):
        bp = z["b"][:, z["ind"]]
        # R = Q^T B P for the first k rows
        rr = z["q"].T @ bp
        d = np.abs(rr - z["r"])
        colerr = d.max(axis=0)
        bad = np.nonzero(colerr > 1e-9 * np.abs(z["r"]).max())[0]
        print(nm, "max |Q^T B P - R|", d.max(), "bad permuted cols", len(bad), bad[:12])
        if len(bad):
            norms = np.linalg.norm(z["b"], axis=0)
            order = np.argsort(-norms)
            rank_of = np.empty_like(order); rank_of[order] = np.arange(len(order))
            phys = z["ind"][bad]
            print("   norm-ranks of the bad columns (0 = largest):", np.sort(rank_of[phys])[:12], "...", np.sort(rank_of[phys])[-5:])
            print("   rows with error for first bad col:", np.nonzero(d[:, bad[0]] > 1e-9)[0])
            print("   R col", z["r"][:, bad[0]][40:60])
            print("   QtBP col", rr[:, bad[0]][40:60])
