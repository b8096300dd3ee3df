"""Shared helpers of the parity tests."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    wide = np.complex128 if (np.iscomplexobj(a) or np.iscomplexobj(b)) else np.float64
    a = a.astype(wide)
    b = b.astype(wide)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def npy(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def sign_normalise(u, vt):
    """Largest-|.| entry of each left singular vector positive (per-pair sign ambiguity of an SVD)."""
    u = np.array(u, copy=True)
    vt = np.array(vt, copy=True)
    for j in range(u.shape[1]):
        i = int(np.argmax(np.abs(u[:, j])))
        if u[i, j] < 0:
            u[:, j] *= -1
            vt[j, :] *= -1
    return u, vt


def is_permutation(ind, n):
    return sorted(np.asarray(ind).tolist()) == list(range(n))


# tolerances (stated once, used everywhere)
#   f64: factors <= 1e-10 relative Frobenius (BASELINE.json north_star), singular values <= 1e-12 relative
#   f32: the reference's own f32 thresholds (1e-4 factors, 1e-5 singular values / round trips)
TOL = {
    np.dtype(np.float64): dict(factor=1e-10, sval=1e-12, orth=1e-12, recon=1e-12),
    np.dtype(np.float32): dict(factor=1e-4, sval=1e-5, orth=1e-5, recon=1e-5),
}


def stable_prefix(r, dtype):
    """Number of leading pivots that are determined by the data rather than by rounding.

    Column-pivoted QR picks the largest remaining partial norm; once |r_jj| has
    dropped to the rounding level of the working precision the remaining columns
    are numerical noise and their order legitimately depends on summation order
    (SURVEY.md F8).  Pivot indices are compared bit-exactly on this prefix only."""
    d = np.abs(np.diag(np.asarray(r))).astype(np.float64)
    floor = {np.dtype(np.float64): 1e-12, np.dtype(np.float32): 2e-5}[np.dtype(dtype)] * d[0]
    below = np.nonzero(d < floor)[0]
    return int(below[0]) if below.size else int(d.size)


def agreed_pivot_prefix(ind, r, ind_ref, r_ref, dtype):
    """Length of the leading run on which two pivoted QRs took the same pivots, with the
    near-tie rule of SURVEY.md section 7 applied at the first disagreement.

    LAPACK only keeps the down-dated partial norms accurate to ~sqrt(eps) (the tol3z
    recompute rule), so when two candidate columns' partial norms agree to that level the
    choice depends on summation order.  A disagreement at position j is accepted only if
    (1) j lies beyond nothing else (all earlier pivots identical) and (2) both
    factorizations report the same |r_jj| to the tie tolerance, i.e. each picked a column
    whose partial norm was maximal to within rounding.  f64 callers additionally assert
    that no disagreement happens inside the stable prefix at all."""
    ns = min(stable_prefix(r_ref, dtype), np.asarray(r).shape[0], np.asarray(r_ref).shape[0])
    ind = np.asarray(ind)[:ns]
    ind_ref = np.asarray(ind_ref)[:ns]
    diff = np.nonzero(ind != ind_ref)[0]
    if diff.size == 0:
        return ns
    j = int(diff[0])
    tie = {np.dtype(np.float64): 1e-6, np.dtype(np.float32): 5e-3}[np.dtype(dtype)]
    a, b = float(abs(r[j, j])), float(abs(r_ref[j, j]))
    assert abs(a - b) <= tie * max(a, b), f"pivot {j} differs and is not a near tie: |r_jj| {a} vs {b}"
    return j


def greedy_pivot_slack(a, r, ind, k):
    """How far every one of the first k pivots of A[:, ind] = Q R is from being THE largest remaining partial column norm.

    Column-pivoted QR (?geqp3) picks at step j the column of largest partial norm; two correct implementations may pick
    different columns only where the candidates tie to the accuracy the down-dated norms are kept to.  From the factorization
    itself (R: k x n in pivoted column order) the partial norms at step j are, in f64,
        vn_j(c)^2 = ||A[:, ind[c]]||^2 - sum_{i < j} R[i, c]^2          (c >= j),
    and the pivot's own is |R[j, j]|.  Returns the list of (max_c vn_j(c) - |R[j, j]|) / max_c vn_j(c), j = 0 .. k-1: every entry
    must be below the tie tolerance for every pivot -- not only the first disagreement with a reference -- to be a legitimate choice."""
    a = np.asarray(a, dtype=np.float64)
    r = np.asarray(r, dtype=np.float64)
    ind = np.asarray(ind)
    vn2 = (a[:, ind] ** 2).sum(axis=0)
    slack = []
    for j in range(k):
        rest = np.sqrt(np.maximum(vn2[j:], 0.0))
        mx = float(rest.max())
        slack.append((mx - abs(r[j, j])) / mx if mx > 0 else 0.0)
        vn2 = vn2 - r[j] ** 2
    return slack
