"""Worker of test_results_do_not_depend_on_what_fresh_workspace_memory_holds: a fresh process runs one call of every kind (all of
them FIRST calls on their workspace) and prints a digest of every output.  With RC_DEBUG_POISON_WORKSPACE=1 the library fills
workspace memory that is new to a context with small integers before handing it out; the digests must not change."""
import hashlib, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import rusty_compression_amd as rc
from rusty_compression_amd import batch


def dg(*ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()[:16]


out = {}
# (the small SVDs come first: their workspace is then memory no earlier call of this process has written)
for shape in ((100, 60), (64, 128), (300, 200)):
    u, s, vt = rc.compute_svd(rc.random_gaussian(shape, rc.Rng(6), torch.float64))
    out[f"svd_{shape[0]}x{shape[1]}"] = dg(s, (u * s) @ vt)
a = rc.random_gaussian((4096, 2048), rc.Rng(3), torch.float64)
q = rc.sample_range_by_rank(a, 96, 5, rc.Rng(4))
svd = rc.SVD.compute_from_range_estimate(q, a)
qr = rc.QR.compute_from_range_estimate(q, a)
cid = qr.column_id()
out["rsvd_f64"] = dg(q, svd.u, svd.s, svd.vt, qr.q, qr.r, qr.ind, cid.c, cid.z)
b = rc.random_gaussian((2048, 2048), rc.Rng(5), torch.float32)
out["column_id_rank_f32"] = dg(*batch.column_id_rank(b, 64))
out["batch_f32"] = dg(*[t for tri in batch.batch_column_id([b, b.t().contiguous(), b + 1], 48) for t in tri])
out["pivoted_qr_tall_f32"] = dg(*rc.pivoted_qr(rc.random_gaussian((3000, 100), rc.Rng(7), torch.float32)))
out["pivoted_qr_wide_f64"] = dg(*rc.pivoted_qr(rc.random_gaussian((96, 3000), rc.Rng(8), torch.float64)))
c = rc.random_gaussian((1500, 900), rc.Rng(9), torch.float64)
out["pivoted_qr_general_f64"] = dg(*rc.pivoted_qr(c))
torch.cuda.synchronize()
print("DIGESTS " + json.dumps(out))
