"""Developer tool: soak of the cfg5 paths (single column IDs and the batch of 8) with the per-panel permutation check on
(RC_QRCP_CHECK=1: a bookkeeping fault becomes an error message instead of a wild gather)."""
import os, sys, time
os.environ.setdefault("RC_QRCP_CHECK", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
from rusty_compression_amd import _lib, batch
rounds = int(os.environ.get("ROUNDS", "40"))
mats = [rc.random_gaussian((4096, 4096), rc.Rng(500 + i), torch.float32) for i in range(8)]
ctx = _lib.default_context()
t0 = time.time()
for r in range(rounds):
    out = batch.batch_column_id(mats, 64)
    c, z, ind = batch.column_id_rank(mats[r % 8], 64)
    torch.cuda.synchronize()
    h = ctx.get_health()
    ok = sorted(ind.cpu().tolist()) == list(range(4096))
    if h or not ok:
        print("round", r, "health", h, "single perm valid", ok, flush=True)
        sys.exit(1)
    if r % 10 == 0:
        print("round", r, "ok", round(time.time() - t0, 1), "s", flush=True)
print("soak ok:", rounds, "rounds")
