import os, sys
os.environ.setdefault("RC_QRCP_CHECK", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
from rusty_compression_amd import batch
for seed in range(500, 520):
    a = rc.random_gaussian((4096, 4096), rc.Rng(seed), torch.float32)
    try:
        c, z, ind = batch.column_id_rank(a, 64)
        torch.cuda.synchronize()
        print("seed", seed, "ok", flush=True)
    except Exception as e:
        print("seed", seed, "FAILED:", str(e)[:100], flush=True)
