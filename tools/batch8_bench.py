"""Developer tool: the cfg5 batch of 8 (4096 x 4096 f32, rank-64 column ID) on one GPU; env knobs are read by the library."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
from rusty_compression_amd import batch
NB = int(os.environ.get("NB", "8"))
mats = [rc.random_gaussian((4096, 4096), rc.Rng(500 + i), torch.float32) for i in range(NB)]
for _ in range(2):
    batch.batch_column_id(mats, 64)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    batch.batch_column_id(mats, 64)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 5
tag = " ".join(f"{k}={os.environ[k]}" for k in sorted(os.environ) if k.startswith("RC_"))
print(f"[{tag}] batch of {NB}: {t*1e3:.3f} ms  {NB/t:.1f} matrices/s")
