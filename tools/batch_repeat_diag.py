"""Developer tool: repeat the cfg5 batch of 8 and compare every factor bit for bit with the one-matrix call; on a mismatch print
where and by how much (which matrix, C or Z, how many entries, the largest difference, the rows / columns touched)."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
from rusty_compression_amd import batch

ROUNDS = int(os.environ.get("ROUNDS", "100"))
NB = int(os.environ.get("NB", "8"))
M, N, K = (int(os.environ.get(k, d)) for k, d in (("M", 4096), ("N", 4096), ("K", 64)))
DT = getattr(torch, os.environ.get("DTYPE", "float32"))
mats = [rc.random_gaussian((M, N), rc.Rng(500 + i), DT) for i in range(NB)]
want = [batch.column_id_rank(a, K) for a in mats]
again = [batch.column_id_rank(a, K) for a in mats]
for i, (w, g) in enumerate(zip(want, again)):
    print("one-matrix call repeats:", i, all(torch.equal(x, y) for x, y in zip(w, g)))
bad = 0
for r in range(ROUNDS):
    print(f'== round {r}', file=sys.stderr, flush=True)
    out = batch.batch_column_id(mats, K)
    for i, ((c, z, ind), (c1, z1, i1)) in enumerate(zip(out, want)):
        for name, x, y in (("ind", ind, i1), ("C", c, c1), ("Z", z, z1)):
            if not torch.equal(x, y):
                bad += 1
                d = (x.double() - y.double()).abs()
                nz = d.nonzero()
                rows = torch.unique(nz[:, 0]) if nz.dim() == 2 and nz.shape[1] > 1 else nz.flatten()
                cols = torch.unique(nz[:, 1]) if nz.dim() == 2 and nz.shape[1] > 1 else nz.flatten()
                print(f"round {r} matrix {i} {name}: {nz.shape[0]} entries differ, max |diff| {d.max().item():.3e} (max |value| {y.abs().max().item():.3e}); "
                      f"{rows.numel()} rows [{rows[:6].tolist()}...], {cols.numel()} cols [{cols[:6].tolist()}...]", flush=True)
    if bad >= 6:
        break
print("rounds", r + 1, "mismatches", bad)
