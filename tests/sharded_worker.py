"""Worker of tests/test_gpu_sharded.py: one rank of the row-sharded rSVD + ID (rusty_compression_amd/sharded.py).
Launched as `python tests/sharded_worker.py <out.npz>` with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set; every rank uses
GPU 0 of the box (the test machine has one) and a gloo group: the native call gets a host communicator whose callbacks run the
group's collectives on the library's staged host copies."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

import rusty_compression_amd as rc
from rusty_compression_amd import sharded


def test_matrix(m, n, dtype=torch.float64):
    """Seeded m x n matrix with a decaying spectrum, the same bits on every rank (device generator of the library)."""
    r = min(m, n, 96)  # numerical rank well below the k + p of the tests
    g1 = rc.random_gaussian((m, r), rc.Rng(71), dtype)
    g2 = rc.random_gaussian((r, n), rc.Rng(72), dtype)
    sig = torch.logspace(0, -8, r, dtype=dtype, device="cuda")
    return rc.dot(g1, sig[:, None] * g2) * (1.0 / np.sqrt(m * n))


def main():
    out = sys.argv[1]
    m, n, k, p, seed = (int(os.environ.get(x, d)) for x, d in (("SH_M", 2048), ("SH_N", 1024), ("SH_K", 64), ("SH_P", 5), ("SH_SEED", 9)))
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a = test_matrix(m, n)
        rows = np.array_split(np.arange(m), world)[rank]
        a_loc = a[int(rows[0]):int(rows[-1]) + 1].contiguous()
        dtype = getattr(torch, os.environ.get("SH_DTYPE", "float64"))
        a_loc = a_loc.to(dtype)
        res = sharded.rsvd_id_row_sharded(a_loc, k, p, seed)  # the native call rc_rsvd_id_row_sharded_* over a host communicator
        torch.cuda.synchronize()
        comp = sharded.rsvd_id_row_sharded(a_loc, k, p, seed, ops=sharded.DeviceOps)  # the same algebra from the one-matrix calls
        torch.cuda.synchronize()
        fields = ("range_q", "u", "s", "vt", "qr_q", "r", "ind", "c", "z")
        np.savez(out, rows=rows, **{f: getattr(res, f).cpu().numpy() for f in fields}, **{"comp_" + f: getattr(comp, f).cpu().numpy() for f in fields})
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
