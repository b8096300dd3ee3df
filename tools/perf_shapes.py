#!/usr/bin/env python3
"""Wall time of the main entry points over a spread of shapes (f64 and f32): a net for performance pathologies
outside the headline workload (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
from rusty_compression_amd import batch


def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for dt in (torch.float64, torch.float32):
    for (m, n) in [(2048, 2048), (4096, 512), (512, 4096), (8192, 64), (64, 8192), (1000, 777), (16384, 256), (300, 20000)]:
        a = rc.random_gaussian((m, n), rc.Rng(1), dt)
        k = min(m, n)
        row = [f"{str(dt)[-7:]:8s} {m:6d}x{n:<6d}"]
        row.append(f"qr_full {t(lambda: rc.pivoted_qr(a)):9.2f} ms")
        row.append(f"qr_k32 {t(lambda: rc.pivoted_qr(a, rank=min(32, k))):8.2f} ms")
        if k <= 1024:
            row.append(f"svd {t(lambda: rc.compute_svd(a)):9.2f} ms")
        row.append(f"colid32 {t(lambda: batch.column_id_rank(a, min(32, k))):8.2f} ms")
        row.append(f"sketch32 {t(lambda: rc.sample_range_by_rank(a, min(32, k // 2), 5, rc.Rng(3))):8.2f} ms")
        print("  ".join(row), flush=True)
        del a
