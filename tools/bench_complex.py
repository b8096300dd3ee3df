#!/usr/bin/env python3
"""Timings of the complex instantiations (c64 / c32) beside their real twins, one stream, eager (the complex
entry points had never been timed).  Informational: the complex kernels are correctness-first (column-parallel Householder
chain, 4M products on the real MFMA GEMM); parity is in tests/test_gpu_complex.py."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import rusty_compression_amd as rc  # noqa: E402


def timed(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def gauss(shape, dtype, seed):
    if dtype.is_complex:
        real = torch.float64 if dtype == torch.complex128 else torch.float32
        re, im = rc.random_gaussian(shape, rc.Rng(seed), real), rc.random_gaussian(shape, rc.Rng(seed + 1000), real)
        return torch.complex(re, im)
    return rc.random_gaussian(shape, rc.Rng(seed), dtype)


def main():
    out = {}
    for name, real, cplx in (("f64/c64", torch.float64, torch.complex128), ("f32/c32", torch.float32, torch.complex64)):
        row = {}
        for tag, dt in (("real", real), ("complex", cplx)):
            r = {}
            a = gauss((2048, 2048), dt, 1)
            x = gauss((2048, 133), dt, 2)
            r["matmat_2048x2048x133_ms"] = round(timed(lambda: rc.matmat(a, x), reps=5) * 1e3, 3)
            y = gauss((2048, 133), dt, 3)
            r["pivoted_qr_2048x133_ms"] = round(timed(lambda: rc.pivoted_qr(y)) * 1e3, 3)
            b = gauss((128, 2048), dt, 4)
            r["pivoted_qr_128x2048_ms"] = round(timed(lambda: rc.pivoted_qr(b)) * 1e3, 3)
            r["compute_svd_128x2048_ms"] = round(timed(lambda: rc.compute_svd(b)) * 1e3, 3)
            r["sample_range_by_rank_2048_k128_ms"] = round(timed(lambda: rc.sample_range_by_rank(a, 128, 5, rc.Rng(7))) * 1e3, 3)
            sq = gauss((512, 512), dt, 5)
            r["pivoted_qr_512x512_ms"] = round(timed(lambda: rc.pivoted_qr(sq)) * 1e3, 3)
            row[tag] = r
            del a, x, y, b, sq
        row["complex_over_real"] = {k: round(row["complex"][k] / max(row["real"][k], 1e-9), 1) for k in row["real"]}
        out[name] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
