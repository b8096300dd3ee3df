#!/usr/bin/env python3
"""Timings of the non-headline workloads of BASELINE.json `configs` on one MI355X.
Informational (DESIGN.md quotes them); the parity of the same workloads is in tests/test_gpu_parity.py.

  cfg2  4096 x 4096 f64 N(0,1): sample_range_by_rank(k = 64, p = 5)          (sketch + pivoted QR + form Q)
  cfg4  16384 x 4096 f64, sigma = geomspace(1, 1e-10): sample_range_adaptive(1e-6, s) + QR-from-range + column ID + two-sided ID
  cfg5  N x (4096 x 4096 f32) N(0,1): rank-64 column ID per matrix (8 per GPU in the config), S streams
  h2d   upload of one 8192 x 8192 f64 matrix (the host-buffer variant of the headline workload)
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import rusty_compression_amd as rc  # noqa: E402
from rusty_compression_amd import batch  # noqa: E402


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip", default="")
    a = ap.parse_args()
    out = {}
    if "cfg2" not in a.skip:
        m = rc.random_gaussian((4096, 4096), rc.Rng(2))
        t = timed(lambda: rc.sample_range_by_rank(m, 64, 5, rc.Rng(7)), reps=20)
        gf = 2.315 + 2 * 0.0388
        out["cfg2"] = {"ms": round(t * 1e3, 3), "per_s": round(1 / t, 1), "tflops_algorithmic": round(gf / t / 1e3, 2), "note": "single stream, eager"}
        del m
    if "cfg4" not in a.skip:
        # input recipe of the reference (src/random_matrix.rs:70-93); the orthogonal factors come from torch's QR here
        # (input generation only: the engine's own SVD core stops at min(m, n) = 1024)
        g = torch.Generator(device="cuda").manual_seed(4)
        u = torch.linalg.qr(torch.randn(16384, 4096, dtype=torch.float64, device="cuda", generator=g)).Q
        v = torch.linalg.qr(torch.randn(4096, 4096, dtype=torch.float64, device="cuda", generator=g)).Q
        sig = torch.logspace(0, -10, 4096, dtype=torch.float64, device="cuda")
        mat = (u * sig) @ v.T
        del u, v
        res = {}
        for s in (64, 256):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            q, hist = rc.sample_range_adaptive(mat, 1e-6, s, rc.Rng(11))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            qr = rc.QR.compute_from_range_estimate(q, mat)
            cid = qr.column_id()
            tid = cid.two_sided_id()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            err = float(rc.rel_diff_fro(tid.to_mat(), mat))
            res[f"s={s}"] = {"rank": int(q.shape[1]), "iterations": len(hist), "adaptive_s": round(t1 - t0, 3), "qr_id_two_sided_s": round(t2 - t1, 3),
                             "rel_err_two_sided": err}
            del q, qr, cid, tid
        out["cfg4"] = res
        del mat
    if "cfg5" not in a.skip:
        n_mat, S = 8, 8
        mats = [rc.random_gaussian((4096, 4096), rc.Rng(500 + i), torch.float32) for i in range(n_mat)]
        streams = [torch.cuda.Stream() for _ in range(S)]

        def run():
            for i, mt in enumerate(mats):
                with torch.cuda.stream(streams[i % S]):
                    batch.column_id_rank(mt, 64)

        t = timed(run, reps=3, warm=1)
        out["cfg5"] = {"matrices": n_mat, "streams": S, "s_per_batch_of_8": round(t, 4), "matrices_per_s": round(n_mat / t, 2),
                       "tflops_algorithmic": round(n_mat * 4.245 / t / 1e3, 3)}
        del mats
    if "h2d" not in a.skip:
        host = torch.randn(8192, 8192, dtype=torch.float64)
        pinned = host.pin_memory()
        dev = torch.empty_like(host, device="cuda")
        t_page = timed(lambda: dev.copy_(host), reps=3, warm=1)
        t_pin = timed(lambda: dev.copy_(pinned, non_blocking=True), reps=5, warm=1)
        out["h2d_8192x8192_f64"] = {"pageable_ms": round(t_page * 1e3, 2), "pinned_ms": round(t_pin * 1e3, 2), "pinned_GBps": round(host.numel() * 8 / t_pin / 1e9, 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
