#!/usr/bin/env python3
"""Timing of the general-shape pivoted QR (blocked ?laqps path vs the per-step chain) on the cfg5 / cfg4 shapes.
Usage: python tools/qrblk_bench.py [--skip-cfg4]"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import rusty_compression_amd as rc  # noqa: E402
from rusty_compression_amd import _lib, batch  # noqa: E402


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
    ap.add_argument("--skip-cfg4", action="store_true")
    a = ap.parse_args()
    out = {}
    ctx = _lib.default_context()
    m1 = rc.random_gaussian((4096, 4096), rc.Rng(500), torch.float32)
    for blocked in (1, 0):
        ctx.set_option(_lib.RC_OPT_BLOCKED_QRCP, blocked)
        t = timed(lambda: batch.column_id_rank(m1, 64), reps=10)
        out[f"cfg5_single_stream_blocked={blocked}"] = {"ms_per_matrix": round(t * 1e3, 3), "matrices_per_s": round(1 / t, 1)}
    ctx.set_option(_lib.RC_OPT_BLOCKED_QRCP, 1)
    mats = [rc.random_gaussian((4096, 4096), rc.Rng(500 + i), torch.float32) for i in range(8)]
    t = timed(lambda: batch.batch_column_id(mats, 64), reps=5, warm=2)
    out["cfg5_batch_of_8"] = {"s_per_batch": round(t, 5), "matrices_per_s": round(8 / t, 1), "tflops_algorithmic": round(8 * 4.245 / t / 1e3, 3)}
    del mats, m1
    if not a.skip_cfg4:
        g = torch.Generator(device="cuda").manual_seed(4)
        u = torch.linalg.qr(torch.randn(16384, 4096, dtype=torch.float64, device="cuda", generator=g)).Q
        v = torch.linalg.qr(torch.randn(4096, 4096, dtype=torch.float64, device="cuda", generator=g)).Q
        sig = torch.logspace(0, -10, 4096, dtype=torch.float64, device="cuda")
        mat = (u * sig) @ v.T
        del u, v
        q, hist = rc.sample_range_adaptive(mat, 1e-6, 64, rc.Rng(11))
        for blocked in (1, 0):
            ctx.set_option(_lib.RC_OPT_BLOCKED_QRCP, blocked)
            warm = rc.QR.compute_from_range_estimate(q, mat).column_id().two_sided_id()  # workspace growth, module load, graph capture
            del warm
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            qr = rc.QR.compute_from_range_estimate(q, mat)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            cid = qr.column_id()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            tid = cid.two_sided_id()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            err = float(rc.rel_diff_fro(tid.to_mat(), mat))
            out[f"cfg4_rank{q.shape[1]}_blocked={blocked}"] = {"qr_from_range_s": round(t1 - t0, 4), "column_id_s": round(t2 - t1, 4),
                                                             "two_sided_id_s": round(t3 - t2, 4), "rel_err": err}
        ctx.set_option(_lib.RC_OPT_BLOCKED_QRCP, 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
