#!/usr/bin/env python3
"""How well do independent HIP streams overlap on this box?  (diagnostic, not a test)

 (1) S streams, each replaying a hipGraph with one 128x128 SVD (one long single-workgroup
     Jacobi kernel): ideal = time(S=1) for every S <= 256.
 (2) the same S-1 streams + one stream replaying the 8192x133x8192 sketch GEMM: how much
     does the full-chip GEMM slow down while single-CU kernels hold some CUs?
"""
import ctypes
import os
import sys
import time

S_LIST = [int(x) for x in (sys.argv[1:] or ["1", "4", "8", "16", "32"])]
os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(S_LIST) + 1))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rusty_compression_amd as rc  # noqa: E402
from rusty_compression_amd import _lib  # noqa: E402

lib = _lib.lib()
_seen = set()
dt = torch.float64


def make_lane(kind):
    st = torch.cuda.Stream()
    assert st.cuda_stream not in _seen, "torch's stream pool wrapped around (32 streams): use fewer lanes"
    _seen.add(st.cuda_stream)
    with torch.cuda.stream(st):
        ctx = _lib.Context(torch.cuda.current_device(), st.cuda_stream)
        if kind == "svd":
            a = torch.randn(128, 128, dtype=dt, device="cuda")
            u, s, vt = torch.empty(128, 128, dtype=dt, device="cuda"), torch.empty(128, dtype=dt, device="cuda"), torch.empty(128, 128, dtype=dt, device="cuda")
            call = lambda: ctx.call("rc_compute_svd_f64", _lib.mat(a), _lib.mat(u), ctypes.c_void_p(s.data_ptr()), _lib.mat(vt))  # noqa: E731
            keep = (a, u, s, vt)
        else:
            a = torch.randn(8192, 8192, dtype=dt, device="cuda")
            x = torch.randn(8192, 133, dtype=dt, device="cuda")
            y = torch.empty(8192, 133, dtype=dt, device="cuda")
            call = lambda: ctx.call("rc_matmat_f64", _lib.mat(a), _lib.mat(x), _lib.mat(y))  # noqa: E731
            keep = (a, x, y)
        call()
        ctx.synchronize()
        g = ctypes.c_void_p(None)
        ctx.check(lib.rc_graph_begin_capture(ctx._h))
        for _ in range(4):
            call()
        ctx.check(lib.rc_graph_end_capture(ctx._h, ctypes.byref(g)))
    return dict(ctx=ctx, graph=g, keep=keep, stream=st)


def run(lanes, reps):
    for ln in lanes:
        ln["ctx"].synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for ln in lanes:
            ln["ctx"].check(lib.rc_graph_launch(ln["ctx"]._h, ln["graph"]))
    for ln in lanes:
        ln["ctx"].synchronize()
    return (time.perf_counter() - t0) / (reps * 4) * 1e3


svd_lanes = [make_lane("svd") for _ in range(max(S_LIST))]
gemm = make_lane("gemm")
run(svd_lanes[:1], 2)
print("gemm alone: %.3f ms per launch" % run([gemm], 8))
for S in S_LIST:
    t = run(svd_lanes[:S], 6)
    print("svd x%-3d streams: %.3f ms per round (ideal = S=1 value)" % (S, t), flush=True)
# GEMM stream against S-1 busy SVD streams: time the GEMM stream alone while the others run
for S in S_LIST:
    if S < 2:
        continue
    others = svd_lanes[: S - 1]
    for ln in others:
        for _ in range(12):
            ln["ctx"].check(lib.rc_graph_launch(ln["ctx"]._h, ln["graph"]))
    t0 = time.perf_counter()
    for _ in range(8):
        gemm["ctx"].check(lib.rc_graph_launch(gemm["ctx"]._h, gemm["graph"]))
    gemm["ctx"].synchronize()
    tg = (time.perf_counter() - t0) / 32 * 1e3
    for ln in others:
        ln["ctx"].synchronize()
    print("gemm with %-3d svd streams busy: %.3f ms per launch" % (S - 1, tg), flush=True)
