"""Developer tool: S lanes that run ONLY the two big products of a cfg3 compression (sketch Y = A Omega, projection B = Q^H A)
back to back, un-traced: the chip time the products of one compression need when the chip holds nothing else.  The difference
to the headline's time per compression is what everything else (and idling) costs.
    python tools/gemm_lanes.py [--streams 42] [--rounds 12] [--size 8192] [--rank 128] [--oversample 5]"""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
from rusty_compression_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=42)
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--rank", type=int, default=128)
ap.add_argument("--oversample", type=int, default=5)
ap.add_argument("--pad", type=int, default=0, help="lane s starts its matrix s*pad elements into its allocation (address alignment between lanes)")
ap.add_argument("--stagger-us", type=int, default=0, help="host sleep between the first launches of the lanes")
ap.add_argument("--graph", type=int, default=0, help="1: every lane replays a captured hipGraph of its two products (as bench.py does with the whole compression)")
ap.add_argument("--hint", type=int, default=-1, help="RC_OPT_CONCURRENCY_HINT (default: the number of lanes)")
args = ap.parse_args()
m = n = args.size; k = args.rank; l = k + args.oversample; S = args.streams
dt = torch.float64
lib = _lib.lib()
lanes = []
for s in range(S):
    if s < 32:
        st = torch.cuda.Stream()
    else:
        raw = ctypes.c_void_p(); assert lib.rc_stream_create(ctypes.c_int32(0), ctypes.byref(raw)) == 0
        st = torch.cuda.ExternalStream(raw.value)
    with torch.cuda.stream(st):
        ctx = _lib.default_context()
        ctx.set_option(_lib.RC_OPT_CONCURRENCY_HINT, S if args.hint < 0 else args.hint)
        a = rc.random_gaussian((m, n), rc.Rng(s + 1), dt)
        if args.pad:
            big = torch.empty(m * n + S * args.pad, dtype=dt, device="cuda")
            a2 = big[s * args.pad: s * args.pad + m * n].view(m, n); a2.copy_(a); a = a2
        omega = rc.random_gaussian((n, l), rc.Rng(99), dt)
        y = torch.empty((l, m), dtype=dt, device="cuda").t()       # column-major m x l (the library's working layout)
        q = rc.random_gaussian((m, k), rc.Rng(98), dt)
        b = torch.empty((k, n), dtype=dt, device="cuda")
        def call(ctx=ctx, a=a, omega=omega, y=y, q=q, b=b):
            ctx.call("rc_matmat_f64", _lib.mat(a), _lib.mat(omega), _lib.mat(y))
            ctx.call("rc_gemm_f64", ctypes.c_int32(1), ctypes.c_int32(0), ctypes.c_double(1.0), _lib.mat(q), _lib.mat(a), ctypes.c_double(0.0), _lib.mat(b))
        call(); ctx.synchronize()
        if args.graph:
            g = ctypes.c_void_p(None)
            ctx.check(lib.rc_graph_begin_capture(ctx._h)); call(); ctx.check(lib.rc_graph_end_capture(ctx._h, ctypes.byref(g)))
            call = (lambda ctx=ctx, g=g: ctx.check(lib.rc_graph_launch(ctx._h, g)))
            call(); ctx.synchronize()
        lanes.append((st, ctx, call, a, omega, y, q, b))
names = set()
for r in range(2):
    for st, ctx, call, *_ in lanes:
        with torch.cuda.stream(st): call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(args.rounds):
    for st, ctx, call, *_ in lanes:
        with torch.cuda.stream(st): call()
torch.cuda.synchronize()
t = time.perf_counter() - t0
per = t / (args.rounds * S)
flops = 2.0 * m * n * (l + k)
print(f"graph={args.graph} hwq={os.environ.get('GPU_MAX_HW_QUEUES')} pad={args.pad} lanes={S} pairs/s={1/per:.1f} ms_per_pair={per*1e3:.4f} TFLOP/s={flops/per/1e12:.2f} frac_of_78.6={flops/per/78.6e12:.3f}")
