"""Developer tool: un-traced chip time per compression of cfg3 with parts of the pipeline left out (40 lanes replaying hipGraphs, as
bench.py): what the SVD consumer, the QR + ID consumer and the range finder's QR cost beside the two big products.
    python tools/ablate_pipeline.py [--streams 40] [--rounds 10]"""
import argparse, ctypes, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
from rusty_compression_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=40)
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--modes", default="pair,range,svd,qr,id,full")
args = ap.parse_args()
m = n = 8192; k = 128; p = 5; l = k + p; S = args.streams
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
        ctx.set_option(_lib.RC_OPT_CONCURRENCY_HINT, S)
        a = rc.random_gaussian((m, n), rc.Rng(s + 1), dt)
        mk = lambda r, c: torch.empty((r, c), dtype=dt, device="cuda")
        b = dict(range_q=mk(m, k), u=mk(m, k), s=torch.empty(k, dtype=dt, device="cuda"), vt=mk(k, n), qr_q=mk(m, k), qr_r=mk(k, n),
                 qr_ind=torch.empty(n, dtype=torch.int64, device="cuda"), id_c=mk(m, k), id_z=mk(k, n),
                 omega=rc.random_gaussian((n, l + 1), rc.Rng(99), dt)[:, :l], y=torch.empty((l, m), dtype=dt, device="cuda").t(), bb=mk(k, n))
    lanes.append((st, ctx, a, b))
none = _lib.mat(None)

def make_call(mode, ctx, a, b, seed):
    if mode == "pair":
        def call():
            ctx.call("rc_matmat_f64", _lib.mat(a), _lib.mat(b["omega"]), _lib.mat(b["y"]))
            ctx.call("rc_gemm_f64", ctypes.c_int32(1), ctypes.c_int32(0), ctypes.c_double(1.0), _lib.mat(b["range_q"]), _lib.mat(a), ctypes.c_double(0.0), _lib.mat(b["bb"]))
        return call
    if mode == "range":   # sample_range_by_rank + the projection: the two products and the range finder's pivoted QR
        def call():
            ctx.call("rc_sample_range_by_rank_f64", _lib.mat(a), ctypes.c_int64(k), ctypes.c_int64(p), _lib.mat(None), ctypes.c_uint64(seed), _lib.mat(b["range_q"]))
            ctx.call("rc_gemm_f64", ctypes.c_int32(1), ctypes.c_int32(0), ctypes.c_double(1.0), _lib.mat(b["range_q"]), _lib.mat(a), ctypes.c_double(0.0), _lib.mat(b["bb"]))
        return call
    if mode == "geqp3":   # range + the pivoted QR kernel of B alone (rc_geqp3: copy, cooperative kernel, gather)
        def call():
            ctx.call("rc_sample_range_by_rank_f64", _lib.mat(a), ctypes.c_int64(k), ctypes.c_int64(p), _lib.mat(None), ctypes.c_uint64(seed), _lib.mat(b["range_q"]))
            ctx.call("rc_gemm_f64", ctypes.c_int32(1), ctypes.c_int32(0), ctypes.c_double(1.0), _lib.mat(b["range_q"]), _lib.mat(a), ctypes.c_double(0.0), _lib.mat(b["bb"]))
            ctx.call("rc_geqp3_f64", _lib.mat(b["bb"]), ctypes.c_int64(k), _lib.i64p(b["qr_ind"]), ctypes.c_void_p(b["s"].data_ptr()))
        return call
    svd = mode in ("svd", "full"); idb = mode in ("id", "full"); qrb = idb or mode == "qr"   # qr: QR::compute_from_range_estimate without the ID
    out = _lib.rc_rsvd_id_out(_lib.mat(b["range_q"]), _lib.mat(b["u"]) if svd else none, ctypes.c_void_p(b["s"].data_ptr() if svd else None), _lib.mat(b["vt"]) if svd else none,
                              _lib.mat(b["qr_q"]) if qrb else none, _lib.mat(b["qr_r"]) if qrb else none, ctypes.c_void_p(b["qr_ind"].data_ptr() if qrb else None),
                              _lib.mat(b["id_c"]) if idb else none, _lib.mat(b["id_z"]) if idb else none)
    return lambda: ctx.call("rc_rsvd_id_f64", _lib.mat(a), ctypes.c_int64(k), ctypes.c_int64(p), _lib.mat(None), ctypes.c_uint64(seed), ctypes.byref(out)) or out

res = {}
for mode in args.modes.split(","):
    graphs = []
    for i, (st, ctx, a, b) in enumerate(lanes):
        with torch.cuda.stream(st):
            call = make_call(mode, ctx, a, b, 7 + i)
            call(); ctx.synchronize()
            g = ctypes.c_void_p(None)
            ctx.check(lib.rc_graph_begin_capture(ctx._h)); keep = call(); ctx.check(lib.rc_graph_end_capture(ctx._h, ctypes.byref(g)))
            ctx.check(lib.rc_graph_launch(ctx._h, g)); ctx.synchronize()
            graphs.append((g, keep))
    def rnd():
        for (st, ctx, a, b), (g, _) in zip(lanes, graphs):
            with torch.cuda.stream(st): ctx.check(lib.rc_graph_launch(ctx._h, g))
    for _ in range(2): rnd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.rounds): rnd()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / (args.rounds * S)
    res[mode] = per * 1e3
    print(f"{mode:6s} {per*1e3:.4f} ms chip time per lane-job   {1/per:8.1f} /s", flush=True)
    assert not any(ctx.get_health() for _, ctx, _, _ in lanes)
    for (st, ctx, a, b), (g, _) in zip(lanes, graphs): lib.rc_graph_destroy(ctx._h, g)
if all(x in res for x in ("pair", "range", "svd", "id", "full")):
    print(f"products {res['pair']:.3f} | range finder's QR {res['range']-res['pair']:.3f} | SVD consumer {res['svd']-res['range']:.3f} | QR + ID consumer {res['id']-res['range']:.3f} | "
          f"sum {res['svd']+res['id']-res['range']:.3f} vs full {res['full']:.3f} ms")
