"""Developer experiment: does partitioning the chip help cfg3?  The big products run on CU-masked streams that own P_G CUs, everything
else (range finder's QR, the consumers with the cooperative QRCP) on streams that own the other CUs; a compression is four graphs
handed between the two streams of a pair with events; two lanes per pair software-pipeline each other.
    python tools/partition_probe.py [--pairs 12] [--gemm-cus 192] [--mask 1] [--rounds 8]
Work per compression (public entry points only; roughly the consumers of rc_rsvd_id, not bit-identical to it):
  G1: Omega, Y = A Omega                      F1: pivoted QR of Y truncated to k -> range
  G2: B = range^T A                           F2: rc_geqp3(B copy) [cooperative QRCP], rc_compute_svd(B), U = range U_b"""
import argparse, ctypes, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
from rusty_compression_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=12)
ap.add_argument("--lanes-per-pair", type=int, default=2)
ap.add_argument("--gemm-cus", type=int, default=192)
ap.add_argument("--mask", type=int, default=1)
ap.add_argument("--rounds", type=int, default=8)
args = ap.parse_args()
m = n = 8192; k = 128; p = 5; l = k + p
dt = torch.float64
lib = _lib.lib()
hip = ctypes.CDLL("libamdhip64.so.7")
torch.cuda.init(); torch.zeros(1, device="cuda")

def masked_stream(lo, hi):
    """a stream whose kernels may use the CUs [lo, hi) of the 256 (bit i of the mask = CU i in the runtime's numbering)"""
    if not args.mask:
        return torch.cuda.Stream()
    words = (ctypes.c_uint32 * 8)()
    for i in range(lo, hi):
        words[i // 32] |= (1 << (i % 32))
    st = ctypes.c_void_p()
    e = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(8), words)
    assert e == 0, "hipExtStreamCreateWithCUMask -> %d" % e
    return torch.cuda.ExternalStream(st.value)

pairs = []
for ip in range(args.pairs):
    sG, sF = masked_stream(0, args.gemm_cus), masked_stream(args.gemm_cus, 256)
    with torch.cuda.stream(sG):
        cG = _lib.default_context(); cG.set_option(_lib.RC_OPT_CONCURRENCY_HINT, 40)
    with torch.cuda.stream(sF):
        cF = _lib.default_context(); cF.set_option(_lib.RC_OPT_CONCURRENCY_HINT, 40)
    lanes = []
    for il in range(args.lanes_per_pair):
        with torch.cuda.stream(sG):
            mk = lambda r, c: torch.empty((r, c), dtype=dt, device="cuda")
            b = dict(a=rc.random_gaussian((m, n), rc.Rng(100 * ip + il + 1), dt), omega=mk(n, l + 1)[:, :l], y=torch.empty((l, m), dtype=dt, device="cuda").t(),
                     range=mk(m, k), rr=mk(k, l), ind=torch.empty(l, dtype=torch.int64, device="cuda"), bb=mk(k, n), bw=mk(k, n),
                     jp=torch.empty(n, dtype=torch.int64, device="cuda"), tau=torch.empty(k, dtype=dt, device="cuda"),
                     ub=mk(k, k), s=torch.empty(k, dtype=dt, device="cuda"), vt=mk(k, n), u=mk(m, k))
        lanes.append(b)
    pairs.append(dict(sG=sG, sF=sF, cG=cG, cF=cF, lanes=lanes))

def stage(name, c, b, seed):
    M = _lib.mat
    if name == "G1":
        c.call("rc_random_gaussian_f64", M(b["omega"]), ctypes.c_uint64(seed), ctypes.c_uint64(0))
        c.call("rc_matmat_f64", M(b["a"]), M(b["omega"]), M(b["y"]))
    elif name == "F1":
        c.call("rc_pivoted_qr_f64", M(b["y"]), M(b["range"]), M(b["rr"]), _lib.i64p(b["ind"]))
    elif name == "G2":
        c.call("rc_gemm_f64", ctypes.c_int32(1), ctypes.c_int32(0), ctypes.c_double(1.0), M(b["range"]), M(b["a"]), ctypes.c_double(0.0), M(b["bb"]))
    else:
        b["bw"].copy_(b["bb"])
        c.call("rc_geqp3_f64", M(b["bw"]), ctypes.c_int64(k), _lib.i64p(b["jp"]), ctypes.c_void_p(b["tau"].data_ptr()))
        c.call("rc_compute_svd_f64", M(b["bb"]), M(b["ub"]), ctypes.c_void_p(b["s"].data_ptr()), M(b["vt"]))
        c.call("rc_gemm_f64", ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_double(1.0), M(b["range"]), M(b["ub"]), ctypes.c_double(0.0), M(b["u"]))

# eager warm-up (sizes the arenas), then one graph per (lane, stage)
for pr in pairs:
    for il, b in enumerate(pr["lanes"]):
        for name in ("G1", "F1", "G2", "F2"):
            c, st = (pr["cG"], pr["sG"]) if name[0] == "G" else (pr["cF"], pr["sF"])
            with torch.cuda.stream(st):
                stage(name, c, b, 7 + il)
            c.synchronize()
    pr["graphs"] = []
    for il, b in enumerate(pr["lanes"]):
        gs = {}
        for name in ("G1", "F1", "G2", "F2"):
            c, st = (pr["cG"], pr["sG"]) if name[0] == "G" else (pr["cF"], pr["sF"])
            with torch.cuda.stream(st):
                g = ctypes.c_void_p(None)
                if name == "F2":
                    b["bw"].copy_(b["bb"])  # (torch copy is not capturable through the library's capture: keep it eager, outside)
                c.check(lib.rc_graph_begin_capture(c._h))
                if name == "F2":
                    M = _lib.mat
                    c.call("rc_geqp3_f64", M(b["bw"]), ctypes.c_int64(k), _lib.i64p(b["jp"]), ctypes.c_void_p(b["tau"].data_ptr()))
                    c.call("rc_compute_svd_f64", M(b["bb"]), M(b["ub"]), ctypes.c_void_p(b["s"].data_ptr()), M(b["vt"]))
                    c.call("rc_gemm_f64", ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_double(1.0), M(b["range"]), M(b["ub"]), ctypes.c_double(0.0), M(b["u"]))
                else:
                    stage(name, c, b, 7 + il)
                c.check(lib.rc_graph_end_capture(c._h, ctypes.byref(g)))
            gs[name] = g
        pr["graphs"].append(gs)
torch.cuda.synchronize()

def launch(pr, il, name):
    c, st = (pr["cG"], pr["sG"]) if name[0] == "G" else (pr["cF"], pr["sF"])
    with torch.cuda.stream(st):
        c.check(lib.rc_graph_launch(c._h, pr["graphs"][il][name]))

def one_round():
    # per pair: G1(A) G1(B) | F1(A) F1(B) | G2(A) G2(B) | F2(A) F2(B), each stage after its predecessor of the same lane
    for pr in pairs:
        L = range(len(pr["lanes"]))
        ev = {}
        for il in L:
            if "f2" in pr and il in pr["f2"]:
                pr["sG"].wait_event(pr["f2"][il])   # the lane's buffers are free again
            launch(pr, il, "G1"); ev[il] = torch.cuda.Event(); ev[il].record(pr["sG"])
        for il in L:
            pr["sF"].wait_event(ev[il]); launch(pr, il, "F1"); ev[il] = torch.cuda.Event(); ev[il].record(pr["sF"])
        for il in L:
            pr["sG"].wait_event(ev[il]); launch(pr, il, "G2"); ev[il] = torch.cuda.Event(); ev[il].record(pr["sG"])
        pr["f2"] = {}
        for il in L:
            pr["sF"].wait_event(ev[il]); launch(pr, il, "F2"); e = torch.cuda.Event(); e.record(pr["sF"]); pr["f2"][il] = e

for _ in range(2): one_round()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.rounds): one_round()
torch.cuda.synchronize()
t = time.perf_counter() - t0
nc = args.rounds * args.pairs * args.lanes_per_pair
assert not any(pr["cG"].get_health() or pr["cF"].get_health() for pr in pairs)
print(f"mask={args.mask} gemm_cus={args.gemm_cus} pairs={args.pairs} x {args.lanes_per_pair} lanes: {nc/t:.1f} compressions/s ({t/nc*1e3:.4f} ms each)", flush=True)
