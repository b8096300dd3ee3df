"""Developer tool: the cfg3 pipeline (rc_rsvd_id_f64, 8192 x 8192, k = 128) replayed from hipGraphs on LANES streams at once, every
lane on the SAME matrix and seed, ROUNDS times; every output of every lane must equal lane 0's eager result bit for bit (the big
GEMMs of the other lanes are the memory load under which the cooperative QRCP exchanges its columns)."""
import ctypes, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
from rusty_compression_amd import _lib

LANES, ROUNDS = int(os.environ.get("LANES", "16")), int(os.environ.get("ROUNDS", "40"))
SEEDS = int(os.environ.get("SEEDS", "0"))  # 1: every lane its own Omega stream (lanes drift apart: GEMMs of one beside the cooperative QRCP of another)
REPS = int(os.environ.get("REPS", "1"))    # graph launches per lane and round, in shuffled lane order, before the round is compared
m = n = int(os.environ.get("N", "8192")); k, p = 128, 5
dt = torch.float64
a = rc.random_gaussian((m, n), rc.Rng(11), dt)
lib = _lib.lib()
lanes = []
for s in range(LANES):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ctx = _lib.Context(torch.cuda.current_device(), st.cuda_stream)
        mk = lambda r, c: torch.zeros((r, c), dtype=dt, device="cuda")
        b = dict(range_q=mk(m, k), u=mk(m, k), s=torch.zeros(k, dtype=dt, device="cuda"), vt=mk(k, n), qr_q=mk(m, k), qr_r=mk(k, n),
                 qr_ind=torch.zeros(n, dtype=torch.int64, device="cuda"), id_c=mk(m, k), id_z=mk(k, n))
        out = _lib.rc_rsvd_id_out(_lib.mat(b["range_q"]), _lib.mat(b["u"]), ctypes.c_void_p(b["s"].data_ptr()), _lib.mat(b["vt"]), _lib.mat(b["qr_q"]),
                                  _lib.mat(b["qr_r"]), ctypes.c_void_p(b["qr_ind"].data_ptr()), _lib.mat(b["id_c"]), _lib.mat(b["id_z"]))
        call = lambda ctx=ctx, out=out, s=s: ctx.call("rc_rsvd_id_f64", _lib.mat(a), ctypes.c_int64(k), ctypes.c_int64(p), _lib.mat(None), ctypes.c_uint64(7 + (s if SEEDS else 0)), ctypes.byref(out))
        ctx.set_option(_lib.RC_OPT_CONCURRENCY_HINT, LANES)
        call(); ctx.synchronize()
        graph = ctypes.c_void_p(None)
        ctx.check(lib.rc_graph_begin_capture(ctx._h)); call(); ctx.check(lib.rc_graph_end_capture(ctx._h, ctypes.byref(graph)))
        lanes.append(dict(ctx=ctx, st=st, b=b, out=out, graph=graph))
refs = [{f: t.clone() for f, t in ln["b"].items()} for ln in lanes]
ref = refs[0]
import random
rng = random.Random(5)
for i, ln in enumerate(lanes[1:] if not SEEDS else [], 1):
    for f, t in ln["b"].items():
        if not torch.equal(t, ref[f]):
            d = (t.double() - ref[f].double()).abs()
            print(f"eager result of lane {i} differs from lane 0's in {f}: {int((d != 0).sum())} entries, max |diff| {d.max().item():.3e} (a certified fast path "
                  f"that fell back computes the same factors along another route; the replays below are compared with lane 0)")
bad = 0
for r in range(ROUNDS):
    for ln in lanes:
        for t in ln["b"].values(): t.zero_()
    torch.cuda.synchronize()
    for _ in range(REPS):
        order = list(range(LANES))
        if REPS > 1: rng.shuffle(order)
        for i in order: lanes[i]["ctx"].check(lib.rc_graph_launch(lanes[i]["ctx"]._h, lanes[i]["graph"]))
    for ln in lanes: ln["ctx"].synchronize()
    for i, ln in enumerate(lanes):
        h = ln["ctx"].get_health()
        if h: print(f"round {r} lane {i}: health {h:#x}")
        for f, t in ln["b"].items():
            want = refs[i][f] if SEEDS else ref[f]
            if not torch.equal(t, want):
                bad += 1
                d = (t.double() - want.double()).abs()
                print(f"round {r} lane {i} {f}: {int((d != 0).sum())} entries differ, max |diff| {d.max().item():.3e}", flush=True)
    if bad > 12: break
print("lanes", LANES, "rounds", r + 1, "mismatching outputs", bad)
