#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native compression engine.

Workload (BASELINE.json configs[2], the one `metric` is quoted on; it fits one GPU):
  8192 x 8192 f64 dense i.i.d. N(0,1), rank k = 128, oversampling p = 5:
  one STEP = one "rSVD + ID" compression through the C ABI (rc_rsvd_id_f64):
      sample_range_by_rank -> SVD::compute_from_range_estimate
                           -> QR::compute_from_range_estimate -> column_id
  with A resident in HBM and Omega generated on the device.

  python bench.py --gpus N --steps K --warmup W      (N > 1: under torch.distributed.run, or alone -- bench.py then starts
                                                      its N ranks itself, see launch_ranks)
  python bench.py --config cfg5 [--gpus N]           BASELINE.json configs[4]: the sharded batch of column IDs + factor gather

Independent matrices are the unit of parallelism: every rank
compresses its own matrices (weak scaling, no data-path collective); within a rank
`--streams S` independent compressions are in flight on S HIP streams ("lanes"), each
replayed from a hipGraph, so the launch-bound pivot chain of one overlaps the GEMMs of
another.  ONE STEP = one compression on EACH of the S lanes (config.compressions_per_step = S):
`--steps K --warmup W` runs W untimed and K timed rounds over all lanes, so every command
line measures the steady state (K*S compressions per GPU), `value` = K*S*N / elapsed and
`ms_per_step` = elapsed / K.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# hardware queues for the in-flight compressions (the HIP default of 4 serialises the streams); must be set before
# the HIP runtime initialises.  Measured (tools/stream_sweep.sh): 24 queues x 42 streams is the best point (975-980
# compressions/s; 40 streams 951-960, 44-46 streams 961, 48 streams 908), 16 or 20 queues 885-900, 28-32 queues
# 845-890, 48 queues with >= 48 streams collapses to 316.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

F64_MFMA_PEAK_TFLOPS = 78.6   # MI355X f64 matrix peak (vendor datasheet value)
HBM_PEAK_GBS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def work_model(m, n, k, p, with_id=True):
    """Algorithmic flops / bytes of one compression (DESIGN.md 'Work model')."""
    l = k + p
    fl = {
        "sketch_gemm": 2.0 * m * n * l,
        "qrcp_y": 2.0 * m * l * l - (2.0 / 3.0) * l ** 3,
        "form_q": 2.0 * m * l * l - (2.0 / 3.0) * l ** 3,
        "project_gemm": 2.0 * m * n * k,
        "svd_b": 6.0 * n * k * k + 22.0 * k ** 3,
        "u_gemm": 2.0 * m * k * k,
    }
    by = 2 * 8.0 * m * n + 8.0 * (m * k + k + k * n)
    if with_id:
        fl["qrcp_b"] = 2.0 * n * k * k - (2.0 / 3.0) * k ** 3
        fl["trsm"] = 1.0 * k * k * (n - k)
        fl["c_gemm"] = 2.0 * m * k * k
        by += 8.0 * (m * k + k * n) + 8.0 * n
    return fl, by


def host_threads():
    """Threads the oracle's BLAS / LAPACK (SciPy's OpenBLAS) runs on, and where that number comes from."""
    try:
        from threadpoolctl import threadpool_info

        pools = [(int(x.get("num_threads", 1)), x.get("internal_api", "?")) for x in threadpool_info()]
        if pools:
            nthr, api = max(pools)
            return nthr, "threadpoolctl (%s pool)" % api
    except Exception:
        pass
    return os.cpu_count() or 1, "os.cpu_count()"


# The contract is ONE JSON line on stdout.  Libraries below us write there too (RCCL prints a version banner at communicator
# creation), so the process's fd 1 is pointed at stderr for the whole run and the line goes to the original stdout.
_REAL_STDOUT = None


def _guard_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def _emit(text):
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (text + "\n").encode())


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12, help="timed rounds; one round = one compression on each lane (cfg3) / one batch per GPU (cfg5)")
    ap.add_argument("--warmup", type=int, default=2, help="untimed rounds")
    ap.add_argument("--config", choices=("cfg3", "cfg5"), default="cfg3",
                    help="cfg3 (default, the headline): 8192^2 f64 rank-128 rSVD+ID, independent matrices per GPU.  cfg5: BASELINE.json configs[4], "
                         "4096^2 f32 rank-64 column ID, --matrices-per-gpu (8) per rank through rc_batch_column_id_f32 + the factor gather to rank 0")
    ap.add_argument("--streams", type=int, default=44, help="independent compressions in flight per GPU")
    ap.add_argument("--size", type=int, default=None, help="matrix size (cfg3: 8192, cfg5: 4096)")
    ap.add_argument("--rank", type=int, default=None, help="target rank (cfg3: 128, cfg5: 64)")
    ap.add_argument("--oversample", type=int, default=5)
    ap.add_argument("--matrices-per-gpu", type=int, default=8, help="cfg5: matrices of the batch held by each rank (weak scaling: 8 N in total)")
    ap.add_argument("--batch-lanes", type=int, default=8, help="cfg5: contexts / streams the batch call spreads its matrices over")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--no-id", action="store_true", help="rSVD only (skip QR-from-range + column ID)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-reps", type=int, default=2, help="compressions timed on the host (about 5 s each on the GPU box)")
    ap.add_argument("--no-concurrency-hint", action="store_true", help="leave RC_OPT_CONCURRENCY_HINT at 1 (every GEMM splits K for a lone launch)")
    ap.add_argument("--no-gemm-lanes", action="store_true", help="skip roofline.in_flight_live (every lane running only the two big products, after the timed region)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the second throughput figure that re-uploads A before every compression")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous rehearsal WITHOUT a GPU: ranks meet over gloo, run barriers and the MAX reduction around sleeps; "
                         "the line carries value null and dry_run true -- never a measurement (tests/test_dist_cpu.py)")
    ap.add_argument("--lane-events", action="store_true",
                    help="diagnostic: HIP events around every timed step on its lane's stream; start/end offsets go to stderr")
    ap.add_argument("--profile-concurrent", action="store_true",
                    help="diagnostic: per-stage HIP-event timers with ALL streams busy (eager launches), printed to stderr")
    return ap.parse_args(argv)


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside a launcher (no WORLD_SIZE in the environment): this process starts the N ranks itself,
    as plain child processes with the rendezvous variables torch.distributed.run would set, forwards rank 0's single JSON line and
    exits with the worst child status.  It never imports torch and never touches the GPU (and nothing here replaces a process:
    children are started with Popen).  Returns the exit status."""
    import subprocess
    import threading

    n = args.gpus
    env0 = dict(os.environ)
    env0.setdefault("MASTER_ADDR", "127.0.0.1")
    env0["MASTER_PORT"] = str(_free_port())
    env0["WORLD_SIZE"] = env0["LOCAL_WORLD_SIZE"] = str(n)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
        # rank 0's stdout carries the line; whatever another rank writes to its stdout is diagnostics
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None))
    captured = []
    reader = threading.Thread(target=lambda: captured.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("RC_BENCH_LAUNCH_TIMEOUT", "1500"))
    first_failure = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad and first_failure is None:
            first_failure = time.time()  # the others would wait at the rendezvous / a barrier for ever: give them a moment, then end them
        if (first_failure is not None and time.time() - first_failure > 5.0) or time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.terminate()   # exactly the processes started above
            t_kill = time.time() + 10.0
            while any(p.poll() is None for p in procs) and time.time() < t_kill:
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
    reader.join(timeout=10.0)
    codes = [p.wait() for p in procs]
    own = [c for c in codes if c > 0]   # a rank's own failure status ranks above the signal this launcher ended its peers with
    worst = max(own) if own else max((128 - c if c < 0 else 0) for c in codes)
    text = (captured[0] if captured else b"").decode(errors="replace")
    lines = [ln for ln in text.splitlines() if ln.strip().startswith("{")]
    if worst == 0 and len(lines) != 1:
        print("bench.py launcher: rank 0 printed %d JSON lines, expected 1" % len(lines), file=sys.stderr)
        worst = 1
    if worst == 0:
        rec = json.loads(lines[0])
        if rec.get("n_gpus") != n:
            print("bench.py launcher: rank 0 reports n_gpus=%r, %d ranks were started" % (rec.get("n_gpus"), n), file=sys.stderr)
            worst = 1
        else:
            _emit(lines[0])
    else:
        print("bench.py launcher: rank exit codes %s" % codes, file=sys.stderr)
    return worst


def run_dry(args):
    """Rendezvous / barrier / MAX-over-ranks / one-line path of an N-rank run with the compute replaced by sleeps (no GPU needed;
    the CPU suite starts `bench.py --gpus 2 --dry-run` and checks that the launcher produced two ranks)."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("RC_BENCH_DRY_FAIL_RANK") == str(rank):
        raise SystemExit(3)   # test hook: a rank that dies before the rendezvous
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (rank + 1))
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.destroy_process_group()
    if rank == 0:
        _emit(json.dumps({"metric": "dry run of the launcher -- not a measurement", "value": None, "unit": None, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3), "dry_run": True, "data": "none",
                          "config": {"workload": "%s launcher rehearsal, no GPU work" % args.config}}))


def setup_ranks(args):
    """Device selection + process group of a (possibly multi-rank) GPU run.  Returns (torch, dist or None, world, rank, local_rank, rehearsal)."""
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    # rehearsal of the N > 1 path on a one-GPU box: RC_BENCH_REHEARSAL=1 puts every rank on device 0 and uses gloo
    # for the barrier / max-time reduction (RCCL refuses two ranks on one device); never set by the driver
    rehearsal = os.environ.get("RC_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d wants GPU %d, this node shows %d (use --gpus <= the node's GPU count)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    # RC_BENCH_FORCE_DIST=1 (never set by the driver): take the N > 1 code path -- RCCL init, barriers, MAX all-reduce on a device
    # tensor -- with a single rank, which is all the RCCL a one-GPU box can run
    if world > 1 or os.environ.get("RC_BENCH_FORCE_DIST", "0") == "1":
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # forced single-rank run outside torchrun: the rendezvous variables torchrun would have set
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    return torch, dist, world, rank, local_rank, rehearsal


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    _guard_stdout()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver may start `python bench.py --gpus N` exactly as it starts `--gpus 1`: be the launcher, before torch or
        # the GPU is touched in this process
        raise SystemExit(launch_ranks(args, argv))
    if args.dry_run:
        return run_dry(args)
    if args.config == "cfg5":
        return run_cfg5(args)
    return run_cfg3(args)


def run_cfg3(args):
    import numpy as np

    torch, dist, world, rank, local_rank, rehearsal = setup_ranks(args)
    import rusty_compression_amd as rc
    from rusty_compression_amd import _lib

    args.size = args.size or 8192
    args.rank = args.rank or 128
    m = n = args.size
    k, p = args.rank, args.oversample
    l = k + p
    S = max(1, args.streams)
    with_id = not args.no_id
    dt = torch.float64

    # ---- per-stream state: own matrix, own outputs, own context ------------------------
    lanes = []
    seen_streams = set()
    for s in range(S):
        if s < 32:
            st = torch.cuda.Stream()
        else:
            # torch hands out 32 distinct streams per device; further lanes get their own HIP streams (through the
            # library, so that they come from the HIP runtime the library itself is linked against)
            raw = ctypes.c_void_p()
            assert _lib.lib().rc_stream_create(ctypes.c_int32(local_rank), ctypes.byref(raw)) == 0
            st = torch.cuda.ExternalStream(raw.value)
        assert st.cuda_stream not in seen_streams, "duplicate stream handle"
        seen_streams.add(st.cuda_stream)
        with torch.cuda.stream(st):
            ctx = _lib.default_context()
            a = rc.random_gaussian((m, n), rc.Rng(1000 * rank + s + 1), dt)  # resident in HBM
            mk = lambda r, c: torch.empty((r, c), dtype=dt, device="cuda")  # noqa: E731
            bufs = dict(range_q=mk(m, k), u=mk(m, k), s=torch.empty(k, dtype=dt, device="cuda"), vt=mk(k, n))
            if with_id:
                bufs.update(qr_q=mk(m, k), qr_r=mk(k, n), qr_ind=torch.empty(n, dtype=torch.int64, device="cuda"), id_c=mk(m, k), id_z=mk(k, n))
            none = _lib.mat(None)
            out = _lib.rc_rsvd_id_out(
                _lib.mat(bufs["range_q"]), _lib.mat(bufs["u"]), ctypes.c_void_p(bufs["s"].data_ptr()), _lib.mat(bufs["vt"]),
                _lib.mat(bufs["qr_q"]) if with_id else none, _lib.mat(bufs["qr_r"]) if with_id else none,
                ctypes.c_void_p(bufs["qr_ind"].data_ptr()) if with_id else ctypes.c_void_p(None),
                _lib.mat(bufs["id_c"]) if with_id else none, _lib.mat(bufs["id_z"]) if with_id else none)

            def call(ctx=ctx, a=a, out=out, seed=7 + s):
                ctx.call("rc_rsvd_id_f64", _lib.mat(a), ctypes.c_int64(k), ctypes.c_int64(p), _lib.mat(None), ctypes.c_uint64(seed), ctypes.byref(out))

            if not args.no_concurrency_hint:
                ctx.set_option(_lib.RC_OPT_CONCURRENCY_HINT, S)  # S compressions in flight: wide GEMMs stay un-split (no partial slabs)
            call()  # eager warm-up: sizes the workspace arena (required before capture)
            ctx.synchronize()
            # what the replays of the timed region must reproduce bit for bit (checked after it): EVERY output buffer
            eager_ref = {name: t.clone() for name, t in bufs.items()}
            graph = ctypes.c_void_p(None)
            if not args.no_graph:
                ctx.check(_lib.lib().rc_graph_begin_capture(ctx._h))
                call()
                ctx.check(_lib.lib().rc_graph_end_capture(ctx._h, ctypes.byref(graph)))
                # first launch of the executable graph (uploads it to the device) belongs to the lane's set-up, so that a
                # small --warmup does not leave first launches inside the timed region
                ctx.check(_lib.lib().rc_graph_launch(ctx._h, graph))
                ctx.synchronize()
            lanes.append(dict(stream=st, ctx=ctx, a=a, bufs=bufs, out=out, call=call, graph=graph, eager_ref=eager_ref))

    lane_events = []

    def step(i, timed=False):
        ln = lanes[i % S]
        if timed and args.lane_events:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ln["stream"])
        if ln["graph"]:
            ln["ctx"].check(_lib.lib().rc_graph_launch(ln["ctx"]._h, ln["graph"]))
        else:
            ln["call"]()
        if timed and args.lane_events:
            e1.record(ln["stream"])
            lane_events.append((i, e0, e1))

    ctx_arr = (ctypes.c_void_p * S)(*[ln["ctx"]._h.value for ln in lanes])

    def sync_all():
        # rc_synchronize_all: a completion event on EVERY lane's stream first, the waits afterwards.  Measured on this
        # stack (tools/steps_sweep.sh, --lane-events): with more than ~28 streams holding work, stream-by-stream waits
        # return after max(259 ms, work); with an event recorded on every stream before the first wait they return on
        # time.  RC_BENCH_SYNC=ctx reproduces the old behaviour (rc_synchronize context by context).
        if os.environ.get("RC_BENCH_SYNC") == "ctx":
            for ln in lanes:
                ln["ctx"].synchronize()
            return
        st = _lib.lib().rc_synchronize_all(ctx_arr, ctypes.c_int32(S))
        assert st == 0, "rc_synchronize_all failed with status %d" % st

    # ---- sanity of what is being timed (lane 0) -----------------------------------------
    sync_all()
    q0 = lanes[0]["bufs"]["range_q"]
    with torch.cuda.stream(lanes[0]["stream"]):
        gram = rc.dot(q0.t(), q0)
    sync_all()
    orth = float((gram - torch.eye(k, dtype=dt, device="cuda")).abs().max())
    s0 = lanes[0]["bufs"]["s"]
    assert orth < 1e-10 and bool((s0[:-1] >= s0[1:]).all()) and float(s0[-1]) > 0, "bench sanity check failed"

    def profile_samples(ln, lib, n):
        out = []
        # the kernel is timed as a lone launch: with the hint at 1 it splits K to cover the chip (what `roofline` describes);
        # the captured graphs of the timed region keep the launch shape they were recorded with
        ln["ctx"].set_option(_lib.RC_OPT_CONCURRENCY_HINT, 1)
        for _ in range(2):  # untimed: lane's buffers back into the TLBs / caches
            ln["call"]()
        ln["ctx"].synchronize()
        lib.rc_profile_enable(ln["ctx"]._h, 1)
        for _ in range(n):
            lib.rc_profile_reset(ln["ctx"]._h)
            ln["call"]()
            cnt = ctypes.c_int32(0)
            ln["ctx"].check(lib.rc_profile_count(ln["ctx"]._h, ctypes.byref(cnt)))
            one = {}
            for i in range(cnt.value):
                name = ctypes.create_string_buffer(192)
                ms = ctypes.c_double(0)
                calls = ctypes.c_int64(0)
                lib.rc_profile_get(ln["ctx"]._h, i, name, 192, ctypes.byref(ms), ctypes.byref(calls))
                one[name.value.decode()] = ms.value / max(calls.value, 1)
            out.append(one)
        lib.rc_profile_enable(ln["ctx"]._h, 0)
        if not args.no_concurrency_hint:
            ln["ctx"].set_option(_lib.RC_OPT_CONCURRENCY_HINT, S)
        return out

    # ---- warm-up, then K timed steps (a step = one compression on each of the S lanes) ----
    for _ in range(args.warmup):
        for i in range(S):
            step(i)
    sync_all()
    prof_before = profile_samples(lanes[0], _lib.lib(), 4)
    for i in range(S):   # every lane busy again after the single-lane profiling samples
        step(i)
    sync_all()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for i in range(S):
            step(i, True)
    t_issue = time.perf_counter()
    sync_all()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if lane_events and rank == 0:
        ref = lane_events[0][1]
        for i, e0, e1 in lane_events:
            print(f"step {i:4d} lane {i % S:3d} start {ref.elapsed_time(e0):9.3f} ms end {ref.elapsed_time(e1):9.3f} ms", file=sys.stderr)
    if dist is not None:
        dist.barrier()
    # every captured tall-skinny fast path must have certified itself (no fallback exists inside a graph)
    health = [ln["ctx"].get_health() for ln in lanes]
    assert not any(health), f"fast-path certificate failed during the timed region: {health}"
    # the last compression of every lane (a graph replay under full concurrency) against that lane's eager warm-up result:
    # every output buffer (range_q, u, s, vt, qr_q, qr_r, qr_ind, id_c, id_z), bit for bit
    replay_bad = {}
    for i, ln in enumerate(lanes):
        diff = [name for name, t in ln["bufs"].items() if not torch.equal(t, ln["eager_ref"][name])]
        if diff:
            replay_bad[i] = diff
    replay_ok = S - len(replay_bad)
    for ln in lanes:
        ln["eager_ref"] = None   # 2.4 GB of copies back to the allocator
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ... and LIVE, un-traced: every lane runs ONLY the two big products of a compression (sketch, projection: the launches of the
    # timed graphs, un-split) back to back for a few rounds -- the rate the products reach when the chip holds nothing else but
    # products.  (A kernel trace serialises part of the concurrency, so the traced durations above are a LOWER bound of the
    # in-flight cost: this figure is the one to price the pipeline's GEMM share with.)
    in_flight_live = None
    if rank == 0 and world == 1 and not args.no_gemm_lanes and S >= 8:
        try:
            fl_live, _ = work_model(m, n, k, p, with_id)
            pl = []
            for ln in lanes:
                with torch.cuda.stream(ln["stream"]):
                    om_ = rc.random_gaussian((n, l + (l & 1)), rc.Rng(3), dt)[:, :l]
                    y_ = torch.empty((l, m), dtype=dt, device="cuda").t()
                    b_ = torch.empty((k, n), dtype=dt, device="cuda")
                pl.append((om_, y_, b_))

            def products(ln, om_, y_, b_):
                ln["ctx"].call("rc_matmat_f64", _lib.mat(ln["a"]), _lib.mat(om_), _lib.mat(y_))
                ln["ctx"].call("rc_gemm_f64", ctypes.c_int32(1), ctypes.c_int32(0), ctypes.c_double(1.0), _lib.mat(ln["bufs"]["range_q"]), _lib.mat(ln["a"]),
                               ctypes.c_double(0.0), _lib.mat(b_))
            pgraphs = []
            for ln, bufs_ in zip(lanes, pl):   # replayed as graphs like the timed region (same stream -> queue regime)
                with torch.cuda.stream(ln["stream"]):
                    g_ = ctypes.c_void_p(None)
                    if ln["graph"]:
                        products(ln, *bufs_)
                        ln["ctx"].synchronize()
                        ln["ctx"].check(_lib.lib().rc_graph_begin_capture(ln["ctx"]._h))
                        products(ln, *bufs_)
                        ln["ctx"].check(_lib.lib().rc_graph_end_capture(ln["ctx"]._h, ctypes.byref(g_)))
                    pgraphs.append(g_)

            def products_round():
                for ln, bufs_, g_ in zip(lanes, pl, pgraphs):
                    with torch.cuda.stream(ln["stream"]):
                        if g_:
                            ln["ctx"].check(_lib.lib().rc_graph_launch(ln["ctx"]._h, g_))
                        else:
                            products(ln, *bufs_)
            products_round()
            sync_all()
            rounds = 8
            tg0 = time.perf_counter()
            for _ in range(rounds):
                products_round()
            sync_all()
            tp = (time.perf_counter() - tg0) / (rounds * S)
            gf = fl_live["sketch_gemm"] + fl_live["project_gemm"]
            in_flight_live = {"what": "S lanes x (sketch + projection), un-split launches, nothing else on the chip, no trace", "lanes": S, "rounds": rounds,
                              "ms_per_pair_chip_time": round(tp * 1e3, 4), "tflops": round(gf / tp / 1e12, 2), "frac": round(gf / tp / 1e12 / F64_MFMA_PEAK_TFLOPS, 4),
                              "share_of_a_compression": round(tp / (elapsed / (args.steps * S)), 4)}
            for ln, g_ in zip(lanes, pgraphs):
                if g_:
                    _lib.lib().rc_graph_destroy(ln["ctx"]._h, g_)
            del pl
        except Exception as e:  # a diagnostic: never fails the bench line
            in_flight_live = {"error": repr(e)}
    # ---- second figure: every compression first re-uploads its A from pinned host memory on its own
    # lane (the reference's API takes host ndarrays); one round over all lanes, never `value` ------------------------
    h2d = None
    if rank == 0 and world == 1 and not args.no_h2d:
        a_pinned = torch.empty((m, n), dtype=dt, pin_memory=True)
        a_pinned.copy_(lanes[0]["a"])
        sync_all()
        th0 = time.perf_counter()
        for i in range(S):
            with torch.cuda.stream(lanes[i]["stream"]):
                lanes[i]["a"].copy_(a_pinned, non_blocking=True)
            step(i)
        sync_all()
        torch.cuda.synchronize()
        th = time.perf_counter() - th0
        h2d = {"value_including_h2d": round(S / th, 3), "unit": "compressions/s", "compressions": S,
               "h2d_gb_per_s": round(S * 8.0 * m * n / th / 1e9, 2),
               "note": "A (%.0f MiB) copied from pinned host memory to the lane's device buffer before each compression, copies and compressions of different lanes overlapping" % (8.0 * m * n / 2 ** 20)}

    # ---- stage / kernel timers: HIP events on lane 0's own stream, eager launches --------
    # one (reset -> call -> read) cycle per sample, so every launch is seen individually; the samples taken before
    # the timed region (prof_before, below the warm-up) and after it are both reported and the MEDIAN is used:
    # right after minutes of sustained f64 MFMA load single eager launches sporadically run several times slower
    # (power management), which an average would fold into the kernel's figure.
    lib = _lib.lib()
    ln = lanes[0]
    samples_after = profile_samples(ln, lib, 8)
    samples = prof_before + samples_after
    prof = {}
    for name in set().union(*[set(x) for x in samples]):
        vals = sorted(x[name] for x in samples if name in x)
        prof[name] = (vals[len(vals) // 2], 1)

    if args.profile_concurrent and rank == 0:
        for l2 in lanes:
            lib.rc_profile_enable(l2["ctx"]._h, 1)
            lib.rc_profile_reset(l2["ctx"]._h)
        tq0 = time.perf_counter()
        for _ in range(4):
            for l2 in lanes:
                l2["call"]()
        sync_all()
        tq = (time.perf_counter() - tq0) / (4 * S)
        agg = {}
        for l2 in lanes:
            ln_cnt = ctypes.c_int32(0)
            l2["ctx"].check(lib.rc_profile_count(l2["ctx"]._h, ctypes.byref(ln_cnt)))
            for i in range(ln_cnt.value):
                name = ctypes.create_string_buffer(192)
                ms = ctypes.c_double(0)
                calls = ctypes.c_int64(0)
                lib.rc_profile_get(l2["ctx"]._h, i, name, 192, ctypes.byref(ms), ctypes.byref(calls))
                e = agg.setdefault(name.value.decode(), [0.0, 0])
                e[0] += ms.value
                e[1] += calls.value
            lib.rc_profile_enable(l2["ctx"]._h, 0)
        print("concurrent eager pass: %.3f ms/step, %d streams" % (tq * 1e3, S), file=sys.stderr)
        for kk, (ms_, c_) in sorted(agg.items()):
            single = prof.get(kk, (0.0, 0))
            s_ms = single[0] / max(single[1], 1)
            print("  %-64s %9.3f ms   (single-stream %8.3f)  x%.1f" % (kk, ms_ / max(c_, 1), s_ms, (ms_ / max(c_, 1)) / s_ms if s_ms else 0), file=sys.stderr)

    fl, by = work_model(m, n, k, p, with_id)
    total_flops = sum(fl.values())

    def sketch_kernel_name():
        # the instantiation the library launches for the sketch shape: issue that product once on lane 0 and ask
        fn = lib.rc_last_gemm_kernel_name
        fn.restype = ctypes.c_char_p
        fn.argtypes = [ctypes.c_void_p]
        with torch.cuda.stream(lanes[0]["stream"]):
            # the same operand layout as inside the library (even leading dimensions: 16-byte vector staging), otherwise the
            # scalar-staging instantiation would answer
            om = torch.empty((n, l + (l & 1)), dtype=dt, device="cuda")[:, :l]
            y = torch.empty((m, l + (l & 1)), dtype=dt, device="cuda")[:, :l]
            lanes[0]["ctx"].call("rc_matmat_f64", _lib.mat(lanes[0]["a"]), _lib.mat(om), _lib.mat(y))
        lanes[0]["ctx"].synchronize()
        nm = fn(lanes[0]["ctx"]._h)
        return nm.decode() if nm else None

    # the library runs the skinny-N sketch as the transposed problem (M = l), the timer carries that shape
    key = f"kernel:k_gemm_mfma<f64> M={l} N={m} K={n}"
    if key not in prof:
        key = f"kernel:k_gemm_mfma<f64> M={m} N={l} K={n}"
    roof = None
    if key in prof and prof[key][1] > 0:
        ms_gemm = prof[key][0] / prof[key][1]
        # the deterministic split-K slab reduction is part of the same product: it is timed with it
        red = [kk for kk in prof if kk.startswith("kernel:k_splitk_reduce M=%d N=%d " % (l, m)) or kk.startswith("kernel:k_splitk_reduce M=%d N=%d " % (m, l))]
        ms_reduce = sum(prof[kk][0] / prof[kk][1] for kk in red)
        ms_launch = ms_gemm + ms_reduce
        achieved = fl["sketch_gemm"] / (ms_launch * 1e-3) / 1e12
        # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, tools/pmc_gemm.sh); the
        # record names the kernel instantiation it was taken on, so a stale figure is detectable: it is reported only when
        # that instantiation is the one the library still launches for this shape (rc_gemm_kernel_name)
        traffic, traffic_src = None, None
        for tname in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if not os.path.exists(tpath):
                continue
            try:
                rec = json.load(open(tpath))
            except Exception:
                continue
            cur = sketch_kernel_name()
            norm = lambda x: x.replace(" ", "") if x else x
            per_kernel = {norm(kn): v for kn, v in rec.get("kernels", {}).items()}
            if cur is not None and norm(cur) in per_kernel:
                # one record per kernel instantiation (tools/gpu_round2_profiles.sh writes them from the PMC passes)
                ent = per_kernel[norm(cur)]
                traffic = ent.get("bytes_per_launch")
                traffic_src = {"file": "profiles/" + tname, "source": rec.get("source"), "kernel": cur, "fetch_bytes": ent.get("fetch_bytes"),
                               "write_bytes": ent.get("write_bytes"), "launches": ent.get("launches")}
            else:
                traffic_src = {"file": "profiles/" + tname, "source": rec.get("source"), "kernel": rec.get("kernel")}
                if rec.get("kernel") is None or cur is None or norm(rec.get("kernel")) == norm(cur):
                    traffic = rec.get("k_gemm_mfma_sketch_bytes_per_launch")
                else:
                    traffic_src["stale"] = "PMC pass was taken on %s, the library now launches %s" % (rec.get("kernel"), cur)
            break
        # What the TIMED REGION runs is not this lone launch: with S lanes in flight the big products are not split (32 workgroups
        # each, the other lanes fill the chip).  Their in-flight cost comes from a kernel trace of this very command (a committed
        # record written by tools/gpu_round3_profiles.sh, named per kernel instantiation): CU-time = duration x workgroups / 256.
        in_flight = None
        ipath = os.path.join(ROOT, "profiles", "r03_in_flight.json")
        cur_name = sketch_kernel_name()
        if os.path.exists(ipath) and cur_name:
            try:
                rec = json.load(open(ipath))
                ent = [e for e in rec.get("kernels", {}).get(cur_name.replace(" ", ""), []) if e.get("workgroups", 0) < 256]
                if ent:
                    e = max(ent, key=lambda x: x.get("launches", 0))
                    cu_ms = e["cu_time_us"] * 1e-3
                    in_flight = {"kernel": cur_name, "workgroups": e["workgroups"], "launches_in_trace": e["launches"], "avg_launch_us": e["avg_us"],
                                 "cu_time_ms": round(cu_ms, 4), "tflops_per_chip_equivalent": round(fl["sketch_gemm"] / (cu_ms * 1e-3) / 1e12, 2),
                                 "frac": round(fl["sketch_gemm"] / (cu_ms * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS, 4),
                                 "source": "profiles/r03_in_flight.json: " + rec.get("source", "")}
            except Exception:
                in_flight = None
        roof = {"bound": "mfma", "kernel": "%s (sketch Y = A*Omega, %dx%dx%d, run as Y^T = Omega^T A^T)" % (sketch_kernel_name() or "k_gemm_f64", m, l, n), "achieved": round(achieved, 3),
                "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / F64_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src, "in_flight": in_flight, "in_flight_live": in_flight_live,
                "avg_launch_ms": round(ms_launch, 4), "gemm_ms": round(ms_gemm, 4), "splitk_reduce_ms": round(ms_reduce, 4), "launches_timed": len(samples),
                "launch_ms_samples_before_timed_region": [round(x[key], 4) for x in prof_before if key in x],
                "launch_ms_samples_after_timed_region": [round(x[key], 4) for x in samples_after if key in x],
                "flops_per_launch": fl["sketch_gemm"], "hbm_gbs_algorithmic": round(8.0 * m * n / (ms_launch * 1e-3) / 1e9, 1),
                "method": "HIP events on the launching stream around each launch (rc_profile_*): median of the eager single-stream samples taken right before and right after the timed region"}

    # ---- CPU baseline: the oracle in the reference's call shape on this box's host cores --
    cpu = cpu_gemm = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_lapack as o

        cores, cores_source = host_threads()
        a_h = lanes[0]["a"].cpu().numpy()
        om_h = np.random.default_rng(0).standard_normal((n, l))
        tc0 = time.perf_counter()
        for _ in range(args.cpu_baseline_reps):
            o.rsvd_id_reference_shape(a_h, om_h, k, faithful=True)
        tc = (time.perf_counter() - tc0) / args.cpu_baseline_reps
        tg0 = time.perf_counter()
        o.rsvd_id_reference_shape(a_h, om_h, k, faithful=False)
        tg = time.perf_counter() - tg0
        cpu_gemm = {"value": round(1.0 / tg, 4), "unit": "compressions/s", "cores": cores, "cores_source": cores_source, "kind": "port",
                    "sample": "1 compression of the same matrix with the two operator products as single GEMMs (numpy @ / OpenBLAS dgemm) "
                              "instead of the reference's per-column gemv loops; everything else as cpu_baseline; %.2f s" % tg,
                    "seconds_per_compression": round(tg, 3)}
        cpu = {"value": round(1.0 / tc, 4), "unit": "compressions/s", "cores": cores, "cores_source": cores_source, "kind": "port",
               "sample": "%d compression(s) of the same 8192x8192 f64 matrix, rSVD+ID, reference call shape "
                         "(per-column gemv loops for A*Omega and A^H*Q, ?geqp3+?orgqr, ?gesdd, per-column ?trtrs) via oracle/ref_lapack.py "
                         "(SciPy LAPACK/OpenBLAS), %.2f s each" % (args.cpu_baseline_reps, tc),
               "seconds_per_compression": round(tc, 3)}

    if rank == 0:
        steps_total = args.steps * S * world   # compressions in the timed region, all ranks
        value = steps_total / elapsed
        stage_ms = {kk: round(v[0] / max(v[1], 1), 4) for kk, v in sorted(prof.items()) if kk.startswith(("stage:", "op:", "info:"))}
        line = {
            "metric": "GB/s + compressions/sec, 8192x8192 f64 rank-128 rSVD+ID",
            "value": round(value, 3),
            "unit": "compressions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "host_issue_ms_per_step": round((t_issue - t0) / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "cfg3: %dx%d f64 dense N(0,1), rank-%d rSVD%s, p=%d (BASELINE.json configs[2])" % (m, n, k, "+ID" if with_id else "", p),
                       "compressions_per_step": S, "compressions_timed": steps_total,
                       "concurrency_hint": 1 if args.no_concurrency_hint else S,
                       "streams_per_gpu": S, "hipgraph": not args.no_graph, "parallelism": "independent matrices, %d per GPU in flight" % S},
            "gb_per_s": round(value * by / 1e9, 2),
            "tflops_algorithmic": round(value * total_flops / 1e12, 3),
            "frac_of_f64_mfma_peak_whole_pipeline": round(value * total_flops / 1e12 / (F64_MFMA_PEAK_TFLOPS * world), 4),
            "bytes_per_compression": by,
            "flops_per_compression": total_flops,
            "roofline": roof,
            "cpu_baseline": cpu,
            "cpu_baseline_gemm_form": cpu_gemm,
            "value_including_h2d": h2d,
            "stage_ms_single_stream_eager": stage_ms,
            "timed_results_check": {"lanes_whose_last_replay_equals_their_eager_result_bitwise": replay_ok, "lanes": S,
                                    "compared": "every output buffer: " + ", ".join(sorted(lanes[0]["bufs"])),
                                    "lanes_that_differ": {str(i): v for i, v in sorted(replay_bad.items())}},
        }
        if replay_ok != S:
            line["invalid"] = "the timed replays of %d lane(s) did not reproduce their eager result" % (S - replay_ok)
        _emit(json.dumps(line))

    for ln in lanes:
        if ln["graph"]:
            _lib.lib().rc_graph_destroy(ln["ctx"]._h, ln["graph"])
    if dist is not None:
        dist.destroy_process_group()
    if replay_ok != S:
        print("bench.py: timed_results_check FAILED on rank %d: %s" % (rank, replay_bad), file=sys.stderr)
        raise SystemExit(4)



F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X f32 matrix peak (vendor datasheet value)


def run_cfg5(args):
    """BASELINE.json configs[4]: a batch of independent 4096 x 4096 f32 matrices, rank-64 column ID
    (QR::compute_from -> compress(RANK(k)) -> column_id, examples/interpolative_decomposition.rs:25-32), sharded by matrix
    (rc_batch_shard_range: 8 per GPU, 64 on 8 GPUs -- weak scaling), each rank's share through ONE rc_batch_column_id_f32 call,
    then the ONLY collective of the path: the gather of the packed factor blocks to rank 0 (rc_comm_gather, RCCL over xGMI).
    One STEP = one such batch + gather on every rank; value = matrices/s over all ranks."""
    import numpy as np

    torch, dist, world, rank, local_rank, rehearsal = setup_ranks(args)
    import rusty_compression_amd as rc
    from rusty_compression_amd import _lib, batch

    m = n = args.size or 4096
    k = args.rank or 64
    nloc = max(1, args.matrices_per_gpu)
    total = nloc * world
    start, count = ctypes.c_int64(0), ctypes.c_int64(0)
    assert _lib.lib().rc_batch_shard_range(ctypes.c_int64(total), ctypes.c_int32(world), ctypes.c_int32(rank), ctypes.byref(start), ctypes.byref(count)) == 0
    mine = list(range(start.value, start.value + count.value))
    assert len(mine) == nloc
    mats = [rc.random_gaussian((m, n), rc.Rng(500 + i), torch.float32) for i in mine]   # seeds 500..563
    torch.cuda.synchronize()
    per = batch.packed_bytes(m, n, k, 4)
    comm = None
    if dist is not None and world > 1 and not rehearsal:
        comm = batch.Comm.from_process_group(None, local_rank)
    elif os.environ.get("RC_BENCH_FORCE_DIST", "0") == "1" and world == 1:
        comm = batch.Comm(1, 0, batch.Comm.unique_id(), local_rank)   # one-rank RCCL communicator: the entry points themselves

    gather_s = []

    def gather(packed):
        t0 = time.perf_counter()
        if comm is not None:
            got = comm.gather(packed, 0)   # grouped ncclSend / ncclRecv on the context's stream + rc_synchronize
        elif dist is not None and world > 1:  # rehearsal: all ranks on one device, gloo on host copies
            src = packed.cpu()
            parts = [torch.empty_like(src) for _ in range(world)] if rank == 0 else None
            dist.gather(src, parts, dst=0)
            got = torch.cat(parts).to(packed.device) if rank == 0 else None
        else:
            got = packed
        gather_s.append(time.perf_counter() - t0)
        return got

    def step():
        packed = batch.batch_column_id_packed(mats, k, lanes=args.batch_lanes)
        return gather(packed)

    first = step()
    first = first.clone() if first is not None else None
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    gather_s.clear()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if dist is not None:
        dist.barrier()
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- what was timed is checked on rank 0: the last gathered batch equals the first bit for bit, and every matrix of every rank
    # satisfies the ID's defining identities against the matrix regenerated from its seed -------------------------------------------
    check = None
    bad = []
    if rank == 0:
        same = bool(torch.equal(first, last))
        facs = batch.unpack_factors(last, total, m, n, k, torch.float32)
        for i, (c, z, ind) in enumerate(facs):
            a = mats[i - mine[0]] if mine[0] <= i < mine[0] + nloc else rc.random_gaussian((m, n), rc.Rng(500 + i), torch.float32)
            is_perm = bool(torch.equal(torch.sort(ind).values, torch.arange(n, device=ind.device)))
            if not is_perm:
                bad.append((i, "col_ind is not a permutation"))
                continue
            sel = a[:, ind[:k]]
            e_c = float((c - sel).norm() / sel.norm())
            e_i = float((z[:, ind[:k]] - torch.eye(k, dtype=z.dtype, device=z.device)).abs().max())
            if not (e_c <= 1e-4 and e_i <= 1e-5):
                bad.append((i, "C vs A[:, col_ind[:k]] %.2e, Z[:, col_ind[:k]] vs I %.2e" % (e_c, e_i)))
        check = {"matrices_checked": total, "identities": "col_ind a permutation; C = A[:, col_ind[:k]] (rel. Frobenius <= 1e-4); Z[:, col_ind[:k]] = I (<= 1e-5)",
                 "failed": [list(b) for b in bad], "last_timed_batch_equals_first_bitwise": same}
        if not same:
            bad.append((-1, "the last timed batch differs from the first one"))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_lapack as o

        cores, cores_source = host_threads()
        a_h = mats[0].cpu().numpy()
        tc0 = time.perf_counter()
        o.QR.compute_from(a_h).compress("RANK", k).column_id()
        tc = time.perf_counter() - tc0
        cpu = {"value": round(1.0 / tc, 4), "unit": "matrices/s", "cores": cores, "cores_source": cores_source, "kind": "port",
               "sample": "1 matrix of the batch, reference call sequence QR::compute_from (full sgeqp3 + sorgqr of 4096 x 4096) -> compress(RANK(64)) -> "
                         "column_id (per-column strtrs) via oracle/ref_lapack.py (SciPy LAPACK/OpenBLAS), %.2f s" % tc,
               "seconds_per_matrix": round(tc, 3)}

    if rank == 0:
        n_mat = args.steps * total
        value = n_mat / elapsed
        flops = 4.0 * m * n * k - 2.0 * (m + n) * k * k + (4.0 / 3.0) * k ** 3 + 1.0 * k * k * (n - k)   # truncated QRCP + TRSM
        by = 4.0 * m * n + 4.0 * (m * k + k * n) + 8.0 * n                                              # A once + C, Z, col_ind
        g_ms = sorted(x * 1e3 for x in gather_s)
        line = {
            "metric": "matrices/sec, batch of 4096x4096 f32 rank-64 column ID, 8 per GPU + RCCL gather of the factors",
            "value": round(value, 2), "unit": "matrices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg5: batch of %d independent %dx%d f32 dense N(0,1) matrices (seeds 500..%d), rank-%d column ID, %d per GPU "
                                   "(BASELINE.json configs[4])" % (total, m, n, 499 + total, k, nloc),
                       "matrices_per_step": total, "matrices_timed": n_mat, "batch_lanes": args.batch_lanes,
                       "parallelism": "independent matrices, rc_batch_shard_range; gather: %s" % (
                           "rc_comm_gather (RCCL)" if comm is not None else ("gloo on host copies (rehearsal)" if world > 1 else "single rank, none"))},
            "gather_ms_median": round(g_ms[len(g_ms) // 2], 4) if g_ms else None,
            "gather_bytes_per_rank": nloc * per,
            "gb_per_s": round(value * by / 1e9, 2),
            "tflops_algorithmic": round(value * flops / 1e12, 3),
            "frac_of_f32_mfma_peak_whole_pipeline": round(value * flops / 1e12 / (F32_MFMA_PEAK_TFLOPS * world), 4),
            "bytes_per_matrix": by, "flops_per_matrix": flops,
            "roofline": {"bound": "hbm", "kernel": "rc_batch_column_id_f32 (the whole batch call; its kernels are latency-bound pivot panels + streaming panel ends)",
                         "achieved": round(value / world * by / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(value / world * by / 1e9 / HBM_PEAK_GBS, 5),
                         "traffic": None, "method": "algorithmic bytes per matrix x matrices per second per GPU (host clock around the timed region)"},
            "cpu_baseline": cpu,
            "timed_results_check": check,
        }
        if bad:
            line["invalid"] = "result check failed: %s" % bad
        _emit(json.dumps(line))
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()
    if bad:
        raise SystemExit(4)


if __name__ == "__main__":
    main()
