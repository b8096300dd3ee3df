"""CPU tests of the N > 1 path (gloo, world_size 2): the batch is sharded by matrix, each rank
compresses its own block with no data-path collective, and ONE gather brings the packed factor
blocks to rank 0 in global order.  The compute function is injected (here: the CPU oracle) --
what is under test is the partition / pack / gather plumbing that bench-scale runs use over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rusty_compression_amd import batch


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_column_id(a, k):
    from oracle import ref_lapack as o

    cid = o.QR.compute_from(a.numpy()).compress("RANK", k).column_id()
    return torch.from_numpy(cid.c), torch.from_numpy(cid.z), torch.from_numpy(cid.col_ind)


def _make_batch(n_items, m, n, dtype):
    rng = np.random.default_rng(123)
    return [torch.from_numpy(rng.standard_normal((m, n)).astype(dtype)) for _ in range(n_items)]


def _worker(rank, world, port, n_items, k, dtype, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mats = _make_batch(n_items, 24, 40, dtype)
        mine = [mats[i] for i in batch.shard_range(n_items, world, rank)]
        out = batch.batch_column_id(mine, k, compute=_oracle_column_id)
        if rank == 0:
            ok = len(out) == n_items
            for i, (c, z, ind) in enumerate(out):
                rc_, rz, rind = _oracle_column_id(mats[i], k)
                ok = ok and torch.equal(c, rc_) and torch.equal(z, rz) and torch.equal(ind, rind)
            ret["ok"] = bool(ok)
        else:
            ret[f"none{rank}"] = out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sharded_batch_gathers_in_global_order(dtype):
    world, n_items, k = 2, 6, 5
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_items, k, dtype, ret), nprocs=world, join=True)
    assert ret.get("ok") is True and ret.get("none1") is True


def test_shard_range_is_a_contiguous_partition():
    for n_items in (64, 10, 3, 0):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                seen += list(batch.shard_range(n_items, world, r))
            assert seen == list(range(n_items))
    assert list(batch.shard_range(64, 8, 3)) == list(range(24, 32))  # 8 per GPU, matrix i -> GPU i // 8


def test_pack_unpack_round_trip_is_exact():
    rng = np.random.default_rng(0)
    for dtype in (torch.float64, torch.float32):
        fs = [(torch.from_numpy(rng.standard_normal((7, 3))).to(dtype), torch.from_numpy(rng.standard_normal((3, 10))).to(dtype),
               torch.from_numpy(rng.permutation(10) + 2 ** 40)) for _ in range(3)]
        got = batch.unpack_factors(batch.pack_factors(fs), 3, 7, 10, 3)
        for (c, z, i), (c2, z2, i2) in zip(fs, got):
            assert torch.equal(c, c2) and torch.equal(z, z2) and torch.equal(i, i2)


def test_c_abi_partition_and_packed_size_agree_with_the_python_mirror():
    """rc_batch_shard_range / rc_batch_packed_bytes (pure host arithmetic of the C ABI) against batch.shard_range /
    batch.packed_bytes, including sizes whose factor block is not a multiple of 8 bytes."""
    import ctypes

    from rusty_compression_amd import _lib

    lib = _lib.lib()
    lib.rc_batch_packed_bytes.restype = ctypes.c_size_t
    for n_items in (64, 10, 3, 0):
        for world in (1, 2, 3, 8):
            for r in range(world):
                s, c = ctypes.c_int64(-1), ctypes.c_int64(-1)
                assert lib.rc_batch_shard_range(ctypes.c_int64(n_items), world, r, ctypes.byref(s), ctypes.byref(c)) == 0
                rng = batch.shard_range(n_items, world, r)
                assert (s.value, c.value) == (rng.start, len(rng))
    s, c = ctypes.c_int64(), ctypes.c_int64()
    assert lib.rc_batch_shard_range(ctypes.c_int64(5), 2, 2, ctypes.byref(s), ctypes.byref(c)) == _lib.RC_INVALID_ARGUMENT
    for (m, n, k, es) in ((4096, 4096, 64, 4), (7, 10, 3, 4), (7, 10, 3, 8), (5, 3, 1, 4), (1, 1, 1, 4)):
        assert lib.rc_batch_packed_bytes(ctypes.c_int64(m), ctypes.c_int64(n), ctypes.c_int64(k), ctypes.c_int32(es)) == batch.packed_bytes(m, n, k, es)
    # the packed layout: C | Z | pad | col_ind, read back through unpack_factors
    c_, z_, i_ = torch.arange(15, dtype=torch.float32).reshape(5, 3), torch.arange(9, dtype=torch.float32).reshape(3, 3) + 100, torch.tensor([2, 0, 1])
    buf = batch.pack_factors([(c_[:, :1].contiguous(), z_[:1].contiguous(), i_)])
    assert buf.numel() == batch.packed_bytes(5, 3, 1, 4) and buf.numel() % 8 == 0
    (c2, z2, i2), = batch.unpack_factors(buf, 1, 5, 3, 1, torch.float32)
    assert torch.equal(c2, c_[:, :1]) and torch.equal(z2, z_[:1]) and torch.equal(i2, i_)


# ---- one matrix sharded by rows (rusty_compression_amd/sharded.py): collective plumbing + TSQR algebra with the oracle injected ----
class _OracleOps:
    """CPU stand-ins for the C-ABI steps (oracle/ref_lapack.py on torch CPU tensors): only the tests do this."""

    @staticmethod
    def prepare(a):
        return a

    @staticmethod
    def random_gaussian(shape, seed, like):
        return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape)).to(like.dtype)

    @staticmethod
    def matmat(a, x):
        return a @ x

    @staticmethod
    def conj_matmat(a, x):
        return a.t() @ x

    @staticmethod
    def dot(a, b):
        return a @ b

    @staticmethod
    def pivoted_qr(a):
        from oracle import ref_lapack as o

        q, r, ind = o.pivoted_qr(a.numpy())
        return torch.from_numpy(q), torch.from_numpy(r), torch.from_numpy(ind.astype(np.int64))

    @staticmethod
    def compute_svd(a):
        from oracle import ref_lapack as o

        u, s_, vt = o.compute_svd(a.numpy())
        return torch.from_numpy(u), torch.from_numpy(s_), torch.from_numpy(vt)

    @staticmethod
    def column_id(q, r, ind):
        from oracle import ref_lapack as o

        cid = o.QR(q.numpy(), r.numpy(), ind.numpy()).column_id()
        return torch.from_numpy(cid.c), torch.from_numpy(cid.z)


def _sharded_matrix(m, n):
    rng = np.random.default_rng(77)
    r = 40
    sig = np.logspace(0, -8, r)
    return torch.from_numpy((rng.standard_normal((m, r)) * sig) @ rng.standard_normal((r, n)))


def _sharded_worker(rank, world, port, ret):
    from rusty_compression_amd import sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a = _sharded_matrix(300, 120)
        rows = np.array_split(np.arange(300), world)[rank]
        res = sharded.rsvd_id_row_sharded(a[int(rows[0]):int(rows[-1]) + 1], 24, 6, 5, ops=_OracleOps)
        ret[rank] = {f: getattr(res, f).numpy() for f in ("range_q", "u", "s", "vt", "qr_q", "r", "ind", "c", "z")}
    finally:
        dist.destroy_process_group()


def test_row_sharded_pipeline_plumbing_with_the_oracle_injected():
    from rusty_compression_amd import sharded

    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sharded_worker, args=(world, port, ret), nprocs=world, join=True)
    parts = [ret[r] for r in range(world)]
    for f in ("s", "vt", "r", "ind", "z"):
        assert np.array_equal(parts[0][f], parts[1][f]), f   # replicated outputs: the same bits on every rank
    a = _sharded_matrix(300, 120)
    one = sharded.rsvd_id_row_sharded(a, 24, 6, 5, ops=_OracleOps)   # no process group here: world 1
    u = np.concatenate([q["u"] for q in parts])
    rq = np.concatenate([q["range_q"] for q in parts])
    c = np.concatenate([q["c"] for q in parts])
    an = a.numpy()
    assert np.abs(rq.T @ rq - np.eye(24)).max() <= 1e-12 and np.abs(u.T @ u - np.eye(24)).max() <= 1e-12
    assert np.allclose(parts[0]["s"], one.s.numpy(), rtol=1e-10, atol=0)
    p2, p1 = rq @ (rq.T @ an), one.range_q.numpy() @ (one.range_q.numpy().T @ an)
    assert np.linalg.norm(p2 - p1) <= 1e-9 * np.linalg.norm(an)
    assert np.array_equal(parts[0]["ind"][:12], one.ind.numpy()[:12])
    e2 = np.linalg.norm(c @ parts[0]["z"] - an) / np.linalg.norm(an)
    e1 = np.linalg.norm(one.c.numpy() @ one.z.numpy() - an) / np.linalg.norm(an)
    assert abs(e2 - e1) <= 1e-9 and e2 < 1e-3


def _uneven_worker(rank, world, port, ret):
    from rusty_compression_amd import sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a = _sharded_matrix(300, 120)
        block = a[:280] if rank == 0 else a[280:]   # rank 1 holds 20 rows < k + p = 30
        try:
            sharded.rsvd_id_row_sharded(block, 24, 6, 5, ops=_OracleOps)
            ret[rank] = "no error"
        except AssertionError as e:
            ret[rank] = "AssertionError: " + str(e)
    finally:
        dist.destroy_process_group()


def test_row_sharded_precondition_fails_on_every_rank_together():
    """a rank whose block is too short must not raise alone while its peers wait in the all-gather."""
    world = 2
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_uneven_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0].startswith("AssertionError") and ret[1].startswith("AssertionError"), dict(ret)
    assert "this one has 20" in ret[1] and "this one has" not in ret[0]


# ---- bench.py as its own launcher: `python bench.py --gpus N` must produce N ranks ----------------------------
def _run_bench(argv, env_extra=None, timeout=180):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_2_starts_two_ranks_by_itself_and_prints_one_line():
    import json

    r = _run_bench(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True and rec["value"] is None and rec["steps"] == 2
    # rank 1 sleeps twice as long as rank 0 per step: the reported time is the MAX over the ranks
    assert rec["ms_per_step"] >= 19.0


def test_bench_under_an_external_launcher_does_not_start_ranks_again():
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.strip().startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2, r.stdout


def test_bench_launcher_reports_a_dead_rank_and_ends_the_others():
    r = _run_bench(["--gpus", "2", "--dry-run", "--steps", "1"], {"RC_BENCH_DRY_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
    assert not [x for x in r.stdout.splitlines() if x.strip().startswith("{")]   # no line from a run that lost a rank


def test_bench_cfg5_is_a_selectable_config_with_the_batch_knobs():
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    a = mod.parse_args(["--config", "cfg5", "--gpus", "8"])
    assert a.config == "cfg5" and a.matrices_per_gpu == 8 and a.gpus == 8   # 8 per GPU, 64 on 8 GPUs (BASELINE.json configs[4])
    assert mod.parse_args([]).config == "cfg3" and mod.parse_args([]).gpus == 1
