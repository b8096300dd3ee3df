"""One matrix sharded by ROWS over the GPUs of a node: the randomized SVD + column ID of
A = [A_0; A_1; ...; A_{W-1}], rank r holding the block A_r (m_r x n, m_r >= k + p), for a matrix that does not fit one GPU or
whose latency matters more than throughput.  Everything heavy stays local; what crosses xGMI is small:

    Omega (n x l)            the same on every rank (Philox stream of the shared seed): nothing is sent
    Y_r = A_r Omega          local GEMM                                   (reference: src/random_sampling.rs:103-118)
    Y_r P_r = Q_r R_r        local pivoted QR; S_r = R_r P_r^T (l x l)
    all-gather S_r           W l^2 numbers;  S = [S_0; ...; S_{W-1}]
    S P = Q_S R              pivoted QR of the small stack, redundantly on every rank (same input bits, same result)
                             => Y P = blockdiag(Q_r) Q_S R is THE pivoted QR of Y (TSQR: column norms and inner products of S
                             are those of Y, so pivots and R are ?geqp3's of the whole Y)
    range_r = Q_r Q_S[r-th block, :k]                                      local GEMM, rows of the range basis
    B = sum_r range_r^H A_r  local GEMM + ONE all-reduce of k x n          (src/svd.rs:171-183, src/qr.rs:311-323)
    SVD(B), pivoted QR(B), column ID coefficients                          redundantly on every rank (k x n: small)
    U_r = range_r U_b,  C_r = (range_r Q_b) R11                            local GEMMs: rows of U and of the ID's column matrix

Outputs: row-sharded `range_q`, `u`, `qr_q`, `c`; replicated `s`, `vt`, `r`, `ind`, `z`.

The product path is ONE C-ABI call per rank, `rc_rsvd_id_row_sharded_*` (include/rusty_compression_amd.h), with an `rc_comm`
built from the torch.distributed group: for an "nccl" group the library's own RCCL communicator (its unique id travels through
the group once), for a gloo group (the tests: several ranks on one GPU) a host communicator whose two callbacks run the group's
collectives on the staged host copies.  The step-by-step composition below (`ops=`) is the same algebra spelled with the
one-matrix calls; tests/test_dist_cpu.py injects the CPU oracle there to check plumbing and TSQR algebra without a GPU, and the
GPU test runs it beside the native call.  No CPU fallback in either."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import ctypes
import traceback

import torch


class DeviceOps:
    """The compute steps of the sharded pipeline on the GPU of this rank, through the C ABI (the default).  tests/test_dist_cpu.py
    injects the CPU oracle here to exercise the collective plumbing and the TSQR algebra without a GPU -- the same arrangement as
    `batch.batch_column_id(compute=...)`; the product path never does."""

    @staticmethod
    def prepare(a_local):
        from .types import as_device

        return as_device(a_local)

    @staticmethod
    def random_gaussian(shape, seed, like):
        from .random_matrix import Rng, random_gaussian

        return random_gaussian(shape, Rng(seed), like.dtype)

    @staticmethod
    def matmat(a, x):
        from .types import matmat

        return matmat(a, x)

    @staticmethod
    def conj_matmat(a, x):
        from .types import conj_matmat

        return conj_matmat(a, x)

    @staticmethod
    def dot(a, b):
        from .types import dot

        return dot(a, b)

    @staticmethod
    def pivoted_qr(a):
        from .qr import pivoted_qr

        return pivoted_qr(a)

    @staticmethod
    def compute_svd(a):
        from .svd import compute_svd

        return compute_svd(a)

    @staticmethod
    def column_id(q, r, ind):
        from .qr import QR

        cid = QR(q, r, ind).column_id()
        return cid.c, cid.z


@dataclass
class ShardedRsvdId:
    range_q: torch.Tensor  # m_r x k   (rows of this rank)
    u: torch.Tensor        # m_r x k
    s: torch.Tensor        # k
    vt: torch.Tensor       # k x n
    qr_q: torch.Tensor     # m_r x k
    r: torch.Tensor        # k x n
    ind: torch.Tensor      # n
    c: Optional[torch.Tensor]  # m_r x k   (A[:, ind[:k]] restricted to this rank's rows, as Q R11)
    z: Optional[torch.Tensor]  # k x n


def _world(group):
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0, None
    return dist.get_world_size(group), dist.get_rank(group), dist


def _host_staged(dist, group) -> bool:
    return dist.get_backend(group) == "gloo"


def all_gather_rows(x: torch.Tensor, group=None) -> torch.Tensor:
    """[x_0; x_1; ...] of equally shaped blocks, identical on every rank."""
    world, _, dist = _world(group)
    if world == 1:
        return x.contiguous()
    src = x.contiguous()
    if _host_staged(dist, group):
        src = src.cpu()
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src, group=group)
    return out.to(x.device)


def all_reduce_sum(x: torch.Tensor, group=None) -> torch.Tensor:
    """Sum over the ranks, the same bits on every rank."""
    world, _, dist = _world(group)
    if world == 1:
        return x
    buf = x.contiguous()
    if _host_staged(dist, group):
        buf = buf.cpu()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf.to(x.device)


def sample_range_by_rank_sharded(a_local, k: int, p: int, seed: int, group=None, ops=DeviceOps) -> torch.Tensor:
    """Rows of the range basis `sample_range_by_rank` (src/random_sampling.rs:103-118) returns for the stacked matrix."""
    a = ops.prepare(a_local)
    m_r, n = a.shape
    l = k + p
    world, rank, _ = _world(group)
    _agree_on_shapes(m_r, n, l, group)
    omega = ops.random_gaussian((n, l), seed, a)         # the same stream on every rank
    y = ops.matmat(a, omega)                             # m_r x l
    q_r, r_r, ind_r = ops.pivoted_qr(y)                  # Y_r[:, ind_r] = Q_r R_r
    inv = torch.empty_like(ind_r)
    inv[ind_r] = torch.arange(l, device=ind_r.device, dtype=ind_r.dtype)
    s_r = r_r[:, inv].contiguous()                       # Y_r = Q_r S_r
    s_all = all_gather_rows(s_r, group)                  # (W l) x l
    q_s, _, _ = ops.pivoted_qr(s_all)                    # (W l) x l, identical on every rank
    block = q_s[rank * l:(rank + 1) * l, :min(k, l)].contiguous()
    return ops.dot(q_r, block)                           # m_r x k


def _agree_on_shapes(m_r: int, n: int, l: int, group=None) -> None:
    """The precondition of the sharded calls (every rank holds >= k + p rows, all ranks the same column count) is checked
    COLLECTIVELY before the first data collective: a rank that fails alone would raise while its peers block in the all-gather
    (for ever on RCCL).  One all-reduce of three integers; every rank raises the same AssertionError or none does.
    A failing host callback / compute step on one rank later on still strands the others: the group has to be torn down then."""
    world, _, dist = _world(group)
    ok = m_r >= l
    if world > 1:
        t = torch.tensor([1 if ok else 0, n, -n], dtype=torch.int64)
        if not _host_staged(dist, group):
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        all_ok, n_min, n_max = int(t[0]), int(t[1]), -int(t[2])
        assert n_min == n_max, f"the row blocks disagree on the column count ({n_min} .. {n_max})"
        assert all_ok == 1, f"every rank needs at least k + p = {l} rows" + ("" if ok else f", this one has {m_r}")
    else:
        assert ok, f"every rank needs at least k + p = {l} rows, this one has {m_r}"


_GATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)
_REDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32)


class Communicator:
    """`rc_comm` of a torch.distributed group (see the module text): RCCL transport for "nccl" groups, host callbacks otherwise."""

    def __init__(self, group=None, device=None):
        from . import _lib

        world, rank, dist = _world(group)
        assert dist is not None and world > 1, "a communicator needs an initialised process group with more than one rank"
        self.world, self.rank, self.group = world, rank, group
        self.device = torch.cuda.current_device() if device is None else int(device)
        self._h = ctypes.c_void_p()
        lib = _lib.lib()
        if dist.get_backend(group) == "nccl":
            ident = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                buf = (ctypes.c_char * 128)()
                if lib.rc_comm_unique_id(buf) != _lib.RC_OK:
                    raise _lib.HipRuntimeError("rc_comm_unique_id failed (librccl not available?)")
                ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            ident = ident.cuda(self.device)
            dist.broadcast(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            raw = bytes(ident.cpu().numpy().tobytes())
            st = lib.rc_comm_init(ctypes.byref(self._h), ctypes.c_int32(world), ctypes.c_int32(rank), raw, ctypes.c_int32(self.device))
        else:
            def gather(_user, send, recv, nbytes):
                try:
                    src = torch.frombuffer((ctypes.c_char * nbytes).from_address(send), dtype=torch.uint8)
                    dst = torch.frombuffer((ctypes.c_char * (nbytes * world)).from_address(recv), dtype=torch.uint8)
                    dist.all_gather_into_tensor(dst, src, group=group)
                    return 0
                except Exception:  # a callback must not unwind into C
                    traceback.print_exc()
                    return 1

            def reduce(_user, buf, count, elem_size):
                try:
                    t = torch.frombuffer((ctypes.c_char * (count * elem_size)).from_address(buf), dtype=torch.float64 if elem_size == 8 else torch.float32)
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                    return 0
                except Exception:
                    traceback.print_exc()
                    return 1

            self._cb = (_GATHER_FN(gather), _REDUCE_FN(reduce))  # kept alive as long as the communicator
            lib.rc_comm_init_host.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _GATHER_FN, _REDUCE_FN, ctypes.c_void_p]
            st = lib.rc_comm_init_host(ctypes.byref(self._h), world, rank, self.device, self._cb[0], self._cb[1], None)
        if st != _lib.RC_OK:
            raise _lib.HipRuntimeError(f"creating the communicator failed with status {st}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            from . import _lib

            _lib.lib().rc_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_comms = {}


def communicator(group=None) -> Optional[Communicator]:
    """The cached communicator of `group` (None for a single rank: the C call then takes a null communicator)."""
    world, _, _ = _world(group)
    if world == 1:
        return None
    key = (id(group), torch.cuda.current_device())
    if key not in _comms:
        _comms[key] = Communicator(group)
    return _comms[key]


def _rsvd_id_row_sharded_native(a_local, k: int, p: int, seed: int, group, with_id: bool) -> ShardedRsvdId:
    from . import _lib
    from .types import as_device

    a = as_device(a_local)
    m_r, n = a.shape
    _agree_on_shapes(m_r, n, k + p, group)
    comm = communicator(group)
    mk = lambda r, c: torch.empty((r, c), dtype=a.dtype, device=a.device)  # noqa: E731
    rq, u, vt = mk(m_r, k), mk(m_r, k), mk(k, n)
    s = torch.empty(k, dtype=_lib.real_dtype(a.dtype), device=a.device)
    qq, r, ind = mk(m_r, k), mk(k, n), torch.empty(n, dtype=torch.int64, device=a.device)
    c, z = (mk(m_r, k), mk(k, n)) if with_id else (None, None)
    out = _lib.rc_rsvd_id_out(_lib.mat(rq), _lib.mat(u), ctypes.c_void_p(s.data_ptr()), _lib.mat(vt), _lib.mat(qq), _lib.mat(r),
                              ctypes.c_void_p(ind.data_ptr()), _lib.mat(c), _lib.mat(z))
    ctx = _lib.default_context()
    fn = getattr(_lib.lib(), f"rc_rsvd_id_row_sharded_{_lib.suffix(a.dtype)}")
    ctx.check(fn(comm._h if comm is not None else ctypes.c_void_p(None), ctx._h, _lib.mat(a), ctypes.c_int64(k), ctypes.c_int64(p),
                 ctypes.c_uint64(seed), ctypes.byref(out)))
    return ShardedRsvdId(rq, u, s, vt, qq, r, ind, c, z)


def rsvd_id_row_sharded(a_local, k: int, p: int, seed: int, group=None, with_id: bool = True, ops=None) -> ShardedRsvdId:
    """Randomized SVD + pivoted QR + column ID of the row-sharded matrix (the cfg3 pipeline, one matrix over several GPUs).
    ops=None: the native call `rc_rsvd_id_row_sharded_*`; ops=DeviceOps (or an injected set of steps): the composition below."""
    if ops is None:
        return _rsvd_id_row_sharded_native(a_local, k, p, seed, group, with_id)
    a = ops.prepare(a_local)
    rq = sample_range_by_rank_sharded(a, k, p, seed, group, ops)
    kk = rq.shape[1]
    b_t = ops.conj_matmat(a, rq)                         # n x k: this rank's share of B^H
    b = all_reduce_sum(b_t, group).t().contiguous()      # k x n, the same bits everywhere
    ub, s, vt = ops.compute_svd(b)
    u = ops.dot(rq, ub)
    qb, r, ind = ops.pivoted_qr(b)
    qr_q = ops.dot(rq, qb)
    c = z = None
    if with_id:
        c, z = ops.column_id(qr_q, r, ind)
    return ShardedRsvdId(rq, u, s[:kk], vt, qr_q, r, ind, c, z)
