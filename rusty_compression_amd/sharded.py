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

Outputs: row-sharded `range_q`, `u`, `qr_q`, `c`; replicated `s`, `vt`, `r`, `ind`, `z`.  The collectives are
torch.distributed's (backend "nccl" = RCCL over xGMI on GPUs); under a gloo group (the tests: two ranks on one GPU) the two small
buffers are staged through the host.  Compute goes through the C ABI like everything else: no CPU fallback."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

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
    assert m_r >= l, f"every rank needs at least k + p = {l} rows, this one has {m_r}"
    world, rank, _ = _world(group)
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


def rsvd_id_row_sharded(a_local, k: int, p: int, seed: int, group=None, with_id: bool = True, ops=DeviceOps) -> ShardedRsvdId:
    """Randomized SVD + pivoted QR + column ID of the row-sharded matrix (the cfg3 pipeline, one matrix over several GPUs)."""
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
