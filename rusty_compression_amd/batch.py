"""Batches of independent matrices sharded over the GPUs of one node (BASELINE.json configs[4]).

The path shards by MATRIX: matrix i of N goes to rank i // (N / world) (8 per GPU for the
64-matrix config); nothing is exchanged while compressing.  The one exchange step is the
gather of the finished factor blocks (C: m x k, Z: k x n, col_ind: n) to rank 0 --
`torch.distributed.gather` of one packed, equal-sized buffer per rank: RCCL over xGMI on
the GPUs ("nccl" backend), gloo in the CPU tests.  Every peer has its own direct xGMI link
to the root, so at ~16 MiB per rank the gather is latency-, not bandwidth-bound.

reference call sequence per matrix (examples/interpolative_decomposition.rs:25-32):
    QR::compute_from(a) -> compress(RANK(k)) -> column_id()
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch


def shard_range(n_items: int, world: int, rank: int) -> range:
    """Contiguous block partition: the first (n_items % world) ranks take one extra item."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def column_id_rank(a: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Rank-k column ID of one matrix through the C ABI (rc_column_id_rank_*)."""
    import ctypes

    from . import _lib
    from .types import as_device, empty

    a = as_device(a)
    m, n = a.shape
    k = min(int(k), m, n)
    c, z = empty(m, k, a), empty(k, n, a)
    ind = torch.empty(n, dtype=torch.int64, device=a.device)
    _lib.default_context().call(f"rc_column_id_rank_{_lib.suffix(a.dtype)}", _lib.mat(a), ctypes.c_int64(k), _lib.mat(c), _lib.mat(z), _lib.i64p(ind))
    return c, z, ind


def packed_bytes(m: int, n: int, k: int, elem_size: int) -> int:
    """Bytes one matrix's factors take in the packed buffer: C (m x k) | Z (k x n) | col_ind (n int64)."""
    return (m * k + k * n) * elem_size + n * 8


def pack_factors(factors: Sequence[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]) -> torch.Tensor:
    """[(C, Z, col_ind), ...] -> one flat buffer of the factors' dtype (indices bit-cast, exact)."""
    parts = []
    for c, z, ind in factors:
        idx = ind.to(torch.int64).contiguous()
        parts += [c.contiguous().reshape(-1), z.contiguous().reshape(-1), idx.view(c.dtype).reshape(-1)]
    return torch.cat(parts) if parts else torch.empty(0)


def unpack_factors(buf: torch.Tensor, count: int, m: int, n: int, k: int):
    out = []
    per_idx = n * 8 // buf.element_size()
    off = 0
    for _ in range(count):
        c = buf[off:off + m * k].reshape(m, k); off += m * k
        z = buf[off:off + k * n].reshape(k, n); off += k * n
        ind = buf[off:off + per_idx].clone().view(torch.int64); off += per_idx
        out.append((c, z, ind))
    return out


def batch_column_id(matrices: Sequence[torch.Tensor], k: int, compute: Optional[Callable] = None,
                    group=None, dst: int = 0) -> Optional[List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]]:
    """Compress the local shard `matrices` (this rank's block of the global batch) and gather every
    rank's factors on `dst` in global matrix order.  Returns the full list on `dst`, None elsewhere.
    All ranks must hold the same number of same-shaped matrices (8 per GPU in configs[4])."""
    import torch.distributed as dist

    compute = compute or column_id_rank
    local = [compute(a, k) for a in matrices]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    packed = pack_factors(local)
    gathered = [torch.empty_like(packed) for _ in range(world)] if rank == dst else None
    dist.gather(packed, gathered, dst=dst, group=group)   # the ONLY collective on this path
    if rank != dst:
        return None
    m, n = matrices[0].shape
    kk = local[0][0].shape[1]
    out = []
    for r in range(world):
        out += unpack_factors(gathered[r], len(matrices), m, n, kk)
    return out
