"""Batches of independent matrices sharded over the GPUs of one node (BASELINE.json configs[4]).

The path shards by MATRIX: matrix i of N goes to rank i // (N / world) (8 per GPU for the
64-matrix config); nothing is exchanged while compressing.  The one exchange step is the
gather of the finished factor blocks (C: m x k, Z: k x n, col_ind: n) to rank 0, one packed,
equal-sized buffer per rank:

  * `Comm` + `gather_packed`: the library's own RCCL path (rc_comm_init / rc_comm_gather, grouped
    ncclSend / ncclRecv over xGMI) -- what a Rust / C++ host uses;
  * `torch.distributed.gather` when a torch process group is active (gloo in the CPU tests).

Per GPU the matrices run through rc_batch_column_id_* (include/rusty_compression_amd.h): spread over
`lanes` contexts / HIP streams and advanced in lock step, one host wait per pivoting panel for all of them.

reference call sequence per matrix (examples/interpolative_decomposition.rs:25-32):
    QR::compute_from(a) -> compress(RANK(k)) -> column_id()
"""
from __future__ import annotations

import ctypes
from typing import Callable, List, Optional, Sequence, Tuple

import torch


def shard_range(n_items: int, world: int, rank: int) -> range:
    """Contiguous block partition: the first (n_items % world) ranks take one extra item (rc_batch_shard_range)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def column_id_rank(a: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Rank-k column ID of one matrix through the C ABI (rc_column_id_rank_*)."""
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
    """Bytes one matrix's factors take in the packed buffer: C (m x k) | Z (k x n) | pad to 8 | col_ind (n int64)
    (rc_batch_packed_bytes)."""
    return ((m * k + k * n) * elem_size + 7) // 8 * 8 + n * 8


def pack_factors(factors: Sequence[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]) -> torch.Tensor:
    """[(C, Z, col_ind), ...] -> one flat uint8 buffer in the layout of rc_batch_column_id_* (indices bit-cast, exact)."""
    parts = []
    for c, z, ind in factors:
        body = torch.cat([c.contiguous().reshape(-1), z.contiguous().reshape(-1)]).view(torch.uint8)
        pad = (-body.numel()) % 8
        if pad:
            body = torch.cat([body, torch.zeros(pad, dtype=torch.uint8, device=body.device)])
        parts += [body, ind.to(torch.int64).contiguous().view(torch.uint8)]
    return torch.cat(parts) if parts else torch.empty(0, dtype=torch.uint8)


def unpack_factors(buf: torch.Tensor, count: int, m: int, n: int, k: int, dtype: Optional[torch.dtype] = None):
    """Inverse of pack_factors / view of the buffer rc_batch_column_id_* fills: [(C, Z, col_ind), ...]."""
    if buf.dtype != torch.uint8:  # older callers packed in the factors' dtype
        dtype = dtype or buf.dtype
        buf = buf.contiguous().view(torch.uint8)
    if dtype is None:  # infer the scalar type from the buffer size
        dtype = next((d for d in (torch.float32, torch.float64) if packed_bytes(m, n, k, torch.empty(0, dtype=d).element_size()) * count == buf.numel()), None)
        assert dtype is not None, "unpack_factors: buffer size matches neither f32 nor f64 factors"
    es = torch.empty(0, dtype=dtype).element_size()
    per = packed_bytes(m, n, k, es)
    out = []
    for i in range(count):
        b = buf[i * per:(i + 1) * per]
        c = b[: m * k * es].view(dtype).reshape(m, k)
        z = b[m * k * es: (m * k + k * n) * es].view(dtype).reshape(k, n)
        ind = b[per - n * 8:].view(torch.int64)
        out.append((c, z, ind))
    return out


class _LanePool:
    """Contexts + HIP streams for the lock-step batch (created once per device, reused)."""

    _pools = {}

    @classmethod
    def get(cls, device: int, lanes: int):
        from . import _lib

        pool = cls._pools.setdefault(device, [])
        while len(pool) < lanes:
            raw = ctypes.c_void_p()
            st = _lib.lib().rc_stream_create(ctypes.c_int32(device), ctypes.byref(raw))
            if st != 0:
                raise _lib.HipRuntimeError("rc_stream_create failed")
            pool.append(_lib.Context(device, raw.value))
        return pool[:lanes]


def batch_column_id_packed(matrices: Sequence[torch.Tensor], k: int, lanes: int = 8) -> torch.Tensor:
    """rc_batch_column_id_*: rank-k column ID of same-shaped device matrices -> packed uint8 device buffer."""
    from . import _lib
    from .types import as_device

    mats = [as_device(a) for a in matrices]
    if not mats:
        return torch.empty(0, dtype=torch.uint8, device="cuda")
    m, n = mats[0].shape
    k = min(int(k), m, n)
    dev = mats[0].device.index if mats[0].device.index is not None else torch.cuda.current_device()
    es = mats[0].element_size()
    per = packed_bytes(m, n, k, es)
    out = torch.empty(len(mats) * per, dtype=torch.uint8, device=mats[0].device)
    ctxs = _LanePool.get(dev, max(1, min(lanes, len(mats))))
    arr = (ctypes.c_void_p * len(ctxs))(*[c._h.value for c in ctxs])
    marr = (_lib.rc_matrix * len(mats))(*[_lib.mat(a) for a in mats])
    torch.cuda.current_stream(dev).synchronize()  # the inputs were produced on torch's stream, the lanes have their own
    fn = getattr(_lib.lib(), f"rc_batch_column_id_{_lib.suffix(mats[0].dtype)}")
    ctxs[0].check(fn(arr, ctypes.c_int32(len(ctxs)), marr, ctypes.c_int32(len(mats)), ctypes.c_int64(k), ctypes.c_void_p(out.data_ptr())))
    return out


class Comm:
    """rc_comm_*: the library's RCCL communicator for the factor gather (one process per GPU)."""

    def __init__(self, world: int, rank: int, unique_id: bytes, device: Optional[int] = None):
        from . import _lib

        self.world, self.rank = int(world), int(rank)
        self.device = torch.cuda.current_device() if device is None else int(device)
        self._h = ctypes.c_void_p()
        st = _lib.lib().rc_comm_init(ctypes.byref(self._h), ctypes.c_int32(world), ctypes.c_int32(rank), ctypes.c_char_p(unique_id), ctypes.c_int32(self.device))
        if st != 0:
            raise _lib.HipRuntimeError(f"rc_comm_init failed with status {st}")

    @classmethod
    def from_process_group(cls, group=None, device: Optional[int] = None) -> "Comm":
        """The library's RCCL communicator for the ranks of a torch.distributed group: rank 0 draws the unique id
        (rc_comm_unique_id) and the group carries its 128 bytes to the others once; every later byte moves through rc_comm_gather."""
        import torch.distributed as dist

        world, rank = dist.get_world_size(group), dist.get_rank(group)
        dev = torch.cuda.current_device() if device is None else int(device)
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            ident = torch.frombuffer(bytearray(cls.unique_id()), dtype=torch.uint8).clone()
        if dist.get_backend(group) == "nccl":
            ident = ident.cuda(dev)
        dist.broadcast(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(world, rank, bytes(ident.cpu().numpy().tobytes()), dev)

    @staticmethod
    def unique_id() -> bytes:
        from . import _lib

        buf = ctypes.create_string_buffer(128)
        st = _lib.lib().rc_comm_unique_id(buf)
        if st != 0:
            raise _lib.HipRuntimeError(f"rc_comm_unique_id failed with status {st} (librccl not available?)")
        return buf.raw

    def gather(self, send: torch.Tensor, root: int = 0) -> Optional[torch.Tensor]:
        """Gather equal-sized uint8 device buffers to `root` (returns the concatenation there, None elsewhere)."""
        from . import _lib

        send = send.contiguous()
        nbytes = send.numel() * send.element_size()
        recv = torch.empty(self.world * nbytes, dtype=torch.uint8, device=send.device) if self.rank == root else None
        ctx = _lib.default_context()
        st = _lib.lib().rc_comm_gather(self._h, ctx._h, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr() if recv is not None else None),
                                       ctypes.c_size_t(nbytes), ctypes.c_int32(root))
        ctx.check(st)
        ctx.synchronize()
        return recv

    def close(self):
        from . import _lib

        if self._h.value:
            _lib.lib().rc_comm_destroy(self._h)
            self._h = ctypes.c_void_p()


def batch_column_id(matrices: Sequence[torch.Tensor], k: int, compute: Optional[Callable] = None,
                    group=None, dst: int = 0, comm: Optional[Comm] = None, lanes: int = 8) -> Optional[List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]]:
    """Compress the local shard `matrices` (this rank's block of the global batch) and gather every
    rank's factors on `dst` in global matrix order.  Returns the full list on `dst`, None elsewhere.
    All ranks must hold the same number of same-shaped matrices (8 per GPU in configs[4]).

    compute: optional per-matrix function (a, k) -> (C, Z, col_ind) replacing the device path (the gloo tests inject the CPU
    oracle); comm: the library's RCCL communicator (otherwise torch.distributed when a process group is active)."""
    import torch.distributed as dist

    if not matrices:
        return []
    m, n = matrices[0].shape
    if compute is None:
        packed = batch_column_id_packed(matrices, k, lanes)
        kk = min(int(k), m, n)
        dtype = matrices[0].dtype
    else:
        local = [compute(a, k) for a in matrices]
        kk = local[0][0].shape[1]
        dtype = local[0][0].dtype
        packed = pack_factors(local)
    nloc = len(matrices)
    if comm is not None and comm.world > 1:
        got = comm.gather(packed, dst)
        if comm.rank != dst:
            return None
        return unpack_factors(got, comm.world * nloc, m, n, kk, dtype)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return unpack_factors(packed, nloc, m, n, kk, dtype)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    gathered = [torch.empty_like(packed) for _ in range(world)] if rank == dst else None
    dist.gather(packed, gathered, dst=dst, group=group)   # the ONLY collective on this path
    if rank != dst:
        return None
    out = []
    for r in range(world):
        out += unpack_factors(gathered[r], nloc, m, n, kk, dtype)
    return out
