"""Mirror of the reference's `random_sampling` module (src/random_sampling.rs).

`rng_or_omega` is either an `Rng` (Philox seed/offset, advanced like the
reference's `&mut R: Rng`) or an explicit Gaussian matrix Omega, which is how
parity tests feed the same samples to the oracle and to the HIP path."""
from __future__ import annotations

import ctypes
from typing import List, Tuple

import torch

from . import _lib
from .operator import OperatorTable, is_operator
from .random_matrix import Rng, random_gaussian
from .types import as_device, empty


def _ctx():
    return _lib.default_context()


def _omega(rng_or_omega, shape, dtype):
    if isinstance(rng_or_omega, Rng):
        return random_gaussian(shape, rng_or_omega, dtype)
    om = as_device(rng_or_omega, dtype)
    assert tuple(om.shape) == tuple(shape), f"Omega must be {shape}, got {tuple(om.shape)}"
    return om


def _empty(rows, cols, dtype):
    return torch.empty((rows, cols), dtype=dtype, device="cuda")


def sample_range_by_rank(op, k: int, p: int, rng_or_omega) -> torch.Tensor:
    """`SampleRange::sample_range_by_rank` (src/random_sampling.rs:103-118); `op`: a dense matrix or an operator (operator.py)."""
    if is_operator(op):
        tab = OperatorTable(op)
        m, n = op.nrows(), op.ncols()
        omega = _omega(rng_or_omega, (n, k + p), tab.dtype)
        q = _empty(m, min(k, m, k + p), tab.dtype)
        tab.call(_ctx(), f"rc_sample_range_by_rank_op_{_lib.suffix(tab.dtype)}", tab.byref(), ctypes.c_int64(k), ctypes.c_int64(p), _lib.mat(omega),
                 ctypes.c_uint64(0), _lib.mat(q))
        return q
    a = as_device(op)
    m, n = a.shape
    omega = _omega(rng_or_omega, (n, k + p), a.dtype)
    kk = min(k, m, k + p)
    q = empty(m, kk, a)
    _ctx().call(f"rc_sample_range_by_rank_{_lib.suffix(a.dtype)}", _lib.mat(a), ctypes.c_int64(k), ctypes.c_int64(p), _lib.mat(omega),
                ctypes.c_uint64(0), _lib.mat(q))
    return q


def sample_range_power_iteration(op, k: int, p: int, it_count: int, rng_or_omega) -> torch.Tensor:
    """`SampleRangePowerIteration::sample_range_power_iteration` (src/random_sampling.rs:131-160),
    shadowing quirk included (exactly one power step for any it_count >= 1)."""
    tab = OperatorTable(op) if is_operator(op) else None
    a = None if tab else as_device(op)
    m, n = (op.nrows(), op.ncols()) if tab else a.shape
    dtype = tab.dtype if tab else a.dtype
    l = k + p
    omega = _omega(rng_or_omega, (n, l), dtype)
    if it_count <= 0:
        kk = min(k, m, l)
    else:
        c0 = min(m, l)
        c1 = min(n, c0)
        kk = min(k, m, c1)
    q = _empty(m, kk, dtype)
    if tab:
        tab.call(_ctx(), f"rc_sample_range_power_iteration_op_{_lib.suffix(dtype)}", tab.byref(), ctypes.c_int64(k), ctypes.c_int64(p),
                 ctypes.c_int64(it_count), _lib.mat(omega), ctypes.c_uint64(0), _lib.mat(q))
        return q
    _ctx().call(f"rc_sample_range_power_iteration_{_lib.suffix(a.dtype)}", _lib.mat(a), ctypes.c_int64(k), ctypes.c_int64(p),
                ctypes.c_int64(it_count), _lib.mat(omega), ctypes.c_uint64(0), _lib.mat(q))
    return q


def max_col_norm(mat) -> float:
    """`MaxColNorm::max_col_norm` (src/random_sampling.rs:184-191)."""
    y = as_device(mat)
    out = _lib.real_out(y.dtype)
    _ctx().call(f"rc_max_col_norm_{_lib.suffix(y.dtype)}", _lib.mat(y), ctypes.byref(out))
    return float(out.value)


def sample_range_adaptive(op, rel_tol: float, sample_size: int, rng_or_omegas, max_rank: int = None) -> Tuple[torch.Tensor, List[Tuple[int, float]]]:
    """`AdaptiveSampling::sample_range_adaptive` (src/random_sampling.rs:223-274).

    Returns (q, residuals) with residuals = [(rank, estimated relative residual), ...].
    `rng_or_omegas`: an `Rng`, or an explicit n x (sample_size * blocks) matrix whose
    column blocks are the successive Omegas.  Raises CompressionError when
    `max_rank` columns (default min(m, n) rounded up to a block) do not reach the tolerance."""
    tab = OperatorTable(op) if is_operator(op) else None
    a = None if tab else as_device(op)
    m, n = (op.nrows(), op.ncols()) if tab else a.shape
    dtype = tab.dtype if tab else a.dtype
    s = int(sample_size)
    if max_rank is None:
        max_rank = ((min(m, n) + s - 1) // s) * s
    cap = int(max_rank)
    if isinstance(rng_or_omegas, Rng):
        omegas = None
        seed = rng_or_omegas.seed
        if rng_or_omegas.offset != 0:
            # the device generator addresses blocks from offset 0: draw them up front instead
            nblocks = cap // max(min(m, s), 1) + 1
            omegas = random_gaussian((n, s * nblocks), rng_or_omegas, dtype)
    else:
        omegas = as_device(rng_or_omegas, dtype)
        seed = 0
    qcap = torch.empty((cap, m), dtype=dtype, device="cuda").t()  # column-major m x cap
    hist_cap = cap // max(min(m, s), 1) + 2
    hist_rank = (ctypes.c_int64 * hist_cap)()
    hist_res = (ctypes.c_double * hist_cap)()
    rank = ctypes.c_int64(0)
    hist_len = ctypes.c_int64(0)
    tail = (ctypes.c_double(rel_tol), ctypes.c_int64(s), _lib.mat(omegas), ctypes.c_uint64(seed), _lib.mat(qcap), ctypes.byref(rank), hist_rank, hist_res,
            ctypes.c_int64(hist_cap), ctypes.byref(hist_len))
    if tab:
        tab.call(_ctx(), f"rc_sample_range_adaptive_op_{_lib.suffix(dtype)}", tab.byref(), *tail)
    else:
        _ctx().call(f"rc_sample_range_adaptive_{_lib.suffix(dtype)}", _lib.mat(a), *tail)
    if isinstance(rng_or_omegas, Rng) and omegas is None:
        rng_or_omegas.offset += (hist_len.value + 1) * n * s
    q = qcap[:, : rank.value].contiguous()
    residuals = [(int(hist_rank[i]), float(hist_res[i])) for i in range(hist_len.value)]
    return q, residuals
