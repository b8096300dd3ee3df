"""Mirror of the reference's `random_matrix` module (src/random_matrix.rs).

`Rng` stands in for the caller-owned `&mut R: Rng` of the reference: a Philox
(seed, offset) pair; every draw advances the offset by the number of samples, so
consumption order is observable exactly as in the reference."""
from __future__ import annotations

import ctypes

import torch

from . import _lib


class Rng:
    def __init__(self, seed: int = 0, offset: int = 0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.offset = int(offset)


def _dtype(dtype):
    return {"f64": torch.float64, "f32": torch.float32, "c64": torch.complex128, "c32": torch.complex64}.get(dtype, dtype)


def random_gaussian(dimension, rng: Rng, dtype=torch.float64) -> torch.Tensor:
    """`RandomMatrix::random_gaussian` (src/random_matrix.rs:21, :120-125): row-major draw order."""
    dtype = _dtype(dtype)
    rows, cols = dimension
    out = torch.empty((rows, cols), dtype=dtype, device="cuda")
    _lib.default_context().call(f"rc_random_gaussian_{_lib.suffix(dtype)}", _lib.mat(out), ctypes.c_uint64(rng.seed), ctypes.c_uint64(rng.offset))
    rng.offset += rows * cols
    return out


def random_bits_u32(n: int, seed: int, word_offset: int = 0) -> torch.Tensor:
    """Raw uint32 words of the Philox4x32-10 stream behind `random_gaussian` (rc_random_bits_u32), as an int64 tensor."""
    out = torch.empty(int(n), dtype=torch.int32, device="cuda")
    _lib.default_context().call("rc_random_bits_u32", ctypes.c_void_p(out.data_ptr()), ctypes.c_int64(int(n)),
                                ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), ctypes.c_uint64(int(word_offset)))
    return out.to(torch.int64) & 0xFFFFFFFF


def random_orthogonal_matrix(dimension, rng: Rng, dtype=torch.float64) -> torch.Tensor:
    """src/random_matrix.rs:35-56: U of the thin SVD of a Gaussian (rows orthonormal if wide)."""
    from .svd import SVD

    m, n = dimension
    swap = n > m
    if swap:
        m, n = n, m
    g = random_gaussian((m, n), rng, dtype)
    u = SVD.compute_from(g).u
    return u.t().conj().contiguous() if swap else u  # conjugate transpose (src/random_matrix.rs:51-53)


def random_approximate_low_rank_matrix(dimension, sigma_max: float, sigma_min: float, rng: Rng, dtype=torch.float64) -> torch.Tensor:
    """src/random_matrix.rs:70-93: U diag(geomspace(sigma_min, sigma_max)) Vt."""
    assert sigma_min < sigma_max, "`sigma_min` must be smaller than `sigma_max`"
    assert sigma_min > 0.0, "`sigma_min` must be positive."
    dtype = _dtype(dtype)
    m, n = dimension
    r = min(m, n)
    u = random_orthogonal_matrix((m, r), rng, dtype)
    vt = random_orthogonal_matrix((r, n), rng, dtype)
    s = torch.logspace(torch.log10(torch.tensor(float(sigma_min))).item(), torch.log10(torch.tensor(float(sigma_max))).item(), r,
                       dtype=torch.float64, device="cuda").to(_lib.real_dtype(dtype))
    from .svd import SVD

    return SVD(u, s, vt).to_mat()
