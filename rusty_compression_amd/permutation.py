"""Mirror of the reference's `permutation` module (src/permutation.rs)."""
from __future__ import annotations

import ctypes
from enum import IntEnum

import torch

from . import _lib
from .types import as_device, as_index


class MatrixPermutationMode(IntEnum):
    """src/permutation.rs:7-16"""

    COL = 0
    ROW = 1
    COLINV = 2
    ROWINV = 3


class VectorPermutationMode(IntEnum):
    """src/permutation.rs:19-24"""

    INV = 0
    NOINV = 1


def invert_permutation_vector(perm) -> torch.Tensor:
    """src/permutation.rs:28-38"""
    p = as_index(perm)
    inv = torch.empty_like(p)
    _lib.default_context().call("rc_invert_permutation", _lib.i64p(p), ctypes.c_int64(p.numel()), _lib.i64p(inv))
    return inv


def apply_permutation(arr, index_array, mode) -> torch.Tensor:
    """`ApplyPermutationToMatrix/Vector::apply_permutation` (src/permutation.rs:84-184).

    A 2-D `arr` takes a MatrixPermutationMode, a 1-D `arr` a VectorPermutationMode.
    Length mismatches raise AssertionError (the reference asserts)."""
    a = as_device(arr)
    idx = as_index(index_array)
    out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    suf = _lib.suffix(a.dtype)
    if a.dim() == 2:
        _lib.default_context().call(f"rc_apply_permutation_matrix_{suf}", ctypes.c_int32(int(MatrixPermutationMode(mode))),
                                    _lib.mat(a), _lib.i64p(idx), ctypes.c_int64(idx.numel()), _lib.mat(out))
    else:
        _lib.default_context().call(f"rc_apply_permutation_vector_{suf}", ctypes.c_int32(int(VectorPermutationMode(mode))),
                                    _lib.mat(a), _lib.i64p(idx), ctypes.c_int64(idx.numel()), _lib.mat(out))
    return out
