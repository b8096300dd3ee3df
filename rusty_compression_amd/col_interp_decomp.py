"""Mirror of the reference's `col_interp_decomp` module (src/col_interp_decomp.rs): A ~ C Z."""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import _lib
from .types import dot, empty


@dataclass
class ColumnID:
    """`struct ColumnID` (src/col_interp_decomp.rs:23-31): col_ind[i] = j <=> C[:, i] = A[:, j]."""

    c: torch.Tensor
    z: torch.Tensor
    col_ind: torch.Tensor

    @staticmethod
    def new(c, z, col_ind) -> "ColumnID":
        """src/col_interp_decomp.rs:113-115"""
        return ColumnID(c, z, col_ind)

    # -- ColumnIDTraits (src/col_interp_decomp.rs:44-86) ----------------------
    def nrows(self) -> int:
        return self.c.shape[0]

    def ncols(self) -> int:
        return self.z.shape[1]

    def rank(self) -> int:
        return self.c.shape[1]

    def get_c(self):
        return self.c

    def get_z(self):
        return self.z

    def get_col_ind(self):
        return self.col_ind

    def to_mat(self) -> torch.Tensor:
        """src/col_interp_decomp.rs:63-65"""
        return dot(self.c, self.z)

    def two_sided_id(self):
        """src/col_interp_decomp.rs:116-125"""
        from .two_sided_interp_decomp import TwoSidedID

        m, k = self.c.shape
        kk = min(m, k)
        c_out, x = empty(m, kk, self.c), empty(kk, k, self.c)
        row_ind = torch.empty(m, dtype=torch.int64, device=self.c.device)
        _lib.default_context().call(f"rc_column_id_two_sided_{_lib.suffix(self.c.dtype)}", _lib.mat(self.c), _lib.mat(c_out), _lib.mat(x), _lib.i64p(row_ind))
        return TwoSidedID(c_out, x, self.z.clone(), row_ind, self.col_ind.clone())

    def dot(self, rhs) -> torch.Tensor:
        """`Apply` (src/col_interp_decomp.rs:134-154): C (Z rhs)."""
        return dot(self.c, dot(self.z, rhs))
