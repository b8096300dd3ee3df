"""Mirror of the reference's `row_interp_decomp` module (src/row_interp_decomp.rs): A ~ X R."""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import _lib
from .types import dot, empty


@dataclass
class RowID:
    """`struct RowID` (src/row_interp_decomp.rs:25-33): row_ind[i] = j <=> R[i, :] = A[j, :]."""

    x: torch.Tensor
    r: torch.Tensor
    row_ind: torch.Tensor

    @staticmethod
    def new(x, r, row_ind) -> "RowID":
        """src/row_interp_decomp.rs:116-118"""
        return RowID(x, r, row_ind)

    # -- RowIDTraits (src/row_interp_decomp.rs:46-89) -------------------------
    def nrows(self) -> int:
        return self.x.shape[0]

    def ncols(self) -> int:
        return self.r.shape[1]

    def rank(self) -> int:
        return self.r.shape[0]

    def get_x(self):
        return self.x

    def get_r(self):
        return self.r

    def get_row_ind(self):
        return self.row_ind

    def to_mat(self) -> torch.Tensor:
        """src/row_interp_decomp.rs:65-67"""
        return dot(self.x, self.r)

    def two_sided_id(self):
        """src/row_interp_decomp.rs:120-130"""
        from .two_sided_interp_decomp import TwoSidedID

        k, n = self.r.shape
        kk = min(k, n)
        x, r_out = empty(k, kk, self.r), empty(kk, n, self.r)
        col_ind = torch.empty(n, dtype=torch.int64, device=self.r.device)
        _lib.default_context().call(f"rc_row_id_two_sided_{_lib.suffix(self.r.dtype)}", _lib.mat(self.r), _lib.mat(x), _lib.mat(r_out), _lib.i64p(col_ind))
        return TwoSidedID(self.x.clone(), x, r_out, self.row_ind.clone(), col_ind)

    def dot(self, rhs) -> torch.Tensor:
        """`Apply` (src/row_interp_decomp.rs:134-154): X (R rhs)."""
        return dot(self.x, dot(self.r, rhs))
