"""Mirror of the reference's `two_sided_interp_decomp` module (src/two_sided_interp_decomp.rs): A ~ C X R."""
from __future__ import annotations

from dataclasses import dataclass

import torch

from .types import dot


@dataclass
class TwoSidedID:
    """`struct TwoSidedID` (src/two_sided_interp_decomp.rs:19-30); X = A[row_ind[:k], col_ind[:k]]."""

    c: torch.Tensor
    x: torch.Tensor
    r: torch.Tensor
    row_ind: torch.Tensor
    col_ind: torch.Tensor

    @staticmethod
    def new(x, r, c, col_ind, row_ind) -> "TwoSidedID":
        """Argument order of the reference constructor (src/two_sided_interp_decomp.rs:89-95)."""
        return TwoSidedID(c, x, r, row_ind, col_ind)

    # -- TwoSidedIDTraits (src/two_sided_interp_decomp.rs:43-96) --------------
    def nrows(self) -> int:
        return self.c.shape[0]

    def ncols(self) -> int:
        return self.r.shape[1]

    def rank(self) -> int:
        return self.c.shape[1]

    def get_c(self):
        return self.c

    def get_x(self):
        return self.x

    def get_r(self):
        return self.r

    def get_col_ind(self):
        return self.col_ind

    def get_row_ind(self):
        return self.row_ind

    def to_mat(self) -> torch.Tensor:
        """src/two_sided_interp_decomp.rs:62-64: C (X R)."""
        return dot(self.c, dot(self.x, self.r))

    def dot(self, rhs) -> torch.Tensor:
        """`Apply` (src/two_sided_interp_decomp.rs:154-171): C (X (R rhs))."""
        return dot(self.c, dot(self.x, dot(self.r, rhs)))
