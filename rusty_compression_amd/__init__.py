"""rusty_compression_amd -- MI355X-native randomized low-rank compression.

Host-side mirror of the public surface of the Rust crate `rusty-compression`
v0.1.1 (reference src/lib.rs:90-102) over the C ABI of
librusty_compression_amd.so (include/rusty_compression_amd.h).  Only the hot
path is here: Gaussian sketch -> column-pivoted QR (+ permutation) -> small SVD /
column, row and two-sided interpolative decompositions.
"""
from ._lib import (CompressionError, Context, HipRuntimeError, LayoutError, LinalgError, PivotedQRError,  # noqa: F401
                   RustyCompressionError, default_context)
from .col_interp_decomp import ColumnID  # noqa: F401
from .operator import DenseOperator, LowRankOperator, Operator  # noqa: F401
from .permutation import (MatrixPermutationMode, VectorPermutationMode, apply_permutation,  # noqa: F401
                          invert_permutation_vector)
from .qr import LQ, QR, pivoted_lq, pivoted_qr  # noqa: F401
from .random_matrix import (Rng, random_approximate_low_rank_matrix, random_bits_u32, random_gaussian,  # noqa: F401
                            random_orthogonal_matrix)
from .random_sampling import (max_col_norm, sample_range_adaptive, sample_range_by_rank,  # noqa: F401
                              sample_range_power_iteration)
from .row_interp_decomp import RowID  # noqa: F401
from .svd import SVD, compute_svd  # noqa: F401
from .lapack import geqp3, orgqr, trsm_upper  # noqa: F401
from .two_sided_interp_decomp import TwoSidedID  # noqa: F401
from .types import CompressionType, conj_matmat, dot, matmat, rel_diff_fro, rel_diff_l2  # noqa: F401

__all__ = [
    "QR", "LQ", "SVD", "ColumnID", "RowID", "TwoSidedID", "CompressionType", "Rng",
    "MatrixPermutationMode", "VectorPermutationMode", "apply_permutation", "invert_permutation_vector",
    "random_gaussian", "random_bits_u32", "random_orthogonal_matrix", "random_approximate_low_rank_matrix",
    "sample_range_by_rank", "sample_range_power_iteration", "sample_range_adaptive", "max_col_norm",
    "matmat", "conj_matmat", "dot", "rel_diff_fro", "rel_diff_l2", "pivoted_qr", "pivoted_lq", "compute_svd", "geqp3", "orgqr", "trsm_upper",
    "RustyCompressionError", "LinalgError", "CompressionError", "LayoutError", "PivotedQRError", "HipRuntimeError",
    "Context", "default_context", "Operator", "DenseOperator", "LowRankOperator",
]
