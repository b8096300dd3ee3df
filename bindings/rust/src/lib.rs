//! The reference crate's public surface (rusty-compression v0.1.1, `src/lib.rs:67-102`) over the C ABI of
//! librusty_compression_amd.so (MI355X).  Module names, trait names, method names, argument meaning and error
//! behaviour are the reference's, so that a user of the reference can switch by changing the crate name:
//!
//! ```ignore
//! use rusty_compression_amd::*;            // was: use rusty_compression::*;
//! let qr = QR::<f64>::compute_from(mat.view())?.compress(CompressionType::RANK(20))?;
//! let two_sided = qr.column_id()?.two_sided_id()?;
//! ```
//!
//! No arithmetic happens on the host: every numerical method uploads its `ndarray` views, issues ONE call of the C ABI
//! and downloads the owned results (the reference returns owned arrays too).  Data that should stay on the GPU between
//! calls uses `device::DeviceMatrix`, which implements the operator traits (`MatVec` ... `ConjMatMat`) as well.
//!
//! What a maintainer adopting this must know:
//! * `MatMat` / `ConjMatMat` of dense matrices are one GEMM each, not the reference's per-column `matvec` loop
//!   (`src/types.rs:60-70`, `:145-146`): results differ by summation order only.
//! * `RandomMatrix::random_gaussian` draws on the HOST from the caller's `rand::Rng` exactly as the reference does
//!   (`src/random_matrix.rs:120-125`), so a seeded run reproduces the reference's Omega bit for bit; the samplers upload
//!   it.  `random_matrix::random_gaussian_device` is the on-device Philox stream for callers that do not need that.
//! * The range finders and `compute_from_range_estimate` are generic over the operator as in the reference
//!   (`src/random_sampling.rs:102`, `:130`, `:222`): for `f32` / `f64` the library runs the algorithm and calls back for the
//!   two products only (`operator.rs`, `rc_*_op_*`), so a custom `MatVec` operator needs no dense matrix anywhere.
//! * `CompressionType`, `RustyCompressionError`, `Result` are the reference's (`src/lib.rs:82-87`, `src/types.rs:11-23`).
pub mod col_interp_decomp;
pub mod compute_svd;
pub mod device;
pub mod ffi;
pub mod operator;
pub mod permutation;
pub(crate) mod pivoted_qr;
pub mod qr;
pub mod random_matrix;
pub mod random_sampling;
pub mod row_interp_decomp;
pub mod svd;
pub mod two_sided_interp_decomp;
pub mod types;

/// `CompressionType` (reference `src/lib.rs:82-87`).
pub enum CompressionType {
    /// Adaptive compression with a specified tolerance
    ADAPTIVE(f64),
    /// Rank based compression with specified rank
    RANK(usize),
}

pub use col_interp_decomp::{ColumnID, ColumnIDTraits};
pub use permutation::*;
pub use qr::{LQTraits, QRTraits, LQ, QR};
pub use random_matrix::RandomMatrix;
pub use random_sampling::*;
pub use row_interp_decomp::{RowID, RowIDTraits};
pub use svd::{SVDTraits, SVD};
pub use two_sided_interp_decomp::{TwoSidedID, TwoSidedIDTraits};
pub use operator::{DeviceOperator, HostConjMatMat, HostMatMat};
pub use types::RelDiff;

pub use types::{c32, c64, Scalar};

pub use types::Result;
