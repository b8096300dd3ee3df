//! Operators behind the C ABI's callback table (`rc_operator`, include/rusty_compression_amd.h).
//!
//! The reference implements its range finders for ANY operator (`impl<Op: MatMat<A = $scalar>> SampleRange for Op`,
//! `src/random_sampling.rs:102`, `:130`, `:222`; `compute_from_range_estimate<Op: ConjMatMat>`, `src/qr.rs:311-323`,
//! `src/svd.rs:171-183`).  For the real scalar types the library runs those algorithms itself and calls back for the two
//! products only (`rc_*_op_f32` / `_f64`): this module turns a host operator (`MatMat` / `ConjMatMat` on `ndarray` views) into
//! that table -- the right-hand side comes down from the device, the host's product runs, the result goes up into the view the
//! library handed over -- so the crate does no arithmetic of its own (no Gram-Schmidt, no subtraction on the host).
//! A device-resident operator (`DeviceMatrix`, or a user type implementing `DeviceOperator`) skips the transfers.
//! The complex types have no callback entry points in the C ABI yet: their generic samplers stay compositions of the one-call
//! factorizations (`OpScalar` for `c32` / `c64` in `random_sampling.rs`).
use crate::device::{self, Context, DeviceMatrix};
use crate::ffi::*;
use crate::types::{ConjMatMat, MatMat, Result, RustyCompressionError, Scalar};
use ndarray::Array2;
use std::os::raw::c_void;

/// The typed `rc_*_op_*` entry points (real scalars).
pub trait RealOpFfi: Scalar {
    unsafe fn ffi_sample_range_by_rank_op(ctx: *mut rc_context, op: *const rc_operator, k: i64, p: i64, omega: rc_matrix, seed: u64, q: rc_matrix) -> rc_status;
    unsafe fn ffi_sample_range_power_iteration_op(ctx: *mut rc_context, op: *const rc_operator, k: i64, p: i64, it: i64, omega: rc_matrix, seed: u64, q: rc_matrix) -> rc_status;
    #[allow(clippy::too_many_arguments)]
    unsafe fn ffi_sample_range_adaptive_op(ctx: *mut rc_context, op: *const rc_operator, rel_tol: f64, sample_size: i64, omegas: rc_matrix, seed: u64, q_cap: rc_matrix,
                                           rank: *mut i64, hist_rank: *mut i64, hist_res: *mut f64, hist_cap: i64, hist_len: *mut i64) -> rc_status;
    unsafe fn ffi_qr_from_range_estimate_op(ctx: *mut rc_context, range: rc_matrix, op: *const rc_operator, q: rc_matrix, r: rc_matrix, ind: *mut i64) -> rc_status;
    unsafe fn ffi_svd_from_range_estimate_op(ctx: *mut rc_context, range: rc_matrix, op: *const rc_operator, u: rc_matrix, s: *mut Self::Real, vt: rc_matrix) -> rc_status;
}
macro_rules! impl_real_op_ffi {
    ($t:ty, $suf:ident) => {
        paste::paste! {
        impl RealOpFfi for $t {
            unsafe fn ffi_sample_range_by_rank_op(ctx: *mut rc_context, op: *const rc_operator, k: i64, p: i64, omega: rc_matrix, seed: u64, q: rc_matrix) -> rc_status { [<rc_sample_range_by_rank_op_ $suf>](ctx, op, k, p, omega, seed, q) }
            unsafe fn ffi_sample_range_power_iteration_op(ctx: *mut rc_context, op: *const rc_operator, k: i64, p: i64, it: i64, omega: rc_matrix, seed: u64, q: rc_matrix) -> rc_status { [<rc_sample_range_power_iteration_op_ $suf>](ctx, op, k, p, it, omega, seed, q) }
            unsafe fn ffi_sample_range_adaptive_op(ctx: *mut rc_context, op: *const rc_operator, rel_tol: f64, sample_size: i64, omegas: rc_matrix, seed: u64, q_cap: rc_matrix,
                                                   rank: *mut i64, hist_rank: *mut i64, hist_res: *mut f64, hist_cap: i64, hist_len: *mut i64) -> rc_status {
                [<rc_sample_range_adaptive_op_ $suf>](ctx, op, rel_tol, sample_size, omegas, seed, q_cap, rank, hist_rank, hist_res, hist_cap, hist_len)
            }
            unsafe fn ffi_qr_from_range_estimate_op(ctx: *mut rc_context, range: rc_matrix, op: *const rc_operator, q: rc_matrix, r: rc_matrix, ind: *mut i64) -> rc_status { [<rc_qr_from_range_estimate_op_ $suf>](ctx, range, op, q, r, ind) }
            unsafe fn ffi_svd_from_range_estimate_op(ctx: *mut rc_context, range: rc_matrix, op: *const rc_operator, u: rc_matrix, s: *mut $t, vt: rc_matrix) -> rc_status { [<rc_svd_from_range_estimate_op_ $suf>](ctx, range, op, u, s, vt) }
        }
        }
    };
}
impl_real_op_ffi!(f32, f32);
impl_real_op_ffi!(f64, f64);

/// An operator whose products run on the device: `y = A x` / `y = A^H x` on the strided device views the library hands over,
/// enqueued on the context (through this crate's FFI or with the host's own kernels on `rc_get_stream`).
pub trait DeviceOperator {
    type A: Scalar;
    fn nrows(&self) -> usize;
    fn ncols(&self) -> usize;
    fn matmat_device(&self, ctx: &Context, x: rc_matrix, y: rc_matrix) -> Result<()>;
    /// `None`: the operator is only `MatMat` (enough for `SampleRange`, `src/random_sampling.rs:102`).
    fn conj_matmat_device(&self, ctx: &Context, x: rc_matrix, y: rc_matrix) -> Option<Result<()>>;
}

impl<A: Scalar> DeviceOperator for DeviceMatrix<A> {
    type A = A;
    fn nrows(&self) -> usize { self.dims().0 }
    fn ncols(&self) -> usize { self.dims().1 }
    fn matmat_device(&self, ctx: &Context, x: rc_matrix, y: rc_matrix) -> Result<()> { ctx.check(unsafe { A::ffi_matmat(ctx.raw(), self.view(), x, y) }) }
    fn conj_matmat_device(&self, ctx: &Context, x: rc_matrix, y: rc_matrix) -> Option<Result<()>> {
        Some(ctx.check(unsafe { A::ffi_conj_matmat(ctx.raw(), self.view(), x, y) }))
    }
}

/// A host operator (the reference's `MatMat`, optionally `ConjMatMat`) as a `DeviceOperator`: operands cross PCIe per product.
pub struct HostMatMat<'a, Op: MatMat>(pub &'a Op);
pub struct HostConjMatMat<'a, Op: ConjMatMat>(pub &'a Op);

fn host_product<A: Scalar>(ctx: &Context, x: rc_matrix, y: rc_matrix, f: impl FnOnce(&Array2<A>) -> Array2<A>) -> Result<()> {
    let xh = device::download_view::<A>(ctx, x)?;
    let yh = f(&xh);
    assert_eq!((yh.nrows() as i64, yh.ncols() as i64), (y.rows, y.cols), "operator product has the wrong shape");
    device::upload_into_view::<A>(ctx, yh.view(), y)
}
impl<'a, Op: MatMat> DeviceOperator for HostMatMat<'a, Op> {
    type A = Op::A;
    fn nrows(&self) -> usize { self.0.nrows() }
    fn ncols(&self) -> usize { self.0.ncols() }
    fn matmat_device(&self, ctx: &Context, x: rc_matrix, y: rc_matrix) -> Result<()> { host_product::<Op::A>(ctx, x, y, |xh| self.0.matmat(xh.view())) }
    fn conj_matmat_device(&self, _ctx: &Context, _x: rc_matrix, _y: rc_matrix) -> Option<Result<()>> { None }
}
impl<'a, Op: ConjMatMat> DeviceOperator for HostConjMatMat<'a, Op> {
    type A = Op::A;
    fn nrows(&self) -> usize { self.0.nrows() }
    fn ncols(&self) -> usize { self.0.ncols() }
    fn matmat_device(&self, ctx: &Context, x: rc_matrix, y: rc_matrix) -> Result<()> { host_product::<Op::A>(ctx, x, y, |xh| self.0.matmat(xh.view())) }
    fn conj_matmat_device(&self, ctx: &Context, x: rc_matrix, y: rc_matrix) -> Option<Result<()>> {
        Some(host_product::<Op::A>(ctx, x, y, |xh| self.0.conj_matmat(xh.view())))
    }
}

struct Shim<'a, D: DeviceOperator> {
    op: &'a D,
    ctx: &'a Context,
}
fn status_of(e: &RustyCompressionError) -> rc_status {
    match e {
        RustyCompressionError::LinalgError(_) => RC_LINALG_ERROR,
        RustyCompressionError::CompressionError => RC_COMPRESSION_ERROR,
        RustyCompressionError::LayoutError => RC_LAYOUT_ERROR,
        RustyCompressionError::PivotedQRError => RC_PIVOTED_QR_ERROR,
        RustyCompressionError::Runtime(_) => RC_RUNTIME_ERROR,
    }
}
unsafe extern "C" fn matmat_cb<D: DeviceOperator>(user: *mut c_void, _ctx: *mut rc_context, x: rc_matrix, y: rc_matrix) -> i32 {
    let shim = &*(user as *const Shim<D>);
    // a callback must not unwind into C
    match std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| shim.op.matmat_device(shim.ctx, x, y))) {
        Ok(Ok(())) => RC_OK,
        Ok(Err(e)) => status_of(&e),
        Err(_) => RC_RUNTIME_ERROR,
    }
}
unsafe extern "C" fn conj_matmat_cb<D: DeviceOperator>(user: *mut c_void, _ctx: *mut rc_context, x: rc_matrix, y: rc_matrix) -> i32 {
    let shim = &*(user as *const Shim<D>);
    match std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| shim.op.conj_matmat_device(shim.ctx, x, y))) {
        Ok(Some(Ok(()))) => RC_OK,
        Ok(Some(Err(e))) => status_of(&e),
        Ok(None) => RC_INVALID_ARGUMENT,
        Err(_) => RC_RUNTIME_ERROR,
    }
}

/// Runs `f` with the `rc_operator` of `op` (valid for the duration of the call only).
pub fn with_table<D: DeviceOperator, T>(ctx: &Context, op: &D, needs_conj: bool, f: impl FnOnce(*const rc_operator) -> T) -> T {
    let shim = Shim { op, ctx };
    let table = rc_operator {
        rows: op.nrows() as i64,
        cols: op.ncols() as i64,
        matmat: Some(matmat_cb::<D>),
        conj_matmat: if needs_conj { Some(conj_matmat_cb::<D>) } else { None },
        user: &shim as *const Shim<D> as *mut c_void,
    };
    f(&table as *const rc_operator)
}
