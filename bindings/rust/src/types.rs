//! Traits and errors of the reference's `types` module (`src/types.rs`) + the scalar dispatch table of the C ABI.
use crate::device::{self, Context};
use crate::ffi::*;
use ndarray::{Array1, Array2, ArrayBase, ArrayView1, ArrayView2, Data, Ix2};
use std::os::raw::c_void;
use thiserror::Error;

#[allow(non_camel_case_types)]
pub type c32 = num_complex::Complex<f32>;
#[allow(non_camel_case_types)]
pub type c64 = num_complex::Complex<f64>;

/// `RustyCompressionError` (reference `src/types.rs:11-21`).  `LinalgError` carries the library's message instead of
/// ndarray-linalg's error type; `Runtime` (HIP / RCCL failure) has no reference counterpart.
#[derive(Error, Debug)]
pub enum RustyCompressionError {
    #[error("Lapack Error")]
    LinalgError(String),
    #[error("Could not compress to desired tolerance")]
    CompressionError,
    #[error("Incompatible memory layout")]
    LayoutError,
    #[error("Pivoted QR failed")]
    PivotedQRError,
    #[error("HIP runtime error: {0}")]
    Runtime(String),
}

pub type Result<T> = std::result::Result<T, RustyCompressionError>;

/// The four scalar types of the reference (`ndarray_linalg::Scalar` there).  Besides the real type it carries the
/// table of typed C-ABI entry points (`rc_*_f32`, `_f64`, `_c32`, `_c64`), so generic code dispatches statically.
pub trait Scalar:
    Copy + Clone + Default + PartialEq + std::fmt::Debug + num_traits::Zero + num_traits::One + std::ops::Div<Output = Self> + std::ops::Sub<Output = Self> + std::ops::MulAssign + 'static
{
    type Real: Copy + Clone + Default + PartialOrd + std::fmt::Debug + num_traits::Float + 'static;
    fn conj(&self) -> Self;
    fn abs(&self) -> Self::Real;
    fn from_real(re: Self::Real) -> Self;
    fn real_to_f64(r: Self::Real) -> f64;

    unsafe fn ffi_random_gaussian(ctx: *mut rc_context, out: rc_matrix, seed: u64, offset: u64) -> rc_status;
    unsafe fn ffi_matmat(ctx: *mut rc_context, a: rc_matrix, x: rc_matrix, y: rc_matrix) -> rc_status;
    unsafe fn ffi_conj_matmat(ctx: *mut rc_context, a: rc_matrix, x: rc_matrix, y: rc_matrix) -> rc_status;
    /// c <- c - a b (rc_gemm_* with alpha = -1, beta = 1)
    unsafe fn ffi_gemm_minus(ctx: *mut rc_context, a: rc_matrix, b: rc_matrix, c: rc_matrix) -> rc_status;
    unsafe fn ffi_rel_diff_fro(ctx: *mut rc_context, first: rc_matrix, second: rc_matrix, out: *mut Self::Real) -> rc_status;
    unsafe fn ffi_apply_permutation_matrix(ctx: *mut rc_context, mode: i32, input: rc_matrix, perm: *const i64, n: i64, out: rc_matrix) -> rc_status;
    unsafe fn ffi_pivoted_qr(ctx: *mut rc_context, a: rc_matrix, q: rc_matrix, r: rc_matrix, ind: *mut i64) -> rc_status;
    unsafe fn ffi_pivoted_lq(ctx: *mut rc_context, a: rc_matrix, l: rc_matrix, q: rc_matrix, ind: *mut i64) -> rc_status;
    unsafe fn ffi_compute_svd(ctx: *mut rc_context, a: rc_matrix, u: rc_matrix, s: *mut Self::Real, vt: rc_matrix) -> rc_status;
    /// the LAPACK seam (`$qrf` at reference `src/pivoted_qr.rs:139-172`, `lax::Lapack::q` at `:104-108`, `solve_triangular` at `src/qr.rs:298`)
    unsafe fn ffi_geqp3(ctx: *mut rc_context, a: rc_matrix, kmax: i64, jpvt: *mut i64, tau: *mut std::os::raw::c_void) -> rc_status;
    unsafe fn ffi_orgqr(ctx: *mut rc_context, a: rc_matrix, tau: *const std::os::raw::c_void, k: i64, q: rc_matrix) -> rc_status;
    unsafe fn ffi_trsm_upper(ctx: *mut rc_context, t: rc_matrix, b: rc_matrix) -> rc_status;
    unsafe fn ffi_qr_to_mat(ctx: *mut rc_context, q: rc_matrix, r: rc_matrix, ind: *const i64, out: rc_matrix) -> rc_status;
    unsafe fn ffi_lq_to_mat(ctx: *mut rc_context, l: rc_matrix, q: rc_matrix, ind: *const i64, out: rc_matrix) -> rc_status;
    unsafe fn ffi_qr_column_id(ctx: *mut rc_context, q: rc_matrix, r: rc_matrix, ind: *const i64, c: rc_matrix, z: rc_matrix) -> rc_status;
    unsafe fn ffi_lq_row_id(ctx: *mut rc_context, l: rc_matrix, q: rc_matrix, ind: *const i64, x: rc_matrix, r_rows: rc_matrix) -> rc_status;
    unsafe fn ffi_qr_from_range_estimate(ctx: *mut rc_context, range: rc_matrix, a: rc_matrix, q: rc_matrix, r: rc_matrix, ind: *mut i64) -> rc_status;
    unsafe fn ffi_svd_to_mat(ctx: *mut rc_context, u: rc_matrix, s: *const Self::Real, vt: rc_matrix, out: rc_matrix) -> rc_status;
    unsafe fn ffi_svd_to_qr(ctx: *mut rc_context, u: rc_matrix, s: *const Self::Real, vt: rc_matrix, q: rc_matrix, r: rc_matrix, ind: *mut i64) -> rc_status;
    unsafe fn ffi_svd_from_range_estimate(ctx: *mut rc_context, range: rc_matrix, a: rc_matrix, u: rc_matrix, s: *mut Self::Real, vt: rc_matrix) -> rc_status;
    unsafe fn ffi_column_id_two_sided(ctx: *mut rc_context, c: rc_matrix, c_out: rc_matrix, x: rc_matrix, row_ind: *mut i64) -> rc_status;
    unsafe fn ffi_row_id_two_sided(ctx: *mut rc_context, r: rc_matrix, x: rc_matrix, r_out: rc_matrix, col_ind: *mut i64) -> rc_status;
    unsafe fn ffi_max_col_norm(ctx: *mut rc_context, y: rc_matrix, out: *mut Self::Real) -> rc_status;
    unsafe fn ffi_sample_range_by_rank(ctx: *mut rc_context, a: rc_matrix, k: i64, p: i64, omega: rc_matrix, seed: u64, q: rc_matrix) -> rc_status;
    unsafe fn ffi_sample_range_power_iteration(ctx: *mut rc_context, a: rc_matrix, k: i64, p: i64, it: i64, omega: rc_matrix, seed: u64, q: rc_matrix) -> rc_status;
    #[allow(clippy::too_many_arguments)]
    unsafe fn ffi_sample_range_adaptive(ctx: *mut rc_context, a: rc_matrix, rel_tol: f64, sample_size: i64, omegas: rc_matrix, seed: u64, q_cap: rc_matrix,
                                        rank: *mut i64, hist_rank: *mut i64, hist_res: *mut f64, hist_cap: i64, hist_len: *mut i64) -> rc_status;
    unsafe fn ffi_column_id_rank(ctx: *mut rc_context, a: rc_matrix, k: i64, c: rc_matrix, z: rc_matrix, col_ind: *mut i64) -> rc_status;
}

macro_rules! impl_scalar {
    ($t:ty, $real:ty, $suf:ident, $conj:expr, $abs:expr, $from_real:expr, $minus_one:expr, $one:expr) => {
        paste::paste! {
        impl Scalar for $t {
            type Real = $real;
            fn conj(&self) -> Self { ($conj)(*self) }
            fn abs(&self) -> $real { ($abs)(*self) }
            fn from_real(re: $real) -> Self { ($from_real)(re) }
            fn real_to_f64(r: $real) -> f64 { r as f64 }
            unsafe fn ffi_random_gaussian(ctx: *mut rc_context, out: rc_matrix, seed: u64, offset: u64) -> rc_status { [<rc_random_gaussian_ $suf>](ctx, out, seed, offset) }
            unsafe fn ffi_matmat(ctx: *mut rc_context, a: rc_matrix, x: rc_matrix, y: rc_matrix) -> rc_status { [<rc_matmat_ $suf>](ctx, a, x, y) }
            unsafe fn ffi_conj_matmat(ctx: *mut rc_context, a: rc_matrix, x: rc_matrix, y: rc_matrix) -> rc_status { [<rc_conj_matmat_ $suf>](ctx, a, x, y) }
            unsafe fn ffi_gemm_minus(ctx: *mut rc_context, a: rc_matrix, b: rc_matrix, c: rc_matrix) -> rc_status { [<rc_gemm_ $suf>](ctx, 0, 0, $minus_one, a, b, $one, c) }
            unsafe fn ffi_rel_diff_fro(ctx: *mut rc_context, first: rc_matrix, second: rc_matrix, out: *mut $real) -> rc_status { [<rc_rel_diff_fro_ $suf>](ctx, first, second, out) }
            unsafe fn ffi_apply_permutation_matrix(ctx: *mut rc_context, mode: i32, input: rc_matrix, perm: *const i64, n: i64, out: rc_matrix) -> rc_status { [<rc_apply_permutation_matrix_ $suf>](ctx, mode, input, perm, n, out) }
            unsafe fn ffi_pivoted_qr(ctx: *mut rc_context, a: rc_matrix, q: rc_matrix, r: rc_matrix, ind: *mut i64) -> rc_status { [<rc_pivoted_qr_ $suf>](ctx, a, q, r, ind) }
            unsafe fn ffi_pivoted_lq(ctx: *mut rc_context, a: rc_matrix, l: rc_matrix, q: rc_matrix, ind: *mut i64) -> rc_status { [<rc_pivoted_lq_ $suf>](ctx, a, l, q, ind) }
            unsafe fn ffi_compute_svd(ctx: *mut rc_context, a: rc_matrix, u: rc_matrix, s: *mut $real, vt: rc_matrix) -> rc_status { [<rc_compute_svd_ $suf>](ctx, a, u, s, vt) }
            unsafe fn ffi_geqp3(ctx: *mut rc_context, a: rc_matrix, kmax: i64, jpvt: *mut i64, tau: *mut std::os::raw::c_void) -> rc_status { [<rc_geqp3_ $suf>](ctx, a, kmax, jpvt, tau as *mut _) }
            unsafe fn ffi_orgqr(ctx: *mut rc_context, a: rc_matrix, tau: *const std::os::raw::c_void, k: i64, q: rc_matrix) -> rc_status { [<rc_orgqr_ $suf>](ctx, a, tau as *const _, k, q) }
            unsafe fn ffi_trsm_upper(ctx: *mut rc_context, t: rc_matrix, b: rc_matrix) -> rc_status { [<rc_trsm_upper_ $suf>](ctx, t, b) }
            unsafe fn ffi_qr_to_mat(ctx: *mut rc_context, q: rc_matrix, r: rc_matrix, ind: *const i64, out: rc_matrix) -> rc_status { [<rc_qr_to_mat_ $suf>](ctx, q, r, ind, out) }
            unsafe fn ffi_lq_to_mat(ctx: *mut rc_context, l: rc_matrix, q: rc_matrix, ind: *const i64, out: rc_matrix) -> rc_status { [<rc_lq_to_mat_ $suf>](ctx, l, q, ind, out) }
            unsafe fn ffi_qr_column_id(ctx: *mut rc_context, q: rc_matrix, r: rc_matrix, ind: *const i64, c: rc_matrix, z: rc_matrix) -> rc_status { [<rc_qr_column_id_ $suf>](ctx, q, r, ind, c, z) }
            unsafe fn ffi_lq_row_id(ctx: *mut rc_context, l: rc_matrix, q: rc_matrix, ind: *const i64, x: rc_matrix, r_rows: rc_matrix) -> rc_status { [<rc_lq_row_id_ $suf>](ctx, l, q, ind, x, r_rows) }
            unsafe fn ffi_qr_from_range_estimate(ctx: *mut rc_context, range: rc_matrix, a: rc_matrix, q: rc_matrix, r: rc_matrix, ind: *mut i64) -> rc_status { [<rc_qr_from_range_estimate_ $suf>](ctx, range, a, q, r, ind) }
            unsafe fn ffi_svd_to_mat(ctx: *mut rc_context, u: rc_matrix, s: *const $real, vt: rc_matrix, out: rc_matrix) -> rc_status { [<rc_svd_to_mat_ $suf>](ctx, u, s, vt, out) }
            unsafe fn ffi_svd_to_qr(ctx: *mut rc_context, u: rc_matrix, s: *const $real, vt: rc_matrix, q: rc_matrix, r: rc_matrix, ind: *mut i64) -> rc_status { [<rc_svd_to_qr_ $suf>](ctx, u, s, vt, q, r, ind) }
            unsafe fn ffi_svd_from_range_estimate(ctx: *mut rc_context, range: rc_matrix, a: rc_matrix, u: rc_matrix, s: *mut $real, vt: rc_matrix) -> rc_status { [<rc_svd_from_range_estimate_ $suf>](ctx, range, a, u, s, vt) }
            unsafe fn ffi_column_id_two_sided(ctx: *mut rc_context, c: rc_matrix, c_out: rc_matrix, x: rc_matrix, row_ind: *mut i64) -> rc_status { [<rc_column_id_two_sided_ $suf>](ctx, c, c_out, x, row_ind) }
            unsafe fn ffi_row_id_two_sided(ctx: *mut rc_context, r: rc_matrix, x: rc_matrix, r_out: rc_matrix, col_ind: *mut i64) -> rc_status { [<rc_row_id_two_sided_ $suf>](ctx, r, x, r_out, col_ind) }
            unsafe fn ffi_max_col_norm(ctx: *mut rc_context, y: rc_matrix, out: *mut $real) -> rc_status { [<rc_max_col_norm_ $suf>](ctx, y, out) }
            unsafe fn ffi_sample_range_by_rank(ctx: *mut rc_context, a: rc_matrix, k: i64, p: i64, omega: rc_matrix, seed: u64, q: rc_matrix) -> rc_status { [<rc_sample_range_by_rank_ $suf>](ctx, a, k, p, omega, seed, q) }
            unsafe fn ffi_sample_range_power_iteration(ctx: *mut rc_context, a: rc_matrix, k: i64, p: i64, it: i64, omega: rc_matrix, seed: u64, q: rc_matrix) -> rc_status { [<rc_sample_range_power_iteration_ $suf>](ctx, a, k, p, it, omega, seed, q) }
            unsafe fn ffi_sample_range_adaptive(ctx: *mut rc_context, a: rc_matrix, rel_tol: f64, sample_size: i64, omegas: rc_matrix, seed: u64, q_cap: rc_matrix,
                                                rank: *mut i64, hist_rank: *mut i64, hist_res: *mut f64, hist_cap: i64, hist_len: *mut i64) -> rc_status {
                [<rc_sample_range_adaptive_ $suf>](ctx, a, rel_tol, sample_size, omegas, seed, q_cap, rank, hist_rank, hist_res, hist_cap, hist_len)
            }
            unsafe fn ffi_column_id_rank(ctx: *mut rc_context, a: rc_matrix, k: i64, c: rc_matrix, z: rc_matrix, col_ind: *mut i64) -> rc_status { [<rc_column_id_rank_ $suf>](ctx, a, k, c, z, col_ind) }
        }
        }
    };
}

impl_scalar!(f32, f32, f32, |x: f32| x, |x: f32| x.abs(), |r: f32| r, -1.0f32, 1.0f32);
impl_scalar!(f64, f64, f64, |x: f64| x, |x: f64| x.abs(), |r: f64| r, -1.0f64, 1.0f64);
#[cfg(feature = "complex")]
impl_scalar!(c32, f32, c32, |x: c32| x.conj(), |x: c32| x.norm(), |r: f32| c32::new(r, 0.0), rc_complex32 { re: -1.0, im: 0.0 }, rc_complex32 { re: 1.0, im: 0.0 });
#[cfg(feature = "complex")]
impl_scalar!(c64, f64, c64, |x: c64| x.conj(), |x: c64| x.norm(), |r: f64| c64::new(r, 0.0), rc_complex64 { re: -1.0, im: 0.0 }, rc_complex64 { re: 1.0, im: 0.0 });

/// `Apply` (reference `src/types.rs:25-29`).
pub trait Apply<A, Lhs> {
    type Output;
    fn dot(&self, lhs: &Lhs) -> Self::Output;
}
/// `RApply` (reference `src/types.rs:31-35`).
pub trait RApply<A, Lhs> {
    type Output;
    fn dot(&self, lhs: &Lhs) -> Self::Output;
}

/// `MatVec` (reference `src/types.rs:40-51`).
pub trait MatVec {
    type A: Scalar;
    fn nrows(&self) -> usize;
    fn ncols(&self) -> usize;
    fn matvec(&self, mat: ArrayView1<Self::A>) -> Array1<Self::A>;
}

/// `MatMat` (reference `src/types.rs:58-71`): the default is the reference's per-column loop; dense matrices override it
/// with one GEMM on the device.
pub trait MatMat: MatVec {
    fn matmat(&self, mat: ArrayView2<Self::A>) -> Array2<Self::A> {
        let mut output = Array2::<Self::A>::zeros((self.nrows(), mat.ncols()));
        for (index, col) in mat.axis_iter(ndarray::Axis(1)).enumerate() {
            output.index_axis_mut(ndarray::Axis(1), index).assign(&self.matvec(col));
        }
        output
    }
}

/// `ConjMatVec` (reference `src/types.rs:77-81`).
pub trait ConjMatVec: MatVec {
    fn conj_matvec(&self, vec: ArrayView1<Self::A>) -> Array1<Self::A>;
}

/// `ConjMatMat` (reference `src/types.rs:88-101`).
pub trait ConjMatMat: MatMat + ConjMatVec {
    fn conj_matmat(&self, mat: ArrayView2<Self::A>) -> Array2<Self::A> {
        let mut output = Array2::<Self::A>::zeros((self.ncols(), mat.ncols()));
        for (index, col) in mat.axis_iter(ndarray::Axis(1)).enumerate() {
            output.index_axis_mut(ndarray::Axis(1), index).assign(&self.conj_matvec(col));
        }
        output
    }
}

// Dense host matrices (reference `src/types.rs:103-133`, `:145-146`): every product is one device GEMM.
impl<A: Scalar, S: Data<Elem = A>> MatVec for ArrayBase<S, Ix2> {
    type A = A;
    fn nrows(&self) -> usize { self.nrows() }
    fn ncols(&self) -> usize { self.ncols() }
    fn matvec(&self, vec: ArrayView1<A>) -> Array1<A> {
        let n = vec.len();
        let x = vec.to_owned().into_shape((n, 1)).expect("vector to column");
        let y = device::product::<A>(self.view(), x.view(), false).expect("rc_matmat failed");
        y.into_shape(self.nrows()).expect("column to vector")
    }
}
impl<A: Scalar, S: Data<Elem = A>> ConjMatVec for ArrayBase<S, Ix2> {
    fn conj_matvec(&self, vec: ArrayView1<A>) -> Array1<A> {
        let n = vec.len();
        let x = vec.to_owned().into_shape((n, 1)).expect("vector to column");
        let y = device::product::<A>(self.view(), x.view(), true).expect("rc_conj_matmat failed");
        y.into_shape(self.ncols()).expect("column to vector")
    }
}
impl<A: Scalar, S: Data<Elem = A>> MatMat for ArrayBase<S, Ix2> {
    fn matmat(&self, mat: ArrayView2<A>) -> Array2<A> { device::product::<A>(self.view(), mat, false).expect("rc_matmat failed") }
}
impl<A: Scalar, S: Data<Elem = A>> ConjMatMat for ArrayBase<S, Ix2> {
    fn conj_matmat(&self, mat: ArrayView2<A>) -> Array2<A> { device::product::<A>(self.view(), mat, true).expect("rc_conj_matmat failed") }
}

/// `RelDiff` (reference `src/types.rs:162-196`).
pub trait RelDiff {
    type A: Scalar;
    fn rel_diff_fro(first: ArrayView2<Self::A>, second: ArrayView2<Self::A>) -> <<Self as RelDiff>::A as Scalar>::Real;
    fn rel_diff_l2(first: ArrayView1<Self::A>, second: ArrayView1<Self::A>) -> <<Self as RelDiff>::A as Scalar>::Real;
}

impl<T: Scalar> RelDiff for T {
    type A = T;
    fn rel_diff_fro(first: ArrayView2<T>, second: ArrayView2<T>) -> T::Real {
        let ctx = Context::current();
        let (a, b) = (device::upload(&ctx, first).unwrap(), device::upload(&ctx, second).unwrap());
        let mut out = T::Real::default();
        ctx.check(unsafe { T::ffi_rel_diff_fro(ctx.raw(), a.view(), b.view(), &mut out as *mut T::Real) }).unwrap();
        out
    }
    fn rel_diff_l2(first: ArrayView1<T>, second: ArrayView1<T>) -> T::Real {
        let n = first.len();
        let a = first.to_owned().into_shape((n, 1)).unwrap();
        let b = second.to_owned().into_shape((n, 1)).unwrap();
        Self::rel_diff_fro(a.view(), b.view())
    }
}

pub(crate) fn null_ptr() -> *mut c_void { std::ptr::null_mut() }
