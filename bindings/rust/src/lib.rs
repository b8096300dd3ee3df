//! The reference crate's trait surface (rusty-compression v0.1.1, `src/lib.rs:90-102`) over the C ABI of
//! librusty_compression_amd.so.  No arithmetic happens on the host: a `DeviceMatrix` owns device memory,
//! every trait method is one FFI call.  Host `ndarray` views are uploaded / downloaded at the edges
//! (`DeviceMatrix::from_view`, `DeviceMatrix::to_array`).
//!
//! Differences from the reference that a maintainer adopting this must know:
//! * `MatMat` / `ConjMatMat` are real traits here, not blanket impls over `MatVec` (reference
//!   `src/types.rs:145-146`), so the dense implementation is one GEMM.
//! * The RNG is a Philox (seed, offset) pair; pass an explicit `Omega` to reproduce reference streams.
pub mod ffi;

use ffi::*;
use ndarray::{Array1, Array2, ArrayView2};
use std::ffi::CStr;
use std::os::raw::c_void;
use std::ptr;

#[derive(thiserror::Error, Debug)]
pub enum RustyCompressionError {
    #[error("Lapack Error")]
    LinalgError,
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

pub enum CompressionType {
    ADAPTIVE(f64),
    RANK(usize),
}

/// One `rc_context` (device + HIP stream + workspace arena).
pub struct Context {
    raw: *mut rc_context,
}

impl Context {
    pub fn new(device: i32) -> Result<Self> {
        let mut raw = ptr::null_mut();
        let st = unsafe { rc_create(&mut raw, device, ptr::null_mut()) };
        if st != RC_OK {
            return Err(RustyCompressionError::Runtime(format!("rc_create: {}", st)));
        }
        Ok(Context { raw })
    }
    fn check(&self, st: rc_status) -> Result<()> {
        match st {
            RC_OK => Ok(()),
            RC_LINALG_ERROR => Err(RustyCompressionError::LinalgError),
            RC_COMPRESSION_ERROR => Err(RustyCompressionError::CompressionError),
            RC_LAYOUT_ERROR => Err(RustyCompressionError::LayoutError),
            RC_PIVOTED_QR_ERROR => Err(RustyCompressionError::PivotedQRError),
            RC_INVALID_ARGUMENT => panic!("{}", self.last_error()), // the reference asserts
            _ => Err(RustyCompressionError::Runtime(self.last_error())),
        }
    }
    fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(rc_last_error_message(self.raw)).to_string_lossy().into_owned() }
    }
}
impl Drop for Context {
    fn drop(&mut self) {
        unsafe { rc_destroy(self.raw) };
    }
}

/// Device-resident, C-order f64 matrix.
pub struct DeviceMatrix<'c> {
    ctx: &'c Context,
    ptr: *mut c_void,
    rows: usize,
    cols: usize,
}

impl<'c> DeviceMatrix<'c> {
    pub fn zeros(ctx: &'c Context, rows: usize, cols: usize) -> Result<Self> {
        let mut ptr = ptr::null_mut();
        ctx.check(unsafe { rc_device_malloc(ctx.raw, rows * cols * 8, &mut ptr) })?;
        Ok(DeviceMatrix { ctx, ptr, rows, cols })
    }
    pub fn from_view(ctx: &'c Context, view: ArrayView2<f64>) -> Result<Self> {
        let owned = view.as_standard_layout();
        let m = Self::zeros(ctx, owned.nrows(), owned.ncols())?;
        ctx.check(unsafe { rc_memcpy_h2d(ctx.raw, m.ptr, owned.as_ptr() as *const c_void, owned.len() * 8) })?;
        Ok(m)
    }
    pub fn to_array(&self) -> Result<Array2<f64>> {
        let mut out = Array2::<f64>::zeros((self.rows, self.cols));
        self.ctx.check(unsafe { rc_memcpy_d2h(self.ctx.raw, out.as_mut_ptr() as *mut c_void, self.ptr, self.rows * self.cols * 8) })?;
        Ok(out)
    }
    fn view(&self) -> rc_matrix {
        rc_matrix { data: self.ptr, rows: self.rows as i64, cols: self.cols as i64, row_stride: self.cols as i64, col_stride: 1 }
    }
    fn null() -> rc_matrix {
        rc_matrix { data: ptr::null_mut(), rows: 0, cols: 0, row_stride: 0, col_stride: 0 }
    }
    pub fn nrows(&self) -> usize { self.rows }
    pub fn ncols(&self) -> usize { self.cols }
}
impl<'c> Drop for DeviceMatrix<'c> {
    fn drop(&mut self) {
        unsafe { rc_device_free(self.ctx.raw, self.ptr) };
    }
}

/// `MatMat` (reference `src/types.rs:58-71`) as a real trait.
pub trait MatMat<'c> {
    fn matmat(&self, x: &DeviceMatrix<'c>) -> Result<DeviceMatrix<'c>>;
}
/// `ConjMatMat` (reference `src/types.rs:88-101`).
pub trait ConjMatMat<'c>: MatMat<'c> {
    fn conj_matmat(&self, x: &DeviceMatrix<'c>) -> Result<DeviceMatrix<'c>>;
}
impl<'c> MatMat<'c> for DeviceMatrix<'c> {
    fn matmat(&self, x: &DeviceMatrix<'c>) -> Result<DeviceMatrix<'c>> {
        let y = DeviceMatrix::zeros(self.ctx, self.rows, x.cols)?;
        self.ctx.check(unsafe { rc_matmat_f64(self.ctx.raw, self.view(), x.view(), y.view()) })?;
        Ok(y)
    }
}
impl<'c> ConjMatMat<'c> for DeviceMatrix<'c> {
    fn conj_matmat(&self, x: &DeviceMatrix<'c>) -> Result<DeviceMatrix<'c>> {
        let y = DeviceMatrix::zeros(self.ctx, self.cols, x.cols)?;
        self.ctx.check(unsafe { rc_conj_matmat_f64(self.ctx.raw, self.view(), x.view(), y.view()) })?;
        Ok(y)
    }
}

/// `SampleRange::sample_range_by_rank` (reference `src/random_sampling.rs:103-118`).
pub fn sample_range_by_rank<'c>(op: &DeviceMatrix<'c>, k: usize, p: usize, seed: u64) -> Result<DeviceMatrix<'c>> {
    let q = DeviceMatrix::zeros(op.ctx, op.rows, k.min(op.rows).min(k + p))?;
    op.ctx.check(unsafe { rc_sample_range_by_rank_f64(op.ctx.raw, op.view(), k as i64, p as i64, DeviceMatrix::null(), seed, q.view()) })?;
    Ok(q)
}

/// `struct QR` (reference `src/qr.rs:31-40`).
pub struct QR<'c> {
    pub q: DeviceMatrix<'c>,
    pub r: DeviceMatrix<'c>,
    pub ind: Array1<usize>,
}
pub struct ColumnID<'c> {
    pub c: DeviceMatrix<'c>,
    pub z: DeviceMatrix<'c>,
    pub col_ind: Array1<usize>,
}
pub struct SVD<'c> {
    pub u: DeviceMatrix<'c>,
    pub s: Array1<f64>,
    pub vt: DeviceMatrix<'c>,
}

fn dev_indices(ctx: &Context, n: usize) -> Result<*mut c_void> {
    let mut p = ptr::null_mut();
    ctx.check(unsafe { rc_device_malloc(ctx.raw, n * 8, &mut p) })?;
    Ok(p)
}
fn fetch_indices(ctx: &Context, p: *mut c_void, n: usize) -> Result<Array1<usize>> {
    let mut h = vec![0i64; n];
    ctx.check(unsafe { rc_memcpy_d2h(ctx.raw, h.as_mut_ptr() as *mut c_void, p, n * 8) })?;
    unsafe { rc_device_free(ctx.raw, p) };
    Ok(h.into_iter().map(|v| v as usize).collect())
}

impl<'c> QR<'c> {
    /// `QRTraits::compute_from` (reference `src/qr.rs:251-253` -> `src/pivoted_qr.rs:25-31`).
    pub fn compute_from(a: &DeviceMatrix<'c>) -> Result<QR<'c>> {
        let k = a.rows.min(a.cols);
        let q = DeviceMatrix::zeros(a.ctx, a.rows, k)?;
        let r = DeviceMatrix::zeros(a.ctx, k, a.cols)?;
        let ind = dev_indices(a.ctx, a.cols)?;
        a.ctx.check(unsafe { rc_pivoted_qr_f64(a.ctx.raw, a.view(), q.view(), r.view(), ind as *mut i64) })?;
        Ok(QR { q, r, ind: fetch_indices(a.ctx, ind, a.cols)? })
    }
    /// `QRTraits::compute_from_range_estimate` (reference `src/qr.rs:311-323`).
    pub fn compute_from_range_estimate(range: &DeviceMatrix<'c>, op: &DeviceMatrix<'c>) -> Result<QR<'c>> {
        let k = range.cols.min(op.cols);
        let q = DeviceMatrix::zeros(op.ctx, op.rows, k)?;
        let r = DeviceMatrix::zeros(op.ctx, k, op.cols)?;
        let ind = dev_indices(op.ctx, op.cols)?;
        op.ctx.check(unsafe { rc_qr_from_range_estimate_f64(op.ctx.raw, range.view(), op.view(), q.view(), r.view(), ind as *mut i64) })?;
        Ok(QR { q, r, ind: fetch_indices(op.ctx, ind, op.cols)? })
    }
    /// `QRTraits::column_id` (reference `src/qr.rs:270-309`).
    pub fn column_id(&self) -> Result<ColumnID<'c>> {
        let ctx = self.q.ctx;
        let c = DeviceMatrix::zeros(ctx, self.q.rows, self.q.cols)?;
        let z = DeviceMatrix::zeros(ctx, self.q.cols, self.r.cols)?;
        let ind: Vec<i64> = self.ind.iter().map(|&v| v as i64).collect();
        let d = dev_indices(ctx, ind.len())?;
        ctx.check(unsafe { rc_memcpy_h2d(ctx.raw, d, ind.as_ptr() as *const c_void, ind.len() * 8) })?;
        ctx.check(unsafe { rc_qr_column_id_f64(ctx.raw, self.q.view(), self.r.view(), d as *const i64, c.view(), z.view()) })?;
        unsafe { rc_device_free(ctx.raw, d) };
        Ok(ColumnID { c, z, col_ind: self.ind.clone() })
    }
}

impl<'c> SVD<'c> {
    /// `SVDTraits::compute_from_range_estimate` (reference `src/svd.rs:171-183`).
    pub fn compute_from_range_estimate(range: &DeviceMatrix<'c>, op: &DeviceMatrix<'c>) -> Result<SVD<'c>> {
        let r = range.cols.min(op.cols);
        let u = DeviceMatrix::zeros(op.ctx, op.rows, r)?;
        let vt = DeviceMatrix::zeros(op.ctx, r, op.cols)?;
        let mut s_dev = ptr::null_mut();
        op.ctx.check(unsafe { rc_device_malloc(op.ctx.raw, r * 8, &mut s_dev) })?;
        op.ctx.check(unsafe { rc_svd_from_range_estimate_f64(op.ctx.raw, range.view(), op.view(), u.view(), s_dev as *mut f64, vt.view()) })?;
        let mut s = Array1::<f64>::zeros(r);
        op.ctx.check(unsafe { rc_memcpy_d2h(op.ctx.raw, s.as_mut_ptr() as *mut c_void, s_dev, r * 8) })?;
        unsafe { rc_device_free(op.ctx.raw, s_dev) };
        Ok(SVD { u, s, vt })
    }
}
