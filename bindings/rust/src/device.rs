//! Context, device matrices and the upload / call / download plumbing shared by every trait implementation.
use crate::ffi::*;
use crate::types::{ConjMatMat, ConjMatVec, MatMat, MatVec, Result, RustyCompressionError, Scalar};
use ndarray::{Array1, Array2, ArrayView1, ArrayView2};
use std::cell::RefCell;
use std::ffi::CStr;
use std::marker::PhantomData;
use std::os::raw::c_void;
use std::ptr;
use std::rc::Rc;

struct RawContext(*mut rc_context);
impl Drop for RawContext {
    fn drop(&mut self) {
        unsafe { rc_destroy(self.0) };
    }
}

/// One `rc_context` (device + HIP stream + workspace arena).  Cheap to clone; one per host thread (thread local default).
#[derive(Clone)]
pub struct Context {
    raw: Rc<RawContext>,
}

thread_local! {
    static CURRENT: RefCell<Option<Context>> = RefCell::new(None);
}

impl Context {
    pub fn new(device: i32) -> Result<Self> {
        let mut raw = ptr::null_mut();
        let st = unsafe { rc_create(&mut raw, device, ptr::null_mut()) };
        if st != RC_OK {
            return Err(RustyCompressionError::Runtime(format!("rc_create failed with status {} (no MI355X visible?)", st)));
        }
        Ok(Context { raw: Rc::new(RawContext(raw)) })
    }
    /// The calling thread's context (created on device 0 at first use; `set_current` replaces it).
    pub fn current() -> Context {
        CURRENT.with(|c| {
            let mut c = c.borrow_mut();
            if c.is_none() {
                *c = Some(Context::new(0).expect("no HIP device: the engine has no CPU path"));
            }
            c.as_ref().unwrap().clone()
        })
    }
    pub fn set_current(ctx: Context) {
        CURRENT.with(|c| *c.borrow_mut() = Some(ctx));
    }
    pub fn raw(&self) -> *mut rc_context { self.raw.0 }
    pub fn synchronize(&self) -> Result<()> { self.check(unsafe { rc_synchronize(self.raw()) }) }
    pub fn set_option(&self, option: i32, value: i64) -> Result<()> { self.check(unsafe { rc_set_option(self.raw(), option, value) }) }
    /// Status -> `RustyCompressionError` (reference `src/types.rs:11-21`); argument errors panic as the reference's `assert!`s do.
    pub fn check(&self, st: rc_status) -> Result<()> {
        match st {
            RC_OK => Ok(()),
            RC_LINALG_ERROR => Err(RustyCompressionError::LinalgError(self.last_error())),
            RC_COMPRESSION_ERROR => Err(RustyCompressionError::CompressionError),
            RC_LAYOUT_ERROR => Err(RustyCompressionError::LayoutError),
            RC_PIVOTED_QR_ERROR => Err(RustyCompressionError::PivotedQRError),
            RC_INVALID_ARGUMENT => panic!("{}", self.last_error()),
            _ => Err(RustyCompressionError::Runtime(self.last_error())),
        }
    }
    fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(rc_last_error_message(self.raw())).to_string_lossy().into_owned() }
    }
}

/// Device-resident C-order matrix of scalar `A`.
pub struct DeviceMatrix<A: Scalar> {
    ctx: Context,
    ptr: *mut c_void,
    rows: usize,
    cols: usize,
    _a: PhantomData<A>,
}

impl<A: Scalar> DeviceMatrix<A> {
    pub fn zeros(ctx: &Context, rows: usize, cols: usize) -> Result<Self> {
        let mut ptr = ptr::null_mut();
        let bytes = (rows * cols).max(1) * std::mem::size_of::<A>();
        ctx.check(unsafe { rc_device_malloc(ctx.raw(), bytes, &mut ptr) })?;
        Ok(DeviceMatrix { ctx: ctx.clone(), ptr, rows, cols, _a: PhantomData })
    }
    pub fn from_view(ctx: &Context, view: ArrayView2<A>) -> Result<Self> { upload(ctx, view) }
    pub fn to_array(&self) -> Result<Array2<A>> {
        let mut out = Array2::<A>::zeros((self.rows, self.cols));
        if self.rows * self.cols > 0 {
            self.ctx.check(unsafe {
                rc_memcpy_d2h(self.ctx.raw(), out.as_mut_ptr() as *mut c_void, self.ptr, self.rows * self.cols * std::mem::size_of::<A>())
            })?;
        }
        Ok(out)
    }
    pub fn view(&self) -> rc_matrix {
        rc_matrix { data: self.ptr, rows: self.rows as i64, cols: self.cols as i64, row_stride: self.cols as i64, col_stride: 1 }
    }
    pub fn null() -> rc_matrix { rc_matrix { data: ptr::null_mut(), rows: 0, cols: 0, row_stride: 0, col_stride: 0 } }
    pub fn ctx(&self) -> &Context { &self.ctx }
    pub fn dims(&self) -> (usize, usize) { (self.rows, self.cols) }
}
impl<A: Scalar> Drop for DeviceMatrix<A> {
    fn drop(&mut self) {
        unsafe { rc_device_free(self.ctx.raw(), self.ptr) };
    }
}

pub fn upload<A: Scalar>(ctx: &Context, view: ArrayView2<A>) -> Result<DeviceMatrix<A>> {
    let owned = view.as_standard_layout();
    let m = DeviceMatrix::<A>::zeros(ctx, owned.nrows(), owned.ncols())?;
    if owned.len() > 0 {
        ctx.check(unsafe { rc_memcpy_h2d(ctx.raw(), m.ptr, owned.as_ptr() as *const c_void, owned.len() * std::mem::size_of::<A>()) })?;
    }
    Ok(m)
}

/// Host copy of a strided device view (the operands of an operator callback): one strided gather on the device into a
/// contiguous staging matrix (the permutation entry point with the identity index), then one transfer.
pub fn download_view<A: Scalar>(ctx: &Context, view: rc_matrix) -> Result<Array2<A>> {
    let (rows, cols) = (view.rows as usize, view.cols as usize);
    let stage = DeviceMatrix::<A>::zeros(ctx, rows, cols)?;
    if rows * cols > 0 {
        let idx = DeviceVec::<i64>::from_slice(ctx, &(0..cols as i64).collect::<Vec<_>>())?;
        ctx.check(unsafe { A::ffi_apply_permutation_matrix(ctx.raw(), RC_PERM_COL, view, idx.ptr as *const i64, cols as i64, stage.view()) })?;
        ctx.synchronize()?;
    }
    stage.to_array()
}
/// The reverse: a host array into a strided device view.
pub fn upload_into_view<A: Scalar>(ctx: &Context, src: ArrayView2<A>, view: rc_matrix) -> Result<()> {
    let stage = upload(ctx, src)?;
    if src.len() > 0 {
        let idx = DeviceVec::<i64>::from_slice(ctx, &(0..src.ncols() as i64).collect::<Vec<_>>())?;
        ctx.check(unsafe { A::ffi_apply_permutation_matrix(ctx.raw(), RC_PERM_COL, stage.view(), idx.ptr as *const i64, src.ncols() as i64, view) })?;
        ctx.synchronize()?;
    }
    Ok(())
}

/// Device vector of `n` values of `T` (singular values, index arrays).
pub struct DeviceVec<T: Copy + Default> {
    ctx: Context,
    pub ptr: *mut c_void,
    n: usize,
    _t: PhantomData<T>,
}
impl<T: Copy + Default> DeviceVec<T> {
    pub fn new(ctx: &Context, n: usize) -> Result<Self> {
        let mut ptr = ptr::null_mut();
        ctx.check(unsafe { rc_device_malloc(ctx.raw(), n.max(1) * std::mem::size_of::<T>(), &mut ptr) })?;
        Ok(DeviceVec { ctx: ctx.clone(), ptr, n, _t: PhantomData })
    }
    pub fn from_slice(ctx: &Context, s: &[T]) -> Result<Self> {
        let v = Self::new(ctx, s.len())?;
        if !s.is_empty() {
            ctx.check(unsafe { rc_memcpy_h2d(ctx.raw(), v.ptr, s.as_ptr() as *const c_void, s.len() * std::mem::size_of::<T>()) })?;
        }
        Ok(v)
    }
    pub fn to_vec(&self) -> Result<Vec<T>> {
        let mut h = vec![T::default(); self.n];
        if self.n > 0 {
            self.ctx.check(unsafe { rc_memcpy_d2h(self.ctx.raw(), h.as_mut_ptr() as *mut c_void, self.ptr, self.n * std::mem::size_of::<T>()) })?;
        }
        Ok(h)
    }
}
impl<T: Copy + Default> Drop for DeviceVec<T> {
    fn drop(&mut self) {
        unsafe { rc_device_free(self.ctx.raw(), self.ptr) };
    }
}

pub fn upload_indices(ctx: &Context, ind: ArrayView1<usize>) -> Result<DeviceVec<i64>> {
    let h: Vec<i64> = ind.iter().map(|&v| v as i64).collect();
    DeviceVec::<i64>::from_slice(ctx, &h)
}
pub fn download_indices(v: &DeviceVec<i64>) -> Result<Array1<usize>> { Ok(v.to_vec()?.into_iter().map(|x| x as usize).collect()) }

/// `a x` (conj = false) or `a^H x` (conj = true) for host views: upload, one GEMM, download.
pub fn product<A: Scalar>(a: ArrayView2<A>, x: ArrayView2<A>, conj: bool) -> Result<Array2<A>> {
    let ctx = Context::current();
    let (da, dx) = (upload(&ctx, a)?, upload(&ctx, x)?);
    let rows = if conj { a.ncols() } else { a.nrows() };
    let y = DeviceMatrix::<A>::zeros(&ctx, rows, x.ncols())?;
    let st = unsafe {
        if conj { A::ffi_conj_matmat(ctx.raw(), da.view(), dx.view(), y.view()) } else { A::ffi_matmat(ctx.raw(), da.view(), dx.view(), y.view()) }
    };
    ctx.check(st)?;
    y.to_array()
}

/// `c - a b` for host views: upload, ONE device GEMM with alpha = -1, beta = 1 (rc_gemm_*), download.
pub fn gemm_update<A: Scalar>(a: ArrayView2<A>, b: ArrayView2<A>, c: ArrayView2<A>) -> Result<Array2<A>> {
    let ctx = Context::current();
    let (da, db, dc) = (upload(&ctx, a)?, upload(&ctx, b)?, upload(&ctx, c)?);
    ctx.check(unsafe { A::ffi_gemm_minus(ctx.raw(), da.view(), db.view(), dc.view()) })?;
    dc.to_array()
}

// A device-resident operator: the products never leave the GPU except for the (small) right-hand sides / results.
impl<A: Scalar> MatVec for DeviceMatrix<A> {
    type A = A;
    fn nrows(&self) -> usize { self.rows }
    fn ncols(&self) -> usize { self.cols }
    fn matvec(&self, vec: ArrayView1<A>) -> Array1<A> {
        let x = vec.to_owned().into_shape((vec.len(), 1)).unwrap();
        self.matmat(x.view()).into_shape(self.rows).unwrap()
    }
}
impl<A: Scalar> ConjMatVec for DeviceMatrix<A> {
    fn conj_matvec(&self, vec: ArrayView1<A>) -> Array1<A> {
        let x = vec.to_owned().into_shape((vec.len(), 1)).unwrap();
        self.conj_matmat(x.view()).into_shape(self.cols).unwrap()
    }
}
impl<A: Scalar> MatMat for DeviceMatrix<A> {
    fn matmat(&self, mat: ArrayView2<A>) -> Array2<A> {
        let dx = upload(&self.ctx, mat).unwrap();
        let y = DeviceMatrix::<A>::zeros(&self.ctx, self.rows, mat.ncols()).unwrap();
        self.ctx.check(unsafe { A::ffi_matmat(self.ctx.raw(), self.view(), dx.view(), y.view()) }).unwrap();
        y.to_array().unwrap()
    }
}
impl<A: Scalar> ConjMatMat for DeviceMatrix<A> {
    fn conj_matmat(&self, mat: ArrayView2<A>) -> Array2<A> {
        let dx = upload(&self.ctx, mat).unwrap();
        let y = DeviceMatrix::<A>::zeros(&self.ctx, self.cols, mat.ncols()).unwrap();
        self.ctx.check(unsafe { A::ffi_conj_matmat(self.ctx.raw(), self.view(), dx.view(), y.view()) }).unwrap();
        y.to_array().unwrap()
    }
}
