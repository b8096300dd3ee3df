//! `qr` module of the reference (`src/qr.rs`): `QR`, `LQ`, `QRTraits`, `LQTraits`.
use crate::col_interp_decomp::ColumnID;
use crate::device::{self, Context, DeviceMatrix};
use crate::pivoted_qr::PivotedQR;
use crate::row_interp_decomp::RowID;
use crate::types::{ConjMatMat, Result, RustyCompressionError, Scalar};
use crate::CompressionType;
use ndarray::{s, Array1, Array2, ArrayView1, ArrayView2, ArrayViewMut1, ArrayViewMut2};

/// reference `src/qr.rs:31-40`: `ind[j] = k` <=> column j of Q R is column k of A
pub struct QR<A: Scalar> {
    pub q: Array2<A>,
    pub r: Array2<A>,
    pub ind: Array1<usize>,
}
/// reference `src/qr.rs:42-51`
pub struct LQ<A: Scalar> {
    pub l: Array2<A>,
    pub q: Array2<A>,
    pub ind: Array1<usize>,
}

/// reference `src/qr.rs:141-238`
pub trait QRTraits {
    type A: Scalar;
    fn nrows(&self) -> usize { self.get_q().nrows() }
    fn ncols(&self) -> usize { self.get_r().ncols() }
    fn rank(&self) -> usize { self.get_q().ncols() }
    /// Q (R P^T) (reference `src/qr.rs:160-166`)
    fn to_mat(&self) -> Array2<Self::A> {
        let ctx = Context::current();
        let q = device::upload(&ctx, self.get_q()).unwrap();
        let r = device::upload(&ctx, self.get_r()).unwrap();
        let ind = device::upload_indices(&ctx, self.get_ind()).unwrap();
        let out = DeviceMatrix::<Self::A>::zeros(&ctx, self.nrows(), self.ncols()).unwrap();
        ctx.check(unsafe { Self::A::ffi_qr_to_mat(ctx.raw(), q.view(), r.view(), ind.ptr as *const i64, out.view()) }).unwrap();
        out.to_array().unwrap()
    }
    /// reference `src/qr.rs:169-184`: slices; `ind` keeps its full length
    fn compress_qr_rank(&self, mut max_rank: usize) -> Result<QR<Self::A>> {
        let (q, r) = (self.get_q(), self.get_r());
        if max_rank > q.ncols() {
            max_rank = q.ncols()
        }
        Ok(QR { q: q.slice(s![.., 0..max_rank]).to_owned(), r: r.slice(s![0..max_rank, ..]).to_owned(), ind: self.get_ind().to_owned() })
    }
    /// reference `src/qr.rs:187-200`: first i with |r_ii / r_00| < tol, `CompressionError` if none
    fn compress_qr_tolerance(&self, tol: f64) -> Result<QR<Self::A>> {
        assert!((tol < 1.0) && (0.0 <= tol), "Require 0 <= tol < 1.0");
        let r = self.get_r();
        let pos = r.diag().iter().position(|&item| <Self::A as Scalar>::real_to_f64((item / r[[0, 0]]).abs()) < tol);
        match pos {
            Some(index) => self.compress_qr_rank(index),
            None => Err(RustyCompressionError::CompressionError),
        }
    }
    /// reference `src/qr.rs:203-208`
    fn compress(&self, compression_type: CompressionType) -> Result<QR<Self::A>> {
        match compression_type {
            CompressionType::ADAPTIVE(tol) => self.compress_qr_tolerance(tol),
            CompressionType::RANK(rank) => self.compress_qr_rank(rank),
        }
    }
    fn column_id(&self) -> Result<ColumnID<Self::A>>;
    fn compute_from(arr: ArrayView2<Self::A>) -> Result<QR<Self::A>>;
    fn compute_from_range_estimate<Op: ConjMatMat<A = Self::A>>(range: ArrayView2<Self::A>, op: &Op) -> Result<QR<Self::A>>;
    fn get_q(&self) -> ArrayView2<Self::A>;
    fn get_r(&self) -> ArrayView2<Self::A>;
    fn get_ind(&self) -> ArrayView1<usize>;
    fn get_q_mut(&mut self) -> ArrayViewMut2<Self::A>;
    fn get_r_mut(&mut self) -> ArrayViewMut2<Self::A>;
    fn get_ind_mut(&mut self) -> ArrayViewMut1<usize>;
}

/// reference `src/qr.rs:54-139`
pub trait LQTraits {
    type A: Scalar;
    fn nrows(&self) -> usize { self.get_l().nrows() }
    fn ncols(&self) -> usize { self.get_q().ncols() }
    fn rank(&self) -> usize { self.get_q().nrows() }
    /// (P^T L) Q (reference `src/qr.rs:73-77`)
    fn to_mat(&self) -> Array2<Self::A> {
        let ctx = Context::current();
        let l = device::upload(&ctx, self.get_l()).unwrap();
        let q = device::upload(&ctx, self.get_q()).unwrap();
        let ind = device::upload_indices(&ctx, self.get_ind()).unwrap();
        let out = DeviceMatrix::<Self::A>::zeros(&ctx, self.nrows(), self.ncols()).unwrap();
        ctx.check(unsafe { Self::A::ffi_lq_to_mat(ctx.raw(), l.view(), q.view(), ind.ptr as *const i64, out.view()) }).unwrap();
        out.to_array().unwrap()
    }
    /// reference `src/qr.rs:80-96`
    fn compress_lq_rank(&self, mut max_rank: usize) -> Result<LQ<Self::A>> {
        let (l, q) = (self.get_l(), self.get_q());
        if max_rank > q.nrows() {
            max_rank = q.nrows()
        }
        Ok(LQ { l: l.slice(s![.., 0..max_rank]).to_owned(), q: q.slice(s![0..max_rank, ..]).to_owned(), ind: self.get_ind().to_owned() })
    }
    /// reference `src/qr.rs:98-112`
    fn compress_lq_tolerance(&self, tol: f64) -> Result<LQ<Self::A>> {
        assert!((tol < 1.0) && (0.0 <= tol), "Require 0 <= tol < 1.0");
        let l = self.get_l();
        let pos = l.diag().iter().position(|&item| <Self::A as Scalar>::real_to_f64((item / l[[0, 0]]).abs()) < tol);
        match pos {
            Some(index) => self.compress_lq_rank(index),
            None => Err(RustyCompressionError::CompressionError),
        }
    }
    /// reference `src/qr.rs:114-119`
    fn compress(&self, compression_type: CompressionType) -> Result<LQ<Self::A>> {
        match compression_type {
            CompressionType::ADAPTIVE(tol) => self.compress_lq_tolerance(tol),
            CompressionType::RANK(rank) => self.compress_lq_rank(rank),
        }
    }
    fn get_q(&self) -> ArrayView2<Self::A>;
    fn get_l(&self) -> ArrayView2<Self::A>;
    fn get_ind(&self) -> ArrayView1<usize>;
    fn get_q_mut(&mut self) -> ArrayViewMut2<Self::A>;
    fn get_l_mut(&mut self) -> ArrayViewMut2<Self::A>;
    fn get_ind_mut(&mut self) -> ArrayViewMut1<usize>;
    fn compute_from(arr: ArrayView2<Self::A>) -> Result<LQ<Self::A>>;
    fn row_id(&self) -> Result<RowID<Self::A>>;
}

impl<T: Scalar> QRTraits for QR<T> {
    type A = T;
    fn get_q(&self) -> ArrayView2<T> { self.q.view() }
    fn get_r(&self) -> ArrayView2<T> { self.r.view() }
    fn get_ind(&self) -> ArrayView1<usize> { self.ind.view() }
    fn get_q_mut(&mut self) -> ArrayViewMut2<T> { self.q.view_mut() }
    fn get_r_mut(&mut self) -> ArrayViewMut2<T> { self.r.view_mut() }
    fn get_ind_mut(&mut self) -> ArrayViewMut1<usize> { self.ind.view_mut() }
    /// reference `src/qr.rs:251-253`
    fn compute_from(arr: ArrayView2<T>) -> Result<QR<T>> { T::pivoted_qr(arr) }
    /// reference `src/qr.rs:270-309`: C = Q R11, Z = [I | R11^-1 R12] P^T (one batched triangular solve on the device)
    fn column_id(&self) -> Result<ColumnID<T>> {
        let ctx = Context::current();
        let (m, k, n) = (self.q.nrows(), self.q.ncols(), self.r.ncols());
        let q = device::upload(&ctx, self.q.view())?;
        let r = device::upload(&ctx, self.r.view())?;
        let ind = device::upload_indices(&ctx, self.ind.view())?;
        let c = DeviceMatrix::<T>::zeros(&ctx, m, k)?;
        let z = DeviceMatrix::<T>::zeros(&ctx, k, n)?;
        ctx.check(unsafe { T::ffi_qr_column_id(ctx.raw(), q.view(), r.view(), ind.ptr as *const i64, c.view(), z.view()) })?;
        Ok(ColumnID { c: c.to_array()?, z: z.to_array()?, col_ind: self.ind.clone() })
    }
    /// reference `src/qr.rs:311-323`: B = (A^H range)^H, B P = Q_b R, Q = range Q_b.  The operator is only touched
    /// through `conj_matmat`, exactly as in the reference, so any `ConjMatMat` works (dense ones are one device GEMM).
    fn compute_from_range_estimate<Op: ConjMatMat<A = T>>(range: ArrayView2<T>, op: &Op) -> Result<QR<T>> {
        let b = op.conj_matmat(range).t().map(|item| item.conj());
        let qr = QR::<T>::compute_from(b.view())?;
        Ok(QR { q: crate::device::product::<T>(range, qr.q.view(), false)?, r: qr.r, ind: qr.ind })
    }
}

impl<T: Scalar> LQTraits for LQ<T> {
    type A = T;
    fn get_q(&self) -> ArrayView2<T> { self.q.view() }
    fn get_l(&self) -> ArrayView2<T> { self.l.view() }
    fn get_ind(&self) -> ArrayView1<usize> { self.ind.view() }
    fn get_q_mut(&mut self) -> ArrayViewMut2<T> { self.q.view_mut() }
    fn get_l_mut(&mut self) -> ArrayViewMut2<T> { self.l.view_mut() }
    fn get_ind_mut(&mut self) -> ArrayViewMut1<usize> { self.ind.view_mut() }
    /// reference `src/qr.rs:354-362`
    fn compute_from(arr: ArrayView2<T>) -> Result<LQ<T>> { T::pivoted_lq(arr) }
    /// reference `src/qr.rs:363-403`
    fn row_id(&self) -> Result<RowID<T>> {
        let ctx = Context::current();
        let (m, k, n) = (self.l.nrows(), self.q.nrows(), self.q.ncols());
        let l = device::upload(&ctx, self.l.view())?;
        let q = device::upload(&ctx, self.q.view())?;
        let ind = device::upload_indices(&ctx, self.ind.view())?;
        let x = DeviceMatrix::<T>::zeros(&ctx, m, k)?;
        let r = DeviceMatrix::<T>::zeros(&ctx, k, n)?;
        ctx.check(unsafe { T::ffi_lq_row_id(ctx.raw(), l.view(), q.view(), ind.ptr as *const i64, x.view(), r.view()) })?;
        Ok(RowID { x: x.to_array()?, r: r.to_array()?, row_ind: self.ind.clone() })
    }
}
