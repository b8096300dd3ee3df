//! `svd` module of the reference (`src/svd.rs`): `SVD`, `SVDTraits`.
use crate::compute_svd::ComputeSVD;
use crate::device::{self, Context, DeviceMatrix, DeviceVec};
use crate::qr::QR;
use crate::types::{ConjMatMat, Result, RustyCompressionError, Scalar};
use crate::CompressionType;
use ndarray::{s, Array1, Array2, ArrayView1, ArrayView2, ArrayViewMut1, ArrayViewMut2};

/// reference `src/svd.rs:13-20`
pub struct SVD<A: Scalar> {
    pub u: Array2<A>,
    pub s: Array1<A::Real>,
    pub vt: Array2<A>,
}

/// reference `src/svd.rs:23-122`
pub trait SVDTraits {
    type A: Scalar;
    fn nrows(&self) -> usize { self.get_u().nrows() }
    fn ncols(&self) -> usize { self.get_vt().ncols() }
    fn rank(&self) -> usize { self.get_u().ncols() }
    /// U diag(S) V^T (reference `src/svd.rs:42-54`)
    fn to_mat(&self) -> Array2<Self::A> {
        let ctx = Context::current();
        let u = device::upload(&ctx, self.get_u()).unwrap();
        let vt = device::upload(&ctx, self.get_vt()).unwrap();
        let s = DeviceVec::<<Self::A as Scalar>::Real>::from_slice(&ctx, self.get_s().to_vec().as_slice()).unwrap();
        let out = DeviceMatrix::<Self::A>::zeros(&ctx, self.nrows(), self.ncols()).unwrap();
        ctx.check(unsafe { Self::A::ffi_svd_to_mat(ctx.raw(), u.view(), s.ptr as *const <Self::A as Scalar>::Real, vt.view(), out.view()) }).unwrap();
        out.to_array().unwrap()
    }
    fn to_qr(self) -> Result<QR<Self::A>>;
    /// reference `src/svd.rs:60-65`
    fn compress(&self, compression_type: CompressionType) -> Result<SVD<Self::A>> {
        match compression_type {
            CompressionType::ADAPTIVE(tol) => self.compress_svd_tolerance(tol),
            CompressionType::RANK(rank) => self.compress_svd_rank(rank),
        }
    }
    /// reference `src/svd.rs:68-84`
    fn compress_svd_rank(&self, mut max_rank: usize) -> Result<SVD<Self::A>> {
        let (u, sv, vt) = (self.get_u(), self.get_s(), self.get_vt());
        if max_rank > sv.len() {
            max_rank = sv.len()
        }
        Ok(SVD { u: u.slice(s![.., 0..max_rank]).to_owned(), s: sv.slice(s![0..max_rank]).to_owned(), vt: vt.slice(s![0..max_rank, ..]).to_owned() })
    }
    /// reference `src/svd.rs:87-101`: first i with s_i / s_0 < tol, `CompressionError` if none
    fn compress_svd_tolerance(&self, tol: f64) -> Result<SVD<Self::A>> {
        assert!((tol < 1.0) && (0.0 <= tol), "Require 0 <= tol < 1.0");
        let first_val = self.get_s()[0];
        let pos = self.get_s().iter().position(|&item| <Self::A as Scalar>::real_to_f64(item / first_val) < tol);
        match pos {
            Some(index) => self.compress_svd_rank(index),
            None => Err(RustyCompressionError::CompressionError),
        }
    }
    fn compute_from(arr: ArrayView2<Self::A>) -> Result<SVD<Self::A>>;
    fn compute_from_range_estimate<Op: ConjMatMat<A = Self::A>>(range: ArrayView2<Self::A>, op: &Op) -> Result<SVD<Self::A>>;
    fn get_u(&self) -> ArrayView2<Self::A>;
    fn get_s(&self) -> ArrayView1<<Self::A as Scalar>::Real>;
    fn get_vt(&self) -> ArrayView2<Self::A>;
    fn get_u_mut(&mut self) -> ArrayViewMut2<Self::A>;
    fn get_s_mut(&mut self) -> ArrayViewMut1<<Self::A as Scalar>::Real>;
    fn get_vt_mut(&mut self) -> ArrayViewMut2<Self::A>;
}

impl<T: Scalar> SVDTraits for SVD<T> {
    type A = T;
    fn get_u(&self) -> ArrayView2<T> { self.u.view() }
    fn get_s(&self) -> ArrayView1<T::Real> { self.s.view() }
    fn get_vt(&self) -> ArrayView2<T> { self.vt.view() }
    fn get_u_mut(&mut self) -> ArrayViewMut2<T> { self.u.view_mut() }
    fn get_s_mut(&mut self) -> ArrayViewMut1<T::Real> { self.s.view_mut() }
    fn get_vt_mut(&mut self) -> ArrayViewMut2<T> { self.vt.view_mut() }
    /// reference `src/svd.rs:150-163`: pivoted QR of diag(S) V^T, Q = U Q_b
    fn to_qr(self) -> Result<QR<T>> {
        let ctx = Context::current();
        let (m, r, n) = (self.u.nrows(), self.s.len(), self.vt.ncols());
        let k = r.min(n);
        let u = device::upload(&ctx, self.u.view())?;
        let vt = device::upload(&ctx, self.vt.view())?;
        let s = DeviceVec::<T::Real>::from_slice(&ctx, self.s.to_vec().as_slice())?;
        let q = DeviceMatrix::<T>::zeros(&ctx, m, k)?;
        let rr = DeviceMatrix::<T>::zeros(&ctx, k, n)?;
        let ind = DeviceVec::<i64>::new(&ctx, n)?;
        ctx.check(unsafe { T::ffi_svd_to_qr(ctx.raw(), u.view(), s.ptr as *const T::Real, vt.view(), q.view(), rr.view(), ind.ptr as *mut i64) })?;
        Ok(QR { q: q.to_array()?, r: rr.to_array()?, ind: device::download_indices(&ind)? })
    }
    /// reference `src/svd.rs:165-169`
    fn compute_from(arr: ArrayView2<T>) -> Result<SVD<T>> { T::compute_svd(arr) }
    /// reference `src/svd.rs:171-183`
    fn compute_from_range_estimate<Op: ConjMatMat<A = T>>(range: ArrayView2<T>, op: &Op) -> Result<SVD<T>> {
        let b = op.conj_matmat(range).t().map(|item| item.conj());
        let svd = SVD::<T>::compute_from(b.view())?;
        Ok(SVD { u: crate::device::product::<T>(range, svd.u.view(), false)?, s: svd.s, vt: svd.vt })
    }
}
