//! `random_sampling` module of the reference (`src/random_sampling.rs`): `SampleRange`, `SampleRangePowerIteration`,
//! `MaxColNorm`, `AdaptiveSampling`.
//!
//! The generic implementations work for ANY operator behind `MatMat` / `ConjMatMat`, exactly as in the reference: Omega is
//! drawn on the host from the caller's generator (so a seeded run consumes the generator like the reference), every product
//! goes through the operator's traits (a dense matrix: one device GEMM), every factorization is one call of the C ABI.
//! A dense DEVICE-resident operator additionally has the fused entry points (`*_device` methods below), where Omega is the
//! on-device Philox stream and nothing returns to the host in between.
use crate::device::{self, Context, DeviceMatrix};
use crate::qr::{QRTraits, QR};
use crate::random_matrix::RandomMatrix;
use crate::types::{ConjMatMat, MatMat, Result, Scalar};
use crate::CompressionType;
use ndarray::{concatenate, Array2, ArrayBase, Axis, Data, Ix2};
use rand::Rng;

/// reference `src/random_sampling.rs:58-72`
pub trait SampleRange<A: Scalar> {
    fn sample_range_by_rank<R: Rng>(&self, k: usize, p: usize, rng: &mut R) -> Result<Array2<A>>;
}
/// reference `src/random_sampling.rs:82-98`
pub trait SampleRangePowerIteration<A: Scalar> {
    fn sample_range_power_iteration<R: Rng>(&self, k: usize, p: usize, it_count: usize, rng: &mut R) -> Result<Array2<A>>;
}
/// reference `src/random_sampling.rs:175-178`
pub trait MaxColNorm<A: Scalar> {
    fn max_col_norm(&self) -> A::Real;
}
/// reference `src/random_sampling.rs:202-218`
pub trait AdaptiveSampling<A: Scalar> {
    fn sample_range_adaptive<R: Rng>(&self, rel_tol: f64, sample_size: usize, rng: &mut R) -> Result<(Array2<A>, Vec<(usize, f64)>)>;
}

/// reference `src/random_sampling.rs:103-118`
impl<A: Scalar + RandomMatrix, Op: MatMat<A = A>> SampleRange<A> for Op {
    fn sample_range_by_rank<R: Rng>(&self, k: usize, p: usize, rng: &mut R) -> Result<Array2<A>> {
        let m = self.ncols();
        let omega = A::random_gaussian((m, k + p), rng);
        let basis = self.matmat(omega.view());
        let qr = QR::<A>::compute_from(basis.view())?.compress(CompressionType::RANK(k))?;
        Ok(qr.get_q().to_owned())
    }
}

/// reference `src/random_sampling.rs:131-160`, INCLUDING its behaviour for `it_count >= 1`: the inner product shadows the
/// outer one, so every iteration restarts from `A Omega` and only the last one is kept.
impl<A: Scalar + RandomMatrix, Op: ConjMatMat<A = A>> SampleRangePowerIteration<A> for Op {
    fn sample_range_power_iteration<R: Rng>(&self, k: usize, p: usize, it_count: usize, rng: &mut R) -> Result<Array2<A>> {
        let m = self.ncols();
        let omega = A::random_gaussian((m, k + p), rng);
        let op_omega = self.matmat(omega.view());
        let mut res = op_omega.clone();
        for index in 0..it_count {
            let qr = QR::<A>::compute_from(op_omega.view())?;
            let qr = QR::<A>::compute_from(self.conj_matmat(qr.get_q()).view())?;
            let inner = self.matmat(qr.get_q());
            if index == it_count - 1 {
                res.assign(&inner);
            }
        }
        let compressed = QR::<A>::compute_from(res.view())?.compress(CompressionType::RANK(k))?;
        Ok(compressed.get_q().to_owned())
    }
}

/// reference `src/random_sampling.rs:184-191`
impl<A: Scalar, S: Data<Elem = A>> MaxColNorm<A> for ArrayBase<S, Ix2> {
    fn max_col_norm(&self) -> A::Real {
        let ctx = Context::current();
        let y = device::upload(&ctx, self.view()).unwrap();
        let mut out = A::Real::default();
        ctx.check(unsafe { A::ffi_max_col_norm(ctx.raw(), y.view(), &mut out as *mut A::Real) }).unwrap();
        out
    }
}

/// reference `src/random_sampling.rs:223-274` (HMT section 4.3 estimator, factor 10 sqrt(2 / pi))
impl<A: Scalar + RandomMatrix, Op: ConjMatMat<A = A>> AdaptiveSampling<A> for Op {
    fn sample_range_adaptive<R: Rng>(&self, rel_tol: f64, sample_size: usize, rng: &mut R) -> Result<(Array2<A>, Vec<(usize, f64)>)> {
        let tol_factor = num_traits::cast::<f64, A::Real>(10.0 * std::f64::consts::FRAC_2_PI.sqrt()).unwrap();
        let m = self.ncols();
        let rel_tol = num_traits::cast::<f64, A::Real>(rel_tol).unwrap();
        let omega = A::random_gaussian((m, sample_size), rng);
        let mut op_omega = self.matmat(omega.view());
        let operator_norm = op_omega.max_col_norm() * tol_factor;
        let mut max_norm = operator_norm;
        let mut q = Array2::<A>::zeros((self.nrows(), 0));
        let mut b = Array2::<A>::zeros((0, self.ncols()));
        let mut residuals = Vec::<(usize, f64)>::new();
        while max_norm / operator_norm >= rel_tol {
            if q.ncols() > 0 {
                // op_omega -= q (q^H op_omega): two device GEMMs
                let t = device::product::<A>(q.view(), op_omega.view(), true)?;
                let corr = device::product::<A>(q.view(), t.view(), false)?;
                op_omega.zip_mut_with(&corr, |x, &c| *x = sub(*x, c));
            }
            let qr = QR::<A>::compute_from(op_omega.view())?;
            b = concatenate![Axis(0), b, self.conj_matmat(qr.get_q()).t().map(|item| item.conj())];
            q = concatenate![Axis(1), q, qr.get_q()];
            let omega = A::random_gaussian((m, sample_size), rng);
            let bo = device::product::<A>(b.view(), omega.view(), false)?;
            let qbo = device::product::<A>(q.view(), bo.view(), false)?;
            op_omega = self.matmat(omega.view());
            op_omega.zip_mut_with(&qbo, |x, &c| *x = sub(*x, c));
            max_norm = op_omega.max_col_norm() * tol_factor;
            residuals.push((q.ncols(), A::real_to_f64(max_norm / operator_norm)));
        }
        Ok((q, residuals))
    }
}

fn sub<A: Scalar>(x: A, c: A) -> A { x - c }

impl<A: Scalar> DeviceMatrix<A> {
    /// Fused `sample_range_by_rank` on a device-resident operator (rc_sample_range_by_rank_*): Omega = Philox stream `seed`.
    pub fn sample_range_by_rank_device(&self, k: usize, p: usize, seed: u64) -> Result<DeviceMatrix<A>> {
        let (m, _n) = self.dims();
        let kk = k.min(m).min(k + p);
        let q = DeviceMatrix::<A>::zeros(self.ctx(), m, kk)?;
        self.ctx().check(unsafe { A::ffi_sample_range_by_rank(self.ctx().raw(), self.view(), k as i64, p as i64, DeviceMatrix::<A>::null(), seed, q.view()) })?;
        Ok(q)
    }
    /// Fused power iteration (rc_sample_range_power_iteration_*; the reference's single surviving step unless
    /// RC_OPT_POWER_ITERATION_FIXED is set on the context).
    pub fn sample_range_power_iteration_device(&self, k: usize, p: usize, it_count: usize, seed: u64) -> Result<DeviceMatrix<A>> {
        let (m, n) = self.dims();
        let kk = k.min(m).min(n.min(m.min(k + p)));
        let q = DeviceMatrix::<A>::zeros(self.ctx(), m, kk)?;
        self.ctx().check(unsafe {
            A::ffi_sample_range_power_iteration(self.ctx().raw(), self.view(), k as i64, p as i64, it_count as i64, DeviceMatrix::<A>::null(), seed, q.view())
        })?;
        Ok(q)
    }
    /// Fused adaptive range finder (rc_sample_range_adaptive_*): returns the basis and the residual history.
    pub fn sample_range_adaptive_device(&self, rel_tol: f64, sample_size: usize, seed: u64) -> Result<(Array2<A>, Vec<(usize, f64)>)> {
        let (m, n) = self.dims();
        let cap = m.min(n);
        let q_cap = DeviceMatrix::<A>::zeros(self.ctx(), m, cap)?;
        let hist_cap = cap / sample_size.max(1) + 2;
        let (mut rank, mut hist_len) = (0i64, 0i64);
        let mut hr = vec![0i64; hist_cap];
        let mut he = vec![0f64; hist_cap];
        self.ctx().check(unsafe {
            A::ffi_sample_range_adaptive(self.ctx().raw(), self.view(), rel_tol, sample_size as i64, DeviceMatrix::<A>::null(), seed, q_cap.view(), &mut rank, hr.as_mut_ptr(),
                                         he.as_mut_ptr(), hist_cap as i64, &mut hist_len)
        })?;
        let full = q_cap.to_array()?;
        let q = full.slice(ndarray::s![.., 0..rank as usize]).to_owned();
        let hist = (0..hist_len as usize).map(|i| (hr[i] as usize, he[i])).collect();
        Ok((q, hist))
    }
    /// Rank-k column ID of a device-resident matrix through the truncated factorization (rc_column_id_rank_*):
    /// the unit of work of batches of independent matrices.
    pub fn column_id_rank(&self, k: usize) -> Result<crate::col_interp_decomp::ColumnID<A>> {
        let (m, n) = self.dims();
        let k = k.min(m).min(n);
        let c = DeviceMatrix::<A>::zeros(self.ctx(), m, k)?;
        let z = DeviceMatrix::<A>::zeros(self.ctx(), k, n)?;
        let ind = device::DeviceVec::<i64>::new(self.ctx(), n)?;
        self.ctx().check(unsafe { A::ffi_column_id_rank(self.ctx().raw(), self.view(), k as i64, c.view(), z.view(), ind.ptr as *mut i64) })?;
        Ok(crate::col_interp_decomp::ColumnID { c: c.to_array()?, z: z.to_array()?, col_ind: device::download_indices(&ind)? })
    }
}
