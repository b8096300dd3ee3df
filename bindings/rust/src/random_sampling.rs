//! `random_sampling` module of the reference (`src/random_sampling.rs`): `SampleRange`, `SampleRangePowerIteration`,
//! `MaxColNorm`, `AdaptiveSampling`.
//!
//! The generic implementations work for ANY operator behind `MatMat` / `ConjMatMat`, exactly as in the reference: Omega is
//! drawn on the host from the caller's generator (so a seeded run consumes the generator like the reference) and uploaded; for
//! `f32` / `f64` the whole algorithm is ONE call of the C ABI that calls back for the operator's products (`operator.rs`), for
//! the complex types a composition of one-call factorizations and device GEMMs.
//! A dense DEVICE-resident operator additionally has the fused entry points (`*_device` methods below), where Omega is the
//! on-device Philox stream and nothing returns to the host in between.
use crate::device::{self, Context, DeviceMatrix};
use crate::operator::{with_table, HostConjMatMat, HostMatMat, RealOpFfi};
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

/// How a scalar type runs the operator-generic algorithms.  `f32` / `f64`: ONE call of the C ABI per algorithm, the operator
/// behind the callback table (`operator.rs`; nothing is computed in this crate).  `c32` / `c64` (no callback entry points in
/// the C ABI yet): the reference's own composition, every product through the operator's traits and every factorization one call.
pub trait OpScalar: Scalar + RandomMatrix {
    fn op_sample_range_by_rank<Op: MatMat<A = Self>, R: Rng>(op: &Op, k: usize, p: usize, rng: &mut R) -> Result<Array2<Self>>;
    fn op_sample_range_power_iteration<Op: ConjMatMat<A = Self>, R: Rng>(op: &Op, k: usize, p: usize, it_count: usize, rng: &mut R) -> Result<Array2<Self>>;
    fn op_sample_range_adaptive<Op: ConjMatMat<A = Self>, R: Rng>(op: &Op, rel_tol: f64, sample_size: usize, rng: &mut R) -> Result<(Array2<Self>, Vec<(usize, f64)>)>;
}

macro_rules! impl_op_scalar_real {
    ($t:ty) => {
        impl OpScalar for $t {
            /// reference `src/random_sampling.rs:103-118`: Omega from the caller's generator, uploaded; `rc_sample_range_by_rank_op_*`
            fn op_sample_range_by_rank<Op: MatMat<A = $t>, R: Rng>(op: &Op, k: usize, p: usize, rng: &mut R) -> Result<Array2<$t>> {
                let ctx = Context::current();
                let (m, n) = (op.nrows(), op.ncols());
                let omega = device::upload(&ctx, <$t>::random_gaussian((n, k + p), rng).view())?;
                let q = DeviceMatrix::<$t>::zeros(&ctx, m, k.min(m).min(k + p))?;
                let host = HostMatMat(op);
                with_table(&ctx, &host, false, |tab| ctx.check(unsafe { <$t>::ffi_sample_range_by_rank_op(ctx.raw(), tab, k as i64, p as i64, omega.view(), 0, q.view()) }))?;
                q.to_array()
            }
            /// reference `src/random_sampling.rs:131-160` (one surviving power step); `rc_sample_range_power_iteration_op_*`
            fn op_sample_range_power_iteration<Op: ConjMatMat<A = $t>, R: Rng>(op: &Op, k: usize, p: usize, it_count: usize, rng: &mut R) -> Result<Array2<$t>> {
                let ctx = Context::current();
                let (m, n) = (op.nrows(), op.ncols());
                let omega = device::upload(&ctx, <$t>::random_gaussian((n, k + p), rng).view())?;
                let kk = if it_count == 0 { k.min(m).min(k + p) } else { k.min(m).min(n.min(m.min(k + p))) };
                let q = DeviceMatrix::<$t>::zeros(&ctx, m, kk)?;
                let host = HostConjMatMat(op);
                with_table(&ctx, &host, true, |tab| {
                    ctx.check(unsafe { <$t>::ffi_sample_range_power_iteration_op(ctx.raw(), tab, k as i64, p as i64, it_count as i64, omega.view(), 0, q.view()) })
                })?;
                q.to_array()
            }
            /// reference `src/random_sampling.rs:223-274`; `rc_sample_range_adaptive_op_*`.  The Omega blocks are drawn from the caller's
            /// generator UP FRONT for the largest rank the loop can reach (the reference draws one block per iteration: a seeded run sees
            /// the same blocks in the same order, the generator is left further advanced).
            fn op_sample_range_adaptive<Op: ConjMatMat<A = $t>, R: Rng>(op: &Op, rel_tol: f64, sample_size: usize, rng: &mut R) -> Result<(Array2<$t>, Vec<(usize, f64)>)> {
                let ctx = Context::current();
                let (m, n) = (op.nrows(), op.ncols());
                let s = sample_size.max(1);
                let cap = ((m.min(n) + s - 1) / s) * s;
                let blocks = cap / s.min(m).max(1) + 1;
                let omegas = device::upload(&ctx, <$t>::random_gaussian((n, s * blocks), rng).view())?;
                let q_cap = DeviceMatrix::<$t>::zeros(&ctx, m, cap)?;
                let hist_cap = blocks + 1;
                let (mut rank, mut hist_len) = (0i64, 0i64);
                let mut hr = vec![0i64; hist_cap];
                let mut he = vec![0f64; hist_cap];
                let host = HostConjMatMat(op);
                with_table(&ctx, &host, true, |tab| {
                    ctx.check(unsafe {
                        <$t>::ffi_sample_range_adaptive_op(ctx.raw(), tab, rel_tol, s as i64, omegas.view(), 0, q_cap.view(), &mut rank, hr.as_mut_ptr(), he.as_mut_ptr(),
                                                           hist_cap as i64, &mut hist_len)
                    })
                })?;
                let q = q_cap.to_array()?.slice(ndarray::s![.., 0..rank as usize]).to_owned();
                Ok((q, (0..hist_len as usize).map(|i| (hr[i] as usize, he[i])).collect()))
            }
        }
    };
}
impl_op_scalar_real!(f32);
impl_op_scalar_real!(f64);

#[cfg(feature = "complex")]
macro_rules! impl_op_scalar_complex {
    ($t:ty) => {
        impl OpScalar for $t {
            fn op_sample_range_by_rank<Op: MatMat<A = $t>, R: Rng>(op: &Op, k: usize, p: usize, rng: &mut R) -> Result<Array2<$t>> {
                let omega = <$t>::random_gaussian((op.ncols(), k + p), rng);
                let basis = op.matmat(omega.view());
                let qr = QR::<$t>::compute_from(basis.view())?.compress(CompressionType::RANK(k))?;
                Ok(qr.get_q().to_owned())
            }
            fn op_sample_range_power_iteration<Op: ConjMatMat<A = $t>, R: Rng>(op: &Op, k: usize, p: usize, it_count: usize, rng: &mut R) -> Result<Array2<$t>> {
                let omega = <$t>::random_gaussian((op.ncols(), k + p), rng);
                let op_omega = op.matmat(omega.view());
                let mut res = op_omega.clone();
                for index in 0..it_count {
                    let qr = QR::<$t>::compute_from(op_omega.view())?;
                    let qr = QR::<$t>::compute_from(op.conj_matmat(qr.get_q()).view())?;
                    let inner = op.matmat(qr.get_q());
                    if index == it_count - 1 {
                        res.assign(&inner);
                    }
                }
                Ok(QR::<$t>::compute_from(res.view())?.compress(CompressionType::RANK(k))?.get_q().to_owned())
            }
            fn op_sample_range_adaptive<Op: ConjMatMat<A = $t>, R: Rng>(op: &Op, rel_tol: f64, sample_size: usize, rng: &mut R) -> Result<(Array2<$t>, Vec<(usize, f64)>)> {
                // y <- y - q (q^H y) and y <- A Omega - q (b Omega) are device GEMMs with alpha = -1, beta = 1 (rc_gemm_*): no host arithmetic
                let tol_factor = 10.0 * std::f64::consts::FRAC_2_PI.sqrt();
                let m = op.ncols();
                let omega = <$t>::random_gaussian((m, sample_size), rng);
                let mut op_omega = op.matmat(omega.view());
                let operator_norm = <$t>::real_to_f64(op_omega.max_col_norm()) * tol_factor;
                let mut max_norm = operator_norm;
                let mut q = Array2::<$t>::zeros((op.nrows(), 0));
                let mut b = Array2::<$t>::zeros((0, op.ncols()));
                let mut residuals = Vec::<(usize, f64)>::new();
                while max_norm / operator_norm >= rel_tol {
                    if q.ncols() > 0 {
                        let t = device::product::<$t>(q.view(), op_omega.view(), true)?;
                        op_omega = device::gemm_update::<$t>(q.view(), t.view(), op_omega.view())?;   // op_omega - q t
                    }
                    let qr = QR::<$t>::compute_from(op_omega.view())?;
                    b = concatenate![Axis(0), b, op.conj_matmat(qr.get_q()).t().map(|item| item.conj())];
                    q = concatenate![Axis(1), q, qr.get_q()];
                    let omega = <$t>::random_gaussian((m, sample_size), rng);
                    let bo = device::product::<$t>(b.view(), omega.view(), false)?;
                    op_omega = device::gemm_update::<$t>(q.view(), bo.view(), op.matmat(omega.view()).view())?;
                    max_norm = <$t>::real_to_f64(op_omega.max_col_norm()) * tol_factor;
                    residuals.push((q.ncols(), max_norm / operator_norm));
                }
                Ok((q, residuals))
            }
        }
    };
}
#[cfg(feature = "complex")]
impl_op_scalar_complex!(crate::types::c32);
#[cfg(feature = "complex")]
impl_op_scalar_complex!(crate::types::c64);

/// reference `src/random_sampling.rs:102-121`: `impl<Op: MatMat<A = $scalar>> SampleRange for Op`
impl<A: OpScalar, Op: MatMat<A = A>> SampleRange<A> for Op {
    fn sample_range_by_rank<R: Rng>(&self, k: usize, p: usize, rng: &mut R) -> Result<Array2<A>> { A::op_sample_range_by_rank(self, k, p, rng) }
}
/// reference `src/random_sampling.rs:130-163`
impl<A: OpScalar, Op: ConjMatMat<A = A>> SampleRangePowerIteration<A> for Op {
    fn sample_range_power_iteration<R: Rng>(&self, k: usize, p: usize, it_count: usize, rng: &mut R) -> Result<Array2<A>> {
        A::op_sample_range_power_iteration(self, k, p, it_count, rng)
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
/// reference `src/random_sampling.rs:222-277`
impl<A: OpScalar, Op: ConjMatMat<A = A>> AdaptiveSampling<A> for Op {
    fn sample_range_adaptive<R: Rng>(&self, rel_tol: f64, sample_size: usize, rng: &mut R) -> Result<(Array2<A>, Vec<(usize, f64)>)> {
        A::op_sample_range_adaptive(self, rel_tol, sample_size, rng)
    }
}

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
