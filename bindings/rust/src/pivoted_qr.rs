//! `PivotedQR` (reference `src/pivoted_qr.rs:11-41`): ?geqp3 + ?orgqr / ?ungqr on the device.
use crate::device::{self, Context, DeviceMatrix, DeviceVec};
use crate::qr::{LQ, QR};
use crate::types::{Result, Scalar};
use ndarray::{ArrayBase, Data, Ix2};

pub(crate) trait PivotedQR: Scalar {
    /// A P = Q R; `ind[j]` = column of A at position j (reference `src/pivoted_qr.rs:25-31`, `:81-119`)
    fn pivoted_qr<S: Data<Elem = Self>>(arr: ArrayBase<S, Ix2>) -> Result<QR<Self>> {
        let ctx = Context::current();
        let (m, n) = (arr.nrows(), arr.ncols());
        let k = m.min(n);
        let a = device::upload(&ctx, arr.view())?;
        let q = DeviceMatrix::<Self>::zeros(&ctx, m, k)?;
        let r = DeviceMatrix::<Self>::zeros(&ctx, k, n)?;
        let ind = DeviceVec::<i64>::new(&ctx, n)?;
        ctx.check(unsafe { Self::ffi_pivoted_qr(ctx.raw(), a.view(), q.view(), r.view(), ind.ptr as *mut i64) })?;
        Ok(QR { q: q.to_array()?, r: r.to_array()?, ind: device::download_indices(&ind)? })
    }
    /// P A = L Q (reference `src/pivoted_qr.rs:32-41`)
    fn pivoted_lq<S: Data<Elem = Self>>(arr: ArrayBase<S, Ix2>) -> Result<LQ<Self>> {
        let ctx = Context::current();
        let (m, n) = (arr.nrows(), arr.ncols());
        let k = m.min(n);
        let a = device::upload(&ctx, arr.view())?;
        let l = DeviceMatrix::<Self>::zeros(&ctx, m, k)?;
        let q = DeviceMatrix::<Self>::zeros(&ctx, k, n)?;
        let ind = DeviceVec::<i64>::new(&ctx, m)?;
        ctx.check(unsafe { Self::ffi_pivoted_lq(ctx.raw(), a.view(), l.view(), q.view(), ind.ptr as *mut i64) })?;
        Ok(LQ { l: l.to_array()?, q: q.to_array()?, ind: device::download_indices(&ind)? })
    }
    /// The same factorization through the LAPACK seam only: `rc_geqp3` where the reference calls `$qrf` (`src/pivoted_qr.rs:139-172`),
    /// `rc_orgqr` where it calls `lax::Lapack::q` (`:104-108`); R is the upper triangle of the first k rows (`:100-102`).
    fn pivoted_qr_at_the_lapack_seam<S: Data<Elem = Self>>(arr: ArrayBase<S, Ix2>) -> Result<QR<Self>> {
        let ctx = Context::current();
        let (m, n) = (arr.nrows(), arr.ncols());
        let k = m.min(n);
        let a = device::upload(&ctx, arr.view())?;
        let jpvt = DeviceVec::<i64>::new(&ctx, n)?;
        let tau = DeviceVec::<Self>::new(&ctx, k)?;
        ctx.check(unsafe { Self::ffi_geqp3(ctx.raw(), a.view(), k as i64, jpvt.ptr as *mut i64, tau.ptr) })?;
        let q = DeviceMatrix::<Self>::zeros(&ctx, m, k)?;
        ctx.check(unsafe { Self::ffi_orgqr(ctx.raw(), a.view(), tau.ptr, k as i64, q.view()) })?;
        let f = a.to_array()?;
        let r = ndarray::Array2::from_shape_fn((k, n), |(i, j)| if i <= j { f[[i, j]] } else { Self::default() });
        Ok(QR { q: q.to_array()?, r, ind: device::download_indices(&jpvt)? })
    }
}
impl<T: Scalar> PivotedQR for T {}

#[cfg(test)]
mod tests {
    use super::*;
    use crate::random_matrix::RandomMatrix;
    use crate::types::RelDiff;
    use rand::SeedableRng;

    /// the two routes -- one fused call, and `rc_geqp3` + `rc_orgqr` at the LAPACK seam -- give the same factorization
    #[test]
    fn test_pivoted_qr_at_the_lapack_seam_matches_the_fused_call() {
        let mut rng = rand::rngs::StdRng::seed_from_u64(7);
        let mat = f64::random_approximate_low_rank_matrix((120, 80), 1.0, 1E-8, &mut rng);
        let fused = f64::pivoted_qr(mat.view()).unwrap();
        let seam = f64::pivoted_qr_at_the_lapack_seam(mat.view()).unwrap();
        assert_eq!(fused.ind, seam.ind);
        assert!(f64::rel_diff_fro(seam.r.view(), fused.r.view()) < 1E-12);
        assert!(f64::rel_diff_fro(seam.q.view(), fused.q.view()) < 1E-10);
    }
}
