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
}
impl<T: Scalar> PivotedQR for T {}
