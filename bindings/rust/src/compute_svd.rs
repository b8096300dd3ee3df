//! `ComputeSVD` (reference `src/compute_svd.rs:8-30`): thin ?gesdd-equivalent on the device.
use crate::device::{self, Context, DeviceMatrix, DeviceVec};
use crate::svd::SVD;
use crate::types::{Result, Scalar};
use ndarray::{Array1, ArrayView2};

pub trait ComputeSVD: Scalar {
    fn compute_svd(arr: ArrayView2<Self>) -> Result<SVD<Self>> {
        let ctx = Context::current();
        let (m, n) = (arr.nrows(), arr.ncols());
        let r = m.min(n);
        let a = device::upload(&ctx, arr)?;
        let u = DeviceMatrix::<Self>::zeros(&ctx, m, r)?;
        let vt = DeviceMatrix::<Self>::zeros(&ctx, r, n)?;
        let s = DeviceVec::<Self::Real>::new(&ctx, r)?;
        ctx.check(unsafe { Self::ffi_compute_svd(ctx.raw(), a.view(), u.view(), s.ptr as *mut Self::Real, vt.view()) })?;
        Ok(SVD { u: u.to_array()?, s: Array1::from(s.to_vec()?), vt: vt.to_array()? })
    }
}
impl<T: Scalar> ComputeSVD for T {}
