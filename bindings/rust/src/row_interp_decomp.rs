//! `row_interp_decomp` module of the reference (`src/row_interp_decomp.rs`).
use crate::device::{self, Context, DeviceMatrix, DeviceVec};
use crate::two_sided_interp_decomp::TwoSidedID;
use crate::types::{Apply, Result, Scalar};
use ndarray::{Array1, Array2, ArrayBase, ArrayView1, ArrayView2, ArrayViewMut1, ArrayViewMut2, Data, Ix1, Ix2};

/// A ~ X R with R = A[row_ind[:k], :] (reference `src/row_interp_decomp.rs:25-33`)
pub struct RowID<A: Scalar> {
    pub x: Array2<A>,
    pub r: Array2<A>,
    pub row_ind: Array1<usize>,
}

/// reference `src/row_interp_decomp.rs:46-89`
pub trait RowIDTraits {
    type A: Scalar;
    fn nrows(&self) -> usize { self.get_x().nrows() }
    fn ncols(&self) -> usize { self.get_r().ncols() }
    fn rank(&self) -> usize { self.get_r().nrows() }
    fn to_mat(&self) -> Array2<Self::A> { device::product::<Self::A>(self.get_x(), self.get_r(), false).unwrap() }
    fn get_x(&self) -> ArrayView2<Self::A>;
    fn get_r(&self) -> ArrayView2<Self::A>;
    fn get_row_ind(&self) -> ArrayView1<usize>;
    fn get_x_mut(&mut self) -> ArrayViewMut2<Self::A>;
    fn get_r_mut(&mut self) -> ArrayViewMut2<Self::A>;
    fn get_row_ind_mut(&mut self) -> ArrayViewMut1<usize>;
    fn new(x: Array2<Self::A>, r: Array2<Self::A>, row_ind: Array1<usize>) -> Self;
    fn two_sided_id(&self) -> Result<TwoSidedID<Self::A>>;
}

impl<T: Scalar> RowIDTraits for RowID<T> {
    type A = T;
    fn get_x(&self) -> ArrayView2<T> { self.x.view() }
    fn get_r(&self) -> ArrayView2<T> { self.r.view() }
    fn get_row_ind(&self) -> ArrayView1<usize> { self.row_ind.view() }
    fn get_x_mut(&mut self) -> ArrayViewMut2<T> { self.x.view_mut() }
    fn get_r_mut(&mut self) -> ArrayViewMut2<T> { self.r.view_mut() }
    fn get_row_ind_mut(&mut self) -> ArrayViewMut1<usize> { self.row_ind.view_mut() }
    fn new(x: Array2<T>, r: Array2<T>, row_ind: Array1<usize>) -> Self { RowID { x, r, row_ind } }
    /// reference `src/row_interp_decomp.rs:120-130`: column ID of R
    fn two_sided_id(&self) -> Result<TwoSidedID<T>> {
        let ctx = Context::current();
        let (k, n) = (self.r.nrows(), self.r.ncols());
        let r = device::upload(&ctx, self.r.view())?;
        let x = DeviceMatrix::<T>::zeros(&ctx, k, k)?;
        let r_out = DeviceMatrix::<T>::zeros(&ctx, k, n)?;
        let col_ind = DeviceVec::<i64>::new(&ctx, n)?;
        ctx.check(unsafe { T::ffi_row_id_two_sided(ctx.raw(), r.view(), x.view(), r_out.view(), col_ind.ptr as *mut i64) })?;
        Ok(TwoSidedID { c: self.x.clone(), x: x.to_array()?, r: r_out.to_array()?, row_ind: self.row_ind.clone(), col_ind: device::download_indices(&col_ind)? })
    }
}

/// `Apply` (reference `src/row_interp_decomp.rs:134-154`): X (R x)
impl<T: Scalar, S: Data<Elem = T>> Apply<T, ArrayBase<S, Ix1>> for RowID<T> {
    type Output = Array1<T>;
    fn dot(&self, rhs: &ArrayBase<S, Ix1>) -> Array1<T> {
        let x = rhs.to_owned().into_shape((rhs.len(), 1)).unwrap();
        Apply::<T, Array2<T>>::dot(self, &x).into_shape(self.x.nrows()).unwrap()
    }
}
impl<T: Scalar, S: Data<Elem = T>> Apply<T, ArrayBase<S, Ix2>> for RowID<T> {
    type Output = Array2<T>;
    fn dot(&self, rhs: &ArrayBase<S, Ix2>) -> Array2<T> {
        let rx = device::product::<T>(self.r.view(), rhs.view(), false).unwrap();
        device::product::<T>(self.x.view(), rx.view(), false).unwrap()
    }
}
