//! `col_interp_decomp` module of the reference (`src/col_interp_decomp.rs`).
use crate::device::{self, Context, DeviceMatrix, DeviceVec};
use crate::two_sided_interp_decomp::TwoSidedID;
use crate::types::{Apply, Result, Scalar};
use ndarray::{Array1, Array2, ArrayBase, ArrayView1, ArrayView2, ArrayViewMut1, ArrayViewMut2, Data, Ix1, Ix2};

/// A ~ C Z with C = A[:, col_ind[:k]] (reference `src/col_interp_decomp.rs:23-31`)
pub struct ColumnID<A: Scalar> {
    pub c: Array2<A>,
    pub z: Array2<A>,
    pub col_ind: Array1<usize>,
}

/// reference `src/col_interp_decomp.rs:44-86`
pub trait ColumnIDTraits {
    type A: Scalar;
    fn nrows(&self) -> usize { self.get_c().nrows() }
    fn ncols(&self) -> usize { self.get_z().ncols() }
    fn rank(&self) -> usize { self.get_c().ncols() }
    fn to_mat(&self) -> Array2<Self::A> { device::product::<Self::A>(self.get_c(), self.get_z(), false).unwrap() }
    fn get_c(&self) -> ArrayView2<Self::A>;
    fn get_z(&self) -> ArrayView2<Self::A>;
    fn get_col_ind(&self) -> ArrayView1<usize>;
    fn get_c_mut(&mut self) -> ArrayViewMut2<Self::A>;
    fn get_z_mut(&mut self) -> ArrayViewMut2<Self::A>;
    fn get_col_ind_mut(&mut self) -> ArrayViewMut1<usize>;
    fn new(c: Array2<Self::A>, z: Array2<Self::A>, col_ind: Array1<usize>) -> Self;
    fn two_sided_id(&self) -> Result<TwoSidedID<Self::A>>;
}

impl<T: Scalar> ColumnIDTraits for ColumnID<T> {
    type A = T;
    fn get_c(&self) -> ArrayView2<T> { self.c.view() }
    fn get_z(&self) -> ArrayView2<T> { self.z.view() }
    fn get_col_ind(&self) -> ArrayView1<usize> { self.col_ind.view() }
    fn get_c_mut(&mut self) -> ArrayViewMut2<T> { self.c.view_mut() }
    fn get_z_mut(&mut self) -> ArrayViewMut2<T> { self.z.view_mut() }
    fn get_col_ind_mut(&mut self) -> ArrayViewMut1<usize> { self.col_ind.view_mut() }
    fn new(c: Array2<T>, z: Array2<T>, col_ind: Array1<usize>) -> Self { ColumnID { c, z, col_ind } }
    /// reference `src/col_interp_decomp.rs:116-125`: row ID of C
    fn two_sided_id(&self) -> Result<TwoSidedID<T>> {
        let ctx = Context::current();
        let (m, k) = (self.c.nrows(), self.c.ncols());
        let c = device::upload(&ctx, self.c.view())?;
        let c_out = DeviceMatrix::<T>::zeros(&ctx, m, k)?;
        let x = DeviceMatrix::<T>::zeros(&ctx, k, k)?;
        let row_ind = DeviceVec::<i64>::new(&ctx, m)?;
        ctx.check(unsafe { T::ffi_column_id_two_sided(ctx.raw(), c.view(), c_out.view(), x.view(), row_ind.ptr as *mut i64) })?;
        Ok(TwoSidedID { c: c_out.to_array()?, x: x.to_array()?, r: self.z.clone(), row_ind: device::download_indices(&row_ind)?, col_ind: self.col_ind.clone() })
    }
}

/// `Apply` on vectors and matrices (reference `src/col_interp_decomp.rs:134-154`): C (Z x)
impl<T: Scalar, S: Data<Elem = T>> Apply<T, ArrayBase<S, Ix1>> for ColumnID<T> {
    type Output = Array1<T>;
    fn dot(&self, rhs: &ArrayBase<S, Ix1>) -> Array1<T> {
        let x = rhs.to_owned().into_shape((rhs.len(), 1)).unwrap();
        Apply::<T, Array2<T>>::dot(self, &x).into_shape(self.c.nrows()).unwrap()
    }
}
impl<T: Scalar, S: Data<Elem = T>> Apply<T, ArrayBase<S, Ix2>> for ColumnID<T> {
    type Output = Array2<T>;
    fn dot(&self, rhs: &ArrayBase<S, Ix2>) -> Array2<T> {
        let zx = device::product::<T>(self.z.view(), rhs.view(), false).unwrap();
        device::product::<T>(self.c.view(), zx.view(), false).unwrap()
    }
}
