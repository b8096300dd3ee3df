//! `two_sided_interp_decomp` module of the reference (`src/two_sided_interp_decomp.rs`).
use crate::device;
use crate::types::{Apply, Scalar};
use ndarray::{Array1, Array2, ArrayBase, ArrayView1, ArrayView2, ArrayViewMut1, ArrayViewMut2, Data, Ix1, Ix2};

/// A ~ C X R, X = A[row_ind[:k], col_ind[:k]] (reference `src/two_sided_interp_decomp.rs:19-30`)
pub struct TwoSidedID<A: Scalar> {
    pub c: Array2<A>,
    pub x: Array2<A>,
    pub r: Array2<A>,
    pub row_ind: Array1<usize>,
    pub col_ind: Array1<usize>,
}

/// reference `src/two_sided_interp_decomp.rs:43-96`
pub trait TwoSidedIDTraits {
    type A: Scalar;
    fn nrows(&self) -> usize { self.get_c().nrows() }
    fn ncols(&self) -> usize { self.get_r().ncols() }
    fn rank(&self) -> usize { self.get_c().ncols() }
    /// C (X R) (reference `:62-64`)
    fn to_mat(&self) -> Array2<Self::A> {
        let xr = device::product::<Self::A>(self.get_x(), self.get_r(), false).unwrap();
        device::product::<Self::A>(self.get_c(), xr.view(), false).unwrap()
    }
    fn get_c(&self) -> ArrayView2<Self::A>;
    fn get_x(&self) -> ArrayView2<Self::A>;
    fn get_r(&self) -> ArrayView2<Self::A>;
    fn get_col_ind(&self) -> ArrayView1<usize>;
    fn get_row_ind(&self) -> ArrayView1<usize>;
    fn get_c_mut(&mut self) -> ArrayViewMut2<Self::A>;
    fn get_x_mut(&mut self) -> ArrayViewMut2<Self::A>;
    fn get_r_mut(&mut self) -> ArrayViewMut2<Self::A>;
    fn get_col_ind_mut(&mut self) -> ArrayViewMut1<usize>;
    fn get_row_ind_mut(&mut self) -> ArrayViewMut1<usize>;
    /// note the argument order of the reference (`:89-95`): x, r, c, col_ind, row_ind
    fn new(x: Array2<Self::A>, r: Array2<Self::A>, c: Array2<Self::A>, col_ind: Array1<usize>, row_ind: Array1<usize>) -> Self;
}

impl<T: Scalar> TwoSidedIDTraits for TwoSidedID<T> {
    type A = T;
    fn get_c(&self) -> ArrayView2<T> { self.c.view() }
    fn get_x(&self) -> ArrayView2<T> { self.x.view() }
    fn get_r(&self) -> ArrayView2<T> { self.r.view() }
    fn get_col_ind(&self) -> ArrayView1<usize> { self.col_ind.view() }
    fn get_row_ind(&self) -> ArrayView1<usize> { self.row_ind.view() }
    fn get_c_mut(&mut self) -> ArrayViewMut2<T> { self.c.view_mut() }
    fn get_x_mut(&mut self) -> ArrayViewMut2<T> { self.x.view_mut() }
    fn get_r_mut(&mut self) -> ArrayViewMut2<T> { self.r.view_mut() }
    fn get_col_ind_mut(&mut self) -> ArrayViewMut1<usize> { self.col_ind.view_mut() }
    fn get_row_ind_mut(&mut self) -> ArrayViewMut1<usize> { self.row_ind.view_mut() }
    fn new(x: Array2<T>, r: Array2<T>, c: Array2<T>, col_ind: Array1<usize>, row_ind: Array1<usize>) -> Self { TwoSidedID { c, x, r, row_ind, col_ind } }
}

/// `Apply` (reference `:154-171`): C (X (R x))
impl<T: Scalar, S: Data<Elem = T>> Apply<T, ArrayBase<S, Ix1>> for TwoSidedID<T> {
    type Output = Array1<T>;
    fn dot(&self, rhs: &ArrayBase<S, Ix1>) -> Array1<T> {
        let x = rhs.to_owned().into_shape((rhs.len(), 1)).unwrap();
        Apply::<T, Array2<T>>::dot(self, &x).into_shape(self.c.nrows()).unwrap()
    }
}
impl<T: Scalar, S: Data<Elem = T>> Apply<T, ArrayBase<S, Ix2>> for TwoSidedID<T> {
    type Output = Array2<T>;
    fn dot(&self, rhs: &ArrayBase<S, Ix2>) -> Array2<T> {
        let rx = device::product::<T>(self.r.view(), rhs.view(), false).unwrap();
        let xrx = device::product::<T>(self.x.view(), rx.view(), false).unwrap();
        device::product::<T>(self.c.view(), xrx.view(), false).unwrap()
    }
}
