//! `permutation` module of the reference (`src/permutation.rs`): same enums, same function, same traits.
use crate::device::{self, Context, DeviceMatrix};
use crate::ffi::*;
use crate::types::Scalar;
use ndarray::{Array1, Array2, ArrayBase, Data, Ix1, Ix2};

/// reference `src/permutation.rs:7-16`
pub enum MatrixPermutationMode {
    COL,
    ROW,
    COLINV,
    ROWINV,
}
/// reference `src/permutation.rs:19-24`
pub enum VectorPermutationMode {
    INV,
    NOINV,
}

/// reference `src/permutation.rs:28-38` (index arithmetic on the host, as there)
pub fn invert_permutation_vector<S: Data<Elem = usize>>(perm: &ArrayBase<S, Ix1>) -> Array1<usize> {
    let mut inverse = Array1::<usize>::zeros(perm.len());
    for (index, &elem) in perm.iter().enumerate() {
        inverse[elem] = index;
    }
    inverse
}

/// reference `src/permutation.rs:40-57`
pub trait ApplyPermutationToMatrix {
    type A;
    fn apply_permutation(&self, index_array: ndarray::ArrayView1<usize>, mode: MatrixPermutationMode) -> Array2<Self::A>;
}
/// reference `src/permutation.rs:59-75`
pub trait ApplyPermutationToVector {
    type A;
    fn apply_permutation(&self, index_array: ndarray::ArrayView1<usize>, mode: VectorPermutationMode) -> Array1<Self::A>;
}

impl<A: Scalar, S: Data<Elem = A>> ApplyPermutationToMatrix for ArrayBase<S, Ix2> {
    type A = A;
    fn apply_permutation(&self, index_array: ndarray::ArrayView1<usize>, mode: MatrixPermutationMode) -> Array2<A> {
        let ctx = Context::current();
        let m = match mode {
            MatrixPermutationMode::COL => RC_PERM_COL,
            MatrixPermutationMode::ROW => RC_PERM_ROW,
            MatrixPermutationMode::COLINV => RC_PERM_COLINV,
            MatrixPermutationMode::ROWINV => RC_PERM_ROWINV,
        };
        let input = device::upload(&ctx, self.view()).unwrap();
        let out = DeviceMatrix::<A>::zeros(&ctx, self.nrows(), self.ncols()).unwrap();
        let perm = device::upload_indices(&ctx, index_array).unwrap();
        // a length mismatch is RC_INVALID_ARGUMENT -> panic, like the reference's assert (src/permutation.rs:96-99)
        ctx.check(unsafe { A::ffi_apply_permutation_matrix(ctx.raw(), m, input.view(), perm.ptr as *const i64, index_array.len() as i64, out.view()) })
            .unwrap();
        out.to_array().unwrap()
    }
}

impl<A: Scalar, S: Data<Elem = A>> ApplyPermutationToVector for ArrayBase<S, Ix1> {
    type A = A;
    fn apply_permutation(&self, index_array: ndarray::ArrayView1<usize>, mode: VectorPermutationMode) -> Array1<A> {
        // vectors are n x 1 matrices: NOINV = ROW, INV = ROWINV (reference src/permutation.rs:153-183)
        let n = self.len();
        let col = self.to_owned().into_shape((n, 1)).unwrap();
        let mode = match mode {
            VectorPermutationMode::NOINV => MatrixPermutationMode::ROW,
            VectorPermutationMode::INV => MatrixPermutationMode::ROWINV,
        };
        col.apply_permutation(index_array, mode).into_shape(n).unwrap()
    }
}
