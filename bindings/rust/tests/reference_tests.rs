//! Ports of the reference crate's 89 unit tests (rusty-compression v0.1.1) to this crate's surface:
//!   src/pivoted_qr.rs:193-317 (16), src/qr.rs:418-616 (31), src/svd.rs:193-321 (24),
//!   src/col_interp_decomp.rs:163-242 (8), src/row_interp_decomp.rs:163-236 (8), src/permutation.rs:187-240 (2).
//! Same names, same matrices (the reference's generator recipe), same assertions and tolerances; the only change is a
//! SEEDED generator instead of `rand::thread_rng()`, so a failure is reproducible.  They need an MI355X at run time
//! (`cargo test` on the GPU box); the compiled twin that runs in this repository's GPU suite is tests/cpp/reference_tests.cpp.
use ndarray::Axis;
use rand::rngs::StdRng;
use rand::SeedableRng;
use rusty_compression_amd::permutation::*;
use rusty_compression_amd::types::{c32, c64, RelDiff, Scalar};
use rusty_compression_amd::*;

fn rng(tag: &str) -> StdRng {
    // one stream per test, derived from its name
    let mut h: u64 = 0xcbf29ce484222325;
    for b in tag.bytes() {
        h = (h ^ b as u64).wrapping_mul(0x100000001b3);
    }
    StdRng::seed_from_u64(h)
}
fn f64_of<A: Scalar>(r: A::Real) -> f64 { A::real_to_f64(r) }

macro_rules! pivoted_qr_tests {
    ($($name:ident: $scalar:ty, $dim:expr,)*) => { $(
        #[test]
        fn $name() {
            let (m, n) = $dim;
            let mut rng = rng(stringify!($name));
            let mat = <$scalar>::random_approximate_low_rank_matrix((m, n), 1.0, 1E-5, &mut rng);
            let qr_result = QR::<$scalar>::compute_from(mat.view()).unwrap();
            let prod = qr_result.q.dot(&qr_result.r);
            let qtq = qr_result.q.t().map(|&item| item.conj()).dot(&qr_result.q);
            for ((i, j), &val) in qtq.indexed_iter() {
                if i == j { assert!(f64_of::<$scalar>((val - <$scalar as num_traits::One>::one()).abs()) < 1E-6); }
                else { assert!(f64_of::<$scalar>(val.abs()) < 1E-6); }
            }
            for (col_index, col) in prod.axis_iter(Axis(1)).enumerate() {
                let perm_index = qr_result.ind[col_index];
                let rel_diff = <$scalar>::rel_diff_l2(col, mat.index_axis(Axis(1), perm_index));
                assert!(f64_of::<$scalar>(rel_diff) < 1E-6);
            }
        }
    )* };
}

macro_rules! pivoted_lq_tests {
    ($($name:ident: $scalar:ty, $dim:expr,)*) => { $(
        #[test]
        fn $name() {
            let (m, n) = $dim;
            let mut rng = rng(stringify!($name));
            let mat = <$scalar>::random_approximate_low_rank_matrix((m, n), 1.0, 1E-5, &mut rng);
            let lq_result = LQ::<$scalar>::compute_from(mat.view()).unwrap();
            let prod = lq_result.l.dot(&lq_result.q);
            let qqt = lq_result.q.dot(&lq_result.q.t().map(|&item| item.conj()));
            for ((i, j), &val) in qqt.indexed_iter() {
                if i == j { assert!(f64_of::<$scalar>((val - <$scalar as num_traits::One>::one()).abs()) < 1E-6); }
                else { assert!(f64_of::<$scalar>(val.abs()) < 1E-6); }
            }
            for (row_index, row) in prod.axis_iter(Axis(0)).enumerate() {
                let perm_index = lq_result.ind[row_index];
                let rel_diff = <$scalar>::rel_diff_l2(row, mat.index_axis(Axis(0), perm_index));
                assert!(f64_of::<$scalar>(rel_diff) < 1E-6);
            }
        }
    )* };
}

macro_rules! qr_compression_by_rank_tests {
    ($($name:ident: $scalar:ty, $dim:expr, $tol:expr,)*) => { $(
        #[test]
        fn $name() {
            let (m, n) = $dim;
            let rank: usize = 30;
            let mut rng = rng(stringify!($name));
            let mat = <$scalar>::random_approximate_low_rank_matrix((m, n), 1.0, 1E-10, &mut rng);
            let qr = QR::<$scalar>::compute_from(mat.view()).unwrap().compress(CompressionType::RANK(rank)).unwrap();
            assert!(qr.q.len_of(Axis(1)) == rank);
            assert!(qr.r.len_of(Axis(0)) == rank);
            assert!(f64_of::<$scalar>(<$scalar>::rel_diff_fro(qr.to_mat().view(), mat.view())) < $tol);
        }
    )* };
}

macro_rules! qr_compression_by_tol_tests {
    ($($name:ident: $scalar:ty, $dim:expr, $tol:expr,)*) => { $(
        #[test]
        fn $name() {
            let (m, n): (usize, usize) = $dim;
            let mut rng = rng(stringify!($name));
            let mat = <$scalar>::random_approximate_low_rank_matrix((m, n), 1.0, 1E-10, &mut rng);
            let qr = QR::<$scalar>::compute_from(mat.view()).unwrap().compress(CompressionType::ADAPTIVE($tol)).unwrap();
            assert!(f64_of::<$scalar>(<$scalar>::rel_diff_fro(qr.to_mat().view(), mat.view())) < 5.0 * $tol);
            assert!(qr.q.ncols() < m.min(n));
        }
    )* };
}

macro_rules! col_id_compression_tests {
    ($($name:ident: $scalar:ty, $dim:expr, $tol:expr,)*) => { $(
        #[test]
        fn $name() {
            let (m, n) = $dim;
            let mut rng = rng(stringify!($name));
            let mat = <$scalar>::random_approximate_low_rank_matrix((m, n), 1.0, 1E-10, &mut rng);
            let qr = QR::<$scalar>::compute_from(mat.view()).unwrap().compress(CompressionType::ADAPTIVE($tol)).unwrap();
            let rank = qr.rank();
            let column_id = qr.column_id().unwrap();
            assert!(f64_of::<$scalar>(<$scalar>::rel_diff_fro(column_id.to_mat().view(), mat.view())) < 5.0 * $tol);
            let mat_permuted = mat.apply_permutation(column_id.get_col_ind(), MatrixPermutationMode::COL);
            for index in 0..rank {
                assert!(f64_of::<$scalar>(<$scalar>::rel_diff_l2(mat_permuted.index_axis(Axis(1), index), column_id.get_c().index_axis(Axis(1), index))) < $tol);
            }
        }
    )* };
}

macro_rules! row_id_compression_tests {
    ($($name:ident: $scalar:ty, $dim:expr, $tol:expr,)*) => { $(
        #[test]
        fn $name() {
            let (m, n) = $dim;
            let mut rng = rng(stringify!($name));
            let mat = <$scalar>::random_approximate_low_rank_matrix((m, n), 1.0, 1E-10, &mut rng);
            let lq = LQ::<$scalar>::compute_from(mat.view()).unwrap().compress(CompressionType::ADAPTIVE($tol)).unwrap();
            let rank = lq.rank();
            let row_id = lq.row_id().unwrap();
            assert!(f64_of::<$scalar>(<$scalar>::rel_diff_fro(row_id.to_mat().view(), mat.view())) < 5.0 * $tol);
            let mat_permuted = mat.apply_permutation(row_id.get_row_ind(), MatrixPermutationMode::ROW);
            for index in 0..rank {
                assert!(f64_of::<$scalar>(<$scalar>::rel_diff_l2(mat_permuted.index_axis(Axis(0), index), row_id.get_r().index_axis(Axis(0), index))) < $tol);
            }
        }
    )* };
}

macro_rules! svd_to_qr_tests {
    ($($name:ident: $scalar:ty, $dim:expr, $tol:expr,)*) => { $(
        #[test]
        fn $name() {
            let (m, n) = $dim;
            let mut rng = rng(stringify!($name));
            let mat = <$scalar>::random_approximate_low_rank_matrix((m, n), 1.0, 1E-10, &mut rng);
            let svd = SVD::<$scalar>::compute_from(mat.view()).unwrap();
            // Perform a QR decomposition and recover the original matrix.
            let actual = svd.to_qr().unwrap().to_mat();
            assert!(f64_of::<$scalar>(<$scalar>::rel_diff_fro(actual.view(), mat.view())) < $tol);
        }
    )* };
}

macro_rules! svd_compression_by_rank_tests {
    ($($name:ident: $scalar:ty, $dim:expr, $tol:expr,)*) => { $(
        #[test]
        fn $name() {
            let (m, n) = $dim;
            let max_rank = 20;
            let mut rng = rng(stringify!($name));
            let mat = <$scalar>::random_approximate_low_rank_matrix((m, n), 1.0, 1E-10, &mut rng);
            let svd = SVD::<$scalar>::compute_from(mat.view()).unwrap().compress(CompressionType::RANK(max_rank)).unwrap();
            assert!(svd.u.len_of(Axis(1)) == max_rank);
            assert!(svd.vt.len_of(Axis(0)) == max_rank);
            assert!(f64_of::<$scalar>(<$scalar>::rel_diff_fro(svd.to_mat().view(), mat.view())) < $tol);
        }
    )* };
}

macro_rules! svd_compression_by_tol_tests {
    ($($name:ident: $scalar:ty, $dim:expr, $tol:expr,)*) => { $(
        #[test]
        fn $name() {
            let (m, n) = $dim;
            let mut rng = rng(stringify!($name));
            let mat = <$scalar>::random_approximate_low_rank_matrix((m, n), 1.0, 1E-10, &mut rng);
            let svd = SVD::<$scalar>::compute_from(mat.view()).unwrap().compress(CompressionType::ADAPTIVE($tol)).unwrap();
            assert!(f64_of::<$scalar>(<$scalar>::rel_diff_fro(svd.to_mat().view(), mat.view())) < $tol);
        }
    )* };
}

macro_rules! two_sided_from_col_id_tests {
    ($($name:ident: $scalar:ty, $dim:expr, $tol:expr,)*) => { $(
        #[test]
        fn $name() {
            let (m, n) = $dim;
            let mut rng = rng(stringify!($name));
            let mat = <$scalar>::random_approximate_low_rank_matrix((m, n), 1.0, 1E-10, &mut rng);
            let qr = QR::<$scalar>::compute_from(mat.view()).unwrap().compress(CompressionType::ADAPTIVE($tol)).unwrap();
            let rank = qr.rank();
            let two_sided_id = qr.column_id().unwrap().two_sided_id().unwrap();
            assert!(f64_of::<$scalar>(<$scalar>::rel_diff_fro(two_sided_id.to_mat().view(), mat.view())) < 5.0 * $tol);
            // X = A[row_ind[:k], col_ind[:k]]
            let row_perm = mat.apply_permutation(two_sided_id.row_ind.view(), MatrixPermutationMode::ROW);
            let mat_permuted = row_perm.apply_permutation(two_sided_id.col_ind.view(), MatrixPermutationMode::COL);
            assert!(two_sided_id.x.nrows() == two_sided_id.x.ncols());
            assert!(two_sided_id.x.nrows() == rank);
            for row_index in 0..rank {
                for col_index in 0..rank {
                    let diff = (two_sided_id.x[[row_index, col_index]] - mat_permuted[[row_index, col_index]]).abs();
                    assert!(f64_of::<$scalar>(diff) < 10.0 * $tol * f64_of::<$scalar>(mat_permuted[[row_index, col_index]].abs()));
                }
            }
        }
    )* };
}

macro_rules! two_sided_from_row_id_tests {
    ($($name:ident: $scalar:ty, $dim:expr, $tol:expr,)*) => { $(
        #[test]
        fn $name() {
            let (m, n) = $dim;
            let mut rng = rng(stringify!($name));
            let mat = <$scalar>::random_approximate_low_rank_matrix((m, n), 1.0, 1E-10, &mut rng);
            let lq = LQ::<$scalar>::compute_from(mat.view()).unwrap().compress(CompressionType::ADAPTIVE($tol)).unwrap();
            let rank = lq.rank();
            let two_sided_id = lq.row_id().unwrap().two_sided_id().unwrap();
            assert!(f64_of::<$scalar>(<$scalar>::rel_diff_fro(two_sided_id.to_mat().view(), mat.view())) < 5.0 * $tol);
            let row_perm = mat.apply_permutation(two_sided_id.row_ind.view(), MatrixPermutationMode::ROW);
            let mat_permuted = row_perm.apply_permutation(two_sided_id.col_ind.view(), MatrixPermutationMode::COL);
            assert!(two_sided_id.x.nrows() == two_sided_id.x.ncols());
            assert!(two_sided_id.x.nrows() == rank);
            for row_index in 0..rank {
                for col_index in 0..rank {
                    let diff = (two_sided_id.x[[row_index, col_index]] - mat_permuted[[row_index, col_index]]).abs();
                    assert!(f64_of::<$scalar>(diff) < 10.0 * $tol * f64_of::<$scalar>(mat_permuted[[row_index, col_index]].abs()));
                }
            }
        }
    )* };
}

pivoted_qr_tests! {
    pivoted_qr_test_thin_f64: f64, (100, 50),
    pivoted_qr_test_thin_f32: f32, (100, 50),
    pivoted_qr_test_thin_c64: c64, (100, 50),
    pivoted_qr_test_thin_c32: c32, (100, 50),
    pivoted_qr_test_thick_f64: f64, (50, 100),
    pivoted_qr_test_thick_f32: f32, (50, 100),
    pivoted_qr_test_thick_c64: c64, (50, 100),
    pivoted_qr_test_thick_c32: c32, (50, 100),
}
pivoted_lq_tests! {
    pivoted_lq_test_thin_f64: f64, (100, 50),
    pivoted_lq_test_thin_f32: f32, (100, 50),
    pivoted_lq_test_thin_c64: c64, (100, 50),
    pivoted_lq_test_thin_c32: c32, (100, 50),
    pivoted_lq_test_thick_f64: f64, (50, 100),
    pivoted_lq_test_thick_f32: f32, (50, 100),
    pivoted_lq_test_thick_c64: c64, (50, 100),
    pivoted_lq_test_thick_c32: c32, (50, 100),
}
row_id_compression_tests! {
    test_row_id_compression_by_tol_f32_thin: f32, (100, 50), 1E-4,
    test_row_id_compression_by_tol_c32_thin: c32, (100, 50), 1E-4,
    test_row_id_compression_by_tol_f64_thin: f64, (100, 50), 1E-4,
    test_row_id_compression_by_tol_c64_thin: c64, (100, 50), 1E-4,
    test_row_id_compression_by_tol_f32_thick: f32, (50, 100), 1E-4,
    test_row_id_compression_by_tol_c32_thick: c32, (50, 100), 1E-4,
    test_row_id_compression_by_tol_f64_thick: f64, (50, 100), 1E-4,
    test_row_id_compression_by_tol_c64_thick: c64, (50, 100), 1E-4,
}
col_id_compression_tests! {
    test_col_id_compression_by_tol_f32_thin: f32, (100, 50), 1E-4,
    test_col_id_compression_by_tol_c32_thin: c32, (100, 50), 1E-4,
    test_col_id_compression_by_tol_f64_thin: f64, (100, 50), 1E-4,
    test_col_id_compression_by_tol_c64_thin: c64, (100, 50), 1E-4,
    test_col_id_compression_by_tol_f32_thick: f32, (50, 100), 1E-4,
    test_col_id_compression_by_tol_c32_thick: c32, (50, 100), 1E-4,
    test_col_id_compression_by_tol_f64_thick: f64, (50, 100), 1E-4,
    test_col_id_compression_by_tol_c64_thick: c64, (50, 100), 1E-4,
}
qr_compression_by_rank_tests! {
    test_qr_compression_by_rank_f32_thin: f32, (100, 50), 1E-4,
    test_qr_compression_by_rank_f64_thin: f64, (100, 50), 1E-4,
    test_qr_compression_by_rank_c64_thin: c64, (100, 50), 1E-4,
    test_qr_compression_by_rank_f32_thick: f32, (50, 100), 1E-4,
    test_qr_compression_by_rank_c32_thick: c32, (50, 100), 1E-4,
    test_qr_compression_by_rank_f64_thick: f64, (50, 100), 1E-4,
    test_qr_compression_by_rank_c64_thick: c64, (50, 100), 1E-4,
}
qr_compression_by_tol_tests! {
    test_qr_compression_by_tol_f32_thin: f32, (100, 50), 1E-4,
    test_qr_compression_by_tol_c32_thin: c32, (100, 50), 1E-4,
    test_qr_compression_by_tol_f64_thin: f64, (100, 50), 1E-4,
    test_qr_compression_by_tol_c64_thin: c64, (100, 50), 1E-4,
    test_qr_compression_by_tol_f32_thick: f32, (50, 100), 1E-4,
    test_qr_compression_by_tol_c32_thick: c32, (50, 100), 1E-4,
    test_qr_compression_by_tol_f64_thick: f64, (50, 100), 1E-4,
    test_qr_compression_by_tol_c64_thick: c64, (50, 100), 1E-4,
}
svd_to_qr_tests! {
    test_svd_to_qr_f32_thin: f32, (100, 50), 1E-5,
    test_svd_to_qr_c32_thin: c32, (100, 50), 1E-5,
    test_svd_to_qr_f64_thin: f64, (100, 50), 1E-12,
    test_svd_to_qr_c64_thin: c64, (100, 50), 1E-12,
    test_svd_to_qr_f32_thick: f32, (50, 100), 1E-5,
    test_svd_to_qr_c32_thick: c32, (50, 100), 1E-5,
    test_svd_to_qr_f64_thick: f64, (50, 100), 1E-12,
    test_svd_to_qr_c64_thick: c64, (50, 100), 1E-12,
}
svd_compression_by_rank_tests! {
    test_svd_compression_by_rank_f32_thin: f32, (100, 50), 1E-4,
    test_svd_compression_by_rank_c32_thin: c32, (100, 50), 1E-4,
    test_svd_compression_by_rank_f64_thin: f64, (100, 50), 1E-4,
    test_svd_compression_by_rank_c64_thin: c64, (100, 50), 1E-4,
    test_svd_compression_by_rank_f32_thick: f32, (50, 100), 1E-4,
    test_svd_compression_by_rank_c32_thick: c32, (50, 100), 1E-4,
    test_svd_compression_by_rank_f64_thick: f64, (50, 100), 1E-4,
    test_svd_compression_by_rank_c64_thick: c64, (50, 100), 1E-4,
}
svd_compression_by_tol_tests! {
    test_svd_compression_by_tol_f32_thin: f32, (100, 50), 1E-4,
    test_svd_compression_by_tol_c32_thin: c32, (100, 50), 1E-4,
    test_svd_compression_by_tol_f64_thin: f64, (100, 50), 1E-4,
    test_svd_compression_by_tol_c64_thin: c64, (100, 50), 1E-4,
    test_svd_compression_by_tol_f32_thick: f32, (50, 100), 1E-4,
    test_svd_compression_by_tol_c32_thick: c32, (50, 100), 1E-4,
    test_svd_compression_by_tol_f64_thick: f64, (50, 100), 1E-4,
    test_svd_compression_by_tol_c64_thick: c64, (50, 100), 1E-4,
}
two_sided_from_col_id_tests! {
    test_two_sided_from_col_id_compression_by_tol_f32_thin: f32, (100, 50), 1E-4,
    test_two_sided_from_col_id_compression_by_tol_c32_thin: c32, (100, 50), 1E-4,
    test_two_sided_from_col_id_compression_by_tol_f64_thin: f64, (100, 50), 1E-4,
    test_two_sided_from_col_id_compression_by_tol_c64_thin: c64, (100, 50), 1E-4,
    test_two_sided_from_col_id_compression_by_tol_f32_thick: f32, (50, 100), 1E-4,
    test_two_sided_from_col_id_compression_by_tol_c32_thick: c32, (50, 100), 1E-4,
    test_two_sided_from_col_id_compression_by_tol_f64_thick: f64, (50, 100), 1E-4,
    test_two_sided_from_col_id_compression_by_tol_c64_thick: c64, (50, 100), 1E-4,
}
two_sided_from_row_id_tests! {
    test_two_sided_from_row_id_compression_by_tol_f32_thin: f32, (100, 50), 1E-4,
    test_two_sided_from_row_id_compression_by_tol_c32_thin: c32, (100, 50), 1E-4,
    test_two_sided_from_row_id_compression_by_tol_f64_thin: f64, (100, 50), 1E-4,
    test_two_sided_from_row_id_compression_by_tol_c64_thin: c64, (100, 50), 1E-4,
    test_two_sided_from_row_id_compression_by_tol_f32_thick: f32, (50, 100), 5E-4,
    test_two_sided_from_row_id_compression_by_tol_c32_thick: c32, (50, 100), 1E-4,
    test_two_sided_from_row_id_compression_by_tol_f64_thick: f64, (50, 100), 1E-4,
    test_two_sided_from_row_id_compression_by_tol_c64_thick: c64, (50, 100), 1E-4,
}

// src/permutation.rs:192-239 (known answers)
#[test]
fn test_matrix_permutation() {
    use ndarray::arr2;
    let mat = arr2(&[[1.0, 2.0, 3.0], [4.0, 5.0, 6.0], [7.0, 8.0, 9.0]]);
    let perm = ndarray::arr1(&[2usize, 0, 1]);
    assert_eq!(mat.apply_permutation(perm.view(), MatrixPermutationMode::COL), arr2(&[[3.0, 1.0, 2.0], [6.0, 4.0, 5.0], [9.0, 7.0, 8.0]]));
    assert_eq!(mat.apply_permutation(perm.view(), MatrixPermutationMode::COLINV), arr2(&[[2.0, 3.0, 1.0], [5.0, 6.0, 4.0], [8.0, 9.0, 7.0]]));
    assert_eq!(mat.apply_permutation(perm.view(), MatrixPermutationMode::ROW), arr2(&[[7.0, 8.0, 9.0], [1.0, 2.0, 3.0], [4.0, 5.0, 6.0]]));
    assert_eq!(mat.apply_permutation(perm.view(), MatrixPermutationMode::ROWINV), arr2(&[[4.0, 5.0, 6.0], [7.0, 8.0, 9.0], [1.0, 2.0, 3.0]]));
}
#[test]
fn test_vector_permutaiton() {
    let vec = ndarray::arr1(&[1.0, 2.0, 3.0]);
    let perm = ndarray::arr1(&[2usize, 0, 1]);
    assert_eq!(vec.apply_permutation(perm.view(), VectorPermutationMode::NOINV), ndarray::arr1(&[3.0, 1.0, 2.0]));
    assert_eq!(vec.apply_permutation(perm.view(), VectorPermutationMode::INV), ndarray::arr1(&[2.0, 3.0, 1.0]));
}

// The range finders over a custom operator (`impl<Op: MatMat<A = $scalar>> SampleRange for Op`, src/random_sampling.rs:102, :130,
// :222).  The reference has no test of these; this one mirrors tests/cpp/reference_tests.cpp::operator_tests: an operator that
// is ONLY a pair of matvecs (A = U V^T held as factors; `matmat` / `conj_matmat` are the traits' per-column defaults) goes through
// rc_*_op_f64, and agrees with the same operator given as a dense array.
struct Factored {
    u: ndarray::Array2<f64>,
    v: ndarray::Array2<f64>,
}
impl types::MatVec for Factored {
    type A = f64;
    fn nrows(&self) -> usize { self.u.nrows() }
    fn ncols(&self) -> usize { self.v.nrows() }
    fn matvec(&self, x: ndarray::ArrayView1<f64>) -> ndarray::Array1<f64> { self.u.dot(&self.v.t().dot(&x)) }
}
impl types::ConjMatVec for Factored {
    fn conj_matvec(&self, x: ndarray::ArrayView1<f64>) -> ndarray::Array1<f64> { self.v.dot(&self.u.t().dot(&x)) }
}
impl types::MatMat for Factored {}
impl types::ConjMatMat for Factored {}

#[test]
fn test_range_finders_over_a_matvec_only_operator() {
    let (m, n, r, k) = (300usize, 200usize, 30usize, 12usize);
    let mut u = f64::random_orthogonal_matrix((m, r), &mut rng("op-u"));
    for (j, mut col) in u.axis_iter_mut(Axis(1)).enumerate() {
        col.mapv_inplace(|x| x * 10f64.powf(-6.0 * j as f64 / (r - 1) as f64));
    }
    let v = f64::random_orthogonal_matrix((n, r), &mut rng("op-v"));
    let op = Factored { u, v };
    let dense = op.u.dot(&op.v.t());
    let q_op = op.sample_range_by_rank(k, 8, &mut rng("omega")).unwrap();
    let q_dense = dense.sample_range_by_rank(k, 8, &mut rng("omega")).unwrap();
    assert!(f64::rel_diff_fro(q_op.view(), q_dense.view()) < 1E-9);
    let svd = SVD::<f64>::compute_from_range_estimate(q_op.view(), &op).unwrap();
    for j in 0..3 {
        let sj = 10f64.powf(-6.0 * j as f64 / (r - 1) as f64);
        assert!((svd.get_s()[j] - sj).abs() < 1E-2 * sj);   // the accuracy a rank-12 sketch of this spectrum gives
    }
    let (q_ad, hist) = op.sample_range_adaptive(1E-4, 10, &mut rng("adaptive")).unwrap();
    assert!(!hist.is_empty() && hist.last().unwrap().1 < 1E-4 && q_ad.ncols() <= 60);
}
