//! `random_matrix` module of the reference (`src/random_matrix.rs`).
use crate::device::{Context, DeviceMatrix};
use crate::svd::{SVDTraits, SVD};
use crate::types::{c32, c64, Scalar};
use ndarray::{Array1, Array2};
use rand::Rng;
use rand_distr::{Distribution, Normal};

/// reference `src/random_matrix.rs:11-93`
pub trait RandomMatrix: Scalar {
    /// i.i.d. N(0, 1), drawn on the host from the caller's generator in row-major order, in f64, then cast
    /// (reference `src/random_matrix.rs:120-125`): a seeded generator reproduces the reference's matrix bit for bit.
    fn random_gaussian<R: Rng>(dimension: (usize, usize), rng: &mut R) -> Array2<Self>;

    /// reference `src/random_matrix.rs:35-56`: U of the thin SVD of a Gaussian (rows orthonormal if wide)
    fn random_orthogonal_matrix<R: Rng>(dimension: (usize, usize), rng: &mut R) -> Array2<Self>
    where
        SVD<Self>: SVDTraits<A = Self>,
    {
        let (m, n) = dimension;
        let swap = n > m;
        let (mm, nn) = if swap { (n, m) } else { (m, n) };
        let mat = Self::random_gaussian((mm, nn), rng);
        let u = SVD::<Self>::compute_from(mat.view()).expect("SVD failed").u;
        if swap { u.t().to_owned() } else { u }
    }

    /// reference `src/random_matrix.rs:70-93`: U diag(geomspace(sigma_min, sigma_max)) V^T.
    fn random_approximate_low_rank_matrix<R: Rng>(dimension: (usize, usize), sigma_max: f64, sigma_min: f64, rng: &mut R) -> Array2<Self>
    where
        SVD<Self>: SVDTraits<A = Self>,
    {
        assert!(sigma_min < sigma_max, "`sigma_min` must be smaller than `sigma_max`");
        assert!(sigma_min > 0.0, "`sigma_min` must be positive.");
        let (m, n) = dimension;
        let r = m.min(n);
        let u = Self::random_orthogonal_matrix((m, r), rng);
        let vt = Self::random_orthogonal_matrix((r, n), rng);
        let s: Array1<Self::Real> = Array1::geomspace(sigma_min, sigma_max, r)
            .expect("geomspace")
            .iter()
            .map(|&x| num_traits::cast::<f64, Self::Real>(x).unwrap())
            .collect();
        SVD::<Self> { u, s, vt }.to_mat()
    }
}

impl RandomMatrix for f64 {
    fn random_gaussian<R: Rng>(dimension: (usize, usize), rng: &mut R) -> Array2<f64> {
        let normal = Normal::new(0.0, 1.0).unwrap();
        let mut mat = Array2::<f64>::zeros(dimension);
        mat.map_inplace(|item| *item = normal.sample(rng));
        mat
    }
}
impl RandomMatrix for f32 {
    fn random_gaussian<R: Rng>(dimension: (usize, usize), rng: &mut R) -> Array2<f32> {
        let normal = Normal::new(0.0, 1.0).unwrap();
        let mut mat = Array2::<f32>::zeros(dimension);
        mat.map_inplace(|item| { let v: f64 = normal.sample(rng); *item = v as f32 });
        mat
    }
}
#[cfg(feature = "complex")]
impl RandomMatrix for c64 {
    fn random_gaussian<R: Rng>(dimension: (usize, usize), rng: &mut R) -> Array2<c64> {
        // real and imaginary part each N(0, 1) (reference src/random_matrix.rs:136-143)
        let normal = Normal::new(0.0, 1.0).unwrap();
        let mut mat = Array2::<c64>::zeros(dimension);
        mat.map_inplace(|item| { let re: f64 = normal.sample(rng); let im: f64 = normal.sample(rng); *item = c64::new(re, im) });
        mat
    }
}
#[cfg(feature = "complex")]
impl RandomMatrix for c32 {
    fn random_gaussian<R: Rng>(dimension: (usize, usize), rng: &mut R) -> Array2<c32> {
        let normal = Normal::new(0.0, 1.0).unwrap();
        let mut mat = Array2::<c32>::zeros(dimension);
        mat.map_inplace(|item| { let re: f64 = normal.sample(rng); let im: f64 = normal.sample(rng); *item = c32::new(re as f32, im as f32) });
        mat
    }
}

/// The engine's own generator: Philox4x32-10 + Box-Muller on the device (stream contract in
/// include/rusty_compression_amd.h); element (i, j) is number `offset + i * cols + j` of the stream `seed`.
pub fn random_gaussian_device<A: Scalar>(ctx: &Context, dimension: (usize, usize), seed: u64, offset: u64) -> crate::types::Result<DeviceMatrix<A>> {
    let out = DeviceMatrix::<A>::zeros(ctx, dimension.0, dimension.1)?;
    ctx.check(unsafe { A::ffi_random_gaussian(ctx.raw(), out.view(), seed, offset) })?;
    Ok(out)
}
