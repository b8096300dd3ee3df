// Complex scalars (c32 / c64) of the hot path: the `_c32` / `_c64` entry points of the C ABI.
//
// The reference instantiates every trait for f32, f64, c32, c64 through macros
// (/root/reference/src/qr.rs:408-416, src/pivoted_qr.rs:187-190, src/svd.rs:188-191, src/random_sampling.rs:123-126,
// :165-168, :277-280, src/types.rs:198-204); conjugation semantics: `conj_matmat` is A^H X (src/types.rs:128-132),
// Q^H everywhere the real code has Q^T.  LAPACK routines behind it: ?geqp3 with rwork (src/pivoted_qr.rs:161-172),
// ?ungqr, ?gesdd, ?trtrs on complex data.
//
// Design.  Complex data is interleaved (re, im) -- the layout of ndarray's Complex<T> / numpy complex -- and every
// operand is first brought into a column-major working copy (any strides, optional conjugate / transpose folded into
// that copy), so the kernels see one layout:
//   * products: "4M" on the real MFMA GEMM -- the operands are split into real and imaginary planes, four real GEMMs
//     (kernels_gemm.hip) form Re C and Im C, one kernel recombines with the complex alpha / beta;
//   * pivoted QR: ?geqp3 / ?laqp2 semantics with complex Householder reflectors (?larfg: beta real, tau complex;
//     applied as H^H = I - conj(tau) v v^H), LAPACK's norm down-dating, first-maximum pivoting; Q by ?ung2r;
//   * SVD: one-sided (Hestenes) Jacobi with complex rotations on the tall orientation, one launch per round of the
//     circle-method schedule (the pairs of a round are disjoint); singular values are real, descending;
//   * triangular solves: back substitution, one thread per right-hand side.
// These are correct, deterministic and column-parallel, but not tuned like the real-scalar kernels: complex is the
// breadth row of SURVEY.md 8(f), the measured hot path (BASELINE.json) is real.
#include "rc_common.hpp"
#include "rc_device.hpp"

#include <algorithm>
#include <cmath>

using namespace rc;

namespace {

template <typename R>
struct cplx {
    R re, im;
};
template <typename R> __host__ __device__ inline cplx<R> mk(R a, R b) { return cplx<R>{a, b}; }
template <typename R> __host__ __device__ inline cplx<R> operator+(cplx<R> a, cplx<R> b) { return {a.re + b.re, a.im + b.im}; }
template <typename R> __host__ __device__ inline cplx<R> operator-(cplx<R> a, cplx<R> b) { return {a.re - b.re, a.im - b.im}; }
template <typename R> __host__ __device__ inline cplx<R> operator*(cplx<R> a, cplx<R> b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
template <typename R> __host__ __device__ inline cplx<R> operator*(R s, cplx<R> a) { return {s * a.re, s * a.im}; }
template <typename R> __host__ __device__ inline cplx<R> cj(cplx<R> a) { return {a.re, -a.im}; }
template <typename R> __host__ __device__ inline R abs2(cplx<R> a) { return a.re * a.re + a.im * a.im; }
template <typename R> __device__ inline R cabs(cplx<R> a) { return hypot(a.re, a.im); }
template <typename R> __device__ inline cplx<R> cdiv(cplx<R> a, cplx<R> b) {  // Smith's algorithm (?ladiv)
    if (fabs(b.re) >= fabs(b.im)) {
        const R r = b.im / b.re, d = b.re + b.im * r;
        return {(a.re + a.im * r) / d, (a.im - a.re * r) / d};
    }
    const R r = b.re / b.im, d = b.im + b.re * r;
    return {(a.re * r + a.im) / d, (a.im * r - a.re) / d};
}

static inline int64_t cdivi(int64_t a, int64_t b) { return (a + b - 1) / b; }

template <typename R> struct NumC;
template <> struct NumC<double> { static __host__ __device__ double tol3z() { return 1.0536712127723509e-08; } static __host__ __device__ double eps() { return 1.1102230246251565e-16; } };
template <> struct NumC<float> { static __host__ __device__ float tol3z() { return 2.44140625e-04f; } static __host__ __device__ float eps() { return 5.9604645e-08f; } };

// strided complex view (mirror of rc_matrix) and the column-major working matrix
template <typename R>
struct CV {
    cplx<R> *p;
    int64_t rows, cols, rs, cs;
    __host__ __device__ cplx<R> &at(int64_t i, int64_t j) const { return p[i * rs + j * cs]; }
    CV sub(int64_t r0, int64_t nr, int64_t c0, int64_t nc) const { return CV{p + r0 * rs + c0 * cs, nr, nc, rs, cs}; }
    CV t() const { return CV{p, cols, rows, cs, rs}; }
    bool empty() const { return rows == 0 || cols == 0; }
};
template <typename R>
CV<R> view_of(const rc_matrix &m) { return CV<R>{static_cast<cplx<R> *>(m.data), m.rows, m.cols, m.row_stride, m.col_stride}; }
template <typename R>
CV<R> tmp_cm(rc_context *c, int64_t rows, int64_t cols) {
    const int64_t ld = std::max<int64_t>(rows, 1);
    return CV<R>{c->alloc<cplx<R>>((size_t)ld * std::max<int64_t>(cols, 1)), rows, cols, 1, ld};
}

// ------------------------------------------------------------------------------------------------ elementwise kernels
template <typename R>
__global__ __launch_bounds__(256) void k_c_copy(CV<R> src, CV<R> dst, int conj) {  // dst = src or conj(src); any strides
    const int64_t total = dst.rows * dst.cols;
    const bool col_fast = dst.cs <= dst.rs;
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int64_t i, j;
        if (col_fast) { i = e / dst.cols; j = e - i * dst.cols; }
        else { j = e / dst.rows; i = e - j * dst.rows; }
        cplx<R> v = src.at(i, j);
        if (conj) v.im = -v.im;
        dst.at(i, j) = v;
    }
}
template <typename R>
void c_copy(rc_context *c, CV<R> src, CV<R> dst, bool conj = false) {
    RC_REQUIRE(src.rows == dst.rows && src.cols == dst.cols, RC_INVALID_ARGUMENT, "complex copy: shape mismatch");
    if (dst.empty()) return;
    hipLaunchKernelGGL(k_c_copy<R>, dim3((unsigned)std::min<int64_t>(cdivi(dst.rows * dst.cols, 256), 8192)), dim3(256), 0, c->stream, src, dst, conj ? 1 : 0);
}
template <typename R>
__global__ __launch_bounds__(256) void k_c_fill(CV<R> dst, int identity) {
    const int64_t total = dst.rows * dst.cols;
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t j = e / dst.rows, i = e - j * dst.rows;
        dst.at(i, j) = cplx<R>{(identity && i == j) ? (R)1 : (R)0, (R)0};
    }
}
template <typename R>
void c_fill(rc_context *c, CV<R> dst, bool identity) {
    if (dst.empty()) return;
    hipLaunchKernelGGL(k_c_fill<R>, dim3((unsigned)std::min<int64_t>(cdivi(dst.rows * dst.cols, 256), 8192)), dim3(256), 0, c->stream, dst, identity ? 1 : 0);
}
template <typename R>
__global__ __launch_bounds__(256) void k_c_gather_cols(CV<R> src, const int64_t *idx, CV<R> dst, int *health) {  // dst[:, j] = src[:, idx[j]]
    const int64_t total = dst.rows * dst.cols;
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t j = e / dst.rows, i = e - j * dst.rows;
        const int64_t s = idx[j];
        if ((uint64_t)s < (uint64_t)src.cols) {
            dst.at(i, j) = src.at(i, s);
        } else {  // (as k_gather_cols: no wild addresses; health bit 32)
            dst.at(i, j) = cplx<R>{0, 0};
            if (i == 0) atomicOr(health, 32);
        }
    }
}
template <typename R>
void c_gather_cols(rc_context *c, CV<R> src, const int64_t *idx, CV<R> dst) {
    if (dst.empty()) return;
    hipLaunchKernelGGL(k_c_gather_cols<R>, dim3((unsigned)std::min<int64_t>(cdivi(dst.rows * dst.cols, 256), 8192)), dim3(256), 0, c->stream, src, idx, dst, c->health_word());
}
template <typename R>
__global__ __launch_bounds__(256) void k_c_scale_rows(const R *s, CV<R> m) {  // m[i, :] *= s[i]
    const int64_t total = m.rows * m.cols;
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t j = e / m.rows, i = e - j * m.rows;
        cplx<R> v = m.at(i, j);
        m.at(i, j) = s[i] * v;
    }
}
template <typename R>
__global__ __launch_bounds__(256) void k_c_sub(CV<R> y, CV<R> corr) {  // y -= corr
    const int64_t total = y.rows * y.cols;
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t j = e / y.rows, i = e - j * y.rows;
        y.at(i, j) = y.at(i, j) - corr.at(i, j);
    }
}
// planes for the 4M product: re / im as real column-major matrices with the same leading dimension
template <typename R>
__global__ __launch_bounds__(256) void k_c_split(CV<R> src, R *re, R *im, int64_t ld) {
    const int64_t total = src.rows * src.cols;
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t j = e / src.rows, i = e - j * src.rows;
        const cplx<R> v = src.at(i, j);
        re[i + j * ld] = v.re;
        im[i + j * ld] = v.im;
    }
}
template <typename R>
__global__ __launch_bounds__(256) void k_c_combine(const R *re, const R *im, int64_t ld, cplx<R> alpha, cplx<R> beta, CV<R> dst) {
    const int64_t total = dst.rows * dst.cols;
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t j = e / dst.rows, i = e - j * dst.rows;
        cplx<R> v = alpha * cplx<R>{re[i + j * ld], im[i + j * ld]};
        if (beta.re != (R)0 || beta.im != (R)0) v = v + beta * dst.at(i, j);
        dst.at(i, j) = v;
    }
}

template <typename R>
__device__ inline R wsumc(R v) { return wave_sum_dpp(v); }

// out[j] = ||a[:, j]||^2 (column-major a), one wave per column
template <typename R>
__global__ __launch_bounds__(256) void k_c_col_sumsq(CV<R> a, R *out) {
    const int lane = threadIdx.x & 63;
    for (int64_t j = blockIdx.x * 4 + (threadIdx.x >> 6); j < a.cols; j += (int64_t)gridDim.x * 4) {
        R acc = 0;
        for (int64_t i = lane; i < a.rows; i += 64) acc += abs2(a.at(i, j));
        acc = wsumc(acc);
        if (lane == 0) out[j] = acc;
    }
}
// out2[0] = ||a - b||_F^2, out2[1] = ||b||_F^2 : one workgroup, fixed order (small matrices only on this path)
template <typename R>
__global__ __launch_bounds__(1024) void k_c_fro(CV<R> a, CV<R> b, R *out2) {
    __shared__ R sh[32];
    R d2 = 0, b2 = 0;
    const int64_t total = a.rows * a.cols;
    for (int64_t e = threadIdx.x; e < total; e += 1024) {
        const int64_t j = e / a.rows, i = e - j * a.rows;
        const cplx<R> x = a.at(i, j), y = b.at(i, j);
        d2 += abs2(x - y);
        b2 += abs2(y);
    }
    d2 = wsumc(d2);
    b2 = wsumc(b2);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = d2; sh[16 + (threadIdx.x >> 6)] = b2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        R s1 = 0, s2 = 0;
        for (int i = 0; i < 16; ++i) { s1 += sh[i]; s2 += sh[16 + i]; }
        out2[0] = s1;
        out2[1] = s2;
    }
}

// ------------------------------------------------------------------------------------------------ products (4M)
// C = alpha op(A) op(B) + beta C;  A, B, C: any strided views;  op: 0 none, 1 transpose, 2 conjugate transpose
template <typename R>
void c_gemm(rc_context *c, int opa, int opb, cplx<R> alpha, CV<R> a, CV<R> b, cplx<R> beta, CV<R> cm) {
    if (opa) a = a.t();
    if (opb) b = b.t();
    const R sa = opa == 2 ? (R)-1 : (R)1, sb = opb == 2 ? (R)-1 : (R)1;  // sign of the imaginary plane
    RC_REQUIRE(a.cols == b.rows && a.rows == cm.rows && b.cols == cm.cols, RC_INVALID_ARGUMENT, "complex gemm: shapes (%lld x %lld) * (%lld x %lld) -> (%lld x %lld)",
               (long long)a.rows, (long long)a.cols, (long long)b.rows, (long long)b.cols, (long long)cm.rows, (long long)cm.cols);
    if (cm.empty()) return;
    ArenaMark mark(c);
    const int64_t m = a.rows, k = a.cols, n = b.cols;
    const int64_t lda = even_ld(std::max<int64_t>(m, 1)), ldb = even_ld(std::max<int64_t>(k, 1)), ldc = even_ld(std::max<int64_t>(m, 1));
    R *ar = c->alloc<R>((size_t)lda * std::max<int64_t>(k, 1)), *ai = c->alloc<R>((size_t)lda * std::max<int64_t>(k, 1));
    R *br = c->alloc<R>((size_t)ldb * n), *bi = c->alloc<R>((size_t)ldb * n);
    R *cr = c->alloc<R>((size_t)ldc * n), *ci = c->alloc<R>((size_t)ldc * n);
    if (k == 0) {  // empty inner dimension: C = beta C
        fill_words(c, cr, (size_t)ldc * n * sizeof(R), 0u);
        fill_words(c, ci, (size_t)ldc * n * sizeof(R), 0u);
        hipLaunchKernelGGL(k_c_combine<R>, dim3((unsigned)std::min<int64_t>(cdivi(m * n, 256), 8192)), dim3(256), 0, c->stream, cr, ci, ldc, alpha, beta, cm);
        return;
    }
    if (k > 0) {
        hipLaunchKernelGGL(k_c_split<R>, dim3((unsigned)std::min<int64_t>(cdivi(m * k, 256), 8192)), dim3(256), 0, c->stream, a, ar, ai, lda);
        hipLaunchKernelGGL(k_c_split<R>, dim3((unsigned)std::min<int64_t>(cdivi(k * n, 256), 8192)), dim3(256), 0, c->stream, b, br, bi, ldb);
    }
    Mat<R> Ar = colmajor(ar, m, k, lda), Ai = colmajor(ai, m, k, lda), Br = colmajor(br, k, n, ldb), Bi = colmajor(bi, k, n, ldb);
    Mat<R> Cr = colmajor(cr, m, n, ldc), Ci = colmajor(ci, m, n, ldc);
    // Re C = Ar Br - (sa sb) Ai Bi ;  Im C = sb Ar Bi + sa Ai Br   (four products on the real MFMA GEMM)
    gemm<R>(c, (R)1, Ar, Br, (R)0, Cr);
    gemm<R>(c, -(sa * sb), Ai, Bi, (R)1, Cr);
    gemm<R>(c, sb, Ar, Bi, (R)0, Ci);
    gemm<R>(c, sa, Ai, Br, (R)1, Ci);
    hipLaunchKernelGGL(k_c_combine<R>, dim3((unsigned)std::min<int64_t>(cdivi(m * n, 256), 8192)), dim3(256), 0, c->stream, cr, ci, ldc, alpha, beta, cm);
}
template <typename R> cplx<R> one() { return cplx<R>{(R)1, (R)0}; }
template <typename R> cplx<R> zero() { return cplx<R>{(R)0, (R)0}; }

// ------------------------------------------------------------------------------------------------ pivoted QR
// step j: pivot (first maximum), swap by index, ?larfg on the pivot column; one workgroup
template <typename R>
__global__ __launch_bounds__(1024) void k_c_qr_pivot_reflect(CV<R> w, int64_t j, int pivot, int64_t *jpvt, R *vn1, R *vn2, cplx<R> *tau) {
    __shared__ R shv[16];
    __shared__ long long shi[16];
    __shared__ R shs[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t n = w.cols, m = w.rows;
    if (pivot) {
        R best = (R)-1;
        long long bi = 0x7fffffffffffffffLL;
        for (int64_t p = j + tid; p < n; p += 1024) {
            const R v = fabs(vn1[p]);
            if (v > best) { best = v; bi = p; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const R ob = __shfl_xor(best, off, 64);
            const long long oi = __shfl_xor(bi, off, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) { shv[wv] = best; shi[wv] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int q = 1; q < 16; ++q)
                if (shv[q] > best || (shv[q] == best && shi[q] < bi)) { best = shv[q]; bi = shi[q]; }
            const int64_t pvt = (bi >= j && bi < n) ? (int64_t)bi : j;
            if (pvt != j) {
                const int64_t t = jpvt[pvt]; jpvt[pvt] = jpvt[j]; jpvt[j] = t;
                vn1[pvt] = vn1[j];
                vn2[pvt] = vn2[j];
            }
        }
        __syncthreads();
    }
    cplx<R> *col = w.p + jpvt[j] * w.cs;
    const cplx<R> alpha = col[j];
    R acc = 0;
    for (int64_t i = j + 1 + tid; i < m; i += 1024) acc += abs2(col[i]);
    acc = wsumc(acc);
    if (lane == 0) shs[wv] = acc;
    __syncthreads();
    R ssq = 0;
    for (int q = 0; q < 16; ++q) ssq += shs[q];
    const R xnorm = sqrt(ssq);
    if (xnorm == (R)0 && alpha.im == (R)0) {  // ?larfg: H = I
        if (tid == 0) tau[j] = cplx<R>{0, 0};
        return;
    }
    const R beta = -copysign(sqrt(alpha.re * alpha.re + alpha.im * alpha.im + ssq), alpha.re);  // ?lapy3
    const cplx<R> scal = cdiv(cplx<R>{(R)1, (R)0}, cplx<R>{alpha.re - beta, alpha.im});
    for (int64_t i = j + 1 + tid; i < m; i += 1024) col[i] = col[i] * scal;
    if (tid == 0) {
        tau[j] = cplx<R>{(beta - alpha.re) / beta, -alpha.im / beta};
        col[j] = cplx<R>{beta, (R)0};
    }
}
// apply H_j^H = I - conj(tau) v v^H to every remaining column + LAPACK norm down-date; one workgroup per column
template <typename R>
__global__ __launch_bounds__(256) void k_c_qr_apply(CV<R> w, int64_t j, int pivot, const int64_t *jpvt, R *vn1, R *vn2, const cplx<R> *tau) {
    __shared__ R sh[8];
    const int64_t n = w.cols, m = w.rows;
    const int64_t p = j + 1 + blockIdx.x;
    if (p >= n) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const cplx<R> tj = tau[j];
    const cplx<R> *v = w.p + jpvt[j] * w.cs;
    cplx<R> *x = w.p + jpvt[p] * w.cs;
    if (tj.re != (R)0 || tj.im != (R)0) {
        cplx<R> dot{0, 0};  // v^H x
        for (int64_t i = j + tid; i < m; i += 256) {
            const cplx<R> vi = i == j ? cplx<R>{(R)1, (R)0} : v[i];
            dot = dot + cj(vi) * x[i];
        }
        R dr = wsumc(dot.re), di = wsumc(dot.im);
        __syncthreads();
        if (lane == 0) { sh[wv] = dr; sh[4 + wv] = di; }
        __syncthreads();
        dr = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        di = (sh[4] + sh[5]) + (sh[6] + sh[7]);
        const cplx<R> f = cj(tj) * cplx<R>{dr, di};
        for (int64_t i = j + tid; i < m; i += 256) {
            const cplx<R> vi = i == j ? cplx<R>{(R)1, (R)0} : v[i];
            x[i] = x[i] - f * vi;
        }
    }
    if (!pivot) return;
    __syncthreads();
    const R vn = vn1[p];
    if (vn == (R)0) return;
    const R t = cabs(x[j]) / vn;
    R temp = (R)1 - t * t;
    temp = temp > (R)0 ? temp : (R)0;
    const R r = vn / vn2[p];
    if (temp * r * r <= NumC<R>::tol3z()) {
        R ss = 0;
        for (int64_t i = j + 1 + tid; i < m; i += 256) ss += abs2(x[i]);
        ss = wsumc(ss);
        __syncthreads();
        if (lane == 0) sh[wv] = ss;
        __syncthreads();
        if (tid == 0) { const R nn = (j < m - 1) ? sqrt((sh[0] + sh[1]) + (sh[2] + sh[3])) : (R)0; vn1[p] = nn; vn2[p] = nn; }
    } else if (tid == 0) {
        vn1[p] = vn * sqrt(temp);
    }
}
template <typename R>
__global__ __launch_bounds__(256) void k_c_qr_init(CV<R> w, int64_t *jpvt, R *vn1, R *vn2) {
    const int lane = threadIdx.x & 63;
    for (int64_t j = blockIdx.x * 4 + (threadIdx.x >> 6); j < w.cols; j += (int64_t)gridDim.x * 4) {
        R acc = 0;
        for (int64_t i = lane; i < w.rows; i += 64) acc += abs2(w.at(i, j));
        acc = sqrt(wsumc(acc));
        if (lane == 0) { vn1[j] = acc; vn2[j] = acc; jpvt[j] = j; }
    }
}
// w: column-major m x n, overwritten with the ?geqp3 output format (columns never moved: jpvt maps position -> column)
template <typename R>
void c_geqp3(rc_context *c, CV<R> w, int64_t kmax, int64_t *jpvt, cplx<R> *tau) {
    const int64_t m = w.rows, n = w.cols;
    if (m == 0 || n == 0) return;
    kmax = std::min(kmax, std::min(m, n));
    ProfScope ps(c, "op:geqp3<complex> %lldx%lld k=%lld", (long long)m, (long long)n, (long long)kmax);
    ArenaMark mark(c);
    R *vn1 = c->alloc<R>((size_t)n), *vn2 = c->alloc<R>((size_t)n);
    hipLaunchKernelGGL(k_c_qr_init<R>, dim3((unsigned)std::min<int64_t>(cdivi(n, 4), 8192)), dim3(256), 0, c->stream, w, jpvt, vn1, vn2);
    for (int64_t j = 0; j < kmax; ++j) {
        hipLaunchKernelGGL(k_c_qr_pivot_reflect<R>, dim3(1), dim3(1024), 0, c->stream, w, j, 1, jpvt, vn1, vn2, tau);
        if (n - j - 1 > 0) hipLaunchKernelGGL(k_c_qr_apply<R>, dim3((unsigned)(n - j - 1)), dim3(256), 0, c->stream, w, j, 1, jpvt, vn1, vn2, tau);
    }
}
// r(i, p) = (i <= p) ? w(i, jpvt[p]) : 0
template <typename R>
__global__ __launch_bounds__(256) void k_c_extract_r(CV<R> w, const int64_t *jpvt, CV<R> r) {
    const int64_t total = r.rows * r.cols;
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t p = e / r.rows, i = e - p * r.rows;
        r.at(i, p) = i <= p ? w.p[jpvt[p] * w.cs + i] : cplx<R>{0, 0};
    }
}
// q(:, cq) = H_0 ... H_{k-1} e_cq (?ung2r), one workgroup per column, the column lives in q
template <typename R>
__global__ __launch_bounds__(256) void k_c_form_q(CV<R> w, const int64_t *jpvt, const cplx<R> *tau, int64_t k, CV<R> q) {
    __shared__ R sh[8];
    const int64_t m = w.rows, cq = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    cplx<R> *x = q.p + cq * q.cs;
    for (int64_t i = tid; i < m; i += 256) x[i] = cplx<R>{i == cq ? (R)1 : (R)0, (R)0};
    for (int64_t j = (cq < k - 1 ? cq : k - 1); j >= 0; --j) {
        const cplx<R> tj = tau[j];
        if (tj.re == (R)0 && tj.im == (R)0) continue;
        const cplx<R> *v = w.p + jpvt[j] * w.cs;
        // thread tid owns the rows i = tid (mod 256) for the whole kernel: x[i] is only ever touched by its owner (the loops used
        // to start at j + tid, which moved a row from thread to thread between steps with no barrier in between: a race between
        // waves, visible as ~1e-4 errors of the c32 factor at k = 200)
        const int64_t i0 = tid >= j ? tid : tid + ((j - tid + 255) / 256) * 256;
        cplx<R> dot{0, 0};
        for (int64_t i = i0; i < m; i += 256) {
            const cplx<R> vi = i == j ? cplx<R>{(R)1, (R)0} : v[i];
            dot = dot + cj(vi) * x[i];
        }
        R dr = wsumc(dot.re), di = wsumc(dot.im);
        __syncthreads();
        if (lane == 0) { sh[wv] = dr; sh[4 + wv] = di; }
        __syncthreads();
        const cplx<R> f = tj * cplx<R>{(sh[0] + sh[1]) + (sh[2] + sh[3]), (sh[4] + sh[5]) + (sh[6] + sh[7])};
        for (int64_t i = i0; i < m; i += 256) {
            const cplx<R> vi = i == j ? cplx<R>{(R)1, (R)0} : v[i];
            x[i] = x[i] - f * vi;
        }
    }
}
// T X = B in place, T upper triangular k x k, one thread per right-hand side column
template <typename R>
__global__ __launch_bounds__(256) void k_c_trsm_upper(CV<R> t, CV<R> b) {
    const int64_t col = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (col >= b.cols) return;
    const int64_t k = t.rows;
    for (int64_t i = k - 1; i >= 0; --i) {
        cplx<R> acc = b.at(i, col);
        for (int64_t l = i + 1; l < k; ++l) acc = acc - t.at(i, l) * b.at(l, col);
        b.at(i, col) = cdiv(acc, t.at(i, i));
    }
}

// A P = Q R with k = q.cols = r.rows steps;  a: any view (not modified), q / r: any views (may be empty)
template <typename R>
void c_pivoted_qr(rc_context *c, CV<R> a, CV<R> q, CV<R> r, int64_t *ind, int64_t k) {
    const int64_t m = a.rows, n = a.cols;
    RC_REQUIRE(k <= std::min(m, n), RC_INVALID_ARGUMENT, "pivoted_qr: rank %lld exceeds min(m, n)", (long long)k);
    if (n == 0) return;
    ArenaMark mark(c);
    CV<R> w = tmp_cm<R>(c, m, n);
    c_copy(c, a, w);
    cplx<R> *tau = c->alloc<cplx<R>>((size_t)std::max<int64_t>(k, 1));
    c_geqp3(c, w, k, ind, tau);
    if (!r.empty()) {
        CV<R> rw = tmp_cm<R>(c, r.rows, n);
        hipLaunchKernelGGL(k_c_extract_r<R>, dim3((unsigned)std::min<int64_t>(cdivi(r.rows * n, 256), 8192)), dim3(256), 0, c->stream, w, ind, rw);
        c_copy(c, rw, r);
    }
    if (!q.empty()) {
        CV<R> qw = tmp_cm<R>(c, m, q.cols);
        hipLaunchKernelGGL(k_c_form_q<R>, dim3((unsigned)q.cols), dim3(256), 0, c->stream, w, ind, tau, k, qw);
        c_copy(c, qw, q);
    }
}

// LAPACK granularity (see lapack_geqp3 / lapack_orgqr in rc_api.hip): ?geqp3 with the columns in pivoted order, ?ungqr
template <typename R>
void c_lapack_geqp3(rc_context *c, CV<R> a, int64_t kmax, int64_t *jpvt, cplx<R> *tau) {
    const int64_t m = a.rows, n = a.cols;
    RC_REQUIRE(kmax >= 0 && kmax <= std::min(m, n), RC_INVALID_ARGUMENT, "geqp3: need 0 <= kmax <= min(m, n)");
    RC_REQUIRE(jpvt != nullptr && (kmax == 0 || tau != nullptr), RC_INVALID_ARGUMENT, "geqp3: null jpvt / tau");
    if (n == 0) return;
    if (kmax == 0 || m == 0) { iota_i64(c, jpvt, n); return; }
    ArenaMark mark(c);
    CV<R> w = tmp_cm<R>(c, m, n);
    c_copy(c, a, w);
    c_geqp3(c, w, kmax, jpvt, tau);
    c_gather_cols(c, w, jpvt, a);
}
template <typename R>
void c_lapack_ungqr(rc_context *c, CV<R> a, const cplx<R> *tau, int64_t k, CV<R> q) {
    const int64_t m = a.rows;
    RC_REQUIRE(k >= 0 && k <= std::min(m, a.cols) && q.rows == m && q.cols == k, RC_INVALID_ARGUMENT,
               "orgqr: need k <= min(m, n) reflectors in a and q of m x k");
    if (k == 0 || m == 0) return;
    RC_REQUIRE(tau != nullptr, RC_INVALID_ARGUMENT, "orgqr: null tau");
    ArenaMark mark(c);
    int64_t *ident = c->alloc<int64_t>((size_t)k);
    iota_i64(c, ident, k);
    CV<R> w = tmp_cm<R>(c, m, k), qw = tmp_cm<R>(c, m, k);
    CV<R> ak = a; ak.cols = k;
    c_copy(c, ak, w);
    hipLaunchKernelGGL(k_c_form_q<R>, dim3((unsigned)k), dim3(256), 0, c->stream, w, ident, tau, k, qw);
    c_copy(c, qw, q);
}

// ------------------------------------------------------------------------------------------------ SVD (one-sided Jacobi)
__device__ inline void rr_pair_c(int N, int r, int pi, int &p, int &q) {  // circle method, round r, pair pi
    if (pi == 0) { p = N - 1; q = r; }
    else { p = (r + pi) % (N - 1); q = (r - pi + (N - 1)) % (N - 1); }
    if (p > q) { const int t = p; p = q; q = t; }
}
template <typename R>
__global__ __launch_bounds__(256) void k_c_jacobi_round(CV<R> g, CV<R> v, int r, int *state) {
    if (state[1]) return;
    const int n = (int)g.cols, N = (n + 1) & ~1;
    const int lane = threadIdx.x & 63;
    const int pi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pi >= N / 2) return;
    int p, q;
    rr_pair_c(N, r, pi, p, q);
    if (q >= n) return;
    const int64_t M = g.rows;
    cplx<R> *gp = g.p + (int64_t)p * g.cs, *gq = g.p + (int64_t)q * g.cs;
    R app = 0, aqq = 0;
    cplx<R> apq{0, 0};
    for (int64_t i = lane; i < M; i += 64) {
        const cplx<R> a = gp[i], b = gq[i];
        app += abs2(a);
        aqq += abs2(b);
        apq = apq + cj(a) * b;
    }
    app = wsumc(app);
    aqq = wsumc(aqq);
    apq.re = wsumc(apq.re);
    apq.im = wsumc(apq.im);
    const R habs = hypot(apq.re, apq.im);
    const R tol = sqrt((R)M) * NumC<R>::eps();
    if (habs == (R)0 || habs <= tol * sqrt(app) * sqrt(aqq)) return;
    // q~ = e^{-i phi} q makes p^H q~ = |apq| real: a real Jacobi rotation between p and q~.
    // The phase and the rotation parameters are evaluated in f64 for both precisions (as in the real kernels): in f32,
    // c^2 + s^2 - 1 and |phase|^2 - 1 have a systematic sign and the thousands of rotations a column undergoes inflate the
    // singular values by a few 1e-5 (the reference's c32 tests ask for 1e-5: src/svd.rs:291).
    const double hd = hypot((double)apq.re, (double)apq.im);
    const cplx<R> ph{(R)((double)apq.re / hd), (R)(-(double)apq.im / hd)};
    const double zeta = ((double)aqq - (double)app) / (2.0 * hd);
    const double td = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double cd = 1.0 / sqrt(1.0 + td * td);
    const R cs = (R)cd, sn = (R)(cd * td);
    for (int64_t i = lane; i < M; i += 64) {
        const cplx<R> a = gp[i], b = ph * gq[i];
        gp[i] = cs * a - sn * b;
        gq[i] = sn * a + cs * b;
    }
    cplx<R> *vp = v.p + (int64_t)p * v.cs, *vq = v.p + (int64_t)q * v.cs;
    for (int64_t i = lane; i < n; i += 64) {
        const cplx<R> a = vp[i], b = ph * vq[i];
        vp[i] = cs * a - sn * b;
        vq[i] = sn * a + cs * b;
    }
    if (lane == 0) state[0] = 1;
}
__global__ void k_c_jacobi_sweep_end(int *state) {
    if (threadIdx.x != 0 || state[1]) return;
    state[2] += 1;
    if (state[0] == 0) state[1] = 1;
    state[0] = 0;
}
template <typename R>
__global__ __launch_bounds__(256) void k_c_jacobi_norms(CV<R> g, R *sig) {
    const int lane = threadIdx.x & 63;
    const int64_t j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= g.cols) return;
    R acc = 0;
    for (int64_t i = lane; i < g.rows; i += 64) acc += abs2(g.at(i, j));
    acc = wsumc(acc);
    if (lane == 0) sig[j] = sqrt(acc);
}
template <typename R>
__global__ __launch_bounds__(256) void k_c_jacobi_rank(int n, const R *sig, int *order, R *s) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const R si = sig[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += (sig[j] > si || (sig[j] == si && j < i)) ? 1 : 0;
    order[i] = rank;
    s[rank] = si;
}
// uc(:, order[j]) = g(:, j) / sig[j], vc(:, order[j]) = v(:, j)
template <typename R>
__global__ __launch_bounds__(256) void k_c_jacobi_emit(CV<R> g, CV<R> v, const R *sig, const int *order, CV<R> uc, CV<R> vc) {
    const int lane = threadIdx.x & 63;
    const int64_t j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= g.cols) return;
    const int dst = order[j];
    const R sj = sig[j];
    const R inv = sj > (R)0 ? (R)1 / sj : (R)0;
    for (int64_t i = lane; i < g.rows; i += 64) uc.at(i, dst) = inv * g.at(i, j);
    for (int64_t i = lane; i < v.rows; i += 64) vc.at(i, dst) = v.at(i, j);
}
// thin SVD of the TALL column-major g (M x n, M >= n, destroyed): g = uc diag(s) vc^H
template <typename R>
void c_jacobi_svd_tall(rc_context *c, CV<R> g, CV<R> uc, R *s, CV<R> vc) {
    const int n = (int)g.cols, N = (n + 1) & ~1;
    ProfScope ps(c, "op:jacobi_svd<complex> %lldx%d", (long long)g.rows, n);
    ArenaMark mark(c);
    int *state = c->alloc<int>(4);
    R *sig = c->alloc<R>((size_t)n);
    int *order = c->alloc<int>((size_t)n);
    CV<R> v = tmp_cm<R>(c, n, n);
    fill_words(c, state, 4 * sizeof(int), 0u);
    c_fill(c, v, true);
    RC_REQUIRE(!c->capturing, RC_RUNTIME_ERROR, "complex SVD reads its convergence flag back: not capturable");
    const unsigned grid = (unsigned)((N / 2 + 3) / 4);
    bool converged = false;
    for (int sweep = 0; sweep < 60 && n > 1; ++sweep) {
        for (int r = 0; r < N - 1; ++r) hipLaunchKernelGGL(k_c_jacobi_round<R>, dim3(grid), dim3(256), 0, c->stream, g, v, r, state);
        hipLaunchKernelGGL(k_c_jacobi_sweep_end, dim3(1), dim3(64), 0, c->stream, state);
        int h[4] = {0, 0, 0, 0};
        RC_HIP(hipMemcpyAsync(h, state, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        RC_HIP(hipStreamSynchronize(c->stream));
        if (h[1]) { converged = true; break; }
    }
    RC_REQUIRE(converged || n <= 1, RC_LINALG_ERROR, "complex Jacobi SVD did not converge in 60 sweeps");
    hipLaunchKernelGGL(k_c_jacobi_norms<R>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, g, sig);
    hipLaunchKernelGGL(k_c_jacobi_rank<R>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, sig, order, s);
    hipLaunchKernelGGL(k_c_jacobi_emit<R>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, g, v, sig, order, uc, vc);
}
template <typename R>
void c_colinv(rc_context *c, CV<R> in, const int64_t *ind, CV<R> out);
// ComputeSVD::compute_svd (src/compute_svd.rs:18-27): a (m x n) = u diag(s) vt, u: m x r, vt: r x n, r = min(m, n)
template <typename R>
void c_compute_svd(rc_context *c, CV<R> a, CV<R> u, R *s, CV<R> vt) {
    const int64_t m = a.rows, n = a.cols, r = std::min(m, n);
    RC_REQUIRE(u.rows == m && u.cols == r && vt.rows == r && vt.cols == n, RC_INVALID_ARGUMENT, "compute_svd: output shapes");
    if (r == 0) return;
    ArenaMark mark(c);
    // Round 3: a clearly rectangular input goes through a QR factorization first (as ?gesdd does): the one-launch-per-round Jacobi
    // then works on the r x r triangle instead of on M-row columns (128 x 2048 c64: 23 -> see tools/bench_complex.py)
    const bool qr_first = std::max(m, n) >= 2 * r && r >= 2;
    if (m >= n) {
        if (qr_first) {
            CV<R> q = tmp_cm<R>(c, m, n), rr = tmp_cm<R>(c, n, n), rp = tmp_cm<R>(c, n, n), ur = tmp_cm<R>(c, n, n), vc = tmp_cm<R>(c, n, n);
            int64_t *ind = c->alloc<int64_t>((size_t)n);
            c_pivoted_qr(c, a, q, rr, ind, n);   // a P = q rr
            c_colinv(c, rr, ind, rp);            // rr P^H: a = q (rr P^H)
            c_jacobi_svd_tall(c, rp, ur, s, vc);
            c_gemm<R>(c, 0, 0, one<R>(), q, ur, zero<R>(), u);
            c_copy(c, vc.t(), vt, true);
        } else {
            CV<R> g = tmp_cm<R>(c, m, n), uc = tmp_cm<R>(c, m, n), vc = tmp_cm<R>(c, n, n);
            c_copy(c, a, g);
            c_jacobi_svd_tall(c, g, uc, s, vc);
            c_copy(c, uc, u);
            c_copy(c, vc.t(), vt, true);  // vt = vc^H
        }
    } else {
        // a^H = U' S V'^H  =>  a = V' S U'^H
        if (qr_first) {
            CV<R> ah = tmp_cm<R>(c, n, m), q = tmp_cm<R>(c, n, m), rr = tmp_cm<R>(c, m, m), rp = tmp_cm<R>(c, m, m), ur = tmp_cm<R>(c, m, m), vc = tmp_cm<R>(c, m, m);
            int64_t *ind = c->alloc<int64_t>((size_t)m);
            c_copy(c, a.t(), ah, true);
            c_pivoted_qr(c, ah, q, rr, ind, m);  // a^H P = q rr
            c_colinv(c, rr, ind, rp);
            c_jacobi_svd_tall(c, rp, ur, s, vc); // rr P^H = ur S vc^H  =>  a^H = (q ur) S vc^H  =>  a = vc S (q ur)^H
            CV<R> qu = tmp_cm<R>(c, n, m);
            c_gemm<R>(c, 0, 0, one<R>(), q, ur, zero<R>(), qu);
            c_copy(c, vc, u);
            c_copy(c, qu.t(), vt, true);
        } else {
            CV<R> g = tmp_cm<R>(c, n, m), uc = tmp_cm<R>(c, n, m), vc = tmp_cm<R>(c, m, m);
            c_copy(c, a.t(), g, true);
            c_jacobi_svd_tall(c, g, uc, s, vc);
            c_copy(c, vc, u);
            c_copy(c, uc.t(), vt, true);
        }
    }
}

// ------------------------------------------------------------------------------------------------ compositions
template <typename R>
void read_back_r(rc_context *c, const R *dev, R *host, size_t n) {
    RC_HIP(hipMemcpyAsync(host, dev, n * sizeof(R), hipMemcpyDeviceToHost, c->stream));
    RC_HIP(hipStreamSynchronize(c->stream));
}
// out = in with columns gathered by the inverse permutation: out[:, i] = in[:, inv[i]]
template <typename R>
void c_colinv(rc_context *c, CV<R> in, const int64_t *ind, CV<R> out) {
    int64_t *inv = c->alloc<int64_t>((size_t)std::max<int64_t>(in.cols, 1));
    invert_perm(c, ind, in.cols, inv);
    c_gather_cols(c, in, inv, out);
}
// QRTraits::column_id (src/qr.rs:270-309)
template <typename R>
void c_qr_column_id(rc_context *c, CV<R> q, CV<R> r, const int64_t *ind, CV<R> cm, CV<R> z) {
    const int64_t m = q.rows, k = q.cols, n = r.cols;
    RC_REQUIRE(r.rows == k && cm.rows == m && cm.cols == k && z.rows == k && z.cols == n && k <= n, RC_INVALID_ARGUMENT, "column_id: shapes");
    if (n == 0) return;
    ArenaMark mark(c);
    CV<R> zt = tmp_cm<R>(c, k, n);
    if (k == n) {
        c_gemm(c, 0, 0, one<R>(), q, r, zero<R>(), cm);
        c_fill(c, zt, true);
    } else {
        c_fill(c, zt.sub(0, k, 0, k), true);
        c_copy(c, r.sub(0, k, k, n - k), zt.sub(0, k, k, n - k));
        CV<R> r11 = tmp_cm<R>(c, k, k);
        c_copy(c, r.sub(0, k, 0, k), r11);
        hipLaunchKernelGGL(k_c_trsm_upper<R>, dim3((unsigned)cdivi(n - k, 256)), dim3(256), 0, c->stream, r11, zt.sub(0, k, k, n - k));
        c_gemm(c, 0, 0, one<R>(), q, r11, zero<R>(), cm);
    }
    CV<R> zo = tmp_cm<R>(c, k, n);
    c_colinv(c, zt, ind, zo);
    c_copy(c, zo, z);
}
// LQTraits::row_id (src/qr.rs:363-403) = the adjoint of the column ID of (Q^H, L^H)
template <typename R>
void c_lq_row_id(rc_context *c, CV<R> l, CV<R> q, const int64_t *ind, CV<R> x, CV<R> rrows) {
    const int64_t m = l.rows, k = l.cols, n = q.cols;
    ArenaMark mark(c);
    CV<R> qh = tmp_cm<R>(c, n, k), lh = tmp_cm<R>(c, k, m), cp = tmp_cm<R>(c, n, k), zp = tmp_cm<R>(c, k, m);
    c_copy(c, q.t(), qh, true);
    c_copy(c, l.t(), lh, true);
    c_qr_column_id(c, qh, lh, ind, cp, zp);
    c_copy(c, zp.t(), x, true);      // X = Z'^H   (m x k)
    c_copy(c, cp.t(), rrows, true);  // R = C'^H   (k x n)
}
// P A = L Q: pivoted QR of A^H, conjugate-transposed back (src/pivoted_qr.rs:32-41)
template <typename R>
void c_pivoted_lq(rc_context *c, CV<R> a, CV<R> l, CV<R> q, int64_t *ind, int64_t k) {
    const int64_t m = a.rows, n = a.cols;
    ArenaMark mark(c);
    CV<R> ah = tmp_cm<R>(c, n, m), qp = tmp_cm<R>(c, n, k), rp = tmp_cm<R>(c, k, m);
    c_copy(c, a.t(), ah, true);
    c_pivoted_qr(c, ah, qp, rp, ind, k);
    c_copy(c, rp.t(), l, true);
    c_copy(c, qp.t(), q, true);
}
template <typename R>
void c_gaussian(rc_context *c, CV<R> out, uint64_t seed, uint64_t offset) {
    // element (i, j): re = normal number 2 (i cols + j), im = the next one (src/random_matrix.rs:136-143), row-major order
    if (out.empty()) return;
    ArenaMark mark(c);
    cplx<R> *tmp = c->alloc<cplx<R>>((size_t)out.rows * out.cols);
    fill_gaussian<R>(c, Mat<R>(reinterpret_cast<R *>(tmp), out.rows, 2 * out.cols, 2 * out.cols, 1), seed, 2 * offset);
    c_copy(c, CV<R>{tmp, out.rows, out.cols, out.cols, 1}, out);
}
template <typename R>
void c_max_col_norm_dev(rc_context *c, CV<R> y, R *out_dev) {
    ArenaMark mark(c);
    CV<R> w = tmp_cm<R>(c, y.rows, y.cols);
    c_copy(c, y, w);
    R *ss = c->alloc<R>((size_t)std::max<int64_t>(y.cols, 1));
    hipLaunchKernelGGL(k_c_col_sumsq<R>, dim3((unsigned)std::min<int64_t>(cdivi(std::max<int64_t>(y.cols, 1), 4), 8192)), dim3(256), 0, c->stream, w, ss);
    max_sqrt<R>(c, ss, y.cols, out_dev);
}
// SampleRange::sample_range_by_rank (src/random_sampling.rs:103-118)
// The operator the complex range finders sample (the real twin is OpView in rc_api.hip): a dense device matrix or the host's
// callback table (rc_operator; the views handed over are interleaved-complex rc_matrix descriptors, strides in complex elements).
// conj_matmat is A^H x (src/types.rs:128-132); the projection B = Q^H A is the conjugate transpose of the callback's A^H Q
// (the reference: conj_matmat(..).t().map(conj), src/qr.rs:316-317), one conjugating copy.
template <typename R>
struct COp {
    int64_t rows = 0, cols = 0;
    CV<R> dense{nullptr, 0, 0, 0, 0};
    const rc_operator *cb = nullptr;
    static COp of(CV<R> a) { COp o; o.rows = a.rows; o.cols = a.cols; o.dense = a; return o; }
    static COp of(const rc_operator *op) {
        RC_REQUIRE(op != nullptr && op->matmat != nullptr && op->rows >= 0 && op->cols >= 0, RC_INVALID_ARGUMENT, "rc_operator: null table / matmat or negative extent");
        COp o; o.rows = op->rows; o.cols = op->cols; o.cb = op; return o;
    }
    static rc_matrix to_c(CV<R> m) { rc_matrix r; r.data = m.p; r.rows = m.rows; r.cols = m.cols; r.row_stride = m.rs; r.col_stride = m.cs; return r; }
    void call(rc_context *c, rc_operator_product_fn fn, const char *what, CV<R> x, CV<R> y) const {
        RC_REQUIRE(fn != nullptr, RC_INVALID_ARGUMENT, "rc_operator: this call needs the operator's %s", what);
        RC_REQUIRE(!c->capturing, RC_INVALID_ARGUMENT, "rc_operator: callbacks cannot be recorded into a hipGraph");
        const rc_status st = fn(cb->user, c, to_c(x), to_c(y));
        if (st != RC_OK) fail(st, "operator callback %s (%lld x %lld -> %lld x %lld) returned status %d", what, (long long)x.rows, (long long)x.cols, (long long)y.rows, (long long)y.cols, (int)st);
    }
    void matmat(rc_context *c, CV<R> x, CV<R> y) const {
        RC_REQUIRE(x.rows == cols && y.rows == rows && x.cols == y.cols, RC_INVALID_ARGUMENT, "matmat: shape mismatch");
        if (cb) call(c, cb->matmat, "matmat", x, y);
        else c_gemm(c, 0, 0, one<R>(), dense, x, zero<R>(), y);
    }
    void conj_matmat(rc_context *c, CV<R> x, CV<R> y) const {
        RC_REQUIRE(x.rows == rows && y.rows == cols && x.cols == y.cols, RC_INVALID_ARGUMENT, "conj_matmat: shape mismatch");
        if (cb) call(c, cb->conj_matmat, "conj_matmat", x, y);
        else c_gemm(c, 2, 0, one<R>(), dense, x, zero<R>(), y);
    }
    // b (k x n) = range^H A
    void project(rc_context *c, CV<R> range, CV<R> b) const {
        if (!cb) { c_gemm(c, 2, 0, one<R>(), range, dense, zero<R>(), b); return; }
        ArenaMark mark(c);
        CV<R> t = tmp_cm<R>(c, cols, range.cols);
        conj_matmat(c, range, t);
        c_copy(c, t.t(), b, true);
    }
};

template <typename R>
void c_sample_range_by_rank(rc_context *c, const COp<R> &a, int64_t k, int64_t p, CV<R> omega, uint64_t seed, CV<R> q) {
    const int64_t m = a.rows, n = a.cols, l = k + p;
    RC_REQUIRE(k >= 0 && p >= 0, RC_INVALID_ARGUMENT, "sample_range_by_rank: negative k or p");
    const int64_t kk = std::min(k, std::min(m, l));
    RC_REQUIRE(q.rows == m && q.cols == kk, RC_INVALID_ARGUMENT, "sample_range_by_rank: q must be %lld x %lld", (long long)m, (long long)kk);
    if (kk == 0) return;
    ArenaMark mark(c);
    if (omega.p == nullptr) {
        omega = tmp_cm<R>(c, n, l);
        c_gaussian(c, omega, seed, 0);
    } else {
        RC_REQUIRE(omega.rows == n && omega.cols == l, RC_INVALID_ARGUMENT, "sample_range_by_rank: omega must be %lld x %lld", (long long)n, (long long)l);
    }
    CV<R> y = tmp_cm<R>(c, m, l);
    a.matmat(c, omega, y);
    int64_t *ind = c->alloc<int64_t>((size_t)l);
    c_pivoted_qr(c, y, q, CV<R>{nullptr, 0, 0, 0, 0}, ind, kk);
}
template <typename R>
CV<R> c_orth_full(rc_context *c, CV<R> w) {
    const int64_t k = std::min(w.rows, w.cols);
    CV<R> q = tmp_cm<R>(c, w.rows, k);
    int64_t *ind = c->alloc<int64_t>((size_t)std::max<int64_t>(w.cols, 1));
    c_pivoted_qr(c, w, q, CV<R>{nullptr, 0, 0, 0, 0}, ind, k);
    return q;
}
// SampleRangePowerIteration (src/random_sampling.rs:131-160) with the reference's single surviving step
template <typename R>
void c_sample_range_power(rc_context *c, const COp<R> &a, int64_t k, int64_t p, int64_t it_count, CV<R> omega, uint64_t seed, CV<R> q) {
    if (it_count <= 0) { c_sample_range_by_rank(c, a, k, p, omega, seed, q); return; }
    const int64_t m = a.rows, n = a.cols, l = k + p;
    ArenaMark mark(c);
    if (omega.p == nullptr) {
        omega = tmp_cm<R>(c, n, l);
        c_gaussian(c, omega, seed, 0);
    } else {
        RC_REQUIRE(omega.rows == n && omega.cols == l, RC_INVALID_ARGUMENT, "sample_range_power_iteration: omega must be %lld x %lld", (long long)n, (long long)l);
    }
    CV<R> y1 = tmp_cm<R>(c, m, l);
    a.matmat(c, omega, y1);
    const int64_t steps = c->opt_power_fixed ? it_count : 1;
    for (int64_t it = 0; it < steps; ++it) {
        CV<R> q0 = c_orth_full(c, y1);
        CV<R> z = tmp_cm<R>(c, n, q0.cols);
        a.conj_matmat(c, q0, z);
        CV<R> wq = c_orth_full(c, z);
        y1 = tmp_cm<R>(c, m, wq.cols);
        a.matmat(c, wq, y1);
    }
    const int64_t kk = std::min(k, std::min(m, y1.cols));
    RC_REQUIRE(q.rows == m && q.cols == kk, RC_INVALID_ARGUMENT, "sample_range_power_iteration: q must be %lld x %lld", (long long)m, (long long)kk);
    int64_t *ind = c->alloc<int64_t>((size_t)std::max<int64_t>(y1.cols, 1));
    c_pivoted_qr(c, y1, q, CV<R>{nullptr, 0, 0, 0, 0}, ind, kk);
}
// AdaptiveSampling::sample_range_adaptive (src/random_sampling.rs:223-274)
template <typename R>
void c_sample_range_adaptive(rc_context *c, const COp<R> &a, double rel_tol_d, int64_t s, CV<R> omegas, uint64_t seed, CV<R> qcap, int64_t *rank_out, int64_t *hist_rank,
                             double *hist_res, int64_t hist_cap, int64_t *hist_len) {
    const int64_t m = a.rows, n = a.cols, cap = qcap.cols;
    RC_REQUIRE(s >= 1 && qcap.rows == m, RC_INVALID_ARGUMENT, "sample_range_adaptive: bad sample_size or q buffer");
    const bool explicit_omega = omegas.p != nullptr;
    if (explicit_omega) RC_REQUIRE(omegas.rows == n, RC_INVALID_ARGUMENT, "sample_range_adaptive: omegas must have %lld rows", (long long)n);
    const R tol_factor = (R)(10.0 * std::sqrt(2.0 / 3.14159265358979323846));
    const R rel_tol = (R)rel_tol_d;
    const int64_t sq = std::min(m, s);
    int64_t blocks_used = 0;
    CV<R> omega = tmp_cm<R>(c, n, s), y = tmp_cm<R>(c, m, s), qacc = tmp_cm<R>(c, m, cap), bacc = tmp_cm<R>(c, cap, n), t1 = tmp_cm<R>(c, cap, s);
    int64_t *ind = c->alloc<int64_t>((size_t)s);
    R *scal = c->alloc<R>(1);
    auto next_omega = [&]() {
        if (explicit_omega) {
            RC_REQUIRE((blocks_used + 1) * s <= omegas.cols, RC_COMPRESSION_ERROR, "sample_range_adaptive: explicit Omega blocks exhausted after %lld blocks", (long long)blocks_used);
            c_copy(c, omegas.sub(0, n, blocks_used * s, s), omega);
        } else {
            c_gaussian(c, omega, seed, (uint64_t)blocks_used * (uint64_t)(n * s));
        }
        ++blocks_used;
    };
    next_omega();
    a.matmat(c, omega, y);
    R mc;
    c_max_col_norm_dev(c, y, scal);
    read_back_r(c, scal, &mc, 1);
    const R operator_norm = mc * tol_factor;
    R max_norm = operator_norm;
    int64_t r = 0, nh = 0;
    const cplx<R> minus_one{(R)-1, (R)0};
    while (max_norm / operator_norm >= rel_tol) {
        RC_REQUIRE(r + sq <= cap, RC_COMPRESSION_ERROR, "sample_range_adaptive: basis capacity %lld exhausted at rank %lld", (long long)cap, (long long)r);
        if (r > 0) {  // y -= q (q^H y)
            CV<R> qr_ = qacc.sub(0, m, 0, r), tt = t1.sub(0, r, 0, s);
            c_gemm(c, 2, 0, one<R>(), qr_, y, zero<R>(), tt);
            c_gemm(c, 0, 0, minus_one, qr_, tt, one<R>(), y);
        }
        CV<R> qnew = qacc.sub(0, m, r, sq);
        c_pivoted_qr(c, y, qnew, CV<R>{nullptr, 0, 0, 0, 0}, ind, sq);
        a.project(c, qnew, bacc.sub(r, sq, 0, n));  // b = [b ; (A^H Q_new)^H] = Q_new^H A
        r += sq;
        next_omega();
        {
            CV<R> qr_ = qacc.sub(0, m, 0, r), tt = t1.sub(0, r, 0, s);
            c_gemm(c, 0, 0, one<R>(), bacc.sub(0, r, 0, n), omega, zero<R>(), tt);
            a.matmat(c, omega, y);
            c_gemm(c, 0, 0, minus_one, qr_, tt, one<R>(), y);
        }
        c_max_col_norm_dev(c, y, scal);
        read_back_r(c, scal, &mc, 1);
        max_norm = mc * tol_factor;
        if (nh < hist_cap) {
            if (hist_rank) hist_rank[nh] = r;
            if (hist_res) hist_res[nh] = (double)(max_norm / operator_norm);
        }
        ++nh;
    }
    c_copy(c, qacc.sub(0, m, 0, r), qcap.sub(0, m, 0, r));
    if (rank_out) *rank_out = r;
    if (hist_len) *hist_len = std::min(nh, hist_cap);
    RC_HIP(hipStreamSynchronize(c->stream));
}

template <typename F>
rc_status guarded_c(rc_context *ctx, F &&f) {
    if (!ctx) return RC_INVALID_ARGUMENT;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != ctx->device) (void)hipSetDevice(ctx->device);
    rc_status st = RC_OK;
    // (calls nest: an operator callback may use the library on the same context -- only the outermost call resets the arena)
    struct Scope {
        rc_context *c; size_t off = 0; bool outer;
        explicit Scope(rc_context *x) : c(x), outer(x->call_depth++ == 0) { if (outer) c->reset_arena(); else off = c->arena_off; }
        ~Scope() { --c->call_depth; if (!outer) c->arena_off = off; }
    };
    try {
        Scope scope(ctx);
        f();
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) fail(RC_RUNTIME_ERROR, "kernel launch failed: %s", hipGetErrorString(e));
    } catch (const Error &e) {
        ctx->last_error = e.msg;
        st = e.code;
    } catch (const std::exception &e) {
        ctx->last_error = e.what();
        st = RC_RUNTIME_ERROR;
    }
    if (prev != ctx->device && prev >= 0) (void)hipSetDevice(prev);
    return st;
}

template <typename R>
void c_rank_by_tolerance(rc_context *c, CV<R> tri, double tol, int64_t *rank) {
    RC_REQUIRE(tol < 1.0 && 0.0 <= tol, RC_INVALID_ARGUMENT, "Require 0 <= tol < 1.0");
    const int64_t len = std::min(tri.rows, tri.cols);
    RC_REQUIRE(len >= 1, RC_COMPRESSION_ERROR, "rank_by_tolerance: empty factor");
    ArenaMark mark(c);
    cplx<R> *d = c->alloc<cplx<R>>((size_t)len);
    c_copy(c, CV<R>{tri.p, len, 1, tri.rs + tri.cs, 1}, CV<R>{d, len, 1, 1, len});
    std::vector<cplx<R>> h((size_t)len);
    RC_HIP(hipMemcpyAsync(h.data(), d, (size_t)len * sizeof(cplx<R>), hipMemcpyDeviceToHost, c->stream));
    RC_HIP(hipStreamSynchronize(c->stream));
    const double d0 = std::hypot((double)h[0].re, (double)h[0].im);
    for (int64_t i = 0; i < len; ++i)
        if (std::hypot((double)h[i].re, (double)h[i].im) / d0 < tol) { *rank = i; return; }  // qr.rs:194
    fail(RC_COMPRESSION_ERROR, "Could not compress to desired tolerance");
}

template <typename R>
void c_apply_perm(rc_context *c, int mode, CV<R> in, const int64_t *perm, int64_t plen, CV<R> out) {
    RC_REQUIRE(in.rows == out.rows && in.cols == out.cols, RC_INVALID_ARGUMENT, "apply_permutation: shape mismatch");
    RC_REQUIRE(mode >= 0 && mode <= 3, RC_INVALID_ARGUMENT, "apply_permutation: unknown mode %d", mode);
    ArenaMark mark(c);
    const bool cols = (mode == RC_PERM_COL || mode == RC_PERM_COLINV), inv = (mode == RC_PERM_COLINV || mode == RC_PERM_ROWINV);
    if (cols) RC_REQUIRE(plen == in.cols, RC_INVALID_ARGUMENT, "Length of index array and number of columns differ.");
    else RC_REQUIRE(plen == in.rows, RC_INVALID_ARGUMENT, "Length of index array and number of rows differ.");
    const int64_t *idx = perm;
    if (inv) {
        int64_t *iv = c->alloc<int64_t>((size_t)std::max<int64_t>(plen, 1));
        invert_perm(c, perm, plen, iv);
        idx = iv;
    }
    if (cols) c_gather_cols(c, in, idx, out);
    else c_gather_cols(c, in.t(), idx, out.t());
}

static rc_status c_svd_rank_fwd(rc_context *ctx, const double *s, int64_t len, double tol, int64_t *rank) { return rc_svd_rank_by_tolerance_f64(ctx, s, len, tol, rank); }
static rc_status c_svd_rank_fwd(rc_context *ctx, const float *s, int64_t len, double tol, int64_t *rank) { return rc_svd_rank_by_tolerance_f32(ctx, s, len, tol, rank); }

// the cfg3 pipeline for complex scalars (call sequence of rsvd_id in rc_api.hip): range -> B = Q^H A once -> SVD(B), QRCP(B), column ID
template <typename R>
void c_rsvd_id(rc_context *c, CV<R> a, int64_t k, int64_t p, CV<R> omega, uint64_t seed, const rc_rsvd_id_out &o) {
    const int64_t m = a.rows, n = a.cols;
    RC_REQUIRE(k >= 1 && k + p <= m && k <= n, RC_INVALID_ARGUMENT, "rsvd_id: need 1 <= k, k + p <= m, k <= n");
    ArenaMark mark(c);
    CV<R> range = view_of<R>(o.range_q);
    if (range.p == nullptr) range = tmp_cm<R>(c, m, k);
    RC_REQUIRE(range.rows == m && range.cols == k, RC_INVALID_ARGUMENT, "rsvd_id: range_q must be m x k");
    c_sample_range_by_rank<R>(c, COp<R>::of(a), k, p, omega, seed, range);
    CV<R> b = tmp_cm<R>(c, k, n);
    c_gemm<R>(c, 2, 0, one<R>(), range, a, zero<R>(), b);
    const bool want_id = o.id_c.data || o.id_z.data || o.qr_q.data || o.qr_r.data || o.qr_ind;
    const bool want_svd = o.u.data || o.s || o.vt.data;
    if (want_svd) RC_REQUIRE(o.u.data && o.s && o.vt.data, RC_INVALID_ARGUMENT, "rsvd_id: u, s, vt must be given together");
    if (want_id) {
        CV<R> qb = tmp_cm<R>(c, k, k);
        CV<R> r = o.qr_r.data ? view_of<R>(o.qr_r) : tmp_cm<R>(c, k, n);
        int64_t *ind = o.qr_ind ? o.qr_ind : c->alloc<int64_t>((size_t)n);
        c_pivoted_qr<R>(c, b, qb, r, ind, k);  // works on its own copy of b
        CV<R> q = o.qr_q.data ? view_of<R>(o.qr_q) : tmp_cm<R>(c, m, k);
        c_gemm<R>(c, 0, 0, one<R>(), range, qb, zero<R>(), q);
        if (o.id_c.data || o.id_z.data) {
            RC_REQUIRE(o.id_c.data && o.id_z.data, RC_INVALID_ARGUMENT, "rsvd_id: id_c and id_z must be given together");
            c_qr_column_id<R>(c, q, r, ind, view_of<R>(o.id_c), view_of<R>(o.id_z));
        }
    }
    if (want_svd) {
        CV<R> ub = tmp_cm<R>(c, k, k);
        c_compute_svd<R>(c, b, ub, static_cast<R *>(o.s), view_of<R>(o.vt));
        c_gemm<R>(c, 0, 0, one<R>(), range, ub, zero<R>(), view_of<R>(o.u));
    }
}

// rank-k column IDs of `count` same-shaped complex matrices into the packed buffer (layout of rc_batch_packed_bytes with
// elem_size = sizeof(complex)); matrix i runs on context i % nctx, every context is waited for at the end
template <typename R>
void c_batch_column_id(rc_context *const *ctxs, int nctx, const rc_matrix *mats, int count, int64_t k, void *packed) {
    RC_REQUIRE(count >= 0 && k >= 1, RC_INVALID_ARGUMENT, "batch_column_id: k >= 1");
    if (count == 0) return;
    RC_REQUIRE(mats != nullptr && packed != nullptr, RC_INVALID_ARGUMENT, "batch_column_id: null argument");
    const int64_t m = mats[0].rows, n = mats[0].cols;
    RC_REQUIRE(k <= std::min(m, n), RC_INVALID_ARGUMENT, "batch_column_id: rank exceeds min(m, n)");
    const size_t per = rc_batch_packed_bytes(m, n, k, (int32_t)sizeof(cplx<R>));
    for (int i = 0; i < count; ++i) {
        RC_REQUIRE(mats[i].rows == m && mats[i].cols == n && mats[i].data, RC_INVALID_ARGUMENT, "batch_column_id: all matrices must have the shape of the first");
        rc_context *c = ctxs[i % nctx];
        RC_REQUIRE(c != nullptr, RC_INVALID_ARGUMENT, "batch_column_id: bad context");
        char *base = static_cast<char *>(packed) + (size_t)i * per;
        cplx<R> *cz = reinterpret_cast<cplx<R> *>(base);
        int64_t *ind = reinterpret_cast<int64_t *>(base + per - (size_t)n * sizeof(int64_t));
        CV<R> A = view_of<R>(mats[i]);
        CV<R> cm{cz, m, k, k, 1}, z{cz + (size_t)m * k, k, n, n, 1};
        c->reset_arena();
        ArenaMark mark(c);
        CV<R> q = tmp_cm<R>(c, m, k), r = tmp_cm<R>(c, k, n);
        c_pivoted_qr<R>(c, A, q, r, ind, k);
        c_qr_column_id<R>(c, q, r, ind, cm, z);
    }
    for (int l = 0; l < std::min(nctx, count); ++l) RC_HIP(hipStreamSynchronize(ctxs[l]->stream));
}

}  // namespace

// ================================================================================================ extern "C"
extern "C" {

#define RC_DEFINE_COMPLEX(SUF, R, CT)                                                                                                     \
    rc_status rc_random_gaussian_##SUF(rc_context *ctx, rc_matrix out, uint64_t seed, uint64_t offset) {                                  \
        return guarded_c(ctx, [&] { c_gaussian<R>(ctx, view_of<R>(out), seed, offset); });                                                \
    }                                                                                                                                     \
    rc_status rc_matmat_##SUF(rc_context *ctx, rc_matrix a, rc_matrix x, rc_matrix y) {                                                   \
        return guarded_c(ctx, [&] { c_gemm<R>(ctx, 0, 0, one<R>(), view_of<R>(a), view_of<R>(x), zero<R>(), view_of<R>(y)); });           \
    }                                                                                                                                     \
    rc_status rc_conj_matmat_##SUF(rc_context *ctx, rc_matrix a, rc_matrix x, rc_matrix y) {                                              \
        return guarded_c(ctx, [&] { c_gemm<R>(ctx, 2, 0, one<R>(), view_of<R>(a), view_of<R>(x), zero<R>(), view_of<R>(y)); });           \
    }                                                                                                                                     \
    rc_status rc_gemm_##SUF(rc_context *ctx, int32_t trans_a, int32_t trans_b, CT alpha, rc_matrix a, rc_matrix b, CT beta, rc_matrix c) { \
        return guarded_c(ctx, [&] {                                                                                                       \
            RC_REQUIRE(trans_a >= 0 && trans_a <= 2 && trans_b >= 0 && trans_b <= 2, RC_INVALID_ARGUMENT, "gemm: op must be 0, 1 or 2");  \
            c_gemm<R>(ctx, trans_a, trans_b, cplx<R>{alpha.re, alpha.im}, view_of<R>(a), view_of<R>(b), cplx<R>{beta.re, beta.im}, view_of<R>(c)); \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_rel_diff_fro_##SUF(rc_context *ctx, rc_matrix first, rc_matrix second, R *out) {                                         \
        return guarded_c(ctx, [&] {                                                                                                       \
            CV<R> A = view_of<R>(first), B = view_of<R>(second);                                                                          \
            RC_REQUIRE(A.rows == B.rows && A.cols == B.cols, RC_INVALID_ARGUMENT, "rel_diff_fro: shape mismatch");                        \
            R *d = ctx->alloc<R>(2);                                                                                                      \
            hipLaunchKernelGGL(k_c_fro<R>, dim3(1), dim3(1024), 0, ctx->stream, A, B, d);                                                 \
            R h[2];                                                                                                                       \
            read_back_r<R>(ctx, d, h, 2);                                                                                                 \
            *out = std::sqrt(h[0]) / std::sqrt(h[1]);                                                                                     \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_apply_permutation_matrix_##SUF(rc_context *ctx, int32_t mode, rc_matrix in, const int64_t *perm, int64_t plen, rc_matrix out) { \
        return guarded_c(ctx, [&] { c_apply_perm<R>(ctx, mode, view_of<R>(in), perm, plen, view_of<R>(out)); });                          \
    }                                                                                                                                     \
    rc_status rc_apply_permutation_vector_##SUF(rc_context *ctx, int32_t mode, rc_matrix in, const int64_t *perm, int64_t plen, rc_matrix out) { \
        return guarded_c(ctx, [&] {                                                                                                       \
            RC_REQUIRE(mode == RC_VPERM_INV || mode == RC_VPERM_NOINV, RC_INVALID_ARGUMENT, "unknown vector permutation mode");           \
            RC_REQUIRE(in.cols == 1 && out.cols == 1 && plen == in.rows, RC_INVALID_ARGUMENT, "The input vector and the index array must have the same length"); \
            c_apply_perm<R>(ctx, mode == RC_VPERM_INV ? RC_PERM_ROWINV : RC_PERM_ROW, view_of<R>(in), perm, plen, view_of<R>(out));       \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_pivoted_qr_##SUF(rc_context *ctx, rc_matrix a, rc_matrix q, rc_matrix r, int64_t *ind) {                                 \
        return guarded_c(ctx, [&] {                                                                                                       \
            CV<R> A = view_of<R>(a), Q = view_of<R>(q), Rr = view_of<R>(r);                                                               \
            RC_REQUIRE(Q.rows == A.rows && Rr.cols == A.cols && Rr.rows == Q.cols, RC_INVALID_ARGUMENT, "pivoted_qr: output shapes");     \
            c_pivoted_qr<R>(ctx, A, Q, Rr, ind, Q.cols);                                                                                  \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_geqp3_##SUF(rc_context *ctx, rc_matrix a, int64_t kmax, int64_t *jpvt, CT *tau) {                                        \
        return guarded_c(ctx, [&] { c_lapack_geqp3<R>(ctx, view_of<R>(a), kmax, jpvt, reinterpret_cast<cplx<R> *>(tau)); });             \
    }                                                                                                                                     \
    rc_status rc_orgqr_##SUF(rc_context *ctx, rc_matrix a, const CT *tau, int64_t k, rc_matrix q) {                                       \
        return guarded_c(ctx, [&] { c_lapack_ungqr<R>(ctx, view_of<R>(a), reinterpret_cast<const cplx<R> *>(tau), k, view_of<R>(q)); }); \
    }                                                                                                                                     \
    rc_status rc_trsm_upper_##SUF(rc_context *ctx, rc_matrix t, rc_matrix b) {                                                            \
        return guarded_c(ctx, [&] {                                                                                                       \
            CV<R> tt = view_of<R>(t), bb = view_of<R>(b);                                                                                 \
            RC_REQUIRE(tt.rows == tt.cols && tt.rows == bb.rows, RC_INVALID_ARGUMENT, "trsm_upper: t must be k x k, b k x nrhs");        \
            if (bb.empty()) return;                                                                                                       \
            hipLaunchKernelGGL(k_c_trsm_upper<R>, dim3((unsigned)cdivi(bb.cols, 256)), dim3(256), 0, ctx->stream, tt, bb);                \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_pivoted_lq_##SUF(rc_context *ctx, rc_matrix a, rc_matrix l, rc_matrix q, int64_t *ind) {                                 \
        return guarded_c(ctx, [&] {                                                                                                       \
            CV<R> A = view_of<R>(a), L = view_of<R>(l), Q = view_of<R>(q);                                                                \
            RC_REQUIRE(L.rows == A.rows && Q.cols == A.cols && Q.rows == L.cols, RC_INVALID_ARGUMENT, "pivoted_lq: output shapes");       \
            c_pivoted_lq<R>(ctx, A, L, Q, ind, L.cols);                                                                                   \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_compute_svd_##SUF(rc_context *ctx, rc_matrix a, rc_matrix u, R *s, rc_matrix vt) {                                       \
        return guarded_c(ctx, [&] { c_compute_svd<R>(ctx, view_of<R>(a), view_of<R>(u), s, view_of<R>(vt)); });                           \
    }                                                                                                                                     \
    rc_status rc_rank_by_tolerance_##SUF(rc_context *ctx, rc_matrix tri, double tol, int64_t *rank) {                                     \
        return guarded_c(ctx, [&] { c_rank_by_tolerance<R>(ctx, view_of<R>(tri), tol, rank); });                                          \
    }                                                                                                                                     \
    rc_status rc_qr_to_mat_##SUF(rc_context *ctx, rc_matrix q, rc_matrix r, const int64_t *ind, rc_matrix out) {                          \
        return guarded_c(ctx, [&] {                                                                                                       \
            CV<R> Rr = view_of<R>(r);                                                                                                     \
            CV<R> rp = tmp_cm<R>(ctx, Rr.rows, Rr.cols);                                                                                  \
            c_colinv<R>(ctx, Rr, ind, rp);                                                                                                \
            c_gemm<R>(ctx, 0, 0, one<R>(), view_of<R>(q), rp, zero<R>(), view_of<R>(out));                                                \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_lq_to_mat_##SUF(rc_context *ctx, rc_matrix l, rc_matrix q, const int64_t *ind, rc_matrix out) {                          \
        return guarded_c(ctx, [&] {                                                                                                       \
            CV<R> L = view_of<R>(l);                                                                                                      \
            CV<R> lp = tmp_cm<R>(ctx, L.rows, L.cols);                                                                                    \
            c_colinv<R>(ctx, L.t(), ind, lp.t());                                                                                         \
            c_gemm<R>(ctx, 0, 0, one<R>(), lp, view_of<R>(q), zero<R>(), view_of<R>(out));                                                \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_qr_column_id_##SUF(rc_context *ctx, rc_matrix q, rc_matrix r, const int64_t *ind, rc_matrix c, rc_matrix z) {            \
        return guarded_c(ctx, [&] { c_qr_column_id<R>(ctx, view_of<R>(q), view_of<R>(r), ind, view_of<R>(c), view_of<R>(z)); });          \
    }                                                                                                                                     \
    rc_status rc_lq_row_id_##SUF(rc_context *ctx, rc_matrix l, rc_matrix q, const int64_t *ind, rc_matrix x, rc_matrix rr) {              \
        return guarded_c(ctx, [&] { c_lq_row_id<R>(ctx, view_of<R>(l), view_of<R>(q), ind, view_of<R>(x), view_of<R>(rr)); });            \
    }                                                                                                                                     \
    rc_status rc_qr_from_range_estimate_##SUF(rc_context *ctx, rc_matrix range, rc_matrix a, rc_matrix q, rc_matrix r, int64_t *ind) {    \
        return guarded_c(ctx, [&] {                                                                                                       \
            CV<R> Rg = view_of<R>(range), A = view_of<R>(a), Q = view_of<R>(q), Rr = view_of<R>(r);                                       \
            const int64_t rr = Rg.cols, n = A.cols, k = std::min(rr, n);                                                                  \
            RC_REQUIRE(Rg.rows == A.rows && Q.rows == A.rows && Q.cols == k && Rr.rows == k && Rr.cols == n, RC_INVALID_ARGUMENT, "qr_from_range_estimate: shape mismatch"); \
            CV<R> b = tmp_cm<R>(ctx, rr, n), qb = tmp_cm<R>(ctx, rr, k);                                                                  \
            c_gemm<R>(ctx, 2, 0, one<R>(), Rg, A, zero<R>(), b);                                                                          \
            c_pivoted_qr<R>(ctx, b, qb, Rr, ind, k);                                                                                      \
            c_gemm<R>(ctx, 0, 0, one<R>(), Rg, qb, zero<R>(), Q);                                                                         \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_svd_to_mat_##SUF(rc_context *ctx, rc_matrix u, const R *s, rc_matrix vt, rc_matrix out) {                                \
        return guarded_c(ctx, [&] {                                                                                                       \
            CV<R> VT = view_of<R>(vt);                                                                                                    \
            CV<R> sv = tmp_cm<R>(ctx, VT.rows, VT.cols);                                                                                  \
            c_copy<R>(ctx, VT, sv);                                                                                                       \
            if (!sv.empty()) hipLaunchKernelGGL(k_c_scale_rows<R>, dim3((unsigned)std::min<int64_t>(cdivi(sv.rows * sv.cols, 256), 8192)), dim3(256), 0, ctx->stream, s, sv); \
            c_gemm<R>(ctx, 0, 0, one<R>(), view_of<R>(u), sv, zero<R>(), view_of<R>(out));                                                \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_svd_to_qr_##SUF(rc_context *ctx, rc_matrix u, const R *s, rc_matrix vt, rc_matrix q, rc_matrix r, int64_t *ind) {        \
        return guarded_c(ctx, [&] {                                                                                                       \
            CV<R> VT = view_of<R>(vt), Q = view_of<R>(q);                                                                                 \
            const int64_t k = Q.cols;                                                                                                     \
            RC_REQUIRE(k <= std::min(VT.rows, VT.cols), RC_INVALID_ARGUMENT, "svd_to_qr: rank exceeds min(r, n)");                        \
            CV<R> w = tmp_cm<R>(ctx, VT.rows, VT.cols), qb = tmp_cm<R>(ctx, VT.rows, k);                                                  \
            c_copy<R>(ctx, VT, w);                                                                                                        \
            if (!w.empty()) hipLaunchKernelGGL(k_c_scale_rows<R>, dim3((unsigned)std::min<int64_t>(cdivi(w.rows * w.cols, 256), 8192)), dim3(256), 0, ctx->stream, s, w); \
            c_pivoted_qr<R>(ctx, w, qb, view_of<R>(r), ind, k);                                                                           \
            c_gemm<R>(ctx, 0, 0, one<R>(), view_of<R>(u), qb, zero<R>(), Q);                                                              \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_svd_from_range_estimate_##SUF(rc_context *ctx, rc_matrix range, rc_matrix a, rc_matrix u, R *s, rc_matrix vt) {          \
        return guarded_c(ctx, [&] {                                                                                                       \
            CV<R> Rg = view_of<R>(range), A = view_of<R>(a), U = view_of<R>(u), VT = view_of<R>(vt);                                      \
            const int64_t rr = Rg.cols, n = A.cols, r = std::min(rr, n);                                                                  \
            RC_REQUIRE(Rg.rows == A.rows && U.rows == A.rows && U.cols == r && VT.rows == r && VT.cols == n, RC_INVALID_ARGUMENT, "svd_from_range_estimate: shape mismatch"); \
            CV<R> b = tmp_cm<R>(ctx, rr, n), ub = tmp_cm<R>(ctx, rr, r);                                                                  \
            c_gemm<R>(ctx, 2, 0, one<R>(), Rg, A, zero<R>(), b);                                                                          \
            c_compute_svd<R>(ctx, b, ub, s, VT);                                                                                          \
            c_gemm<R>(ctx, 0, 0, one<R>(), Rg, ub, zero<R>(), U);                                                                         \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_column_id_two_sided_##SUF(rc_context *ctx, rc_matrix c, rc_matrix c_out, rc_matrix x, int64_t *row_ind) {                \
        return guarded_c(ctx, [&] {                                                                                                       \
            CV<R> C = view_of<R>(c);                                                                                                      \
            const int64_t m = C.rows, k = C.cols, kk = std::min(m, k);                                                                    \
            CV<R> l = tmp_cm<R>(ctx, m, kk), ql = tmp_cm<R>(ctx, kk, k);                                                                  \
            c_pivoted_lq<R>(ctx, C, l, ql, row_ind, kk); /* LQ::compute_from, qr.rs:354-362 */                                            \
            c_lq_row_id<R>(ctx, l, ql, row_ind, view_of<R>(c_out), view_of<R>(x));                                                        \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_row_id_two_sided_##SUF(rc_context *ctx, rc_matrix r, rc_matrix x, rc_matrix r_out, int64_t *col_ind) {                   \
        return guarded_c(ctx, [&] {                                                                                                       \
            CV<R> Rr = view_of<R>(r);                                                                                                     \
            const int64_t k = Rr.rows, n = Rr.cols, kk = std::min(k, n);                                                                  \
            CV<R> q = tmp_cm<R>(ctx, k, kk), rr = tmp_cm<R>(ctx, kk, n);                                                                  \
            c_pivoted_qr<R>(ctx, Rr, q, rr, col_ind, kk);                                                                                 \
            c_qr_column_id<R>(ctx, q, rr, col_ind, view_of<R>(x), view_of<R>(r_out));                                                     \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_max_col_norm_##SUF(rc_context *ctx, rc_matrix y, R *out) {                                                               \
        return guarded_c(ctx, [&] {                                                                                                       \
            R *d = ctx->alloc<R>(1);                                                                                                      \
            c_max_col_norm_dev<R>(ctx, view_of<R>(y), d);                                                                                 \
            read_back_r<R>(ctx, d, out, 1);                                                                                               \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_sample_range_by_rank_##SUF(rc_context *ctx, rc_matrix a, int64_t k, int64_t p, rc_matrix omega, uint64_t seed, rc_matrix q) { \
        return guarded_c(ctx, [&] { c_sample_range_by_rank<R>(ctx, COp<R>::of(view_of<R>(a)), k, p, view_of<R>(omega), seed, view_of<R>(q)); }); \
    }                                                                                                                                     \
    rc_status rc_sample_range_power_iteration_##SUF(rc_context *ctx, rc_matrix a, int64_t k, int64_t p, int64_t it, rc_matrix omega, uint64_t seed, rc_matrix q) { \
        return guarded_c(ctx, [&] { c_sample_range_power<R>(ctx, COp<R>::of(view_of<R>(a)), k, p, it, view_of<R>(omega), seed, view_of<R>(q)); }); \
    }                                                                                                                                     \
    rc_status rc_sample_range_adaptive_##SUF(rc_context *ctx, rc_matrix a, double rel_tol, int64_t s, rc_matrix omegas, uint64_t seed, rc_matrix q_cap, \
                                             int64_t *rank, int64_t *hist_rank, double *hist_res, int64_t hist_cap, int64_t *hist_len) {  \
        return guarded_c(ctx, [&] {                                                                                                       \
            c_sample_range_adaptive<R>(ctx, COp<R>::of(view_of<R>(a)), rel_tol, s, view_of<R>(omegas), seed, view_of<R>(q_cap), rank, hist_rank, hist_res, hist_cap, hist_len); \
        });                                                                                                                               \
    }                                                                                                                                     \
    /* the range finders and compute_from_range_estimate over the host's operator callbacks (rc_operator) */                              \
    rc_status rc_sample_range_by_rank_op_##SUF(rc_context *ctx, const rc_operator *op, int64_t k, int64_t p, rc_matrix omega, uint64_t seed, rc_matrix q) { \
        return guarded_c(ctx, [&] { c_sample_range_by_rank<R>(ctx, COp<R>::of(op), k, p, view_of<R>(omega), seed, view_of<R>(q)); });     \
    }                                                                                                                                     \
    rc_status rc_sample_range_power_iteration_op_##SUF(rc_context *ctx, const rc_operator *op, int64_t k, int64_t p, int64_t it, rc_matrix omega, uint64_t seed, rc_matrix q) { \
        return guarded_c(ctx, [&] { c_sample_range_power<R>(ctx, COp<R>::of(op), k, p, it, view_of<R>(omega), seed, view_of<R>(q)); });   \
    }                                                                                                                                     \
    rc_status rc_sample_range_adaptive_op_##SUF(rc_context *ctx, const rc_operator *op, double rel_tol, int64_t s, rc_matrix omegas, uint64_t seed, rc_matrix q_cap, \
                                                int64_t *rank, int64_t *hist_rank, double *hist_res, int64_t hist_cap, int64_t *hist_len) { \
        return guarded_c(ctx, [&] {                                                                                                       \
            c_sample_range_adaptive<R>(ctx, COp<R>::of(op), rel_tol, s, view_of<R>(omegas), seed, view_of<R>(q_cap), rank, hist_rank, hist_res, hist_cap, hist_len); \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_qr_from_range_estimate_op_##SUF(rc_context *ctx, rc_matrix range, const rc_operator *op, rc_matrix q, rc_matrix r, int64_t *ind) { \
        return guarded_c(ctx, [&] {                                                                                                       \
            const COp<R> A = COp<R>::of(op);                                                                                              \
            CV<R> Rg = view_of<R>(range), Q = view_of<R>(q), Rr = view_of<R>(r);                                                          \
            const int64_t rr = Rg.cols, n = A.cols, k = std::min(rr, n);                                                                  \
            RC_REQUIRE(Rg.rows == A.rows && Q.rows == A.rows && Q.cols == k && Rr.rows == k && Rr.cols == n, RC_INVALID_ARGUMENT, "qr_from_range_estimate: shape mismatch"); \
            CV<R> b = tmp_cm<R>(ctx, rr, n), qb = tmp_cm<R>(ctx, rr, k);                                                                  \
            A.project(ctx, Rg, b);                                                                                                        \
            c_pivoted_qr<R>(ctx, b, qb, Rr, ind, k);                                                                                      \
            c_gemm<R>(ctx, 0, 0, one<R>(), Rg, qb, zero<R>(), Q);                                                                         \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_svd_from_range_estimate_op_##SUF(rc_context *ctx, rc_matrix range, const rc_operator *op, rc_matrix u, R *s, rc_matrix vt) { \
        return guarded_c(ctx, [&] {                                                                                                       \
            const COp<R> A = COp<R>::of(op);                                                                                              \
            CV<R> Rg = view_of<R>(range), U = view_of<R>(u), VT = view_of<R>(vt);                                                         \
            const int64_t rr = Rg.cols, n = A.cols, r = std::min(rr, n);                                                                  \
            RC_REQUIRE(Rg.rows == A.rows && U.rows == A.rows && U.cols == r && VT.rows == r && VT.cols == n, RC_INVALID_ARGUMENT, "svd_from_range_estimate: shape mismatch"); \
            CV<R> b = tmp_cm<R>(ctx, rr, n), ub = tmp_cm<R>(ctx, rr, r);                                                                  \
            A.project(ctx, Rg, b);                                                                                                        \
            c_compute_svd<R>(ctx, b, ub, s, VT);                                                                                          \
            c_gemm<R>(ctx, 0, 0, one<R>(), Rg, ub, zero<R>(), U);                                                                         \
        });                                                                                                                               \
    }                                                                                                                                     \
    rc_status rc_column_id_rank_##SUF(rc_context *ctx, rc_matrix a, int64_t k, rc_matrix c, rc_matrix z, int64_t *col_ind) {              \
        return guarded_c(ctx, [&] {                                                                                                       \
            CV<R> A = view_of<R>(a);                                                                                                      \
            const int64_t kk = std::min(k, std::min(A.rows, A.cols));                                                                     \
            CV<R> q = tmp_cm<R>(ctx, A.rows, kk), r = tmp_cm<R>(ctx, kk, A.cols);                                                         \
            c_pivoted_qr<R>(ctx, A, q, r, col_ind, kk);                                                                                   \
            c_qr_column_id<R>(ctx, q, r, col_ind, view_of<R>(c), view_of<R>(z));                                                          \
        });                                                                                                                               \
    }                                                                                                                                     \
    /* singular values are real for every scalar type (src/svd.rs:86-101) */                                                              \
    rc_status rc_svd_rank_by_tolerance_##SUF(rc_context *ctx, const R *s, int64_t len, double tol, int64_t *rank) {                        \
        return c_svd_rank_fwd(ctx, s, len, tol, rank);                                                                                    \
    }                                                                                                                                     \
    /* the fused rSVD + ID call (same members, same "null = skipped" rule as rc_rsvd_id_f64); B = Q^H A is formed once */                \
    rc_status rc_rsvd_id_##SUF(rc_context *ctx, rc_matrix a, int64_t k, int64_t p, rc_matrix omega, uint64_t seed, const rc_rsvd_id_out *out) { \
        if (!out) return RC_INVALID_ARGUMENT;                                                                                             \
        return guarded_c(ctx, [&] { c_rsvd_id<R>(ctx, view_of<R>(a), k, p, view_of<R>(omega), seed, *out); });                            \
    }                                                                                                                                     \
    /* cfg5's batch for complex matrices: the same packed layout, one rank-k column ID per matrix, round-robin over the contexts */      \
    rc_status rc_batch_column_id_##SUF(rc_context *const *ctxs, int32_t nctx, const rc_matrix *mats, int32_t count, int64_t k, void *packed) { \
        if (!ctxs || nctx < 1 || !ctxs[0]) return RC_INVALID_ARGUMENT;                                                                    \
        return guarded_c(ctxs[0], [&] { c_batch_column_id<R>(ctxs, nctx, mats, count, k, packed); });                                     \
    }

RC_DEFINE_COMPLEX(c64, double, rc_complex64)
RC_DEFINE_COMPLEX(c32, float, rc_complex32)

}  // extern "C"
