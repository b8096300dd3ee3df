// Communication-avoiding QR / pivoted QR of TALL-SKINNY matrices (m >> n, n <= ~138 f64)
// for gfx950.  Same results as the Householder chain of kernels_qr.hip (LAPACK ?geqp3 +
// ?orgqr semantics, including the SIGNS of R's diagonal), organised for MI355X:
//
//   Y (m x n)  --CholeskyQR2-->  Y = Q1 R1        three MFMA GEMMs per pass (Gram, apply R^-1),
//                                                  the n x n Cholesky + inverse in ONE workgroup's LDS
//   R1 (n x n) --Householder QRCP in LDS-->  R1 P = Q2 R     (?laqp2 semantics: first-max pivot,
//                                                  LAPACK norm down-dating / tol3z recompute)
//   Q = Q1 Q2  (one GEMM),  then the Householder sign convention of LAPACK is re-imposed:
//   the reflector representation of an orthonormal Q is unique, its signs d_j follow from an
//   LU-type elimination on Q's top k x k block (d_j = -sgn of the j-th pivot candidate), so
//   Q <- Q D, R <- D R reproduce ?geqp3/?orgqr exactly (tests: R to 1e-14 of LAPACK).
//
// Why: the m x n Householder chain needs ~2n dependent kernel launches that each stream the
// whole trailing matrix through single CUs (measured 3.2 ms for 8192 x 133); here the only
// serial parts act on n x n data inside one CU's LDS and everything O(m n^2) is a GEMM.
// CholeskyQR2 is only valid while cond(Y)^2 * eps < 1: k_chol_inv certifies its own result
// (positive pivots, ||Q1^T Q1 - I||_max small before the second pass) and raises a flag, on
// which the caller falls back to the Householder chain (rc_api.hip).
//
// Replaces: ?geqp3 + ?orgqr on sketches (/root/reference/src/pivoted_qr.rs:81-183 as called
// from src/random_sampling.rs:114) and the QR/LQ reduction inside ?gesdd (src/compute_svd.rs:19).
#include "rc_common.hpp"
#include "rc_device.hpp"

namespace rc {

static __host__ __device__ inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

template <typename T> struct TNum;
template <> struct TNum<double> {
    static __device__ inline double tol3z() { return 1.0536712127723509e-08; }
    static __device__ inline double tiny() { return 1e-300; }
};
template <> struct TNum<float> {
    static __device__ inline float tol3z() { return 2.44140625e-04f; }
    static __device__ inline float tiny() { return 1e-30f; }
};

template <typename T>
__device__ inline T wave_sum64(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <typename T>
__device__ inline T row_sum16(T v) {  // sum over an aligned group of 16 lanes
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------
// Cholesky G = R^T R (R upper) + R^{-1}, n x n in the LDS of one workgroup.
//   flag bit 0: non-positive pivot (G not numerically SPD)
//   flag bit 1: check_identity && max|G - I| > ident_tol  (first pass left Q1 too far from orthonormal)
// ---------------------------------------------------------------------------
template <typename T, int NT>
__global__ __launch_bounds__(1024) void k_chol_inv(Mat<T> g, Mat<T> r_out, Mat<T> rinv_out, int check_identity, T ident_tol, T skip_tol, int *flag) {
    // Register-tiled right-looking Cholesky: thread (ti, tc) of the 32 x 32 thread grid owns
    // the entries (ti + 32 a, tc + 32 b), a, b < NT, in registers; per step only pivot row j
    // goes through LDS (double buffered: ONE barrier per step).
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int n = (int)g.rows;
    const int ld = n | 1;
    T *A = reinterpret_cast<T *>(smem_raw);  // A[i * ld + c], used by the inversion phase
    T *svec = A + (size_t)n * ld;            // n
    T *rowbuf = svec + n;                    // 2 * n
    T *red = rowbuf + 2 * n;                 // 16
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ti = tid >> 5, tc = tid & 31;

    T reg[NT][NT];
    T dev = 0;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int i = ti + 32 * a, c = tc + 32 * b;
            T v = 0;
            if (i < n && c < n) {
                v = g.at(i, c);
                dev = max(dev, fabs(v - ((i == c) ? (T)1 : (T)0)));
            }
            reg[a][b] = v;
        }
    if (check_identity) {
        dev = wave_max_dpp(dev);
        if (lane == 0) red[wv] = dev;
        __syncthreads();
        T mx = 0;
        for (int k = 0; k < 16; ++k) mx = max(mx, red[k]);
        if (tid == 0 && !(mx <= ident_tol)) atomicOr(flag, 2);
        // The first pass already delivered orthonormal columns to working accuracy (well-conditioned input:
        // max |Q1^T Q1 - I| <= skip_tol): the second pass would change nothing above that level, so R2 = R2^-1 = I
        // and the 2 n dependent column steps below are skipped.
        if (mx <= skip_tol) {
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const int i = ti + 32 * a, c = tc + 32 * b;
                    if (i < n && c < n) {
                        const T e = (i == c) ? (T)1 : (T)0;
                        r_out.at(i, c) = e;
                        rinv_out.at(i, c) = e;
                    }
                }
            return;
        }
    }

    // One loop for the factor and its inverse.  X = R^-1 by the right-looking column recurrence of X R = I:
    //   X(:, j) /= r_jj ;  X(:, c) -= X(:, j) R(j, c)  for c > j
    // needs exactly the row of R that step j of the factorization completes, so both rank-1 updates share the step's one
    // barrier (the separate bottom-up Gauss-Jordan pass was another n dependent steps).  X is register-tiled like the
    // factor; only tiles on or above the block diagonal exist (a <= b).
    T xr[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) xr[a][b] = (ti + 32 * a == tc + 32 * b) ? (T)1 : (T)0;
    T *colbuf = A;  // 2 * n
    for (int j = 0; j < n; ++j) {
        T *rb = rowbuf + (j & 1) * n;
        T *cb = colbuf + (j & 1) * n;
        const int ja = j >> 5, jt = j & 31;
        if (ti == jt) {  // owners of row j publish it (unscaled; rb[j] is the pivot)
#pragma unroll
            for (int a = 0; a < NT; ++a)
                if (a == ja) {
#pragma unroll
                    for (int b = a; b < NT; ++b) {
                        const int c = tc + 32 * b;
                        if (c >= j && c < n) rb[c] = reg[a][b];
                    }
                }
        }
        if (tc == jt) {  // owners of column j of X publish it (before its scaling; rows <= j)
#pragma unroll
            for (int b = 0; b < NT; ++b)
                if (b == ja) {
#pragma unroll
                    for (int a = 0; a <= b; ++a) {
                        const int i = ti + 32 * a;
                        if (i <= j) cb[i] = xr[a][b];
                    }
                }
        }
        __syncthreads();
        T d = rb[j];
        if (!(d > (T)0)) {
            if (tid == 0) atomicOr(flag, 1);
            d = TNum<T>::tiny();
        }
        const T invr = fast_rsqrt(d), invd = invr * invr;
        T cv[NT];
#pragma unroll
        for (int b = 0; b < NT; ++b) { const int c = tc + 32 * b; cv[b] = (c > j && c < n) ? rb[c] : (T)0; }
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int i = ti + 32 * a;
            if (i == j) {  // row j becomes the final row of R
#pragma unroll
                for (int b = a; b < NT; ++b) {
                    const int c = tc + 32 * b;
                    if (c >= j && c < n) reg[a][b] *= invr;
                }
            } else if (i > j && i < n) {
                const T w = rb[i] * invd;
#pragma unroll
                for (int b = a; b < NT; ++b) reg[a][b] -= w * cv[b];  // entries below the diagonal are never used
            }
            if (i <= j) {  // X(i, c) -= X(i, j) R(j, c) = (cb[i] / r_jj) (rb[c] / r_jj)
                const T xj = cb[i];
                const T w = xj * invd;
#pragma unroll
                for (int b = a; b < NT; ++b) {
                    xr[a][b] -= w * cv[b];
                    if (b == ja && tc == jt) xr[a][b] = xj * invr;  // column j itself: final
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int i = ti + 32 * a, c = tc + 32 * b;
            if (i < n && c < n) {
                r_out.at(i, c) = (b >= a && c >= i) ? reg[a][b] : (T)0;
                rinv_out.at(i, c) = (b >= a && c >= i) ? xr[a][b] : (T)0;
            }
        }
}

template <typename T>
static size_t chol_lds(int64_t n) { return ((size_t)n * (n | 1) + 3 * (size_t)n + 16) * sizeof(T); }

template <typename T, int NT>
static void chol_inv_launch(rc_context *c, Mat<T> g, Mat<T> r, Mat<T> rinv, bool check_identity, T ident_tol, T skip_tol, int *flag) {
    auto kern = k_chol_inv<T, NT>;
    static bool attr_set[64] = {};
    if (!attr_set[c->device & 63]) {
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        attr_set[c->device & 63] = true;
    }
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), chol_lds<T>(g.rows), c->stream, g, r, rinv, check_identity ? 1 : 0, ident_tol, skip_tol, flag);
}
template <typename T>
static void chol_inv(rc_context *c, Mat<T> g, Mat<T> r, Mat<T> rinv, bool check_identity, T ident_tol, T skip_tol, int *flag) {
    const int64_t n = g.rows;
    RC_REQUIRE(n <= 224 && chol_lds<T>(n) <= 160 * 1024 - 1024, RC_INVALID_ARGUMENT, "chol_inv: n = %lld does not fit LDS", (long long)n);
    ProfScope ps(c, "op:chol_inv n=%lld", (long long)n);
    if (n <= 64) chol_inv_launch<T, 2>(c, g, r, rinv, check_identity, ident_tol, skip_tol, flag);
    else if (n <= 160) chol_inv_launch<T, 5>(c, g, r, rinv, check_identity, ident_tol, skip_tol, flag);
    else chol_inv_launch<T, 7>(c, g, r, rinv, check_identity, ident_tol, skip_tol, flag);
}

template <typename T>
bool tsqr_supported(int64_t m, int64_t n) {
    const size_t small_lds = ((size_t)n * (n | 1) + 4 * (size_t)n) * sizeof(T) + (size_t)n * sizeof(int) + 256;
    return n >= 2 && m >= 4 * n && m >= 1024 && small_lds <= 160 * 1024 - 2048;
}

// Y = Q R by CholeskyQR2 with Q left in factored form: Q = q1 r2i.  y: m x n (any strides, not modified), q1: m x n (row-major
// for coalesced GEMM epilogues), r2i: n x n, r: n x n (upper, any strides).  *flag (device int) gets bits OR-ed in on failure.
// Callers fold r2i into whatever small matrix multiplies Q next (Q2 of the small QRCP, U_c / V_c of the small SVD), so the
// tall matrix is read by two Gram products and written / read by two applications instead of three (four with the sign pass).
template <typename T>
void tsqr_cholqr2_factored(rc_context *c, Mat<T> y, Mat<T> q1, Mat<T> r2i, Mat<T> r, int *flag) {
    const int64_t m = y.rows, n = y.cols;
    ProfScope ps(c, "op:cholqr2 %lldx%lld", (long long)m, (long long)n);
    ArenaMark mark(c);
    Mat<T> g = rowmajor(c->alloc<T>((size_t)n * even_ld(n)), n, n, even_ld(n));
    Mat<T> r1 = rowmajor(c->alloc<T>((size_t)n * even_ld(n)), n, n, even_ld(n)), r1i = rowmajor(c->alloc<T>((size_t)n * even_ld(n)), n, n, even_ld(n));
    Mat<T> r2 = rowmajor(c->alloc<T>((size_t)n * even_ld(n)), n, n, even_ld(n));
    gemm<T>(c, 1, y.t(), y, 0, g);
    chol_inv<T>(c, g, r1, r1i, false, 0, 0, flag);
    gemm<T>(c, 1, y, r1i, 0, q1);
    gemm<T>(c, 1, q1.t(), q1, 0, g);
    // after one pass ||Q1^T Q1 - I|| ~ cond(Y)^2 eps; the second pass is only accurate while that is << 1
    static const int skip_ok = [] { const char *e = getenv("RC_CHOLQR_SKIP"); return e ? atoi(e) : 1; }();
    const T skip_tol = skip_ok ? (sizeof(T) == 8 ? (T)2.5e-14 : (T)1e-6) : (T)-1;
    chol_inv<T>(c, g, r2, r2i, true, (T)1e-2, skip_tol, flag);
    gemm<T>(c, 1, r2, r1, 0, r);
}

// Y = Q R with Q formed: q m x n (any strides)
template <typename T>
void tsqr_cholqr2(rc_context *c, Mat<T> y, Mat<T> q, Mat<T> r, int *flag) {
    const int64_t m = y.rows, n = y.cols;
    ArenaMark mark(c);
    Mat<T> q1 = rowmajor(c->alloc<T>((size_t)m * even_ld(n)), m, n, even_ld(n));
    Mat<T> r2i = rowmajor(c->alloc<T>((size_t)n * even_ld(n)), n, n, even_ld(n));
    tsqr_cholqr2_factored<T>(c, y, q1, r2i, r, flag);
    gemm<T>(c, 1, q1, r2i, 0, q);
}

// ---------------------------------------------------------------------------
// Householder QRCP (?laqp2 semantics) of an n x n matrix in the LDS of ONE workgroup,
// plus the explicit orthogonal factor.  Columns are never moved (jp maps position -> column).
//   rin  : n x n input (any strides)
//   rout : kmax x n, rout(i, p) = (i <= p) ? R(i, jp[p]) : 0      (may be empty)
//   q2   : n x q2.cols = H_0 ... H_{kmax-1} [I ; 0]                 (may be empty)
// ---------------------------------------------------------------------------
template <typename T>
__device__ inline T safe_hypot(T a, T b) {  // ?lapy2
    a = fabs(a); b = fabs(b);
    const T w = max(a, b), z = min(a, b);
    if (z == (T)0) return w;
    const T q = z * fast_rcp(w);
    const T e = (T)1 + q * q;
    return w * (e * fast_rsqrt(e));
}
template <typename T>
__device__ inline T fast_sqrt_pos(T x) { return x > (T)0 ? x * fast_rsqrt(x) : (T)0; }

// One Householder step of k_qrcp_small applied to the trailing columns (one LPP-lane group per column, NEJ rows per lane
// from row j on) + the ?laqp2 norm down-date.  safe != 0: a read past the end of a column stays inside the LDS image (it
// lands in the next column or in the norm / tau / permutation arrays behind the matrix), so the loads carry no bounds:
// the reflector entries there are zero, the products are exact zeros and the sums are the same bits as with bounds.
template <typename T, int NEJ, int LPP>
__device__ __forceinline__ void qrcp_small_apply(T *A, int ld, int n, int j, int grp, int ll, int ngrp, const int *jp, T *vn1, T *vn2, T tj, int pivot, bool safe) {
    const T *vcol = A + jp[j] * ld;
    T v[NEJ];
#pragma unroll
    for (int e = 0; e < NEJ; ++e) {
        int i = j + ll + LPP * e;
        v[e] = (i < n) ? ((i == j) ? (T)1 : vcol[i]) : (T)0;
    }
    for (int p = j + 1 + grp; p < n; p += ngrp) {
        T *xcol = A + jp[p] * ld;
        T x[NEJ];
        T dot = 0;
        if (safe) {
#pragma unroll
            for (int e = 0; e < NEJ; ++e) {
                x[e] = xcol[j + ll + LPP * e];
                dot = fma(v[e], x[e], dot);
            }
        } else {
#pragma unroll
            for (int e = 0; e < NEJ; ++e) {
                int i = j + ll + LPP * e;
                x[e] = (i < n) ? xcol[i] : (T)0;
                dot = fma(v[e], x[e], dot);
            }
        }
        if (tj != (T)0) {
            dot = group_sum_dpp<LPP>(dot);
            const T f = tj * dot;
#pragma unroll
            for (int e = 0; e < NEJ; ++e) {
                int i = j + ll + LPP * e;
                if (i < n) { x[e] -= f * v[e]; xcol[i] = x[e]; }
            }
        }
        if (pivot) {
            const T xj = xcol[j];  // written by lane ll == 0 of this group: same wave, LDS ops are in order
            const T vn = vn1[p];
            if (vn != (T)0) {
                T t = fabs(xj) * fast_rcp(vn);
                T temp = (T)1 - t * t;
                temp = temp > (T)0 ? temp : (T)0;
                T rr = vn * fast_rcp(vn2[p]);
                T temp2 = temp * rr * rr;
                if (temp2 <= TNum<T>::tol3z()) {
                    T ss = 0;
#pragma unroll
                    for (int e = 0; e < NEJ; ++e) {
                        int i = j + ll + LPP * e;
                        if (i > j && i < n) ss += x[e] * x[e];
                    }
                    ss = group_sum_dpp<LPP>(ss);
                    if (ll == 0) { T nn = (j < n - 1) ? fast_sqrt_pos(ss) : (T)0; vn1[p] = nn; vn2[p] = nn; }
                } else if (ll == 0) {
                    vn1[p] = vn * fast_sqrt_pos(temp);
                }
            }
        }
    }
}

// x <- H_j x for the Q2 formation of k_qrcp_small: rows ll + LPP e, register blocks E0 .. NE - 1 (the blocks below E0 lie
// above row j, where the reflector is zero).
template <typename T, int NE, int LPP, int E0>
__device__ __forceinline__ void qrcp_small_reflect(T (&x)[NE], const T *vcol, T tj, int j, int n, int ll) {
    T v[NE];
    T dot = 0;
#pragma unroll
    for (int e = E0; e < NE; ++e) {
        int i = ll + LPP * e;
        v[e] = 0;
        if (i < n && i >= j) { v[e] = (i == j) ? (T)1 : vcol[i]; dot = fma(v[e], x[e], dot); }
    }
    dot = group_sum_dpp<LPP>(dot);
    const T f = tj * dot;
#pragma unroll
    for (int e = E0; e < NE; ++e) x[e] -= f * v[e];
}

template <typename T, int NE, int NTHR>
__global__ __launch_bounds__(NTHR) void k_qrcp_small(Mat<T> rin, int kmax, int pivot, int64_t *jpvt_out, Mat<T> rout, Mat<T> q2) {
    constexpr int LPP = 8;     // lanes per column in the update
    // NTHR threads = NTHR / 8 column groups: 1024 (n <= 144) covers the trailing columns of a step in one or two passes
    constexpr int NGRP = NTHR / LPP;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int n = (int)rin.rows;
    const int ld = n | 1;
    T *A = reinterpret_cast<T *>(smem_raw);  // column-major: A[c * ld + i]
    T *vn1 = A + (size_t)n * ld;
    T *vn2 = vn1 + n;
    T *tau = vn2 + n;
    int *jp = reinterpret_cast<int *>(tau + n);
    const int tid = threadIdx.x, lane = tid & 63;
    const int ll = tid % LPP, grp = tid / LPP;
    const bool safe = LPP * NE <= 3 * n;  // rows read past a column's end stay inside the LDS image (see qrcp_small_apply)

    for (int e = tid; e < n * n; e += NTHR) {
        int i = e % n, cc = e / n;
        A[cc * ld + i] = rin.at(i, cc);
    }
    __syncthreads();
    for (int p = grp; p < n; p += NGRP) {
        T acc = 0;
        for (int i = ll; i < n; i += LPP) { T v = A[p * ld + i]; acc += v * v; }
        acc = group_sum_dpp<LPP>(acc);
        if (ll == 0) { T nr = sqrt(acc); vn1[p] = nr; vn2[p] = nr; jp[p] = p; tau[p] = (T)0; }  // tau: unbounded reads may land on it
    }
    __syncthreads();

    for (int j = 0; j < kmax; ++j) {
        if (tid < 64) {  // wave 0: pivot search (first maximum, like idamax), "swap", reflector (?larfg)
            if (pivot) {
                T best = (T)-1;
                int bi = 0x7fffffff;
                for (int p = j + lane; p < n; p += 64) {
                    T v = fabs(vn1[p]);
                    if (v > best) { best = v; bi = p; }
                }
                const T mx = wave_max_dpp(best);
                const int pv = wave_min_dpp(best == mx ? bi : 0x7fffffff);
                const int pvt = (pv >= j && pv < n) ? pv : j;
                if (lane == 0 && pvt != j) {
                    int t = jp[pvt]; jp[pvt] = jp[j]; jp[j] = t;
                    vn1[pvt] = vn1[j];
                    vn2[pvt] = vn2[j];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            T *col = A + jp[j] * ld;
            const T alpha = col[j];
            T acc = 0;
            for (int i = j + 1 + lane; i < n; i += 64) { T v = col[i]; acc += v * v; }
            acc = wave_sum_dpp(acc);
            const T xnorm = fast_sqrt_pos(acc);
            if (xnorm == (T)0) {
                if (lane == 0) tau[j] = 0;
            } else {
                const T beta = -copysign(safe_hypot(alpha, xnorm), alpha);
                const T scal = fast_rcp(alpha - beta);
                for (int i = j + 1 + lane; i < n; i += 64) col[i] *= scal;
                if (lane == 0) { tau[j] = (beta - alpha) * fast_rcp(beta); col[j] = beta; }
            }
        }
        __syncthreads();
        {   // apply H_j to the remaining columns; one LPP-lane group per column.  The register tile covers the rows that are
            // left (the step is bound by the instructions issued, so short tiles for the late steps count)
            const int rem = n - j;
            const T tj = tau[j];
            if (NE > 12 && rem > 8 * 12) qrcp_small_apply<T, NE, LPP>(A, ld, n, j, grp, ll, NGRP, jp, vn1, vn2, tj, pivot, safe);
            else if (NE > 8 && rem > 8 * 8) qrcp_small_apply<T, (NE > 12 ? 12 : NE), LPP>(A, ld, n, j, grp, ll, NGRP, jp, vn1, vn2, tj, pivot, safe);
            else if (NE > 4 && rem > 8 * 4) qrcp_small_apply<T, (NE > 8 ? 8 : NE), LPP>(A, ld, n, j, grp, ll, NGRP, jp, vn1, vn2, tj, pivot, safe);
            else qrcp_small_apply<T, (NE > 4 ? 4 : NE), LPP>(A, ld, n, j, grp, ll, NGRP, jp, vn1, vn2, tj, pivot, safe);
        }
        __syncthreads();
    }

    for (int p = tid; p < n; p += NTHR) jpvt_out[p] = jp[p];
    if (!rout.empty()) {
        const int64_t total = rout.rows * rout.cols;
        for (int64_t e = tid; e < total; e += NTHR) {
            int i = (int)(e / rout.cols), p = (int)(e % rout.cols);
            rout.at(i, p) = (i <= p) ? A[jp[p] * ld + i] : (T)0;
        }
    }
    if (!q2.empty()) {  // column cq of Q2: apply H_min(cq,kmax-1) ... H_0 to e_cq; A is read-only now
        for (int cq = grp; cq < (int)q2.cols; cq += NGRP) {
            T x[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) x[e] = (ll + LPP * e == cq) ? (T)1 : (T)0;
            for (int j = (cq < kmax - 1 ? cq : kmax - 1); j >= 0; --j) {
                const T tj = tau[j];
                if (tj == (T)0) continue;
                const T *vcol = A + jp[j] * ld;
                // register blocks that lie entirely above row j hold only zeros of the reflector: the instance that starts at
                // the largest multiple of four blocks below row j skips them (same sums, the skipped products are exact zeros)
                const int e0 = j / LPP;
                if (NE > 12 && e0 >= 12) qrcp_small_reflect<T, NE, LPP, (NE > 12 ? 12 : 0)>(x, vcol, tj, j, n, ll);
                else if (NE > 8 && e0 >= 8) qrcp_small_reflect<T, NE, LPP, (NE > 8 ? 8 : 0)>(x, vcol, tj, j, n, ll);
                else if (NE > 4 && e0 >= 4) qrcp_small_reflect<T, NE, LPP, (NE > 4 ? 4 : 0)>(x, vcol, tj, j, n, ll);
                else qrcp_small_reflect<T, NE, LPP, 0>(x, vcol, tj, j, n, ll);
            }
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                int i = ll + LPP * e;
                if (i < n) q2.at(i, cq) = x[e];
            }
        }
    }
}

template <typename T>
void qrcp_small(rc_context *c, Mat<T> rin, int64_t kmax, bool pivot, int64_t *jpvt, Mat<T> rout, Mat<T> q2) {
    const int64_t n = rin.rows;
    RC_REQUIRE(rin.rows == rin.cols, RC_INVALID_ARGUMENT, "qrcp_small: square input required");
    const size_t lds = ((size_t)n * (n | 1) + 3 * (size_t)n) * sizeof(T) + (size_t)n * sizeof(int) + 64;
    RC_REQUIRE(lds <= 160 * 1024 - 1024 && n <= 208, RC_INVALID_ARGUMENT, "qrcp_small: n = %lld does not fit LDS", (long long)n);
    ProfScope ps(c, "op:qrcp_small n=%lld k=%lld", (long long)n, (long long)kmax);
#define RC_QS(NE, NT_)                                                                                                 \
    do {                                                                                                               \
        auto kern = k_qrcp_small<T, NE, NT_>;                                                                          \
        static bool attr_set[64] = {};                                                                                 \
        if (!attr_set[c->device & 63]) {                                                                               \
            RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024)); \
            attr_set[c->device & 63] = true;                                                                           \
        }                                                                                                              \
        hipLaunchKernelGGL(kern, dim3(1), dim3(NT_), lds, c->stream, rin, (int)kmax, pivot ? 1 : 0, jpvt, rout, q2);   \
    } while (0)
    static const int wide = [] { const char *e = getenv("RC_QRCP_SMALL_1024"); return e ? atoi(e) : 1; }();
    if (n <= 64) RC_QS(8, 512);
    else if (n <= 144 && wide) RC_QS(18, 1024);
    else if (n <= 144) RC_QS(18, 512);
    else RC_QS(26, 512);
#undef RC_QS
}

// ---------------------------------------------------------------------------
// Householder sign convention: d_j such that Q D = H_0 ... H_{k-1} [I ; 0] with LAPACK's
// reflectors (beta = -sign(alpha) |x|).  LU-type elimination on the top k x k block of Q.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void k_householder_signs(Mat<T> q, T *d) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int k = (int)q.cols;
    const int ld = k | 1;
    T *W = reinterpret_cast<T *>(smem_raw);  // W[i * ld + c]
    const int tid = threadIdx.x;
    for (int e = tid; e < k * k; e += 1024) {
        int i = e % k, cc = e / k;
        W[i * ld + cc] = q.at(i, cc);
    }
    __syncthreads();
    const int ti = tid >> 5, tc = tid & 31;
    for (int j = 0; j < k; ++j) {
        const T s = W[j * ld + j];
        const T dj = (s > (T)0) ? (T)-1 : (T)1;
        const T u = (T)1 + fabs(s);
        if (tid == 0) d[j] = dj;
        const T inv = -dj / u;  // one division per step and thread (it was one per row: the kernel was bound by them)
        for (int i = j + 1 + ti; i < k; i += 32) {
            const T li = W[i * ld + j] * inv;
            for (int cc = j + 1 + tc; cc < k; cc += 32) W[i * ld + cc] -= li * W[j * ld + cc];
        }
        __syncthreads();
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_scale_cols_rows(Mat<T> q, Mat<T> r, const T *d) {
    // q(:, j) *= d_j ; r(j, :) *= d_j
    const int64_t tq = q.rows * q.cols, tr = r.rows * r.cols;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tq + tr; e += (int64_t)gridDim.x * blockDim.x) {
        if (e < tq) {
            int64_t j = e / q.rows, i = e - j * q.rows;
            if (d[j] < (T)0) q.at(i, j) = -q.at(i, j);
        } else {
            int64_t f = e - tq;
            int64_t i = f / r.cols, p = f - i * r.cols;
            if (d[i] < (T)0) r.at(i, p) = -r.at(i, p);
        }
    }
}

// d_j of the sign convention from the top k x k block of Q (k x k, any strides)
template <typename T>
static void householder_signs(rc_context *c, Mat<T> top, T *d) {
    const int64_t k = top.cols;
    const size_t lds = (size_t)k * (k | 1) * sizeof(T);
    RC_REQUIRE(top.rows >= k && lds <= 160 * 1024 - 1024, RC_INVALID_ARGUMENT, "sign fix: k = %lld does not fit LDS", (long long)k);
    auto kern = k_householder_signs<T>;
    static bool attr_set[64] = {};
    if (!attr_set[c->device & 63]) {
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        attr_set[c->device & 63] = true;
    }
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, c->stream, top, d);
}
template <typename T>
static void scale_cols_rows(rc_context *c, Mat<T> q, Mat<T> r, const T *d) {
    const int64_t total = q.rows * q.cols + r.rows * r.cols;
    hipLaunchKernelGGL(k_scale_cols_rows<T>, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 4096)), dim3(256), 0, c->stream, q, r, d);
}

template <typename T>
void householder_sign_fix(rc_context *c, Mat<T> q, Mat<T> r) {
    const int64_t k = q.cols;
    if (k == 0) return;
    RC_REQUIRE(q.rows >= k, RC_INVALID_ARGUMENT, "sign fix: q must be tall");
    ProfScope ps(c, "op:householder_sign_fix k=%lld", (long long)k);
    ArenaMark mark(c);
    T *d = c->alloc<T>((size_t)k);
    householder_signs<T>(c, q, d);
    scale_cols_rows<T>(c, q, r, d);
}

// Pivoted (or plain) QR of the tall-skinny y through CholeskyQR2 + small QRCP + sign fix.
//   q: m x k column-major (may be empty), r: k x n (may be empty), ind: n
// Q = Q1 (R2^-1 Q2 D): the n x k factor W = R2^-1 Q2 is formed first, the signs D (+-1: exact) come from the top k x k block
// Q1[:k, :] W and go into W's columns and R's rows, and ONE tall product writes Q.  (Until round 3: Q1 R2^-1, then (.) Q2, then a
// sign pass over Q: two more passes over the tall matrix; RC_TSQR_FOLD=0 restores that order.)
template <typename T>
void qrcp_tall_fast(rc_context *c, Mat<T> y, int64_t k, bool pivot, Mat<T> q, Mat<T> r, int64_t *ind, int *flag) {
    const int64_t m = y.rows, n = y.cols;
    ProfScope ps(c, "op:qrcp_tall_fast %lldx%lld k=%lld pivot=%d", (long long)m, (long long)n, (long long)k, pivot ? 1 : 0);
    ArenaMark mark(c);
    static const int fold = [] { const char *e = getenv("RC_TSQR_FOLD"); return e ? atoi(e) : 1; }();
    Mat<T> q1 = rowmajor(c->alloc<T>((size_t)m * even_ld(n)), m, n, even_ld(n));
    Mat<T> r1 = colmajor(c->alloc<T>((size_t)n * even_ld(n)), n, n, even_ld(n));
    Mat<T> r2i = rowmajor(c->alloc<T>((size_t)n * even_ld(n)), n, n, even_ld(n));
    if (fold) tsqr_cholqr2_factored<T>(c, y, q1, r2i, r1, flag);
    else tsqr_cholqr2<T>(c, y, q1, r1, flag);
    Mat<T> q2 = colmajor(c->alloc<T>((size_t)even_ld(n) * k), n, k, even_ld(n));
    Mat<T> rr = r.empty() ? rowmajor(c->alloc<T>((size_t)k * even_ld(n)), k, n, even_ld(n)) : r;
    qrcp_small<T>(c, r1, k, pivot, ind, rr, q2);
    Mat<T> qq = q.empty() ? rowmajor(c->alloc<T>((size_t)m * even_ld(k)), m, k, even_ld(k)) : q;
    if (!fold) {
        gemm<T>(c, 1, q1, q2, 0, qq);
        householder_sign_fix<T>(c, qq, rr);
        return;
    }
    Mat<T> w = colmajor(c->alloc<T>((size_t)even_ld(n) * k), n, k, even_ld(n));
    Mat<T> top = colmajor(c->alloc<T>((size_t)even_ld(k) * k), k, k, even_ld(k));
    T *d = c->alloc<T>((size_t)k);
    gemm<T>(c, 1, r2i, q2, 0, w);
    gemm<T>(c, 1, q1.sub(0, k, 0, n), w, 0, top);
    householder_signs<T>(c, top, d);
    scale_cols_rows<T>(c, w, rr, d);
    gemm<T>(c, 1, q1, w, 0, qq);
}

#define RC_INST(T)                                                                                               \
    template bool tsqr_supported<T>(int64_t, int64_t);                                                           \
    template void tsqr_cholqr2<T>(rc_context *, Mat<T>, Mat<T>, Mat<T>, int *);                                  \
    template void tsqr_cholqr2_factored<T>(rc_context *, Mat<T>, Mat<T>, Mat<T>, Mat<T>, int *);                 \
    template void qrcp_small<T>(rc_context *, Mat<T>, int64_t, bool, int64_t *, Mat<T>, Mat<T>);                 \
    template void householder_sign_fix<T>(rc_context *, Mat<T>, Mat<T>);                                         \
    template void qrcp_tall_fast<T>(rc_context *, Mat<T>, int64_t, bool, Mat<T>, Mat<T>, int64_t *, int *);
RC_INST(double)
RC_INST(float)
#undef RC_INST

}  // namespace rc
