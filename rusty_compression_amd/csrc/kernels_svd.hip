// Small-core SVD for gfx950: one-sided (Hestenes) Jacobi on a square n x n
// matrix, n <= 1024, inside ONE workgroup.
//
// Replaces the dense core of LAPACK ?gesdd as the reference reaches it through
// ndarray-linalg `svddc_into(JobSvd::Some)` (/root/reference/src/compute_svd.rs:19).
// The tall/wide input is first reduced to its square triangular factor by the
// Householder QR of kernels_qr.hip (rc_api.hip: compute_svd), so this kernel only
// ever sees min(m, n) x min(m, n).  One-sided Jacobi computes every singular
// value to high RELATIVE accuracy, which is what the f64 <= 1e-12 round-trip
// bound of the reference tests (src/svd.rs:290-297) needs.
//
// Mapping: round-robin (circle) ordering gives n/2 independent column pairs per
// round; 16 lanes (a quarter wave) own one pair, so a 1024-thread workgroup
// rotates 64 pairs per pass with quarter-wave shuffles and one barrier per round.
#include "rc_common.hpp"

namespace rc {

template <typename T> struct JEps;
template <> struct JEps<double> { static __device__ inline double eps() { return 1.1102230246251565e-16; } };
template <> struct JEps<float> { static __device__ inline float eps() { return 5.9604644775390625e-08f; } };

template <typename T>
__device__ inline T qsum16(T v) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// g, v: column-major n x n (cs = ld).  On exit uc/vc hold the singular vectors
// sorted by descending singular value, s the singular values.
template <typename T>
__global__ __launch_bounds__(1024) void k_jacobi_svd(Mat<T> g, Mat<T> v, Mat<T> uc, T *s, Mat<T> vc, int max_sweeps) {
    __shared__ int sh_rot;
    __shared__ T sig[1024];
    __shared__ int order[1024];
    const int tid = threadIdx.x;
    const int l16 = tid & 15, grp = tid >> 4;  // 64 groups of 16 lanes
    const int n = (int)g.rows;
    const int N = (n + 1) & ~1;  // padded to even; column n (if any) is a dummy
    const int npairs = N / 2;
    const T tol = sqrt((T)n) * JEps<T>::eps();

    for (int e = tid; e < n * n; e += 1024) {
        int i = e % n, j = e / n;
        v.p[j * v.cs + i] = (i == j) ? (T)1 : (T)0;
    }
    __syncthreads();

    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        if (tid == 0) sh_rot = 0;
        __syncthreads();
        for (int r = 0; r < N - 1; ++r) {
            for (int pi = grp; pi < npairs; pi += 64) {
                int p, q;
                if (pi == 0) { p = N - 1; q = r; }
                else { p = (r + pi) % (N - 1); q = (r - pi + (N - 1)) % (N - 1); }
                if (p >= n || q >= n) continue;  // dummy column of an odd n
                if (p > q) { int t = p; p = q; q = t; }
                T *gp = g.p + (int64_t)p * g.cs, *gq = g.p + (int64_t)q * g.cs;
                T app = 0, aqq = 0, apq = 0;
                for (int i = l16; i < n; i += 16) {
                    T a = gp[i], b = gq[i];
                    app += a * a; aqq += b * b; apq += a * b;
                }
                app = qsum16(app); aqq = qsum16(aqq); apq = qsum16(apq);
                if (apq == (T)0 || fabs(apq) <= tol * sqrt(app) * sqrt(aqq)) continue;  // uniform over the 16 lanes
                const T zeta = (aqq - app) / ((T)2 * apq);
                const T t = copysign((T)1, zeta) / (fabs(zeta) + sqrt((T)1 + zeta * zeta));
                const T cs = (T)1 / sqrt((T)1 + t * t), sn = cs * t;
                for (int i = l16; i < n; i += 16) {
                    T a = gp[i], b = gq[i];
                    gp[i] = cs * a - sn * b;
                    gq[i] = sn * a + cs * b;
                }
                T *vp = v.p + (int64_t)p * v.cs, *vq = v.p + (int64_t)q * v.cs;
                for (int i = l16; i < n; i += 16) {
                    T a = vp[i], b = vq[i];
                    vp[i] = cs * a - sn * b;
                    vq[i] = sn * a + cs * b;
                }
                if (l16 == 0) sh_rot = 1;
            }
            __syncthreads();  // pairs of one round are disjoint; the next round re-pairs the columns
        }
        const int rotated = sh_rot;
        __syncthreads();
        if (!rotated) break;
    }

    // singular values = column norms
    for (int j = grp; j < n; j += 64) {
        const T *gj = g.p + (int64_t)j * g.cs;
        T acc = 0;
        for (int i = l16; i < n; i += 16) { T a = gj[i]; acc += a * a; }
        acc = qsum16(acc);
        if (l16 == 0) sig[j] = sqrt(acc);
    }
    __syncthreads();
    // rank sort, descending, stable (gesdd returns S descending)
    for (int i = tid; i < n; i += 1024) {
        int rank = 0;
        const T si = sig[i];
        for (int j = 0; j < n; ++j) rank += (sig[j] > si || (sig[j] == si && j < i)) ? 1 : 0;
        order[i] = rank;
        s[rank] = si;
    }
    __syncthreads();
    for (int j = grp; j < n; j += 64) {
        const int dst = order[j];
        const T sj = sig[j];
        const T inv = sj > (T)0 ? (T)1 / sj : (T)0;
        const T *gj = g.p + (int64_t)j * g.cs, *vj = v.p + (int64_t)j * v.cs;
        for (int i = l16; i < n; i += 16) {
            uc.at(i, dst) = gj[i] * inv;
            vc.at(i, dst) = vj[i];
        }
    }
}

template <typename T>
void jacobi_svd(rc_context *c, Mat<T> g, Mat<T> vwork, Mat<T> uc, T *s, Mat<T> vc) {
    RC_REQUIRE(g.rows == g.cols && g.rs == 1 && vwork.rs == 1, RC_LAYOUT_ERROR, "jacobi_svd: square column-major core required");
    RC_REQUIRE(g.rows <= 1024, RC_INVALID_ARGUMENT, "compute_svd: min(m, n) = %lld > 1024 is not supported by the single-workgroup Jacobi core",
               (long long)g.rows);
    if (g.rows == 0) return;
    ProfScope ps(c, "op:jacobi_svd n=%lld", (long long)g.rows);
    hipLaunchKernelGGL(k_jacobi_svd<T>, dim3(1), dim3(1024), 0, c->stream, g, vwork, uc, s, vc, 60);
}

template void jacobi_svd<double>(rc_context *, Mat<double>, Mat<double>, Mat<double>, double *, Mat<double>);
template void jacobi_svd<float>(rc_context *, Mat<float>, Mat<float>, Mat<float>, float *, Mat<float>);

}  // namespace rc
