// Householder QR with column pivoting, form-Q and triangular solves for gfx950.
//
// Replaces LAPACK ?geqp3 (+ ?laqp2 / ?larfg), ?orgqr and ?trtrs as the
// reference reaches them:
//   /root/reference/src/pivoted_qr.rs:139-150, :161-172   (?geqp3)
//   /root/reference/src/pivoted_qr.rs:104-108             (?orgqr via lax::Lapack::q)
//   /root/reference/src/qr.rs:290-301, :384-395           (?trtrs per column / per row)
//
// Numerical contract (what makes the permutation reproducible): ?laqp2
// semantics -- pivot = FIRST maximum of the partial column norms (idamax),
// reflector with beta = -sign(alpha) * hypot(alpha, |x|), partial norms
// down-dated with the LAPACK formula and recomputed from scratch when
// temp2 <= sqrt(eps) (tol3z).
//
// MI355X mapping.  The m x n working matrix is column-major and columns are
// never moved: `jpvt` maps factorization position -> physical column, so a
// "swap" is two index writes.  Reflector application is independent per column,
// so the trailing update is column-parallel (one wave or one workgroup per
// column, the column held in registers between the dot product and the update:
// one read + one write of the trailing matrix per step, nothing else); the only
// serial piece per step is the single-workgroup pivot search + reflector
// generation.
#include "rc_common.hpp"
#include "rc_device.hpp"

namespace rc {

static __host__ __device__ inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

template <typename T> struct Num;
template <> struct Num<double> {
    static __host__ __device__ inline double tol3z() { return 1.0536712127723509e-08; }  // sqrt(2^-53)
};
template <> struct Num<float> {
    static __host__ __device__ inline float tol3z() { return 2.44140625e-04f; }  // sqrt(2^-24)
};

template <typename T>
__device__ inline T wsum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum over a group of TPC threads (TPC = 64: one wave; TPC = 256: the 4-wave workgroup)
template <typename T, int TPC>
__device__ inline T group_sum(T v, T *sh /* 4 entries, TPC == 256 only */) {
    v = wsum(v);
    if (TPC == 256) {
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        v = sh[0] + sh[1] + sh[2] + sh[3];
    }
    return v;
}

// ---------------------------------------------------------------------------
// init: jpvt = iota, vn1 = vn2 = column norms
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_qr_init(Mat<T> w, int pivot, int64_t *jpvt, T *vn1, T *vn2) {
    __shared__ T sh[4];
    for (int64_t j = blockIdx.x; j < w.cols; j += gridDim.x) {
        if (pivot) {
            const T *col = w.p + j * w.cs;
            T acc = 0;
            for (int64_t i = threadIdx.x; i < w.rows; i += 256) { T v = col[i]; acc += v * v; }
            acc = group_sum<T, 256>(acc, sh);
            if (threadIdx.x == 0) { T nrm = sqrt(acc); vn1[j] = nrm; vn2[j] = nrm; }
        }
        if (threadIdx.x == 0) jpvt[j] = j;
    }
}

// ---------------------------------------------------------------------------
// step j, serial part: pivot search + "swap" + reflector generation (?larfg)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void k_qr_pivot_reflect(Mat<T> w, int64_t j, int pivot, int64_t *jpvt, T *vn1, T *vn2, T *tau) {
    __shared__ T shv[16];
    __shared__ long long shi[16];
    __shared__ T shs[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t n = w.cols, m = w.rows;
    if (pivot) {
        T best = (T)-1;
        long long bi = 0x7fffffffffffffffLL;
        for (int64_t p = j + tid; p < n; p += 1024) {
            T v = fabs(vn1[p]);
            if (v > best) { best = v; bi = p; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            T ob = __shfl_xor(best, off, 64);
            long long oi = __shfl_xor(bi, off, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) { shv[wv] = best; shi[wv] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int k = 1; k < 16; ++k)
                if (shv[k] > best || (shv[k] == best && shi[k] < bi)) { best = shv[k]; bi = shi[k]; }
            int64_t pvt = (bi >= j && bi < n) ? (int64_t)bi : j;
            if (pvt != j) {  // dlaqp2: swap columns (here: indices), carry the norms of position j to pvt
                int64_t t = jpvt[pvt]; jpvt[pvt] = jpvt[j]; jpvt[j] = t;
                vn1[pvt] = vn1[j];
                vn2[pvt] = vn2[j];
            }
        }
        __syncthreads();
    }
    // ---- ?larfg on column jpvt[j], rows j..m-1 -------------------------------
    T *col = w.p + jpvt[j] * w.cs;
    const T alpha = col[j];
    T acc = 0;
    for (int64_t i = j + 1 + tid; i < m; i += 1024) { T v = col[i]; acc += v * v; }
    acc = wsum(acc);
    if (lane == 0) shs[wv] = acc;
    __syncthreads();  // also orders every thread's read of alpha before the write of beta below
    T ssq = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) ssq += shs[k];
    const T xnorm = sqrt(ssq);
    if (xnorm == (T)0) {
        if (tid == 0) tau[j] = 0;  // H = I
        return;
    }
    const T beta = -copysign(hypot(alpha, xnorm), alpha);
    const T scal = (T)1 / (alpha - beta);
    for (int64_t i = j + 1 + tid; i < m; i += 1024) col[i] *= scal;
    if (tid == 0) {
        tau[j] = (beta - alpha) / beta;
        col[j] = beta;
    }
}

// ---------------------------------------------------------------------------
// step j, parallel part: apply H_j to every remaining column and down-date its
// partial norm.  TPC threads per column; the column segment lives in registers
// (MAXE elements per thread) between the dot product and the update.
// ---------------------------------------------------------------------------
template <typename T, int TPC, int MAXE>
__global__ __launch_bounds__(256) void k_qr_apply(Mat<T> w, int64_t j, int pivot, const int64_t *jpvt, T *vn1, T *vn2, const T *tau) {
    __shared__ T sh[4];
    __shared__ T shx;
    constexpr int CPW = 256 / TPC;
    const int lt = threadIdx.x % TPC;
    const int64_t n = w.cols, m = w.rows;
    const int64_t p = j + 1 + (int64_t)blockIdx.x * CPW + threadIdx.x / TPC;
    const bool active = p < n;
    if (TPC == 64 && !active) return;  // whole wave idle: no barrier is used on this path
    const T tj = tau[j];
    // 32-bit element offsets from the two (wave-uniform) column bases at row j: one address register per
    // array instead of a 64-bit pointer per element
    const T *vj = w.p + jpvt[j] * w.cs + j;
    T *xj_ = w.p + jpvt[active ? p : j] * w.cs + j;
    const int mrem = (int)(m - j);

    T x[MAXE], v[MAXE];
    T dot = 0;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        // branch-free: out-of-range lanes read row j (always valid) and are masked by a select -- per-element
        // branches made the compiler shuffle the register arrays (thousands of moves, spills in the f32 build)
        const int li = lt + e * TPC;
        const bool ok = li < mrem;
        const int lc = ok ? li : 0;
        const T xv = xj_[lc], vv = vj[lc];
        x[e] = ok ? xv : (T)0;
        v[e] = ok ? (li == 0 ? (T)1 : vv) : (T)0;
        dot = fma(v[e], x[e], dot);
    }
    if (tj != (T)0) {  // tau == 0: H = I (dlarf skips the update)
        dot = group_sum<T, TPC>(dot, sh);
        const T f = tj * dot;
#pragma unroll
        for (int e = 0; e < MAXE; ++e) {
            const int li = lt + e * TPC;
            x[e] = fma(-f, v[e], x[e]);  // v is zero out of range
            if (li < mrem) xj_[li] = x[e];
        }
    }
    if (!pivot) return;
    // ---- partial-norm down-date (dlaqp2) --------------------------------------
    T xj;
    if (TPC == 64) {
        xj = __shfl(x[0], 0, 64);
    } else {
        __syncthreads();
        if (threadIdx.x == 0) shx = x[0];
        __syncthreads();
        xj = shx;
    }
    const T vn = vn1[p];
    if (vn != (T)0) {
        T t = fabs(xj) / vn;
        T temp = (T)1 - t * t;
        temp = temp > (T)0 ? temp : (T)0;
        T r = vn / vn2[p];
        T temp2 = temp * r * r;
        if (temp2 <= Num<T>::tol3z()) {
            T ss = 0;
#pragma unroll
            for (int e = 0; e < MAXE; ++e) {
                const int li = lt + e * TPC;
                if (li > 0 && li < mrem) ss += x[e] * x[e];
            }
            ss = group_sum<T, TPC>(ss, sh);
            if (lt == 0) { T nn = (j < m - 1) ? sqrt(ss) : (T)0; vn1[p] = nn; vn2[p] = nn; }
        } else if (lt == 0) {
            vn1[p] = vn * sqrt(temp);
        }
    }
}

// Short columns (m - j <= 128, the k x n "wide" case, e.g. the 128 x 8192 projection B): one wave
// owns CPW consecutive positions and handles them four at a time, so eight independent column
// loads are in flight per lane and the reflector is read once per wave; reductions are DPP.
template <typename T, int CPW>
__global__ __launch_bounds__(256) void k_qr_apply_short(Mat<T> w, int64_t j, int pivot, const int64_t *jpvt, T *vn1, T *vn2, const T *tau) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t n = w.cols, m = w.rows;
    const int64_t p0 = j + 1 + ((int64_t)blockIdx.x * 4 + wv) * CPW;
    if (p0 >= n) return;  // whole wave
    const T tj = tau[j];
    const T *vcol = w.p + jpvt[j] * w.cs;
    const int64_t i0 = j + lane, i1 = j + lane + 64;
    const T v0 = (i0 < m) ? ((i0 == j) ? (T)1 : vcol[i0]) : (T)0;
    const T v1 = (i1 < m) ? vcol[i1] : (T)0;
    for (int64_t pb = p0; pb < p0 + CPW && pb < n; pb += 4) {
        T *xc[4];
        T x0[4], x1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t p = pb + u;
            xc[u] = w.p + jpvt[p < n ? p : n - 1] * w.cs;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            x0[u] = (i0 < m) ? xc[u][i0] : (T)0;
            x1[u] = (i1 < m) ? xc[u][i1] : (T)0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t p = pb + u;
            if (p >= n || p >= p0 + CPW) continue;  // wave-uniform
            if (tj != (T)0) {
                const T dot = wave_sum_dpp(fma(v0, x0[u], v1 * x1[u]));
                const T f = tj * dot;
                x0[u] -= f * v0;
                x1[u] -= f * v1;
                if (i0 < m) xc[u][i0] = x0[u];
                if (i1 < m) xc[u][i1] = x1[u];
            }
            if (pivot) {
                const T xj = read_lane(x0[u], 0);
                const T vn = vn1[p];
                if (vn != (T)0) {
                    T t = fabs(xj) / vn;
                    T temp = (T)1 - t * t;
                    temp = temp > (T)0 ? temp : (T)0;
                    T r = vn / vn2[p];
                    T temp2 = temp * r * r;
                    if (temp2 <= Num<T>::tol3z()) {
                        T ss = (i0 > j ? x0[u] * x0[u] : (T)0) + x1[u] * x1[u];
                        ss = wave_sum_dpp(ss);
                        if (lane == 0) { T nn = (j < m - 1) ? sqrt(ss) : (T)0; vn1[p] = nn; vn2[p] = nn; }
                    } else if (lane == 0) {
                        vn1[p] = vn * sqrt(temp);
                    }
                }
            }
        }
    }
}

// general fallback: any column length, two passes over memory (256 threads per column)
template <typename T>
__global__ __launch_bounds__(256) void k_qr_apply_general(Mat<T> w, int64_t j, int pivot, const int64_t *jpvt, T *vn1, T *vn2, const T *tau) {
    __shared__ T sh[4];
    const int64_t n = w.cols, m = w.rows;
    const int64_t p = j + 1 + blockIdx.x;
    if (p >= n) return;
    const T tj = tau[j];
    const T *vcol = w.p + jpvt[j] * w.cs;
    T *xcol = w.p + jpvt[p] * w.cs;
    if (tj != (T)0) {
        T dot = 0;
        for (int64_t i = j + threadIdx.x; i < m; i += 256) dot += ((i == j) ? (T)1 : vcol[i]) * xcol[i];
        dot = group_sum<T, 256>(dot, sh);
        const T f = tj * dot;
        for (int64_t i = j + threadIdx.x; i < m; i += 256) xcol[i] -= f * ((i == j) ? (T)1 : vcol[i]);
    }
    if (!pivot) return;
    __syncthreads();  // row j was written by thread 0 of this workgroup
    const T xj = xcol[j];
    const T vn = vn1[p];
    if (vn != (T)0) {
        T t = fabs(xj) / vn;
        T temp = (T)1 - t * t;
        temp = temp > (T)0 ? temp : (T)0;
        T r = vn / vn2[p];
        T temp2 = temp * r * r;
        if (temp2 <= Num<T>::tol3z()) {
            T ss = 0;
            for (int64_t i = j + 1 + threadIdx.x; i < m; i += 256) { T v = xcol[i]; ss += v * v; }
            ss = group_sum<T, 256>(ss, sh);
            if (threadIdx.x == 0) { T nn = (j < m - 1) ? sqrt(ss) : (T)0; vn1[p] = nn; vn2[p] = nn; }
        } else if (threadIdx.x == 0) {
            vn1[p] = vn * sqrt(temp);
        }
    }
}

template <typename T>
void geqp3_inplace(rc_context *c, Mat<T> w, int64_t kmax, bool pivot, int64_t *jpvt, T *tau, T *vn) {
    RC_REQUIRE(w.rs == 1, RC_LAYOUT_ERROR, "geqp3: working matrix must be column-major");
    const int64_t m = w.rows, n = w.cols;
    if (m == 0 || n == 0) return;
    kmax = std::min(kmax, std::min(m, n));
    ProfScope ps(c, "op:geqp3 %lldx%lld k=%lld pivot=%d", (long long)m, (long long)n, (long long)kmax, pivot ? 1 : 0);
    T *vn1 = vn, *vn2 = vn + n;
    const int pv = pivot ? 1 : 0;
    hipLaunchKernelGGL(k_qr_init<T>, dim3((unsigned)std::min<int64_t>(n, 65535)), dim3(256), 0, c->stream, w, pv, jpvt, vn1, vn2);
    for (int64_t j = 0; j < kmax; ++j) {
        hipLaunchKernelGGL(k_qr_pivot_reflect<T>, dim3(1), dim3(1024), 0, c->stream, w, j, pv, jpvt, vn1, vn2, tau);
        const int64_t rem_cols = n - j - 1;
        if (rem_cols <= 0) continue;
        const int64_t rem = m - j;
        const int64_t c64 = cdiv(rem, 64), c256 = cdiv(rem, 256);
#define RC_APPLY(TPC, MAXE)                                                                                       \
    hipLaunchKernelGGL((k_qr_apply<T, TPC, MAXE>), dim3((unsigned)cdiv(rem_cols, 256 / TPC)), dim3(256), 0,      \
                       c->stream, w, j, pv, jpvt, vn1, vn2, tau)
        if (c64 <= 2 && rem_cols >= 2048) {
            constexpr int CPW = 4;
            hipLaunchKernelGGL((k_qr_apply_short<T, CPW>), dim3((unsigned)cdiv(rem_cols, 4 * CPW)), dim3(256), 0, c->stream, w, j, pv, jpvt, vn1, vn2, tau);
        }
        else if (c64 <= 2) RC_APPLY(64, 2);
        else if (c64 <= 8) RC_APPLY(64, 8);
        else if (c256 <= 8) RC_APPLY(256, 8);
        else if (c256 <= 16) RC_APPLY(256, 16);
        else if (c256 <= 32) RC_APPLY(256, 32);
        else
            hipLaunchKernelGGL(k_qr_apply_general<T>, dim3((unsigned)rem_cols), dim3(256), 0, c->stream, w, j, pv, jpvt, vn1, vn2, tau);
#undef RC_APPLY
    }
}

// ---------------------------------------------------------------------------
// R extraction: r(i, p) = (i <= p) ? w(i, jpvt[p]) : 0   (?geqp3 output upper
// trapezoid in position order; IntoTriangular, pivoted_qr.rs:100-102)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_extract_r(Mat<T> w, const int64_t *jpvt, Mat<T> r) {
    const bool col_fast = (r.cs <= r.rs);
    const int64_t total = r.rows * r.cols;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t i, p;
        if (col_fast) { i = e / r.cols; p = e - i * r.cols; }
        else { p = e / r.rows; i = e - p * r.rows; }
        r.at(i, p) = (i <= p) ? w.p[jpvt[p] * w.cs + i] : (T)0;
    }
}
template <typename T>
void extract_r(rc_context *c, Mat<T> w, const int64_t *jpvt, Mat<T> r) {
    if (r.empty()) return;
    int grid = (int)std::min<int64_t>(cdiv(r.rows * r.cols, 256), 8192);
    hipLaunchKernelGGL(k_extract_r<T>, dim3(grid), dim3(256), 0, c->stream, w, jpvt, r);
}

// ---------------------------------------------------------------------------
// form Q (?org2r semantics): column cq of Q = H_0 ... H_{k-1} e_cq; for e_cq
// only H_j with j <= cq act non-trivially, so each output column is an
// independent chain -> one workgroup per column, column in registers.
// ---------------------------------------------------------------------------
template <typename T, int MAXE>
__global__ __launch_bounds__(256) void k_form_q(Mat<T> w, const int64_t *jpvt, const T *tau, int64_t k, Mat<T> qw) {
    __shared__ T sh[4];
    const int64_t m = w.rows;
    const int64_t cq = blockIdx.x;
    const int tid = threadIdx.x;
    T x[MAXE];
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        int64_t i = tid + (int64_t)e * 256;
        x[e] = (i == cq) ? (T)1 : (T)0;
    }
    for (int64_t j = (cq < k - 1 ? cq : k - 1); j >= 0; --j) {
        const T tj = tau[j];
        if (tj == (T)0) continue;
        const T *vcol = w.p + jpvt[j] * w.cs;
        T v[MAXE];
        T dot = 0;
#pragma unroll
        for (int e = 0; e < MAXE; ++e) {
            // branch-free (see k_qr_apply): masked lanes read row j and are zeroed by a select
            const int i = tid + e * 256;
            const bool ok = i < (int)m && i >= (int)j;
            const T vv = vcol[ok ? i : (int)j];
            v[e] = ok ? (i == (int)j ? (T)1 : vv) : (T)0;
            dot = fma(v[e], x[e], dot);
        }
        dot = group_sum<T, 256>(dot, sh);
        const T f = tj * dot;
#pragma unroll
        for (int e = 0; e < MAXE; ++e) x[e] -= f * v[e];
    }
    T *out = qw.p + cq * qw.cs;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        int64_t i = tid + (int64_t)e * 256;
        if (i < m) out[i] = x[e];
    }
}
// general: the column lives in global memory (qw), any m
template <typename T>
__global__ __launch_bounds__(256) void k_form_q_general(Mat<T> w, const int64_t *jpvt, const T *tau, int64_t k, Mat<T> qw) {
    __shared__ T sh[4];
    const int64_t m = w.rows;
    const int64_t cq = blockIdx.x;
    const int tid = threadIdx.x;
    T *x = qw.p + cq * qw.cs;
    for (int64_t i = tid; i < m; i += 256) x[i] = (i == cq) ? (T)1 : (T)0;
    for (int64_t j = (cq < k - 1 ? cq : k - 1); j >= 0; --j) {
        const T tj = tau[j];
        if (tj == (T)0) continue;
        const T *vcol = w.p + jpvt[j] * w.cs;
        // thread tid owns the rows i = tid (mod 256) for the WHOLE kernel (the initialisation above included): x[i] is only ever
        // touched by its owner, so no barrier is needed between a step's update and the next step's reads.  (Until round 3 the
        // loops started at j + tid: the owner of a row changed from step to step without a barrier in between -- a race between waves.)
        const int64_t i0 = tid >= j ? tid : tid + ((j - tid + 255) / 256) * 256;
        T dot = 0;
        for (int64_t i = i0; i < m; i += 256) dot += ((i == j) ? (T)1 : vcol[i]) * x[i];
        dot = group_sum<T, 256>(dot, sh);
        const T f = tj * dot;
        for (int64_t i = i0; i < m; i += 256) x[i] -= f * ((i == j) ? (T)1 : vcol[i]);
    }
}

// short matrices (m <= 128, at most 128 columns of Q): ONE workgroup.  All k reflectors are staged in LDS
// with one bulk load; 8 lanes per output column keep the column in registers, so the k dependent
// reflector applications cost LDS broadcasts only -- one CU instead of one workgroup per column.
template <typename T, int NE>
__global__ __launch_bounds__(1024) void k_form_q_small(Mat<T> w, const int64_t *jpvt, const T *tau, int k, Mat<T> qw) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *V = reinterpret_cast<T *>(smem_raw);  // V[j * 8 * NE + i] = v_j(i): zero above the diagonal, one on it
    T *tl = V + (size_t)k * 8 * NE;
    const int m = (int)w.rows;
    constexpr int LD = 8 * NE;
    for (int e = threadIdx.x; e < k * LD; e += blockDim.x) {
        const int j = e / LD, i = e - j * LD;
        V[e] = (i < j || i >= m) ? (T)0 : (i == j) ? (T)1 : w.p[jpvt[j] * w.cs + i];
    }
    for (int j = threadIdx.x; j < k; j += blockDim.x) tl[j] = tau[j];
    __syncthreads();
    const int l8 = threadIdx.x & 7, cq = threadIdx.x >> 3;
    if (cq >= qw.cols) return;
    T x[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) x[e] = (l8 + 8 * e == cq) ? (T)1 : (T)0;
    for (int j = cq < k - 1 ? cq : k - 1; j >= 0; --j) {
        const T tj = tl[j];
        if (tj == (T)0) continue;
        const T *vj = V + (size_t)j * LD + l8;
        T v[NE];
        T dot = 0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            v[e] = vj[8 * e];
            dot = fma(v[e], x[e], dot);
        }
        dot = group_sum_dpp<8>(dot);
        const T f = tj * dot;
#pragma unroll
        for (int e = 0; e < NE; ++e) x[e] = fma(-f, v[e], x[e]);
    }
    T *out = qw.p + (int64_t)cq * qw.cs;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int i = l8 + 8 * e;
        if (i < m) out[i] = x[e];
    }
}

// ---- compact-WY form-Q for tall matrices: Q = (I - V T V^T) [I ; 0] as three GEMMs ----
// vm(i, j) = v_j(i): zeros above the diagonal, one on it, reflector below
template <typename T>
__global__ __launch_bounds__(256) void k_extract_v(Mat<T> w, const int64_t *jpvt, Mat<T> vm) {
    const int64_t j = blockIdx.y;
    const T *col = w.p + jpvt[j] * w.cs;
    T *out = vm.p + j * vm.cs;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < vm.rows; i += (int64_t)gridDim.x * 256)
        out[i] = (i < j) ? (T)0 : (i == j) ? (T)1 : col[i];
}
// T = D (I + striu(S) D)^{-1}, D = diag(tau), S = V^T V  (tau_j == 0, i.e. H_j = I, needs
// no special case).  Built column by column with the ?larft recurrence
//   T[0:j, j] = -tau_j * T[0:j, 0:j] * S[0:j, j],  T[j, j] = tau_j
// by ONE workgroup: thread i owns row i of T (kept in global/L2, written once per
// column), the current S column is broadcast through LDS.
template <typename T>
__global__ __launch_bounds__(1024) void k_build_t(Mat<T> sgram, const T *tau, Mat<T> tm) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *svec = reinterpret_cast<T *>(smem_raw);  // k entries
    T *trow = svec + tm.rows;                   // k x (k | 1) : T in LDS when it fits, else unused
    const int k = (int)tm.rows;
    const int ld = k | 1;
    const int tid = threadIdx.x;
    for (int e = tid; e < k * ld; e += 1024) trow[e] = 0;
    __syncthreads();
    for (int j = 0; j < k; ++j) {
        for (int l = tid; l < j; l += 1024) svec[l] = sgram.at(l, j);
        __syncthreads();
        const T tj = tau[j];
        for (int i = tid; i <= j; i += 1024) {
            if (i == j) {
                trow[i * ld + j] = tj;
            } else {
                T acc = 0;
                for (int l = i; l < j; ++l) acc += trow[i * ld + l] * svec[l];
                trow[i * ld + j] = -tj * acc;
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < k * k; e += 1024) {
        int i = e % k, j = e / k;
        tm.at(i, j) = trow[i * ld + j];
    }
}

template <typename T>
void form_q(rc_context *c, Mat<T> w, const int64_t *jpvt, const T *tau, int64_t k, Mat<T> qw) {
    RC_REQUIRE(w.rs == 1 && qw.rs == 1 && qw.rows == w.rows, RC_LAYOUT_ERROR, "form_q: column-major operands required");
    if (qw.empty()) return;
    const int64_t m = w.rows;
    const unsigned grid = (unsigned)qw.cols;
    if (k <= 0) { fill_identity(c, qw); return; }
    ProfScope ps(c, "op:form_q %lldx%lld k=%lld", (long long)m, (long long)qw.cols, (long long)k);
    if (m >= 2048 && k >= 16 && ((size_t)k + (size_t)k * (k | 1)) * sizeof(T) <= 160 * 1024 - 1024) {
        // tall: three MFMA GEMMs instead of k dependent reflector applications per column
        ArenaMark mark(c);
        const int64_t kq = qw.cols;
        Mat<T> vm = colmajor(c->alloc<T>((size_t)even_ld(m) * k), m, k, even_ld(m));
        hipLaunchKernelGGL(k_extract_v<T>, dim3((unsigned)std::min<int64_t>(cdiv(m, 256), 64), (unsigned)k), dim3(256), 0, c->stream, w, jpvt, vm);
        Mat<T> sg = rowmajor(c->alloc<T>((size_t)k * even_ld(k)), k, k, even_ld(k));
        gemm<T>(c, 1, vm.t(), vm, 0, sg);
        Mat<T> tm = colmajor(c->alloc<T>((size_t)even_ld(k) * k), k, k, even_ld(k));
        {
            const size_t lds = ((size_t)k + (size_t)k * (k | 1)) * sizeof(T);
            RC_REQUIRE(lds <= 160 * 1024 - 1024, RC_INVALID_ARGUMENT, "form_q: block of %lld reflectors does not fit the LDS T-builder", (long long)k);
            auto kern = k_build_t<T>;
            static bool attr_set[64] = {};
            if (!attr_set[c->device & 63]) {
                RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
                attr_set[c->device & 63] = true;
            }
            hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, c->stream, sg, tau, tm);
        }
        Mat<T> w2 = rowmajor(c->alloc<T>((size_t)k * even_ld(kq)), k, kq, even_ld(kq));
        gemm<T>(c, 1, tm, vm.sub(0, std::min(kq, m), 0, k).t(), 0, w2);
        fill_identity(c, qw);
        gemm<T>(c, -1, vm, w2, 1, qw);
        return;
    }
    if (m <= 128 && qw.cols <= 128 && k <= 128) {
        const unsigned threads = (unsigned)std::max<int64_t>(256, ((qw.cols * 8 + 63) / 64) * 64);
        const int ne = m <= 64 ? 8 : 16;
        const size_t lds = ((size_t)k * 8 * ne + (size_t)k) * sizeof(T);
        auto k8 = k_form_q_small<T, 8>;
        auto k16 = k_form_q_small<T, 16>;
        static bool attr_set[64] = {};
        if (!attr_set[c->device & 63]) {
            RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k8), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
            RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k16), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
            attr_set[c->device & 63] = true;
        }
        if (ne == 8) hipLaunchKernelGGL(k8, dim3(1), dim3(threads), lds, c->stream, w, jpvt, tau, (int)k, qw);
        else hipLaunchKernelGGL(k16, dim3(1), dim3(threads), lds, c->stream, w, jpvt, tau, (int)k, qw);
        return;
    }
    if (m <= 256 * 2) hipLaunchKernelGGL((k_form_q<T, 2>), dim3(grid), dim3(256), 0, c->stream, w, jpvt, tau, k, qw);
    else if (m <= 256 * 8) hipLaunchKernelGGL((k_form_q<T, 8>), dim3(grid), dim3(256), 0, c->stream, w, jpvt, tau, k, qw);
    else if (m <= 256 * 32) hipLaunchKernelGGL((k_form_q<T, 32>), dim3(grid), dim3(256), 0, c->stream, w, jpvt, tau, k, qw);
    else hipLaunchKernelGGL(k_form_q_general<T>, dim3(grid), dim3(256), 0, c->stream, w, jpvt, tau, k, qw);
}

// ---------------------------------------------------------------------------
// T X = B, T upper triangular k x k (any strides), B k x nrhs in place.
// One thread per right-hand side; consecutive threads take consecutive RHS so
// the accesses coalesce when B is row-major (k x nrhs, the Z / X^T blocks).
// Replaces the reference's per-column ?trtrs loop (qr.rs:290-301, :384-395).
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_trsm_upper(Mat<T> t, Mat<T> b) {
    // Blocked back substitution: 16 rows of X per thread in registers, the 16 x 16
    // tiles of the triangle staged through LDS (every thread reads the same tile
    // entry: an LDS broadcast), the 16 X values of a finished block re-read with 16
    // independent, coalesced global loads.
    constexpr int NB = 16;
    __shared__ T tile[NB][NB + 1];
    const int64_t col = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const bool active = col < b.cols;
    const int64_t k = t.rows;
    const int64_t nblk = (k + NB - 1) / NB;
    const int ti = threadIdx.x / NB, tj = threadIdx.x % NB;
    for (int64_t bi = nblk - 1; bi >= 0; --bi) {
        const int64_t r0 = bi * NB;
        T acc[NB];
#pragma unroll
        for (int ii = 0; ii < NB; ++ii) acc[ii] = (active && r0 + ii < k) ? b.at(r0 + ii, col) : (T)0;
        for (int64_t bj = nblk - 1; bj >= bi; --bj) {
            const int64_t c0 = bj * NB;
            __syncthreads();
            tile[ti][tj] = (r0 + ti < k && c0 + tj < k) ? t.at(r0 + ti, c0 + tj) : (T)0;
            __syncthreads();
            if (bj > bi) {
                T x[NB];
#pragma unroll
                for (int jj = 0; jj < NB; ++jj) x[jj] = (active && c0 + jj < k) ? b.at(c0 + jj, col) : (T)0;
#pragma unroll
                for (int jj = 0; jj < NB; ++jj)
#pragma unroll
                    for (int ii = 0; ii < NB; ++ii) acc[ii] -= tile[ii][jj] * x[jj];
            } else {
#pragma unroll
                for (int ii = NB - 1; ii >= 0; --ii) {
                    if (r0 + ii < k) {
                        acc[ii] /= tile[ii][ii];
#pragma unroll
                        for (int i2 = 0; i2 < ii; ++i2) acc[i2] -= tile[i2][ii] * acc[ii];
                    }
                }
            }
        }
        if (active) {
#pragma unroll
            for (int ii = 0; ii < NB; ++ii)
                if (r0 + ii < k) b.at(r0 + ii, col) = acc[ii];
        }
    }
}
template <typename T>
void trsm_upper(rc_context *c, Mat<T> t, Mat<T> b) {
    RC_REQUIRE(t.rows == t.cols && t.rows == b.rows, RC_INVALID_ARGUMENT, "trsm: shape mismatch");
    if (b.empty()) return;
    ProfScope ps(c, "op:trsm_upper k=%lld nrhs=%lld", (long long)t.rows, (long long)b.cols);
    hipLaunchKernelGGL(k_trsm_upper<T>, dim3((unsigned)cdiv(b.cols, 256)), dim3(256), 0, c->stream, t, b);
}

// ---------------------------------------------------------------------------
// Column ID straight from the factored working matrix (the unit of work of batches, QR::compute_from -> compress(RANK(k)) ->
// column_id, /root/reference/src/qr.rs:270-309, examples/interpolative_decomposition.rs:25-32) in TWO launches:
//   Z = [I | R11^-1 R12] P^T : k_id_z -- k_trsm_upper's blocked back substitution (same tiles, same order of operations: the same
//       bits) with the triangle and the right-hand sides read from the ?geqp3-format matrix through jpvt and every solved column
//       written to its FINAL place z[:, jpvt[p]] (apply_permutation(COLINV) folded in; no R copy, no identity fill, no inverse
//       permutation, no gather);
//   C = Q R11 = (A P)[:, :k] : the selected columns of A themselves (gather_cols of the ORIGINAL matrix) -- the reference forms
//       the product Q R11, which equals those columns up to rounding (src/qr.rs:287-288 "first_part"); forming Q (?orgqr: three
//       GEMMs per panel) only to multiply it back with R11 was a third of the launches of a cfg5 matrix.
// ---------------------------------------------------------------------------
// FROM_R: w is the k x n factor R in pivoted order (r(i, p), any strides) instead of the ?geqp3-format matrix
template <typename T, bool FROM_R>
__global__ __launch_bounds__(256) void k_id_z(Mat<T> w, const int64_t *jpvt, int64_t k, Mat<T> z) {
    constexpr int NB = 16;
    __shared__ T tile[NB][NB + 1];
    const int64_t n = w.cols;
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;  // position in the pivoted order
    const bool inside = p < n;
    const int64_t dc = inside ? jpvt[p] : 0;                            // where column p of [I | R11^-1 R12] goes
    const bool active = inside && p >= k;                               // a right-hand side (the first k columns are the identity)
    const T *bcol = FROM_R ? w.p + (inside ? p : 0) * w.cs : w.p + dc * w.cs;  // R12[:, p] (rows 0 .. k-1 of the physical column)
    const int64_t brs = FROM_R ? w.rs : 1;
    const int64_t nblk = (k + NB - 1) / NB;
    const int ti = threadIdx.x / NB, tj = threadIdx.x % NB;
    if (inside && p < k)
        for (int64_t i = 0; i < k; ++i) z.at(i, dc) = (i == p) ? (T)1 : (T)0;
    for (int64_t bi = nblk - 1; bi >= 0; --bi) {
        const int64_t r0 = bi * NB;
        T acc[NB];
#pragma unroll
        for (int ii = 0; ii < NB; ++ii) acc[ii] = (active && r0 + ii < k) ? bcol[(r0 + ii) * brs] : (T)0;
        for (int64_t bj = nblk - 1; bj >= bi; --bj) {
            const int64_t c0 = bj * NB;
            __syncthreads();
            {
                const int64_t i = r0 + ti, l = c0 + tj;
                tile[ti][tj] = (i < k && l < k && i <= l) ? (FROM_R ? w.at(i, l) : w.p[jpvt[l] * w.cs + i]) : (T)0;
            }
            __syncthreads();
            if (bj > bi) {
                T x[NB];
#pragma unroll
                for (int jj = 0; jj < NB; ++jj) x[jj] = (active && c0 + jj < k) ? z.at(c0 + jj, dc) : (T)0;  // this thread's own finished block
#pragma unroll
                for (int jj = 0; jj < NB; ++jj)
#pragma unroll
                    for (int ii = 0; ii < NB; ++ii) acc[ii] -= tile[ii][jj] * x[jj];
            } else {
#pragma unroll
                for (int ii = NB - 1; ii >= 0; --ii) {
                    if (r0 + ii < k) {
                        acc[ii] /= tile[ii][ii];
#pragma unroll
                        for (int i2 = 0; i2 < ii; ++i2) acc[i2] -= tile[i2][ii] * acc[ii];
                    }
                }
            }
        }
        if (active) {
#pragma unroll
            for (int ii = 0; ii < NB; ++ii)
                if (r0 + ii < k) z.at(r0 + ii, dc) = acc[ii];
        }
    }
}
// a: the original m x n matrix (any layout); w: its factorization in the ?geqp3 format (column-major, columns in place, k steps);
// jpvt: position -> physical column; cm: m x k, z: k x n
template <typename T>
void column_id_from_qrcp(rc_context *c, Mat<T> a, Mat<T> w, int64_t k, const int64_t *jpvt, Mat<T> cm, Mat<T> z) {
    RC_REQUIRE(w.rs == 1 && a.rows == w.rows && a.cols == w.cols && cm.rows == a.rows && cm.cols == k && z.rows == k && z.cols == a.cols && k <= std::min(a.rows, a.cols),
               RC_INVALID_ARGUMENT, "column_id_from_qrcp: shape mismatch");
    if (a.cols == 0 || k == 0) return;
    ProfScope ps(c, "op:column_id_from_qrcp %lldx%lld k=%lld", (long long)a.rows, (long long)a.cols, (long long)k);
    hipLaunchKernelGGL((k_id_z<T, false>), dim3((unsigned)cdiv(a.cols, 256)), dim3(256), 0, c->stream, w, jpvt, k, z);
    gather_cols(c, a, jpvt, cm);  // C[:, i] = A[:, jpvt[i]], i < k
}
// Z = [I | R11^-1 R12] P^T from the k x n factor r (pivoted order) in one launch: the bits of
// fill_identity + copy + trsm_upper + invert_perm + gather_cols (qr_column_id's former chain)
template <typename T>
void id_z_from_r(rc_context *c, Mat<T> r, int64_t k, const int64_t *ind, Mat<T> z) {
    RC_REQUIRE(r.rows == k && z.rows == k && z.cols == r.cols && k <= r.cols, RC_INVALID_ARGUMENT, "id_z_from_r: shape mismatch");
    if (r.cols == 0 || k == 0) return;
    ProfScope ps(c, "op:id_z_from_r k=%lld n=%lld", (long long)k, (long long)r.cols);
    hipLaunchKernelGGL((k_id_z<T, true>), dim3((unsigned)cdiv(r.cols, 256)), dim3(256), 0, c->stream, r, ind, k, z);
}

#define RC_INST(T)                                                                                         \
    template void column_id_from_qrcp<T>(rc_context *, Mat<T>, Mat<T>, int64_t, const int64_t *, Mat<T>, Mat<T>); \
    template void id_z_from_r<T>(rc_context *, Mat<T>, int64_t, const int64_t *, Mat<T>);                 \
    template void geqp3_inplace<T>(rc_context *, Mat<T>, int64_t, bool, int64_t *, T *, T *);             \
    template void extract_r<T>(rc_context *, Mat<T>, const int64_t *, Mat<T>);                             \
    template void form_q<T>(rc_context *, Mat<T>, const int64_t *, const T *, int64_t, Mat<T>);           \
    template void trsm_upper<T>(rc_context *, Mat<T>, Mat<T>);
RC_INST(double)
RC_INST(float)
#undef RC_INST

}  // namespace rc

// ===========================================================================
// "Lazy" pivoted QR for SHORT-WIDE matrices (m <= 256 rows, n >> m; the k x n projection B
// of the range finder).  The trailing matrix is never updated: the m x m orthogonal factor
// Qacc = H_0 ... H_j is kept explicitly and row j of R is produced as q_j^T B, so B is
// READ-ONLY (it stays valid in every XCD's L2 / the Infinity Cache; the eager scheme
// rewrote all 8 MB of it at every step and paid the cross-XCD write visibility each launch).
//   per step: one serial workgroup (pivot search, x = Qacc^T b_p, ?larfg, Qacc <- Qacc H_j)
//             one streaming kernel  (R[j, c] = Qacc[:, j]^T b_c, LAPACK norm down-date)
// Same ?laqp2 semantics as the eager chain (first-max pivot, tol3z recompute with the exact
// trailing norm); Q needs no separate ?orgqr pass.
// ===========================================================================
namespace rc {

template <typename T>
__global__ __launch_bounds__(256) void k_wq_init(Mat<T> qacc, Mat<T> rphys) {
    const int64_t tq = qacc.rows * qacc.cols, tr = rphys.rows * rphys.cols;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tq + tr; e += (int64_t)gridDim.x * blockDim.x) {
        if (e < tq) { int64_t i = e % qacc.rows, c2 = e / qacc.rows; qacc.p[c2 * qacc.cs + i] = (i == c2) ? (T)1 : (T)0; }
        else rphys.p[e - tq] = 0;
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void k_wq_pivot(Mat<T> b, Mat<T> qacc, Mat<T> rphys, int64_t j, int64_t *jpvt, T *vn1, T *vn2, T *tau) {
    // The active part of Qacc (columns j..m-1) is staged in LDS for the step; 8 lanes (a DPP
    // half row) share every dot product, so the three O(m^2) passes are ~16 FMAs per lane each.
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int m = (int)b.rows;
    const int ldq = m | 1;
    long long *shi = reinterpret_cast<long long *>(smem_raw);  // 16
    T *shv = reinterpret_cast<T *>(shi + 16);                  // 16
    T *Q = shv + 16;  // Q[i * ldq + r] = Qacc(r, i), i >= j only
    T *bvec = Q + (size_t)m * ldq;
    T *xvec = bvec + m;
    T *vvec = xvec + m;
    T *wvec = vvec + m;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l8 = tid & 7, g8 = tid >> 3;  // 128 groups of 8 lanes
    const int64_t n = b.cols;
    const int jj = (int)j;
    // stage Qacc[:, j:] (coalesced: rows fastest)
    for (int e = tid; e < (m - jj) * m; e += 1024) {
        const int i = jj + e / m, r = e % m;
        Q[i * ldq + r] = qacc.p[(int64_t)i * qacc.cs + r];
    }
    {   // pivot: first maximum of vn1[j..n)
        T best = (T)-1;
        long long bi = 0x7fffffffffffffffLL;
        for (int64_t p = j + tid; p < n; p += 1024) {
            T v = fabs(vn1[p]);
            if (v > best) { best = v; bi = p; }
        }
        const T mx = wave_max_dpp(best);
        long long cand = (best == mx) ? bi : 0x7fffffffffffffffLL;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { long long o2 = __shfl_xor(cand, off, 64); cand = o2 < cand ? o2 : cand; }
        if (lane == 0) { shv[wv] = mx; shi[wv] = cand; }
        __syncthreads();
        if (tid == 0) {
            T bb = shv[0]; long long bidx = shi[0];
            for (int k2 = 1; k2 < 16; ++k2)
                if (shv[k2] > bb || (shv[k2] == bb && shi[k2] < bidx)) { bb = shv[k2]; bidx = shi[k2]; }
            const int64_t pvt = (bidx >= j && bidx < n) ? (int64_t)bidx : j;
            if (pvt != j) {
                int64_t t = jpvt[pvt]; jpvt[pvt] = jpvt[j]; jpvt[j] = t;
                vn1[pvt] = vn1[j];
                vn2[pvt] = vn2[j];
            }
        }
        __syncthreads();
    }
    const int64_t c = jpvt[j];
    if (tid < m) bvec[tid] = b.p[c * b.cs + tid];
    __syncthreads();
    // x_i = Qacc[:, i]^T b_c for i >= j  (the not yet reduced part of the pivot column)
    for (int i = jj + g8; i < m; i += 128) {
        T acc = 0;
        for (int r = l8; r < m; r += 8) acc = fma(Q[i * ldq + r], bvec[r], acc);
        acc = group_sum_dpp<8>(acc);
        if (l8 == 0) xvec[i] = acc;
    }
    __syncthreads();
    // ?larfg on x[j..m)
    T ss = (tid > jj && tid < m) ? xvec[tid] * xvec[tid] : (T)0;
    ss = wave_sum_dpp(ss);
    if (lane == 0) shv[wv] = ss;
    __syncthreads();
    T ssq = 0;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) ssq += shv[k2];
    const T xnorm = sqrt(ssq);
    const T alpha = xvec[jj];
    T tj = 0, beta = alpha;
    if (xnorm != (T)0) {
        beta = -copysign(hypot(alpha, xnorm), alpha);
        tj = (beta - alpha) / beta;
    }
    if (tid == 0) { tau[j] = tj; rphys.p[j * rphys.rs + c] = beta; }
    if (tj == (T)0) return;  // H_j = I: Qacc unchanged
    const T scal = (T)1 / (alpha - beta);
    if (tid >= jj && tid < m) vvec[tid] = (tid == jj) ? (T)1 : xvec[tid] * scal;
    __syncthreads();
    // w = Qacc[:, j:] v ; 8 lanes per row r
    for (int r = g8; r < m; r += 128) {
        T acc = 0;
        for (int i = jj + l8; i < m; i += 8) acc = fma(Q[i * ldq + r], vvec[i], acc);
        acc = group_sum_dpp<8>(acc);
        if (l8 == 0) wvec[r] = tj * acc;
    }
    __syncthreads();
    // Qacc[:, j:] -= (tau w) v^T, written straight back to global (coalesced: rows fastest)
    for (int e = tid; e < (m - jj) * m; e += 1024) {
        const int i = jj + e / m, r = e % m;
        qacc.p[(int64_t)i * qacc.cs + r] = Q[i * ldq + r] - wvec[r] * vvec[i];
    }
}

// R[j, c] = q_j^T b_c for every column still unpivoted + LAPACK partial-norm down-date.
// LPC lanes per column, NE = ceil(m / LPC) rows per lane.
template <typename T, int LPC, int NE>
__global__ __launch_bounds__(256) void k_wq_row(Mat<T> b, Mat<T> qacc, Mat<T> rphys, int64_t j, const int64_t *jpvt, T *vn1, T *vn2) {
    const int ll = threadIdx.x % LPC, grp = threadIdx.x / LPC;
    const int m = (int)b.rows;
    const int64_t n = b.cols;
    const int64_t p = j + 1 + (int64_t)blockIdx.x * (256 / LPC) + grp;
    const bool active = p < n;
    const int64_t c = jpvt[active ? p : n - 1];
    const T *qj = qacc.p + j * qacc.cs;
    const T *bc = b.p + c * b.cs;
    T x[NE];
    T dot = 0;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int i = ll + LPC * e;
        x[e] = (i < m) ? bc[i] : (T)0;
        dot = fma((i < m) ? qj[i] : (T)0, x[e], dot);
    }
    dot = group_sum_dpp<LPC>(dot);
    if (!active) return;
    if (ll == 0) rphys.p[j * rphys.rs + c] = dot;
    const T vn = vn1[p];
    if (vn != (T)0) {
        T t = fabs(dot) / vn;
        T temp = (T)1 - t * t;
        temp = temp > (T)0 ? temp : (T)0;
        T r = vn / vn2[p];
        T temp2 = temp * r * r;
        if (temp2 <= Num<T>::tol3z()) {
            // exact trailing norm: || (Qacc^T b_c)[j+1:] ||
            T ssq = 0;
            for (int i2 = (int)j + 1; i2 < m; ++i2) {
                const T *qi = qacc.p + (int64_t)i2 * qacc.cs;
                T d2 = 0;
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const int i = ll + LPC * e;
                    d2 = fma((i < m) ? qi[i] : (T)0, x[e], d2);
                }
                d2 = group_sum_dpp<LPC>(d2);
                ssq = fma(d2, d2, ssq);
            }
            if (ll == 0) { T nn = (j < m - 1) ? sqrt(ssq) : (T)0; vn1[p] = nn; vn2[p] = nn; }
        } else if (ll == 0) {
            vn1[p] = vn * sqrt(temp);
        }
    }
}

// Vectorised variant: every lane reads its rows as 2-element vectors (16 B for f64), so one
// 8-lane group fetches a column in 128-byte segments.  Needs even leading dimensions and
// 2-element aligned bases (true for the library's own temporaries).
template <typename T> struct Vec2T;
template <> struct Vec2T<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct Vec2T<float> { typedef float type __attribute__((ext_vector_type(2))); };

template <typename T, int LPC, int NV>
__global__ __launch_bounds__(256) void k_wq_row_v2(Mat<T> b, Mat<T> qacc, Mat<T> rphys, int64_t j, const int64_t *jpvt, T *vn1, T *vn2) {
    typedef typename Vec2T<T>::type V2;
    const int ll = threadIdx.x % LPC, grp = threadIdx.x / LPC;
    const int m = (int)b.rows;
    const int64_t n = b.cols;
    const int64_t p = j + 1 + (int64_t)blockIdx.x * (256 / LPC) + grp;
    const bool active = p < n;
    const int64_t c = jpvt[active ? p : n - 1];
    const T *qj = qacc.p + j * qacc.cs;
    const T *bc = b.p + c * b.cs;
    const T vn = vn1[active ? p : n - 1], vnb = vn2[active ? p : n - 1];  // issued early: independent of the dot product
    V2 x[NV];
    T dot = 0;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const int i = 2 * (ll + LPC * e);
        V2 q2 = V2{0, 0};
        x[e] = V2{0, 0};
        if (i + 1 < m) { x[e] = *reinterpret_cast<const V2 *>(bc + i); q2 = *reinterpret_cast<const V2 *>(qj + i); }
        else if (i < m) { x[e][0] = bc[i]; q2[0] = qj[i]; }
        dot = fma(q2[0], x[e][0], dot);
        dot = fma(q2[1], x[e][1], dot);
    }
    dot = group_sum_dpp<LPC>(dot);
    if (!active) return;
    if (ll == 0) rphys.p[j * rphys.rs + c] = dot;
    if (vn != (T)0) {
        T t = fabs(dot) / vn;
        T temp = (T)1 - t * t;
        temp = temp > (T)0 ? temp : (T)0;
        T r = vn / vnb;
        T temp2 = temp * r * r;
        if (temp2 <= Num<T>::tol3z()) {
            T ssq = 0;
            for (int i2 = (int)j + 1; i2 < m; ++i2) {
                const T *qi = qacc.p + (int64_t)i2 * qacc.cs;
                T d2 = 0;
#pragma unroll
                for (int e = 0; e < NV; ++e) {
                    const int i = 2 * (ll + LPC * e);
                    if (i < m) d2 = fma(qi[i], x[e][0], d2);
                    if (i + 1 < m) d2 = fma(qi[i + 1], x[e][1], d2);
                }
                d2 = group_sum_dpp<LPC>(d2);
                ssq = fma(d2, d2, ssq);
            }
            if (ll == 0) { T nn = (j < m - 1) ? sqrt(ssq) : (T)0; vn1[p] = nn; vn2[p] = nn; }
        } else if (ll == 0) {
            vn1[p] = vn * sqrt(temp);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_wq_extract(Mat<T> rphys, const int64_t *jpvt, Mat<T> r) {
    const bool col_fast = (r.cs <= r.rs);
    const int64_t total = r.rows * r.cols;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t i, p;
        if (col_fast) { i = e / r.cols; p = e - i * r.cols; }
        else { p = e / r.rows; i = e - p * r.rows; }
        r.at(i, p) = (i <= p) ? rphys.p[i * rphys.rs + jpvt[p]] : (T)0;
    }
}

template <typename T>
static size_t wq_pivot_lds(int64_t m) { return ((size_t)m * (m | 1) + 4 * (size_t)m + 16) * sizeof(T) + 16 * sizeof(long long); }
template <typename T>
bool wide_lazy_supported(int64_t m, int64_t n) { return m >= 2 && m <= 256 && n >= 4 * m && n >= 256 && wq_pivot_lds<T>(m) <= 160 * 1024 - 1024; }

// b: m x n column-major (NOT modified); q: m x kq (any strides, may be empty); r: kmax x n (may be empty)
template <typename T>
void geqp3_wide_lazy(rc_context *c, Mat<T> b, int64_t kmax, int64_t *jpvt, Mat<T> q, Mat<T> r) {
    RC_REQUIRE(b.rs == 1, RC_LAYOUT_ERROR, "geqp3_wide_lazy: column-major input required");
    const int64_t m = b.rows, n = b.cols;
    kmax = std::min(kmax, std::min(m, n));
    ProfScope ps(c, "op:geqp3_wide_lazy %lldx%lld k=%lld", (long long)m, (long long)n, (long long)kmax);
    ArenaMark mark(c);
    T *vn = c->alloc<T>((size_t)(2 * n));
    T *tau = c->alloc<T>((size_t)std::max<int64_t>(kmax, 1));
    Mat<T> qacc = colmajor(c->alloc<T>((size_t)even_ld(m) * m), m, m, even_ld(m));
    Mat<T> rphys = rowmajor(c->alloc<T>((size_t)kmax * n), kmax, n, n);
    T *vn1 = vn, *vn2 = vn + n;
    hipLaunchKernelGGL(k_qr_init<T>, dim3((unsigned)std::min<int64_t>(n, 65535)), dim3(256), 0, c->stream, b, 1, jpvt, vn1, vn2);
    hipLaunchKernelGGL(k_wq_init<T>, dim3((unsigned)std::min<int64_t>(cdiv(m * m + kmax * n, 256), 4096)), dim3(256), 0, c->stream, qacc, rphys);
    auto pk = k_wq_pivot<T>;
    static bool attr_set[64] = {};
    if (!attr_set[c->device & 63]) {
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pk), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        attr_set[c->device & 63] = true;
    }
    for (int64_t j = 0; j < kmax; ++j) {
        hipLaunchKernelGGL(pk, dim3(1), dim3(1024), wq_pivot_lds<T>(m), c->stream, b, qacc, rphys, j, jpvt, vn1, vn2, tau);
        const int64_t rem = n - j - 1;
        if (rem <= 0) continue;
        const bool vec_ok = (b.cs % 2) == 0 && (reinterpret_cast<uintptr_t>(b.p) % (2 * sizeof(T))) == 0;
        if (vec_ok && m <= 128) hipLaunchKernelGGL((k_wq_row_v2<T, 8, 8>), dim3((unsigned)cdiv(rem, 32)), dim3(256), 0, c->stream, b, qacc, rphys, j, jpvt, vn1, vn2);
        else if (m <= 64) hipLaunchKernelGGL((k_wq_row<T, 8, 8>), dim3((unsigned)cdiv(rem, 32)), dim3(256), 0, c->stream, b, qacc, rphys, j, jpvt, vn1, vn2);
        else if (m <= 128) hipLaunchKernelGGL((k_wq_row<T, 8, 16>), dim3((unsigned)cdiv(rem, 32)), dim3(256), 0, c->stream, b, qacc, rphys, j, jpvt, vn1, vn2);
        else hipLaunchKernelGGL((k_wq_row<T, 16, 16>), dim3((unsigned)cdiv(rem, 16)), dim3(256), 0, c->stream, b, qacc, rphys, j, jpvt, vn1, vn2);
    }
    if (!r.empty()) hipLaunchKernelGGL(k_wq_extract<T>, dim3((unsigned)std::min<int64_t>(cdiv(r.rows * r.cols, 256), 8192)), dim3(256), 0, c->stream, rphys, jpvt, r);
    if (!q.empty()) copy_mat(c, qacc.sub(0, m, 0, q.cols), q);
}

template bool wide_lazy_supported<double>(int64_t, int64_t);
template bool wide_lazy_supported<float>(int64_t, int64_t);
template void geqp3_wide_lazy<double>(rc_context *, Mat<double>, int64_t, int64_t *, Mat<double>, Mat<double>);
template void geqp3_wide_lazy<float>(rc_context *, Mat<float>, int64_t, int64_t *, Mat<float>, Mat<float>);

}  // namespace rc
