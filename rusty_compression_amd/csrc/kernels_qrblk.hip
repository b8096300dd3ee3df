// Blocked column-pivoted QR for general shapes on gfx950: ?geqp3 as LAPACK runs it -- ?laqps panels of
// NB = 32 Householder steps with the delayed "F matrix" update, then ONE GEMM (MFMA) trailing update per
// panel -- with the panel restricted to a CANDIDATE set of columns so that a panel costs about three
// passes over the trailing matrix instead of NB.
//
// Replaces /root/reference/src/pivoted_qr.rs:139-150, :161-172 (?geqp3 through the `lapack` crate) for
// matrices that are neither tall-skinny (kernels_tsqr.hip) nor short-wide (kernels_wqcoop.hip): the
// square rank-k column ID of BASELINE.json configs[4], QR::compute_from_range_estimate / two_sided_id at
// large rank (configs[3]) and plain pivoted_qr / pivoted_lq of any shape.
//
// ?laqps, step k of a panel starting at row / position j0 (rk = j0 + k), restated:
//   1. pivot  = first maximum of the partial column norms vn1 over positions rk .. n-1; swap
//   2. column = A(rk:m, pvt) - V(rk:m, 0:k) F(pvt, 0:k)^T          (bring the pivot column up to date)
//   3. ?larfg on it -> v_k, tau_k, beta
//   4. F(:, k) = tau_k A(rk:m, :)^T v_k + F(:, 0:k) auxv,  auxv = -tau_k V(rk:m, 0:k)^T v_k
//   5. row rk of the trailing columns: A(rk, j) -= V(rk, 0:k+1) F(j, 0:k+1)^T, then the LAPACK norm
//      down-date; a column whose down-dated norm lost its accuracy (temp2 <= sqrt(eps)) ends the panel
//      after this step and gets its norm recomputed once the block update has been applied
//   panel end: A(rk+1:m, rest) -= V F^T (GEMM), recompute flagged norms.
// Step 4 is a GEMV over the whole trailing matrix: NB read passes per panel, which is what bounds LAPACK's
// own ?geqp3 at half BLAS-2 speed.
//
// Candidate set.  Partial column norms never grow, so a column whose norm at the panel start is below
// tau cannot be chosen while the chosen pivots' norms stay above tau.  At a panel start the columns with
// the largest norms (a few hundred: about 8 MB of them, L2-resident) become the candidates; steps 1, 4, 5
// run on the candidates only (their arithmetic is exactly ?laqps'), and the panel additionally ends as
// soon as the best candidate no longer exceeds tau (1 + 4 sqrt(eps)) -- the margin covers the accuracy LAPACK
// itself keeps the down-dated norms to.  For the other columns F = A^T V T follows from ONE product
// Y = V^T A (MFMA GEMM, one read pass) with the panel's T factor (?larft recurrence, accumulated in step 4),
// their rows j0 .. j0+kb-1 of R and the kb sequential norm down-dates are done by one column-parallel
// kernel, and the block update is the same GEMM as LAPACK's.  Pivots are therefore ?geqp3's (same
// first-maximum rule, same down-dating formulas); what differs from LAPACK is the summation order inside
// dot products, as for every other kernel of this library.  If no column can be excluded (all norms tie,
// or the candidate budget covers the matrix) the scheme IS plain ?laqps.
//
// Host synchronisation: one 48-byte read-back per panel (the number of steps a panel completed is data
// dependent, exactly as in LAPACK).  Not capturable in a hipGraph; under capture the per-step chain of
// kernels_qr.hip runs instead.
#include "rc_common.hpp"
#include "rc_device.hpp"

#include <cstdlib>

namespace rc {

static __host__ __device__ inline int64_t cdivb(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kNB = 32;

template <typename T> struct NumB;
template <> struct NumB<double> {
    typedef unsigned long long key_t;
    static constexpr int kKeyBits = 64;
    static __host__ __device__ inline double tol3z() { return 1.0536712127723509e-08; }  // sqrt(2^-53)
    static __device__ inline key_t key(double v) { return (key_t)__double_as_longlong(fabs(v)); }
};
template <> struct NumB<float> {
    typedef unsigned int key_t;
    static constexpr int kKeyBits = 32;
    static __host__ __device__ inline float tol3z() { return 2.44140625e-04f; }  // sqrt(2^-24)
    static __device__ inline key_t key(float v) { return __float_as_uint(fabsf(v)); }
};

// device-side panel state (read back once per panel)
struct QrbState {
    int stopped;       // the panel ended before its last step
    int kb;            // steps completed when it stopped
    int lsticc;        // a candidate's norm lost its accuracy in the previous step (?laqps ends the panel)
    int stop_tau;      // stopped because the best candidate no longer exceeds every excluded column
    int ncand;         // candidates of this panel
    int have_noncand;  // 1: some unpivoted column is not a candidate
    int pad0, pad1;
    int piv[kNB];      // physical column of the panel's pivots
};

// threads of the serial per-step kernel: 8 waves (two per SIMD) may use 256 registers each, which the batched loads need
constexpr int kPT = 512;
template <typename T>
__device__ inline T block_sum_pt(T v, T *sh16) {
    v = wave_sum_dpp(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh16[threadIdx.x >> 6] = v;
    __syncthreads();
    T s = 0;
#pragma unroll
    for (int i = 0; i < kPT / 64; ++i) s += sh16[i];
    return s;
}

// ---------------------------------------------------------------------------
// init: identity permutation, exact column norms (indexed by PHYSICAL column: columns never move)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_qrb_init(Mat<T> w, int64_t *jpvt, int *pos, T *vn1, T *vn2, int *flag) {
    const int lane = threadIdx.x & 63;
    for (int64_t j = blockIdx.x * 4 + (threadIdx.x >> 6); j < w.cols; j += (int64_t)gridDim.x * 4) {
        const T *col = w.p + j * w.cs;
        T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int64_t i = lane;
        for (; i + 192 < w.rows; i += 256) {
            const T x0 = col[i], x1 = col[i + 64], x2 = col[i + 128], x3 = col[i + 192];
            a0 = fma(x0, x0, a0); a1 = fma(x1, x1, a1); a2 = fma(x2, x2, a2); a3 = fma(x3, x3, a3);
        }
        for (; i < w.rows; i += 64) { const T x0 = col[i]; a0 = fma(x0, x0, a0); }
        const T nrm = sqrt(wave_sum_dpp((a0 + a1) + (a2 + a3)));
        if (lane == 0) { vn1[j] = nrm; vn2[j] = nrm; jpvt[j] = j; pos[j] = (int)j; flag[j] = 0; }
    }
}

// ---------------------------------------------------------------------------
// panel start: candidates = the unpivoted columns whose norm is >= the cwant-th largest (all ties at the
// threshold included, so the global first-maximum is always a candidate), tau = largest excluded norm.
// One workgroup; MSB-first radix select over the norms' bit patterns (monotone for non-negative floats).
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void k_qrb_select(int n, int j0, int cwant, const int *pos, const T *vn1, int *cand, unsigned char *is_cand,
                                                     QrbState *st, T *tsc) {
    typedef typename NumB<T>::key_t key_t;
    __shared__ int hist[256];
    __shared__ int sh_scan[1024];
    __shared__ key_t sh_prefix;
    __shared__ int sh_need;
    __shared__ T sh_t[16];
    const int tid = threadIdx.x;
    const int nu = n - j0;
    key_t thr = 0;  // candidates: key >= thr
    if (nu > cwant) {
        if (tid == 0) { sh_prefix = 0; sh_need = cwant; }
        key_t mask = 0;
        for (int shift = NumB<T>::kKeyBits - 8; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const key_t prefix = sh_prefix;
            for (int c = tid; c < n; c += 1024) {
                if (pos[c] < j0) continue;
                const key_t kx = NumB<T>::key(vn1[c]);
                if ((kx & mask) == prefix) atomicAdd(&hist[(int)((kx >> shift) & 255)], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int need = sh_need, cum = 0, b = 255;
                for (; b > 0; --b) {
                    if (cum + hist[b] >= need) break;
                    cum += hist[b];
                }
                sh_need = need - cum;
                sh_prefix = prefix | ((key_t)b << shift);
            }
            mask |= (key_t)255 << shift;
            __syncthreads();
        }
        thr = sh_prefix;
    }
    // ordered compaction (deterministic candidate order) + tau
    const int chunk = (n + 1023) / 1024;
    const int c0 = tid * chunk, c1 = min(n, c0 + chunk);
    int cnt = 0;
    T tmax = 0;
    for (int c = c0; c < c1; ++c) {
        const bool unp = pos[c] >= j0;
        const bool in = unp && NumB<T>::key(vn1[c]) >= thr;
        is_cand[c] = in ? 1 : 0;
        cnt += in ? 1 : 0;
        if (unp && !in) tmax = max(tmax, fabs(vn1[c]));
    }
    sh_scan[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
        const int v = tid >= off ? sh_scan[tid - off] : 0;
        __syncthreads();
        sh_scan[tid] += v;
        __syncthreads();
    }
    int o = sh_scan[tid] - cnt;
    for (int c = c0; c < c1; ++c)
        if (is_cand[c]) cand[o++] = c;
    tmax = wave_max_dpp(tmax);
    if ((tid & 63) == 0) sh_t[tid >> 6] = tmax;
    __syncthreads();
    if (tid == 0) {
        T t = 0;
        for (int i = 0; i < 16; ++i) t = max(t, sh_t[i]);
        const int nc = sh_scan[1023];
        tsc[0] = t;
        st->stopped = 0; st->kb = 0; st->lsticc = 0; st->stop_tau = 0;
        st->ncand = nc;
        st->have_noncand = nc < nu ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------
// step k, serial part (one workgroup): pivot among the candidates, stop tests, "swap", pivot column brought
// up to date, ?larfg, auxv = -tau V^T v and column k of the panel's T factor.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kPT) void k_qrb_pivot(Mat<T> w, int j0, int k, int64_t *jpvt, int *pos, const T *vn1, const T *Fm, const int *cand,
                                                    QrbState *st, const T *tsc, T *tau, T *auxv, T *Tm) {
    __shared__ T sh16[16];
    __shared__ int shp[16], shc[16];
    __shared__ T shF[kNB];
    __shared__ T shaux[kNB];
    __shared__ int shpiv[kNB];
    __shared__ const T *shcol[kNB];
    __shared__ T shred[16 * kNB];
    __shared__ int sh_go;
    __shared__ T sh_alpha;
    if (st->stopped) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int rk = j0 + k;
    const int64_t m = w.rows;
    if (st->lsticc) {  // ?laqps: the panel ends after the step in which a norm lost its accuracy
        if (tid == 0) { st->stopped = 1; st->kb = k; }
        return;
    }
    // ---- pivot: largest norm, lowest position (idamax) over the unpivoted candidates ----------------------
    const int ncand = st->ncand;
    T best = (T)-1;
    int bp = 0x7fffffff, bc = -1;
    for (int i = tid; i < ncand; i += kPT) {
        const int c = cand[i];
        const int p = pos[c];
        if (p < rk) continue;
        const T v = fabs(vn1[c]);
        if (v > best || (v == best && p < bp)) { best = v; bp = p; bc = c; }
    }
    {
        const T mx = wave_max_dpp(best);
        int pc = (best == mx && bc >= 0) ? bp : 0x7fffffff;
        pc = wave_min_dpp(pc);
        if (bc >= 0 && best == mx && bp == pc) { sh16[wv] = mx; shp[wv] = bp; shc[wv] = bc; }
        else if (lane == 0 && pc == 0x7fffffff) { sh16[wv] = (T)-1; shp[wv] = 0x7fffffff; shc[wv] = -1; }
    }
    __syncthreads();
    if (tid == 0) {
        T bb = sh16[0];
        int pp = shp[0], cc = shc[0];
        for (int i = 1; i < kPT / 64; ++i)
            if (shc[i] >= 0 && (cc < 0 || sh16[i] > bb || (sh16[i] == bb && shp[i] < pp))) { bb = sh16[i]; pp = shp[i]; cc = shc[i]; }
        int go = 1;
        if (cc < 0) { st->stopped = 1; st->kb = k; go = 0; }  // every candidate has been used
        else if (k > 0 && st->have_noncand) {
            // a column outside the candidate set may be the true maximum once the best candidate is no longer
            // above every excluded norm (margin: the accuracy LAPACK keeps its down-dated norms to)
            const T lim = tsc[0] * ((T)1 + (T)4 * NumB<T>::tol3z());
            if (!(bb > lim)) { st->stopped = 1; st->kb = k; st->stop_tau = 1; go = 0; }
        }
        if (go) {
            const int cold = (int)jpvt[rk];
            if (pp != rk) {
                jpvt[rk] = cc; jpvt[pp] = cold;
                pos[cc] = rk; pos[cold] = pp;
            }
            st->piv[k] = cc;
            shpiv[k] = cc;
        }
        sh_go = go;
    }
    if (tid < k) shpiv[tid] = st->piv[tid];
    __syncthreads();
    if (!sh_go) return;
    const int c = shpiv[k];
    // F row of the pivot column and the panel's reflector columns, padded to a multiple of the unroll width with F = 0 and
    // a valid column, so that the inner loops carry no predicates and keep RB x 8 independent loads in flight
    if (tid < kNB) {
        shF[tid] = tid < k ? Fm[(int64_t)c * kNB + tid] : (T)0;
        shcol[tid] = w.p + (int64_t)shpiv[tid < k ? tid : 0] * w.cs;
    }
    __syncthreads();
    // ---- pivot column up to date: x = A(rk:m, c) - V(rk:m, 0:k) F(c, 0:k)^T ------------------------------
    T *wc = w.p + (int64_t)c * w.cs;
    constexpr int RB = 4, TB = 8;  // rows per thread and reflectors per batch of loads
    const int kpad = (k + TB - 1) / TB * TB;
    T ss = 0;
    for (int64_t ib = rk + tid; ib < m; ib += kPT * RB) {
        T x[RB];
        int64_t ii[RB];
#pragma unroll
        for (int e = 0; e < RB; ++e) {
            const int64_t i = ib + kPT * e;
            ii[e] = i < m ? i : (int64_t)rk;  // out-of-range rows read row rk and are not written back
            x[e] = wc[ii[e]];
        }
        for (int t0 = 0; t0 < kpad; t0 += TB) {
            T vv[RB][TB];
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                const T *cp = shcol[t0 + u];
#pragma unroll
                for (int e = 0; e < RB; ++e) vv[e][u] = cp[ii[e]];
            }
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                const T fu = shF[t0 + u];
#pragma unroll
                for (int e = 0; e < RB; ++e) x[e] = fma(-vv[e][u], fu, x[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < RB; ++e) {
            const int64_t i = ib + kPT * e;
            if (i < m) {
                wc[i] = x[e];
                if (i > rk) ss = fma(x[e], x[e], ss);
                else sh_alpha = x[e];
            }
        }
    }
    ss = block_sum_pt(ss, sh16);  // (its barriers also publish sh_alpha)
    // ---- ?larfg ----------------------------------------------------------------------------------------
    const T alpha = sh_alpha;
    const T xnorm = sqrt(ss);
    T tk = 0, beta = alpha, scal = 0;
    if (xnorm != (T)0) {
        beta = -copysign(hypot(alpha, xnorm), alpha);
        tk = (beta - alpha) / beta;
        scal = (T)1 / (alpha - beta);
    }
    // ---- v = x * scal (stored), d_t = V_t(rk:m)^T v ---------------------------------------------------------
    constexpr int RB2 = sizeof(T) == 8 ? 2 : 4, TB2 = 8;
    T d[kNB];
#pragma unroll
    for (int t = 0; t < kNB; ++t) d[t] = 0;
    if (tk != (T)0) {
        for (int64_t ib = rk + tid; ib < m; ib += kPT * RB2) {
            T v[RB2];
            int64_t ii[RB2];
#pragma unroll
            for (int e = 0; e < RB2; ++e) {
                const int64_t i = ib + kPT * e;
                ii[e] = i < m ? i : (int64_t)rk;
                const T xv = wc[ii[e]];
                v[e] = i >= m ? (T)0 : (i == rk ? (T)1 : xv * scal);
                if (i < m && i > rk) wc[i] = v[e];
            }
#pragma unroll
            for (int t0 = 0; t0 < kNB; t0 += TB2) {
                if (t0 < k) {
                    T vv[RB2][TB2];
#pragma unroll
                    for (int u = 0; u < TB2; ++u) {
                        const T *cp = shcol[t0 + u];
#pragma unroll
                        for (int e = 0; e < RB2; ++e) vv[e][u] = cp[ii[e]];
                    }
#pragma unroll
                    for (int u = 0; u < TB2; ++u)
#pragma unroll
                        for (int e = 0; e < RB2; ++e) d[t0 + u] = fma(vv[e][u], v[e], d[t0 + u]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < kNB; ++t)
        if (t < k) {
            const T s = wave_sum_dpp(d[t]);
            if (lane == 0) shred[wv * kNB + t] = s;
        }
    __syncthreads();
    if (tid < k) {
        T s = 0;
#pragma unroll
        for (int i = 0; i < kPT / 64; ++i) s += shred[i * kNB + tid];
        const T a = -tk * s;
        shaux[tid] = a;
        auxv[tid] = a;
    }
    if (tid == 0) {
        wc[rk] = beta;
        tau[rk] = tk;
    }
    __syncthreads();
    // ---- column k of T (?larft): T(0:k, k) = T(0:k, 0:k) auxv, T(k, k) = tau_k ------------------------------
    if (tid <= k) {
        T s = tk;
        if (tid < k) {
            s = 0;
            for (int q = tid; q < k; ++q) s = fma(Tm[tid + q * kNB], shaux[q], s);
        }
        Tm[tid + k * kNB] = s;
    }
}

// ---------------------------------------------------------------------------
// step k, parallel part: one wave per unpivoted candidate: F(c, k), row rk of the column, norm down-date
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_qrb_update(Mat<T> w, int j0, int k, const int *pos, T *vn1, const T *vn2, T *Fm, const int *cand, int *flag,
                                                    QrbState *st, const T *tau, const T *auxv) {
    __shared__ T shaux[kNB], shvrow[kNB];
    if (st->stopped) return;
    const int rk = j0 + k, ncand = st->ncand;
    const int cpiv = st->piv[k];
    const int lane = threadIdx.x & 63;
    const int64_t m = w.rows;
    if (threadIdx.x < k) {
        shaux[threadIdx.x] = auxv[threadIdx.x];
        shvrow[threadIdx.x] = w.p[(int64_t)st->piv[threadIdx.x] * w.cs + rk];
    }
    __syncthreads();
    const T tk = tau[rk];
    const T *v = w.p + (int64_t)cpiv * w.cs;
    const int nw = gridDim.x * 4;
    for (int ci = blockIdx.x * 4 + (threadIdx.x >> 6); ci < ncand; ci += nw) {
        const int c = cand[ci];
        if (pos[c] <= rk) continue;  // pivoted (wave-uniform)
        T *x = w.p + (int64_t)c * w.cs;
        // g = A(rk:m, c)^T v, v(rk) = 1
        T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int64_t i = rk + lane;
        if (lane == 0) { a0 = x[rk]; i += 64; }  // unit diagonal of the reflector
        for (; i + 192 < m; i += 256) {
            const T x0 = x[i], x1 = x[i + 64], x2 = x[i + 128], x3 = x[i + 192];
            const T v0 = v[i], v1 = v[i + 64], v2 = v[i + 128], v3 = v[i + 192];
            a0 = fma(x0, v0, a0); a1 = fma(x1, v1, a1); a2 = fma(x2, v2, a2); a3 = fma(x3, v3, a3);
        }
        for (; i < m; i += 64) a1 = fma(x[i], v[i], a1);
        const T g = wave_sum_dpp((a0 + a1) + (a2 + a3));
        const T ft = lane < k ? Fm[(int64_t)c * kNB + lane] : (T)0;
        const T e = wave_sum_dpp(lane < k ? ft * shaux[lane] : (T)0);
        const T fk = tk * g + e;
        const T term = lane < k ? shvrow[lane] * ft : (T)0;
        const T sr = wave_sum_dpp(term) + fk;
        if (lane == 0) {
            Fm[(int64_t)c * kNB + k] = fk;
            const T a = x[rk] - sr;
            x[rk] = a;
            const T vn = vn1[c];
            if (vn != (T)0) {
                T temp = fabs(a) / vn;
                temp = ((T)1 + temp) * ((T)1 - temp);
                temp = temp > (T)0 ? temp : (T)0;
                const T r = vn / vn2[c];
                const T temp2 = temp * r * r;
                if (temp2 <= NumB<T>::tol3z()) { flag[c] = 1; st->lsticc = 1; }
                else vn1[c] = vn * sqrt(temp);
            }
        }
    }
}

// Vp(i, t) = v_t(j0 + i): zeros above the diagonal, one on it, the reflector below  (rows x kb, column-major)
template <typename T>
__global__ __launch_bounds__(256) void k_qrb_build_vp(Mat<T> w, int j0, const QrbState *st, Mat<T> vp) {
    const int t = blockIdx.y;
    const T *col = w.p + (int64_t)st->piv[t] * w.cs + j0;
    T *out = vp.p + (int64_t)t * vp.cs;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < vp.rows; i += (int64_t)gridDim.x * 256) out[i] = (i < t) ? (T)0 : (i == t) ? (T)1 : col[i];
}

// ---------------------------------------------------------------------------
// panel end, one thread per physical column:
//   pivoted columns        : F row = 0 (the block update must not touch them)
//   candidates             : nothing (F row, rows of R and norms were kept current by the steps)
//   every other column     : F(c, :) = Y(:, c)^T T, rows j0 .. j0+kb-1 (= rows of R), kb sequential norm down-dates
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_qrb_finish(Mat<T> w, int j0, int kb, const int *pos, const unsigned char *is_cand, T *vn1, const T *vn2, T *Fm,
                                                    const T *Y, int64_t ldy, const T *Tm, Mat<T> vp, int *flag) {
    __shared__ T Tl[kNB * kNB], Vl[kNB * kNB];
    for (int e = threadIdx.x; e < kNB * kNB; e += 256) {
        const int r = e % kNB, q = e / kNB;  // element (r, q)
        Tl[e] = (r <= q && q < kb) ? Tm[r + q * kNB] : (T)0;
        Vl[e] = (q < r && r < kb && r < vp.rows) ? vp.p[(int64_t)q * vp.cs + r] : (T)0;  // strictly lower part of the panel's unit block
    }
    __syncthreads();
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= w.cols) return;
    T *frow = Fm + c * kNB;
    if (pos[c] < j0 + kb) {
#pragma unroll
        for (int t = 0; t < kNB; ++t) frow[t] = 0;
        return;
    }
    if (is_cand[c]) return;
    T f[kNB];
    {
        T y[kNB];
#pragma unroll
        for (int s = 0; s < kNB; ++s) y[s] = s < kb ? Y[(int64_t)s * ldy + c] : (T)0;
#pragma unroll
        for (int t = 0; t < kNB; ++t) {
            T acc = 0;
#pragma unroll
            for (int s = 0; s < kNB; ++s)
                if (s <= t) acc = fma(y[s], Tl[s + t * kNB], acc);
            f[t] = acc;
        }
    }
#pragma unroll
    for (int t = 0; t < kNB; ++t) frow[t] = f[t];
    T *x = w.p + c * w.cs + j0;
    T vn = vn1[c];
    const T vnb = vn2[c];
    bool lost = false;
    const int nrows = (int)min((int64_t)kb, w.rows - j0);
#pragma unroll
    for (int kk = 0; kk < kNB; ++kk) {
        if (kk < nrows) {
            T acc = f[kk];
#pragma unroll
            for (int t = 0; t < kNB; ++t)
                if (t < kk) acc = fma(Vl[kk + t * kNB], f[t], acc);
            const T a = x[kk] - acc;
            x[kk] = a;
            if (!lost && vn != (T)0) {
                T temp = fabs(a) / vn;
                temp = ((T)1 + temp) * ((T)1 - temp);
                temp = temp > (T)0 ? temp : (T)0;
                const T r = vn / vnb;
                if (temp * r * r <= NumB<T>::tol3z()) lost = true;  // recomputed exactly after the block update
                else vn *= sqrt(temp);
            }
        }
    }
    vn1[c] = vn;
    if (lost) flag[c] = 1;
}

// exact norms of the flagged columns below row `row0` (?laqps: VN1 = VN2 = ?nrm2 after the block update)
template <typename T>
__global__ __launch_bounds__(256) void k_qrb_renorm(Mat<T> w, int row0, const int *pos, int *flag, T *vn1, T *vn2) {
    const int lane = threadIdx.x & 63;
    for (int64_t c = blockIdx.x * 4 + (threadIdx.x >> 6); c < w.cols; c += (int64_t)gridDim.x * 4) {
        if (!flag[c]) continue;
        T acc = 0;
        if (pos[c] >= row0) {
            const T *col = w.p + c * w.cs;
            for (int64_t i = row0 + lane; i < w.rows; i += 64) { const T v = col[i]; acc = fma(v, v, acc); }
            acc = sqrt(wave_sum_dpp(acc));
            if (lane == 0) { vn1[c] = acc; vn2[c] = acc; }
        }
        if (lane == 0) flag[c] = 0;
    }
}

static int env_int_b(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

template <typename T>
bool geqp3_blocked_supported(int64_t m, int64_t n, int64_t kmax) {
    static const int on = env_int_b("RC_QRCP_BLOCKED", 1);
    return on && m >= 128 && n >= 128 && kmax >= 8 && n < (int64_t)1 << 30 && m < (int64_t)1 << 30;
}

// w: m x n column-major working matrix (overwritten with the ?geqp3 output format: R on and above the
// diagonal in position order, reflectors below, columns never moved); jpvt: n; tau: kmax
template <typename T>
void geqp3_blocked(rc_context *c, Mat<T> w, int64_t kmax, int64_t *jpvt, T *tau) {
    RC_REQUIRE(w.rs == 1, RC_LAYOUT_ERROR, "geqp3_blocked: working matrix must be column-major");
    RC_REQUIRE(!c->capturing, RC_RUNTIME_ERROR, "geqp3_blocked: reads one scalar back per panel, not capturable");
    const int64_t m = w.rows, n = w.cols;
    kmax = std::min(kmax, std::min(m, n));
    if (kmax <= 0) return;
    ProfScope ps(c, "op:geqp3_blocked %lldx%lld k=%lld", (long long)m, (long long)n, (long long)kmax);
    ArenaMark mark(c);
    int *pos = c->alloc<int>((size_t)n);
    int *cand = c->alloc<int>((size_t)n);
    int *flag = c->alloc<int>((size_t)n);
    unsigned char *is_cand = c->alloc<unsigned char>((size_t)n);
    T *vn1 = c->alloc<T>((size_t)n), *vn2 = c->alloc<T>((size_t)n);
    T *Fm = c->alloc<T>((size_t)n * kNB);
    T *Tm = c->alloc<T>((size_t)kNB * kNB);
    T *auxv = c->alloc<T>(kNB);
    T *tsc = c->alloc<T>(4);
    QrbState *st = reinterpret_cast<QrbState *>(c->alloc_bytes(sizeof(QrbState)));
    Mat<T> vp = colmajor(c->alloc<T>((size_t)even_ld(m) * kNB), m, kNB, even_ld(m));
    T *Y = c->alloc<T>((size_t)kNB * even_ld(n));
    const int64_t ldy = even_ld(n);

    hipLaunchKernelGGL(k_qrb_init<T>, dim3((unsigned)std::min<int64_t>(cdivb(n, 4), 8192)), dim3(256), 0, c->stream, w, jpvt, pos, vn1, vn2, flag);
    // candidate budget: about RC_QRCP_CAND_MB of column data (L2-resident across the steps of a panel), at least 4 NB columns
    static const int cand_mb = env_int_b("RC_QRCP_CAND_MB", 8);
    int64_t cwant = std::max<int64_t>(4 * kNB, ((int64_t)cand_mb << 20) / (int64_t)(sizeof(T) * (size_t)std::max<int64_t>(m, 1)));
    if (cand_mb <= 0) cwant = n;  // plain ?laqps
    int64_t j0 = 0;
    while (j0 < kmax) {
        const int nbp = (int)std::min<int64_t>(kNB, kmax - j0);
        const int64_t cw = std::min<int64_t>(cwant, n - j0);
        hipLaunchKernelGGL(k_qrb_select<T>, dim3(1), dim3(1024), 0, c->stream, (int)n, (int)j0, (int)cw, pos, vn1, cand, is_cand, st, tsc);
        const unsigned grid2 = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdivb(n - j0, 4), cdivb(2 * cw, 4)));
        for (int k = 0; k < nbp; ++k) {
            hipLaunchKernelGGL(k_qrb_pivot<T>, dim3(1), dim3(kPT), 0, c->stream, w, (int)j0, k, jpvt, pos, vn1, Fm, cand, st, tsc, tau, auxv, Tm);
            hipLaunchKernelGGL(k_qrb_update<T>, dim3(grid2), dim3(256), 0, c->stream, w, (int)j0, k, pos, vn1, vn2, Fm, cand, flag, st, tau, auxv);
        }
        QrbState h;
        RC_HIP(hipMemcpyAsync(&h, st, sizeof(QrbState), hipMemcpyDeviceToHost, c->stream));
        RC_HIP(hipStreamSynchronize(c->stream));
        const int kb = h.stopped ? h.kb : nbp;
        RC_REQUIRE(kb >= 1 && kb <= nbp, RC_PIVOTED_QR_ERROR, "geqp3_blocked: panel at %lld made %d steps", (long long)j0, kb);
        const int64_t rows = m - j0;
        const bool last = j0 + kb >= kmax;
        Mat<T> vpp = Mat<T>(vp.p, rows, kb, 1, vp.cs);
        hipLaunchKernelGGL(k_qrb_build_vp<T>, dim3((unsigned)std::min<int64_t>(cdivb(rows, 256), 64), (unsigned)kb), dim3(256), 0, c->stream, w, (int)j0, st, vpp);
        if (h.have_noncand) {
            // Y = V^T A(j0:m, :) for every column: one read pass (MFMA GEMM); used for the non-candidates only
            Mat<T> ym = rowmajor(Y, kb, n, ldy);
            gemm<T>(c, 1, vpp.t(), w.sub(j0, rows, 0, n), 0, ym);
        }
        hipLaunchKernelGGL(k_qrb_finish<T>, dim3((unsigned)cdivb(n, 256)), dim3(256), 0, c->stream, w, (int)j0, kb, pos, is_cand, vn1, vn2, Fm, Y, ldy, Tm, vpp, flag);
        if (!last && rows - kb > 0) {
            // block update of everything below the panel, written as the transposed product so that the lanes of the
            // MFMA accumulator run along the column-major matrix' contiguous dimension:
            //   A(j0+kb:m, :)^T -= F(:, 0:kb) V(kb:, 0:kb)^T
            Mat<T> ft = Mat<T>(Fm, n, kb, kNB, 1);
            gemm<T>(c, (T)-1, ft, vpp.sub(kb, rows - kb, 0, kb).t(), (T)1, w.sub(j0 + kb, rows - kb, 0, n).t());
            hipLaunchKernelGGL(k_qrb_renorm<T>, dim3((unsigned)std::min<int64_t>(cdivb(n, 4), 4096)), dim3(256), 0, c->stream, w, (int)(j0 + kb), pos, flag, vn1, vn2);
        }
        // a panel that the tau test ended early means the candidate set was too small for this spectrum
        if (h.stopped && h.stop_tau && kb < nbp / 2) cwant = std::min<int64_t>(n, cwant * 2);
        j0 += kb;
    }
}

template bool geqp3_blocked_supported<double>(int64_t, int64_t, int64_t);
template bool geqp3_blocked_supported<float>(int64_t, int64_t, int64_t);
template void geqp3_blocked<double>(rc_context *, Mat<double>, int64_t, int64_t *, double *);
template void geqp3_blocked<float>(rc_context *, Mat<float>, int64_t, int64_t *, float *);

}  // namespace rc
