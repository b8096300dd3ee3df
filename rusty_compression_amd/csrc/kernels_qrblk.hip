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
#include <vector>

namespace rc {

static __host__ __device__ inline int64_t cdivb(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kNB = 32;

constexpr int kQrbVtaMaxSplits = 16;  // row chunks of the streaming Y = V^T A (k_qrb_vta): partial slabs summed by k_qrb_finish
template <typename T> struct NumB;
template <> struct NumB<double> {
    typedef unsigned long long key_t;
    static constexpr int kKeyBits = 64;
    static constexpr int kSelBits = 32;
    static __host__ __device__ inline double tol3z() { return 1.0536712127723509e-08; }  // sqrt(2^-53)
    static __device__ inline key_t key(double v) { return (key_t)__double_as_longlong(fabs(v)); }
};
template <> struct NumB<float> {
    typedef unsigned int key_t;
    static constexpr int kKeyBits = 32;
    static constexpr int kSelBits = 24;
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
                                                     QrbState *st, T *tsc, int *cpos0, T *cvn0) {
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
        // the leading kSelBits bits of the norm decide (f32: sign + exponent + 15 mantissa bits, f64: + 20): the candidate
        // set is "every norm >= the bin of the cwant-th largest", i.e. at least cwant columns, ties and near-ties included
        for (int shift = NumB<T>::kKeyBits - 8; shift >= NumB<T>::kKeyBits - NumB<T>::kSelBits; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const key_t prefix = sh_prefix;
            for (int c = tid; c < n; c += 1024) {
                if (pos[c] < j0) continue;
                const key_t kx = NumB<T>::key(vn1[c]);
                if ((kx & mask) == prefix) atomicAdd(&hist[(int)((kx >> shift) & 255)], 1);
            }
            __syncthreads();
            // suffix counts S[b] = sum_{b' >= b} hist[b'] by the first 256 threads; the bin with S[b] >= need > S[b + 1] is chosen
            if (tid < 256) {
                const int lane = tid & 63, wv = tid >> 6;
                int v = hist[tid];
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int o = __shfl_down(v, off, 64);
                    if (lane + off < 64) v += o;
                }
                if (lane == 0) sh_scan[wv] = v;  // wave totals
                sh_scan[8 + tid] = v;            // in-wave suffix sums
            }
            __syncthreads();
            if (tid < 256) {
                const int wv = tid >> 6;
                int above = 0;
                for (int w2 = wv + 1; w2 < 4; ++w2) above += sh_scan[w2];
                const int sfx = sh_scan[8 + tid] + above;          // S[tid]
                const int nextv = sfx - hist[tid];                   // S[tid + 1]
                const int need = sh_need;
                if (sfx >= need && nextv < need) { sh_scan[300] = tid; sh_scan[301] = need - nextv; }
            }
            __syncthreads();
            if (tid == 0) {
                sh_need = sh_scan[301];
                sh_prefix = prefix | ((key_t)sh_scan[300] << shift);
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
        if (is_cand[c]) { cand[o] = c; cpos0[o] = pos[c]; cvn0[o] = vn1[c]; ++o; }
    tmax = wave_max_dpp(tmax);
    if ((tid & 63) == 0) sh_t[tid >> 6] = tmax;
    __syncthreads();
    if (tid == 0) {
        T t = 0;
        for (int i = 0; i < 16; ++i) t = max(t, sh_t[i]);
        const int nc = sh_scan[1023];
        tsc[0] = t;
        st->stopped = 0; st->kb = 0; st->lsticc = 0; st->stop_tau = 0; st->pad0 = 0;
        st->ncand = nc;
        st->have_noncand = nc < nu ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------
// Per-step kernels.  Both are multi-workgroup and free of serial single-CU passes.  A step is a chain of dependent
// global-memory round trips (~1 us each across XCDs), so each kernel is organised as two rounds of independent loads.
// The few scalar decisions of a step (pivot, stop tests, ?larfg, auxv) are evaluated REDUNDANTLY by every workgroup
// from the same inputs in the same order -- all workgroups agree bit for bit -- and only workgroup 0 writes the shared
// bookkeeping; what one launch both reads and replaces is double buffered (candidate positions) or kept out of place
// (the updated pivot column in xbuf, the candidates' rows of R in Rrow).
//
//   step A(k): pivot = first maximum of the candidates' norms, stop tests, "swap"; the pivot column is brought up to
//              date by row slabs (one per workgroup): x = A(rk:m, c) - V(rk:m, 0:k) F(c, 0:k)^T, with the slab's share
//              of ||x||^2 and of V^T x
//   step C(k): one workgroup per unpivoted candidate: ?larfg and auxv from the slab sums, g = A(rk:m, j)^T v, then the
//              candidate finishes its own step: F(j, k), its entry of row rk of R, the norm down-date
// A panel of nbp steps is A(0) C(0) ... A(nbp-1) C(nbp-1).
// ---------------------------------------------------------------------------
template <typename T>
struct QrbPanel {
    int64_t *jpvt;
    int *pos;             // n: position of physical column (workgroup 0 of step A is the only writer)
    const int *cand;      // candidate -> physical column
    int *cpos[2];         // candidate positions, double buffered by step parity
    T *cvn;               // candidate partial norms (vn1); each entry has one writer (its workgroup in step C)
    const T *vn2;         // n: last exactly computed norm per physical column
    T *Fm;                // n x kNB, row per physical column
    T *Rrow;              // kNB x ncap: row j0 + kk of R for the candidates (scattered into the matrix at panel end)
    int64_t ncap;
    T *xbuf;              // m (+ pad): the pivot column brought up to date, rows rk .. m-1 at offset (rk mod 16 bytes)
    T *pss;               // [g] slab sums of squares, [64] = alpha, [128 + g * kNB + t] slab sums of V_t^T x
    int *flag;            // n: norm lost its accuracy, recompute after the block update
    QrbState *st;
    const T *tsc;         // [0] = tau threshold
    T *tau;
    T *Tm;                // kNB x kNB
};

template <typename T>
__device__ inline bool qrb_downdate(T a, T &vn, T vnb) {  // ?laqps norm down-date; true = accuracy lost (vn unchanged)
    if (vn == (T)0) return false;
    T temp = fabs(a) / vn;
    temp = ((T)1 + temp) * ((T)1 - temp);
    temp = temp > (T)0 ? temp : (T)0;
    const T r = vn / vnb;
    if (temp * r * r <= NumB<T>::tol3z()) return true;
    vn *= sqrt(temp);
    return false;
}

template <typename T>
__global__ __launch_bounds__(256) void k_qrb_step_a(Mat<T> w, int j0, int k, QrbPanel<T> P) {
    __shared__ T shF[kNB];
    __shared__ int shpiv[kNB];
    __shared__ T shv[4];
    __shared__ int shp[4], shi[4], shc[4];
    __shared__ T shpart[4 * (kNB + 1)];
    QrbState *st = P.st;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool wg0 = blockIdx.x == 0;
    const int rk = j0 + k;
    const int cur = k & 1, nxt = cur ^ 1;
    const int64_t m = w.rows;
    const int *cposc = P.cpos[cur];
    // ---- round 1: panel state, every candidate's norm and position ------------------------------------------------------
    const int stopped = st->stopped;
    const int ncand = st->ncand;
    const int have_noncand = st->have_noncand;
    const int lsticc = st->lsticc;
    const T lim = P.tsc[0] * ((T)1 + (T)4 * NumB<T>::tol3z());
    if (tid < kNB) shpiv[tid] = (tid < k) ? st->piv[tid] : 0;
    constexpr int NPF = 4;
    int pf_p[NPF], pf_c[NPF];
    T pf_vn[NPF];
#pragma unroll
    for (int e = 0; e < NPF; ++e) {
        const int ci = tid + 256 * e;
        const bool ok = ci < (int)P.ncap;
        pf_p[e] = ok ? cposc[ci] : -1;
        pf_vn[e] = ok ? P.cvn[ci] : (T)0;
        pf_c[e] = ok ? P.cand[ci] : 0;
    }
    if (stopped) return;
    T best = (T)-1;
    int bp = 0x7fffffff, bi = -1, bcol = 0;
#pragma unroll
    for (int e = 0; e < NPF; ++e) {
        const int ci = tid + 256 * e;
        if (ci < ncand && pf_p[e] >= rk) {
            const T v = fabs(pf_vn[e]);
            if (v > best || (v == best && pf_p[e] < bp)) { best = v; bp = pf_p[e]; bi = ci; bcol = pf_c[e]; }
        }
    }
    for (int ci = tid + 256 * NPF; ci < ncand; ci += 256) {
        const int p = cposc[ci];
        if (p >= rk) {
            const T v = fabs(P.cvn[ci]);
            if (v > best || (v == best && p < bp)) { best = v; bp = p; bi = ci; bcol = P.cand[ci]; }
        }
    }
    {
        const T mx = wave_max_dpp(best);
        int pc = (best == mx && bi >= 0) ? bp : 0x7fffffff;
        pc = wave_min_dpp(pc);
        if (bi >= 0 && best == mx && bp == pc) { shv[wv] = mx; shp[wv] = bp; shi[wv] = bi; shc[wv] = bcol; }
        else if (lane == 0 && pc == 0x7fffffff) { shv[wv] = (T)-1; shp[wv] = 0x7fffffff; shi[wv] = -1; shc[wv] = 0; }
    }
    __syncthreads();
    T bb = shv[0];
    int pp = shp[0], ci_s = shi[0], cs = shc[0];
    for (int i = 1; i < 4; ++i)
        if (shi[i] >= 0 && (ci_s < 0 || shv[i] > bb || (shv[i] == bb && shp[i] < pp))) { bb = shv[i]; pp = shp[i]; ci_s = shi[i]; cs = shc[i]; }
    // ---- stop tests (identical in every workgroup) -----------------------------------------------------------------------
    int stop = 0, why_tau = 0;
    if (lsticc) stop = 1;                           // ?laqps: the panel ends after the step in which a norm lost its accuracy
    else if (ci_s < 0) stop = 1;                    // every candidate has been used
    else if (k > 0 && have_noncand && !(bb > lim)) { stop = 1; why_tau = 1; }
    if (stop) {
        if (wg0 && tid == 0) { st->stopped = 1; st->kb = k; st->stop_tau = why_tau; }
        return;
    }
    // ---- workgroup 0: swap bookkeeping ------------------------------------------------------------------------------------
    if (wg0) {
#pragma unroll
        for (int e = 0; e < NPF; ++e) {
            const int ci = tid + 256 * e;
            if (ci < ncand) P.cpos[nxt][ci] = ci == ci_s ? rk : (pf_p[e] == rk ? pp : pf_p[e]);
        }
        for (int ci = tid + 256 * NPF; ci < ncand; ci += 256) {
            const int p = cposc[ci];
            P.cpos[nxt][ci] = ci == ci_s ? rk : (p == rk ? pp : p);
        }
        if (tid == 0) {
            const int cold = (int)P.jpvt[rk];
            if (pp != rk) {
                P.jpvt[rk] = cs; P.jpvt[pp] = cold;
                P.pos[cs] = rk; P.pos[cold] = pp;
            }
            st->piv[k] = cs;
        }
    }
    // ---- round 2: pivot column slab, the panel's reflector rows, F row of the pivot column -----------------------------------
    const int64_t nrows = m - rk;
    const int64_t slab = (nrows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * slab, r1 = min(nrows, r0 + slab);
    const T *wc = w.p + (int64_t)cs * w.cs + rk;
    const int xoff = rk & (16 / (int)sizeof(T) - 1);  // xbuf(r) sits at the same 16-byte phase as row rk + r of a column
    T ss = 0;
    T pv[kNB];  // this thread's share of V_t(rk+1:m)^T x(1:)
#pragma unroll
    for (int t = 0; t < kNB; ++t) pv[t] = 0;
    // the first row of this thread: loads issued together with the F row, before the barrier that publishes it
    const int64_t rf = r0 + tid;
    const bool hasf = rf < r1;
    T xf = hasf ? wc[rf] : (T)0;
    T vf[kNB];
#pragma unroll
    for (int t = 0; t < kNB; ++t) vf[t] = (hasf && t < k) ? w.p[(int64_t)shpiv[t] * w.cs + rk + rf] : (T)0;
    if (tid < kNB) shF[tid] = tid < k ? P.Fm[(int64_t)cs * kNB + tid] : (T)0;
    __syncthreads();
    auto process = [&](int64_t r, T x, const T *vv) {
#pragma unroll
        for (int t = 0; t < kNB; ++t) x = fma(-vv[t], shF[t], x);  // shF is zero beyond the step (and vv too)
        P.xbuf[xoff + r] = x;
        if (r > 0) {
            ss = fma(x, x, ss);
#pragma unroll
            for (int t = 0; t < kNB; ++t) pv[t] = fma(vv[t], x, pv[t]);
        } else {
            P.pss[64] = x;  // alpha
        }
    };
    if (hasf) process(rf, xf, vf);
    for (int64_t r = rf + 256; r < r1; r += 256) {  // slabs longer than the workgroup
        T x = wc[r];
        T vv[kNB];
#pragma unroll
        for (int t = 0; t < kNB; ++t) vv[t] = t < k ? w.p[(int64_t)shpiv[t] * w.cs + rk + r] : (T)0;
        process(r, x, vv);
    }
    // slab sums: ||x(1:)||^2 and V_t^T x, fixed order (lanes -> waves -> workgroup)
    ss = wave_sum_dpp(ss);
    if (lane == 0) shpart[wv * (kNB + 1) + kNB] = ss;
#pragma unroll
    for (int t = 0; t < kNB; ++t) {
        if (t < k) {
            const T s2 = wave_sum_dpp(pv[t]);
            if (lane == 0) shpart[wv * (kNB + 1) + t] = s2;
        }
    }
    __syncthreads();
    if (tid == 0) P.pss[blockIdx.x] = (shpart[kNB] + shpart[(kNB + 1) + kNB]) + (shpart[2 * (kNB + 1) + kNB] + shpart[3 * (kNB + 1) + kNB]);
    if (tid < k) P.pss[128 + blockIdx.x * kNB + tid] = (shpart[tid] + shpart[(kNB + 1) + tid]) + (shpart[2 * (kNB + 1) + tid] + shpart[3 * (kNB + 1) + tid]);
}

// Step C: one workgroup per candidate.
template <typename T>
__global__ __launch_bounds__(256) void k_qrb_step_c(Mat<T> w, int j0, int k, int nslab, int vec_ok, QrbPanel<T> P) {
    constexpr int VL = 16 / sizeof(T);
    typedef T vecT __attribute__((ext_vector_type(VL)));
    __shared__ T shs[4];
    __shared__ T shw[4];
    __shared__ T shaux[kNB], shvrow[kNB];
    __shared__ T shpg[8 * kNB];
    QrbState *st = P.st;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int rk = j0 + k;
    const int nxt = (k & 1) ^ 1;
    const int64_t m = w.rows;
    const int xoff = rk & (VL - 1);
    // ---- round 1 ----------------------------------------------------------------------------------------------------
    const int stopped = st->stopped;
    const int ncand = st->ncand;
    const int cpiv = st->piv[k];
    const T part = (tid < nslab && tid < 64) ? P.pss[tid] : (T)0;
    const T alpha = P.pss[64];
    // slab sums of V_t^T x: thread (t = tid & 31, group = tid >> 5) fetches 8 of the (at most 64) slabs, all loads independent
    T pgq = 0;
    {
        const int t = tid & (kNB - 1), gq = tid >> 5;
        T part8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int g = gq * 8 + u;
            part8[u] = (t < k && g < nslab) ? P.pss[128 + g * kNB + t] : (T)0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) pgq += part8[u];
    }
    int mypiv = 0;
    if (tid < k) mypiv = st->piv[tid];
    int pf_c = 0, pf_p = 0;
    T pf_vn = 0;
    if ((int)blockIdx.x < (int)P.ncap) { pf_c = P.cand[blockIdx.x]; pf_p = P.cpos[nxt][blockIdx.x]; pf_vn = P.cvn[blockIdx.x]; }
    if (stopped) return;
    const T *xb = P.xbuf + xoff;
    T *vcol = w.p + (int64_t)cpiv * w.cs + rk;
    const int64_t nrows = m - rk;
    const int peel = vec_ok ? (int)min((int64_t)((VL - xoff) & (VL - 1)), nrows) : 0;
    const int64_t nv = vec_ok ? (nrows - peel) / VL : 0;
    const int64_t tail0 = peel + nv * VL;
    // row rk of the panel's reflectors (needed for gv and for the row update)
    const T myvrow = tid < k ? w.p[(int64_t)mypiv * w.cs + rk] : (T)0;
    T tk = 0, beta = 0, scal = 0;
    bool have = false;
    auto larfg = [&]() {  // ?larfg and auxv from the slab sums (every workgroup, identical); uniform call sites only
        shpg[tid] = pgq;  // [group][t]
        if (tid < 64) {
            const T s = wave_sum_dpp(part);
            if (tid == 0) {
                const T xnorm = sqrt(s);
                T tk_ = 0, beta_ = alpha, scal_ = 0;
                if (xnorm != (T)0) {
                    beta_ = -copysign(hypot(alpha, xnorm), alpha);
                    tk_ = (beta_ - alpha) / beta_;
                    scal_ = (T)1 / (alpha - beta_);
                }
                shs[0] = tk_; shs[1] = beta_; shs[2] = scal_;
            }
        }
        __syncthreads();
        tk = shs[0]; beta = shs[1]; scal = shs[2];
        if (tid < kNB) {
            T pg = 0;  // V_t(rk+1:m)^T x(1:), the eight slab groups in fixed order
#pragma unroll
            for (int q = 0; q < 8; ++q) pg += shpg[q * kNB + tid];
            // V_t(rk:m)^T v with v(rk) = 1, v(rk+1:) = x(1:) * scal
            const T gvt = tid < k ? myvrow + scal * pg : (T)0;
            shaux[tid] = tid < k ? -tk * gvt : (T)0;
            shvrow[tid] = myvrow;
        }
        __syncthreads();
        have = true;
    };
    for (int tg = blockIdx.x; tg < ncand; tg += gridDim.x) {
        const bool mine = tg == (int)blockIdx.x;
        const int p = mine ? pf_p : P.cpos[nxt][tg];
        if (p <= rk) continue;  // pivoted (uniform over the workgroup, before any barrier)
        const int c = mine ? pf_c : P.cand[tg];
        T vn = mine ? pf_vn : P.cvn[tg];
        const T *col = w.p + (int64_t)c * w.cs + rk;
        // round 2: the first four vectors per thread of the column and of x (4096 rows of f32 / 2048 of f64 per workgroup),
        // and what thread 0 needs to finish the candidate's step
        const vecT *cv = reinterpret_cast<const vecT *>(col + peel);
        const vecT *xv = reinterpret_cast<const vecT *>(xb + peel);
        vecT cq[4], xq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t v = tid + 256 * q;
            const bool ok = v < nv;
            cq[q] = ok ? cv[v] : vecT{};
            xq[q] = ok ? xv[v] : vecT{};
        }
        const T c_diag = tid == 0 ? col[0] : (T)0;  // unit diagonal of the reflector; also the entry of row rk to update
        T c_head = 0, x_head = 0;
        if (tid >= 1 && tid < peel) { c_head = col[tid]; x_head = xb[tid]; }
        const T ft = (tid < k) ? P.Fm[(int64_t)c * kNB + tid] : (T)0;   // F(c, 0:k)
        const T vnb = tid == 0 ? P.vn2[c] : (T)0;
        if (!have) larfg();
        T a0 = c_diag, a1 = 0, a2 = 0, a3 = 0;
        if (tk != (T)0) {
            a1 = c_head * (x_head * scal);
            for (int64_t r = max(tail0, (int64_t)1) + tid; r < nrows; r += 256) a1 = fma(col[r], xb[r] * scal, a1);
#pragma unroll
            for (int u = 0; u < VL; ++u) {
                const bool diag0 = peel == 0 && u == 0 && tid == 0;  // vector 0, element 0 is row 0 when nothing was peeled
                a0 = fma(cq[0][u], diag0 ? (T)0 : xq[0][u] * scal, a0);
                a1 = fma(cq[1][u], xq[1][u] * scal, a1);
                a2 = fma(cq[2][u], xq[2][u] * scal, a2);
                a3 = fma(cq[3][u], xq[3][u] * scal, a3);
            }
            for (int64_t v = tid + 1024; v < nv; v += 256) {
                const vecT c0 = cv[v];
                const vecT x0 = xv[v];
#pragma unroll
                for (int u = 0; u < VL; ++u) a2 = fma(c0[u], x0[u] * scal, a2);
            }
        }
        T g = wave_sum_dpp((a0 + a1) + (a2 + a3));
        // e = F(c, 0:k) . auxv, sr = V(rk, 0:k) . F(c, 0:k): lanes t < k of wave 0 (entries beyond k are zero)
        T e = 0, sr = 0;
        if (wv == 0) {  // (ft is zero in lanes >= k)
            e = wave_sum_dpp(ft * shaux[lane & (kNB - 1)]);
            sr = wave_sum_dpp(shvrow[lane & (kNB - 1)] * ft);
        }
        __syncthreads();  // shw of the previous target has been read
        if (lane == 0) shw[wv] = g;
        __syncthreads();
        if (tid == 0) {
            g = (shw[0] + shw[1]) + (shw[2] + shw[3]);
            // the candidate finishes its own step: F(c, k), its entry of row rk of R, the norm down-date
            const T fk = tk * g + e;
            const T a = c_diag - (sr + fk);
            P.Fm[(int64_t)c * kNB + k] = fk;
            P.Rrow[(int64_t)k * P.ncap + tg] = a;
            if (qrb_downdate(a, vn, vnb)) { P.flag[c] = 1; st->lsticc = 1; }
            else P.cvn[tg] = vn;
        }
    }
    if (!have) larfg();
    if (blockIdx.x == 0) {
        if (tid == 0) { P.tau[rk] = tk; vcol[0] = beta; }
        // column k of the panel's T factor (?larft): T(0:k, k) = T(0:k, 0:k) auxv, T(k, k) = tau_k
        if (tid <= k && tid < kNB) {
            T sacc = tk;
            if (tid < k) {
                T trow[kNB];
#pragma unroll
                for (int q = 0; q < kNB; ++q) trow[q] = (q >= tid && q < k) ? P.Tm[tid + q * kNB] : (T)0;
                sacc = 0;
#pragma unroll
                for (int q = 0; q < kNB; ++q) sacc = fma(trow[q], shaux[q], sacc);
            }
            P.Tm[tid + k * kNB] = sacc;
        }
    }
    // ---- v stored: rows distributed over the workgroups (nobody reads the column itself in this launch) -----------------------
    {
        const int64_t nr = m - rk - 1;
        const int64_t per = (nr + gridDim.x - 1) / gridDim.x;
        const int64_t a0 = 1 + (int64_t)blockIdx.x * per, a1 = min(m - rk, a0 + per);
        for (int64_t r = a0 + tid; r < a1; r += 256) vcol[r] = tk != (T)0 ? xb[r] * scal : xb[r];
    }
}

// panel end: rows j0 .. j0+kb-1 of R and the final norms of the candidates go back into the matrix / the norm array
template <typename T>
__global__ __launch_bounds__(256) void k_qrb_scatter(Mat<T> w, int j0, int kb, QrbPanel<T> P, T *vn1) {
    const int ci = blockIdx.x * 256 + threadIdx.x;
    if (ci >= P.st->ncand) return;
    const int c = P.cand[ci];
    const int p = P.pos[c];
    T *x = w.p + (int64_t)c * w.cs + j0;
    const int lim = min(kb, p - j0);  // rows of R above the column's own diagonal position
    for (int kk = 0; kk < lim; ++kk) x[kk] = P.Rrow[(int64_t)kk * P.ncap + ci];
    if (p >= j0 + kb) vn1[c] = P.cvn[ci];
}

// Vp(i, t) = v_t(j0 + i): zeros above the diagonal, one on it, the reflector below  (rows x kb, column-major)
template <typename T>
__global__ __launch_bounds__(256) void k_qrb_build_vp(Mat<T> w, int j0, const QrbState *st, Mat<T> vp, const int *ok) {
    if (ok && *ok == 0) return;  // optimistic issue: an earlier panel broke an assumption, nothing behind it may run on its state
    const int t = blockIdx.y;
    const T *col = w.p + (int64_t)st->piv[t] * w.cs + j0;
    T *out = vp.p + (int64_t)t * vp.cs;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < vp.rows; i += (int64_t)gridDim.x * 256) out[i] = (i < t) ? (T)0 : (i == t) ? (T)1 : col[i];
}

// the same from the finished permutation: reflector t of the panel sits in physical column jpvt[j0 + t]
template <typename T>
__global__ __launch_bounds__(256) void k_qrb_build_vp_pos(Mat<T> w, int j0, const int64_t *jpvt, Mat<T> vp) {
    const int t = blockIdx.y;
    const T *col = w.p + jpvt[j0 + t] * w.cs + j0;
    T *out = vp.p + (int64_t)t * vp.cs;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < vp.rows; i += (int64_t)gridDim.x * 256) out[i] = (i < t) ? (T)0 : (i == t) ? (T)1 : col[i];
}

// ---------------------------------------------------------------------------
// panel end, one thread per physical column:
//   pivoted columns        : F row = 0 (the block update must not touch them)
//   candidates             : nothing (F row, rows of R and norms were kept current by the steps); F row = 0 after a cooperative panel
//   every other column     : F(c, :) = Y(:, c)^T T, rows j0 .. j0+kb-1 (= rows of R), kb sequential norm down-dates
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_qrb_finish(Mat<T> w, int j0, int kb, const int *pos, const unsigned char *is_cand, T *vn1, const T *vn2, T *Fm,
                                                    const T *Y, int64_t ldy, const T *Tm, Mat<T> vp, int *flag, int coop, int nsplit, int64_t sstride, const int *ok) {
    if (ok && *ok == 0) return;  // optimistic issue: an earlier panel broke an assumption, nothing behind it may run on its state
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
    // (the candidates of a cooperative panel are already up to date: no F row either)
    if (pos[c] < j0 + kb || (coop && is_cand[c])) {
#pragma unroll
        for (int t = 0; t < kNB; ++t) frow[t] = 0;
        return;
    }
    if (is_cand[c]) return;
    T f[kNB];
    {
        T y[kNB];
#pragma unroll
        for (int s = 0; s < kNB; ++s) {
            // Y arrives as nsplit partial slabs over the rows (k_qrb_vta): summed here in slab order (deterministic), no reduction launch
            T acc = 0;
            if (s < kb)
                for (int sp = 0; sp < nsplit; ++sp) acc += Y[(int64_t)sp * sstride + (int64_t)s * ldy + c];
            y[s] = acc;
        }
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

// ---------------------------------------------------------------------------------------------------------------------
// Y = V^T A for the panel end (f32; round 3): 32 x n from ONE read pass over the trailing matrix, HBM bound (32 flops per element of
// A).  The generic MFMA GEMM reached 1.9 TB/s here (8 KB per workgroup and K tile in flight through LDS, then a split-K reduction
// launch).  This kernel has no LDS and no barrier: a wave owns 64 columns x one chunk of rows and streams them with one 16-byte load
// per lane and column block -- A is column-major, so a lane reads 4 consecutive rows of its column, the four 16-lane groups of a wave
// 16 consecutive rows = one 64-byte sector per column -- and V the same way from L2 (32 x rows, re-read by every wave: 32 MB in all).
// v_mfma_f32_16x16x4f32 with the operand layout validated in kernels_gemm.hip (A: row = lane % 16, k = lane / 16; B: k = lane / 16,
// col = lane % 16; D[r]: row = 4 (lane / 16) + r, col = lane % 16): component t of the lanes' float4s is the MFMA's k-slot, so four
// MFMAs consume the 16 rows of a step.  The row chunks' partial results go to `nsplit` slabs that k_qrb_finish sums in order.
// Requires 16-byte aligned columns (j0 % 4 == 0, leading dimensions % 4 == 0): the host checks and otherwise takes the GEMM.
// ---------------------------------------------------------------------------------------------------------------------
typedef float vta_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_qrb_vta(const float *v, int64_t ldv, const float *a, int64_t lda, int rows, int n, int kb, float *ypart, int64_t ldy,
                                                 int64_t sstride, int rows_per_split, const int *ok) {
    if (ok && *ok == 0) return;  // optimistic issue: an earlier panel broke an assumption, nothing behind it may run on its state
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & 15, g = lane >> 4;
    const int c0 = (blockIdx.x * 4 + wave) * 64;
    if (c0 >= n) return;  // whole wave (no barriers in this kernel)
    const int split = blockIdx.y;
    const int r0 = split * rows_per_split, r1 = min(rows, r0 + rows_per_split);
    vta_f4 acc[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[b][j] = vta_f4{0.f, 0.f, 0.f, 0.f};
    const float *ap[4], *vp[2];
    bool aok[4], vok[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + 16 * j + cl;
        aok[j] = c < n;
        ap[j] = a + (int64_t)(aok[j] ? c : 0) * lda + 4 * g;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int k = 16 * b + cl;
        vok[b] = k < kb;
        vp[b] = v + (int64_t)(vok[b] ? k : 0) * ldv + 4 * g;
    }
    auto load = [&](int i0, vta_f4 (&fa)[4], vta_f4 (&fv)[2]) {
        if (i0 + 16 <= r1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) fa[j] = aok[j] ? *reinterpret_cast<const vta_f4 *>(ap[j] + i0) : vta_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 2; ++b) fv[b] = vok[b] ? *reinterpret_cast<const vta_f4 *>(vp[b] + i0) : vta_f4{0.f, 0.f, 0.f, 0.f};
        } else {  // the last, partial step of the matrix: element-wise with bounds
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) fa[j][t] = (aok[j] && i0 + 4 * g + t < r1) ? ap[j][i0 + t] : 0.f;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int t = 0; t < 4; ++t) fv[b][t] = (vok[b] && i0 + 4 * g + t < r1) ? vp[b][i0 + t] : 0.f;
        }
    };
    auto compute = [&](const vta_f4 (&fa)[4], const vta_f4 (&fv)[2]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[b][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fv[b][t], fa[j][t], acc[b][j], 0, 0, 0);
    };
    // three fragment sets in rotation: two steps of loads (12 x 16 bytes per lane) are in flight while one is consumed
    vta_f4 fa0[4], fv0[2], fa1[4], fv1[2], fa2[4], fv2[2];
    int i = r0;
    if (i < r1) load(i, fa0, fv0);
    if (i + 16 < r1) load(i + 16, fa1, fv1);
    for (; i < r1; i += 48) {
        if (i + 32 < r1) load(i + 32, fa2, fv2);
        compute(fa0, fv0);
        if (i + 16 >= r1) break;
        if (i + 48 < r1) load(i + 48, fa0, fv0);
        compute(fa1, fv1);
        if (i + 32 >= r1) break;
        if (i + 64 < r1) load(i + 64, fa1, fv1);
        compute(fa2, fv2);
    }
    float *yp = ypart + (int64_t)split * sstride;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + 16 * j + cl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * b + 4 * g + r;
                if (k < kb && c < n) yp[(int64_t)k * ldy + c] = acc[b][j][r];
            }
        }
}
// y[k][c] = sum over the slabs, in slab order (deterministic); slab 0 receives the sum.  (Folding this sum into k_qrb_finish was
// measured: its 16 workgroups then issue 512 loads per thread and the panel end got 0.2 ms SLOWER.)
__global__ __launch_bounds__(256) void k_qrb_ysum(float *ypart, int64_t ldy, int64_t sstride, int kb, int n, int nsplit, const int *ok) {
    if (ok && *ok == 0) return;  // optimistic issue: an earlier panel broke an assumption, nothing behind it may run on its state
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)kb * n) return;
    const int k = (int)(e / n), c = (int)(e % n);
    float *p = ypart + (int64_t)k * ldy + c;
    float acc = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) acc += p[(int64_t)sp * sstride];
    p[0] = acc;
}
template <typename T>
static bool qrb_vta_launch(rc_context *, Mat<T>, Mat<T>, int, T *, int64_t, int64_t, int *, const int *) { return false; }
// returns true and sets *nsplit when the streaming kernel ran (ypart: nsplit slabs of kNB x ldy); false: take the GEMM
template <>
bool qrb_vta_launch<float>(rc_context *c, Mat<float> vpp, Mat<float> wsub, int kb, float *ypart, int64_t ldy, int64_t sstride, int *nsplit, const int *ok) {
    static const int on = [] { const char *e = getenv("RC_QRCP_VTA"); return e ? atoi(e) : 1; }();
    const int64_t rows = wsub.rows, n = wsub.cols;
    if (!on || wsub.rs != 1 || vpp.rs != 1 || rows < 256 || n < 64 || kb > kNB) return false;
    if ((wsub.cs % 4) || (vpp.cs % 4) || (reinterpret_cast<uintptr_t>(wsub.p) % 16) || (reinterpret_cast<uintptr_t>(vpp.p) % 16)) return false;
    // row chunks so that ~1024 waves exist (4 per CU): a chunk is a whole number of 16-row steps
    const int64_t col_wgs = cdivb(n, 256);
    int splits = (int)std::max<int64_t>(1, std::min<int64_t>(kQrbVtaMaxSplits, 256 / std::max<int64_t>(col_wgs, 1)));
    int64_t rps = cdivb(cdivb(rows, splits), 16) * 16;
    splits = (int)cdivb(rows, rps);
    ProfScope ps(c, "kernel:k_qrb_vta rows=%lld n=%lld kb=%d splits=%d", (long long)rows, (long long)n, kb, splits);
    hipLaunchKernelGGL(k_qrb_vta, dim3((unsigned)col_wgs, (unsigned)splits), dim3(256), 0, c->stream, vpp.p, vpp.cs, wsub.p, wsub.cs, (int)rows, (int)n, kb, ypart, ldy, sstride,
                       (int)rps, ok);
    if (splits > 1) hipLaunchKernelGGL(k_qrb_ysum, dim3((unsigned)cdivb((int64_t)kb * n, 256)), dim3(256), 0, c->stream, ypart, ldy, sstride, kb, (int)n, splits, ok);
    *nsplit = 1;  // slab 0 holds Y
    return true;
}

// exact norms of the flagged columns below row `row0` (?laqps: VN1 = VN2 = ?nrm2 after the block update)
// all != 0: every unpivoted column (see qrb_finish)
template <typename T>
__global__ __launch_bounds__(256) void k_qrb_renorm(Mat<T> w, int row0, int all, const int *pos, int *flag, T *vn1, T *vn2, const QrbState *st, const int *ok) {
    if (ok && *ok == 0) return;  // optimistic issue: an earlier panel broke an assumption, nothing behind it may run on its state
    const int lane = threadIdx.x & 63;
    if (st) all = st->lsticc;  // optimistic issue: the host has not seen the panel's state (same value, read where it lives)
    for (int64_t c = blockIdx.x * 4 + (threadIdx.x >> 6); c < w.cols; c += (int64_t)gridDim.x * 4) {
        if (!flag[c] && !all) continue;
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

// ---------------------------------------------------------------------------------------------------------------------
// Cooperative panel: the steps of a panel in ONE launch, the candidates resident in registers.
//
// G workgroups of 8 waves; wave (wg, wv) owns candidate ci = 8 wg + wv and keeps rows j0 .. m-1 of its column in registers
// (lane l holds rows VW l + 64 VW e + q, VW = 16 bytes of elements) for the whole panel, so the candidate slab is read once
// and written once per panel and a step costs ONE grid barrier instead of two kernel boundaries.  The columns are kept UP
// TO DATE (every reflector is applied to every unpivoted candidate right away, as ?laqp2 does), so there is no F matrix
// for the candidates and a norm that loses its accuracy is recomputed on the spot from the registers instead of ending the
// panel.  Per step (k_wq_coop's protocol, kernels_wqcoop.hip):
//   A  every workgroup posts its best unpivoted candidate -- norm, position, candidate index, the column itself -- to its
//      slot (write-through stores), then a header whose words carry the step number
//   -- grid barrier = wave 0 polls the G headers until all carry this step's number --
//   B  every workgroup agrees on the pivot (largest norm, first position on ties) and on the stop test (best candidate no
//      longer above tau), fetches the winning column into LDS and generates the reflector redundantly (?larfg)
//   C  every wave applies it to its column (v from LDS), down-dates its norm (?laqp2 formulas), and the waves that own
//      EARLIER pivots of the panel record v_t^T v_k, from which the panel's T factor (?larft) is built afterwards
// Nothing global is modified before the last step has completed: the permutation changes of the non-candidates are
// kept in an overlay by workgroup 0 and flushed at the end, so a launch that had to give up (co-residency time-out, more
// candidates than waves) leaves the panel untouched and the host runs it through the step kernels instead.
//
// PROTOCOL INVARIANTS (who writes what, at which scope; reviewed against the code in round 3 -- keep list and code in step)
//  Q1  Cross-workgroup traffic INSIDE the launch is 8-byte agent-scope relaxed atomics only: a.hdr (headers), a.cols (posted columns),
//      a.sync[0..2] (finished counter, abort word, commit counter).  Everything else a workgroup reads inside the launch was written
//      before the launch (w, P.cand, P.cpos, P.cvn, vn2, st, tsc, jpvt: by kernels in front on the same stream) or by itself.
//  Q2  Header words validate themselves: (payload << 32) | (step + 1); a reader accepts a slot only when all five words carry the
//      step it waits for.  Headers and column slots are double buffered by step parity.  Slot parity p is rewritten at step jj + 2;
//      its owner gets there only through the poll of step jj + 1, which needs every workgroup's header of jj + 1, which a workgroup
//      writes only after it has finished BOTH reads of step jj (the poll and the fetch of the winning column).  Hence no slot is
//      overwritten while anybody may still read it.
//  Q3  A posted column is complete in memory before its header exists: the ONE wave that posts drains its own write-through stores
//      (s_waitcnt vmcnt(0)), then the workgroup barrier, then wave 0 writes the header (the barrier alone waits for LDS traffic only).
//  Q4  Cleared per launch by k_coop_gate on the same stream in front: all header words, sync[0], sync[2]; sync[1] is set by the gate.
//  Q5  Every workgroup derives pivot, stop test and reflector from the same words with the same instructions: identical decisions
//      and bit-identical reflectors everywhere (the v_t^T v_k recorded by the owners of earlier pivots rely on that).
//  Q6  No global state of the factorization (w, vn1, vn2, pos, jpvt, st) is written before the COMMIT: after its last step a
//      workgroup adds 1 to sync[2] and waits (bounded) until it reads G; a workgroup that aborted never adds, so either every
//      workgroup writes back or none does.  Exceptions, harmless when the panel is abandoned: a.D, st->piv, P.tau of the steps taken
//      (overwritten by the step kernels that redo the panel).  Workgroup 0 reads the occupants of the panel's positions (ov_orig)
//      BEFORE the first step, because other workgroups' write-back of jpvt may start while it is still in its last step's bookkeeping.
//  Q7  All spins are bounded (kSpin); on expiry the abort word is set and every workgroup leaves at its next poll; st->pad0 tells
//      the host (1 completed, 2 more candidates than waves, 3 time-out) and the panel is redone by the step kernels.
//  Q8  The device budget taken by the gate (a.need) is released exactly once, by the last arrival at sync[0].
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
struct QrbCoopArgs {
    Mat<T> w;
    int j0, nbp;
    QrbPanel<T> P;              // cand, cpos[0] (initial positions), cvn (initial norms), vn2, st, tsc, tau, jpvt, pos
    T *vn1, *vn2;               // by physical column
    unsigned long long *hdr;    // [2][G][5]
    T *cols;                    // [2][G][mpad]
    int mpad;
    T *D;                       // kNB x kNB: D[t + k * kNB] = v_t^T v_k  (t < k)
    unsigned *sync;             // [0] finished-workgroup counter, [1] abort
    unsigned *sem;
    unsigned need;              // units of the device budget to release
};

__device__ inline void qc_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double qc_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// NE vectors of VW = 16 / sizeof(T) rows per lane: 64 VW NE rows at most
#ifdef RC_QRC_TIMING
__device__ unsigned long long g_qrc_dbg[8];
#endif
// WPE: waves per SIMD the register allocation must leave room for (4 = two workgroups per CU = half the budget units)
template <typename T, int NE, int WPE>
__global__ __launch_bounds__(512, WPE) void k_qrb_coop(QrbCoopArgs<T> a) {
    constexpr int VW = 16 / (int)sizeof(T);
    constexpr int NW = 8;
    constexpr int ROWS = 64 * VW * NE;
    constexpr int kNoInt = 0x7fffffff;
    constexpr int kSpin = 1 << 22;
    typedef unsigned long long u64;
    __shared__ __attribute__((aligned(16))) T vv[ROWS];   // the reflector
    __shared__ T sh_part[NW], sh_alpha;
    __shared__ double sh_v[NW];
    __shared__ int sh_p[NW], sh_j[NW];
    __shared__ int bc[6];
    __shared__ int sh_exit;
    __shared__ int ov_pos[kNB], ov_col[kNB], ov_n;
    __shared__ int ov_orig[kNB];  // who sat at positions j0 .. j0+nbp-1 when the panel started (workgroup 0)
    // (wave-uniform values are forced into scalar registers: the per-wave roles below must compile to scalar branches, not
    // to exec-masked selects that keep two copies of the register-resident column alive)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x, wg = blockIdx.x;
    const int j0 = a.j0;
    const int rows = (int)a.w.rows - j0;  // <= ROWS (host)
    QrbState *st = a.P.st;
    unsigned *done = a.sync, *abortw = a.sync + 1;
    const unsigned ab0 = __hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ab0 == 2u) {  // the gate gave up: no budget is held
        if (wg == 0 && tid == 0) st->pad0 = 3;
        return;
    }
    const int ncand = st->ncand;
    const bool too_many = ncand > NW * G;
    bool aborted = ab0 != 0u || too_many;
    if (tid == 0) { sh_exit = 0; ov_n = 0; }
    // Snapshot of the panel's positions, taken BEFORE any step: the write-back at the end of this kernel stores into jpvt, and a
    // workgroup that is through its last step may do so while workgroup 0 is still in that step's bookkeeping (it used to read
    // jpvt[j] there and could see the pivot just written by its owner: one duplicated column in the permutation).
    if (wg == 0 && tid < kNB) ov_orig[tid] = tid < a.nbp ? (int)a.P.jpvt[j0 + tid] : -1;

    const int ci = wg * NW + wv;
    const bool have = __builtin_amdgcn_readfirstlane((int)(ci < ncand && !aborted)) != 0;
    const int c = __builtin_amdgcn_readfirstlane(have ? a.P.cand[ci] : 0);
    T x[NE][VW];
    {
        // (branch free: the row index is clamped into the matrix and the value masked; per-element predicates are written as
        // "lane-constant < scalar" so that nothing per element stays alive for the write-back at the end)
        const T *col = a.w.p + (int64_t)c * a.w.cs + j0;
        const int lq = VW * lane;
#pragma unroll
        for (int e = 0; e < NE; ++e)
#pragma unroll
            for (int q = 0; q < VW; ++q) {
                const int r = lq + 64 * VW * e + q;
                const T val = col[min(r, rows - 1)];
                x[e][q] = (have && lq + q < rows - 64 * VW * e) ? val : (T)0;
            }
    }
    int mypos = __builtin_amdgcn_readfirstlane(have ? a.P.cpos[0][ci] : kNoInt);
    T vn1 = have ? a.P.cvn[ci] : (T)0, vn2 = have ? a.P.vn2[c] : (T)0;
    bool renormed = false;
    const T lim = a.P.tsc[0] * ((T)1 + (T)4 * NumB<T>::tol3z());
    const int have_noncand = st->have_noncand;
    __syncthreads();

    // per-phase s_memtime totals of workgroup 0 (diagnostic build only: -DRC_QRC_TIMING, tools/qrc_timing.py)
#ifdef RC_QRC_TIMING
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#define RC_QTICK(k)                                                                       \
    {                                                                                     \
        unsigned long long now_;                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");   \
        tph[k] += now_ - tlast;                                                           \
        tlast = now_;                                                                     \
    }
#else
#define RC_QTICK(k)
#endif
    int kb = 0, stop_tau = 0;
    for (int jj = 0; jj < a.nbp && !aborted; ++jj) {
        const int j = j0 + jj, par = jj & 1;
        // (opaque copy of `rows` per step: keeps the compiler from hoisting one row-index register and one mask per
        // element of the column out of this loop; lane-constant parts are written as VW * lane < scalar)
        int rows_v = rows;
        asm volatile("" : "+s"(rows_v));
        const int l0 = VW * lane;
        // ---- A: the workgroup's best unpivoted candidate ----------------------------------------------------------
        if (lane == 0) {
            const bool speak = have && mypos >= j;
            double b = speak ? fabs((double)vn1) : -1.0;
            if (speak && !(b >= 0.0)) b = -0.5;  // NaN norms are taken last
            sh_v[wv] = b;
            sh_p[wv] = speak ? mypos : kNoInt;
            sh_j[wv] = (have && mypos == j) ? ci : -1;
        }
        __syncthreads();
        RC_QTICK(0)
        double lb = -1.0;
        int lp = kNoInt, lc = -1, lj = -1;
#pragma unroll
        for (int k2 = 0; k2 < NW; ++k2) {
            const double v2 = sh_v[k2];
            const int p2 = sh_p[k2];
            if (p2 != kNoInt && (lc < 0 || v2 > lb || (v2 == lb && p2 < lp))) { lb = v2; lp = p2; lc = wg * NW + k2; }
            lj = sh_j[k2] > lj ? sh_j[k2] : lj;
        }
        if (__builtin_amdgcn_readfirstlane((int)(lc == ci && have))) {  // this wave's column is the workgroup's candidate: post it
            T *slot = a.cols + ((size_t)par * G + wg) * a.mpad;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int r = VW * lane + 64 * VW * e;
                if (l0 < rows_v - 64 * VW * e) {  // rows beyond the matrix inside the last vector are zeros
                    if constexpr (sizeof(T) == 8) {
                        qc_st(reinterpret_cast<double *>(slot + r), (double)x[e][0]);
                        qc_st(reinterpret_cast<double *>(slot + r + 1), (double)x[e][1]);
                    } else {
                        qc_st(reinterpret_cast<double *>(slot + r), __hiloint2double(__float_as_int((float)x[e][1]), __float_as_int((float)x[e][0])));
                        qc_st(reinterpret_cast<double *>(slot + r + 2), __hiloint2double(__float_as_int((float)x[e][3]), __float_as_int((float)x[e][2])));
                    }
                }
            }
            // The posting wave drains its write-through stores itself: on gfx950 a workgroup barrier waits for LDS traffic only
            // (s_waitcnt lgkmcnt(0); s_barrier -- no vmcnt(0) outside threadgroup-split mode), and the header below is written by
            // ANOTHER wave.  Without this wait the header could overtake the column under memory load: a reader then fetched a
            // column with a few rows of the posting two steps earlier, formed a slightly different reflector than the other
            // workgroups, and the v_t^T v_k it recorded made the panel's T factor wrong (seen once in ~1000 matrices, and only
            // with streaming kernels of other matrices running beside the panel).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // the column's write-through stores have completed (drained by their wave, above) before the header is written
        __syncthreads();
        RC_QTICK(1)
        if (wv == 0) {
            const u64 tag = (unsigned)(jj + 1);
            if (lane == 0) {
                u64 *h = a.hdr + ((size_t)par * G + wg) * 5;
                const u64 nb = (u64)__double_as_longlong(lc >= 0 ? lb : -1.0);
                __hip_atomic_store(h + 0, (nb & 0xffffffff00000000ull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(h + 1, (nb << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(h + 2, ((u64)(unsigned)(lc >= 0 ? lp : kNoInt) << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(h + 3, ((u64)(unsigned)lc << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(h + 4, ((u64)(unsigned)lj << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // ---- grid barrier + pivot agreement: poll the G headers (one per lane) until all carry this step's tag
            double hb = -2.0;
            int hp = kNoInt, hc = -1, hj = -1;
            bool ab = false;
            for (int it = 0;; ++it) {
                bool ok = true;
                hb = -2.0; hp = kNoInt; hc = -1; hj = -1;
                if (lane < G) {
                    const u64 *h = a.hdr + ((size_t)par * G + lane) * 5;
                    const u64 w0 = __hip_atomic_load(h + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), w1 = __hip_atomic_load(h + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                              w2 = __hip_atomic_load(h + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), w3 = __hip_atomic_load(h + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                              w4 = __hip_atomic_load(h + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (unsigned)w0 == (unsigned)tag && (unsigned)w1 == (unsigned)tag && (unsigned)w2 == (unsigned)tag && (unsigned)w3 == (unsigned)tag &&
                         (unsigned)w4 == (unsigned)tag;
                    hb = __longlong_as_double((long long)((w0 & 0xffffffff00000000ull) | (w1 >> 32)));
                    hp = (int)(w2 >> 32);
                    hc = (int)(w3 >> 32);
                    hj = (int)(w4 >> 32);
                }
                if (__all(ok)) break;
                if (it > kSpin || ((it & 31) == 31 && __any(__hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u))) { ab = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            const double mx = wave_max_dpp(hc >= 0 ? hb : -2.0);
            const int wp = wave_min_dpp((hc >= 0 && hb == mx) ? hp : kNoInt);
            const u64 mask = __ballot(hc >= 0 && hb == mx && hp == wp);
            const int src = mask ? __ffsll((long long)mask) - 1 : 0;
            const int wc = mask ? __builtin_amdgcn_readlane(hc, src) : -1;
            const int cj = wave_max_dpp(hj);
            if (lane == 0) {
                // stop: no unpivoted candidate left, or the best one no longer exceeds every excluded column
                // (the first step of a panel always proceeds: the first maximum over all columns is a candidate by construction)
                const int why_tau = (wc >= 0 && jj > 0 && have_noncand && !((T)mx > lim)) ? 1 : 0;
                const int stop = (wc < 0 || why_tau) ? 1 : 0;
                if (ab) __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sh_exit = ab ? 1 : 0;
                bc[0] = wc; bc[1] = wp; bc[2] = cj; bc[3] = stop; bc[4] = why_tau;
            }
        }
        __syncthreads();
        RC_QTICK(2)
        if (sh_exit) { aborted = true; break; }
        const int wc = __builtin_amdgcn_readfirstlane(bc[0]), wp = __builtin_amdgcn_readfirstlane(bc[1]), cj = __builtin_amdgcn_readfirstlane(bc[2]);
        if (bc[3]) { stop_tau = bc[4]; break; }
        // ---- B: the winning column (complete before its header was posted) -> registers of the whole workgroup, ?larfg from the
        //         workgroup's partial sums (identical arithmetic in every workgroup), the reflector -> LDS ---------------------------
        constexpr int NF = (ROWS * (int)sizeof(T) / 8 + 511) / 512;  // 8-byte words of the column per thread
        double cw8[NF];
        {
            const double *src = reinterpret_cast<const double *>(a.cols + ((size_t)par * G + (wc / NW)) * a.mpad);
            const int nd = ((rows_v + VW - 1) / VW) * 2;  // whole 16-byte vectors were posted (zeros beyond the matrix)
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                const int i = tid + 512 * u;
                cw8[u] = i < nd ? qc_ld(src + i) : 0.0;
            }
        }
        T tj = 0, beta, scal = 0;
        {
            T part = 0;
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                const int i = tid + 512 * u;
                if constexpr (sizeof(T) == 8) {
                    const T xv = (T)cw8[u];
                    if (i > jj) part = fma(xv, xv, part);
                    if (i == jj) sh_alpha = xv;
                } else {
                    const T x0 = (T)__int_as_float(__double2loint(cw8[u])), x1 = (T)__int_as_float(__double2hiint(cw8[u]));
                    if (2 * i > jj) part = fma(x0, x0, part);
                    if (2 * i + 1 > jj) part = fma(x1, x1, part);
                    if (2 * i == jj) sh_alpha = x0;
                    if (2 * i + 1 == jj) sh_alpha = x1;
                }
            }
            part = wave_sum_dpp(part);
            if (lane == 0) sh_part[wv] = part;
        }
        __syncthreads();
        RC_QTICK(3)
        {
            T ss = 0;
#pragma unroll
            for (int k2 = 0; k2 < NW; ++k2) ss += sh_part[k2];
            const T xnorm = sqrt(ss);
            const T alpha = sh_alpha;
            beta = alpha;
            if (xnorm != (T)0) {
                const T aa = fabs(alpha);
                const T wmax = aa > xnorm ? aa : xnorm, zmin = aa > xnorm ? xnorm : aa;
                const T zr = zmin / wmax;
                beta = -copysign(wmax * sqrt(fma(zr, zr, (T)1)), alpha);
                tj = (beta - alpha) / beta;
                scal = (T)1 / (alpha - beta);
            }
        }
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            const int i = tid + 512 * u;
            if (i < ROWS * (int)sizeof(T) / 8) {
                if constexpr (sizeof(T) == 8) {
                    vv[i] = i < jj ? (T)0 : i == jj ? (T)1 : (T)cw8[u] * scal;
                } else {
                    const T x0 = (T)__int_as_float(__double2loint(cw8[u])), x1 = (T)__int_as_float(__double2hiint(cw8[u]));
                    vv[2 * i] = 2 * i < jj ? (T)0 : 2 * i == jj ? (T)1 : x0 * scal;
                    vv[2 * i + 1] = 2 * i + 1 < jj ? (T)0 : 2 * i + 1 == jj ? (T)1 : x1 * scal;
                }
            }
        }
        __syncthreads();
        RC_QTICK(4)
        // ---- C: apply, down-date, bookkeeping -------------------------------------------------------------------------
        if (have) {
            // f32: the reflector stays in registers between the dot product and the update (one LDS pass); f64: two passes of
            // four vectors at a time (register budget)
            constexpr bool VREG = false;  // (the reflector in registers between the two passes: one LDS pass less, but 64 registers that cost the second workgroup per CU)
            T vr[VREG ? NE : 1][VW];
            T dot = 0;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
#pragma unroll
                for (int q = 0; q < VW; ++q) {
                    const T vq = vv[l0 + 64 * VW * e + q];
                    if (VREG) vr[VREG ? e : 0][q] = vq;
                    dot = fma(vq, x[e][q], dot);
                }
                if (!VREG && (e & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // at most four LDS vectors in flight
            }
            dot = wave_sum_dpp(dot);
            const int role = __builtin_amdgcn_readfirstlane(ci == wc ? 1 : (mypos < j ? 2 : 3));
            if (role == 1) {
                // pivot column: R(j, j) = beta, the reflector below the diagonal (?geqp3 format); rows above keep R
                mypos = j;
                if (lane == 0) { st->piv[jj] = c; a.P.tau[j] = tj; }  // (the owner knows its physical column: no load on anybody's critical path)
#pragma unroll
                for (int e = 0; e < NE; ++e) {
#pragma unroll
                    for (int q = 0; q < VW; ++q) {
                        const T vq = VREG ? vr[VREG ? e : 0][q] : vv[l0 + 64 * VW * e + q];  // zero beyond the matrix, like the column
                        if (e > 0) x[e][q] = vq;
                        else if (l0 + q == jj) x[e][q] = beta;
                        else if (l0 + q > jj) x[e][q] = vq;
                    }
                    if (!VREG && (e & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            } else if (role == 2) {
                // an earlier pivot of this panel: rows >= jj of its column hold its reflector, so dot = v_t^T v_k
                if (lane == 0) a.D[(mypos - j0) + jj * kNB] = dot;
            } else {
                if (ci == cj && wp != j) mypos = wp;  // (scalar)
                if (tj != (T)0) {
                    const T f = tj * dot;
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
#pragma unroll
                        for (int q = 0; q < VW; ++q) x[e][q] = fma(-f, VREG ? vr[VREG ? e : 0][q] : vv[l0 + 64 * VW * e + q], x[e][q]);
                        if (!VREG && (e & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // row jj of this column = its entry of row j of R: lane jj / VW, component jj % VW of register vector 0
                T aj = 0;
#pragma unroll
                for (int q = 0; q < VW; ++q) aj = (jj % VW == q) ? x[0][q] : aj;
                aj = __shfl(aj, jj / VW, 64);
                if (vn1 != (T)0) {  // ?laqp2 down-date
                    const T t = fabs(aj) / vn1;
                    T temp = (T)1 - t * t;
                    temp = temp > (T)0 ? temp : (T)0;
                    const T r2 = vn1 / vn2;
                    if (temp * r2 * r2 <= NumB<T>::tol3z()) {
                        T ssl = 0;
#pragma unroll
                        for (int e = 0; e < NE; ++e)
#pragma unroll
                            for (int q = 0; q < VW; ++q)
                                if (e > 0 || l0 + q > jj) ssl = fma(x[e][q], x[e][q], ssl);  // rows beyond the matrix hold zeros
                        ssl = wave_sum_dpp(ssl);
                        vn1 = vn2 = (jj < rows_v - 1) ? sqrt(ssl) : (T)0;
                        renormed = true;
                    } else {
                        vn1 = vn1 * sqrt(temp);
                    }
                }
            }
        }
        if (wg == 0 && tid == 0) {
            if (cj < 0 && wp != j) {
                // the column at position j is not a candidate: it moves to the pivot's old position (overlay; flushed at the end)
                int oldc = -1, slot = -1;
                for (int i = 0; i < ov_n; ++i)
                    if (ov_pos[i] == j) { oldc = ov_col[i]; slot = i; }
                if (oldc < 0) { oldc = ov_orig[jj]; slot = ov_n; ov_n = ov_n + 1; }
                ov_pos[slot] = wp;
                ov_col[slot] = oldc;
            }
        }
        kb = jj + 1;
        RC_QTICK(5)
    }
#undef RC_QTICK
#ifdef RC_QRC_TIMING
    if (wg == 0 && tid == 0)
        for (int k2 = 0; k2 < 6; ++k2) g_qrc_dbg[k2] = tph[k2];
#endif

    // ---- commit: nothing has been written yet; the write-back below must happen in EVERY workgroup or in none.  A workgroup
    // that gave up inside the loop never arrives here un-aborted, so the counter reaches G only if all of them completed every
    // step; anybody who does not see that within the spin bound aborts (and then nobody can have seen G either). ------------------
    if (!aborted) {
        __syncthreads();
        if (tid == 0) {
            unsigned *commit = a.sync + 2;
            __hip_atomic_fetch_add(commit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = false;
            for (int it = 0; it <= kSpin; ++it) {
                if (__hip_atomic_load(commit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)G) { ok = true; break; }
                if ((it & 31) == 31 && __hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok) __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_exit = ok ? 0 : 1;
        }
        __syncthreads();
        if (sh_exit) aborted = true;
    }
    // ---- the panel is done: columns, norms and the permutation go back -------------------------------------------------------
    if (!aborted) {
        if (have) {
            // (the column's addresses are formed afresh from an opaque copy of its index: kept alive from the prologue they
            // would cost one 64-bit register pair per element across the whole loop)
            int c2 = c, rows_e = rows;
            asm volatile("" : "+s"(c2), "+s"(rows_e));
            T *col = a.w.p + (int64_t)c2 * a.w.cs + j0;
            const int lq = VW * lane;
#pragma unroll
            for (int e = 0; e < NE; ++e)
#pragma unroll
                for (int q = 0; q < VW; ++q)
                    if (lq + q < rows_e - 64 * VW * e) col[lq + 64 * VW * e + q] = x[e][q];
            if (lane == 0) {
                a.vn1[c] = vn1;
                if (renormed) a.vn2[c] = vn2;
                a.P.pos[c] = mypos;
                a.P.jpvt[mypos] = c;
            }
        }
        if (wg == 0 && tid == 0) {
            for (int i = 0; i < ov_n; ++i) { a.P.jpvt[ov_pos[i]] = ov_col[i]; a.P.pos[ov_col[i]] = ov_pos[i]; }
            st->stopped = kb < a.nbp ? 1 : 0;
            st->kb = kb;
            st->stop_tau = stop_tau;
            st->lsticc = 0;
            st->pad0 = 1;  // the cooperative panel completed
        }
    } else if (wg == 0 && tid == 0) {
        st->pad0 = too_many ? 2 : 3;
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)G - 1u) __hip_atomic_fetch_sub(a.sem, a.need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// T factor of a cooperative panel from the recorded v_t^T v_k (?larft, forward / columnwise): T(0:k, k) = -tau_k T(0:k, 0:k) d(0:k, k)
template <typename T>
__global__ __launch_bounds__(1024) void k_qrb_build_t(const T *D, const T *tau, int j0, int kb, T *Tm, const int *ok) {
    if (ok && *ok == 0) return;  // optimistic issue: an earlier panel broke an assumption, nothing behind it may run on its state
    // thread (i, t) = (tid / 32, tid % 32): the 32 products of row i with d(:, k), summed over the 32 lanes of a half wave
    __shared__ T Tl[kNB * kNB], Dl[kNB * kNB], taul[kNB];
    static_assert(kNB == 32, "one half wave per row of T");
    const int tid = threadIdx.x, i = tid >> 5, t = tid & 31;
    {
        Tl[tid] = 0;
        const int tt = tid % kNB, kk = tid / kNB;
        Dl[tid] = (tt < kk && kk < kb) ? D[tid] : (T)0;
    }
    if (tid < kNB) taul[tid] = tid < kb ? tau[j0 + tid] : (T)0;
    __syncthreads();
    for (int k = 0; k < kb; ++k) {
        T p = (t >= i && t < k && i < k) ? Tl[i + t * kNB] * Dl[t + k * kNB] : (T)0;
        p = group_sum_dpp<16>(p);
        p += __shfl_xor(p, 16, 64);  // the two rows of 16 lanes of the half wave
        if (t == 0 && i < k) Tl[i + k * kNB] = -taul[k] * p;  // (column k: nobody reads it in this step)
        if (tid == 0) Tl[k + k * kNB] = taul[k];
        __syncthreads();
    }
    Tm[tid] = Tl[tid];
}

// ---------------------------------------------------------------------------------------------------------------------
// Panel-end block update as a streaming kernel:  A(row0 + r, c) -= sum_t V(r, t) F(c, t)  for every column c that still has
// an F row (unpivoted, and not a candidate of a cooperative panel).  It is HBM bound (one read and one write of the trailing
// matrix, 2 kb flops per element), and the 16x16x4 MFMA GEMM reached 1.8 TB/s on it (its accumulator layout reads and writes C
// in 64-byte pieces).  Here a thread owns RPT rows (consecutive threads = consecutive rows: every access is a full line), keeps
// their kNB entries of V in registers for the whole launch, and walks over a strip of columns; F(c, :) is wave-uniform, so it
// arrives through the scalar cache and feeds the FMAs as a scalar operand.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int RPT>
__global__ __launch_bounds__(256) void k_qrb_block_update(Mat<T> w, int row0, int kb, int j0, Mat<T> vp, int vrow0, const T *Fm, const int *pos,
                                                          const unsigned char *is_cand, int coop, int cols_per_wg, const int *proceed) {
    if (proceed && *proceed == 0) return;  // optimistic issue: an earlier panel broke an assumption, nothing behind it may run on its state
    const int tid = threadIdx.x;
    const int64_t rows = w.rows - row0;
    const int64_t rbase = (int64_t)blockIdx.x * 256 * RPT;
    T v[RPT][kNB];
    bool ok[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int64_t r = rbase + tid + 256 * i;
        ok[i] = r < rows;
#pragma unroll
        for (int t = 0; t < kNB; ++t) v[i][t] = (ok[i] && t < kb) ? vp.p[(int64_t)t * vp.cs + vrow0 + r] : (T)0;
    }
    const int64_t c_begin = (int64_t)blockIdx.y * cols_per_wg, c_end = min(w.cols, c_begin + cols_per_wg);
    // (wave-uniform) next column that still has an F row
    auto next_active = [&](int64_t c) {
        while (c < c_end && (pos[c] < j0 + kb || (coop && is_cand[c]))) ++c;
        return c;
    };
    // software pipeline over the columns: the loads of the next active column are in flight while this one is updated.
    // (Measured on 4064 x 4096 f32: 53 us = 2.5 TB/s of the 134 MB against 69 us for the MFMA GEMM; a plain copy of the
    // matrix reaches 3.8 TB/s.  Groups of four columns in flight were slower, 60 us.)
    int64_t c = next_active(c_begin);
    T x[RPT], xn[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) x[i] = (c < c_end && ok[i]) ? w.p[c * w.cs + row0 + rbase + tid + 256 * i] : (T)0;
    while (c < c_end) {
        const int64_t cn = next_active(c + 1);
#pragma unroll
        for (int i = 0; i < RPT; ++i) xn[i] = (cn < c_end && ok[i]) ? w.p[cn * w.cs + row0 + rbase + tid + 256 * i] : (T)0;
        const T *frow = Fm + c * kNB;
#pragma unroll
        for (int t = 0; t < kNB; ++t) {
            const T f = frow[t];
#pragma unroll
            for (int i = 0; i < RPT; ++i) x[i] = fma(-v[i][t], f, x[i]);
        }
        T *col = w.p + c * w.cs + row0;
#pragma unroll
        for (int i = 0; i < RPT; ++i)
            if (ok[i]) col[rbase + tid + 256 * i] = x[i];
#pragma unroll
        for (int i = 0; i < RPT; ++i) x[i] = xn[i];
        c = cn;
    }
}

// diagnostic (RC_QRCP_CHECK=1): jpvt must be a permutation of 0 .. n-1 and pos its inverse; counts the violations
__global__ __launch_bounds__(256) void k_qrb_check_perm(const int64_t *jpvt, const int *pos, int n, int *mark, int *bad) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int64_t c = jpvt[i];
        if (c < 0 || c >= n) { atomicAdd(bad, 1); continue; }
        if (atomicAdd(mark + c, 1) != 0) atomicAdd(bad + 1, 1);
        if (pos[c] != i) atomicAdd(bad + 2, 1);
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

constexpr int kCoopWgs = 64;  // workgroups of a cooperative panel at most (8 candidates each; one header per lane of the polling wave)
template <typename T> constexpr int coop_ne_big() { return sizeof(T) == 8 ? 24 : 16; }  // f64: 96 doubles of slab per lane is what fits beside the working set
template <typename T> constexpr int coop_rows_cap() { return 64 * (16 / (int)sizeof(T)) * coop_ne_big<T>(); }  // f32: 4096 rows, f64: 3072

// units of the device-wide cooperative budget (half CUs) one workgroup of the kernel costs
template <typename K>
static unsigned coop_units_per_wg(K kern) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kern), 512, 0) != hipSuccess) { (void)hipGetLastError(); nb = 1; }
    return nb >= 2 ? 1u : 2u;
}

template <typename T>
static bool qrb_coop_launch(rc_context *c, QrbCoopArgs<T> a, int g, int rows, const int *proceed) {
    constexpr int NE_BIG = coop_ne_big<T>(), NE_SMALL = NE_BIG / 4;
    const bool small = rows <= 64 * (16 / (int)sizeof(T)) * NE_SMALL;
    // f64 with 96 doubles of column per lane needs the whole register file of a SIMD for two waves; everything else leaves room
    // for a second workgroup on the CU
    constexpr int WPE_BIG = sizeof(T) == 8 ? 2 : 4;
    static const unsigned units_big = coop_units_per_wg(k_qrb_coop<T, NE_BIG, WPE_BIG>), units_small = coop_units_per_wg(k_qrb_coop<T, NE_SMALL, 4>);
    a.need = (unsigned)g * (small ? units_small : units_big);
    if (a.need > coop_budget_units(c->device)) return false;  // (a device with few CUs: the step kernels run instead)
    coop_gate_launch(c, a.need, a.sync, a.hdr, 2 * g * 5, proceed);
    if (small) hipLaunchKernelGGL((k_qrb_coop<T, NE_SMALL, 4>), dim3((unsigned)g), dim3(512), 0, c->stream, a);
    else hipLaunchKernelGGL((k_qrb_coop<T, NE_BIG, WPE_BIG>), dim3((unsigned)g), dim3(512), 0, c->stream, a);
    return true;
}

// One factorization as a resumable job: issue() enqueues a panel's kernels and the read-back of its state, finish() -- once
// the stream has been synchronised -- enqueues the panel-end kernels and says whether the factorization is complete.  A host
// that drives several matrices (rc_batch_column_id_*) issues all of them, waits once, finishes all of them: the per-panel
// waits of different matrices overlap.  All buffers come from the context's arena (no reset while the job is alive).
template <typename T>
struct BlockedQrcpJob {
    rc_context *c;
    Mat<T> w;
    int64_t m, n, kmax, j0 = 0, cwant, ldy;
    int64_t *jpvt;
    T *tau;
    int *pos, *cand, *flag;
    unsigned char *is_cand;
    T *vn1, *vn2, *Fm, *Tm, *auxv, *tsc, *Y;
    QrbPanel<T> P;
    QrbState *st;
    QrbState *host_st;  // pinned
    Mat<T> vp;
    int vec_ok;
    int nbp = 0;
    struct PanelRec { int64_t j0; int kb; T *tm; };
    std::vector<PanelRec> panels;  // finished panels with their T factors (block form-Q)
    // cooperative panels (k_qrb_coop)
    QrbCoopArgs<T> coop;
    bool coop_ready = false, coop_issued = false;
    bool keep_t = true;  // the panels' T factors are kept for qrb_form_q (false: the caller needs no Q -- column_id_from_qrcp)
    int coop_fallbacks = 0;
    int64_t cw_issued = 0;
    unsigned grid_a = 0, grid_c = 0;
    // optimistic issue (qrb_issue_all_optimistic): panel p's state goes to host_st[log_base + p]; nbp_log[p] = the steps it was assumed to make
    bool optimistic = false, opt_broken = false;
    int *opt_ok = nullptr;  // device word: 1 while every assumption of the optimistic issue has held (k_qrb_opt_check clears it)
    int log_base = 0, log_n = 0;
    std::vector<int> nbp_log;
};

template <typename T>
BlockedQrcpJob<T> *qrb_begin(rc_context *c, Mat<T> w, int64_t kmax, int64_t *jpvt, T *tau) {
    RC_REQUIRE(w.rs == 1, RC_LAYOUT_ERROR, "geqp3_blocked: working matrix must be column-major");
    RC_REQUIRE(!c->capturing, RC_RUNTIME_ERROR, "geqp3_blocked: reads one scalar back per panel, not capturable");
    auto *J = new BlockedQrcpJob<T>();
    J->c = c; J->w = w; J->jpvt = jpvt; J->tau = tau;
    const int64_t m = w.rows, n = w.cols;
    J->m = m; J->n = n;
    J->kmax = std::min(kmax, std::min(m, n));
    J->pos = c->alloc<int>((size_t)n);
    J->cand = c->alloc<int>((size_t)n);
    J->flag = c->alloc<int>((size_t)n);
    J->is_cand = c->alloc<unsigned char>((size_t)n);
    J->vn1 = c->alloc<T>((size_t)n);
    J->vn2 = c->alloc<T>((size_t)n);
    J->Fm = c->alloc<T>((size_t)n * kNB + (size_t)kNB * kNB);  // F and T in one allocation: one clearing launch
    J->Tm = J->Fm + (size_t)n * kNB;
    J->auxv = c->alloc<T>(kNB);
    J->tsc = c->alloc<T>(4);
    QrbPanel<T> &P = J->P;
    P.jpvt = jpvt; P.pos = J->pos; P.cand = J->cand;
    P.cpos[0] = c->alloc<int>((size_t)n); P.cpos[1] = c->alloc<int>((size_t)n);
    P.cvn = c->alloc<T>((size_t)n);
    P.vn2 = J->vn2; P.Fm = J->Fm;
    P.Rrow = c->alloc<T>((size_t)kNB * n); P.ncap = n;
    P.xbuf = c->alloc<T>((size_t)m + 16); P.pss = c->alloc<T>(128 + 64 * kNB);
    P.flag = J->flag; P.tsc = J->tsc; P.tau = tau; P.Tm = J->Tm;
    J->st = reinterpret_cast<QrbState *>(c->alloc_bytes(sizeof(QrbState)));
    P.st = J->st;
    J->opt_ok = c->alloc<int>(2);
    J->vp = colmajor(c->alloc<T>((size_t)even_ld(m) * kNB), m, kNB, even_ld(m));
    J->ldy = even_ld(n);
    J->Y = c->alloc<T>((size_t)kQrbVtaMaxSplits * kNB * J->ldy);  // up to kQrbVtaMaxSplits partial slabs of kNB x ldy (k_qrb_vta); slab 0 alone for the GEMM path
    if (c->pinned_size < sizeof(QrbState)) {
        if (c->pinned) RC_HIP(hipHostFree(c->pinned));
        c->pinned = nullptr; c->pinned_size = 0;
        RC_HIP(hipHostMalloc(&c->pinned, 1 << 16, hipHostMallocDefault));
        c->pinned_size = 1 << 16;
    }
    J->host_st = reinterpret_cast<QrbState *>(c->pinned);
    // whole F rows are read: keep them finite; the strictly lower part of T stays zero (block form-Q multiplies by the full square)
    fill_words(c, J->Fm, ((size_t)n * kNB + (size_t)kNB * kNB) * sizeof(T), 0u);
    J->vec_ok = (w.cs % (16 / (int64_t)sizeof(T)) == 0 && reinterpret_cast<uintptr_t>(w.p) % 16 == 0) ? 1 : 0;
    hipLaunchKernelGGL(k_qrb_init<T>, dim3((unsigned)std::min<int64_t>(cdivb(n, 4), 8192)), dim3(256), 0, c->stream, w, jpvt, J->pos, J->vn1, J->vn2, J->flag);
    // candidate budget: about RC_QRCP_CAND_MB of column data, at least 4 NB columns.  A cooperative panel holds one wave per candidate,
    // so the budget is also its CU demand: 4 MB = 256 candidates = 32 workgroups for 4096 rows of f32 (8 MB until round 3: 64 workgroups;
    // eight lanes' panels did not fit the device budget at once).  Fewer candidates end a panel earlier on flat norm profiles (the tau
    // test; the next panel then asks for more).  Measured, 8 x 4096^2 f32 Gaussian, k = 64, batch of 8: 2450 (8 MB) -> 2690 (5) -> 2820 (4)
    // -> 2940 (3) matrices/s, 1470 at 2 MB (every panel ends early); one matrix 0.91 -> 0.88 ms.  NOTE: R12 / Z differ in the last bits
    // with the candidate count (candidates are updated reflector by reflector, the other columns through the block update).
    static const int cand_mb = env_int_b("RC_QRCP_CAND_MB", 4);
    J->cwant = std::max<int64_t>(4 * kNB, ((int64_t)cand_mb << 20) / (int64_t)(sizeof(T) * (size_t)std::max<int64_t>(m, 1)));
    if (cand_mb <= 0) J->cwant = n;  // plain ?laqps
    // cooperative panels: header / candidate slots of up to kCoopWgs workgroups, the recorded v_t^T v_k
    static const int coop_on = env_int_b("RC_QRCP_COOP", 1);
    if (coop_on && c->opt_coop_panel && cand_mb > 0) {
        QrbCoopArgs<T> &a = J->coop;
        a.w = w;
        a.P = P;
        a.vn1 = J->vn1; a.vn2 = J->vn2;
        a.mpad = coop_rows_cap<T>();
        a.hdr = c->alloc<unsigned long long>((size_t)2 * kCoopWgs * 5);
        a.cols = c->alloc<T>((size_t)2 * kCoopWgs * a.mpad);
        a.D = c->alloc<T>((size_t)kNB * kNB);
        a.sync = c->alloc<unsigned>(4);
        a.sem = coop_semaphore_of(c->device);
        J->coop_ready = true;
    }
    return J;
}

template <typename T>
static void qrb_issue_launches(BlockedQrcpJob<T> *J, int nbp, int64_t cw, unsigned grid_a, unsigned grid_c) {
    rc_context *c = J->c;
    const int64_t n = J->n, j0 = J->j0;
    hipLaunchKernelGGL(k_qrb_select<T>, dim3(1), dim3(1024), 0, c->stream, (int)n, (int)j0, (int)cw, J->pos, J->vn1, J->cand, J->is_cand, J->st, J->tsc, J->P.cpos[0],
                       J->P.cvn);  // (also resets the panel state)
    for (int k = 0; k < nbp; ++k) {
        hipLaunchKernelGGL(k_qrb_step_a<T>, dim3(grid_a), dim3(256), 0, c->stream, J->w, (int)j0, k, J->P);
        hipLaunchKernelGGL(k_qrb_step_c<T>, dim3(grid_c), dim3(256), 0, c->stream, J->w, (int)j0, k, (int)grid_a, J->vec_ok, J->P);
    }
    RC_HIP(hipMemcpyAsync(J->host_st + (J->optimistic ? J->log_base + J->log_n : 0), J->st, sizeof(QrbState), hipMemcpyDeviceToHost, c->stream));
}

template <typename T>
void qrb_issue(BlockedQrcpJob<T> *J) {
    rc_context *c = J->c;
    const int64_t m = J->m, n = J->n, j0 = J->j0;
    const int nbp = (int)std::min<int64_t>(kNB, J->kmax - j0);
    J->nbp = nbp;
    const int64_t cw = std::min<int64_t>(J->cwant, n - j0);
    // step A: one row slab of the pivot column per workgroup (every workgroup scans all candidate norms: few workgroups
    // when the candidates are many); step C: one workgroup per candidate
    const int64_t cbound = std::min<int64_t>(n - j0, 2 * cw);
    const unsigned grid_a = (unsigned)std::max<int64_t>(4, std::min<int64_t>(std::min<int64_t>(64, cdivb(m - j0, 64)), 65536 / std::max<int64_t>(cbound, 1)));
    const unsigned grid_c = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cbound, 4096));
    J->grid_a = grid_a; J->grid_c = grid_c; J->cw_issued = cw; J->coop_issued = false;
    // Cooperative panel (one launch for all steps, candidates in registers) when the active rows fit the registers of a
    // wave; fewer candidates are requested than the launch has waves, because ties at the threshold all become candidates
    if (J->coop_ready && J->coop_fallbacks < 4 && m - j0 <= coop_rows_cap<T>() && m - j0 >= 8) {
        const int64_t cw_c = std::min<int64_t>(cw, 8 * kCoopWgs - 32);
        const int g = (int)std::min<int64_t>(kCoopWgs, cdivb(cw_c + std::max<int64_t>(8, cw_c / 16), 8));
        hipLaunchKernelGGL(k_qrb_select<T>, dim3(1), dim3(1024), 0, c->stream, (int)n, (int)j0, (int)cw_c, J->pos, J->vn1, J->cand, J->is_cand, J->st, J->tsc, J->P.cpos[0],
                           J->P.cvn);
        QrbCoopArgs<T> a = J->coop;
        a.j0 = (int)j0;
        a.nbp = nbp;
        if (qrb_coop_launch<T>(c, a, g, (int)(m - j0), J->optimistic ? J->opt_ok : nullptr)) {
            RC_HIP(hipMemcpyAsync(J->host_st + (J->optimistic ? J->log_base + J->log_n : 0), J->st, sizeof(QrbState), hipMemcpyDeviceToHost, c->stream));
            J->coop_issued = true;
            J->cw_issued = cw_c;
            return;
        }
        J->coop_ready = false;  // does not fit this device: step kernels from here on (they redo the selection)
    }
    // A panel is ~65 dependent launches at ~3.5 us of host time each: replay them from a hipGraph, cached on the context
    // under everything the launches bake in (the arena hands out the same addresses for the same call sequence, so a
    // host that compresses many same-shaped matrices replays).  Opt-in (RC_QRCP_GRAPH=1): measured on MI355X the replay does not
    // beat eager issue -- with 8 matrices in flight the path is bound by the command processor (~3-4 us per kernel over all
    // streams), not by the host: 1085 matrices/s replayed vs 1207 eager for 8 x (4096 x 4096 f32, k = 64).
    static const int use_graph = env_int_b("RC_QRCP_GRAPH", 0);
    // (only for factorizations of a few panels -- the truncated rank-k case: a long one has a different (j0, candidates)
    // at every panel and would capture each graph for a single use)
    if (!use_graph || c->prof_on || J->kmax > 4 * kNB) { qrb_issue_launches(J, nbp, cw, grid_a, grid_c); return; }
    const std::vector<uint64_t> key = {(uint64_t)(uintptr_t)J->w.p, (uint64_t)J->w.cs, (uint64_t)m, (uint64_t)n, (uint64_t)(uintptr_t)J->jpvt, (uint64_t)(uintptr_t)J->tau,
                                       (uint64_t)(uintptr_t)J->st, (uint64_t)(uintptr_t)J->host_st, (uint64_t)j0, (uint64_t)nbp, (uint64_t)cw, (uint64_t)sizeof(T),
                                       (uint64_t)(uintptr_t)c->stream};
    auto it = c->qrb_graphs.find(key);
    if (it == c->qrb_graphs.end()) {
        if (c->qrb_graphs.size() >= 64) {  // bounded: a host cycling through many shapes simply re-captures
            for (auto &kv : c->qrb_graphs) (void)hipGraphExecDestroy(kv.second);
            c->qrb_graphs.clear();
        }
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); qrb_issue_launches(J, nbp, cw, grid_a, grid_c); return; }
        bool ok = true;
        try { qrb_issue_launches(J, nbp, cw, grid_a, grid_c); } catch (const Error &) { ok = false; }
        if (hipStreamEndCapture(c->stream, &graph) != hipSuccess || !graph) ok = false;
        if (ok && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) ok = false;
        if (graph) (void)hipGraphDestroy(graph);
        if (!ok) { (void)hipGetLastError(); qrb_issue_launches(J, nbp, cw, grid_a, grid_c); return; }
        it = c->qrb_graphs.emplace(key, exec).first;
    }
    RC_HIP(hipGraphLaunch(it->second, c->stream));
}

// to be called after the context's stream has been synchronised since qrb_issue(); true = factorization complete
template <typename T>
static bool qrb_finish_with(BlockedQrcpJob<T> *J, QrbState h, bool coop, bool optimistic);

template <typename T>
bool qrb_finish(BlockedQrcpJob<T> *J) {
    rc_context *c = J->c;
    QrbState h = *J->host_st;
    bool coop = false;
    if (J->coop_issued) {
        if (h.pad0 == 1) {
            coop = true;
        } else {
            // the cooperative launch left the panel untouched (2: more candidates than its waves, 3: co-residency time-out):
            // run the panel through the step kernels (rare; costs one more wait)
            J->coop_fallbacks += (h.pad0 == 2) ? 1 : 4;
            qrb_issue_launches(J, J->nbp, J->cw_issued, J->grid_a, J->grid_c);
            RC_HIP(hipStreamSynchronize(c->stream));
            h = *J->host_st;
        }
        J->coop_issued = false;
    }
    return qrb_finish_with(J, h, coop, false);
}

// the panel-end kernels for a panel whose state is h (read back by the host, or ASSUMED by the optimistic issue: cooperative
// launch completed, all nbp steps made, some column is not a candidate; the norms flag is read on the device)
template <typename T>
static bool qrb_finish_with(BlockedQrcpJob<T> *J, QrbState h, bool coop, bool optimistic) {
    rc_context *c = J->c;
    const int64_t m = J->m, n = J->n, j0 = J->j0;
    const int kb = h.stopped ? h.kb : J->nbp;
    RC_REQUIRE(kb >= 1 && kb <= J->nbp, RC_PIVOTED_QR_ERROR, "geqp3_blocked: panel at %lld made %d steps", (long long)j0, kb);
    static const int check = env_int_b("RC_QRCP_CHECK", 0);
    if (check && !optimistic) {
        ArenaMark mk(c);
        int *mark = c->alloc<int>((size_t)n + 4);
        fill_words(c, mark, ((size_t)n + 4) * sizeof(int), 0u);
        hipLaunchKernelGGL(k_qrb_check_perm, dim3((unsigned)std::min<int64_t>(cdivb(n, 256), 1024)), dim3(256), 0, c->stream, J->jpvt, J->pos, (int)n, mark, mark + n);
        int bad[3] = {0, 0, 0};
        RC_HIP(hipMemcpyAsync(bad, mark + n, sizeof(bad), hipMemcpyDeviceToHost, c->stream));
        RC_HIP(hipStreamSynchronize(c->stream));
        if (bad[0] || bad[1] || bad[2]) {
            fprintf(stderr, "RC_QRCP_CHECK: INVALID permutation after panel j0=%lld kb=%d/%d coop=%d pad0=%d ncand=%d noncand=%d stop_tau=%d: out of range %d, duplicates %d, pos mismatches %d\n",
                    (long long)j0, kb, J->nbp, coop ? 1 : 0, h.pad0, h.ncand, h.have_noncand, h.stop_tau, bad[0], bad[1], bad[2]);
            fail(RC_PIVOTED_QR_ERROR, "geqp3_blocked: permutation invalid after the panel at %lld", (long long)j0);
        }
    }
    const int64_t rows = m - j0;
    const bool last = j0 + kb >= J->kmax;
    const int *okp = optimistic ? J->opt_ok : nullptr;
    Mat<T> w = J->w;
    Mat<T> vpp = Mat<T>(J->vp.p, rows, kb, 1, J->vp.cs);
    hipLaunchKernelGGL(k_qrb_build_vp<T>, dim3((unsigned)std::min<int64_t>(cdivb(rows, 256), 64), (unsigned)kb), dim3(256), 0, c->stream, w, (int)j0, J->st, vpp, okp);
    int ysplits = 1;
    if (h.have_noncand) {
        // Y = V^T A(j0:m, :) for every column: one read pass; used for the non-candidates only.  f32 with 16-byte aligned columns: the
        // streaming kernel k_qrb_vta (partial slabs, summed by k_qrb_finish); otherwise the MFMA GEMM
        if (!qrb_vta_launch<T>(c, vpp, w.sub(j0, rows, 0, n), kb, J->Y, J->ldy, (int64_t)kNB * J->ldy, &ysplits, okp)) {
            ysplits = 1;
            Mat<T> ym = rowmajor(J->Y, kb, n, J->ldy);
            gemm<T>(c, 1, vpp.t(), w.sub(j0, rows, 0, n), 0, ym);
        }
    }
    if (coop) hipLaunchKernelGGL(k_qrb_build_t<T>, dim3(1), dim3(1024), 0, c->stream, J->coop.D, J->tau, (int)j0, kb, J->Tm, okp);
    else hipLaunchKernelGGL(k_qrb_scatter<T>, dim3((unsigned)cdivb(std::max(h.ncand, 1), 256)), dim3(256), 0, c->stream, w, (int)j0, kb, J->P, J->vn1);
    hipLaunchKernelGGL(k_qrb_finish<T>, dim3((unsigned)cdivb(n, 256)), dim3(256), 0, c->stream, w, (int)j0, kb, J->pos, J->is_cand, J->vn1, J->vn2, J->Fm, J->Y, J->ldy,
                       J->Tm, vpp, J->flag, coop ? 1 : 0, ysplits, (int64_t)kNB * J->ldy, okp);
    if (!last && rows - kb > 0) {
        // block update of everything below the panel, written as the transposed product so that the lanes of the
        // MFMA accumulator run along the column-major matrix' contiguous dimension:
        //   A(j0+kb:m, :)^T -= F(:, 0:kb) V(kb:, 0:kb)^T
        static const int stream_upd = env_int_b("RC_QRCP_STREAM_UPDATE", 1);
        if (stream_upd) {
            // streaming rank-kb update (k_qrb_block_update): rows in strips of 256 * RPT, columns in strips sized for ~4 waves of
            // workgroups on the chip
            constexpr int RPT = sizeof(T) == 8 ? 1 : 2;  // 64 registers of V per lane: four waves per SIMD
            const int64_t ur = rows - kb;
            const unsigned gx = (unsigned)cdivb(ur, 256 * RPT);
            static const int upd_wgs = env_int_b("RC_QRCP_UPDATE_WGS", 1024);  // every workgroup re-reads its rows of V: wide column strips amortise it
            const int cols_per_wg = (int)std::max<int64_t>(8, cdivb(n, std::max<int64_t>(1, upd_wgs / gx)));
            hipLaunchKernelGGL((k_qrb_block_update<T, RPT>), dim3(gx, (unsigned)cdivb(n, cols_per_wg)), dim3(256), 0, c->stream, w, (int)(j0 + kb), kb, (int)j0, vpp,
                               kb, J->Fm, J->pos, J->is_cand, coop ? 1 : 0, cols_per_wg, okp);
        } else {
            Mat<T> ft = Mat<T>(J->Fm, n, kb, kNB, 1);
            gemm<T>(c, (T)-1, ft, vpp.sub(kb, rows - kb, 0, kb).t(), (T)1, w.sub(j0 + kb, rows - kb, 0, n).t());
        }
        // ?laqps recomputes the norms it flagged.  When a CANDIDATE lost its accuracy (the panel ended for it) every
        // unpivoted column is recomputed instead: on matrices with a steadily decaying spectrum all columns drift towards the
        // accuracy threshold together (vn1 / vn2 shrinks at the same rate everywhere), and recomputing them one flag at a
        // time would end a panel after every single step for hundreds of steps (which is what LAPACK's own ?geqp3 does there)
        hipLaunchKernelGGL(k_qrb_renorm<T>, dim3((unsigned)std::min<int64_t>(cdivb(n, 4), 4096)), dim3(256), 0, c->stream, w, (int)(j0 + kb), h.lsticc, J->pos, J->flag,
                           J->vn1, J->vn2, optimistic ? J->st : (const QrbState *)nullptr, okp);
    }
    // a panel that the tau test ended early means the candidate set was too small for this spectrum
    if (J->keep_t) {   // keep the panel's T factor for the block form-Q
        T *tsave = c->alloc<T>((size_t)kNB * kNB);
        RC_HIP(hipMemcpyAsync(tsave, J->Tm, (size_t)kNB * kNB * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
        J->panels.push_back({j0, kb, tsave});
    }
    static const int dbg = env_int_b("RC_QRCP_DEBUG", 0);
    if (dbg) fprintf(stderr, "qrb panel w=%p j0=%lld kb=%d/%d ncand=%d noncand=%d lsticc=%d stop_tau=%d cwant=%lld coop=%d pad0=%d fallbacks=%d\n", (void *)J->w.p, (long long)j0, kb, J->nbp, h.ncand, h.have_noncand, h.lsticc, h.stop_tau, (long long)J->cwant, coop ? 1 : 0, h.pad0, J->coop_fallbacks);
    if (h.stop_tau && kb < J->nbp) J->cwant = std::min<int64_t>(n, kb < J->nbp / 2 ? J->cwant * 2 : J->cwant * 3 / 2);
    J->j0 += kb;
    return J->j0 >= J->kmax;
}

// ---------------------------------------------------------------------------------------------------------------------
// Optimistic issue.  The host reads a panel's state back only to learn what almost always holds for a truncated factorization
// of a few panels (cfg5: two): the cooperative launch completed (pad0 == 1) and made all its steps.  Here ALL panels and their
// panel ends are enqueued back to back on those assumptions, each panel's state is copied to its own pinned slot, and the host
// checks the slots ONCE afterwards (qrb_verify_optimistic): one wait per factorization instead of one per panel, and in a batch
// no wait at all until every matrix has been issued.  A failed check means the working matrix is garbage: the caller restores it
// from its source and runs the per-panel path (qrb_issue / wait / qrb_finish), which handles every case.
// Needs: cooperative panels for every panel of the job, no T factors kept (no Q wanted), no per-panel diagnostics.
template <typename T>
bool qrb_optimistic_possible(BlockedQrcpJob<T> *J) {
    static const int on = env_int_b("RC_QRCP_OPTIMISTIC", 1), check = env_int_b("RC_QRCP_CHECK", 0), graph = env_int_b("RC_QRCP_GRAPH", 0);
    if (!on || check || graph || !J->coop_ready || J->keep_t || J->j0 != 0) return false;
    const int64_t panels = cdivb(J->kmax, kNB);
    if (panels < 1 || panels > 16) return false;
    // every panel must take the cooperative path: rows fit a wave's registers (qrb_issue's own test) at the panel's first row
    const int64_t rows_first = J->m, rows_last = J->m - (panels - 1) * kNB;
    return rows_first <= coop_rows_cap<T>() && rows_last >= 8;
}

// after a cooperative panel of the optimistic issue: did it complete and make the steps the host assumed?  (sticky: once cleared,
// every kernel enqueued behind it on these assumptions returns at once, and the gates of the later panels do not open)
__global__ void k_qrb_opt_check(const QrbState *st, int assumed_kb, int *ok) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && (st->pad0 != 1 || (st->stopped && st->kb != assumed_kb))) *ok = 0;
}

// slots of pinned state this context hands to optimistic jobs before somebody has to wait (rc_batch resets the cursor per call)
template <typename T>
bool qrb_issue_all_optimistic(BlockedQrcpJob<T> *J) {
    rc_context *c = J->c;
    const int panels = (int)cdivb(J->kmax, kNB);
    const int cap = (int)(c->pinned_size / sizeof(QrbState));
    if (c->pinned_cursor + panels > cap) return false;  // the caller waits, checks what is pending and resets the cursor
    J->optimistic = true;
    J->log_base = c->pinned_cursor;
    J->log_n = 0;
    c->pinned_cursor += panels;
    fill_words(c, J->opt_ok, 2 * sizeof(int), 1u);
    for (int p = 0; p < panels; ++p) {
        qrb_issue(J);
        if (!J->coop_issued) {  // (the device budget shrank: the step kernels were enqueued instead -- nothing behind them may run)
            fill_words(c, J->opt_ok, 2 * sizeof(int), 0u);
            J->opt_broken = true;
            return true;
        }
        hipLaunchKernelGGL(k_qrb_opt_check, dim3(1), dim3(64), 0, c->stream, J->st, J->nbp, J->opt_ok);
        J->coop_issued = false;
        J->nbp_log.push_back(J->nbp);
        QrbState h;
        memset(&h, 0, sizeof(h));
        h.pad0 = 1;
        h.have_noncand = 1;
        J->log_n = p + 1;
        if (qrb_finish_with(J, h, true, true)) break;
    }
    return true;
}

// after the stream has been waited for: did every assumption hold?
template <typename T>
bool qrb_verify_optimistic(const QrbState *host_log, const std::vector<int> &nbp_log, bool broken) {
    if (broken) return false;
    for (size_t p = 0; p < nbp_log.size(); ++p) {
        const QrbState &h = host_log[p];
        if (h.pad0 != 1) return false;                       // the cooperative launch gave up (time-out, more candidates than waves)
        if (h.stopped && h.kb != nbp_log[p]) return false;   // the panel ended early (tau test)
    }
    return true;
}
template <typename T>
bool qrb_verify_optimistic(BlockedQrcpJob<T> *J) { return qrb_verify_optimistic<T>(J->host_st + J->log_base, J->nbp_log, J->opt_broken); }
template <typename T>
void qrb_optimistic_log(BlockedQrcpJob<T> *J, const QrbState **log, std::vector<int> *nbp_log, bool *broken) {
    *log = J->host_st + J->log_base; *nbp_log = J->nbp_log; *broken = J->opt_broken;
}

// Q(:, 0:kq) = H_0 ... H_{k-1} [I ; 0] from the finished job, panel by panel from the last to the first:
//   X <- (I - V_p T_p V_p^T) X  on rows j0_p .. m-1, columns j0_p .. kq-1 (the other columns are still unit vectors above row j0_p)
// = three MFMA GEMMs per panel (?orgqr's blocked form with the T factors the panels already built).  q: m x kq column-major.
template <typename T>
void qrb_form_q(BlockedQrcpJob<T> *J, Mat<T> q) {
    rc_context *c = J->c;
    RC_REQUIRE(q.rs == 1 && q.rows == J->m, RC_LAYOUT_ERROR, "qrb_form_q: column-major m x kq output required");
    if (q.empty()) return;
    ProfScope ps(c, "op:form_q_blocked %lldx%lld panels=%d", (long long)J->m, (long long)q.cols, (int)J->panels.size());
    ArenaMark mark(c);
    fill_identity(c, q);
    const int64_t m = J->m, kq = q.cols;
    T *w2p = c->alloc<T>((size_t)kNB * even_ld(kq));
    T *w3p = c->alloc<T>((size_t)kNB * even_ld(kq));
    for (int pi = (int)J->panels.size() - 1; pi >= 0; --pi) {
        const auto &pr = J->panels[(size_t)pi];
        if (pr.j0 >= kq) continue;  // reflectors beyond the requested columns act on zero rows only
        const int64_t rows = m - pr.j0, ncols = kq - pr.j0;
        const int kb = pr.kb;
        Mat<T> vpp = Mat<T>(J->vp.p, rows, kb, 1, J->vp.cs);
        // the panel's pivot columns: positions j0 .. j0+kb-1 of jpvt (device) -> build V from them
        hipLaunchKernelGGL(k_qrb_build_vp_pos<T>, dim3((unsigned)std::min<int64_t>(cdivb(rows, 256), 64), (unsigned)kb), dim3(256), 0, c->stream, J->w, (int)pr.j0, J->jpvt, vpp);
        Mat<T> qs = q.sub(pr.j0, rows, pr.j0, ncols);
        Mat<T> w2 = rowmajor(w2p, kb, ncols, even_ld(kq)), w3 = rowmajor(w3p, kb, ncols, even_ld(kq));
        gemm<T>(c, 1, vpp.t(), qs, 0, w2);
        gemm<T>(c, 1, Mat<T>(pr.tm, kb, kb, 1, kNB), w2, 0, w3);  // T is upper triangular with exact zeros below
        gemm<T>(c, (T)-1, vpp, w3, (T)1, qs);
    }
}

template <typename T>
void qrb_end(BlockedQrcpJob<T> *J) { delete J; }
template <typename T>
void qrb_keep_t(BlockedQrcpJob<T> *J, bool keep) { J->keep_t = keep; }

// w: m x n column-major working matrix (overwritten with the ?geqp3 output format: R on and above the
// diagonal in position order, reflectors below, columns never moved); jpvt: n; tau: kmax
template <typename T>
void geqp3_blocked(rc_context *c, Mat<T> w, int64_t kmax, int64_t *jpvt, T *tau, Mat<T> q_out, Mat<T> restore_from) {
    if (std::min(kmax, std::min(w.rows, w.cols)) <= 0) return;
    ProfScope ps(c, "op:geqp3_blocked %lldx%lld k=%lld", (long long)w.rows, (long long)w.cols, (long long)kmax);
    if (restore_from.p && q_out.empty()) {
        // all panels enqueued on the usual outcome, ONE wait, then the check (see qrb_issue_all_optimistic)
        ArenaMark mark(c);
        BlockedQrcpJob<T> *J = qrb_begin<T>(c, w, kmax, jpvt, tau);
        struct Guard { BlockedQrcpJob<T> *j; ~Guard() { qrb_end(j); } } guard{J};
        J->keep_t = false;
        c->pinned_cursor = 0;
        if (qrb_optimistic_possible(J) && qrb_issue_all_optimistic(J)) {
            RC_HIP(hipStreamSynchronize(c->stream));
            if (qrb_verify_optimistic(J)) return;
            copy_mat(c, restore_from, w);  // an assumption failed: start over on a fresh copy, panel by panel
        }
    }
    ArenaMark mark(c);
    BlockedQrcpJob<T> *J = qrb_begin<T>(c, w, kmax, jpvt, tau);
    struct Guard { BlockedQrcpJob<T> *j; ~Guard() { qrb_end(j); } } guard{J};
    J->keep_t = !q_out.empty();
    for (;;) {
        qrb_issue(J);
        RC_HIP(hipStreamSynchronize(c->stream));
        if (qrb_finish(J)) break;
    }
    if (!q_out.empty()) qrb_form_q(J, q_out);
}

#ifdef RC_QRC_TIMING
extern "C" void rc_debug_qrc_timing(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qrc_dbg), 8 * sizeof(unsigned long long)); }
#endif
#define RC_INST_JOB(T)                                                                           \
    template BlockedQrcpJob<T> *qrb_begin<T>(rc_context *, Mat<T>, int64_t, int64_t *, T *);     \
    template void qrb_issue<T>(BlockedQrcpJob<T> *);                                             \
    template bool qrb_finish<T>(BlockedQrcpJob<T> *);                                            \
    template void qrb_form_q<T>(BlockedQrcpJob<T> *, Mat<T>);                                    \
    template void qrb_end<T>(BlockedQrcpJob<T> *);                                               \
    template void qrb_keep_t<T>(BlockedQrcpJob<T> *, bool);                                      \
    template bool qrb_optimistic_possible<T>(BlockedQrcpJob<T> *);                               \
    template bool qrb_issue_all_optimistic<T>(BlockedQrcpJob<T> *);                              \
    template bool qrb_verify_optimistic<T>(BlockedQrcpJob<T> *);                                 \
    template bool qrb_verify_optimistic<T>(const QrbState *, const std::vector<int> &, bool);    \
    template void qrb_optimistic_log<T>(BlockedQrcpJob<T> *, const QrbState **, std::vector<int> *, bool *);
RC_INST_JOB(double)
RC_INST_JOB(float)
#undef RC_INST_JOB

template bool geqp3_blocked_supported<double>(int64_t, int64_t, int64_t);
template bool geqp3_blocked_supported<float>(int64_t, int64_t, int64_t);
template void geqp3_blocked<double>(rc_context *, Mat<double>, int64_t, int64_t *, double *, Mat<double>, Mat<double>);
template void geqp3_blocked<float>(rc_context *, Mat<float>, int64_t, int64_t *, float *, Mat<float>, Mat<float>);

}  // namespace rc
