// Small-core SVD for gfx950: one-sided (Hestenes) Jacobi on a square n x n
// matrix, n <= 1024.
//
// Replaces the dense core of LAPACK ?gesdd as the reference reaches it through
// ndarray-linalg `svddc_into(JobSvd::Some)` (/root/reference/src/compute_svd.rs:19).
// The tall/wide input is first reduced to its square triangular factor by the
// Householder QR of kernels_qr.hip (rc_api.hip: svd_core), so this file only
// ever sees min(m, n) x min(m, n).  One-sided Jacobi computes every singular
// value to high RELATIVE accuracy, which is what the f64 <= 1e-12 round-trip
// bound of the reference tests (src/svd.rs:290-297) needs.
//
// MI355X mapping
//  * k_jacobi_lds: the whole core lives in the 160 KiB LDS of ONE CU (128 x 128
//    f64 = 128 KiB).  Round-robin (circle) ordering gives n/2 independent column
//    pairs per round; 16 lanes (one DPP row) own a pair, keep both columns in
//    registers between the three dot products and the rotation (one LDS read +
//    one LDS write of the pair per round), reduce with DPP row operations (no
//    LDS traffic for the reductions) and the 1024-thread workgroup needs ONE
//    barrier per round.  The right singular vectors are NOT accumulated here:
//    every rotation (c, s) is appended to a log in HBM (write-only stream).
//  * k_jacobi_replay_v: V = product of the logged rotations.  Rotations act on
//    columns, so every ROW of V is independent: one wave per row keeps its row
//    in LDS and streams the log -- no barrier at all, 128 waves in parallel.
//  * k_jacobi_global: fallback for cores that do not fit LDS (n <= 1024),
//    everything in L2-resident global memory.
#include "rc_common.hpp"
#include "rc_device.hpp"

#include <cstdlib>

namespace rc {

template <typename T> struct JEps;
template <> struct JEps<double> { static __device__ inline double eps() { return 1.1102230246251565e-16; } };
template <> struct JEps<float> { static __device__ inline float eps() { return 5.9604644775390625e-08f; } };

template <typename T> __device__ inline T dpp_row_sum(T v) { return group_sum_dpp<16>(v); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for
// outstanding GLOBAL stores (vmcnt(0)); the rotation-log stores are write-only and
// must stay in flight across rounds, so the round barrier waits for lgkmcnt alone.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// circle-method pairing: round r of N-1 (N even), pair slot pi of N/2
__device__ inline void rr_pair(int N, int r, int pi, int &p, int &q) {
    if (pi == 0) { p = N - 1; q = r; }
    else { p = (r + pi) % (N - 1); q = (r - pi + (N - 1)) % (N - 1); }
    if (p > q) { int t = p; p = q; q = t; }
}

// Rotation that annihilates the (p, q) entry of the Gram matrix: with d = aqq - app, h = 2 apq (zeta = d / h)
//   t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)) = sign(d h) |h| / (|d| + sqrt(d^2 + h^2)),   c = 1 / sqrt(1 + t^2),  s = c t
// -- the second form has one reciprocal less on the round's dependent chain.  t only decides how completely the entry is
// annihilated, so its reciprocal and root take ONE Newton step (1e-15); c decides the orthogonality of the rotation and keeps two.
// Always evaluated in f64: with f32 parameters c^2 + s^2 - 1 has a systematic sign, and the thousands of rotations a
// column goes through inflate the singular values (3e-5 at n = 300).  A tiny angle (|d| >> |h|) is safe: t -> 0.
template <typename T>
__device__ inline void jacobi_rotation(T app, T aqq, T apq, T &c, T &s, T *t_out = nullptr) {
    const double d = (double)aqq - (double)app, h = 2.0 * (double)apq;
    const double w = fma(d, d, h * h);
    double ri = __builtin_amdgcn_rsq(w);
    ri = ri * fma(-0.5 * w, ri * ri, 1.5);
    const double den = fabs(d) + w * ri;
    double rd = __builtin_amdgcn_rcp(den);
    rd = fma(fma(-den, rd, 1.0), rd, rd);
    const double t = copysign(fabs(h) * rd, d * h);
    const double cd = fast_rsqrt(fma(t, t, 1.0));
    c = (T)cd;
    s = (T)(cd * t);
    if (t_out) *t_out = (T)t;
}

// one rotation record of the log
template <typename T> struct Rot { T c, s; };

// Fused right-vector accumulation (second workgroup of k_jacobi_lds): the records travel from the producer to the
// consumer workgroup through agent-scope atomics and the producer never waits for its stores.  Every record has a
// check word hash(c, s) ^ magic ^ f(epoch); the epoch is a per-context device counter that both workgroups read at
// their start and the consumer bumps at its end, so a torn combination, or a complete record of an EARLIER launch
// that still sits at the same workspace address, never validates -- nothing has to be cleared before a launch.
#define RC_AGENT __HIP_MEMORY_SCOPE_AGENT
constexpr unsigned long long kRotMagic = 0x9e3779b97f4a7c15ull;
__device__ inline unsigned long long rot_hash(Rot<double> r) { return (unsigned long long)__double_as_longlong(r.c) ^ ((unsigned long long)__double_as_longlong(r.s) * 3ull); }
__device__ inline unsigned long long rot_hash(Rot<float> r) { return ((unsigned long long)__float_as_uint(r.s) << 32) | __float_as_uint(r.c); }
__device__ inline unsigned long long epoch_key(unsigned e) { return kRotMagic ^ ((unsigned long long)e * 0xd1342543de82ef95ull); }
__device__ inline void rot_publish(Rot<double> *p, unsigned long long *chk, Rot<double> r, unsigned long long key) {
    __hip_atomic_store(&p->c, r.c, __ATOMIC_RELAXED, RC_AGENT);
    __hip_atomic_store(&p->s, r.s, __ATOMIC_RELAXED, RC_AGENT);
    __hip_atomic_store(chk, rot_hash(r) ^ key, __ATOMIC_RELAXED, RC_AGENT);
}
__device__ inline void rot_publish(Rot<float> *p, unsigned long long *chk, Rot<float> r, unsigned long long key) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), rot_hash(r), __ATOMIC_RELAXED, RC_AGENT);
    __hip_atomic_store(chk, rot_hash(r) ^ key, __ATOMIC_RELAXED, RC_AGENT);
}
__device__ inline bool rot_fetch(const Rot<double> *p, const unsigned long long *chk, Rot<double> &r, unsigned long long key) {
    r.c = __hip_atomic_load(&p->c, __ATOMIC_RELAXED, RC_AGENT);
    r.s = __hip_atomic_load(&p->s, __ATOMIC_RELAXED, RC_AGENT);
    return __hip_atomic_load(chk, __ATOMIC_RELAXED, RC_AGENT) == (rot_hash(r) ^ key);
}
__device__ inline bool rot_fetch(const Rot<float> *p, const unsigned long long *chk, Rot<float> &r, unsigned long long key) {
    const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, RC_AGENT);
    r.c = __uint_as_float((unsigned)w);
    r.s = __uint_as_float((unsigned)(w >> 32));
    return __hip_atomic_load(chk, __ATOMIC_RELAXED, RC_AGENT) == (w ^ key);
}
// small integers (sweep count, sorted position + 1) travel as (epoch << 8) | payload
__device__ inline void tagged_put(unsigned *p, unsigned e, unsigned payload) { __hip_atomic_store(p, (e << 8) | payload, __ATOMIC_RELAXED, RC_AGENT); }
__device__ inline unsigned tagged_get(const unsigned *p, unsigned e) {  // 0 = not there yet
    const unsigned w = __hip_atomic_load(p, __ATOMIC_RELAXED, RC_AGENT);
    return (w >> 8) == (e & 0xffffffu) ? (w & 0xffu) : 0u;
}

__global__ void k_clear_words(unsigned *p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}

// ---------------------------------------------------------------------------
// LDS-resident one-sided Jacobi.  LPP lanes own one column pair and keep NE = ceil(n / LPP)
// rows of both columns in registers; with LPP = 4 a 128 x 128 core needs only 4 waves
// (one per SIMD), which minimises the per-pair overhead (reductions + rotation
// parameters are paid per wave instruction, not per element).
//   g      : n x n column-major input (global), destroyed
//   log    : [max_sweeps][N-1][N/2] rotations (c = 1, s = 0 where none)
//   sweeps : number of sweeps performed (device scalar out)
//   uc, s  : left singular vectors / singular values, sorted descending
//   order  : order[j] = sorted position of original column j (for the V replay)
// ---------------------------------------------------------------------------
//   fused  : != 0: launched with TWO workgroups; the second one accumulates V from the published records while the first
//            is still rotating (vsync[0] = number of sweeps once known, vsync[1 + j] = order[j] + 1; both zeroed before)
// RC_JAC_ABL: timing ablations of the producer's round (diagnostic builds only, tools/jacobi_ablation.sh; results are WRONG):
// 1 no rotation records, 2 no rcp/rsqrt in the rotation, 4 no write-back, 8 no group reduction, 16 no LDS reads, 32 rotate always
#ifndef RC_JAC_ABL
#define RC_JAC_ABL 0
#endif
// RC_JAC_TIMING: s_memtime stamps of wave 0 of the producer, summed per phase of the round (diagnostic builds only,
// tools/jacobi_timing.py; a stamp costs a few hundred cycles and drains the wave's LDS queue)
#ifdef RC_JAC_TIMING
__device__ unsigned long long g_jac_dbg[8];
#define RC_JTICK(k)                                                                      \
    {                                                                                    \
        unsigned long long now_;                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");  \
        jt[k] += now_ - jlast;                                                           \
        jlast = now_;                                                                    \
    }
#else
#define RC_JTICK(k)
#endif
// PROTOCOL INVARIANTS of the fused launch (two workgroups; reviewed against the code in round 3 -- keep list and code in step)
//  J1  One direction only: workgroup 0 (producer) publishes, workgroup 1 (consumer) reads; the producer never waits for the
//      consumer, so the two need not be co-resident and the producer's result (U, S) never depends on the consumer.
//  J2  Everything that crosses is an 8-byte (records, check words) or 4-byte (vsync) agent-scope relaxed atomic.  A rotation record
//      is valid iff its check word equals hash(c, s) ^ key(epoch): a torn combination of two publications and a complete record of
//      an EARLIER launch at the same workspace address (other epoch, other key) both fail the test, so records need no clearing.
//  J3  epoch is a per-context device counter, started at a pseudo-random 23-bit value, read by both workgroups at their start and
//      incremented by the consumer at its very end; launches of one context are stream ordered, so every launch sees a new epoch.
//  J4  The small hand-over words vsync[0] (number of sweeps) and vsync[1 + j] (sorted position + 1) travel as (epoch << 8) | payload
//      AND are cleared (k_clear_words, same stream, in front) before every launch: payload 0 means "not there yet".
//  J5  The consumer learns that sweep s exists from the first record of sweep s validating, and that it does not from
//      vsync[0] <= s; the producer writes exactly one of the two after sweep s - 1.  Index of a record: (sweep, round, pair slot),
//      the same expression on both sides (FULL: pair slot = group).
//  J6  Every consumer spin is bounded (kSpin); on expiry health bit 8 is raised and V is reported incomplete -- never silently wrong.
//   FULL   : n == LPP * NE and one group per pair slot: no row / column bounds, no slot loop (the round is bound by the
//            number of instructions the 16 waves issue, and the predicates were a quarter of them)
//   CN     : squared column norms are carried in LDS and updated by the rotation (app -= t apq, aqq += t apq) instead of
//            being recomputed by every pair in every round; refreshed from the columns at the start of each sweep
template <typename T, int LPP, int NE, bool FULL, bool CN>
__global__ __launch_bounds__(LPP == 16 ? 1024 : 512) void k_jacobi_lds(Mat<T> g, Rot<T> *log, int *sweeps_out, Mat<T> uc, T *s, int *order_out, int max_sweeps,
                                                                       int fused, unsigned *vsync, unsigned long long *chk, unsigned *epoch_p, Mat<T> vc, int *health, int ld) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int n = (int)g.rows;
    // ld: column pitch chosen by the host (jacobi_pitch): the two column groups of a 32-lane half read neighbouring columns
    T *G = reinterpret_cast<T *>(smem_raw);
    T *sig = G + (size_t)ld * n;
    int *order = reinterpret_cast<int *>(sig + n);
    // a static LDS word: behind a pointer into the dynamic array the compiler lost the address space and issued FLAT
    // stores + s_waitcnt vmcnt(0) for it, which also waited for the rotation records in flight
    __shared__ int sh_rot_word;
#define sh_rot sh_rot_word
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int ll = tid % LPP, grp = tid / LPP, ngrp = nthr / LPP;
    const int N = (n + 1) & ~1;
    const int npairs = N / 2;
    const unsigned epoch = fused ? __hip_atomic_load(epoch_p, __ATOMIC_RELAXED, RC_AGENT) & 0xffffffu : 0u;
    const unsigned long long key = epoch_key(epoch);
    if (fused && blockIdx.x == 1) {
        // ---- consumer: V = product of the rotations, columns in LDS, the producer's pairing (one slot per group) ----
        constexpr int kSpin = 1 << 24;
        T *V = G;
        for (int e = tid; e < n * n; e += nthr) {
            const int i = e % n, j = e / n;
            V[j * ld + i] = (i == j) ? (T)1 : (T)0;
        }
        __syncthreads();
        bool lost = false;
        for (int sweep = 0;; ++sweep) {
            if (tid == 0) {  // has the producer started this sweep, or did it finish before it?
                int fin = 2;
                for (int it = 0; it < kSpin; ++it) {
                    const unsigned d = tagged_get(vsync, epoch);
                    if (d != 0u && (int)d <= sweep) { fin = 1; break; }
                    Rot<T> r0;
                    if (sweep < max_sweeps && rot_fetch(log + (size_t)sweep * (N - 1) * npairs, chk + (size_t)sweep * (N - 1) * npairs, r0, key)) { fin = 0; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
                sh_rot = fin;
            }
            __syncthreads();
            const int fin = sh_rot;
            __syncthreads();
            if (fin) { lost = fin == 2; break; }
            int pr = grp % (N - 1), qr = ((N - 1) - grp % (N - 1)) % (N - 1);
            for (int r = 0; r < N - 1; ++r) {
                if (FULL || grp < npairs) {
                    int p = grp == 0 ? N - 1 : pr, q = grp == 0 ? pr : qr;
                    if (p > q) { const int t = p; p = q; q = t; }
                    pr = pr + 1 == N - 1 ? 0 : pr + 1;
                    qr = qr + 1 == N - 1 ? 0 : qr + 1;
                    Rot<T> rot{(T)1, (T)0};
                    const size_t rec = ((size_t)sweep * (N - 1) + r) * npairs + grp;
                    bool ok = false;
                    for (int it = 0; it < kSpin && !(ok = rot_fetch(log + rec, chk + rec, rot, key)); ++it) __builtin_amdgcn_s_sleep(2);
                    if (!ok) lost = true;
                    if (ok && (FULL || q < n) && rot.s != (T)0) {
                        T *vp = V + p * ld, *vq = V + q * ld;
#pragma unroll
                        for (int e = 0; e < NE; ++e) {
                            const int i = ll + LPP * e;
                            if (FULL || i < n) {
                                const T a = vp[i], b = vq[i];
                                vp[i] = rot.c * a - rot.s * b;
                                vq[i] = rot.s * a + rot.c * b;
                            }
                        }
                    }
                }
                lds_barrier();
            }
        }
        // columns go out in the sorted order the producer publishes at its very end
        for (int j = grp; j < n; j += ngrp) {
            unsigned enc = 0;
            for (int it = 0; it < kSpin && (enc = tagged_get(vsync + 1 + j, epoch)) == 0u; ++it) __builtin_amdgcn_s_sleep(8);
            if (enc == 0u) { lost = true; continue; }
            const int dst = (int)enc - 1;
            for (int i = ll; i < n; i += LPP) vc.at(i, dst) = V[j * ld + i];
        }
        if (lost && ll == 0) atomicOr(health, 8);  // the producer never showed up within the spin bound: V is incomplete
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(epoch_p, 1u, __ATOMIC_RELAXED, RC_AGENT);  // the next launch uses a new key
        return;
    }
    const T tol = sqrt((T)n) * JEps<T>::eps();
    const T tol2 = tol * tol;

    for (int e = tid; e < n * n; e += nthr) {
        int i = e % n, j = e / n;
        G[j * ld + i] = g.p[(int64_t)j * g.cs + i];
    }
    __syncthreads();

#ifdef RC_JAC_TIMING
    unsigned long long jt[6] = {0, 0, 0, 0, 0, 0}, jlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(jlast)::"memory");
#endif
    int sweep = 0;
    bool converged = false;
    for (; sweep < max_sweeps; ++sweep) {
        if (tid == 0) sh_rot = 0;
        if (CN) {
            for (int j = grp; j < n; j += ngrp) {
                const T *gj = G + j * ld;
                T acc = 0;
                for (int i = ll; i < n; i += LPP) acc = fma(gj[i], gj[i], acc);
                acc = group_sum_dpp<LPP>(acc);
                if (ll == 0) sig[j] = acc;
            }
        }
        __syncthreads();
        // circle-method pair of this group, advanced round by round when the group owns one pair slot (no integer
        // modulo on the per-round critical path): slot 0 pairs N - 1 with r, slot pi pairs (r + pi) with (r - pi) mod N - 1
        const bool one_slot = FULL || npairs <= ngrp;
        int pr = grp % (N - 1), qr = ((N - 1) - grp % (N - 1)) % (N - 1);
        for (int r = 0; r < N - 1; ++r) {
            RC_JTICK(5)  // barrier + loop control
            for (int pi = grp; pi < npairs; pi += FULL ? (1 << 20) : ngrp) {  // FULL: exactly one trip
                int p, q;
                if (one_slot) {
                    p = grp == 0 ? N - 1 : pr;
                    q = grp == 0 ? pr : qr;
                    if (p > q) { const int t = p; p = q; q = t; }
                    pr = pr + 1 == N - 1 ? 0 : pr + 1;
                    qr = qr + 1 == N - 1 ? 0 : qr + 1;
                } else {
                    rr_pair(N, r, pi, p, q);
                }
                Rot<T> rot{(T)1, (T)0};
                if (FULL || q < n) {  // p < q; q == n is the dummy column of an odd n
                    T *gp = G + p * ld, *gq = G + q * ld;
                    T a[NE], b[NE];
                    T app = 0, aqq = 0, apq = 0;
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
                        int i = ll + LPP * e;
                        if (RC_JAC_ABL & 16) { a[e] = (T)(i + p) * (T)1e-3; b[e] = (T)(i - q) * (T)1e-3; }
                        else {
                        a[e] = (FULL || i < n) ? gp[i] : (T)0;
                        b[e] = (FULL || i < n) ? gq[i] : (T)0;
                        }
                        if (!CN) { app = fma(a[e], a[e], app); aqq = fma(b[e], b[e], aqq); }
                        apq = fma(a[e], b[e], apq);
                    }
                    RC_JTICK(0)  // LDS reads + dot products
                    if (CN) { app = sig[p]; aqq = sig[q]; apq = group_sum_dpp<LPP>(apq); }
                    else if (!(RC_JAC_ABL & 8)) { app = group_sum_dpp<LPP>(app); aqq = group_sum_dpp<LPP>(aqq); apq = group_sum_dpp<LPP>(apq); }
                    RC_JTICK(1)  // group reduction
                    // rotate iff |apq| > tol * sqrt(app * aqq)   (uniform over the LPP lanes)
                    if ((RC_JAC_ABL & 32) || apq * apq > tol2 * app * aqq) {
                        if (RC_JAC_ABL & 2) { rot.c = (T)0.8 + apq * (T)1e-30; rot.s = (T)0.6 + app * (T)1e-30; }
                        else if (CN) {
                            T t;
                            jacobi_rotation(app, aqq, apq, rot.c, rot.s, &t);
                            if (ll == 0) { sig[p] = app - t * apq; sig[q] = aqq + t * apq; }
                        } else jacobi_rotation(app, aqq, apq, rot.c, rot.s);
                        RC_JTICK(2)  // rotation parameters
#pragma unroll
                        for (int e = 0; e < NE; ++e) {
                            int i = ll + LPP * e;
                            if (RC_JAC_ABL & 4) { if (rot.c * a[e] - rot.s * b[e] == (T)123.456 && rot.s * a[e] + rot.c * b[e] == (T)654.321) gp[i] = 0; }
                            else if (FULL || i < n) {
                                gp[i] = rot.c * a[e] - rot.s * b[e];
                                gq[i] = rot.s * a[e] + rot.c * b[e];
                            }
                        }
                        // flag 2 = another sweep is needed.  A rotation by the angle (c, s) of a pair whose cosine was g leaves
                        // at most |s| * (largest cosine of this sweep) behind in pairs that were already annihilated, so a sweep
                        // may be the last one only if every rotation in it had BOTH a small cosine (g <= sqrt(tol) / 4, 9e-9 in
                        // f64) AND a small angle (|s| <= 4 sqrt(tol)): for well separated singular values the second follows
                        // from the first (quadratic convergence), for clustered / repeated ones (sigma_p ~ sigma_q: the angle is
                        // O(1) however small g is) it does not, and such sweeps are followed by another one until an all-quiet
                        // or all-small sweep has been seen
                        if (ll == 0 && (apq * apq > tol * (T)0.0625 * app * aqq || rot.s * rot.s > (T)16 * tol)) sh_rot = 2;  // plain store: every writer writes 2
                    }
                }
                RC_JTICK(3)  // rotation applied, written back (LDS queue drained by the stamp)
                if (ll == 0 && !(RC_JAC_ABL & 1)) {
                    if (fused) rot_publish(log + ((size_t)sweep * (N - 1) + r) * npairs + pi, chk + ((size_t)sweep * (N - 1) + r) * npairs + pi, rot, key);
                    else log[((size_t)sweep * (N - 1) + r) * npairs + pi] = rot;
                }
            }
            RC_JTICK(4)  // record published
            lds_barrier();  // pairs of one round are disjoint; the next round re-pairs the columns
        }
        const int rotated = sh_rot;
        __syncthreads();
        if (!(RC_JAC_ABL & 32) && rotated < 2) { ++sweep; converged = true; break; }
    }
    // max_sweeps exhausted with rotations still above the thresholds: reported, never silent (health bit 4, value 16)
    if (tid == 0 && !converged && !(RC_JAC_ABL & 32) && health) atomicOr(health, 16);
#ifdef RC_JAC_TIMING
    if (tid == 0) {
        for (int k2 = 0; k2 < 6; ++k2) g_jac_dbg[k2] = jt[k2];
        g_jac_dbg[6] = (unsigned long long)sweep * (N - 1);
    }
#endif
    if (tid == 0) {
        *sweeps_out = sweep;
        if (fused) tagged_put(vsync, epoch, (unsigned)sweep);
    }

    // singular values = column norms; stable descending rank sort (gesdd order)
    for (int j = grp; j < n; j += ngrp) {
        const T *gj = G + j * ld;
        T acc = 0;
        for (int i = ll; i < n; i += LPP) acc += gj[i] * gj[i];
        acc = group_sum_dpp<LPP>(acc);
        if (ll == 0) sig[j] = sqrt(acc);
    }
    __syncthreads();
    for (int i = tid; i < n; i += nthr) {
        int rank = 0;
        const T si = sig[i];
        for (int j = 0; j < n; ++j) rank += (sig[j] > si || (sig[j] == si && j < i)) ? 1 : 0;
        order[i] = rank;
        order_out[i] = rank;
        if (fused) tagged_put(vsync + 1 + i, epoch, (unsigned)(rank + 1));
        s[rank] = si;
    }
    __syncthreads();
    for (int j = grp; j < n; j += ngrp) {
        const int dst = order[j];
        const T sj = sig[j];
        const T inv = sj > (T)0 ? (T)1 / sj : (T)0;
        const T *gj = G + j * ld;
        for (int i = ll; i < n; i += LPP) uc.at(i, dst) = gj[i] * inv;
    }
#undef sh_rot
}

// ---------------------------------------------------------------------------
// Two-column BLOCK schedule of the fused one-sided Jacobi (round 3; n = 128 = 16 lanes x 8 rows, the core of the headline
// pipeline).  k_jacobi_lds visits the 8128 column pairs of a sweep in 127 rounds of 64 disjoint pairs: every rotation costs two
// column loads, two column stores and a share of a workgroup barrier.  Here the columns form 64 blocks of two neighbours; a sweep
// is ONE round of the 64 intra-block pairs plus a 63-round tournament of the blocks (circle method), and a group of 16 lanes that
// meets block pair {X, Y} loads its FOUR columns once, rotates the four cross pairs in two phases of two INDEPENDENT rotations --
// (x0, y0) & (x1, y1), then (x0, y1) & (x1, y0) -- and stores them once: half the LDS traffic and half the barriers per
// rotation (64 instead of 127 per sweep), and two independent dependent chains per lane to fill the issue slots.  Every pair of
// columns still meets exactly once per sweep (a cyclic-by-blocks ordering); the rotation, its threshold, the two-part
// convergence test, the singular-value sort and the record protocol (J1 - J6 above; records indexed (sweep, 64 intra | round,
// slot, 0..3)) are k_jacobi_lds's.  512 threads = 32 groups = one per block pair.  The consumer workgroup mirrors the schedule on V.
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void jb2_rotate(T (&a)[8], T (&b)[8], T tol, T tol2, Rot<T> &rot, int ll, int *sh_rot_p) {
    T app = 0, aqq = 0, apq = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { app = fma(a[e], a[e], app); aqq = fma(b[e], b[e], aqq); apq = fma(a[e], b[e], apq); }
    app = group_sum_dpp<16>(app);
    aqq = group_sum_dpp<16>(aqq);
    apq = group_sum_dpp<16>(apq);
    rot.c = (T)1;
    rot.s = (T)0;
    if (apq * apq > tol2 * app * aqq) {  // uniform over the 16 lanes
        jacobi_rotation(app, aqq, apq, rot.c, rot.s);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const T x = a[e], y = b[e];
            a[e] = rot.c * x - rot.s * y;
            b[e] = rot.s * x + rot.c * y;
        }
        if (ll == 0 && (apq * apq > tol * (T)0.0625 * app * aqq || rot.s * rot.s > (T)16 * tol)) *sh_rot_p = 2;  // another sweep is needed (k_jacobi_lds)
    }
}
template <typename T>
__device__ __forceinline__ void jb2_apply(T (&a)[8], T (&b)[8], Rot<T> rot) {
    if (rot.s != (T)0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const T x = a[e], y = b[e];
            a[e] = rot.c * x - rot.s * y;
            b[e] = rot.s * x + rot.c * y;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(512) void k_jacobi_b2(Mat<T> g, Rot<T> *log, int *sweeps_out, Mat<T> uc, T *s, int *order_out, int max_sweeps, unsigned *vsync,
                                                    unsigned long long *chk, unsigned *epoch_p, Mat<T> vc, int *health, int ld) {
    constexpr int n = 128, NB = 64, NR = NB - 1, NSLOT = NB / 2, LPP = 16, NE = 8;
    constexpr int REC = NB + NR * NSLOT * 4;  // records per sweep = 8128 = (n - 1) * n / 2: the log of k_jacobi_lds has the same size
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *G = reinterpret_cast<T *>(smem_raw);
    T *sig = G + (size_t)ld * n;
    int *order = reinterpret_cast<int *>(sig + n);
    __shared__ int sh_rot_b2;
    const int tid = threadIdx.x, nthr = 512;
    const int ll = tid % LPP, grp = tid / LPP, ngrp = nthr / LPP;  // 32 groups
    // The two groups of a 32-lane half would both read an even column, then both an odd one: same offset in the bank row (the pitch is
    // 16 mod 32 elements), a two-way conflict on every access.  Odd groups take the columns of a block in the other order, so one
    // instruction reads an even and an odd column per half: conflict-free.  The pairs of a phase are the same sets either way --
    // (x0, y0) & (x1, y1), then (x0, y1) & (x1, y0) -- only their order inside a phase differs, and the consumer uses the same rule.
    const int sw = grp & 1;
    const unsigned epoch = __hip_atomic_load(epoch_p, __ATOMIC_RELAXED, RC_AGENT) & 0xffffffu;
    const unsigned long long key = epoch_key(epoch);
    // the block pair of this group in tournament round rr: slot 0 pairs block NB - 1 with rr, slot i pairs (rr + i) with (rr - i) mod NR
    auto blocks_of = [&](int pr, int qr, int &X, int &Y) {
        X = grp == 0 ? NB - 1 : pr;
        Y = grp == 0 ? pr : qr;
        if (X > Y) { const int t = X; X = Y; Y = t; }  // X < Y: every cross pair (x_i, y_j) has x_i < y_j, the (p < q) convention of the rotation
    };
    if (blockIdx.x == 1) {
        // ---- consumer: V = product of the rotations, the producer's schedule ----
        constexpr int kSpin = 1 << 24;
        T *V = G;
        for (int e = tid; e < n * n; e += nthr) {
            const int i = e % n, j = e / n;
            V[j * ld + i] = (i == j) ? (T)1 : (T)0;
        }
        __syncthreads();
        bool lost = false;
        auto fetch = [&](size_t rec) -> Rot<T> {
            Rot<T> rot{(T)1, (T)0};
            bool ok = false;
            for (int it = 0; it < kSpin && !(ok = rot_fetch(log + rec, chk + rec, rot, key)); ++it) __builtin_amdgcn_s_sleep(2);
            if (!ok) { lost = true; rot.c = (T)1; rot.s = (T)0; }
            return rot;
        };
        for (int sweep = 0;; ++sweep) {
            if (tid == 0) {  // has the producer started this sweep, or did it finish before it?
                int fin = 2;
                for (int it = 0; it < kSpin; ++it) {
                    const unsigned d = tagged_get(vsync, epoch);
                    if (d != 0u && (int)d <= sweep) { fin = 1; break; }
                    Rot<T> r0;
                    if (sweep < max_sweeps && rot_fetch(log + (size_t)sweep * REC, chk + (size_t)sweep * REC, r0, key)) { fin = 0; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
                sh_rot_b2 = fin;
            }
            __syncthreads();
            const int fin = sh_rot_b2;
            __syncthreads();
            if (fin) { lost = lost || fin == 2; break; }
            const size_t base = (size_t)sweep * REC;
            for (int h = 0; h < 2; ++h) {  // intra-block pairs
                const int b = grp + ngrp * h;
                const Rot<T> rot = fetch(base + b);
                if (rot.s != (T)0) {
                    T *vp = V + (2 * b) * ld, *vq = V + (2 * b + 1) * ld;
                    T a[NE], c2[NE];
#pragma unroll
                    for (int e = 0; e < NE; ++e) { a[e] = vp[ll + LPP * e]; c2[e] = vq[ll + LPP * e]; }
                    jb2_apply(a, c2, rot);
#pragma unroll
                    for (int e = 0; e < NE; ++e) { vp[ll + LPP * e] = a[e]; vq[ll + LPP * e] = c2[e]; }
                }
            }
            lds_barrier();
            int pr = grp % NR, qr = (NR - grp % NR) % NR;
            for (int rr = 0; rr < NR; ++rr) {
                int X, Y;
                blocks_of(pr, qr, X, Y);
                pr = pr + 1 == NR ? 0 : pr + 1;
                qr = qr + 1 == NR ? 0 : qr + 1;
                const size_t rec = base + NB + ((size_t)rr * NSLOT + grp) * 4;
                const Rot<T> r0 = fetch(rec), r1 = fetch(rec + 1), r2 = fetch(rec + 2), r3 = fetch(rec + 3);
                if (r0.s != (T)0 || r1.s != (T)0 || r2.s != (T)0 || r3.s != (T)0) {
                    T *v0 = V + (2 * X + sw) * ld, *v1 = V + (2 * X + 1 - sw) * ld, *v2 = V + (2 * Y + sw) * ld, *v3 = V + (2 * Y + 1 - sw) * ld;
                    T c0[NE], c1[NE], c2[NE], c3[NE];
#pragma unroll
                    for (int e = 0; e < NE; ++e) { const int i = ll + LPP * e; c0[e] = v0[i]; c1[e] = v1[i]; c2[e] = v2[i]; c3[e] = v3[i]; }
                    jb2_apply(c0, c2, r0);
                    jb2_apply(c1, c3, r1);
                    jb2_apply(c0, c3, r2);
                    jb2_apply(c1, c2, r3);
#pragma unroll
                    for (int e = 0; e < NE; ++e) { const int i = ll + LPP * e; v0[i] = c0[e]; v1[i] = c1[e]; v2[i] = c2[e]; v3[i] = c3[e]; }
                }
                lds_barrier();
            }
        }
        // columns go out in the sorted order the producer publishes at its very end
        for (int j = grp; j < n; j += ngrp) {
            unsigned enc = 0;
            for (int it = 0; it < kSpin && (enc = tagged_get(vsync + 1 + j, epoch)) == 0u; ++it) __builtin_amdgcn_s_sleep(8);
            if (enc == 0u) { lost = true; continue; }
            const int dst = (int)enc - 1;
            for (int i = ll; i < n; i += LPP) vc.at(i, dst) = V[j * ld + i];
        }
        if (lost && ll == 0) atomicOr(health, 8);  // the producer never showed up within the spin bound: V is incomplete
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(epoch_p, 1u, __ATOMIC_RELAXED, RC_AGENT);  // the next launch uses a new key
        return;
    }
    // ---- producer ----
    const T tol = sqrt((T)n) * JEps<T>::eps();
    const T tol2 = tol * tol;
    for (int e = tid; e < n * n; e += nthr) {
        const int i = e % n, j = e / n;
        G[j * ld + i] = g.p[(int64_t)j * g.cs + i];
    }
    __syncthreads();
    int sweep = 0;
    bool converged = false;
    for (; sweep < max_sweeps; ++sweep) {
        if (tid == 0) sh_rot_b2 = 0;
        __syncthreads();
        const size_t base = (size_t)sweep * REC;
        for (int h = 0; h < 2; ++h) {  // the 64 intra-block pairs (2 b, 2 b + 1)
            const int b = grp + ngrp * h;
            T *gp = G + (2 * b) * ld, *gq = G + (2 * b + 1) * ld;
            T a[NE], c2[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) { a[e] = gp[ll + LPP * e]; c2[e] = gq[ll + LPP * e]; }
            Rot<T> rot;
            jb2_rotate(a, c2, tol, tol2, rot, ll, &sh_rot_b2);
            if (rot.s != (T)0) {
#pragma unroll
                for (int e = 0; e < NE; ++e) { gp[ll + LPP * e] = a[e]; gq[ll + LPP * e] = c2[e]; }
            }
            if (ll == 0) rot_publish(log + base + b, chk + base + b, rot, key);
        }
        lds_barrier();
        int pr = grp % NR, qr = (NR - grp % NR) % NR;
        for (int rr = 0; rr < NR; ++rr) {
            int X, Y;
            blocks_of(pr, qr, X, Y);
            pr = pr + 1 == NR ? 0 : pr + 1;
            qr = qr + 1 == NR ? 0 : qr + 1;
            T *g0 = G + (2 * X + sw) * ld, *g1 = G + (2 * X + 1 - sw) * ld, *g2 = G + (2 * Y + sw) * ld, *g3 = G + (2 * Y + 1 - sw) * ld;
            T c0[NE], c1[NE], c2[NE], c3[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) { const int i = ll + LPP * e; c0[e] = g0[i]; c1[e] = g1[i]; c2[e] = g2[i]; c3[e] = g3[i]; }
            Rot<T> r0, r1, r2, r3;
            jb2_rotate(c0, c2, tol, tol2, r0, ll, &sh_rot_b2);   // phase 1: (x0, y0) and (x1, y1) are independent
            jb2_rotate(c1, c3, tol, tol2, r1, ll, &sh_rot_b2);
            jb2_rotate(c0, c3, tol, tol2, r2, ll, &sh_rot_b2);   // phase 2: (x0, y1) and (x1, y0)
            jb2_rotate(c1, c2, tol, tol2, r3, ll, &sh_rot_b2);
            if (r0.s != (T)0 || r1.s != (T)0 || r2.s != (T)0 || r3.s != (T)0) {
#pragma unroll
                for (int e = 0; e < NE; ++e) { const int i = ll + LPP * e; g0[i] = c0[e]; g1[i] = c1[e]; g2[i] = c2[e]; g3[i] = c3[e]; }
            }
            if (ll == 0) {
                const size_t rec = base + NB + ((size_t)rr * NSLOT + grp) * 4;
                rot_publish(log + rec, chk + rec, r0, key);
                rot_publish(log + rec + 1, chk + rec + 1, r1, key);
                rot_publish(log + rec + 2, chk + rec + 2, r2, key);
                rot_publish(log + rec + 3, chk + rec + 3, r3, key);
            }
            lds_barrier();  // block pairs of one round are disjoint; the next round re-pairs the blocks
        }
        const int rotated = sh_rot_b2;
        __syncthreads();
        if (rotated < 2) { ++sweep; converged = true; break; }
    }
    if (tid == 0 && !converged && health) atomicOr(health, 16);  // sweep budget exhausted with rotations above the thresholds
    if (tid == 0) {
        *sweeps_out = sweep;
        tagged_put(vsync, epoch, (unsigned)sweep);
    }
    // singular values = column norms; stable descending rank sort (gesdd order) -- as k_jacobi_lds
    for (int j = grp; j < n; j += ngrp) {
        const T *gj = G + j * ld;
        T acc = 0;
        for (int i = ll; i < n; i += LPP) acc += gj[i] * gj[i];
        acc = group_sum_dpp<LPP>(acc);
        if (ll == 0) sig[j] = sqrt(acc);
    }
    __syncthreads();
    for (int i = tid; i < n; i += nthr) {
        int rank = 0;
        const T si = sig[i];
        for (int j = 0; j < n; ++j) rank += (sig[j] > si || (sig[j] == si && j < i)) ? 1 : 0;
        order[i] = rank;
        order_out[i] = rank;
        tagged_put(vsync + 1 + i, epoch, (unsigned)(rank + 1));
        s[rank] = si;
    }
    __syncthreads();
    for (int j = grp; j < n; j += ngrp) {
        const int dst = order[j];
        const T sj = sig[j];
        const T inv = sj > (T)0 ? (T)1 / sj : (T)0;
        const T *gj = G + j * ld;
        for (int i = ll; i < n; i += LPP) uc.at(i, dst) = gj[i] * inv;
    }
}

// ---------------------------------------------------------------------------
// V = J_1 J_2 ... applied to I, row by row: one wave per row of V.
// ---------------------------------------------------------------------------
template <typename T, int RPW>
__global__ __launch_bounds__(256) void k_jacobi_replay_v(int n, const Rot<T> *log, const int *sweeps, const int *order, Mat<T> vc) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *rows = reinterpret_cast<T *>(smem_raw);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int N = (n + 1) & ~1, npairs = N / 2;
    const int ns = *sweeps;
    if (npairs <= 64) {
        // One pair per lane.  A wave carries RPW independent rows through the rounds: the chain of a row is a
        // dependent LDS read -> FMA -> LDS write per round, so the rows' chains interleave and a quarter of the
        // waves (and CUs) does the same work in the same time.  The log is one 64-entry line per round, fetched
        // PF rounds ahead so that the chain does not pay an L2 latency per round.
        constexpr int PF = 8;
        const int row0 = (blockIdx.x * 4 + wv) * RPW;
        if (row0 >= n) return;  // whole wave; no barriers in this kernel
        T *v = rows + (size_t)wv * RPW * (n + 1);
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
            for (int j = lane; j < n; j += 64) v[rr * (n + 1) + j] = (j == row0 + rr) ? (T)1 : (T)0;
        const int total = ns * (N - 1);
        const bool has = lane < npairs;
        // circle-method pair of this lane, advanced round by round (no integer modulo in the chain):
        // lane 0 pairs N - 1 with r; lane pi pairs (r + pi) mod (N - 1) with (r - pi) mod (N - 1)
        int pr = lane % (N - 1), qr = ((N - 1) - lane % (N - 1)) % (N - 1);
        for (int base = 0; base < total; base += PF) {
            Rot<T> rt[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                rt[u].c = (T)1;
                rt[u].s = (T)0;
                if (has && base + u < total) rt[u] = log[(size_t)(base + u) * npairs + lane];
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (base + u < total) {
                    int p = lane == 0 ? N - 1 : pr, q = lane == 0 ? pr : qr;  // lane 0: pr runs through r itself
                    if (p > q) { const int t = p; p = q; q = t; }
                    pr = pr + 1 == N - 1 ? 0 : pr + 1;
                    qr = qr + 1 == N - 1 ? 0 : qr + 1;
                    if (rt[u].s != (T)0) {
                        T a[RPW], b[RPW];
#pragma unroll
                        for (int rr = 0; rr < RPW; ++rr) { a[rr] = v[rr * (n + 1) + p]; b[rr] = v[rr * (n + 1) + q]; }
#pragma unroll
                        for (int rr = 0; rr < RPW; ++rr) {
                            v[rr * (n + 1) + p] = rt[u].c * a[rr] - rt[u].s * b[rr];
                            v[rr * (n + 1) + q] = rt[u].s * a[rr] + rt[u].c * b[rr];
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
            if (row0 + rr < n)
                for (int j = lane; j < n; j += 64) vc.at(row0 + rr, order[j]) = v[rr * (n + 1) + j];
        return;
    }
    const int row = blockIdx.x * 4 + wv;
    if (row >= n) return;  // whole wave; no barriers in this kernel
    T *v = rows + (size_t)wv * (n + 1);
    for (int j = lane; j < n; j += 64) v[j] = (j == row) ? (T)1 : (T)0;
    for (int sw = 0; sw < ns; ++sw)
        for (int r = 0; r < N - 1; ++r) {
            const Rot<T> *lr = log + ((size_t)sw * (N - 1) + r) * npairs;
            for (int pi = lane; pi < npairs; pi += 64) {
                const Rot<T> rot = lr[pi];
                if (rot.s != (T)0) {
                    int p, q;
                    rr_pair(N, r, pi, p, q);
                    T a = v[p], b = v[q];
                    v[p] = rot.c * a - rot.s * b;
                    v[q] = rot.s * a + rot.c * b;
                }
            }
            // a wave executes its LDS operations in order: the next round (other pairing of the
            // same row) sees these writes without a barrier
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    for (int j = lane; j < n; j += 64) vc.at(row, order[j]) = v[j];
}

// ---------------------------------------------------------------------------
// Cores that do not fit the LDS of one CU: the same one-sided Jacobi with G and V in (L2-resident)
// global memory, one LAUNCH per round of the circle-method schedule -- the N/2 pairs of a round are
// disjoint, so one wave per pair needs no synchronisation inside the round and the whole chip works
// on it.  state[0] = "some pair rotated in this sweep", state[1] = converged, state[2] = sweeps done.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_jacobi_round(Mat<T> g, Mat<T> v, int r, int *state) {
    if (state[1]) return;
    const int n = (int)g.rows, N = (n + 1) & ~1;
    const int lane = threadIdx.x & 63;
    const int pi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pi >= N / 2) return;
    int p, q;
    rr_pair(N, r, pi, p, q);
    if (q >= n) return;  // the dummy column of an odd n
    const T tol = sqrt((T)n) * JEps<T>::eps();
    T *gp = g.p + (int64_t)p * g.cs, *gq = g.p + (int64_t)q * g.cs;
    T app = 0, aqq = 0, apq = 0;
    for (int i = lane; i < n; i += 64) {
        const T a = gp[i], b = gq[i];
        app = fma(a, a, app);
        aqq = fma(b, b, aqq);
        apq = fma(a, b, apq);
    }
    app = wave_sum_dpp(app);
    aqq = wave_sum_dpp(aqq);
    apq = wave_sum_dpp(apq);
    if (apq == (T)0 || fabs(apq) <= tol * sqrt(app) * sqrt(aqq)) return;
    T cs, sn;
    jacobi_rotation(app, aqq, apq, cs, sn);
    for (int i = lane; i < n; i += 64) {
        const T a = gp[i], b = gq[i];
        gp[i] = cs * a - sn * b;
        gq[i] = sn * a + cs * b;
    }
    T *vp = v.p + (int64_t)p * v.cs, *vq = v.p + (int64_t)q * v.cs;
    for (int i = lane; i < n; i += 64) {
        const T a = vp[i], b = vq[i];
        vp[i] = cs * a - sn * b;
        vq[i] = sn * a + cs * b;
    }
    if (lane == 0) state[0] = 1;
}
__global__ void k_jacobi_check(const int *state, int *health) {
    if (threadIdx.x == 0 && !state[1]) atomicOr(health, 16);  // sweep budget exhausted before an all-quiet sweep
}
__global__ void k_jacobi_sweep_end(int *state) {
    if (threadIdx.x != 0 || state[1]) return;
    state[2] += 1;
    if (state[0] == 0) state[1] = 1;
    state[0] = 0;
}
// singular values = column norms of the rotated G; sig/order live in global memory
template <typename T>
__global__ __launch_bounds__(256) void k_jacobi_norms(Mat<T> g, T *sig) {
    const int n = (int)g.rows, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    const T *gj = g.p + (int64_t)j * g.cs;
    T acc = 0;
    for (int i = lane; i < n; i += 64) { const T a = gj[i]; acc = fma(a, a, acc); }
    acc = wave_sum_dpp(acc);
    if (lane == 0) sig[j] = sqrt(acc);
}
template <typename T>
__global__ __launch_bounds__(256) void k_jacobi_rank(int n, const T *sig, int *order, T *s) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const T si = sig[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
        const T sj = sig[j];
        rank += (sj > si || (sj == si && j < i)) ? 1 : 0;
    }
    order[i] = rank;
    s[rank] = si;
}
template <typename T>
__global__ __launch_bounds__(256) void k_jacobi_emit(Mat<T> g, Mat<T> v, const T *sig, const int *order, Mat<T> uc, Mat<T> vc) {
    const int n = (int)g.rows, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    const int dst = order[j];
    const T sj = sig[j];
    const T inv = sj > (T)0 ? (T)1 / sj : (T)0;
    const T *gj = g.p + (int64_t)j * g.cs, *vj = v.p + (int64_t)j * v.cs;
    for (int i = lane; i < n; i += 64) {
        uc.at(i, dst) = gj[i] * inv;
        vc.at(i, dst) = vj[i];
    }
}

template <typename T>
static void jacobi_global(rc_context *c, Mat<T> g, Mat<T> v, Mat<T> uc, T *s, Mat<T> vc) {
    const int n = (int)g.rows, N = (n + 1) & ~1;
    ArenaMark mark(c);
    int *state = c->alloc<int>(4);
    T *sig = c->alloc<T>((size_t)n);
    int *order = c->alloc<int>((size_t)n);
    fill_words(c, state, 4 * sizeof(int), 0u);
    fill_identity(c, v);
    // outside a graph capture the convergence flag is read back after every sweep; inside one a fixed number
    // of sweeps is recorded (converged sweeps return immediately)
    const int max_sweeps = c->capturing ? 16 : 60;
    const unsigned grid = (unsigned)((N / 2 + 3) / 4);
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        for (int r = 0; r < N - 1; ++r) hipLaunchKernelGGL(k_jacobi_round<T>, dim3(grid), dim3(256), 0, c->stream, g, v, r, state);
        hipLaunchKernelGGL(k_jacobi_sweep_end, dim3(1), dim3(64), 0, c->stream, state);
        if (!c->capturing) {
            int h[4] = {0, 0, 0, 0};
            RC_HIP(hipMemcpyAsync(h, state, sizeof(h), hipMemcpyDeviceToHost, c->stream));
            RC_HIP(hipStreamSynchronize(c->stream));
            if (h[1]) break;
        }
    }
    hipLaunchKernelGGL(k_jacobi_check, dim3(1), dim3(64), 0, c->stream, state, c->health_word());
    hipLaunchKernelGGL(k_jacobi_norms<T>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, g, sig);
    hipLaunchKernelGGL(k_jacobi_rank<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, sig, order, s);
    hipLaunchKernelGGL(k_jacobi_emit<T>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, g, v, sig, order, uc, vc);
}

template <typename T, int LPP, int NE, bool FULL, bool CN>
static void launch_lds_impl(rc_context *c, Mat<T> g, Mat<T> uc, T *s, Mat<T> vc, size_t lds, int ld, int max_sweeps) {
    const int n = (int)g.rows, N = (n + 1) & ~1;
    ArenaMark mark(c);
    Rot<T> *log = c->alloc<Rot<T>>((size_t)max_sweeps * (N - 1) * (N / 2));
    int *sweeps = c->alloc<int>(1);
    int *order = c->alloc<int>((size_t)n);
    auto kern = k_jacobi_lds<T, LPP, NE, FULL, CN>;
    static bool attr_set[64] = {};
    if (!attr_set[c->device & 63]) {
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        attr_set[c->device & 63] = true;
    }
    // one LPP-lane group per pair, rounded up to whole waves
    const int threads = std::min(LPP == 16 ? 1024 : 512, std::max(64, (((N / 2) * LPP + 63) / 64) * 64));
    // fused right vectors: a second workgroup applies the rotations to V as they are published (no replay kernel
    // after the fact); needs one pair slot per group and the padded n of the register tiling
    static const int fuse_env = [] { const char *e = getenv("RC_JACOBI_FUSED_V"); return e ? atoi(e) : 1; }();
    const bool fused = fuse_env && (LPP == 16 || LPP == 8) && (N / 2) * LPP <= threads && n <= LPP * NE && n < 255 && max_sweeps < 255;
    if (fused) {
        unsigned *vsync = c->alloc<unsigned>((size_t)n + 1);
        unsigned long long *chk = c->alloc<unsigned long long>((size_t)max_sweeps * (N - 1) * (N / 2));
        // The tagged words are cleared before every launch: a word counts as "published" when its upper 24 bits equal the launch's
        // epoch, and workspace memory that is new to the context (first call, arena growth) holds whatever its last owner left --
        // on a context's first launch (epoch 0 until round 2) any small integer there passed for a sweep count or a sorted
        // position, and the consumer wrote V's columns to the wrong places without noticing (seen as one wrong `vt` among 16
        // contexts' first calls).  The epoch itself now starts at a per-context pseudo-random value (rc_context::epoch_word).
        static const int no_clear = [] { const char *e = getenv("RC_DEBUG_JACOBI_NO_CLEAR"); return e ? atoi(e) : 0; }();  // (diagnostic: the round-2 behaviour)
        if (!no_clear) hipLaunchKernelGGL(k_clear_words, dim3(1), dim3(256), 0, c->stream, vsync, n + 1);
        // two-column block schedule (k_jacobi_b2) for the 128-column core: opt-in (RC_JACOBI_BLOCK2=1).  Measured in round 3: the same
        // 9 sweeps, 1.80 ms against 1.77 ms for the pair-per-round schedule, headline unchanged -- the kernel is bound by the DEPENDENT
        // chain of a rotation (dot products -> DPP reduction -> rsq / rcp + Newton -> update -> LDS), not by LDS traffic or barriers: a
        // block round has two dependent phases, so half the rounds carry the same chain length (SVD)
        static const int block2 = [] { const char *e = getenv("RC_JACOBI_BLOCK2"); return e ? atoi(e) : 0; }();
        if (block2 && FULL && !CN && LPP == 16 && NE == 8 && n == 128) {
            auto kb = k_jacobi_b2<T>;
            static bool attr_b2[64] = {};
            if (!attr_b2[c->device & 63]) {
                RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
                attr_b2[c->device & 63] = true;
            }
            hipLaunchKernelGGL(kb, dim3(2), dim3(512), lds, c->stream, g, log, sweeps, uc, s, order, max_sweeps, vsync, chk, c->epoch_word(), vc, c->health_word(), ld);
        } else
        hipLaunchKernelGGL(kern, dim3(2), dim3(threads), lds, c->stream, g, log, sweeps, uc, s, order, max_sweeps, 1, vsync, chk, c->epoch_word(), vc, c->health_word(), ld);
    } else {
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), lds, c->stream, g, log, sweeps, uc, s, order, max_sweeps, 0, (unsigned *)nullptr, (unsigned long long *)nullptr,
                           (unsigned *)nullptr, vc, c->health_word(), ld);
    }
    static const int rpw_env = [] { const char *e = getenv("RC_REPLAY_RPW"); return e ? atoi(e) : 1; }();
    const int rpw = (N / 2 <= 64) ? (rpw_env == 2 || rpw_env == 4 ? rpw_env : 1) : 1;  // rows per wave: 1 measured best (913 vs 903 compressions/s at 4)
    const size_t lds_v = 4 * (size_t)rpw * (n + 1) * sizeof(T);
    const dim3 grid((unsigned)((n + 4 * rpw - 1) / (4 * rpw)));
    if (fused) { /* V is already complete */ }
    else if (rpw == 4) hipLaunchKernelGGL((k_jacobi_replay_v<T, 4>), grid, dim3(256), lds_v, c->stream, n, log, sweeps, order, vc);
    else if (rpw == 2) hipLaunchKernelGGL((k_jacobi_replay_v<T, 2>), grid, dim3(256), lds_v, c->stream, n, log, sweeps, order, vc);
    else hipLaunchKernelGGL((k_jacobi_replay_v<T, 1>), grid, dim3(256), lds_v, c->stream, n, log, sweeps, order, vc);
    if (c->prof_on && !c->capturing) {  // diagnostic: number of sweeps, reported through the profile table
        int h = 0;
        (void)hipMemcpyAsync(&h, sweeps, sizeof(int), hipMemcpyDeviceToHost, c->stream);  // on the context's own stream
        (void)hipStreamSynchronize(c->stream);
        char nm[64];
        snprintf(nm, sizeof(nm), "info:jacobi_sweeps n=%d", n);
        auto &a = c->prof_acc[nm];
        a.ms += h;
        a.calls += 1;
    }
}

template <typename T, int LPP, int NE>
static void launch_lds(rc_context *c, Mat<T> g, Mat<T> uc, T *s, Mat<T> vc, size_t lds, int ld, int max_sweeps) {
    // the bound-free instance: every lane row and every pair slot is real (n = 32, 64, 128 with 16 lanes per pair)
    static const int full_env = [] { const char *e = getenv("RC_JACOBI_FULL"); return e ? atoi(e) : 1; }();
    const int n = (int)g.rows;
    static const int cn_env = [] { const char *e = getenv("RC_JACOBI_CACHED_NORMS"); return e ? atoi(e) : 0; }();
    const bool full = full_env && (LPP == 16 || LPP == 8) && n == LPP * NE && (n / 2) * LPP <= (LPP == 16 ? 1024 : 512);
    if (full && cn_env && LPP == 16) launch_lds_impl<T, LPP, NE, true, true>(c, g, uc, s, vc, lds, ld, max_sweeps);
    else if (full) launch_lds_impl<T, LPP, NE, true, false>(c, g, uc, s, vc, lds, ld, max_sweeps);
    else launch_lds_impl<T, LPP, NE, false, false>(c, g, uc, s, vc, lds, ld, max_sweeps);
}

template <typename T, int LPP>
static void launch_lds_lpp(rc_context *c, Mat<T> g, Mat<T> uc, T *s, Mat<T> vc, size_t lds, int ld, int max_sweeps) {
    const int n = (int)g.rows;
    constexpr int U = 32 / LPP;  // rows per lane at n = 32
    if (n <= 32) launch_lds<T, LPP, U>(c, g, uc, s, vc, lds, ld, max_sweeps);
    else if (n <= 64) launch_lds<T, LPP, 2 * U>(c, g, uc, s, vc, lds, ld, max_sweeps);
    else if (n <= 128) launch_lds<T, LPP, 4 * U>(c, g, uc, s, vc, lds, ld, max_sweeps);
    else launch_lds<T, LPP, 6 * U>(c, g, uc, s, vc, lds, ld, max_sweeps);  // f32 up to n = 192 (the LDS bound is ~200)
}

// ?gesdd returns an orthonormal U also for a rank-deficient matrix; the one-sided Jacobi iteration leaves a ZERO left vector for
// every zero singular value.  This pass completes them (LAPACK's convention up to the choice of the basis of the null space, which
// no caller can observe through U S V^T): for each zero column, the unit vector with the largest component outside the span of the
// columns before it, orthogonalised twice against them (classical Gram-Schmidt with one re-orthogonalisation) and normalised.
// One workgroup; returns at once when the smallest singular value is positive (the values are sorted), which is every call of
// the hot path.  uc: n x n column-major.
template <typename T>
__global__ __launch_bounds__(1024) void k_complete_left_basis(Mat<T> uc, const T *s) {
    extern __shared__ __attribute__((aligned(16))) char cb_raw[];
    T *v = reinterpret_cast<T *>(cb_raw);          // n
    T *d = v + uc.rows;                            // n: projections
    __shared__ T red_v[16];
    __shared__ int red_i[16];
    __shared__ int sh_p;
    __shared__ T sh_nrm;
    const int n = (int)uc.rows, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (!(s[n - 1] == (T)0)) return;
    int z = 0;  // number of positive singular values (sorted descending): columns z .. n-1 are the zero ones
    for (int j = 0; j < n; ++j) z += s[j] > (T)0 ? 1 : 0;
    for (int j = z; j < n; ++j) {
        // residual of every unit vector: 1 - sum_c U(p, c)^2 over the columns before j
        T best = (T)-1;
        int bp = 0;
        for (int p = tid; p < n; p += 1024) {
            T acc = 1;
            for (int c2 = 0; c2 < j; ++c2) { const T x = uc.p[p + (int64_t)c2 * uc.cs]; acc = fma(-x, x, acc); }
            if (acc > best) { best = acc; bp = p; }
        }
        const T mx = wave_max_dpp(best);
        const int cand = wave_min_dpp(best == mx ? bp : 0x7fffffff);
        if (lane == 0) { red_v[wv] = mx; red_i[wv] = cand; }
        __syncthreads();
        if (tid == 0) {
            T b = red_v[0];
            int bi = red_i[0];
            for (int w = 1; w < 16; ++w)
                if (red_v[w] > b || (red_v[w] == b && red_i[w] < bi)) { b = red_v[w]; bi = red_i[w]; }
            sh_p = bi;
        }
        __syncthreads();
        const int p = sh_p;
        for (int i = tid; i < n; i += 1024) v[i] = i == p ? (T)1 : (T)0;
        __syncthreads();
        for (int pass = 0; pass < 2; ++pass) {
            for (int c2 = tid; c2 < j; c2 += 1024) {
                T acc = 0;
                for (int i = 0; i < n; ++i) acc = fma(uc.p[i + (int64_t)c2 * uc.cs], v[i], acc);
                d[c2] = acc;
            }
            __syncthreads();
            for (int i = tid; i < n; i += 1024) {
                T acc = v[i];
                for (int c2 = 0; c2 < j; ++c2) acc = fma(-uc.p[i + (int64_t)c2 * uc.cs], d[c2], acc);
                v[i] = acc;
            }
            __syncthreads();
        }
        T part = 0;
        for (int i = tid; i < n; i += 1024) part = fma(v[i], v[i], part);
        part = wave_sum_dpp(part);
        if (lane == 0) red_v[wv] = part;
        __syncthreads();
        if (tid == 0) {
            T t = 0;
            for (int w = 0; w < 16; ++w) t += red_v[w];
            sh_nrm = sqrt(t);
        }
        __syncthreads();
        const T inv = (T)1 / sh_nrm;
        for (int i = tid; i < n; i += 1024) uc.p[i + (int64_t)j * uc.cs] = v[i] * inv;
        __threadfence();
        __syncthreads();
    }
}

template <typename T>
void complete_left_basis(rc_context *c, Mat<T> uc, const T *s) {
    static const bool on = [] { const char *e = getenv("RC_SVD_COMPLETE_BASIS"); return !e || atoi(e) != 0; }();  // experiments
    if (!on || uc.rows == 0 || uc.rows != uc.cols || uc.rs != 1) return;
    const size_t lds = 2 * (size_t)uc.rows * sizeof(T);
    if (lds > 64 * 1024) return;  // (cores beyond 4096 x 4096 f64 keep the zero vectors)
    hipLaunchKernelGGL(k_complete_left_basis<T>, dim3(1), dim3(1024), lds, c->stream, uc, s);
}
template void complete_left_basis<double>(rc_context *, Mat<double>, const double *);
template void complete_left_basis<float>(rc_context *, Mat<float>, const float *);

template <typename T>
void jacobi_svd(rc_context *c, Mat<T> g, Mat<T> vwork, Mat<T> uc, T *s, Mat<T> vc) {
    RC_REQUIRE(g.rows == g.cols && g.rs == 1 && vwork.rs == 1, RC_LAYOUT_ERROR, "jacobi_svd: square column-major core required");
    const int n = (int)g.rows;
    if (n == 0) return;
    ProfScope ps(c, "op:jacobi_svd n=%lld", (long long)g.rows);
    static const int max_sweeps_env = [] { const char *e = getenv("RC_JACOBI_MAX_SWEEPS"); return e ? atoi(e) : 30; }();  // experiments only
    const int max_sweeps = max_sweeps_env;
    // Column pitch in LDS.  A 32-lane half of a wave holds the groups of two neighbouring pair slots, whose columns are
    // neighbours too (p, p + 1 and q, q - 1): with a pitch of 16 elements modulo 32 the two 16-lane groups read opposite
    // halves of the bank row (ds_read_b64: 64 banks, f32 ds_read_b32: 32 banks) -- conflict-free, where the odd pitch n | 1
    // made every such read two-way conflicted.  The padded pitch is used whenever it fits the CU's LDS.
    static const int pitch_env = [] { const char *e = getenv("RC_JACOBI_PITCH"); return e ? atoi(e) : 1; }();
    const size_t lds_cap = 160 * 1024 - 2048 - 64;
    auto lds_bytes = [&](int pitch) { return ((size_t)pitch * n + n) * sizeof(T) + (size_t)n * sizeof(int) + 64; };
    static const int lpp = [] { const char *e = getenv("RC_JACOBI_LPP"); return e ? atoi(e) : 16; }();
    int ld = lpp == 8 ? ((n + 23) / 32) * 32 + 8 : ((n + 15) / 32) * 32 + 16;  // 8 lanes per pair: four groups per half, a quarter row apart
    if (!pitch_env || lds_bytes(ld) > lds_cap) ld = n | 1;
    const size_t lds = lds_bytes(ld);
    if (lds <= lds_cap && n <= 192) {
        if (lpp == 4) launch_lds_lpp<T, 4>(c, g, uc, s, vc, lds, ld, max_sweeps);
        else if (lpp == 8) launch_lds_lpp<T, 8>(c, g, uc, s, vc, lds, ld, max_sweeps);
        else launch_lds_lpp<T, 16>(c, g, uc, s, vc, lds, ld, max_sweeps);
    } else {
        jacobi_global<T>(c, g, vwork, uc, s, vc);
    }
}

template void jacobi_svd<double>(rc_context *, Mat<double>, Mat<double>, Mat<double>, double *, Mat<double>);
template void jacobi_svd<float>(rc_context *, Mat<float>, Mat<float>, Mat<float>, float *, Mat<float>);

}  // namespace rc

#ifdef RC_JAC_TIMING
extern "C" void rc_debug_jacobi_timing(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(rc::g_jac_dbg), 8 * sizeof(unsigned long long)); }
#endif
