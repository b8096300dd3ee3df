// Register-resident cooperative pivoted QR for SHORT-WIDE matrices (m <= 256 rows, n >> m):
// the k x n projection B = Q^H A of the range finder, reference call site
// /root/reference/src/qr.rs:311-323 (QR::compute_from_range_estimate -> ?geqp3 of B).
//
// One launch factors the whole matrix.  G workgroups of 512 threads each own a slab of
// columns and keep it in REGISTERS for the whole factorization (8 lanes per column, rows
// interleaved: 8192 x 128 f64 = 8 MB lives in the VGPRs of 32 CUs).  Per Householder step:
//   A  every workgroup finds its best remaining column (largest partial norm, first position
//      on ties = idamax) and posts (norm, position, column index, the column itself) to a
//      per-workgroup slot in global memory
//   -- grid barrier = polling the G slot headers until all carry this step's number --
//   B  every workgroup agrees on the pivot from the headers, fetches the winning column and
//      redundantly generates the reflector (?larfg; identical arithmetic everywhere)
//   C  applies it to its own columns in registers and down-dates their partial norms (?laqp2)
// so HBM sees the matrix once on the way in and once on the way out, and the dependent chain
// of k steps costs k grid barriers instead of 2k kernel launches.  Same ?laqp2 semantics and
// the same in-place output format (R on/above the diagonal, reflectors below it, in the
// physical column jpvt[j]) as the eager chain in kernels_qr.hip.
//
// Communication.  The 8 XCDs of the chip have private L2s, so everything that crosses workgroups
// goes through agent-scope atomics (write-through stores / cache-bypassing loads of 8-byte words):
// no cache write-back or invalidate fences, which would also flush the L2 contents of unrelated
// kernels running beside this one.  A header word is valid iff its low half holds the step number;
// a candidate column is complete before its header is written (workgroup barrier in between).
//
// Co-residency.  The barrier needs all G workgroups on the chip at once.  A device-wide
// budget (semaphore, in CUs) is taken by a one-thread gate kernel launched in front, so that
// concurrently running factorizations of other streams never demand more CUs than exist;
// every other kernel of the library finishes without waiting on anyone, so the G workgroups
// always become resident.  All spins are bounded: on expiry the kernel sets an abort word,
// every workgroup leaves, the certificate flag is raised and the caller falls back to the
// multi-kernel path (or, inside a hipGraph, reports through the health word).
#include "rc_common.hpp"
#include <cstdlib>
#include "rc_device.hpp"

#include <chrono>
#include <mutex>

#include <cerrno>
#include <csignal>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <unistd.h>

namespace rc {

namespace {

// The budget is counted in HALF compute units (a cooperative workgroup that leaves room for a second one on its CU costs
// one unit, one that fills the CU two): three quarters of the device, the rest stays free for everything that is not cooperative.
constexpr int kCoopMaxWgs = 128;
constexpr int kSpinLimit = 1 << 22;      // ~ seconds

template <typename T> struct Tol3z;
template <> struct Tol3z<double> { static __device__ inline double v() { return 1.0536712127723509e-08; } };
template <> struct Tol3z<float> { static __device__ inline float v() { return 2.44140625e-04f; } };

#define RC_AGENT __HIP_MEMORY_SCOPE_AGENT
__device__ inline void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, RC_AGENT); }
__device__ inline void st_agent(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, RC_AGENT); }
__device__ inline void st_agent(long long *p, long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, RC_AGENT); }
__device__ inline void st_agent(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, RC_AGENT); }
__device__ inline double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, RC_AGENT); }
__device__ inline float ld_agent(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, RC_AGENT); }
__device__ inline long long ld_agent(const long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, RC_AGENT); }
__device__ inline unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, RC_AGENT); }

// sync words: [0] finished-workgroup counter, [1] abort (1 = run time, 2 = gate)
// The header slots are cleared on every launch (also on every replay of a captured graph): their
// words carry the step number they belong to, and a stale word of an earlier launch must not pass.
// proceed (optional): a device word of the caller; 0 = do not even try (the optimistic issue of the blocked QRCP found one of its
// assumptions broken: the cooperative kernel behind this gate then leaves through its "the gate gave up" exit, no budget is held)
__global__ void k_coop_gate(unsigned *sem, unsigned need, unsigned budget, unsigned *sync, unsigned long long *hdr, int hdr_words, const int *proceed) {
    if (blockIdx.x != 0) return;
    for (int i = threadIdx.x; i < hdr_words; i += blockDim.x) __hip_atomic_store(hdr + i, 0ull, __ATOMIC_RELAXED, RC_AGENT);
    if (threadIdx.x != 0) return;
    st_agent(sync + 0, 0u);
    st_agent(sync + 2, 0u);  // (k_qrb_coop's commit counter)
    if (proceed && *proceed == 0) { st_agent(sync + 1, 2u); return; }
    unsigned ab = 2u;
    for (int it = 0; it < kSpinLimit; ++it) {
        const unsigned old = __hip_atomic_fetch_add(sem, need, __ATOMIC_RELAXED, RC_AGENT);
        if (old + need <= budget) { ab = 0u; break; }
        __hip_atomic_fetch_sub(sem, need, __ATOMIC_RELAXED, RC_AGENT);
        __builtin_amdgcn_s_sleep(64);
    }
    st_agent(sync + 1, ab);
}

}  // namespace

#ifdef RC_COOP_TIMING
__device__ unsigned long long g_coop_dbg[8];
#endif

template <typename T>
struct WqCoopArgs {
    Mat<T> w;          // m x n column-major input (read once, never written)
    Mat<T> wf;         // m x n column-major output: the factored matrix in ?geqp3 format
    int kmax;
    int cpw;           // columns per workgroup
    int64_t *jpvt;     // n
    T *tau;            // kmax
    unsigned long long *hdr;  // [2][G][5] self-validating words (payload << 32 | step + 1): norm hi, norm lo, position, column, column at position j
    T *cols;           // [2][G][mp]: the candidate columns
    int mp;            // 8 * NE
    unsigned *sync;
    unsigned *sem;
    int *flag;         // certificate word of run_certified (bit 4 = aborted)
    // ---- stages (round 3): the rows above the current step are final, so the slab shrinks as the factorization proceeds; a launch
    // handles the steps [row0, jend) on the rows [row0, m) only -- with the column state handed over through global memory
    int row0;          // first live row = first step of this launch (a multiple of 8); rows < row0 of wf are final
    int jend;          // one past the last step of this launch
    const int *pos_in; // [n] position of every column at entry (nullptr: the identity, first launch)
    int *pos_out;      // [n] position of every column at exit (nullptr: last launch)
    T *vn;             // [2][n] down-dated / reference partial norms (?laqp2's vn1, vn2): read at entry when pos_in != nullptr, written at exit when pos_out != nullptr
    unsigned units;    // budget units to release (2 per workgroup that fills its CU, 1 where two share one)
};

// NE 8-row blocks per column (8 lanes per column, lane l8 holds rows l8 + 8 e), CPG columns per group.
// Register block 0 always holds the 8-row block that contains row j: after every 8 steps the finished
// block (final entries of R / of the reflectors) is written out and the register blocks shift down,
// so row j sits in a register known at compile time and there is no dynamic register indexing.
//
// PROTOCOL INVARIANTS (who writes what, at which scope; reviewed against the code in round 3 -- keep this list and the code in step)
//  I1  Everything that crosses workgroups is an 8-byte agent-scope relaxed atomic (write-through store / L1-bypassing load): the
//      header words a.hdr, the candidate columns a.cols, a.sync[0..2].  Nothing else in global memory is read by a workgroup other
//      than the one that wrote it inside one launch (a.w / a.wf / a.vn / a.pos_* are per-column data of their owner).
//  I2  A header word is (payload << 32) | (step + 1): it validates itself, and a reader accepts a slot only when ALL five words of
//      it carry the step it is waiting for.  Headers are double buffered by step parity; a slot of parity p is rewritten at step
//      j + 2 only after its owner has passed the poll of step j + 1, which needed every workgroup's header of step j + 1, which
//      every workgroup writes after it finished reading the slots of step j.
//  I3  A candidate column is COMPLETE in memory before the header that announces it is written: every storing wave drains its own
//      stores (s_waitcnt vmcnt(0)) -> workgroup barrier -> wave 0 writes the header.  (The barrier alone does not wait for global
//      stores on gfx950.)  Column slots are double buffered like the headers (same argument as I2).
//  I4  Cleared per launch, on the stream, by k_coop_gate in front: all header words (a stale word of an earlier launch at the same
//      address must not validate), sync[0] (finished counter), sync[2]; sync[1] (abort) is written by the gate last.
//  I5  Pivot agreement is computed redundantly from the same 5 x G words by every workgroup with the same instruction sequence,
//      so all workgroups take the same pivot, bit for bit; the reflector is generated redundantly from the same fetched column.
//  I6  Every spin is bounded (kSpinLimit); on expiry the abort word is set (agent scope), every workgroup leaves at its next poll,
//      bit 4 of the certificate is raised and NOTHING of wf / jpvt / tau may be trusted (the caller falls back).
//  I7  The budget taken by the gate (a.units) is released exactly once, by the workgroup whose arrival at sync[0] is the last.
//  I8  Stages: a later launch reads what the earlier one wrote (wf rows >= row0, pos_out, vn) across a kernel boundary on the same
//      stream -- no in-kernel protocol; an aborted earlier stage (flag bit 4) makes the later ones return at once.
template <typename T, int NE, int CPG>
__global__ __launch_bounds__(512, NE <= 4 ? 4 : 2) void k_wq_coop(WqCoopArgs<T> a) {
    constexpr int NG = 64;  // 8-lane groups per workgroup (512 threads: 256 VGPRs per lane, the slab needs 128)
    constexpr int NW = 8;
    constexpr int kNoInt = 0x7fffffff;
    __shared__ double sh_v[NW];
    __shared__ int sh_p[NW], sh_c[NW], sh_j[NW];
    __shared__ int sh_exit;
    __shared__ int bc[4];
    __shared__ T xw[8 * NE];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l8 = tid & 7, g8 = tid >> 3;
    const int m = (int)a.w.rows, mp = a.mp;
    const int64_t n = a.w.cols;
    const int G = gridDim.x, wg = blockIdx.x;
    const int c0 = wg * a.cpw;
    const int nloc = (int)((n - c0) < a.cpw ? (n - c0) : a.cpw);
    unsigned *done = a.sync, *abortw = a.sync + 1;

    const int row0 = a.row0;
    const unsigned ab0 = ld_agent(abortw);
    if (ab0 == 2u) {  // the gate gave up: no budget is held, nothing to release
        if (tid == 0 && wg == 0) atomicOr(a.flag, 4);
        return;
    }
    // an earlier stage gave up (I8): take part in nothing, but release what this launch's gate took
    const bool dead = a.pos_in != nullptr && (__hip_atomic_load(a.flag, __ATOMIC_RELAXED, RC_AGENT) & 4) != 0;
    if (tid == 0) sh_exit = (ab0 != 0u) || dead;

    // ---- load the slab into registers, initial partial norms ---------------------------
    // lane l8 of a group owns the partial norms of the group's column q = l8 (l8 < CPG)
    T x[CPG][NE];
    int pos[CPG];
    T myvn1 = 0, myvn2 = 0;
    const int mycol = c0 + g8 + NG * l8;  // meaningful for l8 < CPG
    const bool resumed = a.pos_in != nullptr;
    const Mat<T> src = resumed ? a.wf : a.w;  // a later stage continues on the partially factored output of the one before
#pragma unroll
    for (int q = 0; q < CPG; ++q) {
        const int cl = g8 + NG * q;
        const bool valid = cl < nloc;
        const T *col = src.p + (int64_t)(c0 + (valid ? cl : 0)) * src.cs;
        const int64_t srs = src.rs;  // 1 for wf; the INPUT may have any strides (the row-major B = Q^H A is read where it lies)
        T ss = 0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int i = row0 + l8 + 8 * e;
            x[q][e] = (valid && i < m) ? col[(int64_t)i * srs] : (T)0;
            ss = fma(x[q][e], x[q][e], ss);
        }
        ss = group_sum_dpp<8>(ss);
        if (l8 == q && !resumed) myvn1 = myvn2 = sqrt(ss);
        pos[q] = valid ? (resumed ? a.pos_in[c0 + cl] : c0 + cl) : -1;
    }
    if (resumed && l8 < CPG && g8 + NG * l8 < nloc) {  // the norms the previous stage left (the down-dated values, not fresh ones: ?laqp2's state)
        myvn1 = a.vn[mycol];
        myvn2 = a.vn[n + mycol];
    }
    __syncthreads();
    bool aborted = sh_exit != 0;

    // per-phase s_memtime totals of workgroup 0 (diagnostic build only: -DRC_COOP_TIMING, tools/coop_timing.py;
    // one stamp costs ~470 cycles)
#ifdef RC_COOP_TIMING
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#define RC_TICK(k)                                                                        \
    {                                                                                     \
        unsigned long long now_;                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");   \
        tph[k] += now_ - tlast;                                                           \
        tlast = now_;                                                                     \
    }
#else
#define RC_TICK(k)
#endif
    int rdone = row0;  // rows below rdone have been written out
    for (int r0 = row0; r0 < a.jend && !aborted; r0 += 8) {
    const int jend = r0 + 8 < a.jend ? r0 + 8 : a.jend;
    for (int j = r0; j < jend && !aborted; ++j) {
        const int par = j & 1;
        const int lj8 = j - r0;  // row j = r0 + lj8 lives in register block 0, lane lj8
        // ---- A: local candidate: largest partial norm, lowest position on ties (idamax) ----
        int mypos = -1;
#pragma unroll
        for (int q = 0; q < CPG; ++q) mypos = (l8 == q) ? pos[q] : mypos;
        const bool speak = l8 < CPG && mypos >= j;
        {
            double best = -1.0;
            if (speak) {
                best = fabs((double)myvn1);
                if (!(best >= 0.0)) best = -0.5;  // NaN norms are taken last
            }
            const double mx = wave_max_dpp(best);
            const int cmin = wave_min_dpp((speak && best == mx) ? mypos : kNoInt);
            const unsigned long long mask = __ballot(speak && best == mx && mypos == cmin);
            const unsigned long long maskj = __ballot(speak && mypos == j);
            if (lane == 0) {
                sh_v[wv] = mask ? mx : -1.0;
                sh_p[wv] = cmin;
                sh_c[wv] = -1;
                sh_j[wv] = -1;
            }
            if (mask && lane == __ffsll((long long)mask) - 1) sh_c[wv] = mycol;
            if (maskj && lane == __ffsll((long long)maskj) - 1) sh_j[wv] = mycol;
        }
        __syncthreads();
        RC_TICK(0)
        double lb = -1.0;
        int lp = kNoInt, lc = -1, lj = -1;
#pragma unroll
        for (int k2 = 0; k2 < NW; ++k2) {
            const int c2 = sh_c[k2];
            if (c2 >= 0 && (lc < 0 || sh_v[k2] > lb || (sh_v[k2] == lb && sh_p[k2] < lp))) { lb = sh_v[k2]; lp = sh_p[k2]; lc = c2; }
            lj = sh_j[k2] > lj ? sh_j[k2] : lj;
        }
        {
            T *slot = a.cols + ((size_t)par * G + wg) * mp + (r0 - row0);  // rows r0 ... of the candidate (slot entries are relative to row0)
#pragma unroll
            for (int q = 0; q < CPG; ++q) {
                if (lc >= 0 && c0 + g8 + NG * q == lc) {
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
                        const int i = r0 + l8 + 8 * e;
                        if (i < m) st_agent(slot + l8 + 8 * e, x[q][e]);
                    }
                }
            }
        }
        // Every store of the candidate column has completed before the header that announces it is written: the storing waves
        // drain their write-through stores themselves -- on gfx950 a workgroup barrier waits for LDS traffic only (s_waitcnt
        // lgkmcnt(0); s_barrier, no vmcnt(0) outside threadgroup-split mode) and the header is written by another wave, so under
        // memory load it could overtake the column (kernels_qrblk.hip, k_qrb_coop, has the failure this caused there).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        RC_TICK(1)
        if (wv == 0) {
            const unsigned long long tag = (unsigned)(j + 1);
            if (lane == 0) {
                unsigned long long *h = a.hdr + ((size_t)par * G + wg) * 5;
                const unsigned long long nb = (unsigned long long)__double_as_longlong(lc >= 0 ? lb : -1.0);
                __hip_atomic_store(h + 0, (nb & 0xffffffff00000000ull) | tag, __ATOMIC_RELAXED, RC_AGENT);
                __hip_atomic_store(h + 1, (nb << 32) | tag, __ATOMIC_RELAXED, RC_AGENT);
                __hip_atomic_store(h + 2, ((unsigned long long)(unsigned)(lc >= 0 ? lp : kNoInt) << 32) | tag, __ATOMIC_RELAXED, RC_AGENT);
                __hip_atomic_store(h + 3, ((unsigned long long)(unsigned)lc << 32) | tag, __ATOMIC_RELAXED, RC_AGENT);
                __hip_atomic_store(h + 4, ((unsigned long long)(unsigned)lj << 32) | tag, __ATOMIC_RELAXED, RC_AGENT);
            }
            // ---- grid barrier + pivot agreement in one: poll the G headers until all carry this step's tag
            double hb = -2.0;
            int hp = kNoInt, hc = -1, hs = -1, hj = -1;
            bool ab = false;
            for (int it = 0;; ++it) {
                bool ok = true;
                hb = -2.0; hp = kNoInt; hc = -1; hs = -1; hj = -1;
                for (int g = lane; g < G; g += 64) {
                    const unsigned long long *h = a.hdr + ((size_t)par * G + g) * 5;
                    const unsigned long long w0 = __hip_atomic_load(h + 0, __ATOMIC_RELAXED, RC_AGENT), w1 = __hip_atomic_load(h + 1, __ATOMIC_RELAXED, RC_AGENT),
                                             w2 = __hip_atomic_load(h + 2, __ATOMIC_RELAXED, RC_AGENT), w3 = __hip_atomic_load(h + 3, __ATOMIC_RELAXED, RC_AGENT),
                                             w4 = __hip_atomic_load(h + 4, __ATOMIC_RELAXED, RC_AGENT);
                    ok = ok && (unsigned)w0 == (unsigned)tag && (unsigned)w1 == (unsigned)tag && (unsigned)w2 == (unsigned)tag && (unsigned)w3 == (unsigned)tag &&
                         (unsigned)w4 == (unsigned)tag;
                    const double nb = __longlong_as_double((long long)((w0 & 0xffffffff00000000ull) | (w1 >> 32)));
                    const int p2 = (int)(w2 >> 32), c2 = (int)(w3 >> 32), j2 = (int)(w4 >> 32);
                    if (c2 >= 0 && (hc < 0 || nb > hb || (nb == hb && p2 < hp))) { hb = nb; hp = p2; hc = c2; hs = g; }
                    hj = j2 > hj ? j2 : hj;
                }
                if (__all(ok)) break;
                if (it > kSpinLimit || ((it & 31) == 31 && __any(ld_agent(abortw) != 0u))) { ab = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            const double mx = wave_max_dpp(hc >= 0 ? hb : -2.0);
            const int wp = wave_min_dpp((hc >= 0 && hb == mx) ? hp : kNoInt);
            const unsigned long long mask = __ballot(hc >= 0 && hb == mx && hp == wp);
            const int src = mask ? __ffsll((long long)mask) - 1 : 0;
            const int wc = mask ? __builtin_amdgcn_readlane(hc, src) : -1;
            const int ws = mask ? __builtin_amdgcn_readlane(hs, src) : -1;
            const int cj = wave_max_dpp(hj);
            if (lane == 0) {
                if (!ab && (wc < 0 || ws < 0)) ab = true;  // cannot happen while j < n; leave cleanly instead of trusting it
                if (ab) st_agent(abortw, 1u);
                sh_exit = ab ? 1 : 0;
                bc[0] = wc; bc[1] = ws; bc[2] = wp; bc[3] = cj;
            }
        }
        __syncthreads();
        RC_TICK(2)
        if (sh_exit) { aborted = true; break; }
        const int wcol = bc[0], wslot = bc[1], wpos = bc[2], colj = bc[3];
        // ---- B: fetch the pivot column (complete before its header was posted); each wave then
        //         generates the reflector redundantly (?larfg): identical arithmetic everywhere ----
        if (tid < mp && row0 + tid >= j && row0 + tid < m) xw[tid] = ld_agent(a.cols + ((size_t)par * G + wslot) * mp + tid);  // xw[i - row0] = row i
        __syncthreads();
        RC_TICK(3)
        T tj = 0, beta, scal = 0;
        {
            T ss = 0;
            for (int i = j + 1 + lane; i < m; i += 64) ss = fma(xw[i - row0], xw[i - row0], ss);
            ss = wave_sum_dpp(ss);
            const T xnorm = sqrt(ss);
            const T alpha = xw[j - row0];
            beta = alpha;
            if (xnorm != (T)0) {
                // ?lapy2: w sqrt(1 + (z / w)^2) with w = max(|alpha|, xnorm)
                const T aa = fabs(alpha);
                const T wmax = aa > xnorm ? aa : xnorm, zmin = aa > xnorm ? xnorm : aa;
                const T zr = zmin / wmax;
                beta = -copysign(wmax * sqrt(fma(zr, zr, (T)1)), alpha);
                tj = (beta - alpha) / beta;
                scal = (T)1 / (alpha - beta);
            }
        }
        if (wg == 0 && tid == 0) a.tau[j] = tj;
        T v[NE];  // the reflector, this lane's rows r0 + l8 + 8 e: zero outside [j, m)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int i = r0 + l8 + 8 * e;
            v[e] = (i < j || i >= m) ? (T)0 : (i == j) ? (T)1 : xw[i - row0] * scal;
        }

        // ---- C: bookkeeping, reflector application, norm down-date -------------------------
        RC_TICK(4)
        T myaj = 0;
#pragma unroll
        for (int q = 0; q < CPG; ++q) {
            const int c = c0 + g8 + NG * q;
            if (pos[q] < 0) continue;
            if (c == wcol) {
                // pivot column: R(j, j) = beta, the reflector goes below the diagonal (?geqp3 format)
                pos[q] = j;
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const int i = r0 + l8 + 8 * e;
                    if (i == j) x[q][e] = beta;
                    else if (i > j && i < m) x[q][e] = v[e];
                }
                continue;
            }
            if (c == colj && wpos != j) pos[q] = wpos;
            if (pos[q] <= j) continue;  // already factored
            if (tj != (T)0) {
                T dot = 0;
#pragma unroll
                for (int e = 0; e < NE; ++e) dot = fma(v[e], x[q][e], dot);
                dot = group_sum_dpp<8>(dot);
                const T f = tj * dot;
#pragma unroll
                for (int e = 0; e < NE; ++e) x[q][e] = fma(-f, v[e], x[q][e]);
            }
            const T aj = group_sum_dpp<8>((l8 == lj8) ? x[q][0] : (T)0);  // exactly one lane contributes
            if (l8 == q) myaj = aj;
        }
        {
            // ?laqp2 down-date by the lane that owns the column's norms; the rare exact recomputation
            // (temp2 <= tol3z) is voted within the group and done by all 8 lanes of it
            int np = -1;
#pragma unroll
            for (int q = 0; q < CPG; ++q) np = (l8 == q) ? pos[q] : np;
            int need = 0;
            if (l8 < CPG && np > j && myvn1 != (T)0) {
                const T t = fabs(myaj) / myvn1;
                T temp = (T)1 - t * t;
                temp = temp > (T)0 ? temp : (T)0;
                const T r = myvn1 / myvn2;
                const T temp2 = temp * r * r;
                if (temp2 <= Tol3z<T>::v()) need = 1 << l8;
                else myvn1 = myvn1 * sqrt(temp);
            }
            need = group_sum_dpp<8>(need);
            if (need) {
#pragma unroll
                for (int q = 0; q < CPG; ++q) {
                    if (!(need & (1 << q))) continue;
                    T ssl = 0;
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
                        const int i = r0 + l8 + 8 * e;
                        if (i > j) ssl = fma(x[q][e], x[q][e], ssl);  // rows >= m hold zeros
                    }
                    ssl = group_sum_dpp<8>(ssl);
                    if (l8 == q) {
                        const T nn = (j < m - 1) ? sqrt(ssl) : (T)0;
                        myvn1 = nn;
                        myvn2 = nn;
                    }
                }
            }
        }
        RC_TICK(5)
    }
    // ---- the 8-row block is finished: write it out, shift the register blocks down -------------
#pragma unroll
    for (int q = 0; q < CPG; ++q) {
        if (!aborted && pos[q] >= 0 && r0 + l8 < m) a.wf.p[(int64_t)(c0 + g8 + NG * q) * a.wf.cs + r0 + l8] = x[q][0];
#pragma unroll
        for (int e = 0; e + 1 < NE; ++e) x[q][e] = x[q][e + 1];
        x[q][NE - 1] = 0;
    }
    rdone = r0 + 8;
    }
#undef RC_TICK
#ifdef RC_COOP_TIMING
    if (wg == 0 && tid == 0)
        for (int k2 = 0; k2 < 6; ++k2) g_coop_dbg[k2] = tph[k2];
#endif

    // ---- write the not yet written blocks, the permutation, and release the budget ------------
    if (!aborted) {
#pragma unroll
        for (int q = 0; q < CPG; ++q) {
            if (pos[q] < 0) continue;
            T *col = a.wf.p + (int64_t)(c0 + g8 + NG * q) * a.wf.cs;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int i = rdone + l8 + 8 * e;
                if (i < m) col[i] = x[q][e];
            }
            // (positions are a permutation at every step: the entries written by the last stage that held a column are the final ones)
            if (l8 == 0) a.jpvt[pos[q]] = c0 + g8 + NG * q;
            if (l8 == 0 && a.pos_out) a.pos_out[c0 + g8 + NG * q] = pos[q];
        }
        if (a.pos_out && l8 < CPG && g8 + NG * l8 < nloc) {  // hand the norms over to the next stage
            a.vn[mycol] = myvn1;
            a.vn[n + mycol] = myvn2;
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (aborted && !dead) atomicOr(a.flag, 4);
        const unsigned old = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, RC_AGENT);
        if (old == (unsigned)G - 1u) __hip_atomic_fetch_sub(a.sem, a.units, __ATOMIC_RELAXED, RC_AGENT);
    }
}

// ---- host side ---------------------------------------------------------------------------
static unsigned *coop_semaphore(int device) {
    static std::mutex mu;
    static unsigned *sems[64] = {};
    std::lock_guard<std::mutex> lk(mu);
    unsigned *&s = sems[device & 63];
    if (!s) {
        RC_HIP(hipMalloc(reinterpret_cast<void **>(&s), 64));
        const unsigned zeros[16] = {};
        RC_HIP(hipMemcpy(s, zeros, 64, hipMemcpyHostToDevice));  // blocking copy from pageable memory: complete before any stream can see the word
    }
    return s;
}
void coop_prepare(int device) { (void)coop_semaphore(device); }
unsigned *coop_semaphore_of(int device) { return coop_semaphore(device); }

// ---- the budget is shared between the PROCESSES that use a device ------------------------------------------------------------
// The semaphore above lives in one process' device memory, so by itself the budget is per process, and processes that share a GPU
// (tests/test_gpu_sharded.py: two or three ranks on one device) could together ask for more co-resident workgroups than the chip has;
// the cooperative kernels then ran into their bounded spins and fell back -- correct, but seconds late.  Every process that uses the
// cooperative kernels on a device therefore registers its pid in a small shared-memory table keyed by the device's PCI bus id
// (/dev/shm/rc_amd_coop_<uid>_<busid>), and a process' budget is the device budget divided by the number of LIVE registered processes
// (kill(pid, 0)), re-evaluated at most every 100 ms: the sum of what all processes may hold never exceeds the device budget once the
// shares have settled (a process that was alone keeps its larger share for at most that long after a second one arrives; the
// bounded spins cover that window).  A share too small for a kernel makes *_supported() false: the non-cooperative path runs.
// One process per GPU -- the deployment -- always has the whole budget.  RC_COOP_SHARE=0 switches the table off.
namespace {
struct ShareTable {
    uint32_t magic;
    uint32_t reserved;
    int32_t pids[62];
};
constexpr uint32_t kShareMagic = 0x52434350u;  // "RCCP"

struct DeviceShare {
    ShareTable *tab = nullptr;
    int fd = -1;
    bool tried = false;
    int live = 1;
    std::chrono::steady_clock::time_point stamp{};
};

int live_processes(int device, DeviceShare &ds) {
    static const bool enabled = [] { const char *e = getenv("RC_COOP_SHARE"); return !(e && atoi(e) == 0); }();
    if (!enabled) return 1;
    const auto now = std::chrono::steady_clock::now();
    if (ds.tried && (ds.tab == nullptr || now - ds.stamp < std::chrono::milliseconds(100))) return ds.live;
    const int me = (int)getpid();
    if (!ds.tried) {
        ds.tried = true;
        char bus[64] = "dev";
        if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) snprintf(bus, sizeof(bus), "dev%d", device);
        for (char *p = bus; *p; ++p)
            if (*p == ':' || *p == '.' || *p == '/') *p = '_';
        char path[160];
        snprintf(path, sizeof(path), "/dev/shm/rc_amd_coop_%u_%s", (unsigned)getuid(), bus);
        ds.fd = open(path, O_RDWR | O_CREAT | O_CLOEXEC, 0600);
        if (ds.fd >= 0 && ftruncate(ds.fd, (off_t)sizeof(ShareTable)) == 0) {
            void *m = mmap(nullptr, sizeof(ShareTable), PROT_READ | PROT_WRITE, MAP_SHARED, ds.fd, 0);
            if (m != MAP_FAILED) ds.tab = static_cast<ShareTable *>(m);
        }
        if (!ds.tab) {  // no shared memory here: the budget stays per process (as before round 3)
            if (ds.fd >= 0) close(ds.fd);
            ds.fd = -1;
            return ds.live = 1;
        }
    }
    // register (once) and count under the file lock; slots of dead processes are reclaimed
    int live = 0;
    bool mine = false;
    if (flock(ds.fd, LOCK_EX) == 0) {
        if (ds.tab->magic != kShareMagic) {
            std::memset(ds.tab, 0, sizeof(ShareTable));
            ds.tab->magic = kShareMagic;
        }
        int free_slot = -1;
        for (int i = 0; i < 62; ++i) {
            const int pid = ds.tab->pids[i];
            if (pid == me) { mine = true; ++live; continue; }
            if (pid > 0 && (kill(pid, 0) == 0 || errno == EPERM)) { ++live; continue; }
            ds.tab->pids[i] = 0;
            if (free_slot < 0) free_slot = i;
        }
        if (!mine && free_slot >= 0) { ds.tab->pids[free_slot] = me; mine = true; ++live; }
        (void)flock(ds.fd, LOCK_UN);
    }
    ds.stamp = now;
    return ds.live = std::max(1, mine ? live : live + 1);
}
}  // namespace

unsigned coop_budget_units(int device) {
    static std::mutex mu;
    static unsigned total[64] = {};
    static DeviceShare share[64];
    std::lock_guard<std::mutex> lk(mu);
    unsigned &b = total[device & 63];
    if (!b) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
        const char *e = getenv("RC_COOP_BUDGET");  // in CUs (experiments)
        const int v = e ? atoi(e) : 0;
        b = (unsigned)(v >= 32 && v <= cus ? 2 * v : 2 * cus * 3 / 4);
    }
    const int procs = live_processes(device, share[device & 63]);
    return std::max(2u, b / (unsigned)procs);
}

// one-thread gate in front of a cooperative kernel: takes `need` units of the device-wide budget (released by the cooperative
// kernel's last workgroup), clears the kernel's header words and its sync words
void coop_gate_launch(rc_context *c, unsigned need, unsigned *sync, unsigned long long *hdr, int hdr_words, const int *proceed) {
    // (the caller checked need <= budget; the share of this process may have shrunk since -- another process arrived on the device --
    // in which case this launch still goes through once nothing else of this process holds units)
    const unsigned budget = std::max(coop_budget_units(c->device), need);
    hipLaunchKernelGGL(k_coop_gate, dim3(1), dim3(256), 0, c->stream, coop_semaphore(c->device), need, budget, sync, hdr, hdr_words, proceed);
}

// One stage: NE 8-row blocks per column in registers (the live rows [row0, m) must fit: 8 NE >= m - row0), CPG columns per 8-lane group.
static void stage_shape(int ne, int64_t n, int *cpg, int *cpw, int *g) {
    *cpg = ne >= 32 ? 2 : ne >= 16 ? 4 : 8;  // at most 64 matrix elements per lane; a group owns at most 8 columns (one norm owner each)
    const int64_t cap = 64 * (int64_t)*cpg;
    const int64_t wgs = (n + cap - 1) / cap;
    *g = (int)wgs;
    *cpw = (int)((n + wgs - 1) / wgs);
}
static int stage_ne(int64_t live_rows) { return live_rows <= 32 ? 4 : live_rows <= 64 ? 8 : live_rows <= 128 ? 16 : 32; }

template <typename T>
bool wide_coop_supported(int64_t m, int64_t n, int device) {
    if (!(m >= 2 && m <= 256 && n >= 4 * m && n >= 256)) return false;
    int cpg, cpw, g;
    stage_shape(stage_ne(m), n, &cpg, &cpw, &g);
    // (a device with few CUs -- a partitioned one -- cannot hold the launch: the non-cooperative path runs instead)
    return g >= 1 && g <= kCoopMaxWgs && n < (int64_t)1 << 30 && 2u * (unsigned)g <= coop_budget_units(device);
}

// w: m x n column-major input (left untouched); wf: m x n column-major output, the factorization in
// the format of geqp3_inplace (R on/above the diagonal, the reflector below it in physical column
// jpvt[j]); flag gets bit 4 when the kernel had to give up (wf, jpvt, tau are then garbage).
//
// Stages (round 3).  The rows above the current step are final, so the slab a launch has to keep in registers shrinks: the
// factorization runs as a short sequence of launches, each on HALF the live rows' worth of steps with the smallest register
// layout that holds the live rows -- 128 x 8192: steps 0..63 on 32 workgroups (16 blocks per column), 64..95 on 16 workgroups
// (8 blocks), 96..127 on 16 workgroups of half a CU each (4 blocks; two share a CU) = 2816 instead of 4096 CU-steps.  The column
// state a later stage needs besides the matrix -- positions and ?laqp2's two norm vectors -- travels through global memory, so
// the pivots, R and the reflectors are bit for bit those of the single launch (RC_WQ_STAGES=0 runs that; test
// test_wide_qrcp_paths_match_lapack_and_each_other compares them).
template <typename T>
void geqp3_wide_coop(rc_context *c, Mat<T> w, Mat<T> wf, int64_t kmax, int64_t *jpvt, T *tau, int *flag) {
    RC_REQUIRE(wf.rs == 1 && wf.rows == w.rows && wf.cols == w.cols, RC_LAYOUT_ERROR, "geqp3_wide_coop: column-major output required");  // w: any strides, read once, never written
    const int64_t m = w.rows, n = w.cols;
    kmax = std::min(kmax, std::min(m, n));
    RC_REQUIRE(wide_coop_supported<T>(m, n, c->device), RC_INVALID_ARGUMENT, "geqp3_wide_coop: unsupported shape");
    static const int staged = [] { const char *e = getenv("RC_WQ_STAGES"); return e ? atoi(e) : 1; }();
    int *pos = nullptr;
    T *vn = nullptr;
    int64_t row0 = 0;
    bool first = true;
    while (row0 < kmax) {
        const int ne = stage_ne(m - row0);
        // this stage runs until half of its live rows are consumed (then the next smaller layout fits), the smallest layout to the end
        int64_t jend = kmax;
        if (staged && ne > 4) jend = std::min<int64_t>(kmax, row0 + 4 * (int64_t)ne);
        const bool last = jend >= kmax;
        int cpg, cpw, g;
        stage_shape(ne, n, &cpg, &cpw, &g);
        ProfScope ps(c, "op:geqp3_wide_coop %lldx%lld steps %lld..%lld wgs=%d", (long long)m, (long long)n, (long long)row0, (long long)jend, g);
        if (!last || !first) {
            if (!pos) {
                pos = c->alloc<int>((size_t)n);
                vn = c->alloc<T>((size_t)2 * n);
            }
        }
        WqCoopArgs<T> a;
        a.w = w;
        a.wf = wf;
        a.kmax = (int)kmax;
        a.cpw = cpw;
        a.jpvt = jpvt;
        a.tau = tau;
        a.mp = 8 * ne;
        a.hdr = c->alloc<unsigned long long>((size_t)2 * g * 5);
        a.cols = c->alloc<T>((size_t)2 * g * a.mp);
        a.sync = c->alloc<unsigned>(4);
        a.sem = coop_semaphore(c->device);
        a.flag = flag;
        a.row0 = (int)row0;
        a.jend = (int)jend;
        a.pos_in = first ? nullptr : pos;
        a.pos_out = last ? nullptr : pos;
        a.vn = vn;
        // 512 threads x up to 256 VGPRs: one workgroup fills its CU = 2 units; the 4-block layout is held to 128 VGPRs: 1 unit
        a.units = (ne <= 4 ? 1u : 2u) * (unsigned)g;
        coop_gate_launch(c, a.units, a.sync, a.hdr, 2 * g * 5);
#define RC_COOP(NE_, CPG_) hipLaunchKernelGGL((k_wq_coop<T, NE_, CPG_>), dim3((unsigned)g), dim3(512), 0, c->stream, a)
        if (ne == 4) RC_COOP(4, 8);
        else if (ne == 8) RC_COOP(8, 8);
        else if (ne == 16) RC_COOP(16, 4);
        else RC_COOP(32, 2);
#undef RC_COOP
        row0 = jend;
        first = false;
    }
}

#ifdef RC_COOP_TIMING
extern "C" void rc_debug_coop_timing(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_coop_dbg), 8 * sizeof(unsigned long long)); }
#endif
template bool wide_coop_supported<double>(int64_t, int64_t, int);
template bool wide_coop_supported<float>(int64_t, int64_t, int);
template void geqp3_wide_coop<double>(rc_context *, Mat<double>, Mat<double>, int64_t, int64_t *, double *, int *);
template void geqp3_wide_coop<float>(rc_context *, Mat<float>, Mat<float>, int64_t, int64_t *, float *, int *);

}  // namespace rc
