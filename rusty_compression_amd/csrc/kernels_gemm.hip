// MFMA GEMM for gfx950: C = alpha * A * B + beta * C on strided views.
//
// Replaces the reference's per-column gemv loops (MatMat / ConjMatMat blanket
// impls, /root/reference/src/types.rs:60-70, :90-100, :145-146) and every
// ndarray `.dot` of two matrices on the hot path.
//
//  * f64: v_mfma_f64_16x16x4_f64, f32: v_mfma_f32_16x16x4_f32 (exact f32).
//    A-operand lane l holds A[l&15][l>>4], B-operand lane l holds B[l>>4][l&15];
//    C/D: col = l&15, row = (l>>4) + 4*reg (f64) or 4*(l>>4) + reg (f32).
//  * Tiles are staged global -> registers -> LDS (double buffered); the LDS
//    image keeps the operand's own contiguous dimension fastest, so global
//    reads coalesce for either orientation and no transposing store is needed;
//    row pitches are chosen so every fragment ds_read is bank-conflict free.
//    When both operands are 16-byte aligned along their contiguous dimension
//    (VEC = 2) the staging uses 2-element vector loads / LDS stores, and every
//    thread keeps its global pointers in registers (no per-element 64-bit
//    index arithmetic in the K loop).
//  * Split-K with a DETERMINISTIC slab reduction (no float atomics): pivot
//    decisions downstream must not depend on arrival order.
//  * XCD-aware tile order: consecutive block ids land on different XCDs, so the
//    linear id is remapped to give each XCD a contiguous run of row tiles that
//    share the B panel in that XCD's L2.
#include "rc_common.hpp"
#include "rc_gemm.hpp"

#include <cstdlib>

namespace rc {

static __host__ __device__ inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

template <typename T> struct Acc;
template <> struct Acc<double> {
    typedef double4_t type;
    typedef double2_t vec2;
    static __device__ inline type mfma(double a, double b, type c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ inline int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct Acc<float> {
    typedef float4_t type;
    typedef float2_t vec2;
    static __device__ inline type mfma(float a, float b, type c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ inline int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

// smallest pitch >= n with pitch % 32 == 16 (conflict-free M/N-fastest fragment reads, 16x16x4 MFMA)
constexpr int pitch16(int n) { return n + ((16 - n % 32) + 32) % 32; }
// smallest pitch >= n with pitch % 8 == 4 (same for the 4x4x4 f64 MFMA fragments)
constexpr int pitch4(int n) { return n + ((4 - n % 8) + 8) % 8; }

// One operand tile: R rows (the operand's non-reduction index) x BK, staged by NT threads.
//   LAY == 0: the reduction index k is contiguous in memory (stride_k == 1 when VEC == 2)
//   LAY == 1: the row index is contiguous in memory (stride_r == 1 when VEC == 2)
// LDS image: LAY == 0 -> [R][P], P = BK + 2 (16x16x4 MFMA fragments) or BK + 8 (4x4x4 fragments);
//            LAY == 1 -> [BK][P], P = pitch16(R) or pitch4(R).  Both choices make every fragment
//            ds_read_b64 of the corresponding MFMA shape bank-conflict free.
template <typename T, int LAY, int R, int BK, int NT, int VEC, int MODE = 0>
struct TileStager {
    static constexpr int P = LAY == 0 ? (MODE == 0 ? BK + 2 : BK + 8) : (MODE == 0 ? pitch16(R) : pitch4(R));
    static constexpr int ELEMS = LAY == 0 ? R * P : BK * P;
    static constexpr int NVEC = R * BK / VEC;
    static constexpr int PER_T = (NVEC + NT - 1) / NT;
    typedef typename Acc<T>::vec2 vec2;

    const T *ptr[PER_T];  // global address of the vector at k-step 0
    T val[PER_T][VEC];

    static __device__ inline void coords(int idx, int &r, int &k) {
        if (LAY == 0) { k = (idx % (BK / VEC)) * VEC; r = idx / (BK / VEC); }
        else { r = (idx % (R / VEC)) * VEC; k = idx / (R / VEC); }
    }
    __device__ inline void init(const T *base, int64_t r0, int64_t kbeg, int64_t sr, int64_t sk, int tid) {
#pragma unroll
        for (int e = 0; e < PER_T; ++e) {
            int r, k;
            coords(tid + e * NT, r, k);
            ptr[e] = base + (r0 + r) * sr + (kbeg + k) * sk;
        }
    }
    // loads the tile whose first reduction index is k0 (absolute); koff_elems = (k0 - kbeg) * sk
    __device__ inline void load(int64_t r0, int64_t rmax, int64_t k0, int64_t kend, int64_t koff_elems, int64_t sr, int64_t sk, int tid) {
#pragma unroll
        for (int e = 0; e < PER_T; ++e) {
            const int idx = tid + e * NT;
            int r, k;
            coords(idx, r, k);
            const T *p = ptr[e] + koff_elems;
            const int64_t gr = r0 + r, gk = k0 + k;
#pragma unroll
            for (int v = 0; v < VEC; ++v) val[e][v] = 0;
            if (idx < NVEC) {
                if (VEC == 2) {
                    const bool full = LAY == 0 ? (gr < rmax && gk + 2 <= kend) : (gr + 2 <= rmax && gk < kend);
                    if (full) {
                        const vec2 t = *reinterpret_cast<const vec2 *>(p);
                        val[e][0] = t[0];
                        val[e][VEC - 1] = t[1];
                    } else {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const int64_t r2 = LAY == 0 ? gr : gr + v, k2 = LAY == 0 ? gk + v : gk;
                            if (r2 < rmax && k2 < kend) val[e][v] = p[LAY == 0 ? v * sk : v * sr];
                        }
                    }
                } else {
                    if (gr < rmax && gk < kend) val[e][0] = *p;
                }
            }
        }
    }
    __device__ inline void store(T *lds, int tid) const {
#pragma unroll
        for (int e = 0; e < PER_T; ++e) {
            const int idx = tid + e * NT;
            int r, k;
            coords(idx, r, k);
            const int off = LAY == 0 ? r * P + k : k * P + r;
            if (idx < NVEC) {
                if (VEC == 2) {
                    vec2 t;
                    t[0] = val[e][0];
                    t[1] = val[e][VEC - 1];
                    *reinterpret_cast<vec2 *>(lds + off) = t;
                } else {
                    lds[off] = val[e][0];
                }
            }
        }
    }
};

// ALAY: 0 = A is K-contiguous, 1 = A is M-contiguous ; BLAY: 0 = B is N-contiguous, 1 = B is K-contiguous
// NBUF: 2 = double-buffered LDS (one barrier per K step), 1 = single LDS buffer + register
// prefetch (two barriers per K step, half the LDS: deeper BK or more workgroups per CU)
template <typename T, int ALAY, int BLAY, int BM, int BN, int BK, int WM, int WN, int VEC, int NBUF>
__global__ __launch_bounds__(WM *WN * 64) void k_gemm_mfma(GemmArgs<T> g) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 16;  // 16x16 tiles per wave along M
    constexpr int TN = BN / WN / 16;
    static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0 && BK % 4 == 0, "tile shape");
    typedef TileStager<T, ALAY, BM, BK, NT, VEC> StA;               // rows = m
    typedef TileStager<T, BLAY == 1 ? 0 : 1, BN, BK, NT, VEC> StB;  // rows = n (B N-contiguous == row-contiguous)
    constexpr int PA = StA::P, PB = StB::P;
    constexpr int A_ELEMS = StA::ELEMS, B_ELEMS = StB::ELEMS;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *smem = reinterpret_cast<T *>(smem_raw);
    // LDS addresses are formed arithmetically from the one __shared__ array (an array of
    // pointers would decay to generic pointers: flat_load instead of ds_read, and every
    // fragment read would then also wait for the global prefetch, vmcnt(0))
    constexpr int STAGE = NBUF == 2 ? A_ELEMS + B_ELEMS : 0;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, kk4 = lane >> 4;

    // ---- tile id with XCD-aware remap (bijective for any tile count) --------
    const int ntiles = g.tiles_m * g.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % g.tiles_n, tile_m = bid / g.tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    const int split = blockIdx.y;
    const int64_t kbeg = (int64_t)split * g.kchunk;
    const int64_t kend = min(g.K, kbeg + g.kchunk);

    typename Acc<T>::type acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = typename Acc<T>::type{0, 0, 0, 0};

    StA sa;
    StB sb;
    sa.init(g.a, m0, kbeg, g.sam, g.sak, tid);
    sb.init(g.b, n0, kbeg, g.sbn, g.sbk, tid);

    const int64_t nk = kend > kbeg ? cdiv(kend - kbeg, BK) : 0;
    if (nk > 0) {
        sa.load(m0, g.M, kbeg, kend, 0, g.sam, g.sak, tid);
        sb.load(n0, g.N, kbeg, kend, 0, g.sbn, g.sbk, tid);
        sa.store(smem, tid);
        sb.store(smem + A_ELEMS, tid);
    }
    __syncthreads();
    for (int64_t it = 0; it < nk; ++it) {
        const int buf = (int)(it & 1);
        if (it + 1 < nk) {  // global loads in flight during the MFMAs
            const int64_t koff = (it + 1) * BK;
            sa.load(m0, g.M, kbeg + koff, kend, koff * g.sak, g.sam, g.sak, tid);
            sb.load(n0, g.N, kbeg + koff, kend, koff * g.sbk, g.sbn, g.sbk, tid);
        }
        const T *as = smem + buf * STAGE, *bs = smem + A_ELEMS + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            T af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int mm = wm * (TM * 16) + i * 16 + r16, kk = ks * 4 + kk4;
                af[i] = ALAY == 0 ? as[mm * PA + kk] : as[kk * PA + mm];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int nn = wn * (TN * 16) + j * 16 + r16, kk = ks * 4 + kk4;
                bf[j] = BLAY == 1 ? bs[nn * PB + kk] : bs[kk * PB + nn];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Acc<T>::mfma(af[i], bf[j], acc[i][j]);
        }
        if (NBUF == 1) __syncthreads();  // everyone is done reading the single buffer
        if (it + 1 < nk) {  // NBUF == 2: the other buffer was last read one iteration ago
            sa.store(smem + (buf ^ 1) * STAGE, tid);
            sb.store(smem + A_ELEMS + (buf ^ 1) * STAGE, tid);
        }
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int64_t gm = m0 + wm * (TM * 16) + i * 16 + Acc<T>::row(lane, r);
                int64_t gn = n0 + wn * (TN * 16) + j * 16 + r16;
                if (gm < g.M && gn < g.N) {
                    T v = acc[i][j][r];
                    if (g.splits > 1) {
                        g.partial[((int64_t)split * g.M + gm) * g.N + gn] = v;
                    } else {
                        T *cp = g.c + gm * g.scm + gn * g.scn;
                        *cp = g.beta == (T)0 ? g.alpha * v : g.alpha * v + g.beta * (*cp);
                    }
                }
            }
}

// ---------------------------------------------------------------------------
// f64 GEMM on v_mfma_f64_4x4x4_4b_f64.
//
// Measured on MI355X (tools/microbench, profiles/r01_microbench_mfma_f64.txt): the 16x16x4
// f64 MFMA sustains only 35-49 TFLOP/s, the 4x4x4 four-block form 70-75 TFLOP/s (the
// 78.6 TFLOP/s datasheet rate) once >= ~64 independent accumulators are in flight, so every
// f64 product of the hot path uses this shape.
//
// Lane layout (probed, tools/microbench/mfma_f64_4x4x4_layout.hip): with kl = l >> 4,
// b = (l >> 2) & 3, i/j = l & 3:
//   A: lane holds A_b[i][kl];  B: lane holds B_b[kl][j];  D: lane l = 16*i + 4*b + j holds D_b[i][j]
// for four independent blocks b.  The blocks are used as four output sub-tiles that SHARE one
// operand (an LDS broadcast read):
//   ORIENT 0: A shared  -> one instruction is a  4 x 16 x 4 product (row = l >> 4, col = l & 15)
//   ORIENT 1: B shared  -> one instruction is a 16 x  4 x 4 product
// so all 64 result lanes are distinct outputs, rows of C are written in 128-byte segments
// (ORIENT 0) and the LDS images / pitches are exactly those of the 16x16x4 kernel.
// ---------------------------------------------------------------------------
#ifdef RC_GEMM_TIMING
// diagnostic build (tools/gemm_timing.py): wave 0 of workgroup (0, 0) accumulates s_memtime deltas per phase of the main loop
__device__ unsigned long long g_gemm_dbg[8];
#define RC_STAMP(k)                                                                              \
    {                                                                                            \
        unsigned long long now_;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        tacc[k] += now_ - tlast;                                                                 \
        tlast = now_;                                                                            \
    }
#else
#define RC_STAMP(k)
#endif

// GLDS: the B operand tile goes global -> LDS directly (global_load_lds_dwordx4: no staging registers, no LDS write
// pass); needs B stored [K][N] with unit N stride (BLAY == 0: every k-row of the tile is BN contiguous doubles, copied
// as BN / 128 pieces of 1 KiB into the padded LDS row), full tiles, 16-byte aligned rows (checked by the host).
// GLDS == 2: the A tile as well (A stored [K][M] with unit M stride, ALAY == 1: one 1 KiB piece per k-row for the
// first 128 columns, plus a masked tail piece when 128 < M <= BM = 136; rows of the tile beyond M keep stale LDS
// contents, which only reach output rows that are never stored).
// With BLAY == 1 (B stored [N][K], unit K stride: a tile row is 16 doubles = 128 bytes) one piece covers 8 tile rows, so
// the LDS image cannot be padded; it is [BN][16] with the eight 16-byte chunks of row n stored at chunk ^ ((n >> 1) & 7)
// (applied to the per-lane SOURCE address), which keeps the 4 x 16 fragment reads at the two-pass minimum.
// GLDS == 3: as 2 with THREE LDS buffers: the copies of tile it + 2 are issued before tile it is computed and stay in
// flight across the barrier (counted s_waitcnt vmcnt, raw LDS-only barrier), so no wave waits for HBM at the barrier.
template <int ALAY, int BLAY, int BM, int BN, int BK, int WM, int WN, int VEC, int ORIENT, int GLDS = 0>
__global__ __launch_bounds__(WM *WN * 64) void k_gemm_f64q(GemmArgs<double> g) {
    typedef double T;
    constexpr int NT = WM * WN * 64;
    constexpr int WR = BM / WM, WC = BN / WN;               // wave tile
    constexpr int TM = ORIENT == 0 ? WR / 4 : WR / 16;      // micro tiles per wave along M
    constexpr int TN = ORIENT == 0 ? WC / 16 : WC / 4;
    static_assert(BM % WM == 0 && BN % WN == 0 && BK % 4 == 0, "tile shape");
    static_assert(ORIENT == 0 ? (WR % 4 == 0 && WC % 16 == 0) : (WR % 16 == 0 && WC % 4 == 0), "wave tile shape");
    typedef TileStager<T, ALAY, BM, BK, NT, VEC> StA;
    typedef TileStager<T, BLAY == 1 ? 0 : 1, BN, BK, NT, VEC> StB;
    constexpr bool SWZ = GLDS != 0 && BLAY == 1;  // swizzled, unpadded B image (see above)
    constexpr int PA = StA::P, PB = SWZ ? BK : StB::P;
    constexpr int A_ELEMS = StA::ELEMS, B_ELEMS = SWZ ? BN * BK : StB::ELEMS;
    static_assert(!SWZ || BK == 16, "swizzled B image: 16-deep tiles");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *smem = reinterpret_cast<T *>(smem_raw);
    constexpr int STAGE = A_ELEMS + B_ELEMS;  // see k_gemm_mfma: keep LDS pointers out of arrays

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lk = lane >> 4;
    // lane's row inside an A micro tile / column inside a B micro tile
    const int la = ORIENT == 0 ? (lane & 3) : (lane & 15);
    const int lbn = ORIENT == 0 ? (lane & 15) : (lane & 3);

    const int ntiles = g.tiles_m * g.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % g.tiles_n, tile_m = bid / g.tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    const int split = blockIdx.y;
    const int64_t kbeg = (int64_t)split * g.kchunk;
    const int64_t kend = min(g.K, kbeg + g.kchunk);

    double acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = 0.0;

    StA sa;
    StB sb;
    if (GLDS < 2) sa.init(g.a, m0, kbeg, g.sam, g.sak, tid);
    if (!GLDS) sb.init(g.b, n0, kbeg, g.sbn, g.sbk, tid);
    static_assert(!GLDS || BLAY == 1 || (BN % 128 == 0 && (BK * (BN / 128)) % (NT / 64) == 0), "direct-to-LDS B tile: shape");
    static_assert(!GLDS || BLAY == 0 || (BN / 8) % (NT / 64) == 0, "direct-to-LDS B tile (K-contiguous): shape");
    static_assert(GLDS < 2 || (ALAY == 1 && BM >= 128 && BM <= 256 && BK % (NT / 64) == 0), "direct-to-LDS A tile: shape");
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto glds_b = [&](int64_t k0, T *bdst) {
        if constexpr (BLAY == 1) {
            constexpr int PER_WAVE = (BN / 8) / (NT / 64) > 0 ? (BN / 8) / (NT / 64) : 1;
#pragma unroll
            for (int i = 0; i < PER_WAVE; ++i) {
                const int nb = (wave_u * PER_WAVE + i) * 8;         // eight tile rows per 1 KiB piece
                const int nrow = nb + (lane >> 3);
                const int chunk = (lane & 7) ^ ((nrow >> 1) & 7);    // source chunk that lands at position lane & 7
                const T *src = g.b + (n0 + nrow) * g.sbn + (k0 + 2 * chunk);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(bdst + nb * BK), 16, 0, 0);
            }
            return;
        }
        constexpr int PIECES_ROW = BN >= 128 ? BN / 128 : 1, PER_WAVE = BK * PIECES_ROW / (NT / 64) > 0 ? BK * PIECES_ROW / (NT / 64) : 1;
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int piece = wave_u * PER_WAVE + i;
            const int kk = piece / PIECES_ROW, h = piece % PIECES_ROW;
            const T *src = g.b + (k0 + kk) * g.sbk + (n0 + h * 128 + 2 * lane);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(bdst + kk * PB + h * 128), 16, 0, 0);
        }
    };

    const int tail_lanes = (int)((g.M - m0 > 128 ? g.M - m0 - 128 : 0) + 1) / 2;
    auto glds_a = [&](int64_t k0, T *adst) {
        constexpr int PER_WAVE = BK / (NT / 64) > 0 ? BK / (NT / 64) : 1;
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int kk = wave_u * PER_WAVE + i;
            const T *src = g.a + (k0 + kk) * g.sak + (m0 + 2 * lane);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(adst + kk * PA), 16, 0, 0);
            if constexpr (BM > 128) {
                // columns 128 ... M - 1 (rounded up to a pair: the host checked that the pair stays inside the row)
                if (lane < tail_lanes)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 128),
                                                     (__attribute__((address_space(3))) void *)(adst + kk * PA + 128), 16, 0, 0);
            }
        }
    };

    const int64_t nk = kend > kbeg ? cdiv(kend - kbeg, BK) : 0;
    if (nk > 0) {
        if (GLDS >= 2) glds_a(kbeg, smem);
        else sa.load(m0, g.M, kbeg, kend, 0, g.sam, g.sak, tid);
        if (GLDS) glds_b(kbeg, smem + A_ELEMS);
        else sb.load(n0, g.N, kbeg, kend, 0, g.sbn, g.sbk, tid);
        if (GLDS < 2) sa.store(smem, tid);
        if (!GLDS) sb.store(smem + A_ELEMS, tid);
    }
    // copies per wave and tile (GLDS == 3 counts them in s_waitcnt vmcnt)
    constexpr int GL_PER_TILE = (BK / (NT / 64) > 0 ? BK / (NT / 64) : 1) * (BM > 128 ? 2 : 1) + (BLAY == 1 ? (BN / 8) / (NT / 64) : BK * (BN >= 128 ? BN / 128 : 1) / (NT / 64));
    if constexpr (GLDS == 3) {
        static_assert(BM <= 128, "counted waits assume no masked tail piece");
        if (nk > 1) {
            glds_a(kbeg + BK, smem + STAGE);
            glds_b(kbeg + BK, smem + A_ELEMS + STAGE);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GL_PER_TILE) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
        __syncthreads();
    }
    constexpr int AM = ORIENT == 0 ? 4 : 16, BNW = ORIENT == 0 ? 16 : 4;
#ifdef RC_GEMM_TIMING
    unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#endif
    int buf3 = 0;  // GLDS == 3: it % 3
    for (int64_t it = 0; it < nk; ++it) {
        const int buf = GLDS == 3 ? buf3 : (int)(it & 1);
        if constexpr (GLDS == 3) {
            if (it + 2 < nk) {
                const int nb = buf3 == 0 ? 2 : buf3 - 1;  // (it + 2) % 3: the buffer read in tile it - 1
                const int64_t koff = (it + 2) * BK;
                glds_a(kbeg + koff, smem + nb * STAGE);
                glds_b(kbeg + koff, smem + A_ELEMS + nb * STAGE);
            }
        } else
        if (it + 1 < nk) {
            const int64_t koff = (it + 1) * BK;
            if (GLDS >= 2) glds_a(kbeg + koff, smem + (buf ^ 1) * STAGE);
            else sa.load(m0, g.M, kbeg + koff, kend, koff * g.sak, g.sam, g.sak, tid);
            if (GLDS) glds_b(kbeg + koff, smem + A_ELEMS + (buf ^ 1) * STAGE);  // that buffer was last read in tile it - 1
            else sb.load(n0, g.N, kbeg + koff, kend, koff * g.sbk, g.sbn, g.sbk, tid);
        }
        RC_STAMP(0)
        const T *as = smem + buf * STAGE, *bs = smem + A_ELEMS + buf * STAGE;
        auto load_frags = [&](int ks, T *af, T *bf) {
            const int kk = ks * 4 + lk;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mm = wm * WR + i * AM + la;
                af[i] = ALAY == 0 ? as[mm * PA + kk] : as[kk * PA + mm];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nn = wn * WC + j * BNW + lbn;
                if (SWZ) bf[j] = bs[nn * BK + ((((kk >> 1) ^ ((nn >> 1) & 7)) << 1) | (kk & 1))];
                else bf[j] = BLAY == 1 ? bs[nn * PB + kk] : bs[kk * PB + nn];
            }
        };
        {
        // (two fragment sets with the next sub-step's reads in flight behind the MFMAs -- possible in the direct-to-LDS
        // instance, which has ~60 registers to spare -- measured 4 % SLOWER than the plain loop)
        // not unrolled: with all BK/4 sub-steps in flight the hoisted fragment loads (4 x 17 f64
        // registers) spill; one sub-step = (TM + TN) LDS reads feeding TM * TN MFMAs
#pragma unroll 1
        for (int ks = 0; ks < BK / 4; ++ks) {
            T af[TM], bf[TN];
            load_frags(ks, af, bf);
            RC_STAMP(1)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
            RC_STAMP(2)
        }
        }
        if constexpr (GLDS == 3) {
            // tile it + 1 must have landed; the copies of tile it + 2 (issued above) stay in flight
            if (it + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GL_PER_TILE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            RC_STAMP(3)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            buf3 = buf3 == 2 ? 0 : buf3 + 1;
        } else {
            if (it + 1 < nk) {
                if (GLDS < 2) sa.store(smem + (buf ^ 1) * STAGE, tid);
                if (!GLDS) sb.store(smem + A_ELEMS + (buf ^ 1) * STAGE, tid);
            }
            RC_STAMP(3)
            __syncthreads();
        }
        RC_STAMP(4)
    }
#ifdef RC_GEMM_TIMING
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && g.M * g.N * g.K > (1ll << 32)) {
        for (int k2 = 0; k2 < 5; ++k2) g_gemm_dbg[k2] = tacc[k2];
        g_gemm_dbg[5] = (unsigned long long)nk;
    }
#endif

    // ---- epilogue: D lane l = 16*i + 4*b + j ---------------------------------------------
    //   ORIENT 0: row = i (l >> 4),           col = 4*b + j (l & 15)
    //   ORIENT 1: row = 4*b + i,              col = j (l & 3)
    const int er = ORIENT == 0 ? (lane >> 4) : (((lane >> 2) & 3) * 4 + (lane >> 4));
    const int ec = ORIENT == 0 ? (lane & 15) : (lane & 3);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int64_t gm = m0 + wm * WR + i * AM + er;
            const int64_t gn = n0 + wn * WC + j * BNW + ec;
            if (gm < g.M && gn < g.N) {
                const double v = acc[i][j];
                if (g.splits > 1) {
                    g.partial[((int64_t)split * g.M + gm) * g.N + gn] = v;
                } else {
                    T *cp = g.c + gm * g.scm + gn * g.scn;
                    *cp = g.beta == 0.0 ? g.alpha * v : g.alpha * v + g.beta * (*cp);
                }
            }
        }
}

// sum over the slabs in slab order (deterministic).  Eight loads are issued before the first add: the adds are one dependent chain
// either way, but the loads of a plain loop waited for one another (32 slabs: 32 memory latencies, ~20 us for a 133 x 133 result).
template <typename T>
__device__ inline T slab_sum(const T *p, int64_t stride, int splits) {
    T s = 0;
    int sp = 0;
    for (; sp + 8 <= splits; sp += 8) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(sp + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; sp < splits; ++sp) s += p[(int64_t)sp * stride];
    return s;
}

// C = alpha * sum_s partial[s] + beta * C ; fixed summation order => deterministic
template <typename T>
__global__ __launch_bounds__(256) void k_splitk_reduce(GemmArgs<T> g) {
    const bool col_fast = (g.scn <= g.scm);
    const int64_t total = g.M * g.N;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t gm, gn;
        if (col_fast) { gm = e / g.N; gn = e - gm * g.N; }
        else { gn = e / g.M; gm = e - gn * g.M; }
        const T s = slab_sum(g.partial + gm * g.N + gn, g.M * g.N, g.splits);
        T *cp = g.c + gm * g.scm + gn * g.scn;
        *cp = g.beta == (T)0 ? g.alpha * s : g.alpha * s + g.beta * (*cp);
    }
}

// The same when C's FIRST index is its contiguous one (e.g. the row-major m x l result of the sketch, computed as the transposed
// problem): the slabs are read along their own contiguous index (n) and the sums cross a 32 x 33 LDS tile, so that both the slab
// reads and the writes of C are whole lines (the element-wise kernel above read the slabs with an 8 N byte stride there:
// 86 us instead of 15 for 133 x 8192 x 8 slabs).  Same fixed summation order.
template <typename T>
__global__ __launch_bounds__(256) void k_splitk_reduce_t(GemmArgs<T> g) {
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int64_t n0 = (int64_t)blockIdx.x * 32, m0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t gm = m0 + ty + 8 * r, gn = n0 + tx;
        T s = 0;
        if (gm < g.M && gn < g.N) s = slab_sum(g.partial + gm * g.N + gn, g.M * g.N, g.splits);
        tile[ty + 8 * r][tx] = s;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t gm = m0 + tx, gn = n0 + ty + 8 * r;  // consecutive threads: consecutive m
        if (gm < g.M && gn < g.N) {
            const T s = tile[tx][ty + 8 * r];
            T *cp = g.c + gm * g.scm + gn * g.scn;
            *cp = g.beta == (T)0 ? g.alpha * s : g.alpha * s + g.beta * (*cp);
        }
    }
}

static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

// target_wgs / min_ksteps (0 = the defaults below): a bandwidth-bound product asks for more, shorter workgroups
template <typename T, int ALAY, int BLAY, int BM, int BN, int BK, int WM, int WN, int VEC, int NBUF = 2>
static void launch_cfg(rc_context *c, GemmArgs<T> g, int target_wgs = 0, int min_ksteps = 0) {
    constexpr int NT = WM * WN * 64;
    typedef TileStager<T, ALAY, BM, BK, NT, VEC> StA;
    typedef TileStager<T, BLAY == 1 ? 0 : 1, BN, BK, NT, VEC> StB;
    constexpr size_t lds = NBUF * (size_t)(StA::ELEMS + StB::ELEMS) * sizeof(T);
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    g.tiles_m = (int)cdiv(g.M, BM);
    g.tiles_n = (int)cdiv(g.N, BN);
    const int64_t tiles = (int64_t)g.tiles_m * g.tiles_n;
    // Split K until the grid covers the 256 CUs about twice (only worth it for deep K; tiny
    // outputs with a deep reduction -- the n x n Gram matrices of the CholeskyQR passes --
    // need up to 128 slabs to reach every CU).
    // Split K until the grid covers the chip; a product with very few output tiles (the n x n Gram
    // matrices of the CholeskyQR passes: ONE tile, K = 8192) stops at 32 slabs -- 16 K-tiles per workgroup
    // amortise its prologue / slab write, and the reduction reads 4x less than with 128 slabs
    static const int target_big = env_int("RC_GEMM_TARGET_WGS", 256), target_small = env_int("RC_GEMM_SMALL_TARGET", 32);
    // with many compressions in flight (RC_OPT_CONCURRENCY_HINT) the other streams fill the chip: wide products stay un-split
    const int target = target_wgs > 0 ? target_wgs : tiles >= 8 ? (c->opt_lanes >= 8 && tiles >= 32 ? 1 : target_big) : target_small;
    int splits = 1;
    const int64_t ksteps = cdiv(g.K, BK);
    const int min_ks = min_ksteps > 0 ? min_ksteps : (tiles >= 32 ? 16 : 4);
    while (tiles * splits < target && splits < 128 && ksteps / (splits * 2) >= min_ks) splits *= 2;
    g.kchunk = cdiv(cdiv(g.K, splits), BK) * BK;
    splits = (int)cdiv(g.K, g.kchunk);
    if (splits < 1) splits = 1;
    g.splits = splits;
    ArenaMark mark(c);
    if (splits > 1) g.partial = c->alloc<T>((size_t)splits * g.M * g.N);
    auto kern = k_gemm_mfma<T, ALAY, BLAY, BM, BN, BK, WM, WN, VEC, NBUF>;
    static bool attr_set[64] = {};
    if (lds > 48 * 1024 && !attr_set[c->device & 63]) {
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[c->device & 63] = true;
    }
    {
        char nm[128];
        snprintf(nm, sizeof(nm), "k_gemm_mfma<%s,%d,%d,%d,%d,%d,%d,%d,%d,%d>", sizeof(T) == 8 ? "double" : "float", ALAY, BLAY, BM, BN, BK, WM, WN, VEC, NBUF);
        c->last_gemm_kernel = nm;
    }
    {
        ProfScope ps(c, "kernel:k_gemm_mfma<%s> M=%lld N=%lld K=%lld", sizeof(T) == 8 ? "f64" : "f32", (long long)g.M, (long long)g.N, (long long)g.K);
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)splits), dim3(NT), lds, c->stream, g);
    }
    if (splits > 1) {
        ProfScope ps(c, "kernel:k_splitk_reduce M=%lld N=%lld splits=%d", (long long)g.M, (long long)g.N, splits);
        if (g.scm < g.scn && g.N >= 32) {
            hipLaunchKernelGGL(k_splitk_reduce_t<T>, dim3((unsigned)cdiv(g.N, 32), (unsigned)cdiv(g.M, 32)), dim3(256), 0, c->stream, g);
        } else {
            int grid = (int)std::min<int64_t>(cdiv(g.M * g.N, 256), 4096);
            hipLaunchKernelGGL(k_splitk_reduce<T>, dim3(grid), dim3(256), 0, c->stream, g);
        }
    }
}

template <int ALAY, int BLAY, int BM, int BN, int BK, int WM, int WN, int VEC, int ORIENT, int GLDS = 0>
static void launch_f64q(rc_context *c, GemmArgs<double> g, int target_wgs = 0, int min_ksteps = 0) {
    typedef double T;
    constexpr int NT = WM * WN * 64;
    typedef TileStager<T, ALAY, BM, BK, NT, VEC> StA;
    typedef TileStager<T, BLAY == 1 ? 0 : 1, BN, BK, NT, VEC> StB;
    constexpr size_t lds = (GLDS == 3 ? 3 : 2) * (size_t)(StA::ELEMS + (GLDS != 0 && BLAY == 1 ? BN * BK : StB::ELEMS)) * sizeof(T);
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    g.tiles_m = (int)cdiv(g.M, BM);
    g.tiles_n = (int)cdiv(g.N, BN);
    const int64_t tiles = (int64_t)g.tiles_m * g.tiles_n;
    // Split K until the grid covers the chip; a product with very few output tiles (the n x n Gram
    // matrices of the CholeskyQR passes: ONE tile, K = 8192) stops at 32 slabs -- 16 K-tiles per workgroup
    // amortise its prologue / slab write, and the reduction reads 4x less than with 128 slabs
    static const int target_big = env_int("RC_GEMM_TARGET_WGS", 256), target_small = env_int("RC_GEMM_SMALL_TARGET", 32);
    // with many compressions in flight (RC_OPT_CONCURRENCY_HINT) the other streams fill the chip: wide products stay un-split
    // (RC_GEMM_LANES_TARGET=64: ONE K split in flight -- CUs come free twice as often, which shortens the cooperative kernels' wait for
    // co-residency: 1081 against 1073 compressions/s on average over five A/B pairs, inside the run-to-run spread; 128 / 256: neutral /
    // lower.  The default stays un-split.)
    static const int target_lanes = env_int("RC_GEMM_LANES_TARGET", 1);
    const int target = tiles >= 8 ? (c->opt_lanes >= 8 && tiles >= 32 ? target_lanes : (target_wgs > 0 ? target_wgs : target_big)) : target_small;
    int splits = 1;
    const int64_t ksteps = cdiv(g.K, BK);
    while (tiles * splits < target && splits < 128 && ksteps / (splits * 2) >= (min_ksteps > 0 ? min_ksteps : (tiles >= 32 ? 16 : 4))) splits *= 2;
    g.kchunk = cdiv(cdiv(g.K, splits), BK) * BK;
    splits = (int)cdiv(g.K, g.kchunk);
    if (splits < 1) splits = 1;
    g.splits = splits;
    ArenaMark mark(c);
    if (splits > 1) g.partial = c->alloc<T>((size_t)splits * g.M * g.N);
    auto kern = k_gemm_f64q<ALAY, BLAY, BM, BN, BK, WM, WN, VEC, ORIENT, GLDS>;
    static bool attr_set[64] = {};
    if (lds > 48 * 1024 && !attr_set[c->device & 63]) {
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[c->device & 63] = true;
    }
    {
        char nm[128];
        snprintf(nm, sizeof(nm), "k_gemm_f64q<%d,%d,%d,%d,%d,%d,%d,%d,%d,%d>", ALAY, BLAY, BM, BN, BK, WM, WN, VEC, ORIENT, GLDS);
        c->last_gemm_kernel = nm;
    }
    // the software-pipelined main loop (kernels_gemm_pipe.hip) where it has an instantiation for this tile shape
    if (!gemm_f64p_launch(c, g, ALAY, BLAY, BM, BN, BK, WM, WN, ORIENT, VEC == 2)) {
        ProfScope ps(c, "kernel:k_gemm_mfma<f64> M=%lld N=%lld K=%lld", (long long)g.M, (long long)g.N, (long long)g.K);
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)splits), dim3(NT), lds, c->stream, g);
    }
    if (splits > 1) {
        ProfScope ps(c, "kernel:k_splitk_reduce M=%lld N=%lld splits=%d", (long long)g.M, (long long)g.N, splits);
        if (g.scm < g.scn && g.N >= 32) {
            hipLaunchKernelGGL(k_splitk_reduce_t<T>, dim3((unsigned)cdiv(g.N, 32), (unsigned)cdiv(g.M, 32)), dim3(256), 0, c->stream, g);
        } else {
            int grid = (int)std::min<int64_t>(cdiv(g.M * g.N, 256), 4096);
            hipLaunchKernelGGL(k_splitk_reduce<T>, dim3(grid), dim3(256), 0, c->stream, g);
        }
    }
}

// f64 shapes on the 4x4x4 MFMA (RC_GEMM_F64X4=0 falls back to the 16x16x4 kernel)
template <int ALAY, int BLAY, int VEC>
static bool launch_shape_f64q(rc_context *c, const GemmArgs<double> &g) {
    static const int use = env_int("RC_GEMM_F64X4", 1);
    if (!use) return false;
    // Skinny shapes: the narrow dimension is padded to the micro-tile granularity only -- 16 along the
    // "shared" operand's side, 4 along the other (ORIENT picks which) -- so N = 133 costs 136, M = 128 costs 128.
    // (N = 133 through ORIENT 1 / BN = 136 measured slower than BN = 144: 34 B-fragment reads per sub-step.)
    static const int vn = env_int("RC_GEMM_F64Q_N", 3), vm = env_int("RC_GEMM_F64Q_M", 3), smallk = env_int("RC_GEMM_SMALLK", 1), two_wg = env_int("RC_GEMM_SKETCH_2WG", 0), m32 = env_int("RC_GEMM_F64_M32", 1);
    if (g.M <= 144 && g.N <= 144 && g.M > 80 && g.N > 80) launch_f64q<ALAY, BLAY, 144, 144, 16, 3, 3, VEC, 0>(c, g);
    else if (g.N <= 80) launch_f64q<ALAY, BLAY, 256, 80, 16, 8, 1, VEC, 0>(c, g);
    else if (g.N <= 128 && vn == 3) launch_f64q<ALAY, BLAY, 256, 128, 16, 8, 1, VEC, 0>(c, g);
    else if (g.N <= 144) {
        if (vn == 1) launch_f64q<ALAY, BLAY, 128, 144, 16, 8, 1, VEC, 0>(c, g);
        else if (vn == 2) launch_f64q<ALAY, BLAY, 128, 144, 16, 4, 1, VEC, 0>(c, g);
        else launch_f64q<ALAY, BLAY, 256, 144, 16, 8, 1, VEC, 0>(c, g);
    }
    else if (g.M <= 32 && g.K >= 512 && m32) {
        // the HBM-bound 32-row products of the blocked QRCP's panel ends (Y = V^T A, V^T Q): 32-row tiles instead of 80 and many
        // short workgroups, as for f32 (launch_shape)
        launch_f64q<ALAY, BLAY, 32, 256, 16, 1, 8, VEC, 1>(c, g, 512, 4);
    }
    else if (g.M <= 80) launch_f64q<ALAY, BLAY, 80, 256, 16, 1, 8, VEC, 1>(c, g);
    else if (g.M <= 136 && g.K <= 160 && smallk && g.N >= 2048) {
        // shallow products on the dependent chain of a compression (apply R^-1, Q1 Q2, Q U_b, ...: K = 128 / 133): 128-wide
        // tiles = twice the workgroups of the 256-wide instance, for latency rather than efficiency
        if (g.M <= 128) launch_f64q<ALAY, BLAY, 128, 128, 16, 1, 8, VEC, 1>(c, g);
        else launch_f64q<ALAY, BLAY, 136, 128, 16, 2, 4, VEC, 0>(c, g);
    }
    else if (g.M <= 128 && g.M > 80 && vm == 3 && ALAY == 1 && BLAY == 0 && VEC == 2 && two_wg && g.K % 16 == 0 && g.sam == 1 && g.sbn == 1 && g.N >= 1024) {
        // the projection on two 4-wave workgroups per CU (128 x 128 tiles, 128 x 32 wave tiles; k_gemm_f64a)
        launch_f64q<ALAY, BLAY, 128, 128, 16, 1, 4, VEC, 1>(c, g, 512);
    }
    else if (g.M <= 128 && vm == 3) {
        static const int glds = env_int("RC_GEMM_GLDS", 2);  // 0: register staging, 1: B tile direct to LDS, 2: A tile too, 3: + three LDS buffers (no gain measured)
        const bool direct = glds && BLAY == 0 && VEC == 2 && g.N % 256 == 0 && g.K % 16 == 0 && g.sbn == 1 && g.sbk % 2 == 0 &&
                            reinterpret_cast<uintptr_t>(g.b) % 16 == 0;
        if constexpr (BLAY == 0 && VEC == 2) {
            if constexpr (ALAY == 1) {
                const bool direct_a = glds >= 2 && g.M == 128 && g.sam == 1 && g.sak % 2 == 0 && reinterpret_cast<uintptr_t>(g.a) % 16 == 0;
                if (direct && direct_a && glds >= 3) { launch_f64q<ALAY, BLAY, 128, 256, 16, 1, 8, VEC, 1, 3>(c, g); return true; }
                if (direct && direct_a) { launch_f64q<ALAY, BLAY, 128, 256, 16, 1, 8, VEC, 1, 2>(c, g); return true; }
            }
            if (direct) { launch_f64q<ALAY, BLAY, 128, 256, 16, 1, 8, VEC, 1, 1>(c, g); return true; }
        }
        launch_f64q<ALAY, BLAY, 128, 256, 16, 1, 8, VEC, 1>(c, g);
    }
    else if (g.M <= 136 && g.M > 128 && vm == 3 && ALAY == 1 && BLAY == 1 && VEC == 2 && two_wg && g.N % 128 == 0 && g.K % 16 == 0 && g.sam == 1 && g.sbk == 1) {
        // the sketch on two 4-wave workgroups per CU (136 x 128 tiles, the same 68 x 64 wave tiles; k_gemm_f64a): each one's
        // barrier and copy issue hide behind the other's MFMAs; a lone product is split until 512 workgroups exist
        launch_f64q<ALAY, BLAY, 136, 128, 16, 2, 2, VEC, 0>(c, g, 512);
    }
    else if (g.M <= 136 && vm == 3) {  // 68 x 64 wave tiles: 17 + 4 fragment reads per 68 MFMAs
        static const int glds = env_int("RC_GEMM_GLDS", 2);
        if constexpr (ALAY == 1 && BLAY == 1 && VEC == 2) {
            // the sketch (as the transposed problem): A' = Omega^T rows of M doubles, B' = A^T with unit K stride
            const bool direct = glds >= 2 && g.M > 128 && g.N % 256 == 0 && g.K % 16 == 0 && g.sbk == 1 && g.sbn % 2 == 0 &&
                                reinterpret_cast<uintptr_t>(g.b) % 16 == 0 && g.sam == 1 && g.sak % 2 == 0 && g.sak >= 128 + 2 * ((g.M - 128 + 1) / 2) &&
                                reinterpret_cast<uintptr_t>(g.a) % 16 == 0;
            static const int gs = env_int("RC_GEMM_GLDS_SKETCH", 0);  // measured: 1 (B direct) and 2 (A and B direct) are 1-2 % slower here
            if (direct && gs == 2) { launch_f64q<ALAY, BLAY, 136, 256, 16, 2, 4, VEC, 0, 2>(c, g); return true; }
            if (direct && gs == 1) { launch_f64q<ALAY, BLAY, 136, 256, 16, 2, 4, VEC, 0, 1>(c, g); return true; }
        }
        launch_f64q<ALAY, BLAY, 136, 256, 16, 2, 4, VEC, 0>(c, g);
    }
    else if (g.M <= 144) {
        if (vm == 1) launch_f64q<ALAY, BLAY, 144, 128, 16, 1, 8, VEC, 1>(c, g);
        else if (vm == 2) launch_f64q<ALAY, BLAY, 144, 128, 16, 1, 4, VEC, 1>(c, g);
        else launch_f64q<ALAY, BLAY, 144, 256, 16, 1, 8, VEC, 1>(c, g);
    }
    else launch_f64q<ALAY, BLAY, 128, 128, 16, 2, 2, VEC, 0>(c, g);
    return true;
}
template <typename T, int ALAY, int BLAY, int VEC>
struct F64Q { static bool run(rc_context *, const GemmArgs<T> &) { return false; } };
template <int ALAY, int BLAY, int VEC>
struct F64Q<double, ALAY, BLAY, VEC> { static bool run(rc_context *c, const GemmArgs<double> &g) { return launch_shape_f64q<ALAY, BLAY, VEC>(c, g); } };

template <typename T, int ALAY, int BLAY, int VEC>
static void launch_shape(rc_context *c, const GemmArgs<T> &g) {
    if (F64Q<T, ALAY, BLAY, VEC>::run(c, g)) return;
    // Skinny outputs (the sketch Y = A Omega has N = k + p ~ 69..133; the range
    // projection B = Q^H A has M = k ~ 64..128) get tiles that cover the short
    // side once, so the long operand streams from HBM exactly once.
    static const int vn = env_int("RC_GEMM_SKINNY_N", 0), vm = env_int("RC_GEMM_SKINNY_M", 0), m32 = env_int("RC_GEMM_F32_M32", 2);
    if (g.M <= 144 && g.N <= 144 && g.M > 80 && g.N > 80) launch_cfg<T, ALAY, BLAY, 144, 144, 16, 3, 3, VEC>(c, g);
    else if (g.N <= 80) launch_cfg<T, ALAY, BLAY, 128, 80, 16, 4, 1, VEC>(c, g);
    else if (g.N <= 144) {
        if (vn == 1) launch_cfg<T, ALAY, BLAY, 128, 144, 16, 4, 1, VEC>(c, g);
        else launch_cfg<T, ALAY, BLAY, 256, 144, 16, 8, 1, VEC>(c, g);
    } else if (g.M <= 32 && sizeof(T) == 4 && ALAY == 0 && BLAY == 1 && VEC == 2 && g.K >= 512 && m32) {
        // Y = V^T A of the blocked QRCP (32 x n x m, both operands K-contiguous, HBM-bound at 32 flops per element of A):
        // 32-row tiles instead of 80 and many short workgroups (16 K slabs instead of 8) so that enough loads are in flight;
        // measured on 32 x 4096 x 4096 f32: 76 + 6 us (80-row tiles, 8 slabs) -> 37 + 9 us.  (RC_GEMM_F32_M32=1: 32-deep K tiles,
        // every row of a tile a whole 128-byte line: 43 + 8 us.)
        if constexpr (sizeof(T) == 4 && ALAY == 0 && BLAY == 1 && VEC == 2) {
            // HBM-bound (32 flops per element of A): many short workgroups keep enough loads in flight
            static const int tw = env_int("RC_GEMM_F32_M32_WGS", 512), mk = env_int("RC_GEMM_F32_M32_MINK", 4);
            if (m32 == 2) launch_cfg<T, ALAY, BLAY, 32, 128, 16, 1, 4, VEC>(c, g, tw, mk);
            else launch_cfg<T, ALAY, BLAY, 32, 128, 32, 1, 4, VEC>(c, g, tw, mk);
        }
    } else if (g.M <= 80) launch_cfg<T, ALAY, BLAY, 80, 128, 16, 1, 4, VEC>(c, g);
    else if (g.M <= 144) {
        if (vm == 1) launch_cfg<T, ALAY, BLAY, 144, 128, 16, 1, 4, VEC>(c, g);
        else launch_cfg<T, ALAY, BLAY, 144, 256, 16, 1, 8, VEC>(c, g);
    } else launch_cfg<T, ALAY, BLAY, 128, 128, 16, 2, 2, VEC>(c, g);
}

template <typename T, int ALAY, int BLAY>
static void launch_vec(rc_context *c, const GemmArgs<T> &g) {
    // 2-element vector staging needs the contiguous stride to be 1, the other stride even and
    // the base 2-element aligned, for BOTH operands (odd leading dimensions take the scalar path)
    auto ok = [](const T *p, int64_t s_contig, int64_t s_other) {
        return s_contig == 1 && (s_other % 2) == 0 && (reinterpret_cast<uintptr_t>(p) % (2 * sizeof(T))) == 0;
    };
    const bool va = ALAY == 0 ? ok(g.a, g.sak, g.sam) : ok(g.a, g.sam, g.sak);
    const bool vb = BLAY == 0 ? ok(g.b, g.sbn, g.sbk) : ok(g.b, g.sbk, g.sbn);
    static const int allow = env_int("RC_GEMM_VEC", 1);
    if (va && vb && allow) launch_shape<T, ALAY, BLAY, 2>(c, g);
    else launch_shape<T, ALAY, BLAY, 1>(c, g);
}

template <typename T>
void gemm(rc_context *c, T alpha, Mat<T> a, Mat<T> b, T beta, Mat<T> cm) {
    RC_REQUIRE(a.cols == b.rows && a.rows == cm.rows && b.cols == cm.cols, RC_INVALID_ARGUMENT,
               "gemm: shapes (%lld x %lld) * (%lld x %lld) -> (%lld x %lld)", (long long)a.rows, (long long)a.cols,
               (long long)b.rows, (long long)b.cols, (long long)cm.rows, (long long)cm.cols);
    if (cm.empty()) return;
    ArenaMark mark(c);
    if (a.cols == 0) {  // empty inner dimension: C = beta * C
        RC_REQUIRE(beta == (T)0, RC_INVALID_ARGUMENT, "gemm: K == 0 with beta != 0 unsupported");
        fill_zero(c, cm);
        return;
    }
    // operands whose neither stride is 1 are packed once (never on the hot path)
    if (a.rs != 1 && a.cs != 1) {
        T *p = c->alloc<T>((size_t)a.rows * a.cols);
        Mat<T> pa = rowmajor(p, a.rows, a.cols, a.cols);
        copy_mat(c, a, pa);
        a = pa;
    }
    if (b.rs != 1 && b.cs != 1) {
        T *p = c->alloc<T>((size_t)b.rows * b.cols);
        Mat<T> pb = rowmajor(p, b.rows, b.cols, b.cols);
        copy_mat(c, b, pb);
        b = pb;
    }
    // f64, skinny N under a long M (the sketch Y = A Omega): run the transposed problem C^T = B^T A^T, so the
    // narrow side becomes M, whose micro-tile granularity is 4 (N = 133 costs 136 instead of 144) and whose
    // tile shapes measured faster; only the view descriptors change
    static const int swap_ok = env_int("RC_GEMM_SWAP_SKINNY", 1);
    if (sizeof(T) == 8 && swap_ok && b.cols <= 144 && a.rows > 144) {
        const Mat<T> a2 = b.t(), b2 = a.t();
        a = a2;
        b = b2;
        cm = cm.t();
    }
    GemmArgs<T> g;
    g.a = a.p; g.b = b.p; g.c = cm.p;
    g.M = a.rows; g.N = b.cols; g.K = a.cols;
    g.sam = a.rs; g.sak = a.cs; g.sbk = b.rs; g.sbn = b.cs; g.scm = cm.rs; g.scn = cm.cs;
    g.alpha = alpha; g.beta = beta;
    g.kchunk = g.K; g.splits = 1; g.partial = nullptr; g.tiles_m = g.tiles_n = 0;
    const bool a_kc = (a.cs == 1);
    const bool b_nc = (b.cs == 1);
    if (a_kc) {
        if (b_nc) launch_vec<T, 0, 0>(c, g); else launch_vec<T, 0, 1>(c, g);
    } else {
        if (b_nc) launch_vec<T, 1, 0>(c, g); else launch_vec<T, 1, 1>(c, g);
    }
}

#ifdef RC_GEMM_TIMING
extern "C" void rc_debug_gemm_timing(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_dbg), 8 * sizeof(unsigned long long)); }
#endif
template void gemm<double>(rc_context *, double, Mat<double>, Mat<double>, double, Mat<double>);
template void gemm<float>(rc_context *, float, Mat<float>, Mat<float>, float, Mat<float>);

}  // namespace rc
