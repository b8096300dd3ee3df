// Software-pipelined f64 MFMA GEMM main loop for gfx950 (the two big products of a compression: the sketch Y = A Omega and
// the projection B = Q^H A; they replace the reference's per-column gemv loops, /root/reference/src/types.rs:60-70, :90-100).
//
// Same tiles, LDS images and v_mfma_f64_4x4x4_4b micro-tiling as k_gemm_f64q (kernels_gemm.hip), different schedule.  In
// k_gemm_f64q every K sub-step (4 of them per 16-deep K tile) first issues its TM + TN fragment reads and waits for ALL of
// them before the first MFMA: with two waves per SIMD running in step, the matrix pipe idles for one LDS round trip per
// sub-step, and once more per tile around the LDS stores and the barrier (PMC: matrix pipe busy 73 %).  Here
//   * the A fragments ROTATE: the read of sub-step s + 1's fragment i is issued right behind the TN MFMAs that consumed
//     sub-step s's fragment i, into the same register; the B fragments are double buffered (TN more registers).  Every
//     fragment read has a whole sub-step of MFMAs (> 1000 cycles) to arrive, the waits are counted (s_waitcnt lgkmcnt(n));
//   * the ONE barrier per K tile sits between the last two sub-steps: the next tile's LDS stores are issued at the start
//     of sub-step NKS - 2, the barrier follows that sub-step's MFMAs, and sub-step NKS - 1 already prefetches the first
//     fragments of the next tile from the other buffer.  No fragment read, LDS store or global load is waited for with an
//     empty matrix pipe;
//   * the global loads are branch free (rows beyond the matrix are clamped to its last row / row pair: they only feed output
//     rows that are never stored; the K range must be whole tiles, which the host checks), so a K tile is ONE basic block
//     whose order is pinned with sched_group_barrier.
#include "rc_gemm.hpp"

#include <type_traits>
#include <utility>

namespace rc {

typedef double pdouble2_t __attribute__((ext_vector_type(2)));
typedef unsigned int pint4_t __attribute__((ext_vector_type(4)));

constexpr int pp_pitch16(int n) { return n + ((16 - n % 32) + 32) % 32; }

// One operand tile: R rows x BK, staged by NT threads with 16-byte vectors.
//   LAY == 0: the reduction index is contiguous in memory ; LAY == 1: the row index is contiguous in memory.
// LDS image as in TileStager (kernels_gemm.hip): LAY == 0 -> [R][BK + 2], LAY == 1 -> [BK][pitch16(R)].
// Slots beyond the tile's NVEC vectors duplicate the vectors of other threads (same address, same value) instead of being
// predicated off, so the staging code has no branches.
template <int LAY, int R, int BK, int NT>
struct PipeStage {
    static constexpr int P = LAY == 0 ? BK + 2 : pp_pitch16(R);
    static constexpr int ELEMS = LAY == 0 ? R * P : BK * P;
    static constexpr int NVEC = R * BK / 2;
    static constexpr int PER_T = (NVEC + NT - 1) / NT;
    static_assert(PER_T * NT <= 2 * NVEC && R % 2 == 0 && BK % 2 == 0, "staging shape");

    uint32_t goff[PER_T];  // byte offset of the vector from the tile's first element (buffer load: lane offset + scalar tile offset)
    pint4_t val[PER_T];

    static __device__ inline void coords(int idx, int &r, int &k) {
        if (idx >= NVEC) idx -= NVEC;
        if (LAY == 0) { k = (idx % (BK / 2)) * 2; r = idx / (BK / 2); }
        else { r = (idx % (R / 2)) * 2; k = idx / (R / 2); }
    }
    // rows_left = rows of the matrix from the tile's first row on (>= 1)
    __device__ inline void init(int64_t rows_left, int64_t sr, int64_t sk, int tid) {
        const int rclamp = (int)min((int64_t)R, rows_left) - 1;
#pragma unroll
        for (int e = 0; e < PER_T; ++e) {
            int r, k;
            coords(tid + e * NT, r, k);
            if (LAY == 0) r = min(r, rclamp);
            else r = min(r, rclamp & ~1);
            goff[e] = (uint32_t)((r * sr + k * sk) * 8);
        }
    }
    // tile_off: byte offset of the K tile from the buffer's base (uniform)
    __device__ inline void load(__amdgpu_buffer_rsrc_t rsrc, uint32_t tile_off) {
#pragma unroll
        for (int e = 0; e < PER_T; ++e) val[e] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[e], tile_off, 0);
    }
    __device__ inline void store(double *lds, int tid) const {
#pragma unroll
        for (int e = 0; e < PER_T; ++e) {
            int r, k;
            coords(tid + e * NT, r, k);
            *reinterpret_cast<pint4_t *>(lds + (LAY == 0 ? r * P + k : k * P + r)) = val[e];
        }
    }
};

// ALAY: 0 = A is K-contiguous, 1 = A is M-contiguous ; BLAY: 0 = B is N-contiguous, 1 = B is K-contiguous
// ORIENT 0: A micro tile 4 rows, B micro tile 16 columns ; ORIENT 1: 16 rows / 4 columns (see k_gemm_f64q)
template <int ALAY, int BLAY, int BM, int BN, int BK, int WM, int WN, int ORIENT, int DBG = 0>
__global__ __launch_bounds__(WM *WN * 64) void k_gemm_f64p(GemmArgs<double> g) {
    constexpr int NT = WM * WN * 64;
    constexpr int WR = BM / WM, WC = BN / WN;
    constexpr int AM = ORIENT == 0 ? 4 : 16, BNW = ORIENT == 0 ? 16 : 4;
    constexpr int TM = WR / AM, TN = WC / BNW;
    constexpr int LA = ALAY, LB = BLAY == 1 ? 0 : 1;
    constexpr int NKS = BK / 4;
    static_assert(WR % AM == 0 && WC % BNW == 0 && NKS >= 2, "tile shape");
    typedef PipeStage<LA, BM, BK, NT> SA;
    typedef PipeStage<LB, BN, BK, NT> SB;
    constexpr int PA = SA::P, PB = SB::P, A_ELEMS = SA::ELEMS, STAGE = SA::ELEMS + SB::ELEMS;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *smem = reinterpret_cast<double *>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lk = lane >> 4;
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
    const int nk = (int)((kend - kbeg) / BK);  // whole tiles (host-checked), >= 1

    double acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = 0.0;

    SA sa;
    SB sb;
    sa.init(g.M - m0, g.sam, g.sak, tid);
    sb.init(g.N - n0, g.sbn, g.sbk, tid);
    // buffer resources based at this workgroup's first tile: lane offsets and tile offsets are 32-bit (host-checked spans)
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(g.a + m0 * g.sam + kbeg * g.sak), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(g.b + n0 * g.sbn + kbeg * g.sbk), 0, 0x7fffffff, 0x00020000);
    const uint32_t a_step = (uint32_t)(BK * g.sak * 8), b_step = (uint32_t)(BK * g.sbk * 8);

    // fragment addresses of this lane inside a stage (doubles)
    const int a_lane = LA == 1 ? lk * PA + wm * WR + la : (wm * WR + la) * PA + lk;
    const int b_lane = LB == 0 ? (wn * WC + lbn) * PB + lk : lk * PB + wn * WC + lbn;
    auto a_frag = [&](const double *st, int ks, int i) -> double {
        return st[a_lane + (LA == 1 ? ks * 4 * PA + i * AM : i * AM * PA + ks * 4)];
    };
    auto b_frag = [&](const double *st, int ks, int j) -> double {
        return st[A_ELEMS + b_lane + (LB == 0 ? j * BNW * PB + ks * 4 : ks * 4 * PB + j * BNW)];
    };

    // ---- prologue: tile 0 into buffer 0, tile 1 into the staging registers, fragments of (tile 0, sub-step 0) ------------
    sa.load(a_rsrc, 0);
    sb.load(b_rsrc, 0);
    sa.store(smem, tid);
    sb.store(smem + A_ELEMS, tid);
    {
        const int t1 = min(1, nk - 1);
        sa.load(a_rsrc, t1 * a_step);
        sb.load(b_rsrc, t1 * b_step);
    }
    __syncthreads();
    double af[TM], bf[2][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = b_frag(smem, 0, j);
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = a_frag(smem, 0, i);

    for (int it = 0; it < nk; ++it) {
        const double *cur = smem + (it & 1) * STAGE;
        double *nxt = smem + ((it & 1) ^ 1) * STAGE;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks == NKS - 2 && !(DBG & 1)) {
                // tile it + 1 (in the staging registers since the previous barrier) -> the other buffer; every wave finished
                // reading that buffer before the previous barrier
                sa.store(nxt, tid);
                sb.store(nxt + A_ELEMS, tid);
                __builtin_amdgcn_sched_barrier(0);
            }
            // the fragments of the next sub-step: same buffer, or sub-step 0 of the other buffer behind the barrier
            const double *fs = ks + 1 < NKS ? cur : nxt;
            const int nks = ks + 1 < NKS ? ks + 1 : 0;
            const int pb = ks & 1, qb = pb ^ 1;
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[qb][j] = b_frag(fs, nks, j);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[i], bf[pb][j], acc[i][j], 0, 0, 0);
                af[i] = a_frag(fs, nks, i);
            }
            // pin the order above: TN reads, then TM x (TN MFMAs, 1 read)
            __builtin_amdgcn_sched_group_barrier(0x100, TN, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, TN, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ks == NKS - 2) {
                // the LDS stores above have landed and this wave's reads of `cur` are complete
                if (!(DBG & 2)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const int t2 = min(it + 2, nk - 1);
                if (!(DBG & 1)) {
                    sa.load(a_rsrc, t2 * a_step);
                    sb.load(b_rsrc, t2 * b_step);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue (k_gemm_f64q's): D lane l = 16*i + 4*b + j -----------------------------------------------------------
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
                    double *cp = g.c + gm * g.scm + gn * g.scn;
                    *cp = g.beta == 0.0 ? g.alpha * v : g.alpha * v + g.beta * (*cp);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// Direct-to-LDS variant: the operand tiles go global -> LDS with buffer_load_dwordx4 ... lds (no staging registers, no LDS
// store instructions), one 1 KiB piece per wave instruction.  The copies of tile it + 2 are issued right behind the
// barrier of tile it and are waited for (vmcnt(0)) only in front of the barrier of tile it + 1: a whole tile in flight.
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void *lds_ptr_t;

// Operand stored with its ROW index contiguous (unit row stride, `sk` between reduction indices): LDS image [BK][P],
// one piece = 128 consecutive rows of one reduction index.  R <= 128 -> P = 144; R <= 256 -> P = 272 (the second piece
// of a row is always copied whole -- 1 KiB -- so the pitch covers 256 rows; lanes beyond the matrix re-read its last pair).
template <int R, int BK, int NT, bool MASKTAIL = false>
struct DirectRows {
    // MASKTAIL (tiles of 128 < R <= 144 rows): the rows beyond the first piece go in a piece that only the lanes carrying them
    // copy, so the pitch is pitch16(R) = 144 instead of 272 (18 KB per stage instead of 35) and the copy moves 64 bytes instead
    // of 1 KiB per reduction index
    static constexpr int PIECES_ROW = MASKTAIL ? 1 : (R + 127) / 128;
    static constexpr int TAIL = MASKTAIL ? R - 128 : 0;
    static constexpr int P = MASKTAIL ? pp_pitch16(R) : (R <= 128 ? pp_pitch16(128) : pp_pitch16(256));
    static constexpr int ELEMS = BK * P;
    static constexpr int NW = NT / 64;
    static constexpr int PER_WAVE = BK * PIECES_ROW / NW;
    static_assert(BK * PIECES_ROW % NW == 0 && PER_WAVE >= 1 && R <= 256 && (!MASKTAIL || (R > 128 && R <= 144 && TAIL % 2 == 0)), "direct-to-LDS tile (row-contiguous operand): shape");
    uint32_t voff[PIECES_ROW + (MASKTAIL ? 1 : 0)];
    uint32_t sk8;
    __device__ inline void init(int64_t rows_left, int64_t sk, int lane) {
        const int rclamp = ((int)min((int64_t)R, rows_left) - 1) & ~1;
#pragma unroll
        for (int h = 0; h < PIECES_ROW + (MASKTAIL ? 1 : 0); ++h) voff[h] = (uint32_t)(min(h * 128 + 2 * lane, rclamp) * 8);
        sk8 = (uint32_t)(sk * 8);
    }
    __device__ inline void copy(__amdgpu_buffer_rsrc_t rsrc, uint32_t tile_off, double *stage, int wave_u, int lane = 0) const {
        static_assert(PER_WAVE % PIECES_ROW == 0, "a wave copies whole rows");
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int kk = wave_u * (PER_WAVE / PIECES_ROW) + i / PIECES_ROW;  // uniform
            const int h = i % PIECES_ROW;                                       // compile time
            const uint32_t soff = __builtin_amdgcn_readfirstlane(tile_off + kk * sk8);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(stage + kk * P + h * 128), 16, voff[h], soff, 0, 0);
            if constexpr (MASKTAIL) {
                if (lane < TAIL / 2)  // a lane's 16 bytes land at lane * 16 behind the piece's base
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(stage + kk * P + 128), 16, voff[1], soff, 0, 0);
            }
        }
    }
    // iteration I of copy() alone (one unit = one piece, plus its masked tail piece): k_gemm_f64r spreads a tile's copies over its MFMA stream
    static constexpr int UNITS = PER_WAVE;
    // (I is a literal at every call site; a plain argument instead of a template parameter: clang 19 rejected the member template's
    // substitution in the second and later instantiations of the calling kernel template)
    __device__ __attribute__((always_inline)) void copy_unit(const int I, __amdgpu_buffer_rsrc_t rsrc, uint32_t tile_off, double *stage, int wave_u, int lane) const {
        const int kk = wave_u * (PER_WAVE / PIECES_ROW) + I / PIECES_ROW;
        const int h = I % PIECES_ROW;
        const uint32_t soff = __builtin_amdgcn_readfirstlane(tile_off + kk * sk8);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(stage + kk * P + h * 128), 16, voff[h], soff, 0, 0);
        if constexpr (MASKTAIL) {
            if (lane < TAIL / 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(stage + kk * P + 128), 16, voff[1], soff, 0, 0);
        }
    }
    // fragment address (doubles) of row r, reduction index k
    static __device__ inline int at(int r, int k) { return k * P + r; }
};

// Operand stored with its REDUCTION index contiguous (16 doubles = 128 bytes of a row per K tile, `sr` between rows): one
// piece = 8 rows; LDS image [R][16], unpadded, the eight 16-byte chunks of row n stored at position chunk ^ ((n >> 1) & 7)
// (applied to the per-lane SOURCE address), which keeps the 4 x 16 fragment reads at two passes.  Needs whole tiles of rows.
template <int R, int BK, int NT>
struct DirectK {
    static constexpr int ELEMS = R * BK;
    static constexpr int NW = NT / 64;
    static constexpr int PER_WAVE = R / 8 / NW;
    static_assert(BK == 16 && R % (8 * NW) == 0 && PER_WAVE % 2 == 0, "direct-to-LDS tile (K-contiguous operand): shape");
    uint32_t voff[2];  // by the parity of the piece
    uint32_t sr64;     // 8 rows in bytes
    __device__ inline void init(int64_t sr, int lane) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int chunk = (lane & 7) ^ ((par * 4 + (lane >> 4)) & 7);
            voff[par] = (uint32_t)((lane >> 3) * sr * 8 + chunk * 16);
        }
        sr64 = (uint32_t)(sr * 64);
    }
    __device__ inline void copy(__amdgpu_buffer_rsrc_t rsrc, uint32_t tile_off, double *stage, int wave_u) const {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int piece = wave_u * PER_WAVE + i;  // uniform; its parity is i's (PER_WAVE is even)
            const uint32_t soff = __builtin_amdgcn_readfirstlane(tile_off + piece * sr64);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(stage + piece * 8 * BK), 16, voff[i & 1], soff, 0, 0);
        }
    }
    static constexpr int UNITS = PER_WAVE;
    __device__ __attribute__((always_inline)) void copy_unit(const int I, __amdgpu_buffer_rsrc_t rsrc, uint32_t tile_off, double *stage, int wave_u, int /*lane*/) const {
        const int piece = wave_u * PER_WAVE + I;
        const uint32_t soff = __builtin_amdgcn_readfirstlane(tile_off + piece * sr64);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(stage + piece * 8 * BK), 16, voff[I & 1], soff, 0, 0);
    }
};

// ALAY must be 1 (A row-contiguous).  BLAY 0: B row(N)-contiguous (DirectRows) ; BLAY 1: B K-contiguous (DirectK, ORIENT 0 only)
template <int BLAY, int BM, int BN, int BK, int WM, int WN, int ORIENT>
__global__ __launch_bounds__(WM *WN * 64) void k_gemm_f64d(GemmArgs<double> g) {
    constexpr int NT = WM * WN * 64;
    constexpr int WR = BM / WM, WC = BN / WN;
    constexpr int AM = ORIENT == 0 ? 4 : 16, BNW = ORIENT == 0 ? 16 : 4;
    constexpr int TM = WR / AM, TN = WC / BNW;
    constexpr int NKS = BK / 4;
    static_assert(WR % AM == 0 && WC % BNW == 0 && NKS >= 2 && (BLAY == 0 || ORIENT == 0), "tile shape");
    typedef DirectRows<BM, BK, NT> SA;
    typedef DirectRows<BN, BK, NT> SBR;
    typedef DirectK<BN, BK, NT> SBK;
    constexpr int PA = SA::P, PBR = SBR::P;
    constexpr int A_ELEMS = SA::ELEMS, STAGE = SA::ELEMS + (BLAY == 0 ? SBR::ELEMS : SBK::ELEMS);

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *smem = reinterpret_cast<double *>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WN, wn = wave % WN;
    const int lk = lane >> 4;
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
    const int nk = (int)((kend - kbeg) / BK);  // whole tiles (host-checked), >= 1

    double acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = 0.0;

    SA sa;
    SBR sbr;
    SBK sbk;
    sa.init(g.M - m0, g.sak, lane);
    if (BLAY == 0) sbr.init(g.N - n0, g.sbk, lane);
    else sbk.init(g.sbn, lane);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(g.a + m0 * g.sam + kbeg * g.sak), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(g.b + n0 * g.sbn + kbeg * g.sbk), 0, 0x7fffffff, 0x00020000);
    const uint32_t a_step = (uint32_t)(BK * g.sak * 8), b_step = (uint32_t)(BK * g.sbk * 8);
    auto copy_tile = [&](int t, double *stage) {
        sa.copy(a_rsrc, t * a_step, stage, wave_u);
        if (BLAY == 0) sbr.copy(b_rsrc, t * b_step, stage + A_ELEMS, wave_u);
        else sbk.copy(b_rsrc, t * b_step, stage + A_ELEMS, wave_u);
    };

    // fragment addresses of this lane inside a stage (doubles)
    const int a_lane = lk * PA + wm * WR + la;
    auto a_frag = [&](const double *st, int ks, int i) -> double { return st[a_lane + ks * 4 * PA + i * AM]; };
    // B row-contiguous: [BK][PBR] ; B K-contiguous: swizzled [BN][16], (nn >> 1) & 7 == (lbn >> 1) & 7 (ORIENT 0: nn = 16 * x + lbn)
    const int b_lane = BLAY == 0 ? lk * PBR + wn * WC + lbn : (wn * WC + lbn) * BK + (lk & 1);
    const int b_swz = (((lk >> 1) ^ (lbn >> 1)) & 7) * 2;
    auto b_frag = [&](const double *st, int ks, int j) -> double {
        if (BLAY == 0) return st[A_ELEMS + b_lane + ks * 4 * PBR + j * BNW];
        return st[A_ELEMS + b_lane + j * BNW * BK + (b_swz ^ (ks * 4))];
    };

    // ---- prologue --------------------------------------------------------------------------------------------------------
    copy_tile(0, smem);
    copy_tile(min(1, nk - 1), smem + STAGE);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    double fa[2][TM], fb[2][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = b_frag(smem, 0, j);
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = a_frag(smem, 0, i);

    for (int it = 0; it < nk; ++it) {
        double *cur = smem + (it & 1) * STAGE;
        const double *nxt = smem + ((it & 1) ^ 1) * STAGE;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            // the fragments of the next sub-step: same buffer, or sub-step 0 of the other buffer (behind the barrier)
            const double *fs = ks + 1 < NKS ? cur : nxt;
            const int nks = ks + 1 < NKS ? ks + 1 : 0;
            const int pb = ks & 1, qb = pb ^ 1;
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[qb][j] = b_frag(fs, nks, j);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[qb][i] = a_frag(fs, nks, i);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[pb][i], fb[pb][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == NKS - 2) {
                // tile it + 1 has landed in `nxt` (copies issued behind the previous barrier); this wave's reads of `cur` are
                // complete, so behind the barrier `cur` is free for tile it + 2
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                copy_tile(min(it + 2, nk - 1), cur);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // the (redundant) copies of the last iterations must not outlive the workgroup's LDS allocation
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

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
                    double *cp = g.c + gm * g.scm + gn * g.scn;
                    *cp = g.beta == 0.0 ? g.alpha * v : g.alpha * v + g.beta * (*cp);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// The sketch shape with a HAND-ORDERED main loop.  Same tiles, copies and barrier placement as k_gemm_f64d (A row-contiguous,
// B K-contiguous and swizzled, 4-row x 16-column micro tiles, 16-deep K tiles), but every LDS read and MFMA of the loop is an
// `asm volatile` statement, so the instruction order is the written one and the registers live exactly as long as written:
//   sub-step s:  4 B-fragment reads for s + 1 | for i = 0..TM-1: [wait] 4 MFMAs on A-fragment i, then the read of A-fragment i
//                for s + 1 INTO THE SAME variable (one rotating A set, B double buffered: 50 fragment registers instead of 84)
// LDS data return in issue order and 20 reads are always issued between a fragment's read and its use, so `s_waitcnt
// lgkmcnt(15)` in front of every MFMA group is enough (the counter has 4 bits) and never stalls.  With the masked tail copy
// (DirectRows<..., true>) a stage is 34.8 KB, and with 4 waves per workgroup two workgroups share a CU: each one's barrier and
// copy issue hide behind the other's MFMAs.
// ---------------------------------------------------------------------------------------------------------------------
// single instructions as ordered statements (clang does not accept captured variables as asm operands inside lambdas, so the
// compile-time loops are fold expressions over these function templates)
template <int OFF>
__device__ inline void asm_lds_read(double &dst, uint32_t addr) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF)); }
__device__ inline void asm_mfma(double &acc, const double &a, const double &b) { asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); }
template <int TN, int OFF0, int JSTRIDE, int... J>
__device__ inline void asm_read_b(double (&fb)[TN], uint32_t vb, std::integer_sequence<int, J...>) { (asm_lds_read<OFF0 + J * JSTRIDE>(fb[J], vb), ...); }
template <int TM, int OFF0, int ISTRIDE, int... I>
__device__ inline void asm_read_a(double (&fa)[TM], uint32_t va, std::integer_sequence<int, I...>) { (asm_lds_read<OFF0 + I * ISTRIDE>(fa[I], va), ...); }
template <int TN, int... J>
__device__ inline void asm_mfma_row(double (&acc)[TN], const double &a, const double (&fb)[TN], std::integer_sequence<int, J...>) { (asm_mfma(acc[J], a, fb[J]), ...); }
// one A fragment: wait, TN MFMAs, re-read the fragment for the next sub-step into the same variable
template <int TM, int TN, int OFF0, int ISTRIDE, int I>
__device__ inline void asm_group(double (&acc)[TM][TN], double (&fa)[TM], const double (&fb)[TN], uint32_t va) {
    asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
    asm_mfma_row<TN>(acc[I], fa[I], fb, std::make_integer_sequence<int, TN>{});
    asm_lds_read<OFF0 + I * ISTRIDE>(fa[I], va);
}
template <int TM, int TN, int OFF0, int ISTRIDE, int... I>
__device__ inline void asm_groups(double (&acc)[TM][TN], double (&fa)[TM], const double (&fb)[TN], uint32_t va, std::integer_sequence<int, I...>) {
    (asm_group<TM, TN, OFF0, ISTRIDE, I>(acc, fa, fb, va), ...);
}

// BLAY 1 / ORIENT 0: the sketch (B K-contiguous, swizzled image; 4-row x 16-column micro tiles)
// BLAY 0 / ORIENT 1: the projection (B N-contiguous, padded image; 16-row x 4-column micro tiles)
template <int BLAY, int ORIENT, int BM, int BN, int WM, int WN, bool MASKTAIL>
__global__ __launch_bounds__(WM *WN * 64, 2) void k_gemm_f64a(GemmArgs<double> g) {
    constexpr int BK = 16;
    constexpr int NT = WM * WN * 64;
    constexpr int WR = BM / WM, WC = BN / WN;
    constexpr int AM = ORIENT == 0 ? 4 : 16, BNW = ORIENT == 0 ? 16 : 4;
    constexpr int TM = WR / AM, TN = WC / BNW;
    static_assert((BLAY == 1 && ORIENT == 0) || (BLAY == 0 && ORIENT == 1), "instantiated pairings");
    static_assert(WR % AM == 0 && WC % BNW == 0 && TM + TN - 1 >= 15, "wave tile shape (and at least 15 reads between a fragment's read and its use)");
    typedef DirectRows<BM, BK, NT, MASKTAIL> SA;
    typedef DirectRows<BN, BK, NT> SBR;
    typedef DirectK<BN, BK, NT> SBK;
    constexpr int PA = SA::P, PB = SBR::P;
    constexpr int A_ELEMS = SA::ELEMS, STAGE = SA::ELEMS + (BLAY == 0 ? SBR::ELEMS : SBK::ELEMS);

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *smem = reinterpret_cast<double *>(smem_raw);
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char *)smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WN, wn = wave % WN;
    const int lk = lane >> 4;
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
    const int nk = (int)((kend - kbeg) / BK);

    double acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = 0.0;

    SA sa;
    SBR sbr;
    SBK sbk;
    sa.init(g.M - m0, g.sak, lane);
    if (BLAY == 0) sbr.init(g.N - n0, g.sbk, lane);
    else sbk.init(g.sbn, lane);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(g.a + m0 * g.sam + kbeg * g.sak), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(g.b + n0 * g.sbn + kbeg * g.sbk), 0, 0x7fffffff, 0x00020000);
    const uint32_t a_step = (uint32_t)(BK * g.sak * 8), b_step = (uint32_t)(BK * g.sbk * 8);
    auto copy_tile = [&](int t, double *stage) {
        sa.copy(a_rsrc, t * a_step, stage, wave_u, lane);
        if (BLAY == 0) sbr.copy(b_rsrc, t * b_step, stage + A_ELEMS, wave_u, lane);
        else sbk.copy(b_rsrc, t * b_step, stage + A_ELEMS, wave_u);
    };

    // LDS byte addresses of this lane's fragments inside stage 0.  A fragment i of sub-step s: + (s * 4 * PA + i * AM) * 8.
    // B, K-contiguous image: row base + ((chunk pair) ^ 32 s), fragment j + j * 16 rows; N-contiguous image: + (s * 4 * PB + j * BNW) * 8.
    const uint32_t a_addr = lds0 + (uint32_t)(lk * PA + wm * WR + la) * 8;
    const uint32_t b_addr = BLAY == 1 ? lds0 + (uint32_t)(A_ELEMS + (wn * WC + lbn) * BK + (lk & 1)) * 8
                                      : lds0 + (uint32_t)(A_ELEMS + lk * PB + wn * WC + lbn) * 8;
    const uint32_t swz8 = BLAY == 1 ? (uint32_t)((((lk >> 1) ^ (lbn >> 1)) & 7) * 16) : 0u;
    constexpr int ISTRIDE = AM * 8, AKS = 4 * PA * 8;
    constexpr int JSTRIDE = BLAY == 1 ? 16 * BK * 8 : BNW * 8, BKS = BLAY == 1 ? 0 : 4 * PB * 8;
    auto b_base = [&](uint32_t stage_off, int ks) -> uint32_t { return BLAY == 1 ? b_addr + stage_off + (swz8 ^ (uint32_t)(ks * 32)) : b_addr + stage_off; };

    copy_tile(0, smem);
    copy_tile(min(1, nk - 1), smem + STAGE);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    double fa[TM], fb[2][TN];
    asm_read_b<TN, 0, JSTRIDE>(fb[0], b_base(0, 0), std::make_integer_sequence<int, TN>{});
    asm_read_a<TM, 0, ISTRIDE>(fa, a_addr, std::make_integer_sequence<int, TM>{});

    for (int it = 0; it < nk; ++it) {
        const uint32_t cur_off = (uint32_t)((it & 1) * STAGE * 8), nxt_off = (uint32_t)(((it & 1) ^ 1) * STAGE * 8);
        // sub-steps 0 .. 2: the next fragments come from the same stage; sub-step 3: from sub-step 0 of the other stage
        {
            asm_read_b<TN, 1 * BKS, JSTRIDE>(fb[1], b_base(cur_off, 1), std::make_integer_sequence<int, TN>{});
            asm_groups<TM, TN, 1 * AKS, ISTRIDE>(acc, fa, fb[0], a_addr + cur_off, std::make_integer_sequence<int, TM>{});
        }
        {
            asm_read_b<TN, 2 * BKS, JSTRIDE>(fb[0], b_base(cur_off, 2), std::make_integer_sequence<int, TN>{});
            asm_groups<TM, TN, 2 * AKS, ISTRIDE>(acc, fa, fb[1], a_addr + cur_off, std::make_integer_sequence<int, TM>{});
        }
        {
            asm_read_b<TN, 3 * BKS, JSTRIDE>(fb[1], b_base(cur_off, 3), std::make_integer_sequence<int, TN>{});
            asm_groups<TM, TN, 3 * AKS, ISTRIDE>(acc, fa, fb[0], a_addr + cur_off, std::make_integer_sequence<int, TM>{});
            // tile it + 1 has landed in the other stage; this wave's reads of the current one are complete
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            copy_tile(min(it + 2, nk - 1), smem + (it & 1) * STAGE);
        }
        {
            asm_read_b<TN, 0, JSTRIDE>(fb[0], b_base(nxt_off, 0), std::make_integer_sequence<int, TN>{});
            asm_groups<TM, TN, 0, ISTRIDE>(acc, fa, fb[1], a_addr + nxt_off, std::make_integer_sequence<int, TM>{});
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // outstanding (redundant) copies and prefetched fragments

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
                    double *cp = g.c + gm * g.scm + gn * g.scn;
                    *cp = g.beta == 0.0 ? g.alpha * v : g.alpha * v + g.beta * (*cp);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_gemm_f64r: k_gemm_f64a's hand-ordered loop on a THREE-stage LDS ring (round 3).  What k_gemm_f64a still lost were the phases
// that are synchronous across the workgroup (ablation, DESIGN.md section 3: copy issue 8 %, barrier 4.5 % of the loop): at the
// end of sub-step 2 every wave drained its LDS reads (lgkmcnt(0): the stage it had just read was about to be overwritten), met
// the others, and then all eight waves issued their copies of the next tile at once -- no wave issues an MFMA meanwhile.  Here
//   * the copies of tile t + 2 go into the stage of tile t - 1, whose last reads every wave completed before it reached the
//     barrier of tile t: the barrier needs no drain of LDS reads any more -- fragment prefetches stay in flight across it --
//     and waits (vmcnt(0)) only for copies that were issued more than two sub-steps earlier;
//   * a wave's copies are single instructions INSIDE its MFMA stream (one behind each of the first MFMA groups of a sub-step),
//     the older half of the workgroup in sub-step 3 and the younger half (waves NW/2 ..: the SIMD partners of the first half)
//     in sub-step 0 of the next tile, so the issue of one wave's copy is covered by its partner's MFMAs on the same SIMD.
// Tiles, wave tiles, LDS images, fragment rotation and counted waits are k_gemm_f64a's; one stage more of LDS (154 / 160 KB).
// ---------------------------------------------------------------------------------------------------------------------
// DBG (diagnostic instantiations only, -DRC_GEMM_PIPE_DEBUG + RC_GEMM_RING_DBG): 1 = no copies in the loop, 2 = no tile barrier,
// 4 = no fragment reads / counted waits (the MFMAs run on stale registers), 8 = no MFMAs.  Results are garbage; only the time counts.
template <int DBG, int TM, int TN, int OFF0, int ISTRIDE, int I>
__device__ inline void asm_group_d(double (&acc)[TM][TN], double (&fa)[TM], const double (&fb)[TN], uint32_t va) {
    if constexpr (!(DBG & 4)) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
    if constexpr (!(DBG & 8)) asm_mfma_row<TN>(acc[I], fa[I], fb, std::make_integer_sequence<int, TN>{});
    if constexpr (!(DBG & 4)) asm_lds_read<OFF0 + I * ISTRIDE>(fa[I], va);
}
template <int DBG, int TM, int TN, int OFF0, int ISTRIDE, class F, int... I>
__device__ inline void asm_groups_c(double (&acc)[TM][TN], double (&fa)[TM], const double (&fb)[TN], uint32_t va, F &&after, std::integer_sequence<int, I...>) {
    ((asm_group_d<DBG, TM, TN, OFF0, ISTRIDE, I>(acc, fa, fb, va), after(std::integral_constant<int, I>{})), ...);
}

template <int BLAY, int ORIENT, int BM, int BN, int WM, int WN, bool MASKTAIL, int DBG = 0>
__global__ __launch_bounds__(WM *WN * 64, 2) void k_gemm_f64r(GemmArgs<double> g) {
    constexpr int BK = 16, NSTAGE = 3;
    constexpr int NT = WM * WN * 64, NW = WM * WN;
    constexpr int WR = BM / WM, WC = BN / WN;
    constexpr int AM = ORIENT == 0 ? 4 : 16, BNW = ORIENT == 0 ? 16 : 4;
    constexpr int TM = WR / AM, TN = WC / BNW;
    static_assert((BLAY == 1 && ORIENT == 0) || (BLAY == 0 && ORIENT == 1), "instantiated pairings");
    static_assert(WR % AM == 0 && WC % BNW == 0 && TM + TN - 1 >= 15, "wave tile shape (and at least 15 reads between a fragment's read and its use)");
    typedef DirectRows<BM, BK, NT, MASKTAIL> SA;
    typedef DirectRows<BN, BK, NT> SBR;
    typedef DirectK<BN, BK, NT> SBK;
    constexpr int PA = SA::P, PB = SBR::P;
    constexpr int A_ELEMS = SA::ELEMS, STAGE = SA::ELEMS + (BLAY == 0 ? SBR::ELEMS : SBK::ELEMS);
    constexpr int UA = SA::UNITS, UB = BLAY == 0 ? SBR::UNITS : SBK::UNITS, NUNITS = UA + UB;
    static_assert(NUNITS <= TM - 1, "one copy behind each of the MFMA groups 1 .. NUNITS of a sub-step");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *smem = reinterpret_cast<double *>(smem_raw);
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char *)smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool early = wave_u < NW / 2;
    const int wm = wave / WN, wn = wave % WN;
    const int lk = lane >> 4;
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
    const int nk = (int)((kend - kbeg) / BK);

    double acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = 0.0;

    SA sa;
    SBR sbr;
    SBK sbk;
    sa.init(g.M - m0, g.sak, lane);
    if (BLAY == 0) sbr.init(g.N - n0, g.sbk, lane);
    else sbk.init(g.sbn, lane);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(g.a + m0 * g.sam + kbeg * g.sak), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(g.b + n0 * g.sbn + kbeg * g.sbk), 0, 0x7fffffff, 0x00020000);
    const uint32_t a_step = (uint32_t)(BK * g.sak * 8), b_step = (uint32_t)(BK * g.sbk * 8);
    // copy unit u of tile t into `stage` (u < UA: the A operand's pieces of this wave, then the B operand's)
    // (u is a literal at every call site and everything below is inlined; no nested generic lambdas -- clang 19's host pass fails to
    // substitute them inside the second and later instantiations of a kernel template)
    auto copy_unit = [&](const int u, int t, double *stage) -> void {
        if (u < UA) sa.copy_unit(u, a_rsrc, t * a_step, stage, wave_u, lane);
        else if (BLAY == 0) sbr.copy_unit(u - UA, b_rsrc, t * b_step, stage + A_ELEMS, wave_u, lane);
        else sbk.copy_unit(u - UA, b_rsrc, t * b_step, stage + A_ELEMS, wave_u, lane);
    };
    auto copy_tile = [&](int t, double *stage) {
        sa.copy(a_rsrc, t * a_step, stage, wave_u, lane);
        if (BLAY == 0) sbr.copy(b_rsrc, t * b_step, stage + A_ELEMS, wave_u, lane);
        else sbk.copy(b_rsrc, t * b_step, stage + A_ELEMS, wave_u);
    };

    const uint32_t a_addr = lds0 + (uint32_t)(lk * PA + wm * WR + la) * 8;
    const uint32_t b_addr = BLAY == 1 ? lds0 + (uint32_t)(A_ELEMS + (wn * WC + lbn) * BK + (lk & 1)) * 8
                                      : lds0 + (uint32_t)(A_ELEMS + lk * PB + wn * WC + lbn) * 8;
    const uint32_t swz8 = BLAY == 1 ? (uint32_t)((((lk >> 1) ^ (lbn >> 1)) & 7) * 16) : 0u;
    constexpr int ISTRIDE = AM * 8, AKS = 4 * PA * 8;
    constexpr int JSTRIDE = BLAY == 1 ? 16 * BK * 8 : BNW * 8, BKS = BLAY == 1 ? 0 : 4 * PB * 8;
    auto b_base = [&](uint32_t stage_off, int ks) -> uint32_t { return BLAY == 1 ? b_addr + stage_off + (swz8 ^ (uint32_t)(ks * 32)) : b_addr + stage_off; };
    auto nothing = [](auto) -> void {};

    // prologue: tile 0 by everyone; tile 1 by the older half now, by the younger half inside sub-step 0 of tile 0 (its regular slot)
    copy_tile(0, smem);
    if (early && nk > 1) copy_tile(1, smem + STAGE);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    double fa[TM], fb[2][TN];
    asm_read_b<TN, 0, JSTRIDE>(fb[0], b_base(0, 0), std::make_integer_sequence<int, TN>{});
    asm_read_a<TM, 0, ISTRIDE>(fa, a_addr, std::make_integer_sequence<int, TM>{});

    int s_cur = 0;  // stage of tile `it`
    for (int it = 0; it < nk; ++it) {
        const int s_nxt = s_cur == NSTAGE - 1 ? 0 : s_cur + 1, s_prv = s_cur == 0 ? NSTAGE - 1 : s_cur - 1;
        const uint32_t cur_off = (uint32_t)(s_cur * STAGE * 8), nxt_off = (uint32_t)(s_nxt * STAGE * 8);
        {   // sub-step 0; younger half: its copies of tile it + 1 (stage free since the barrier of tile it - 1)
            const bool go = !(DBG & 1) && !early && it + 1 < nk;
            double *dst = smem + s_nxt * STAGE;
            if constexpr (!(DBG & 4)) asm_read_b<TN, 1 * BKS, JSTRIDE>(fb[1], b_base(cur_off, 1), std::make_integer_sequence<int, TN>{});
            asm_groups_c<DBG, TM, TN, 1 * AKS, ISTRIDE>(acc, fa, fb[0], a_addr + cur_off, [&](auto ic) -> void {
                constexpr int I = decltype(ic)::value;
                if constexpr (I >= 1 && I - 1 < NUNITS) { if (go) copy_unit(I - 1, it + 1, dst); }
            }, std::make_integer_sequence<int, TM>{});
        }
        {
            if constexpr (!(DBG & 4)) asm_read_b<TN, 2 * BKS, JSTRIDE>(fb[0], b_base(cur_off, 2), std::make_integer_sequence<int, TN>{});
            asm_groups_c<DBG, TM, TN, 2 * AKS, ISTRIDE>(acc, fa, fb[1], a_addr + cur_off, nothing, std::make_integer_sequence<int, TM>{});
        }
        {
            if constexpr (!(DBG & 4)) asm_read_b<TN, 3 * BKS, JSTRIDE>(fb[1], b_base(cur_off, 3), std::make_integer_sequence<int, TN>{});
            asm_groups_c<DBG, TM, TN, 3 * AKS, ISTRIDE>(acc, fa, fb[0], a_addr + cur_off, nothing, std::make_integer_sequence<int, TM>{});
            // tile it + 1 has landed in its stage (this wave's copies: vmcnt(0); everyone's: the barrier); every wave that passes has
            // finished reading tile it - 1.  The fragment reads for sub-step 3 stay in flight.
            if constexpr (!(DBG & 2)) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        {   // sub-step 3; older half: its copies of tile it + 2 into the stage of tile it - 1
            const bool go = !(DBG & 1) && early && it + 2 < nk;
            double *dst = smem + s_prv * STAGE;
            if constexpr (!(DBG & 4)) asm_read_b<TN, 0, JSTRIDE>(fb[0], b_base(nxt_off, 0), std::make_integer_sequence<int, TN>{});
            asm_groups_c<DBG, TM, TN, 0, ISTRIDE>(acc, fa, fb[1], a_addr + nxt_off, [&](auto ic) -> void {
                constexpr int I = decltype(ic)::value;
                if constexpr (I >= 1 && I - 1 < NUNITS) { if (go) copy_unit(I - 1, it + 2, dst); }
            }, std::make_integer_sequence<int, TM>{});
        }
        s_cur = s_nxt;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // prefetched fragments of the (non-existent) next tile

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
                    double *cp = g.c + gm * g.scm + gn * g.scn;
                    *cp = g.beta == 0.0 ? g.alpha * v : g.alpha * v + g.beta * (*cp);
                }
            }
        }
}

template <int BLAY, int ORIENT, int BM, int BN, int WM, int WN, bool MASKTAIL, int DBG = 0>
static bool launch_r(rc_context *c, const GemmArgs<double> &g) {
    constexpr int NT = WM * WN * 64;
    constexpr size_t lds = 3 * (size_t)(DirectRows<BM, 16, NT, MASKTAIL>::ELEMS + (BLAY == 0 ? DirectRows<BN, 16, NT>::ELEMS : DirectK<BN, 16, NT>::ELEMS)) * sizeof(double);
    static_assert(lds <= 160 * 1024, "ring does not fit LDS");
    if (g.sam != 1) return false;
    if (BLAY == 0 ? g.sbn != 1 : (g.sbk != 1 || g.N % BN != 0)) return false;
    const int64_t a_span = ((int64_t)256 * g.sam + g.kchunk * g.sak + 2) * 8;
    const int64_t b_span = ((int64_t)BN * g.sbn + g.kchunk * g.sbk + 2) * 8;
    if (a_span >= (1ll << 31) || b_span >= (1ll << 31) || g.sak < 0 || g.sbn < 0 || g.sbk < 0) return false;
    auto kern = k_gemm_f64r<BLAY, ORIENT, BM, BN, WM, WN, MASKTAIL, DBG>;
    static bool attr_set[64] = {};
    if (!attr_set[c->device & 63]) {
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[c->device & 63] = true;
    }
    char nm[128];
    snprintf(nm, sizeof(nm), "k_gemm_f64r<%d,%d,%d,%d,%d,%d,%s,%d>", BLAY, ORIENT, BM, BN, WM, WN, MASKTAIL ? "true" : "false", DBG);  // as rocprofv3 prints it
    c->last_gemm_kernel = nm;
    ProfScope ps(c, "kernel:k_gemm_mfma<f64> M=%lld N=%lld K=%lld", (long long)g.M, (long long)g.N, (long long)g.K);
    hipLaunchKernelGGL(kern, dim3((unsigned)(g.tiles_m * g.tiles_n), (unsigned)g.splits), dim3(NT), lds, c->stream, g);
    return true;
}

template <int BLAY, int ORIENT, int BM, int BN, int WM, int WN, bool MASKTAIL>
static bool launch_a(rc_context *c, const GemmArgs<double> &g) {
    constexpr int NT = WM * WN * 64;
    constexpr size_t lds = 2 * (size_t)(DirectRows<BM, 16, NT, MASKTAIL>::ELEMS + (BLAY == 0 ? DirectRows<BN, 16, NT>::ELEMS : DirectK<BN, 16, NT>::ELEMS)) * sizeof(double);
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    if (g.sam != 1) return false;
    if (BLAY == 0 ? g.sbn != 1 : (g.sbk != 1 || g.N % BN != 0)) return false;
    const int64_t a_span = ((int64_t)256 * g.sam + g.kchunk * g.sak + 2) * 8;
    const int64_t b_span = ((int64_t)BN * g.sbn + g.kchunk * g.sbk + 2) * 8;
    if (a_span >= (1ll << 31) || b_span >= (1ll << 31) || g.sak < 0 || g.sbn < 0 || g.sbk < 0) return false;
    auto kern = k_gemm_f64a<BLAY, ORIENT, BM, BN, WM, WN, MASKTAIL>;
    static bool attr_set[64] = {};
    if (lds > 48 * 1024 && !attr_set[c->device & 63]) {
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[c->device & 63] = true;
    }
    char nm[128];
    snprintf(nm, sizeof(nm), "k_gemm_f64a<%d,%d,%d,%d,%d,%d,%s>", BLAY, ORIENT, BM, BN, WM, WN, MASKTAIL ? "true" : "false");
    c->last_gemm_kernel = nm;
    ProfScope ps(c, "kernel:k_gemm_mfma<f64> M=%lld N=%lld K=%lld", (long long)g.M, (long long)g.N, (long long)g.K);
    hipLaunchKernelGGL(kern, dim3((unsigned)(g.tiles_m * g.tiles_n), (unsigned)g.splits), dim3(NT), lds, c->stream, g);
    return true;
}

template <int BLAY, int BM, int BN, int BK, int WM, int WN, int ORIENT>
static bool launch_d(rc_context *c, const GemmArgs<double> &g) {
    constexpr int NT = WM * WN * 64;
    constexpr size_t lds = 2 * (size_t)(DirectRows<BM, BK, NT>::ELEMS + (BLAY == 0 ? DirectRows<BN, BK, NT>::ELEMS : DirectK<BN, BK, NT>::ELEMS)) * sizeof(double);
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    // unit row stride of A; B: unit column stride (BLAY 0) or unit K stride with whole tiles of columns (BLAY 1)
    if (g.sam != 1) return false;
    if (BLAY == 0 ? g.sbn != 1 : (g.sbk != 1 || g.N % BN != 0)) return false;
    // 32-bit byte offsets from a workgroup's first element over its whole K chunk
    const int64_t a_span = ((int64_t)256 * g.sam + g.kchunk * g.sak + 2) * 8;
    const int64_t b_span = ((int64_t)BN * g.sbn + g.kchunk * g.sbk + 2) * 8;
    if (a_span >= (1ll << 31) || b_span >= (1ll << 31) || g.sak < 0 || g.sbn < 0 || g.sbk < 0) return false;
    auto kern = k_gemm_f64d<BLAY, BM, BN, BK, WM, WN, ORIENT>;
    static bool attr_set[64] = {};
    if (lds > 48 * 1024 && !attr_set[c->device & 63]) {
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[c->device & 63] = true;
    }
    char nm[128];
    snprintf(nm, sizeof(nm), "k_gemm_f64d<%d,%d,%d,%d,%d,%d,%d>", BLAY, BM, BN, BK, WM, WN, ORIENT);
    c->last_gemm_kernel = nm;
    ProfScope ps(c, "kernel:k_gemm_mfma<f64> M=%lld N=%lld K=%lld", (long long)g.M, (long long)g.N, (long long)g.K);
    hipLaunchKernelGGL(kern, dim3((unsigned)(g.tiles_m * g.tiles_n), (unsigned)g.splits), dim3(NT), lds, c->stream, g);
    return true;
}

template <int ALAY, int BLAY, int BM, int BN, int BK, int WM, int WN, int ORIENT, int DBG = 0>
static bool launch_p(rc_context *c, const GemmArgs<double> &g) {
    constexpr int NT = WM * WN * 64;
    typedef PipeStage<ALAY, BM, BK, NT> SA;
    typedef PipeStage<BLAY == 1 ? 0 : 1, BN, BK, NT> SB;
    constexpr size_t lds = 2 * (size_t)(SA::ELEMS + SB::ELEMS) * sizeof(double);
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    // 32-bit byte offsets from a workgroup's first element over its whole K chunk
    const int64_t a_span = ((int64_t)BM * g.sam + g.kchunk * g.sak + 2) * 8;
    const int64_t b_span = ((int64_t)BN * g.sbn + g.kchunk * g.sbk + 2) * 8;
    if (a_span >= (1ll << 31) || b_span >= (1ll << 31) || g.sam < 0 || g.sak < 0 || g.sbn < 0 || g.sbk < 0) return false;
    auto kern = k_gemm_f64p<ALAY, BLAY, BM, BN, BK, WM, WN, ORIENT, DBG>;
    static bool attr_set[64] = {};
    if (lds > 48 * 1024 && !attr_set[c->device & 63]) {
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[c->device & 63] = true;
    }
    char nm[128];
    snprintf(nm, sizeof(nm), "k_gemm_f64p<%d,%d,%d,%d,%d,%d,%d,%d>", ALAY, BLAY, BM, BN, BK, WM, WN, ORIENT);
    c->last_gemm_kernel = nm;
    ProfScope ps(c, "kernel:k_gemm_mfma<f64> M=%lld N=%lld K=%lld", (long long)g.M, (long long)g.N, (long long)g.K);
    hipLaunchKernelGGL(kern, dim3((unsigned)(g.tiles_m * g.tiles_n), (unsigned)g.splits), dim3(NT), lds, c->stream, g);
    return true;
}

bool gemm_f64p_launch(rc_context *c, const GemmArgs<double> &g, int alay, int blay, int bm, int bn, int bk, int wm, int wn, int orient, bool vec2) {
    static const int use = [] { const char *e = getenv("RC_GEMM_PIPE"); return e ? atoi(e) : 1; }();
    if (!use || !vec2 || bk != 16) return false;
    // whole K tiles in every split, at least one
    if (g.K % bk != 0 || g.kchunk % bk != 0 || g.K < bk) return false;
#define RC_PIPE_CASE(AL, BL, BM_, BN_, WM_, WN_, OR_)                                                              \
    if (alay == AL && blay == BL && bm == BM_ && bn == BN_ && wm == WM_ && wn == WN_ && orient == OR_)             \
        return launch_p<AL, BL, BM_, BN_, 16, WM_, WN_, OR_>(c, g);
#ifdef RC_GEMM_PIPE_DEBUG
    {
        static const int dbg = [] { const char *e = getenv("RC_GEMM_PIPE_DBG"); return e ? atoi(e) : 0; }();
        if (alay == 1 && blay == 0 && bm == 128 && bn == 256 && wm == 1 && wn == 8 && orient == 1) {
            if (dbg == 1) return launch_p<1, 0, 128, 256, 16, 1, 8, 1, 1>(c, g);
            if (dbg == 2) return launch_p<1, 0, 128, 256, 16, 1, 8, 1, 2>(c, g);
            if (dbg == 3) return launch_p<1, 0, 128, 256, 16, 1, 8, 1, 3>(c, g);
        }
    }
#endif
    // RC_GEMM_PIPE_DIRECT: 1 (default) = direct-to-LDS copies where they measured faster (the sketch: 377 -> 358 us; its
    // register-staged instance spills), 2 = also for the projection (325 us against 321 us register-staged), 0 = never
    static const int direct = [] { const char *e = getenv("RC_GEMM_PIPE_DIRECT"); return e ? atoi(e) : 1; }();
    if (direct && alay == 1 && bk == 16) {
        // hand-ordered main loop (k_gemm_f64a) on two 4-wave workgroups per CU, where kernels_gemm.hip chose 128-column tiles for it
        static const int wide_asm = [] { const char *e = getenv("RC_GEMM_PIPE_ASM"); return e ? atoi(e) : 1; }();  // 0: the compiler-scheduled loops (k_gemm_f64d / k_gemm_f64p)
        // three-stage ring with the copies inside the MFMA stream (k_gemm_f64r); RC_GEMM_RING=0: the two-stage k_gemm_f64a
        static const int ring = [] { const char *e = getenv("RC_GEMM_RING"); return e ? atoi(e) : 1; }();
#ifdef RC_GEMM_PIPE_DEBUG
        {
            static const int rdbg = [] { const char *e = getenv("RC_GEMM_RING_DBG"); return e ? atoi(e) : 0; }();
            if (rdbg && blay == 1 && bm == 136 && bn == 256 && wm == 2 && wn == 4 && orient == 0) {
#define RC_RDBG(D) if (rdbg == D) return launch_r<1, 0, 136, 256, 2, 4, true, D>(c, g);
                RC_RDBG(1) RC_RDBG(2) RC_RDBG(3) RC_RDBG(4) RC_RDBG(7) RC_RDBG(8) RC_RDBG(12) RC_RDBG(5) RC_RDBG(6)
#undef RC_RDBG
            }
        }
#endif
        // (RC_GEMM_RING: 1 both products, 2 the sketch only, 3 the projection only)
        if (wide_asm && (ring == 1 || ring == 2) && blay == 1 && bm == 136 && bn == 256 && wm == 2 && wn == 4 && orient == 0 && launch_r<1, 0, 136, 256, 2, 4, true>(c, g)) return true;
        if (wide_asm && (ring == 1 || ring == 3) && blay == 0 && bm == 128 && bn == 256 && wm == 1 && wn == 8 && orient == 1 && launch_r<0, 1, 128, 256, 1, 8, false>(c, g)) return true;
        if (wide_asm && blay == 1 && bm == 136 && bn == 256 && wm == 2 && wn == 4 && orient == 0 && launch_a<1, 0, 136, 256, 2, 4, true>(c, g)) return true;
        if (wide_asm && blay == 0 && bm == 128 && bn == 256 && wm == 1 && wn == 8 && orient == 1 && launch_a<0, 1, 128, 256, 1, 8, false>(c, g)) return true;
        if (blay == 1 && bm == 136 && bn == 128 && wm == 2 && wn == 2 && orient == 0 && launch_a<1, 0, 136, 128, 2, 2, true>(c, g)) return true;
        if (blay == 0 && bm == 128 && bn == 128 && wm == 1 && wn == 4 && orient == 1 && launch_a<0, 1, 128, 128, 1, 4, false>(c, g)) return true;
        if (blay == 1 && bm == 136 && bn == 256 && wm == 2 && wn == 4 && orient == 0 && launch_d<1, 136, 256, 16, 2, 4, 0>(c, g)) return true;
        if (direct >= 2 && blay == 0 && bm == 128 && bn == 256 && wm == 1 && wn == 8 && orient == 1 && launch_d<0, 128, 256, 16, 1, 8, 1>(c, g)) return true;
    }
    // the Gram products of the CholeskyQR passes (Y^T Y: K-contiguous both; Q1^T Q1 on the row-major Q1: M- / N-contiguous): one
    // 144 x 144 tile on 3 x 3 waves, K split over the workgroups -- the compiler-scheduled loop of kernels_gemm.hip spends ~5 us per
    // K tile on them (load -> LDS -> MFMA in sequence), this one keeps the next tile's loads in flight (RC_GEMM_PIPE_GRAM=0: off)
    static const int gram = [] { const char *e = getenv("RC_GEMM_PIPE_GRAM"); return e ? atoi(e) : 1; }();
    if (gram) {
        RC_PIPE_CASE(0, 1, 144, 144, 3, 3, 0)
        RC_PIPE_CASE(1, 0, 144, 144, 3, 3, 0)
    }
    // (the K = 128 shallow products -- Q = range Q_b, C = Q R11, U = range U_b -- through this loop on 128 x 128 tiles measured
    // neutral to slightly slower in the headline, 1072 / 1083 against 1084 / 1084: eight K tiles do not amortise the prologue)
    RC_PIPE_CASE(1, 1, 136, 256, 2, 4, 0)  // the sketch (transposed problem): 68 x 64 wave tiles
    RC_PIPE_CASE(1, 0, 128, 256, 1, 8, 1)  // the projection: 128 x 32 wave tiles
#undef RC_PIPE_CASE
    return false;
}

}  // namespace rc

namespace rc {
// (explicit: clang did not emit the host stub of the second instantiation from its use inside the `&&` chain above)
template __global__ void k_gemm_f64d<1, 136, 256, 16, 2, 4, 0>(GemmArgs<double>);
template __global__ void k_gemm_f64a<1, 0, 136, 128, 2, 2, true>(GemmArgs<double>);
template __global__ void k_gemm_f64a<1, 0, 136, 256, 2, 4, true>(GemmArgs<double>);
template __global__ void k_gemm_f64a<0, 1, 128, 256, 1, 8, false>(GemmArgs<double>);
template __global__ void k_gemm_f64a<0, 1, 128, 128, 1, 4, false>(GemmArgs<double>);
template __global__ void k_gemm_f64d<0, 128, 256, 16, 1, 8, 1>(GemmArgs<double>);
template __global__ void k_gemm_f64r<1, 0, 136, 256, 2, 4, true>(GemmArgs<double>);
template __global__ void k_gemm_f64r<0, 1, 128, 256, 1, 8, false>(GemmArgs<double>);
}
