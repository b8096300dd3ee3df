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
//  * Split-K with a DETERMINISTIC slab reduction (no float atomics): pivot
//    decisions downstream must not depend on arrival order.
//  * XCD-aware tile order: consecutive block ids land on different XCDs, so the
//    linear id is remapped to give each XCD a contiguous run of row tiles that
//    share the B panel in that XCD's L2.
#include "rc_common.hpp"

namespace rc {

static __host__ __device__ inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));

template <typename T> struct Acc;
template <> struct Acc<double> {
    typedef double4_t type;
    static __device__ inline type mfma(double a, double b, type c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ inline int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct Acc<float> {
    typedef float4_t type;
    static __device__ inline type mfma(float a, float b, type c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ inline int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

// smallest pitch >= n with pitch % 32 == 16 (conflict-free M/N-fastest fragment reads)
constexpr int pitch16(int n) { return n + ((16 - n % 32) + 32) % 32; }

template <typename T>
struct GemmArgs {
    const T *a, *b;
    T *c;
    int64_t M, N, K;
    int64_t sam, sak;  // A(m, k)
    int64_t sbk, sbn;  // B(k, n)
    int64_t scm, scn;  // C(m, n)
    T alpha, beta;
    int64_t kchunk;    // K range per split
    int splits;
    T *partial;        // [splits][M][N] when splits > 1
    int tiles_m, tiles_n;
};

// ALAY: 0 = A is K-contiguous (sak == 1), 1 = A is M-contiguous (sam == 1)
// BLAY: 0 = B is N-contiguous (sbn == 1), 1 = B is K-contiguous (sbk == 1)
template <typename T, int ALAY, int BLAY, int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(WM *WN * 64) void k_gemm_mfma(GemmArgs<T> g) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 16;  // 16x16 tiles per wave along M
    constexpr int TN = BN / WN / 16;
    static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0 && BK % 4 == 0, "tile shape");
    // LDS pitches (elements): fastest dimension mirrors the global contiguity.
    //  K-fastest image [rows][BK + 2]  : fragment address r*P + kk, P = BK+2 (== 2 mod 4) -> conflict free
    //  M/N-fastest image [BK][BX + pad]: fragment address kk*P + r, P == 16 mod 32        -> conflict free
    constexpr int PA = ALAY == 0 ? BK + 2 : pitch16(BM);
    constexpr int PB = BLAY == 1 ? BK + 2 : pitch16(BN);
    constexpr int A_ELEMS = ALAY == 0 ? BM * PA : BK * PA;
    constexpr int B_ELEMS = BLAY == 1 ? BN * PB : BK * PB;
    constexpr int A_PER_T = (BM * BK + NT - 1) / NT;
    constexpr int B_PER_T = (BN * BK + NT - 1) / NT;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *smem = reinterpret_cast<T *>(smem_raw);
    T *As[2] = {smem, smem + A_ELEMS + B_ELEMS};
    T *Bs[2] = {smem + A_ELEMS, smem + 2 * A_ELEMS + B_ELEMS};

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

    T ra[A_PER_T], rb[B_PER_T];

    auto load_tiles = [&](int64_t k0) {
#pragma unroll
        for (int e = 0; e < A_PER_T; ++e) {
            int idx = tid + e * NT;
            int mm, kk;
            if (ALAY == 0) { kk = idx % BK; mm = idx / BK; } else { mm = idx % BM; kk = idx / BM; }
            int64_t gm = m0 + mm, gk = k0 + kk;
            T v = 0;
            if (idx < BM * BK && gm < g.M && gk < kend) v = g.a[gm * g.sam + gk * g.sak];
            ra[e] = v;
        }
#pragma unroll
        for (int e = 0; e < B_PER_T; ++e) {
            int idx = tid + e * NT;
            int nn, kk;
            if (BLAY == 1) { kk = idx % BK; nn = idx / BK; } else { nn = idx % BN; kk = idx / BN; }
            int64_t gn = n0 + nn, gk = k0 + kk;
            T v = 0;
            if (idx < BN * BK && gn < g.N && gk < kend) v = g.b[gk * g.sbk + gn * g.sbn];
            rb[e] = v;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int e = 0; e < A_PER_T; ++e) {
            int idx = tid + e * NT;
            if (idx < BM * BK) {
                if (ALAY == 0) As[buf][(idx / BK) * PA + (idx % BK)] = ra[e];
                else As[buf][(idx / BM) * PA + (idx % BM)] = ra[e];
            }
        }
#pragma unroll
        for (int e = 0; e < B_PER_T; ++e) {
            int idx = tid + e * NT;
            if (idx < BN * BK) {
                if (BLAY == 1) Bs[buf][(idx / BK) * PB + (idx % BK)] = rb[e];
                else Bs[buf][(idx / BN) * PB + (idx % BN)] = rb[e];
            }
        }
    };

    const int64_t nk = kend > kbeg ? cdiv(kend - kbeg, BK) : 0;
    if (nk > 0) {
        load_tiles(kbeg);
        store_tiles(0);
    }
    __syncthreads();
    for (int64_t it = 0; it < nk; ++it) {
        const int buf = (int)(it & 1);
        if (it + 1 < nk) load_tiles(kbeg + (it + 1) * BK);  // global loads in flight during the MFMAs
        const T *as = As[buf], *bs = Bs[buf];
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
        if (it + 1 < nk) store_tiles(buf ^ 1);  // the other buffer was last read one iteration ago
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

// C = alpha * sum_s partial[s] + beta * C ; fixed summation order => deterministic
template <typename T>
__global__ __launch_bounds__(256) void k_splitk_reduce(GemmArgs<T> g) {
    const bool col_fast = (g.scn <= g.scm);
    const int64_t total = g.M * g.N;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t gm, gn;
        if (col_fast) { gm = e / g.N; gn = e - gm * g.N; }
        else { gn = e / g.M; gm = e - gn * g.M; }
        T s = 0;
        for (int sp = 0; sp < g.splits; ++sp) s += g.partial[((int64_t)sp * g.M + gm) * g.N + gn];
        T *cp = g.c + gm * g.scm + gn * g.scn;
        *cp = g.beta == (T)0 ? g.alpha * s : g.alpha * s + g.beta * (*cp);
    }
}

template <typename T, int ALAY, int BLAY, int BM, int BN, int BK, int WM, int WN>
static void launch_cfg(rc_context *c, GemmArgs<T> g) {
    constexpr int PA = ALAY == 0 ? BK + 2 : pitch16(BM);
    constexpr int PB = BLAY == 1 ? BK + 2 : pitch16(BN);
    constexpr int A_ELEMS = ALAY == 0 ? BM * PA : BK * PA;
    constexpr int B_ELEMS = BLAY == 1 ? BN * PB : BK * PB;
    constexpr size_t lds = 2 * (size_t)(A_ELEMS + B_ELEMS) * sizeof(T);
    g.tiles_m = (int)cdiv(g.M, BM);
    g.tiles_n = (int)cdiv(g.N, BN);
    const int64_t tiles = (int64_t)g.tiles_m * g.tiles_n;
    // split K until the grid covers the 256 CUs about twice (only worth it for deep K)
    int splits = 1;
    const int64_t ksteps = cdiv(g.K, BK);
    // (tiny outputs with a deep reduction -- the n x n Gram matrices of the CholeskyQR passes --
    //  need up to 128 slabs to reach every CU)
    while (tiles * splits < 384 && splits < 128 && ksteps / (splits * 2) >= (tiles >= 32 ? 16 : 4)) splits *= 2;
    g.kchunk = cdiv(cdiv(g.K, splits), BK) * BK;
    splits = (int)cdiv(g.K, g.kchunk);
    if (splits < 1) splits = 1;
    g.splits = splits;
    ArenaMark mark(c);
    if (splits > 1) g.partial = c->alloc<T>((size_t)splits * g.M * g.N);
    auto kern = k_gemm_mfma<T, ALAY, BLAY, BM, BN, BK, WM, WN>;
    static bool attr_set[64] = {};
    if (lds > 48 * 1024 && !attr_set[c->device & 63]) {
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[c->device & 63] = true;
    }
    {
        ProfScope ps(c, "kernel:k_gemm_mfma<%s> M=%lld N=%lld K=%lld", sizeof(T) == 8 ? "f64" : "f32", (long long)g.M, (long long)g.N, (long long)g.K);
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)splits), dim3(WM * WN * 64), lds, c->stream, g);
    }
    if (splits > 1) {
        ProfScope ps(c, "kernel:k_splitk_reduce M=%lld N=%lld splits=%d", (long long)g.M, (long long)g.N, splits);
        int grid = (int)std::min<int64_t>(cdiv(g.M * g.N, 256), 4096);
        hipLaunchKernelGGL(k_splitk_reduce<T>, dim3(grid), dim3(256), 0, c->stream, g);
    }
}

template <typename T, int ALAY, int BLAY>
static void launch_shape(rc_context *c, const GemmArgs<T> &g) {
    // Skinny outputs (the sketch Y = A Omega has N = k + p ~ 69..133; the range
    // projection B = Q^H A has M = k ~ 64..128) get tiles that cover the short
    // side once, so the long operand streams from HBM exactly once.
    if (g.M <= 144 && g.N <= 144 && g.M > 80 && g.N > 80) launch_cfg<T, ALAY, BLAY, 144, 144, 16, 3, 3>(c, g);
    else if (g.N <= 80) launch_cfg<T, ALAY, BLAY, 128, 80, 16, 4, 1>(c, g);
    else if (g.N <= 144) launch_cfg<T, ALAY, BLAY, 128, 144, 16, 4, 1>(c, g);
    else if (g.M <= 80) launch_cfg<T, ALAY, BLAY, 80, 128, 16, 1, 4>(c, g);
    else if (g.M <= 144) launch_cfg<T, ALAY, BLAY, 144, 128, 16, 1, 4>(c, g);
    else launch_cfg<T, ALAY, BLAY, 128, 128, 16, 2, 2>(c, g);
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
    GemmArgs<T> g;
    g.a = a.p; g.b = b.p; g.c = cm.p;
    g.M = a.rows; g.N = b.cols; g.K = a.cols;
    g.sam = a.rs; g.sak = a.cs; g.sbk = b.rs; g.sbn = b.cs; g.scm = cm.rs; g.scn = cm.cs;
    g.alpha = alpha; g.beta = beta;
    g.kchunk = g.K; g.splits = 1; g.partial = nullptr; g.tiles_m = g.tiles_n = 0;
    const bool a_kc = (a.cs == 1);
    const bool b_nc = (b.cs == 1);
    if (a_kc) {
        if (b_nc) launch_shape<T, 0, 0>(c, g); else launch_shape<T, 0, 1>(c, g);
    } else {
        if (b_nc) launch_shape<T, 1, 0>(c, g); else launch_shape<T, 1, 1>(c, g);
    }
}

template void gemm<double>(rc_context *, double, Mat<double>, Mat<double>, double, Mat<double>);
template void gemm<float>(rc_context *, float, Mat<float>, Mat<float>, float, Mat<float>);

}  // namespace rc
