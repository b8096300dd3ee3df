// Utility kernels: Gaussian fill (Philox4x32-10 + Box-Muller), strided copies /
// transposes, permutation gathers, norms and small reductions.  gfx950 only.
#include "rc_common.hpp"

namespace rc {

static __host__ __device__ inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ===========================================================================
// Philox4x32-10 (Salmon et al., SC'11) -- counter-based, so element e of the
// stream is a pure function of (seed, offset + e): layout independent.
// Replaces rand_distr::Normal sampling, /root/reference/src/random_matrix.rs:120-125.
// ===========================================================================
__device__ inline void philox4x32_10(uint32_t ctr[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(M0, ctr[0]), lo0 = M0 * ctr[0];
        uint32_t hi1 = __umulhi(M1, ctr[2]), lo1 = M1 * ctr[2];
        uint32_t n0 = hi1 ^ ctr[1] ^ k0, n1 = lo1, n2 = hi0 ^ ctr[3] ^ k1, n3 = lo0;
        ctr[0] = n0; ctr[1] = n1; ctr[2] = n2; ctr[3] = n3;
        k0 += W0; k1 += W1;
    }
}

// One Philox block -> two 53-bit uniforms -> one Box-Muller pair (z0, z1).
// Normal number e of the stream is z_{e&1} of block (offset + e) >> 1.
template <typename T>
__global__ __launch_bounds__(256) void k_fill_gaussian(Mat<T> out, uint64_t seed, uint64_t offset) {
    const int64_t total = out.rows * out.cols;
    const uint64_t first_pair = offset >> 1;
    const int64_t npairs = (int64_t)(((offset + (uint64_t)total + 1) >> 1) - first_pair);
    for (int64_t pi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; pi < npairs; pi += (int64_t)gridDim.x * blockDim.x) {
        uint64_t blk = first_pair + (uint64_t)pi;
        uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), 0u, 0u};
        philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
        uint64_t a = ((uint64_t)ctr[0] << 32) | ctr[1];
        uint64_t b = ((uint64_t)ctr[2] << 32) | ctr[3];
        // u1 in (0, 1], u2 in [0, 1)
        double u1 = ((double)(a >> 11) + 1.0) * (1.0 / 9007199254740992.0);
        double u2 = (double)(b >> 11) * (1.0 / 9007199254740992.0);
        double rad = sqrt(-2.0 * log(u1));
        double sn, cs;
        sincospi(2.0 * u2, &sn, &cs);
        double z[2] = {rad * cs, rad * sn};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int64_t e = (int64_t)(2 * blk + (uint64_t)h) - (int64_t)offset;
            if (e >= 0 && e < total) {
                int64_t i = e / out.cols, j = e - i * out.cols;
                out.at(i, j) = (T)z[h];
            }
        }
    }
}

// raw words of the same stream (word w = w_{w & 3} of block w >> 2): lets a test compare the integer generator bit for bit
__global__ __launch_bounds__(256) void k_philox_words(uint32_t *out, int64_t n, uint64_t seed, uint64_t word_offset) {
    const uint64_t first = word_offset >> 2;
    const int64_t nblk = (int64_t)(((word_offset + (uint64_t)n + 3) >> 2) - first);
    for (int64_t bi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; bi < nblk; bi += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t blk = first + (uint64_t)bi;
        uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), 0u, 0u};
        philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int64_t e = (int64_t)(4 * blk + (uint64_t)h) - (int64_t)word_offset;
            if (e >= 0 && e < n) out[e] = ctr[h];
        }
    }
}
void philox_words(rc_context *c, uint32_t *out, int64_t n, uint64_t seed, uint64_t word_offset) {
    if (n <= 0) return;
    int grid = (int)std::min<int64_t>(cdiv((n + 6) / 4, 256), 4096);
    hipLaunchKernelGGL(k_philox_words, dim3(grid), dim3(256), 0, c->stream, out, n, seed, word_offset);
}

template <typename T>
void fill_gaussian(rc_context *c, Mat<T> out, uint64_t seed, uint64_t offset) {
    if (out.empty()) return;
    int64_t npairs = (out.rows * out.cols + 2) / 2;
    int grid = (int)std::min<int64_t>(cdiv(npairs, 256), 4096);
    hipLaunchKernelGGL(k_fill_gaussian<T>, dim3(grid), dim3(256), 0, c->stream, out, seed, offset);
}

// ===========================================================================
// strided copy.  32x32 tiles through LDS so both sides coalesce whichever
// dimension is contiguous on each side.
// ===========================================================================
template <typename T>
__global__ __launch_bounds__(256) void k_copy_tiled(Mat<T> src, Mat<T> dst) {
    __shared__ T tile[32][33];
    const int64_t tr = (int64_t)blockIdx.y * 32, tc = (int64_t)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    // read with the thread-fast index along src's contiguous dimension
    const bool src_col_fast = (src.cs <= src.rs);  // columns contiguous in memory => j fastest
    for (int y = ty; y < 32; y += 8) {
        int64_t i = src_col_fast ? tr + y : tr + tx;
        int64_t j = src_col_fast ? tc + tx : tc + y;
        if (i < src.rows && j < src.cols) {
            if (src_col_fast) tile[y][tx] = src.at(i, j);
            else tile[tx][y] = src.at(i, j);
        }
    }
    __syncthreads();
    const bool dst_col_fast = (dst.cs <= dst.rs);
    for (int y = ty; y < 32; y += 8) {
        int64_t i = dst_col_fast ? tr + y : tr + tx;
        int64_t j = dst_col_fast ? tc + tx : tc + y;
        if (i < dst.rows && j < dst.cols) {
            dst.at(i, j) = dst_col_fast ? tile[y][tx] : tile[tx][y];
        }
    }
}

template <typename T>
void copy_mat(rc_context *c, Mat<T> src, Mat<T> dst) {
    RC_REQUIRE(src.rows == dst.rows && src.cols == dst.cols, RC_INVALID_ARGUMENT, "copy_mat: shape mismatch (%lld x %lld) vs (%lld x %lld)",
               (long long)src.rows, (long long)src.cols, (long long)dst.rows, (long long)dst.cols);
    if (src.empty()) return;
    dim3 grid((unsigned)cdiv(src.cols, 32), (unsigned)cdiv(src.rows, 32));
    hipLaunchKernelGGL(k_copy_tiled<T>, grid, dim3(256), 0, c->stream, src, dst);
}

// ---------------------------------------------------------------------------
// elementwise helpers (grid-stride over a logical row-major index; written for
// small matrices -- none of these is on the bandwidth-critical path)
// ---------------------------------------------------------------------------
template <typename T, int MODE>  // 0: zero, 1: identity
__global__ __launch_bounds__(256) void k_fill_const(Mat<T> dst) {
    const bool col_fast = (dst.cs <= dst.rs);
    const int64_t total = dst.rows * dst.cols;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t i, j;
        if (col_fast) { i = e / dst.cols; j = e - i * dst.cols; }
        else { j = e / dst.rows; i = e - j * dst.rows; }
        dst.at(i, j) = (MODE == 1 && i == j) ? (T)1 : (T)0;
    }
}
template <typename T>
void fill_identity(rc_context *c, Mat<T> dst) {
    if (dst.empty()) return;
    int grid = (int)std::min<int64_t>(cdiv(dst.rows * dst.cols, 256), 8192);
    hipLaunchKernelGGL((k_fill_const<T, 1>), dim3(grid), dim3(256), 0, c->stream, dst);
}
template <typename T>
void fill_zero(rc_context *c, Mat<T> dst) {
    if (dst.empty()) return;
    int grid = (int)std::min<int64_t>(cdiv(dst.rows * dst.cols, 256), 8192);
    hipLaunchKernelGGL((k_fill_const<T, 0>), dim3(grid), dim3(256), 0, c->stream, dst);
}

template <typename T>
__global__ __launch_bounds__(256) void k_scale_rows(const T *s, Mat<T> src, Mat<T> dst) {
    const bool col_fast = (dst.cs <= dst.rs);
    const int64_t total = dst.rows * dst.cols;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t i, j;
        if (col_fast) { i = e / dst.cols; j = e - i * dst.cols; }
        else { j = e / dst.rows; i = e - j * dst.rows; }
        dst.at(i, j) = s[i] * src.at(i, j);
    }
}
template <typename T>
void scale_rows(rc_context *c, const T *s, Mat<T> src, Mat<T> dst) {
    if (dst.empty()) return;
    int grid = (int)std::min<int64_t>(cdiv(dst.rows * dst.cols, 256), 8192);
    hipLaunchKernelGGL(k_scale_rows<T>, dim3(grid), dim3(256), 0, c->stream, s, src, dst);
}

// dst[:, j] = src[:, idx[j]]  (row gathers are expressed by transposed views)
// /root/reference/src/permutation.rs:100-139
template <typename T>
__global__ __launch_bounds__(256) void k_gather_cols(Mat<T> src, const int64_t *idx, Mat<T> dst, int *health) {
    const bool col_fast = (dst.cs <= dst.rs);
    const int64_t total = dst.rows * dst.cols;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t i, j;
        if (col_fast) { i = e / dst.cols; j = e - i * dst.cols; }
        else { j = e / dst.rows; i = e - j * dst.rows; }
        const int64_t s = idx[j];
        // an index outside the source (the reference would panic on it) must not become a wild address: zero + health bit 32
        if ((uint64_t)s < (uint64_t)src.cols) {
            dst.at(i, j) = src.at(i, s);
        } else {
            dst.at(i, j) = (T)0;
            if (i == 0) atomicOr(health, 32);
        }
    }
}
template <typename T>
void gather_cols(rc_context *c, Mat<T> src, const int64_t *idx, Mat<T> dst) {
    RC_REQUIRE(src.rows == dst.rows, RC_INVALID_ARGUMENT, "gather_cols: row mismatch");
    if (dst.empty()) return;
    int grid = (int)std::min<int64_t>(cdiv(dst.rows * dst.cols, 256), 8192);
    hipLaunchKernelGGL(k_gather_cols<T>, dim3(grid), dim3(256), 0, c->stream, src, idx, dst, c->health_word());
}

// inverse[perm[i]] = i   /root/reference/src/permutation.rs:28-38
__global__ __launch_bounds__(256) void k_invert_perm(const int64_t *perm, int64_t n, int64_t *inv) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t e = perm[i];
        if (e >= 0 && e < n) inv[e] = i;
    }
}
// Buffers are cleared / filled by kernels of the library on the context's stream, never by hipMemset*: a kernel node is captured into
// a hipGraph like every other node of the path, and its value is a kernel argument ("memset nodes").
__global__ __launch_bounds__(256) void k_fill_words(unsigned *p, size_t nwords, unsigned v) {
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        const size_t n4 = nwords / 4;
        uint4 *q = reinterpret_cast<uint4 *>(p);
        const uint4 v4 = make_uint4(v, v, v, v);
        for (size_t i = tid; i < n4; i += nt) q[i] = v4;
        for (size_t i = n4 * 4 + tid; i < nwords; i += nt) p[i] = v;
    } else {
        for (size_t i = tid; i < nwords; i += nt) p[i] = v;
    }
}
void fill_words(rc_context *c, void *p, size_t bytes, unsigned v) {
    if (bytes == 0) return;
    RC_REQUIRE((bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 3) == 0, RC_INVALID_ARGUMENT, "fill_words: %zu bytes at %p is not a whole number of aligned words", bytes, p);
    const size_t nwords = bytes / 4;
    const unsigned grid = (unsigned)std::min<size_t>((nwords / 4 + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(k_fill_words, dim3(grid), dim3(256), 0, c->stream, static_cast<unsigned *>(p), nwords, v);
}

void invert_perm(rc_context *c, const int64_t *perm, int64_t n, int64_t *inv) {
    if (n <= 0) return;
    // every entry starts at -1: an input that is not a permutation leaves -1 behind, which the gathers reject (health bit 32)
    // instead of following whatever the buffer held
    fill_words(c, inv, (size_t)n * sizeof(int64_t), 0xffffffffu);
    int grid = (int)std::min<int64_t>(cdiv(n, 256), 4096);
    hipLaunchKernelGGL(k_invert_perm, dim3(grid), dim3(256), 0, c->stream, perm, n, inv);
}
__global__ __launch_bounds__(256) void k_iota(int64_t *p, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = i;
}
void iota_i64(rc_context *c, int64_t *p, int64_t n) {
    if (n <= 0) return;
    int grid = (int)std::min<int64_t>(cdiv(n, 256), 4096);
    hipLaunchKernelGGL(k_iota, dim3(grid), dim3(256), 0, c->stream, p, n);
}

// ===========================================================================
// reductions
// ===========================================================================
template <typename T>
__device__ inline T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <typename T>
__device__ inline T wave_max(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}
// block-wide sum for 256 threads; result valid in every thread
template <typename T>
__device__ inline T block_sum_256(T v, T *sh /* >= 4 */) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// out[j] = sum_i a(i, j)^2 ; one workgroup per column.
// /root/reference/src/random_sampling.rs:184-191 (norm_l2 per column)
template <typename T>
__global__ __launch_bounds__(256) void k_col_sumsq(Mat<T> a, T *out) {
    __shared__ T sh[4];
    for (int64_t j = blockIdx.x; j < a.cols; j += gridDim.x) {
        T acc = 0;
        for (int64_t i = threadIdx.x; i < a.rows; i += 256) {
            T v = a.at(i, j);
            acc += v * v;
        }
        acc = block_sum_256(acc, sh);
        if (threadIdx.x == 0) out[j] = acc;
    }
}
template <typename T>
void col_sumsq(rc_context *c, Mat<T> a, T *out) {
    if (a.cols == 0) return;
    int grid = (int)std::min<int64_t>(a.cols, 65535);
    hipLaunchKernelGGL(k_col_sumsq<T>, dim3(grid), dim3(256), 0, c->stream, a, out);
}

template <typename T>
__global__ __launch_bounds__(256) void k_max_sqrt(const T *v, int64_t n, T *out) {
    __shared__ T sh[4];
    T m = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) m = max(m, v[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = sqrt(max(max(sh[0], sh[1]), max(sh[2], sh[3])));
}
template <typename T>
void max_sqrt(rc_context *c, const T *v, int64_t n, T *out) {
    hipLaunchKernelGGL(k_max_sqrt<T>, dim3(1), dim3(256), 0, c->stream, v, n, out);
}

// Frobenius pieces, deterministic two-stage: per-block partials then one block.
// /root/reference/src/types.rs:182-188
template <typename T>
__global__ __launch_bounds__(256) void k_fro_partial(Mat<T> a, Mat<T> b, T *partial) {
    __shared__ T sh[4];
    const bool col_fast = (b.cs <= b.rs);
    const int64_t total = a.rows * a.cols;
    T d2 = 0, b2 = 0;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t i, j;
        if (col_fast) { i = e / a.cols; j = e - i * a.cols; }
        else { j = e / a.rows; i = e - j * a.rows; }
        T x = a.at(i, j), y = b.at(i, j);
        d2 += (x - y) * (x - y);
        b2 += y * y;
    }
    d2 = block_sum_256(d2, sh);
    b2 = block_sum_256(b2, sh);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = d2; partial[2 * blockIdx.x + 1] = b2; }
}
template <typename T>
__global__ __launch_bounds__(256) void k_fro_final(const T *partial, int nblocks, T *out2) {
    __shared__ T sh[4];
    T d2 = 0, b2 = 0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { d2 += partial[2 * i]; b2 += partial[2 * i + 1]; }
    d2 = block_sum_256(d2, sh);
    b2 = block_sum_256(b2, sh);
    if (threadIdx.x == 0) { out2[0] = d2; out2[1] = b2; }
}
template <typename T>
void fro_diff(rc_context *c, Mat<T> a, Mat<T> b, T *out2) {
    RC_REQUIRE(a.rows == b.rows && a.cols == b.cols, RC_INVALID_ARGUMENT, "rel_diff_fro: shape mismatch");
    int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(a.rows * a.cols, 256 * 8), 1024));
    ArenaMark mark(c);
    T *partial = c->alloc<T>(2 * (size_t)grid);
    hipLaunchKernelGGL(k_fro_partial<T>, dim3(grid), dim3(256), 0, c->stream, a, b, partial);
    hipLaunchKernelGGL(k_fro_final<T>, dim3(1), dim3(256), 0, c->stream, partial, grid, out2);
}

// y -= corr   (adaptive sampler: Y = A Omega - q (b Omega), /root/reference/src/random_sampling.rs:266)
template <typename T>
__global__ __launch_bounds__(256) void k_sub_inplace(Mat<T> y, Mat<T> corr) {
    const bool col_fast = (y.cs <= y.rs);
    const int64_t total = y.rows * y.cols;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t i, j;
        if (col_fast) { i = e / y.cols; j = e - i * y.cols; }
        else { j = e / y.rows; i = e - j * y.rows; }
        y.at(i, j) -= corr.at(i, j);
    }
}
template <typename T>
void adaptive_residual_update(rc_context *c, Mat<T> y, Mat<T> corr) {
    if (y.empty()) return;
    int grid = (int)std::min<int64_t>(cdiv(y.rows * y.cols, 256), 8192);
    hipLaunchKernelGGL(k_sub_inplace<T>, dim3(grid), dim3(256), 0, c->stream, y, corr);
}

// explicit instantiations
#define RC_INST(T)                                                                    \
    template void fill_gaussian<T>(rc_context *, Mat<T>, uint64_t, uint64_t);         \
    template void copy_mat<T>(rc_context *, Mat<T>, Mat<T>);                          \
    template void fill_identity<T>(rc_context *, Mat<T>);                             \
    template void fill_zero<T>(rc_context *, Mat<T>);                                 \
    template void scale_rows<T>(rc_context *, const T *, Mat<T>, Mat<T>);             \
    template void gather_cols<T>(rc_context *, Mat<T>, const int64_t *, Mat<T>);      \
    template void col_sumsq<T>(rc_context *, Mat<T>, T *);                            \
    template void max_sqrt<T>(rc_context *, const T *, int64_t, T *);                 \
    template void fro_diff<T>(rc_context *, Mat<T>, Mat<T>, T *);                     \
    template void adaptive_residual_update<T>(rc_context *, Mat<T>, Mat<T>);
RC_INST(double)
RC_INST(float)
#undef RC_INST

}  // namespace rc
