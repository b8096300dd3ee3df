// Shared between the GEMM translation units (kernels_gemm.hip: shape dispatch, split-K policy and the general kernels;
// kernels_gemm_pipe.hip: the software-pipelined f64 main loop of the two big products).
#pragma once
#include "rc_common.hpp"

namespace rc {

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

// Software-pipelined f64 kernel (kernels_gemm_pipe.hip).  Launches the GEMM kernel for an ALREADY split problem (tiles_m,
// tiles_n, kchunk, splits, partial filled in by the caller) if an instantiation for this tile shape exists and the operands
// meet its preconditions (16-byte vector staging, K chunks that are multiples of the K tile); returns false otherwise.
bool gemm_f64p_launch(rc_context *c, const GemmArgs<double> &g, int alay, int blay, int bm, int bn, int bk, int wm, int wn, int orient, bool vec2);

}  // namespace rc
