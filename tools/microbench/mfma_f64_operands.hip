// Does the 4x4x4 f64 MFMA keep its rate when every instruction reads different A / B registers?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int TM, int TN>
__global__ __launch_bounds__(256) void k_grid(double *out, const double *in, int iters) {
    double acc[TM][TN], a[TM], b[TN];
    for (int i = 0; i < TM; ++i) a[i] = in[threadIdx.x + 64 * i];
    for (int j = 0; j < TN; ++j) b[j] = in[threadIdx.x + 64 * (TM + j)];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        // perturb operands a little so they stay live registers and are not hoisted into constants
        if (it == iters + 5) { for (int i = 0; i < TM; ++i) a[i] += 1.0; }
    }
    double s = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char *name, K kern, double *out, const double *in, int blocks, int iters, int nmfma) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, in, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double flops = 512.0 * nmfma * iters * 4.0 * blocks;
    printf("%-24s blocks=%5d  %.3f ms  %.2f TFLOP/s\n", name, blocks, best, flops / best / 1e9);
}
int main() {
    double *od, *in; hipMalloc(&od, 8 * 256 * 4096); hipMalloc(&in, 8 * 64 * 64); hipMemset(in, 0, 8 * 64 * 64);
    for (int mult : {1, 2}) {
        run("grid 1x8 (A shared)", k_grid<1, 8>, od, in, 256 * mult, 20000, 8);
        run("grid 2x4", k_grid<2, 4>, od, in, 256 * mult, 20000, 8);
        run("grid 4x4", k_grid<4, 4>, od, in, 256 * mult, 10000, 16);
        run("grid 8x9", k_grid<8, 9>, od, in, 256 * mult, 2000, 72);
        run("grid 4x9", k_grid<4, 9>, od, in, 256 * mult, 4000, 36);
    }
    return 0;
}
