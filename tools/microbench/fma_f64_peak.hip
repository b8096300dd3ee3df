// Microbenchmark: f64 vector FMA (v_fma_f64) and the 4x4x4 f64 MFMA shape vs the 16x16x4 shape.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double *out, int iters) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double a = threadIdx.x * 1e-9 + 1.0, b = threadIdx.x * 2e-9 - 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(a, acc[i], b);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma4(double *out, int iters) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0;
    double a = threadIdx.x * 1e-3 + 1.0, b = threadIdx.x * 2e-3 - 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma16(double *out, int iters) {
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3 + 1.0, b = threadIdx.x * 2e-3 - 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char *name, K kern, double *out, int blocks, int iters, double flops_per_thread_iter) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double flops = flops_per_thread_iter * iters * 256.0 * blocks;
    printf("%-30s blocks=%5d  %.3f ms  %.2f TFLOP/s\n", name, blocks, best, flops / best / 1e9);
}
int main() {
    double *od;
    hipMalloc(&od, 8 * 256 * 8192);
    for (int mult : {1, 2, 4, 8}) {
        run("v_fma_f64, 8 chains", k_fma<8>, od, 256 * mult, 20000, 2.0 * 8);
        run("v_fma_f64, 32 chains", k_fma<32>, od, 256 * mult, 5000, 2.0 * 32);
        // 4x4x4 (4 blocks): 4*4*4*4*2 = 512 flop per wave instruction = 8 per lane
        run("mfma_f64_4x4x4, 8 acc", k_mfma4<8>, od, 256 * mult, 20000, 8.0 * 8);
        // 16x16x4: 2048 flop per wave instruction = 32 per lane
        run("mfma_f64_16x16x4, 4 acc", k_mfma16<4>, od, 256 * mult, 20000, 32.0 * 4);
        run("mfma_f64_16x16x4, 8 acc", k_mfma16<8>, od, 256 * mult, 10000, 32.0 * 8);
    }
    return 0;
}
