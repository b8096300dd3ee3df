// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 rate with operands in
// registers (no memory traffic): the ceiling any f64 GEMM on this chip can reach.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_f64(double *out, int iters) {
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
template <int NACC>
__global__ __launch_bounds__(256) void k_f32(float *out, int iters) {
    float4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = float4_t{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f + 1.0f, b = threadIdx.x * 2e-3f - 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K, typename T>
void run(const char *name, K kern, T *out, int blocks, int iters, int nacc) {
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
    double flops = 2.0 * 16 * 16 * 4 * (double)nacc * iters * 4.0 * blocks;  // 4 waves per block
    printf("%-28s blocks=%d waves/SIMD=%d  %.3f ms  %.2f TFLOP/s\n", name, blocks, blocks / 256, best, flops / best / 1e9);
}
int main() {
    double *od; float *of;
    hipMalloc(&od, 8 * 256 * 4096); hipMalloc(&of, 4 * 256 * 4096);
    for (int mult : {1, 2, 4}) {
        run("f64 16x16x4, 4 acc", k_f64<4>, od, 256 * mult, 20000, 4);
        run("f64 16x16x4, 16 acc", k_f64<16>, od, 256 * mult, 5000, 16);
        run("f32 16x16x4, 16 acc", k_f32<16>, of, 256 * mult, 5000, 16);
    }
    return 0;
}
