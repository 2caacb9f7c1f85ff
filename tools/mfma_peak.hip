// Micro-benchmark: sustained rate of v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_bf16 on this chip from registers only (no LDS,
// no memory): what the matrix pipe sustains under continuous issue, beside the guide's 157.3 / 2500 TFLOP/s peaks.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int v = 0; v < 16; ++v) s += acc[i][v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_bf16_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)(a0 + threadIdx.x * 1e-3f + j);
        b[j] = (__bf16)(b0 - threadIdx.x * 1e-3f - j);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int v = 0; v < 16; ++v) s += acc[i][v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 4096 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu) {
        const int grid = 256 * wgs_per_cu, iters = 20000;
        hipLaunchKernelGGL(mfma_loop<4>, dim3(grid), dim3(256), 0, 0, out, 100, 1.0f, 2.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 /*waves*/ * iters * 16 /*mfma per iter*/ * 4096.0;
        printf("wgs/CU=%d  %.2f ms  %.1f TFLOP/s\n", wgs_per_cu, ms, flops / ms / 1e9);
    }
    for (int wgs_per_cu = 1; wgs_per_cu <= 4; ++wgs_per_cu) {
        for (int rep = 0; rep < 2; ++rep) {
            const int grid = 256 * wgs_per_cu, iters = rep ? 200000 : 20000;   // ~10 ms and ~100 ms: does the clock hold?
            hipLaunchKernelGGL(mfma_bf16_loop<4>, dim3(grid), dim3(256), 0, 0, out, 100, 1.0f, 2.0f);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_bf16_loop<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)grid * 4 /*waves*/ * iters * 16 /*mfma per iter*/ * 32768.0;
            printf("bf16 32x32x16: wgs/CU=%d  %.2f ms  %.1f TFLOP/s\n", wgs_per_cu, ms, flops / ms / 1e9);
        }
    }
    // dependent-issue cost: NACC independent accumulators per wave, round-robin (NACC = 1: every MFMA waits for the previous one)
    auto run = [&](auto kern, int nacc, int wgs_per_cu) {
        const int grid = 256 * wgs_per_cu, iters = 20000;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 100, 1.0f, 2.0f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 * iters * 4 * nacc * 32768.0;
        printf("bf16 32x32x16: %d accumulators round-robin, wgs/CU=%d  %.2f ms  %.1f TFLOP/s\n", nacc, wgs_per_cu, ms, flops / ms / 1e9);
    };
    for (int w = 1; w <= 2; ++w) {
        run(mfma_bf16_loop<1>, 1, w);
        run(mfma_bf16_loop<2>, 2, w);
        run(mfma_bf16_loop<3>, 3, w);
        run(mfma_bf16_loop<4>, 4, w);
    }
    return 0;
}
