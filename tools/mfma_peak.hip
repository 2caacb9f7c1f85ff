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

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// The inner loop of the split-precision kernels without any staging: per "tap" NREAD conflict-free ds_read_b128 fragment reads feed 12 MFMAs
// (2 x 2 accumulators x 3 products), 4 waves per workgroup.  What does the matrix pipe sustain when its operands arrive from LDS?
template <int NREAD, bool RANDOM = false>
__global__ __launch_bounds__(256, 2) void mfma_lds_loop(float* out, int iters) {
    __shared__ u32x4 L[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) {
        if (RANDOM) {                                              // random signs and mantissas, exponents around 1.0: the bit activity of real operands
            unsigned x = (i + 1) * 2654435761u + blockIdx.x * 40503u;
            u32x4 v;
            for (int k = 0; k < 4; ++k) {
                x ^= x << 13, x ^= x >> 17, x ^= x << 5;
                v[k] = (x & 0x80ff80ffu) | 0x3f003f00u;
            }
            L[i] = v;
        } else {
            L[i] = u32x4{0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        }
    }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 4; ++i)
        for (int v = 0; v < 16; ++v) acc[i >> 1][i & 1][v] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4* base = L + wave * 64 + lane;                      // 64 consecutive 16-byte units per read: conflict-free
    bf16x8 f[8];
    for (int j = 0; j < 8; ++j) f[j] = __builtin_bit_cast(bf16x8, base[256 * j]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int j = 0; j < NREAD; ++j) f[j & 7] = __builtin_bit_cast(bf16x8, base[256 * (j & 7) + ((it + tap) & 7) * 2048 / 8]);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2 + mi], f[4 + 2 * ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[mi], f[5 + 2 * ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[mi], f[4 + 2 * ni], acc[mi][ni], 0, 0, 0);
                }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int v = 0; v < 16; ++v) s += acc[i >> 1][i & 1][v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float f32x4v __attribute__((ext_vector_type(4)));

// The same wave tile (64 x 64 outputs, 3 split-precision products) on v_mfma_f32_16x16x32_bf16: per k = 32 slab 16 conflict-free
// ds_read_b128 (4 A-hi, 4 A-lo, 4 B-hi, 4 B-lo fragments) feed 48 MFMAs of 16 cycles -- the same LDS bytes and the same matrix-pipe
// cycles per FLOP as 2 x (8 reads, 12 MFMAs of 32 cycles).  MI355X_MICROARCH.md "DVFS give-back" item 7: the chip may hold a higher
// clock on this shape.
template <bool RANDOM>
__global__ __launch_bounds__(256, 2) void mfma16_lds_loop(float* out, int iters) {
    __shared__ u32x4 L[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) {
        if (RANDOM) {
            unsigned x = (i + 1) * 2654435761u + blockIdx.x * 40503u;
            u32x4 v;
            for (int k = 0; k < 4; ++k) {
                x ^= x << 13, x ^= x >> 17, x ^= x << 5;
                v[k] = (x & 0x80ff80ffu) | 0x3f003f00u;
            }
            L[i] = v;
        } else {
            L[i] = u32x4{0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        }
    }
    __syncthreads();
    f32x4v acc[4][4];
    for (int i = 0; i < 16; ++i)
        for (int v = 0; v < 4; ++v) acc[i >> 2][i & 3][v] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4* base = L + wave * 64 + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int slab = 0; slab < 5; ++slab) {                     // 5 slabs of k = 32 ~ 9 taps of k = 16 (4.5): same order of work per iteration
            bf16x8 f[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) f[j] = __builtin_bit_cast(bf16x8, base[256 * (j & 7) + ((it + slab + (j >> 3)) & 7) * 2048 / 8]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[4 + mi], f[8 + ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[mi], f[12 + ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[mi], f[8 + ni], acc[mi][ni], 0, 0, 0);
                }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i)
        for (int v = 0; v < 4; ++v) s += acc[i >> 2][i & 3][v];
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
    auto runl = [&](auto kern, int nread, int wgs_per_cu) {
        const int grid = 256 * wgs_per_cu, iters = 20000;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 * iters * 9 * 12 * 32768.0;
        printf("LDS-fed: %d ds_read_b128 per 12 MFMAs, wgs/CU=%d  %.2f ms  %.1f TFLOP/s\n", nread, wgs_per_cu, ms, flops / ms / 1e9);
    };
    for (int w = 1; w <= 2; ++w) {
        runl(mfma_lds_loop<0>, 0, w);
        runl(mfma_lds_loop<4>, 4, w);
        runl(mfma_lds_loop<8>, 8, w);
        printf("random operands: ");
        runl(mfma_lds_loop<8, true>, 8, w);
    }
    auto runl16 = [&](auto kern, const char* tag, int wgs_per_cu) {
        const int grid = 256 * wgs_per_cu, iters = 20000;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 * iters * 5 * 48 * 16384.0;
        printf("16x16x32 LDS-fed (%s): 16 ds_read_b128 per 48 MFMAs, wgs/CU=%d  %.2f ms  %.1f TFLOP/s\n", tag, wgs_per_cu, ms, flops / ms / 1e9);
    };
    for (int rep = 0; rep < 3; ++rep)                               // interleaved rounds in one process (same device, same thermal state)
        for (int w = 1; w <= 2; ++w) {
            printf("random operands: ");
            runl(mfma_lds_loop<8, true>, 8, w);
            runl16(mfma16_lds_loop<true>, "random", w);
            runl16(mfma16_lds_loop<false>, "constant", w);
        }
    for (int w = 1; w <= 2; ++w) {
        run(mfma_bf16_loop<1>, 1, w);
        run(mfma_bf16_loop<2>, 2, w);
        run(mfma_bf16_loop<3>, 3, w);
        run(mfma_bf16_loop<4>, 4, w);
    }
    return 0;
}
