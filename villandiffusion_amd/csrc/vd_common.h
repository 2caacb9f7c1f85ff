// Shared helpers for the gfx950 kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "villan_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void vd_set_error(const char* fmt, ...);
// vd_norm.hip: VD_ETIMEDOUT (with the message set) while an asynchronous GroupNorm poll timeout is pending (vd_async_errors), else 0 -- the check at the
// top of EVERY vd_groupnorm_* entry point, the pre-split producers of vd_presplit.hip included
int vd_gn_sticky(const char* who);

#define VD_REQUIRE(cond, ...)              \
    do {                                   \
        if (!(cond)) {                     \
            vd_set_error(__VA_ARGS__);     \
            return VD_EINVAL;              \
        }                                  \
    } while (0)

#define VD_LAUNCH_CHECK(name)                                               \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            vd_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return (int)e__;                                                \
        }                                                                   \
    } while (0)

// vd_gemm_desc.debug: timing-only ablation bits (wrong results) exist in `make ABLATION=1` builds only; the release library rejects a non-zero value
// in vd_gemm() and compiles the bit tests away.
#ifdef VD_ABLATION
#define VD_DBG(d) ((d).debug)
#else
#define VD_DBG(d) 0
#endif

static inline int vd_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// 64-lane wave sum via DPP-free shuffles (wavefront = 64 on CDNA).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread. red: >= 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// 1 / (1 + e^-z) with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of the IEEE division sequence (v_div_scale / v_rcp / 4 FMAs / v_div_fmas /
// v_div_fixup): the GroupNorm-folding convolution loader evaluates it per staged element.  VD_EXACT_SIGMOID restores the division.
#ifdef VD_EXACT_SIGMOID
__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + __expf(-z)); }
#else
__device__ __forceinline__ float sigmoidf_(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
#endif
