// K1: GroupNorm (+SiLU) forward / backward for NCHW tensors with a batch stride.
//
// In NCHW one (batch item, group) is ONE contiguous slab of cpg*HW floats, so a workgroup streams its slab with
// float4 loads, reduces with wave shuffles (wave = 64) and re-reads the slab (L2-resident: <= 48 KB for the
// CIFAR10 UNet) for the normalise pass.  HBM-bound: algorithmic traffic = read x + write y (8 B / element).
//
// Replaces F.group_norm(+F.silu) of ResnetBlock2D.norm1/norm2, AttentionBlock.group_norm and conv_norm_out in the
// diffusers UNet2DModel the reference trains (loss.py:993) and samples from (VillanDiffusion.py:579).
#include "vd_common.h"
#include <atomic>
#include <stdlib.h>

namespace {

// Register-resident variant: the whole (batch item, group) slab (<= NV float4 per thread) is read ONCE, kept in
// registers for the mean / variance / normalise passes and written once: HBM traffic = the algorithmic 8 B / element.
// Block-wide sum for blockDim.x == NTH (NTH / 64 waves, fixed order); result valid in every thread. red: >= NTH / 64 floats of LDS.
template <int NTH>
__device__ __forceinline__ float block_sum_t(float v, float* red) {
    if (NTH == 256) return block_sum_256(v, red);
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < NTH / 64; ++k) a += red[k];
    return a;
}

// NTH = 512 / 1024 (round 4): groups of up to 28 672 elements (7 float4 per thread) -- the 14 336- and 28 672-element groups of BASELINE config #5's
// 32x32 / 64x64 levels -- stay register-resident (2 tensor sweeps instead of the chunked kernels' 3; backward 3 instead of 5).
template <int NV, int NTH = 256>
__global__ __launch_bounds__(NTH) void gn_fwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out, int C,
                                                         int HW, int G, float eps, int apply_silu, int64_t x_bs, int64_t y_bs,
                                                         float* __restrict__ ss_out) {
    __shared__ float red[2 * (NTH / 64 < 4 ? 4 : NTH / 64)];
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int cpg = C / G;
    const int n4 = (cpg * HW) >> 2;
    const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x + (int64_t)b * x_bs + (int64_t)g * cpg * HW);
    f32x4* __restrict__ y4 = reinterpret_cast<f32x4*>(y + (int64_t)b * y_bs + (int64_t)g * cpg * HW);
    const int tid = threadIdx.x;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * NTH;
        v[i] = (idx < n4) ? x4[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float inv_n = 1.f / (float)(cpg * HW);
    const float mean = block_sum_t<NTH>(s, red) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (tid + i * NTH < n4) {
            const float a0 = v[i][0] - mean, a1 = v[i][1] - mean, a2 = v[i][2] - mean, a3 = v[i][3] - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float rstd = rsqrtf(block_sum_t<NTH>(q, red + (NTH / 64 < 4 ? 4 : NTH / 64)) * inv_n + eps);
    if (tid == 0) {
        mean_out[blockIdx.x] = mean;
        rstd_out[blockIdx.x] = rstd;
    }
    if (ss_out) {        // statistics-only pass: per-channel scale / shift for the convolution that folds GN + SiLU into its loader
        if (tid < cpg) {
            const int c = g * cpg + tid;
            const float ga = gamma[c] * rstd, be = beta[c] - mean * ga;
            ss_out[((int64_t)b * C + c) * 2] = ga;
            ss_out[((int64_t)b * C + c) * 2 + 1] = be;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * NTH;
        if (idx < n4) {
            const int c = g * cpg + (idx * 4) / HW;
            const float ga = gamma[c] * rstd, be = beta[c] - mean * ga;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = v[i][j] * ga + be;
                o[j] = apply_silu ? z * sigmoidf_(z) : z;
            }
            y4[idx] = o;
        }
    }
}

// Small slabs (<= 1024 elements: the 8x8 / 4x4 stages): ONE WAVE per (batch item, group), four groups per workgroup -- no LDS, no barrier; the
// 256-thread kernel above leaves 128 .. 224 of its threads without an element there and pays two block reductions (7.4 us per launch for 8 MB).
template <int NV>
__global__ __launch_bounds__(256) void gn_fwd_wave_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y,
                                                          float* __restrict__ mean_out, float* __restrict__ rstd_out, int C,
                                                          int HW, int G, float eps, int apply_silu, int64_t x_bs, int64_t y_bs,
                                                          float* __restrict__ ss_out, int total) {
    const int lane = threadIdx.x & 63;
    const int gid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gid >= total) return;
    const int b = gid / G, g = gid - b * G;
    const int cpg = C / G;
    const int n4 = (cpg * HW) >> 2;
    const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x + (int64_t)b * x_bs + (int64_t)g * cpg * HW);
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = lane + i * 64;
        v[i] = (idx < n4) ? x4[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float inv_n = 1.f / (float)(cpg * HW);
    const float mean = wave_sum(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + i * 64 < n4) {
            const float a0 = v[i][0] - mean, a1 = v[i][1] - mean, a2 = v[i][2] - mean, a3 = v[i][3] - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) * inv_n + eps);
    if (lane == 0) {
        mean_out[gid] = mean;
        rstd_out[gid] = rstd;
    }
    if (ss_out) {
        for (int cl = lane; cl < cpg; cl += 64) {
            const int c = g * cpg + cl;
            const float ga = gamma[c] * rstd, be = beta[c] - mean * ga;
            ss_out[((int64_t)b * C + c) * 2] = ga;
            ss_out[((int64_t)b * C + c) * 2 + 1] = be;
        }
        return;
    }
    f32x4* __restrict__ y4 = reinterpret_cast<f32x4*>(y + (int64_t)b * y_bs + (int64_t)g * cpg * HW);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = lane + i * 64;
        if (idx < n4) {
            const int c = g * cpg + (idx * 4) / HW;
            const float ga = gamma[c] * rstd, be = beta[c] - mean * ga;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = v[i][j] * ga + be;
                o[j] = apply_silu ? z * sigmoidf_(z) : z;
            }
            y4[idx] = o;
        }
    }
}

__global__ __launch_bounds__(256) void gn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out, int C, int HW,
                                                     int G, float eps, int apply_silu, int64_t x_bs, int64_t y_bs) {
    __shared__ float red[8];
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int cpg = C / G;
    const int n = cpg * HW;
    const float* __restrict__ xs = x + (int64_t)b * x_bs + (int64_t)g * n;
    float* __restrict__ ys = y + (int64_t)b * y_bs + (int64_t)g * n;
    const int tid = threadIdx.x;
    const bool vec = ((HW & 3) == 0) && ((((uintptr_t)xs) & 15) == 0) && ((((uintptr_t)ys) & 15) == 0);

    // pass 1: mean
    float s = 0.f;
    if (vec) {
        const f32x4* x4 = reinterpret_cast<const f32x4*>(xs);
        for (int i = tid; i < (n >> 2); i += 256) {
            f32x4 v = x4[i];
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
    } else {
        for (int i = tid; i < n; i += 256) s += xs[i];
    }
    const float mean = block_sum_256(s, red) / (float)n;
    // pass 2: variance about the mean (two-pass: no cancellation)
    float q = 0.f;
    if (vec) {
        const f32x4* x4 = reinterpret_cast<const f32x4*>(xs);
        for (int i = tid; i < (n >> 2); i += 256) {
            f32x4 v = x4[i];
            const float a0 = v[0] - mean, a1 = v[1] - mean, a2 = v[2] - mean, a3 = v[3] - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    } else {
        for (int i = tid; i < n; i += 256) {
            const float a = xs[i] - mean;
            q += a * a;
        }
    }
    const float var = block_sum_256(q, red + 4) / (float)n;
    const float rstd = rsqrtf(var + eps);
    if (tid == 0) {
        mean_out[blockIdx.x] = mean;
        rstd_out[blockIdx.x] = rstd;
    }
    // pass 3: normalise + affine (+ SiLU)
    if (vec) {
        const f32x4* x4 = reinterpret_cast<const f32x4*>(xs);
        f32x4* y4 = reinterpret_cast<f32x4*>(ys);
        for (int i = tid; i < (n >> 2); i += 256) {
            const int c = g * cpg + (i * 4) / HW;
            const float ga = gamma[c] * rstd, be = beta[c] - mean * ga;
            f32x4 v = x4[i], o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = v[j] * ga + be;
                o[j] = apply_silu ? z * sigmoidf_(z) : z;
            }
            y4[i] = o;
        }
    } else {
        for (int i = tid; i < n; i += 256) {
            const int c = g * cpg + i / HW;
            const float z = (xs[i] - mean) * rstd * gamma[c] + beta[c];
            ys[i] = apply_silu ? z * sigmoidf_(z) : z;
        }
    }
}

// dz = dy * silu'(z) ; dgamma_ws[b][c] = sum dz*xhat ; dbeta_ws[b][c] = sum dz ;
// dx = rstd * (dz*gamma - mean_g(dz*gamma) - xhat * mean_g(dz*gamma*xhat)) (+ extra)
__global__ __launch_bounds__(256) void gn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ extra, float* __restrict__ dx,
                                                     float* __restrict__ dgamma_ws, float* __restrict__ dbeta_ws, int C, int HW,
                                                     int G, int apply_silu, int64_t dy_bs, int64_t x_bs, int64_t ex_bs,
                                                     int64_t dx_bs) {
    __shared__ float ch_s1[64], ch_s2[64];  // per-channel sums of this group (cpg <= 64)
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int cpg = C / G;
    const int n = cpg * HW;
    const int64_t goff = (int64_t)g * n;
    const float* __restrict__ xs = x + (int64_t)b * x_bs + goff;
    const float* __restrict__ dys = dy + (int64_t)b * dy_bs + goff;
    const float* __restrict__ exs = extra ? extra + (int64_t)b * ex_bs + goff : nullptr;
    float* __restrict__ dxs = dx + (int64_t)b * dx_bs + goff;
    const float mean = mean_in[blockIdx.x], rstd = rstd_in[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // pass 1: one wave per channel, fixed summation order -> deterministic per-channel sums
    for (int cl = wave; cl < cpg; cl += 4) {
        const int c = g * cpg + cl;
        const float ga = gamma[c], be = beta[c];
        const float* __restrict__ xc = xs + (int64_t)cl * HW;
        const float* __restrict__ dc = dys + (int64_t)cl * HW;
        float s1 = 0.f, s2 = 0.f;
        for (int i = lane; i < HW; i += 64) {
            const float xh = (xc[i] - mean) * rstd;
            float dz = dc[i];
            if (apply_silu) {
                const float z = xh * ga + be, sg = sigmoidf_(z);
                dz *= sg * (1.f + z * (1.f - sg));
            }
            s1 += dz;
            s2 += dz * xh;
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) {
            ch_s1[cl] = s1;
            ch_s2[cl] = s2;
            dbeta_ws[(int64_t)b * C + c] = s1;
            dgamma_ws[(int64_t)b * C + c] = s2;
        }
    }
    __syncthreads();
    float m1 = 0.f, m2 = 0.f;
    for (int cl = 0; cl < cpg; ++cl) {
        const float ga = gamma[g * cpg + cl];
        m1 += ga * ch_s1[cl];
        m2 += ga * ch_s2[cl];
    }
    const float inv_n = 1.f / (float)n;
    m1 *= inv_n;
    m2 *= inv_n;

    // pass 2: dx
    const bool vec = ((HW & 3) == 0) && ((((uintptr_t)xs) & 15) == 0) && ((((uintptr_t)dys) & 15) == 0) &&
                     ((((uintptr_t)dxs) & 15) == 0) && (!exs || ((((uintptr_t)exs) & 15) == 0));
    if (vec) {
        const f32x4* x4 = reinterpret_cast<const f32x4*>(xs);
        const f32x4* d4 = reinterpret_cast<const f32x4*>(dys);
        const f32x4* e4 = reinterpret_cast<const f32x4*>(exs);
        f32x4* o4 = reinterpret_cast<f32x4*>(dxs);
        for (int i = tid; i < (n >> 2); i += 256) {
            const int c = g * cpg + (i * 4) / HW;
            const float ga = gamma[c], be = beta[c];
            f32x4 xv = x4[i], dv = d4[i], o;
            f32x4 ev = {0.f, 0.f, 0.f, 0.f};
            if (exs) ev = e4[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (xv[j] - mean) * rstd;
                float dz = dv[j];
                if (apply_silu) {
                    const float z = xh * ga + be, sg = sigmoidf_(z);
                    dz *= sg * (1.f + z * (1.f - sg));
                }
                o[j] = rstd * (dz * ga - m1 - xh * m2) + ev[j];
            }
            o4[i] = o;
        }
    } else {
        for (int i = tid; i < n; i += 256) {
            const int c = g * cpg + i / HW;
            const float ga = gamma[c], be = beta[c];
            const float xh = (xs[i] - mean) * rstd;
            float dz = dys[i];
            if (apply_silu) {
                const float z = xh * ga + be, sg = sigmoidf_(z);
                dz *= sg * (1.f + z * (1.f - sg));
            }
            dxs[i] = rstd * (dz * ga - m1 - xh * m2) + (exs ? exs[i] : 0.f);
        }
    }
}

// Register-resident backward: x and dy of the (batch item, group) slab are read ONCE (float4 per thread x NV), turned into
// xhat / dz in registers, reduced to the per-channel sums (deterministic: wave butterflies + fixed-order LDS combine) and
// reused for dx -- 3 tensor reads (x, dy, extra) + 1 write instead of 5 + 1.
// L = float4 per channel (HW/4): L >= 64 (multiple of 64): every (chunk, wave) lies in one channel; L < 64 (power of two):
// channels are L-lane segments of a wave.
// NTH = 512 for the large groups (slab > 4096 elements): half the registers per lane (122 / 170 VGPRs at NV = 8 / 12 allowed only 4 / 2 waves per
// SIMD, and with two resident workgroups per CU nothing overlapped a workgroup's reduce / write phase: 4.4 / 3.4 TB/s against 5.5 for NV <= 4).
template <int NV, int NTH = 256>
__global__ __launch_bounds__(NTH) void gn_bwd_reg_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ extra, float* __restrict__ dx,
                                                         float* __restrict__ dgamma_ws, float* __restrict__ dbeta_ws, int C,
                                                         int HW, int G, int apply_silu, int64_t dy_bs, int64_t x_bs,
                                                         int64_t ex_bs, int64_t dx_bs, const float* __restrict__ extra2,
                                                         int64_t ex2_bs, float* __restrict__ rs_out, int64_t rs_ld) {
    __shared__ float ch_s1[64], ch_s2[64], ch_s3[64];
    __shared__ float part1[NV * (NTH / 64)], part2[NV * (NTH / 64)], part3[NV * (NTH / 64)];
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int cpg = C / G;
    const int n4 = (cpg * HW) >> 2;
    const int L = HW >> 2;
    const int64_t goff = (int64_t)g * cpg * HW;
    const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x + (int64_t)b * x_bs + goff);
    const f32x4* __restrict__ d4 = reinterpret_cast<const f32x4*>(dy + (int64_t)b * dy_bs + goff);
    const f32x4* __restrict__ e4 = extra ? reinterpret_cast<const f32x4*>(extra + (int64_t)b * ex_bs + goff) : nullptr;
    const f32x4* __restrict__ f4 = extra2 ? reinterpret_cast<const f32x4*>(extra2 + (int64_t)b * ex2_bs + goff) : nullptr;
    f32x4* __restrict__ o4 = reinterpret_cast<f32x4*>(dx + (int64_t)b * dx_bs + goff);
    const float mean = mean_in[blockIdx.x], rstd = rstd_in[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    f32x4 xh[NV], dz[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * NTH;
        const bool in = idx < n4;
        const f32x4 xv = in ? x4[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 dv = in ? d4[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = g * cpg + (in ? idx / L : 0);
        const float ga = gamma[c], be = beta[c];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float h_ = (xv[j] - mean) * rstd;
            float z_ = dv[j];
            if (apply_silu) {
                const float z = h_ * ga + be, sg = sigmoidf_(z);
                z_ *= sg * (1.f + z * (1.f - sg));
            }
            if (!in) z_ = 0.f;
            xh[i][j] = h_;
            dz[i][j] = z_;
            s1 += z_;
            s2 += z_ * h_;
        }
        if (L >= 64) {
            s1 = wave_sum(s1);
            s2 = wave_sum(s2);
            if (lane == 0) {
                part1[i * (NTH / 64) + wave] = s1;
                part2[i * (NTH / 64) + wave] = s2;
            }
        } else {
            for (int off = 1; off < L; off <<= 1) {          // butterfly inside the L-lane channel segment
                s1 += __shfl_xor(s1, off, 64);
                s2 += __shfl_xor(s2, off, 64);
            }
            if (in && (lane & (L - 1)) == 0) {
                ch_s1[idx / L] = s1;
                ch_s2[idx / L] = s2;
            }
        }
    }
    __syncthreads();
    if (L >= 64) {
        if (tid < cpg) {                                      // channel tid = parts [tid*L/64, (tid+1)*L/64), fixed order
            const int np = L >> 6;
            float a1 = 0.f, a2 = 0.f;
            for (int k = 0; k < np; ++k) {
                a1 += part1[tid * np + k];
                a2 += part2[tid * np + k];
            }
            ch_s1[tid] = a1;
            ch_s2[tid] = a2;
        }
        __syncthreads();
    }
    if (tid < cpg) {
        dbeta_ws[(int64_t)b * C + g * cpg + tid] = ch_s1[tid];
        dgamma_ws[(int64_t)b * C + g * cpg + tid] = ch_s2[tid];
    }
    float m1 = 0.f, m2 = 0.f;
    for (int cl = 0; cl < cpg; ++cl) {
        const float ga = gamma[g * cpg + cl];
        m1 += ga * ch_s1[cl];
        m2 += ga * ch_s2[cl];
    }
    const float inv_n = 1.f / (float)(cpg * HW);
    m1 *= inv_n;
    m2 *= inv_n;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * NTH;
        const bool in = idx < n4;
        float s3 = 0.f;
        if (in) {
            const float ga = gamma[g * cpg + idx / L];
            f32x4 ev = {0.f, 0.f, 0.f, 0.f}, o;
            if (e4) ev = e4[idx];
            if (f4) {
                const f32x4 fv = f4[idx];
#pragma unroll
                for (int j = 0; j < 4; ++j) ev[j] += fv[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = rstd * (dz[i][j] * ga - m1 - xh[i][j] * m2) + ev[j];
                s3 += o[j];
            }
            o4[idx] = o;
        }
        if (rs_out) {                                         // per-channel sums of the dx just written (the consumer's bias gradient rows)
            if (L >= 64) {
                s3 = wave_sum(s3);
                if (lane == 0) part3[i * (NTH / 64) + wave] = s3;
            } else {
                for (int off = 1; off < L; off <<= 1) s3 += __shfl_xor(s3, off, 64);
                if (in && (lane & (L - 1)) == 0) ch_s3[idx / L] = s3;
            }
        }
    }
    if (rs_out) {
        __syncthreads();
        if (tid < cpg) {
            float a3;
            if (L >= 64) {
                const int np = L >> 6;
                a3 = 0.f;
                for (int k = 0; k < np; ++k) a3 += part3[tid * np + k];
            } else {
                a3 = ch_s3[tid];
            }
            rs_out[(int64_t)b * rs_ld + g * cpg + tid] = a3;
        }
    }
}

// Backward for small slabs (<= 1024 elements, a channel = L = HW/4 < 64 lanes, L a power of two: the 8x8 / 4x4 stages): one WAVE per (batch item,
// group), four groups per workgroup, no LDS and no barrier (gn_bwd_reg_kernel<1, 256>: 10 us per launch for 8-25 MB).  Channels are L-lane segments:
// per-channel sums by butterflies inside the segment, the two group sums m1 = sum ga*dz, m2 = sum ga*dz*xhat by one wave reduction each.
template <int NV>
__global__ __launch_bounds__(256) void gn_bwd_wave_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ extra, float* __restrict__ dx,
                                                          float* __restrict__ dgamma_ws, float* __restrict__ dbeta_ws, int C,
                                                          int HW, int G, int apply_silu, int64_t dy_bs, int64_t x_bs,
                                                          int64_t ex_bs, int64_t dx_bs, const float* __restrict__ extra2,
                                                          int64_t ex2_bs, float* __restrict__ rs_out, int64_t rs_ld, int total) {
    const int lane = threadIdx.x & 63;
    const int gid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gid >= total) return;
    const int b = gid / G, g = gid - b * G;
    const int cpg = C / G;
    const int n4 = (cpg * HW) >> 2;
    const int L = HW >> 2;
    const int64_t goff = (int64_t)g * cpg * HW;
    const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x + (int64_t)b * x_bs + goff);
    const f32x4* __restrict__ d4 = reinterpret_cast<const f32x4*>(dy + (int64_t)b * dy_bs + goff);
    const f32x4* __restrict__ e4 = extra ? reinterpret_cast<const f32x4*>(extra + (int64_t)b * ex_bs + goff) : nullptr;
    const f32x4* __restrict__ f4 = extra2 ? reinterpret_cast<const f32x4*>(extra2 + (int64_t)b * ex2_bs + goff) : nullptr;
    f32x4* __restrict__ o4 = reinterpret_cast<f32x4*>(dx + (int64_t)b * dx_bs + goff);
    const float mean = mean_in[gid], rstd = rstd_in[gid];

    f32x4 xh[NV], dz[NV];
    float gav[NV];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = lane + i * 64;
        const bool in = idx < n4;
        const f32x4 xv = in ? x4[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 dv = in ? d4[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = g * cpg + (in ? idx / L : 0);
        const float ga = gamma[c], be = beta[c];
        gav[i] = ga;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float h_ = (xv[j] - mean) * rstd;
            float z_ = dv[j];
            if (apply_silu) {
                const float z = h_ * ga + be, sg = sigmoidf_(z);
                z_ *= sg * (1.f + z * (1.f - sg));
            }
            if (!in) z_ = 0.f;
            xh[i][j] = h_;
            dz[i][j] = z_;
            s1 += z_;
            s2 += z_ * h_;
        }
        for (int off = 1; off < L; off <<= 1) {                   // butterfly inside the L-lane channel segment
            s1 += __shfl_xor(s1, off, 64);
            s2 += __shfl_xor(s2, off, 64);
        }
        if (in && (lane & (L - 1)) == 0) {
            dbeta_ws[(int64_t)b * C + c] = s1;
            dgamma_ws[(int64_t)b * C + c] = s2;
            m1 += ga * s1;                                        // one lane per channel carries the channel's weighted sums
            m2 += ga * s2;
        }
    }
    const float inv_n = 1.f / (float)(cpg * HW);
    m1 = wave_sum(m1) * inv_n;
    m2 = wave_sum(m2) * inv_n;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = lane + i * 64;
        const bool in = idx < n4;
        float s3 = 0.f;
        if (in) {
            f32x4 ev = {0.f, 0.f, 0.f, 0.f}, o;
            if (e4) ev = e4[idx];
            if (f4) {
                const f32x4 fv = f4[idx];
#pragma unroll
                for (int j = 0; j < 4; ++j) ev[j] += fv[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = rstd * (dz[i][j] * gav[i] - m1 - xh[i][j] * m2) + ev[j];
                s3 += o[j];
            }
            o4[idx] = o;
        }
        if (rs_out) {
            for (int off = 1; off < L; off <<= 1) s3 += __shfl_xor(s3, off, 64);
            if (in && (lane & (L - 1)) == 0) rs_out[(int64_t)b * rs_ld + g * cpg + idx / L] = s3;
        }
    }
}

// ---- multi-workgroup GroupNorm for slabs too large for registers (256x256 images at batch 8: B*G = 256 groups of 1 MB) ----
// One workgroup per group streams a 1 MB slab three times at a fraction of the HBM rate.  Here a group is cut into S
// chunks of <= 8192 floats, one workgroup each: pass A reduces the chunk (register-resident) to (mean, M2) partials,
// pass B combines the S partials in FIXED order (Chan et al.) and normalises the chunk.  2 reads + 1 write, S x the
// parallelism, deterministic.
__device__ __forceinline__ void gn_combine(const float* __restrict__ part, int S, float n_c, float& mean, float& m2) {
    float n = 0.f;
    mean = 0.f;
    m2 = 0.f;
    for (int k = 0; k < S; ++k) {
        const float mk = part[2 * k], qk = part[2 * k + 1];
        const float nn = n + n_c, delta = mk - mean;
        mean += delta * (n_c / nn);
        m2 += qk + delta * delta * (n * n_c / nn);
        n = nn;
    }
}

template <int NV>
__global__ __launch_bounds__(256) void gn_chunk_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int C, int HW,
                                                             int G, int S, int64_t x_bs) {
    __shared__ float red[8];
    const int bg = blockIdx.x / S, sc = blockIdx.x - bg * S;
    const int b = bg / G, g = bg - b * G;
    const int cpg = C / G;
    const int c4 = ((cpg * HW) >> 2) / S;                 // float4 per chunk
    const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x + (int64_t)b * x_bs + (int64_t)g * cpg * HW) + (int64_t)sc * c4;
    const int tid = threadIdx.x;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * 256;
        v[i] = (idx < c4) ? x4[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = block_sum_256(s, red) / (float)(c4 * 4);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (tid + i * 256 < c4) {
            const float a0 = v[i][0] - mean, a1 = v[i][1] - mean, a2 = v[i][2] - mean, a3 = v[i][3] - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    q = block_sum_256(q, red + 4);
    if (tid == 0) {
        part[2 * (int64_t)blockIdx.x] = mean;
        part[2 * (int64_t)blockIdx.x + 1] = q;
    }
}

__global__ __launch_bounds__(256) void gn_chunk_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ y,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                             const float* __restrict__ part, int C, int HW, int G, int S, float eps,
                                                             int apply_silu, int64_t x_bs, int64_t y_bs) {
    const int bg = blockIdx.x / S, sc = blockIdx.x - bg * S;
    const int b = bg / G, g = bg - b * G;
    const int cpg = C / G;
    const int c4 = ((cpg * HW) >> 2) / S;
    float mean, m2;
    gn_combine(part + 2 * (int64_t)bg * S, S, (float)(c4 * 4), mean, m2);
    const float rstd = rsqrtf(m2 / (float)(cpg * HW) + eps);
    if (sc == 0 && threadIdx.x == 0) {
        mean_out[bg] = mean;
        rstd_out[bg] = rstd;
    }
    const int64_t goff = (int64_t)g * cpg * HW;
    const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x + (int64_t)b * x_bs + goff) + (int64_t)sc * c4;
    f32x4* __restrict__ y4 = reinterpret_cast<f32x4*>(y + (int64_t)b * y_bs + goff) + (int64_t)sc * c4;
    const int L = HW >> 2;
    for (int idx = threadIdx.x; idx < c4; idx += 256) {
        const int c = g * cpg + (sc * c4 + idx) / L;
        const float ga = gamma[c] * rstd, be = beta[c] - mean * ga;
        const f32x4 v = x4[idx];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float z = v[j] * ga + be;
            o[j] = apply_silu ? z * sigmoidf_(z) : z;
        }
        y4[idx] = o;
    }
}

// backward: chunks never straddle a channel (S = cpg * chunks-per-channel).  Pass A: per-chunk (sum dz, sum dz*xhat);
// pass B: fixed-order per-channel sums -> group means -> dx of the chunk.
__global__ __launch_bounds__(256) void gn_chunk_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ part, int C, int HW, int G, int S, int apply_silu,
                                                                 int64_t dy_bs, int64_t x_bs) {
    __shared__ float red[8];
    const int bg = blockIdx.x / S, sc = blockIdx.x - bg * S;
    const int b = bg / G, g = bg - b * G;
    const int cpg = C / G;
    const int c4 = ((cpg * HW) >> 2) / S;
    const int64_t goff = (int64_t)g * cpg * HW;
    const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x + (int64_t)b * x_bs + goff) + (int64_t)sc * c4;
    const f32x4* __restrict__ d4 = reinterpret_cast<const f32x4*>(dy + (int64_t)b * dy_bs + goff) + (int64_t)sc * c4;
    const float mean = mean_in[bg], rstd = rstd_in[bg];
    const int c = g * cpg + (sc * c4) / (HW >> 2);
    const float ga = gamma[c], be = beta[c];
    float s1 = 0.f, s2 = 0.f;
    for (int idx = threadIdx.x; idx < c4; idx += 256) {
        const f32x4 xv = x4[idx], dv = d4[idx];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (xv[j] - mean) * rstd;
            float dz = dv[j];
            if (apply_silu) {
                const float z = xh * ga + be, sg = sigmoidf_(z);
                dz *= sg * (1.f + z * (1.f - sg));
            }
            s1 += dz;
            s2 += dz * xh;
        }
    }
    s1 = block_sum_256(s1, red);
    s2 = block_sum_256(s2, red + 4);
    if (threadIdx.x == 0) {
        part[2 * (int64_t)blockIdx.x] = s1;
        part[2 * (int64_t)blockIdx.x + 1] = s2;
    }
}

__global__ __launch_bounds__(256) void gn_chunk_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ extra, float* __restrict__ dx,
                                                                 float* __restrict__ dgamma_ws, float* __restrict__ dbeta_ws,
                                                                 const float* __restrict__ part, int C, int HW, int G, int S,
                                                                 int apply_silu, int64_t dy_bs, int64_t x_bs, int64_t ex_bs,
                                                                 int64_t dx_bs, const float* __restrict__ extra2, int64_t e2_bs,
                                                                 float* __restrict__ rs_part) {
    // extra2 / rs_part (round 4): the second residual gradient and the per-chunk sums of dx (the bias-gradient rows of the producing
    // convolution; gn_chunk_rowsum_kernel adds a channel's chunks in fixed order) ride in this pass instead of vd_add_strided + vd_rowsum
    // passes over dx (3 + 1 tensor sweeps at 256x256 images).
    __shared__ float ch_s1[64], ch_s2[64];
    __shared__ float red[4];
    const int bg = blockIdx.x / S, sc = blockIdx.x - bg * S;
    const int b = bg / G, g = bg - b * G;
    const int cpg = C / G;
    const int spc = S / cpg;                              // chunks per channel
    const int c4 = ((cpg * HW) >> 2) / S;
    const int tid = threadIdx.x;
    if (tid < cpg) {
        const float* __restrict__ pp = part + 2 * ((int64_t)bg * S + (int64_t)tid * spc);
        float a1 = 0.f, a2 = 0.f;
        for (int k = 0; k < spc; ++k) {
            a1 += pp[2 * k];
            a2 += pp[2 * k + 1];
        }
        ch_s1[tid] = a1;
        ch_s2[tid] = a2;
        if (sc == 0) {
            dbeta_ws[(int64_t)b * C + g * cpg + tid] = a1;
            dgamma_ws[(int64_t)b * C + g * cpg + tid] = a2;
        }
    }
    __syncthreads();
    float m1 = 0.f, m2 = 0.f;
    for (int cl = 0; cl < cpg; ++cl) {
        const float ga_ = gamma[g * cpg + cl];
        m1 += ga_ * ch_s1[cl];
        m2 += ga_ * ch_s2[cl];
    }
    const float inv_n = 1.f / (float)(cpg * HW);
    m1 *= inv_n;
    m2 *= inv_n;
    const int64_t goff = (int64_t)g * cpg * HW;
    const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x + (int64_t)b * x_bs + goff) + (int64_t)sc * c4;
    const f32x4* __restrict__ d4 = reinterpret_cast<const f32x4*>(dy + (int64_t)b * dy_bs + goff) + (int64_t)sc * c4;
    const f32x4* __restrict__ e4 = extra ? reinterpret_cast<const f32x4*>(extra + (int64_t)b * ex_bs + goff) + (int64_t)sc * c4 : nullptr;
    const f32x4* __restrict__ f4 = extra2 ? reinterpret_cast<const f32x4*>(extra2 + (int64_t)b * e2_bs + goff) + (int64_t)sc * c4 : nullptr;
    f32x4* __restrict__ o4 = reinterpret_cast<f32x4*>(dx + (int64_t)b * dx_bs + goff) + (int64_t)sc * c4;
    const float mean = mean_in[bg], rstd = rstd_in[bg];
    const int c = g * cpg + sc / spc;
    const float ga = gamma[c], be = beta[c];
    float rs = 0.f;
    for (int idx = tid; idx < c4; idx += 256) {
        const f32x4 xv = x4[idx], dv = d4[idx];
        f32x4 ev = {0.f, 0.f, 0.f, 0.f}, fv = {0.f, 0.f, 0.f, 0.f}, o;
        if (e4) ev = e4[idx];
        if (f4) fv = f4[idx];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (xv[j] - mean) * rstd;
            float dz = dv[j];
            if (apply_silu) {
                const float z = xh * ga + be, sg = sigmoidf_(z);
                dz *= sg * (1.f + z * (1.f - sg));
            }
            o[j] = (rstd * (dz * ga - m1 - xh * m2) + ev[j]) + fv[j];
        }
        rs += (o[0] + o[1]) + (o[2] + o[3]);
        o4[idx] = o;
    }
    if (rs_part) {                                        // block-uniform
        rs = block_sum_256(rs, red);
        if (tid == 0) rs_part[blockIdx.x] = rs;
    }
}

// rowsum[b][c] = the sum of channel c's chunk sums (fixed order): one thread per (b, c)
__global__ __launch_bounds__(256) void gn_chunk_rowsum_kernel(const float* __restrict__ rs_part, float* __restrict__ rowsum, int B, int C, int G,
                                                              int S, int64_t ld) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    const int cpg = C / G, spc = S / cpg;
    const int g = c / cpg, cl = c - g * cpg;
    const float* __restrict__ pp = rs_part + ((int64_t)(b * G + g) * S + (int64_t)cl * spc);
    float a = 0.f;
    for (int k = 0; k < spc; ++k) a += pp[k];
    rowsum[(int64_t)b * ld + c] = a;
}

// ---- ONE-launch form of the chunked kernels (round 4): a chunk stays in its workgroup's registers between the statistics and the apply step --------
// The two-launch form reads x (and dy) twice because the S workgroups of a group can only meet at a kernel boundary.  Here every workgroup publishes
// its partial pair as two 64-bit words (value, launch epoch) with agent-scope atomic stores and then polls the words of its group's S chunks (thread t
// polls chunk t) until all carry this launch's epoch: forward 3 -> 2 tensor sweeps, backward 5 -> 3.  Progress: the S <= 256 workgroups of a group are
// consecutive block ids, workgroups are dispatched in order, and a workgroup only ever waits for blocks of its own group, i.e. for blocks that are resident
// or next in line -- nothing waits for a later group.  The poll is BOUNDED all the same (a timeout publishes NaN statistics: the result is visibly wrong, never
// a hang), value and epoch travel in one atomic word (no fence, no ordering assumption), and stale words of earlier launches carry older epochs (the
// workspace is shared with the split-K slabs: an accidental match needs two specific 32-bit patterns).  HIP-graph captures bake the epoch into the launch, so
// they take the two-launch form (hipStreamIsCapturing).  VD_GN_CHUNK1_OFF=1: always the two-launch form.
constexpr int GN_POLL_MAX = 400000;

__device__ __forceinline__ void gn_publish(unsigned long long* __restrict__ wa, unsigned long long* __restrict__ wb, float a, float b, unsigned epoch) {
    __hip_atomic_store(wa, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(wb, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// thread t < S: the pair of chunk t of this group, into LDS pa[t], pb[t] (NaN after a timeout); all threads leave through a barrier
// A timeout is also REPORTED: `flag` is a word of pinned host memory (system-scope add), which the host reads without synchronising at the top of the
// next vd_groupnorm_* call and through vd_async_errors() -- a sticky error instead of NaN that nobody sees (round-4 review).
__device__ __forceinline__ void gn_gather(const unsigned long long* __restrict__ words, int S, unsigned epoch, float* __restrict__ pa, float* __restrict__ pb,
                                          unsigned* __restrict__ flag, int poll_max) {
    const int t = threadIdx.x;
    if (t < S) {
        const unsigned long long* __restrict__ wa = words + 2 * t;
        unsigned long long a = 0, b = 0;
        bool ok = false;
        for (int it = 0; it < poll_max; ++it) {
            a = __hip_atomic_load(wa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b = __hip_atomic_load(wa + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = (unsigned)(a >> 32) == epoch && (unsigned)(b >> 32) == epoch;
            if (ok) break;
            __builtin_amdgcn_s_sleep(4);
        }
        pa[t] = ok ? __uint_as_float((unsigned)a) : __builtin_nanf("");
        pb[t] = ok ? __uint_as_float((unsigned)b) : __builtin_nanf("");
        if (!ok && flag) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
}

template <int NV>
__global__ __launch_bounds__(256) void gn_chunk1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            unsigned long long* __restrict__ words, int C, int HW, int G, int S, float eps, int apply_silu,
                                                            int64_t x_bs, int64_t y_bs, unsigned epoch, unsigned* __restrict__ flag, int poll_max) {
    __shared__ float red[8];
    __shared__ float pa[256], pb[256];
    const int bg = blockIdx.x / S, sc = blockIdx.x - bg * S;
    const int b = bg / G, g = bg - b * G;
    const int cpg = C / G;
    const int c4 = ((cpg * HW) >> 2) / S;                 // float4 per chunk (<= 256 NV)
    const int64_t goff = (int64_t)g * cpg * HW;
    const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x + (int64_t)b * x_bs + goff) + (int64_t)sc * c4;
    f32x4* __restrict__ y4 = reinterpret_cast<f32x4*>(y + (int64_t)b * y_bs + goff) + (int64_t)sc * c4;
    const int tid = threadIdx.x;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * 256;
        v[i] = (idx < c4) ? x4[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float cmean = block_sum_256(s, red) / (float)(c4 * 4);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (tid + i * 256 < c4) {
            const float a0 = v[i][0] - cmean, a1 = v[i][1] - cmean, a2 = v[i][2] - cmean, a3 = v[i][3] - cmean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    q = block_sum_256(q, red + 4);
    if (tid == 0) gn_publish(words + 2 * (int64_t)blockIdx.x, words + 2 * (int64_t)blockIdx.x + 1, cmean, q, epoch);
    gn_gather(words + 2 * (int64_t)bg * S, S, epoch, pa, pb, flag, poll_max);
    // the S partials in FIXED order (Chan et al.), as gn_combine
    float n = 0.f, mean = 0.f, m2 = 0.f;
    const float n_c = (float)(c4 * 4);
    for (int k = 0; k < S; ++k) {
        const float mk = pa[k], qk = pb[k];
        const float nn = n + n_c, delta = mk - mean;
        mean += delta * (n_c / nn);
        m2 += qk + delta * delta * (n * n_c / nn);
        n = nn;
    }
    const float rstd = rsqrtf(m2 / (float)(cpg * HW) + eps);
    if (sc == 0 && tid == 0) {
        mean_out[bg] = mean;
        rstd_out[bg] = rstd;
    }
    const int L = HW >> 2;
    const int c = g * cpg + (sc * c4) / L;                // chunks never straddle a channel
    const float ga = gamma[c] * rstd, be = beta[c] - mean * ga;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * 256;
        if (idx < c4) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = v[i][j] * ga + be;
                o[j] = apply_silu ? z * sigmoidf_(z) : z;
            }
            y4[idx] = o;
        }
    }
}

template <int NV>
__global__ __launch_bounds__(256) void gn_chunk1_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ extra, float* __restrict__ dx,
                                                            float* __restrict__ dgamma_ws, float* __restrict__ dbeta_ws,
                                                            unsigned long long* __restrict__ words, int C, int HW, int G, int S, int apply_silu,
                                                            int64_t dy_bs, int64_t x_bs, int64_t ex_bs, int64_t dx_bs, const float* __restrict__ extra2,
                                                            int64_t e2_bs, float* __restrict__ rs_part, unsigned epoch, unsigned* __restrict__ flag, int poll_max) {
    __shared__ float red[8];
    __shared__ float pa[256], pb[256];
    __shared__ float ch_s1[64], ch_s2[64];
    const int bg = blockIdx.x / S, sc = blockIdx.x - bg * S;
    const int b = bg / G, g = bg - b * G;
    const int cpg = C / G;
    const int spc = S / cpg;                              // chunks per channel
    const int c4 = ((cpg * HW) >> 2) / S;
    const int tid = threadIdx.x;
    const int64_t goff = (int64_t)g * cpg * HW;
    const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x + (int64_t)b * x_bs + goff) + (int64_t)sc * c4;
    const f32x4* __restrict__ d4 = reinterpret_cast<const f32x4*>(dy + (int64_t)b * dy_bs + goff) + (int64_t)sc * c4;
    const f32x4* __restrict__ e4 = extra ? reinterpret_cast<const f32x4*>(extra + (int64_t)b * ex_bs + goff) + (int64_t)sc * c4 : nullptr;
    const f32x4* __restrict__ f4 = extra2 ? reinterpret_cast<const f32x4*>(extra2 + (int64_t)b * e2_bs + goff) + (int64_t)sc * c4 : nullptr;
    f32x4* __restrict__ o4 = reinterpret_cast<f32x4*>(dx + (int64_t)b * dx_bs + goff) + (int64_t)sc * c4;
    const float mean = mean_in[bg], rstd = rstd_in[bg];
    const int c = g * cpg + sc / spc;
    const float ga = gamma[c], be = beta[c];
    f32x4 xh[NV], dz[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * 256;
        const bool in = idx < c4;
        const f32x4 xv = in ? x4[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 dv = in ? d4[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float h_ = (xv[j] - mean) * rstd;
            float z_ = dv[j];
            if (apply_silu) {
                const float z = h_ * ga + be, sg = sigmoidf_(z);
                z_ *= sg * (1.f + z * (1.f - sg));
            }
            if (!in) z_ = 0.f;
            xh[i][j] = h_;
            dz[i][j] = z_;
            s1 += z_;
            s2 += z_ * h_;
        }
    }
    s1 = block_sum_256(s1, red);
    s2 = block_sum_256(s2, red + 4);
    if (tid == 0) gn_publish(words + 2 * (int64_t)blockIdx.x, words + 2 * (int64_t)blockIdx.x + 1, s1, s2, epoch);
    gn_gather(words + 2 * (int64_t)bg * S, S, epoch, pa, pb, flag, poll_max);
    if (tid < cpg) {                                      // channel sums: the channel's chunks in fixed order
        float a1 = 0.f, a2 = 0.f;
        for (int k = 0; k < spc; ++k) {
            a1 += pa[tid * spc + k];
            a2 += pb[tid * spc + k];
        }
        ch_s1[tid] = a1;
        ch_s2[tid] = a2;
        if (sc == 0) {
            dbeta_ws[(int64_t)b * C + g * cpg + tid] = a1;
            dgamma_ws[(int64_t)b * C + g * cpg + tid] = a2;
        }
    }
    __syncthreads();
    float m1 = 0.f, m2 = 0.f;
    for (int cl = 0; cl < cpg; ++cl) {
        const float ga_ = gamma[g * cpg + cl];
        m1 += ga_ * ch_s1[cl];
        m2 += ga_ * ch_s2[cl];
    }
    const float inv_n = 1.f / (float)(cpg * HW);
    m1 *= inv_n;
    m2 *= inv_n;
    float rs = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * 256;
        if (idx < c4) {
            f32x4 ev = {0.f, 0.f, 0.f, 0.f}, fv = {0.f, 0.f, 0.f, 0.f}, o;
            if (e4) ev = e4[idx];
            if (f4) fv = f4[idx];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (rstd * (dz[i][j] * ga - m1 - xh[i][j] * m2) + ev[j]) + fv[j];
            rs += (o[0] + o[1]) + (o[2] + o[3]);
            o4[idx] = o;
        }
    }
    if (rs_part) {                                        // block-uniform
        rs = block_sum_256(rs, red);
        if (tid == 0) rs_part[blockIdx.x] = rs;
    }
}

constexpr int GN_REG_MAX = 28 * 1024;          // largest group the register-resident kernels take (7 float4 x 1024 threads)

// Chunks per group for the multi-workgroup path (0: not applicable): chunk <= 8192 floats, inside one channel, S <= 256.
static unsigned gn_next_epoch() {                  // never 0: zero-filled workspace words must not match
    static std::atomic<unsigned> e{0};
    unsigned v;
    do v = ++e; while (v == 0);
    return v;
}

// Sticky asynchronous error word of the polling kernels: pinned, device-mapped host memory (allocated at the first one-launch call; never freed).
static unsigned* g_gn_flag_host = nullptr;
static unsigned* g_gn_flag_dev = nullptr;
static unsigned* gn_flag_dev() {
    static std::atomic<int> state{0};              // 0: not tried, 1: ready, -1: no pinned memory (kernels then only emit NaN, as before)
    if (state.load() == 0) {
        void* h = nullptr;
        void* dptr = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && hipHostGetDevicePointer(&dptr, h, 0) == hipSuccess) {
            *reinterpret_cast<volatile unsigned*>(h) = 0u;
            g_gn_flag_host = reinterpret_cast<unsigned*>(h);
            g_gn_flag_dev = reinterpret_cast<unsigned*>(dptr);
            state.store(1);
        } else {
            (void)hipGetLastError();
            state.store(-1);
        }
    }
    return g_gn_flag_dev;
}
static int gn_poll_max() {                         // VD_GN_POLL_MAX=0: every poll times out at once (the sticky-error test)
    static const int v = getenv("VD_GN_POLL_MAX") ? atoi(getenv("VD_GN_POLL_MAX")) : GN_POLL_MAX;
    return v;
}
static unsigned gn_flag_read() { return g_gn_flag_host ? *reinterpret_cast<volatile unsigned*>(g_gn_flag_host) : 0u; }

// Resident-workgroup capacity of the device for the one-launch kernels (occupancy x CU count, the smallest over the four instantiations).  A group's S
// workgroups poll each other, so all S must be resident at once with room to spare for whatever else runs (the weight-gradient side stream, a
// partitioned or CU-masked device): below 2 x S the two-launch form is taken.
static int gn_chunk1_capacity();

// the one-launch chunked kernels: not inside a HIP-graph capture (the epoch would be baked into the launch), not with VD_GN_CHUNK1_OFF=1
static bool gn_chunk1_ok(hipStream_t st, int S) {
    static const int off = getenv("VD_GN_CHUNK1_OFF") ? atoi(getenv("VD_GN_CHUNK1_OFF")) : 0;
    if (off) return false;
    if (gn_chunk1_capacity() < 2 * S) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return cs == hipStreamCaptureStatusNone;
}

static bool gn_wave_ok() {
    static const int off = getenv("VD_GN_WAVE_OFF") ? atoi(getenv("VD_GN_WAVE_OFF")) : 0;
    return !off;
}

// fwd: chunks of up to 16384 floats (16 float4 per thread: fewer chunks to wait for, 5.1 TB/s against 3.4-4.0 at 256x256); the backward keeps two
// arrays per element in registers and stays at 8192 (profiles/r04_gn_chunk_probe.txt)
static int gn_chunks(int B, int C, int HW, int G, bool fwd = false) {
    const int cpg = C / G;
    const int64_t slab = (int64_t)cpg * HW;
    if (slab <= GN_REG_MAX || HW % 4 != 0 || cpg > 64) return 0;
    static const int cmax_f = getenv("VD_GN_CHUNK_MAX_FWD") ? atoi(getenv("VD_GN_CHUNK_MAX_FWD")) : 16384;
    static const int cmax_b = getenv("VD_GN_CHUNK_MAX") ? atoi(getenv("VD_GN_CHUNK_MAX")) : 8192;
    const int cmax = fwd ? cmax_f : cmax_b;
    const int lim = cmax < 1024 ? 1024 : (cmax > (fwd ? 16384 : 8192) ? (fwd ? 16384 : 8192) : cmax);
    int per_ch = 1;
    while (HW / per_ch > lim && (HW / per_ch) % 2 == 0) per_ch *= 2;
    if (HW / per_ch > lim || (HW / per_ch) % 4 != 0 || HW % per_ch != 0) return 0;
    const int S = cpg * per_ch;
    if (S > 256 || (int64_t)B * G * S > (1 << 22)) return 0;
    return S;
}

static int gn_chunk1_capacity() {
    static std::atomic<int> cap{-1};
    int v = cap.load();
    if (v >= 0) return v;
    int dev = 0, cus = 0, best = 1 << 30;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        cap.store(0);
        return 0;
    }
    const void* ks[4] = {reinterpret_cast<const void*>(&gn_chunk1_fwd_kernel<8>), reinterpret_cast<const void*>(&gn_chunk1_fwd_kernel<16>),
                         reinterpret_cast<const void*>(&gn_chunk1_bwd_kernel<8>), reinterpret_cast<const void*>(&gn_chunk1_bwd_kernel<16>)};
    for (int i = 0; i < 4; ++i) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ks[i], 256, 0) != hipSuccess) {
            (void)hipGetLastError();
            nb = 0;
        }
        if (nb * cus < best) best = nb * cus;
    }
    if (const char* e = getenv("VD_GN_CHUNK1_CAPACITY")) best = atoi(e);       // tests: pretend a smaller device
    cap.store(best);
    return best;
}

// a timeout reported by an earlier polling launch fails every later GroupNorm call until vd_async_errors(1) clears it
int gn_sticky_impl(const char* who) {
    const unsigned n = gn_flag_read();
    if (n == 0) return 0;
    vd_set_error("%s: %u GroupNorm poll timeout(s) reported by an earlier one-launch chunked kernel (statistics were NaN); "
                 "vd_async_errors(1) clears the flag, VD_GN_CHUNK1_OFF=1 selects the two-launch form", who, n);
    return VD_ETIMEDOUT;
}

}  // namespace

// (vd_common.h) every vd_groupnorm_* entry point, the pre-split ones of vd_presplit.hip included, fails while the flag is set
int vd_gn_sticky(const char* who) { return gn_sticky_impl(who); }

extern "C" int vd_async_errors(int clear) {
    const unsigned n = gn_flag_read();
    if (clear && g_gn_flag_host) *reinterpret_cast<volatile unsigned*>(g_gn_flag_host) = 0u;
    return (int)(n > 0x7fffffffu ? 0x7fffffffu : n);
}

extern "C" int64_t vd_groupnorm_ws_floats(int B, int C, int HW, int G) {
    if (B <= 0 || C <= 0 || HW <= 0 || G <= 0 || C % G) return 0;
    const int S = gn_chunks(B, C, HW, G);
    // per chunk: two statistics + the sum of its dx (vd_groupnorm_bwd_fused rowsum) + the two 64-bit (value, epoch) words of the one-launch form
    return S ? 4 * (int64_t)B * G * S + 4 * (int64_t)B * G * S : 0;
}

extern "C" int vd_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                int B, int C, int HW, int G, float eps, int apply_silu, int64_t x_bstride, int64_t y_bstride,
                                float* ws, void* stream) {
    VD_REQUIRE(x && gamma && beta && y && mean && rstd, "vd_groupnorm_fwd: null pointer");
    VD_REQUIRE(B > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0, "vd_groupnorm_fwd: bad dims B=%d C=%d HW=%d G=%d", B, C, HW, G);
    const int64_t slab = (int64_t)(C / G) * HW;
    const bool al = (HW % 4 == 0) && (x_bstride % 4 == 0) && (y_bstride % 4 == 0) && ((((uintptr_t)x) & 15) == 0) &&
                    ((((uintptr_t)y) & 15) == 0);
    const bool reg_ok = al && slab <= GN_REG_MAX;
    const int S = (al && ws) ? gn_chunks(B, C, HW, G, true) : 0;
    if (const int rc = vd_gn_sticky("vd_groupnorm_fwd")) return rc;
    if (S && gn_chunk1_ok((hipStream_t)stream, S)) {       // one launch: the chunk stays in registers between statistics and apply
        unsigned long long* words = reinterpret_cast<unsigned long long*>(ws + 4 * (int64_t)B * G * S);
        unsigned* flag = gn_flag_dev();
        if (slab / S > 8192)
            hipLaunchKernelGGL((gn_chunk1_fwd_kernel<16>), dim3(B * G * S), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd, words, C, HW,
                               G, S, eps, apply_silu, x_bstride, y_bstride, gn_next_epoch(), flag, gn_poll_max());
        else
            hipLaunchKernelGGL((gn_chunk1_fwd_kernel<8>), dim3(B * G * S), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd, words, C, HW,
                               G, S, eps, apply_silu, x_bstride, y_bstride, gn_next_epoch(), flag, gn_poll_max());
        VD_LAUNCH_CHECK("vd_groupnorm_fwd");
        return 0;
    }
    if (S) {       // large slabs: S workgroups per group (partials in ws, vd_groupnorm_ws_floats())
        if (slab / S > 8192) hipLaunchKernelGGL((gn_chunk_stats_kernel<16>), dim3(B * G * S), dim3(256), 0, (hipStream_t)stream, x, ws, C, HW, G, S, x_bstride);
        else hipLaunchKernelGGL((gn_chunk_stats_kernel<8>), dim3(B * G * S), dim3(256), 0, (hipStream_t)stream, x, ws, C, HW, G, S, x_bstride);
        hipLaunchKernelGGL(gn_chunk_apply_kernel, dim3(B * G * S), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd,
                           ws, C, HW, G, S, eps, apply_silu, x_bstride, y_bstride);
        VD_LAUNCH_CHECK("vd_groupnorm_fwd");
        return 0;
    }
#define VD_GN_FWD(NVV)                                                                                                       \
    hipLaunchKernelGGL((gn_fwd_reg_kernel<NVV>), dim3(B * G), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd, \
                       C, HW, G, eps, apply_silu, x_bstride, y_bstride, (float*)nullptr)
    if (reg_ok && slab <= 1024 && gn_wave_ok()) {
#define VD_GN_FWDW(NVV)                                                                                                                     \
    hipLaunchKernelGGL((gn_fwd_wave_kernel<NVV>), dim3(vd_cdiv(B * G, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd, \
                       C, HW, G, eps, apply_silu, x_bstride, y_bstride, (float*)nullptr, B * G)
        if (slab <= 256) VD_GN_FWDW(1);
        else if (slab <= 512) VD_GN_FWDW(2);
        else VD_GN_FWDW(4);
#undef VD_GN_FWDW
    } else if (reg_ok && slab <= 1024) VD_GN_FWD(1);
    else if (reg_ok && slab <= 2048) VD_GN_FWD(2);
    else if (reg_ok && slab <= 4096) VD_GN_FWD(4);
    else if (reg_ok && slab <= 8192) VD_GN_FWD(8);
    else if (reg_ok && slab <= 12 * 1024) VD_GN_FWD(12);
    else if (reg_ok && slab <= 14 * 1024)
        hipLaunchKernelGGL((gn_fwd_reg_kernel<7, 512>), dim3(B * G), dim3(512), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd, C, HW, G, eps,
                           apply_silu, x_bstride, y_bstride, (float*)nullptr);
    else if (reg_ok)
        hipLaunchKernelGGL((gn_fwd_reg_kernel<7, 1024>), dim3(B * G), dim3(1024), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd, C, HW, G, eps,
                           apply_silu, x_bstride, y_bstride, (float*)nullptr);
    else
        hipLaunchKernelGGL(gn_fwd_kernel, dim3(B * G), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd, C, HW, G,
                           eps, apply_silu, x_bstride, y_bstride);
#undef VD_GN_FWD
    VD_LAUNCH_CHECK("vd_groupnorm_fwd");
    return 0;
}

extern "C" int vd_groupnorm_stats(const float* x, const float* gamma, const float* beta, float* ss, float* mean, float* rstd, int B,
                                  int C, int HW, int G, float eps, int64_t x_bstride, void* stream) {
    VD_REQUIRE(x && gamma && beta && ss && mean && rstd, "vd_groupnorm_stats: null pointer");
    VD_REQUIRE(B > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0 && C / G <= 256, "vd_groupnorm_stats: bad dims");
    const int64_t slab = (int64_t)(C / G) * HW;
    VD_REQUIRE(HW % 4 == 0 && x_bstride % 4 == 0 && ((((uintptr_t)x) & 15) == 0) && slab <= GN_REG_MAX,
               "vd_groupnorm_stats: needs 16-B aligned groups of at most 28672 elements (got %lld)", (long long)slab);
#define VD_GN_ST(NVV)                                                                                                          \
    hipLaunchKernelGGL((gn_fwd_reg_kernel<NVV>), dim3(B * G), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (float*)nullptr, \
                       mean, rstd, C, HW, G, eps, 0, x_bstride, (int64_t)0, ss)
    if (slab <= 1024 && gn_wave_ok()) {                           // (the same kernel as vd_groupnorm_fwd takes for this slab: identical statistics)
#define VD_GN_STW(NVV)                                                                                                                       \
    hipLaunchKernelGGL((gn_fwd_wave_kernel<NVV>), dim3(vd_cdiv(B * G, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (float*)nullptr, \
                       mean, rstd, C, HW, G, eps, 0, x_bstride, (int64_t)0, ss, B * G)
        if (slab <= 256) VD_GN_STW(1);
        else if (slab <= 512) VD_GN_STW(2);
        else VD_GN_STW(4);
#undef VD_GN_STW
    } else if (slab <= 1024) VD_GN_ST(1);
    else if (slab <= 2048) VD_GN_ST(2);
    else if (slab <= 4096) VD_GN_ST(4);
    else if (slab <= 8192) VD_GN_ST(8);
    else if (slab <= 12 * 1024) VD_GN_ST(12);
    else if (slab <= 14 * 1024)
        hipLaunchKernelGGL((gn_fwd_reg_kernel<7, 512>), dim3(B * G), dim3(512), 0, (hipStream_t)stream, x, gamma, beta, (float*)nullptr, mean, rstd, C,
                           HW, G, eps, 0, x_bstride, (int64_t)0, ss);
    else
        hipLaunchKernelGGL((gn_fwd_reg_kernel<7, 1024>), dim3(B * G), dim3(1024), 0, (hipStream_t)stream, x, gamma, beta, (float*)nullptr, mean, rstd, C,
                           HW, G, eps, 0, x_bstride, (int64_t)0, ss);
#undef VD_GN_ST
    VD_LAUNCH_CHECK("vd_groupnorm_stats");
    return 0;
}

// GroupNorm statistics from per-tile channel sums (vd_gemm_desc.gn_part): one 64-thread workgroup per (batch item, group).
__global__ __launch_bounds__(64) void gn_stats_from_part_kernel(const float* __restrict__ part, int tiles, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ ss,
                                                                 float* __restrict__ mean_out, float* __restrict__ rstd_out, int C, int HW, int G,
                                                                 float eps) {
    __shared__ float mr[2];
    const int b = blockIdx.x / G, g = blockIdx.x - b * G, cpg = C / G;
    if (threadIdx.x == 0) {
        double s1 = 0.0, s2 = 0.0;
        for (int t = 0; t < tiles; ++t) {
            const float* __restrict__ p = part + (((int64_t)b * tiles + t) * C + g * cpg) * 2;
            for (int c = 0; c < cpg; ++c) {
                s1 += (double)p[2 * c];
                s2 += (double)p[2 * c + 1];
            }
        }
        const double n = (double)cpg * HW, m = s1 / n;
        double var = s2 / n - m * m;
        if (var < 0.0) var = 0.0;
        mr[0] = (float)m;
        mr[1] = (float)(1.0 / sqrt(var + (double)eps));
        mean_out[blockIdx.x] = mr[0];
        rstd_out[blockIdx.x] = mr[1];
    }
    __syncthreads();
    const float mean = mr[0], rstd = mr[1];
    for (int c = threadIdx.x; c < cpg; c += 64) {
        const int ch = g * cpg + c;
        const float sc = gamma[ch] * rstd;
        ss[((int64_t)b * C + ch) * 2] = sc;
        ss[((int64_t)b * C + ch) * 2 + 1] = beta[ch] - mean * sc;
    }
}

extern "C" int vd_groupnorm_stats_from_partials(const float* part, int tiles, const float* gamma, const float* beta, float* ss, float* mean,
                                                float* rstd, int B, int C, int HW, int G, float eps, void* stream) {
    VD_REQUIRE(part && gamma && beta && ss && mean && rstd, "vd_groupnorm_stats_from_partials: null pointer");
    VD_REQUIRE(B > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0 && tiles > 0, "vd_groupnorm_stats_from_partials: bad dims");
    hipLaunchKernelGGL(gn_stats_from_part_kernel, dim3(B * G), dim3(64), 0, (hipStream_t)stream, part, tiles, gamma, beta, ss, mean, rstd, C, HW,
                       G, eps);
    VD_LAUNCH_CHECK("vd_groupnorm_stats_from_partials");
    return 0;
}

extern "C" int vd_rowsum(const float* x, float* ws, int B, int M, int HW, int64_t x_bstride, int64_t ws_ld, void* stream);
extern "C" int vd_add_strided(float* dst, const float* src, int B, int64_t inner, int64_t dst_bstride, int64_t src_bstride, int accumulate,
                              void* stream);

// dx = GroupNorm(+SiLU) backward + extra + extra2 (two residual gradients: the block's own and a skip connection's), and optionally
// rowsum[b*rowsum_ld + c] = sum_p dx[b][c][p] -- the bias-gradient rows of whichever convolution produced x (its dY IS this dx).  On the
// register-resident kernels both ride in the pass that writes dx; the other paths fall back to vd_add_strided / vd_rowsum launches.
extern "C" int vd_groupnorm_bwd_fused(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                      const float* beta, const float* extra, const float* extra2, float* dx, float* dgamma_ws,
                                      float* dbeta_ws, float* rowsum, int B, int C, int HW, int G, int apply_silu, int64_t dy_bstride,
                                      int64_t x_bstride, int64_t extra_bstride, int64_t extra2_bstride, int64_t dx_bstride,
                                      int64_t rowsum_ld, float* ws, void* stream) {
    VD_REQUIRE(dy && x && mean && rstd && gamma && beta && dx && dgamma_ws && dbeta_ws, "vd_groupnorm_bwd: null pointer");
    VD_REQUIRE(B > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0 && C / G <= 64, "vd_groupnorm_bwd: bad dims");
    VD_REQUIRE(!rowsum || rowsum_ld >= C, "vd_groupnorm_bwd: rowsum_ld < C");
    if (const int rc = vd_gn_sticky("vd_groupnorm_bwd")) return rc;
    const int64_t slab = (int64_t)(C / G) * HW;
    const int L = HW / 4;
    const bool al = ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)extra) | ((uintptr_t)extra2)) & 15) == 0 &&
                    ((x_bstride | dy_bstride | dx_bstride | extra_bstride | extra2_bstride) & 3) == 0;
    const int S = (al && ws) ? gn_chunks(B, C, HW, G) : 0;
    const bool reg_ok = !S && al && HW % 4 == 0 && slab <= GN_REG_MAX &&
                        ((L >= 64 && L % 64 == 0) || (L < 64 && (L & (L - 1)) == 0 && L > 0));
    if (reg_ok) {
#define VD_GN_BWD(NVV, NT)                                                                                                          \
    hipLaunchKernelGGL((gn_bwd_reg_kernel<NVV, NT>), dim3(B * G), dim3(NT), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma, beta, \
                       extra, dx, dgamma_ws, dbeta_ws, C, HW, G, apply_silu, dy_bstride, x_bstride, extra_bstride, dx_bstride,      \
                       extra2, extra2_bstride, rowsum, rowsum_ld)
        if (slab <= 1024 && L < 64 && gn_wave_ok()) {
#define VD_GN_BWDW(NVV)                                                                                                                          \
    hipLaunchKernelGGL((gn_bwd_wave_kernel<NVV>), dim3(vd_cdiv(B * G, 4)), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma, beta, extra, \
                       dx, dgamma_ws, dbeta_ws, C, HW, G, apply_silu, dy_bstride, x_bstride, extra_bstride, dx_bstride, extra2, extra2_bstride,   \
                       rowsum, rowsum_ld, B * G)
            if (slab <= 256) VD_GN_BWDW(1);
            else if (slab <= 512) VD_GN_BWDW(2);
            else VD_GN_BWDW(4);
#undef VD_GN_BWDW
        } else if (slab <= 1024) VD_GN_BWD(1, 256);
        else if (slab <= 2048) VD_GN_BWD(2, 256);
        else if (slab <= 4096) VD_GN_BWD(4, 256);
        else if (slab <= 8192) VD_GN_BWD(4, 512);
        else if (slab <= 12 * 1024) VD_GN_BWD(6, 512);
        else if (slab <= 14 * 1024) VD_GN_BWD(7, 512);
        else VD_GN_BWD(7, 1024);
#undef VD_GN_BWD
        VD_LAUNCH_CHECK("vd_groupnorm_bwd");
        return 0;
    }
    if (S && gn_chunk1_ok((hipStream_t)stream, S)) {
        unsigned* flag = gn_flag_dev();
        float* rs_part = rowsum ? ws + 2 * (int64_t)B * G * S : nullptr;
        unsigned long long* words = reinterpret_cast<unsigned long long*>(ws + 4 * (int64_t)B * G * S);
        if (slab / S > 8192)
            hipLaunchKernelGGL((gn_chunk1_bwd_kernel<16>), dim3(B * G * S), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma, beta, extra, dx,
                               dgamma_ws, dbeta_ws, words, C, HW, G, S, apply_silu, dy_bstride, x_bstride, extra_bstride, dx_bstride, extra2,
                               extra2_bstride, rs_part, gn_next_epoch(), flag, gn_poll_max());
        else
            hipLaunchKernelGGL((gn_chunk1_bwd_kernel<8>), dim3(B * G * S), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma, beta, extra, dx,
                               dgamma_ws, dbeta_ws, words, C, HW, G, S, apply_silu, dy_bstride, x_bstride, extra_bstride, dx_bstride, extra2,
                               extra2_bstride, rs_part, gn_next_epoch(), flag, gn_poll_max());
        if (rowsum)
            hipLaunchKernelGGL(gn_chunk_rowsum_kernel, dim3(vd_cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)stream, rs_part, rowsum, B, C, G, S,
                               rowsum_ld);
        VD_LAUNCH_CHECK("vd_groupnorm_bwd");
        return 0;
    }
    if (S) {
        hipLaunchKernelGGL(gn_chunk_bwd_stats_kernel, dim3(B * G * S), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma,
                           beta, ws, C, HW, G, S, apply_silu, dy_bstride, x_bstride);
        float* rs_part = rowsum ? ws + 2 * (int64_t)B * G * S : nullptr;
        hipLaunchKernelGGL(gn_chunk_bwd_apply_kernel, dim3(B * G * S), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma,
                           beta, extra, dx, dgamma_ws, dbeta_ws, ws, C, HW, G, S, apply_silu, dy_bstride, x_bstride, extra_bstride,
                           dx_bstride, extra2, extra2_bstride, rs_part);
        if (rowsum)
            hipLaunchKernelGGL(gn_chunk_rowsum_kernel, dim3(vd_cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)stream, rs_part, rowsum, B, C, G, S,
                               rowsum_ld);
        VD_LAUNCH_CHECK("vd_groupnorm_bwd");
        return 0;
    } else {
        hipLaunchKernelGGL(gn_bwd_kernel, dim3(B * G), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma, beta, extra, dx,
                           dgamma_ws, dbeta_ws, C, HW, G, apply_silu, dy_bstride, x_bstride, extra_bstride, dx_bstride);
    }
    VD_LAUNCH_CHECK("vd_groupnorm_bwd");
    if (extra2) {
        const int rc = vd_add_strided(dx, extra2, B, (int64_t)C * HW, dx_bstride, extra2_bstride, 1, stream);
        if (rc) return rc;
    }
    if (rowsum) return vd_rowsum(dx, rowsum, B, C, HW, dx_bstride, rowsum_ld, stream);
    return 0;
}

extern "C" int vd_groupnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                const float* beta, const float* extra, float* dx, float* dgamma_ws, float* dbeta_ws, int B,
                                int C, int HW, int G, int apply_silu, int64_t dy_bstride, int64_t x_bstride,
                                int64_t extra_bstride, int64_t dx_bstride, float* ws, void* stream) {
    return vd_groupnorm_bwd_fused(dy, x, mean, rstd, gamma, beta, extra, nullptr, dx, dgamma_ws, dbeta_ws, nullptr, B, C, HW, G, apply_silu,
                                  dy_bstride, x_bstride, extra_bstride, 0, dx_bstride, 0, ws, stream);
}
