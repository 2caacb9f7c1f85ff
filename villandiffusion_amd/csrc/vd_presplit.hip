// ---- PRE-SPLIT activations (round 5): producers write what the matrix kernels consume ----------------------------------------------------
// The split-precision kernels contract f32 operands as bf16 (hi, lo) pairs: hi = bf16(x), lo = bf16(x - hi).  Until round 4 every consumer made
// that split itself, where it wrote the operand to LDS: the 3x3 convolution once per staged patch element, the 1x1 kernel once per m-tile, the
// weight gradient once per K-step for BOTH operands (14 converted elements per MFMA against the convolution's 3) -- v_cvt_pk_bf16_f32 +
// subtract + v_cvt again, staging VGPRs and a ds_write_b128 pass beside a matrix pipe that the chip already clocks down for power
// (profiles/r04_wgrad_stamps.txt: 45 % of a K-step in convert + store).  Here the PRODUCER of an activation (GroupNorm + SiLU forward, GroupNorm
// backward) writes the pair once, in ONE layout that every consumer can fetch without touching it:
//
//     O8 image of a [C][H*W] tensor:  unit (o, p, part) = 16 bytes = the 8 channels 8 o .. 8 o + 7 of pixel p as bf16, part 0 = hi, 1 = lo,
//     at byte ((o * HW + p) * 2 + part) * 16 of the image  -- the same 4 bytes per element, the same batch stride and the same channel-octet
//     offsets as the f32 NCHW tensor it replaces (a channel slice [8 a, 8 b) of an O8 image is an O8 image).
//
//   * the 3x3 convolution (k = channels: a lane's MFMA fragment is 8 consecutive channels of one pixel) copies units: its patch loader issues two
//     16-byte loads per (octet, pixel) item instead of eight 4-byte loads + 24 conversion instructions (conv3_k32p_kernel<..., PS = true>);
//   * the weight gradient (k = pixels: a fragment is 8 consecutive pixels of one channel) fetches BOTH operands by LDS-DMA (buffer_load ... lds,
//     16 bytes per lane, no VGPRs) into a [octet][part][pixel] image and reads it with ds_read_b64_tr_b16: a 16-lane group reads a block of
//     4 pixels x 16 channels and every lane receives 4 consecutive PIXELS of its channel -- the transpose is free, and because each lane of the
//     group supplies its own row address, the horizontal tap shift (pixels x - 1 .. x + 6) is just another row address: no shifted copies, no
//     ds_bpermute, no v_alignbit (wgrad_ps_group_kernel below).
// Results are bit-identical to the converting kernels on the same values (the producers apply the same two roundings).
//
// Reference work replaced: the cuDNN weight-gradient / forward / input-gradient convolutions autograd runs for ResnetBlock2D (reached from
// reference loss.py:993 and VillanDiffusion.py:1161 accelerator.backward) and the F.group_norm + F.silu that feed them.
#include "vd_common.h"
#include <stdlib.h>

namespace {

#include "vd_wgrad_job.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
    bf16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 t = (__bf16)v[j];
        h[j] = t;
        l[j] = (__bf16)(v[j] - (float)t);
    }
    hi = __builtin_bit_cast(u32x4, h);
    lo = __builtin_bit_cast(u32x4, l);
}

// ---- f32 NCHW <-> O8 (tests, and operands whose producer is not a GroupNorm: gradients out of convolution epilogues) ---------------------
// one thread per (octet, pixel): 8 pixel-coalesced 4-byte loads, two 16-byte stores
__global__ __launch_bounds__(256) void presplit_pack_kernel(const float* __restrict__ x, u32x4* __restrict__ y, int C8, int HW, int64_t x_bs, int64_t y_bs) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int o = blockIdx.y, b = blockIdx.z;
    if (p >= HW) return;
    const float* __restrict__ src = x + (int64_t)b * x_bs + (int64_t)o * 8 * HW + p;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)j * HW];
    u32x4 hi, lo;
    split8(v, hi, lo);
    u32x4* __restrict__ dst = reinterpret_cast<u32x4*>(reinterpret_cast<float*>(y) + (int64_t)b * y_bs) + ((int64_t)o * HW + p) * 2;
    dst[0] = hi;
    dst[1] = lo;
}

__global__ __launch_bounds__(256) void presplit_unpack_kernel(const u32x4* __restrict__ y, float* __restrict__ x, int C8, int HW, int64_t y_bs, int64_t x_bs) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int o = blockIdx.y, b = blockIdx.z;
    if (p >= HW) return;
    const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(y) + (int64_t)b * y_bs) + ((int64_t)o * HW + p) * 2;
    const bf16x8 h = __builtin_bit_cast(bf16x8, src[0]), l = __builtin_bit_cast(bf16x8, src[1]);
    float* __restrict__ dst = x + (int64_t)b * x_bs + (int64_t)o * 8 * HW + p;
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[(int64_t)j * HW] = (float)h[j] + (float)l[j];
}

// ---- GroupNorm (+ SiLU) forward writing the O8 image ---------------------------------------------------------------------------------------
// A workgroup owns NO whole channel octets = NG whole groups of one image (CPG channels per group: 4 -> 1 octet / 2 groups, 8 -> 1 / 1,
// 12 -> 3 / 2, 16 -> 2 / 1), register-resident like gn_fwd_reg_kernel: one read of x, one write of the pair image -- the algorithmic 8 B / element.
// Thread t holds the 8 channels of an octet at pixel t + NTH * it (pixel-coalesced 4-byte loads), so it owns whole 16-byte units on the way out.
template <int NW>
__device__ __forceinline__ void block_sum_n(float (&v)[2], int ng, float* red) {     // sums of v[0 .. ng) over the workgroup (NW waves), fixed order
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 2; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        red[w * 2] = v[0];
        red[w * 2 + 1] = v[1];
    }
    __syncthreads();
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        a0 += red[k * 2];
        a1 += red[k * 2 + 1];
    }
    v[0] = a0;
    v[1] = a1;
}

template <int CPG, int NO, int PIT, int NTH>
__global__ __launch_bounds__(NTH) void gn_fwd_ps_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        u32x4* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out, int C, int HW,
                                                        int G, float eps, int apply_silu, int64_t x_bs, int64_t y_bs) {
    constexpr int NG = NO * 8 / CPG;
    static_assert(NG == 1 || NG == 2, "one or two groups per workgroup");
    static_assert((NO * 8) % CPG == 0, "whole groups");
    __shared__ float red[2 * 2 * (NTH / 64)];
    const int blocks_per_img = C / (NO * 8);
    const int b = blockIdx.x / blocks_per_img, ob = blockIdx.x - b * blocks_per_img;
    const int c0 = ob * NO * 8;
    const int tid = threadIdx.x;
    const float* __restrict__ xb = x + (int64_t)b * x_bs + (int64_t)c0 * HW;
    float v[NO * PIT][8];
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int it = 0; it < PIT; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[o * PIT + it][j] = xb[(int64_t)(o * 8 + j) * HW + tid + it * NTH];
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int it = 0; it < PIT; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) s[(o * 8 + j) / CPG] += v[o * PIT + it][j];
    block_sum_n<NTH / 64>(s, NG, red);
    const float inv_n = 1.f / (float)(CPG * HW);
    const float mean[2] = {s[0] * inv_n, s[1] * inv_n};
    float q[2] = {0.f, 0.f};
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int it = 0; it < PIT; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = v[o * PIT + it][j] - mean[(o * 8 + j) / CPG];
                q[(o * 8 + j) / CPG] += a * a;
            }
    block_sum_n<NTH / 64>(q, NG, red + 2 * (NTH / 64));
    const float rstd[2] = {rsqrtf(q[0] * inv_n + eps), rsqrtf(q[1] * inv_n + eps)};
    if (tid < NG) {
        const int g = c0 / CPG + tid;
        mean_out[b * G + g] = mean[tid];
        rstd_out[b * G + g] = rstd[tid];
    }
    u32x4* __restrict__ yb = reinterpret_cast<u32x4*>(reinterpret_cast<float*>(y) + (int64_t)b * y_bs) + (int64_t)(c0 / 8) * HW * 2;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        float ga[8], be[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + o * 8 + j;
            ga[j] = gamma[c] * rstd[(o * 8 + j) / CPG];
            be[j] = beta[c] - mean[(o * 8 + j) / CPG] * ga[j];
        }
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            float z[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = v[o * PIT + it][j] * ga[j] + be[j];
                z[j] = apply_silu ? t * sigmoidf_(t) : t;
            }
            u32x4 hi, lo;
            split8(z, hi, lo);
            u32x4* __restrict__ dst = yb + ((int64_t)o * HW + tid + it * NTH) * 2;
            dst[0] = hi;
            dst[1] = lo;
        }
    }
}

// ---- GroupNorm (+ SiLU) backward writing dx as the O8 image (and / or as f32 NCHW) -----------------------------------------------------
// The same workgroup shape as gn_fwd_ps_kernel (NO whole octets = NG whole groups, 8 channels of one pixel per thread and item) and the arithmetic of
// gn_bwd_reg_kernel (vd_norm.hip): x and dy read once, xhat / dz kept in registers, per-channel sums (dbeta, dgamma rows) by wave butterflies + a
// fixed-order combine over the waves, then dx = rstd (dz gamma - m1 - xhat m2) + extra + extra2, its per-channel sums (the bias-gradient rows of the
// layer that produced x) from the same pass.  dx goes out as the pre-split image (dx_ps: the operand of the convolution's input gradient AND of its
// weight gradient -- ResnetBlock2D: norm2 backward -> conv1), as f32 (dx: where a residual add still needs it), or both (+4 B / element).
// 64-lane sum by DPP adds (six VALU instructions, no LDS crossbar traffic): the total arrives in LANE 63 only.  Fixed order.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0xB1, 0xF>(v);        // quad_perm [1, 0, 3, 2]
    v = dpp_add<0x4E, 0xF>(v);        // quad_perm [2, 3, 0, 1]
    v = dpp_add<0x141, 0xF>(v);       // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);       // row_mirror: every lane of a row holds the row's sum
    v = dpp_add<0x142, 0xA>(v);       // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);       // row_bcast:31 into rows 2 and 3
    return v;
}

template <int NCH, int NW>
__device__ __forceinline__ void channel_sums(const float (&v)[NCH], float* __restrict__ red, float* __restrict__ out) {     // out[c] = sum over the workgroup
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const float t = wave_sum_to_lane63(v[c]);
        if (lane == 63) red[w * NCH + c] = t;
    }
    __syncthreads();
    if (threadIdx.x < NCH) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) a += red[k * NCH + threadIdx.x];
        out[threadIdx.x] = a;
    }
    __syncthreads();
}

template <int CPG, int NO, int PIT, int NTH>
__global__ __launch_bounds__(NTH) void gn_bwd_ps_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean_in,
                                                        const float* __restrict__ rstd_in, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ extra, const float* __restrict__ extra2, float* __restrict__ dx,
                                                        u32x4* __restrict__ dx_ps, float* __restrict__ dgamma_ws, float* __restrict__ dbeta_ws, int C, int HW,
                                                        int G, int apply_silu, int64_t dy_bs, int64_t x_bs, int64_t ex_bs, int64_t ex2_bs, int64_t dx_bs,
                                                        int64_t ps_bs, float* __restrict__ rs_out, int64_t rs_ld) {
    constexpr int NG = NO * 8 / CPG, NCH = NO * 8, NW = NTH / 64, NI = NO * PIT;
    __shared__ float red[NW * NCH];
    __shared__ float ch1[NCH], ch2[NCH], ch3[NCH];
    const int blocks_per_img = C / NCH;
    const int b = blockIdx.x / blocks_per_img, ob = blockIdx.x - b * blocks_per_img;
    const int c0 = ob * NCH;
    const int tid = threadIdx.x;
    const float* __restrict__ xb = x + (int64_t)b * x_bs + (int64_t)c0 * HW;
    const float* __restrict__ db = dy + (int64_t)b * dy_bs + (int64_t)c0 * HW;
    float mean[NG], rstd[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        mean[k] = mean_in[b * G + c0 / CPG + k];
        rstd[k] = rstd_in[b * G + c0 / CPG + k];
    }
    float xh[NI][8], dz[NI][8];
    float s1[NCH], s2[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) s1[c] = s2[c] = 0.f;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        float ga[8], be[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ga[j] = gamma[c0 + o * 8 + j];
            be[j] = beta[c0 + o * 8 + j];
        }
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int p = tid + it * NTH;
            float xv[8], dv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xv[j] = xb[(int64_t)(o * 8 + j) * HW + p];
                dv[j] = db[(int64_t)(o * 8 + j) * HW + p];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int gi = (o * 8 + j) / CPG;
                const float h_ = (xv[j] - mean[gi]) * rstd[gi];
                float z_ = dv[j];
                if (apply_silu) {
                    const float z = h_ * ga[j] + be[j], sg = sigmoidf_(z);
                    z_ *= sg * (1.f + z * (1.f - sg));
                }
                xh[o * PIT + it][j] = h_;
                dz[o * PIT + it][j] = z_;
                s1[o * 8 + j] += z_;
                s2[o * 8 + j] += z_ * h_;
            }
        }
    }
    channel_sums<NCH, NW>(s1, red, ch1);
    channel_sums<NCH, NW>(s2, red, ch2);
    if (tid < NCH) {
        dbeta_ws[(int64_t)b * C + c0 + tid] = ch1[tid];
        dgamma_ws[(int64_t)b * C + c0 + tid] = ch2[tid];
    }
    float m1[NG], m2[NG];
    const float inv_n = 1.f / (float)(CPG * HW);
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        float a1 = 0.f, a2 = 0.f;
        for (int cl = 0; cl < CPG; ++cl) {
            const float ga = gamma[c0 + k * CPG + cl];
            a1 += ga * ch1[k * CPG + cl];
            a2 += ga * ch2[k * CPG + cl];
        }
        m1[k] = a1 * inv_n;
        m2[k] = a2 * inv_n;
    }
    const float* __restrict__ e1 = extra ? extra + (int64_t)b * ex_bs + (int64_t)c0 * HW : nullptr;
    const float* __restrict__ e2 = extra2 ? extra2 + (int64_t)b * ex2_bs + (int64_t)c0 * HW : nullptr;
    float* __restrict__ of = dx ? dx + (int64_t)b * dx_bs + (int64_t)c0 * HW : nullptr;
    u32x4* __restrict__ op = dx_ps ? reinterpret_cast<u32x4*>(reinterpret_cast<float*>(dx_ps) + (int64_t)b * ps_bs) + (int64_t)(c0 / 8) * HW * 2 : nullptr;
    float s3[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) s3[c] = 0.f;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        float ga[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ga[j] = gamma[c0 + o * 8 + j];
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int p = tid + it * NTH;
            float ov[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int gi = (o * 8 + j) / CPG;
                float ev = 0.f;
                if (e1) ev = e1[(int64_t)(o * 8 + j) * HW + p];
                if (e2) ev += e2[(int64_t)(o * 8 + j) * HW + p];
                ov[j] = rstd[gi] * (dz[o * PIT + it][j] * ga[j] - m1[gi] - xh[o * PIT + it][j] * m2[gi]) + ev;
                s3[o * 8 + j] += ov[j];
            }
            if (of) {
#pragma unroll
                for (int j = 0; j < 8; ++j) of[(int64_t)(o * 8 + j) * HW + p] = ov[j];
            }
            if (op) {
                u32x4 hi, lo;
                split8(ov, hi, lo);
                u32x4* __restrict__ dst = op + ((int64_t)o * HW + p) * 2;
                dst[0] = hi;
                dst[1] = lo;
            }
        }
    }
    if (rs_out) {
        channel_sums<NCH, NW>(s3, red, ch3);
        if (tid < NCH) rs_out[(int64_t)b * rs_ld + c0 + tid] = ch3[tid];
    }
}

// ---- split-precision 3x3 weight gradient with BOTH operands pre-split: LDS-DMA + transposed reads -----------------------------------------
// Tile geometry, grid, split-K plan, slab layout and job table of wgrad_k32_body (vd_wgrad_k32.inc): 128 rows m x 64 channels c x the 3 taps of
// tap row r per workgroup of four waves (64 m x 32 c each), K-step = 32 output pixels, one v_mfma_f32_16x16x32_bf16 per (tile, tap, product).
// What changes is how the operands get there:
//   * a K-step's dY rows (16 octets x 32 pixels x (hi, lo)) and X rows (8 octets) are 24 wave-wide LDS-DMA instructions of 1 KB (6 per wave):
//     lane l of an instruction writes unit [octet][part = l >> 5][slot = l & 31] of the stage image and reads pixel slot ^ 4 (octet & 1) of that
//     octet's row -- the XOR keeps the transposed reads of two neighbouring octets on different banks, and it costs nothing because the source
//     address of a DMA lane is free (the destination is lane-linear);
//   * three stage buffers, DMA issued two K-steps ahead, a counted s_waitcnt vmcnt(6) and a raw s_barrier per step (no drain of the queue);
//   * fragments: lane (g, q, p) of a 16-lane group supplies the address of pixel 8 g + 4 h + q (+ s - 1 for tap s of X), channels 4 p .. 4 p + 3, and
//     receives pixels 8 g + 4 h .. + 3 of channel l & 15: two reads per 8-deep fragment.  Pixels left / right of an image row are ONE zeroed region;
//     rows above / below the image are out-of-range DMA sources (zeros).  All read addresses are loop-invariant VGPRs + immediates (the K loop is
//     unrolled over the three stage buffers).
// No VALU beside the MFMAs except the per-step DMA offsets; 2 workgroups per CU (76 KB of LDS each).
// MODE 2 (the convolution behind Upsample2D: X is the HALF-resolution source of the nearest-2x upsample): a K-step's 32 upsampled pixels read 16
// source pixels per octet (image row (y0 + rr + r - 1) >> 1, pixel (x + s - 1) >> 1), so the X stage is [8 octets][2 parts][16 slots] = 4 KB, ONE
// DMA instruction per wave, and the tap shift / the doubling live in the row addresses of the transposed reads.
constexpr int PS_NST = 3;
constexpr int PS_A_U = 16 * 64;                                 // 16-byte units per stage: dY [16 octets][2 parts][32 slots]
constexpr int ps_b_units(int mode) { return mode == 2 ? 8 * 32 : 8 * 64; }      // X [8 octets][2 parts][32 | 16 slots]
constexpr int ps_stage_units(int mode) { return PS_A_U + ps_b_units(mode); }     // 1536 units = 24 KB (MODE 2: 1280 = 20 KB)
constexpr int PS_Z_U = 256;                                     // zero region: 4 KB (every (octet pair, part, channel tile) immediate of a padding lane lands in it)
constexpr int ps_lds_units(int mode) { return PS_NST * ps_stage_units(mode) + PS_Z_U; }

// The transposed reads are INLINE ASM with hand-counted lgkmcnt waits: hipcc's waitcnt pass treats an LDS read that may alias a pending LDS-DMA as
// dependent on it and puts s_waitcnt vmcnt(0) in front of the first ds_read of a step -- i.e. it waits for the DMAs issued a moment ago for the step
// after next and the prefetch is gone (seen in the ISA of the builtin version).  An asm read carries no memory operand, so only the counted waits below
// order it against the DMAs (vmcnt(6) + s_barrier at the end of the previous step).
template <int OFF>
__device__ __forceinline__ void tr_read(s16x4& dst, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
// s_waitcnt lgkmcnt(0), tied to the registers the caller is about to use (an MFMA on them cannot be scheduled above the wait)
__device__ __forceinline__ void lgkm_wait4(s16x4& a, s16x4& b, s16x4& c, s16x4& e) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(e)::"memory");
}
__device__ __forceinline__ bf16x8 cat8(const s16x4& lo4, const s16x4& hi4) {
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int W, int MODE>   // W: 8 | 16 | 32 (output width = pixels per image row); MODE 0: VD_B_CONV3, 2: VD_B_CONV3_UP
__device__ __forceinline__ void wgrad_ps_body(const vd_wgrad_desc& d, int ksteps_per_split, int gx, int gy, int lin, u32x4* lds) {
    constexpr int ROWS = 32 / W, OPR = W / 8, CT = 64;
    constexpr int PS_ST_U = ps_stage_units(MODE);
    constexpr int NDMA = MODE == 2 ? 5 : 6;          // LDS-DMA instructions per wave and stage (the counted vmcnt waits)
    constexpr int SLOTS = MODE == 2 ? 16 : 32;       // X slots per (octet, part) plane
    constexpr int B_CT = 2 * 2 * SLOTS * 16;         // bytes between the X fragments of channel tiles ct = 0 / 1 (two octets)
    constexpr int B_PART = SLOTS * 16;               // bytes between the hi and lo planes of an X octet
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_m = (d.M + 127) / 128;
    int bx = lin % gx, by = lin / gx;                // XCD-aware remap (as wgrad_k32_body)
    {
        const int T = gx * gy;
        if ((T & 7) == 0) {
            const int v = (lin & 7) * (T >> 3) + (lin >> 3);
            bx = v % gx;
            by = v / gx;
        }
    }
    const int r = bx % 3;
    const int rest = bx / 3;
    const int tm = rest % tiles_m, tc = rest / tiles_m;
    const int m0 = tm * 128, c0 = tc * CT;
    const int steps_per_img = d.OH / ROWS;
    const int ks_total = d.nb * steps_per_img;
    const int ks_begin = by * ksteps_per_split;
    const int ks_end = min(ks_total, ks_begin + ksteps_per_split);
    const int HWs = d.H * d.W;                       // source plane of X (MODE 2: the half-resolution image)

    // zero region (padding lanes of the shifted taps read it); nobody else ever writes it
    for (int i = tid; i < PS_Z_U; i += 256) lds[PS_NST * PS_ST_U + i] = u32x4{0u, 0u, 0u, 0u};

    // ---- DMA roles: wave w moves dY octets w, w + 4, w + 8, w + 12 and X octets w, w + 4 of every stage ----
    const int slot = lane & 31, part = lane >> 5;
    unsigned a_voff[4], b_fix[2];
    bool b_oct_ok[2];
    int b_rr[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int oa = wave + 4 * i;
        const int px = slot ^ (4 * (oa & 1));
        const bool ok = m0 + oa * 8 < d.M;
        a_voff[i] = ok ? 16u * (unsigned)((((m0 >> 3) + oa) * d.NP + px) * 2 + part) : 0xFFFFFFFFu;
    }
    if constexpr (MODE == 2) {                        // ONE instruction per wave: octets 2 w, 2 w + 1; lane = [octet][part][16 slots]
        const int ob = 2 * wave + (lane >> 5), pt = (lane >> 4) & 1;
        const int ls = (lane & 15) ^ (8 * (ob & 1));         // logical slot rr * (W / 2) + sx held by this lane's LDS slot (XOR: see the reads)
        b_oct_ok[0] = c0 + ob * 8 < d.C;
        b_rr[0] = ls / (W / 2);
        b_fix[0] = 16u * (unsigned)((((c0 >> 3) + ob) * HWs + (ls % (W / 2))) * 2 + pt);
        b_oct_ok[1] = false, b_rr[1] = 0, b_fix[1] = 0u;
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ob = wave + 4 * i;
            const int px = slot ^ (4 * (ob & 1));
            b_oct_ok[i] = c0 + ob * 8 < d.C;
            b_rr[i] = px / W;                            // image row of this lane's pixel inside the K-step
            b_fix[i] = 16u * (unsigned)((((c0 >> 3) + ob) * HWs + px) * 2 + part);
        }
    }
    auto dma_stage = [&](int ks, int sb) {
        const int b = ks / steps_per_img;
        const int y0 = (ks - b * steps_per_img) * ROWS;
        const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.dY + (int64_t)b * d.dy_bstride), 0, 0xFFFFFFF0, 0x00020000);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.X + (int64_t)b * d.x_bstride), 0, 0xFFFFFFF0, 0x00020000);
        const unsigned a_so = 32u * (unsigned)(y0 * W);                          // the K-step's 32 pixels are contiguous in every octet row
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(lds + sb * PS_ST_U + (wave + 4 * i) * 64);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dyr, dst, 16, a_voff[i], a_so, 0, 0);
        }
        const int yb = y0 + r - 1;                                               // first input row of the step (may be -1)
        if constexpr (MODE == 2) {
            const int uy = yb + b_rr[0];                                         // row of the (virtual) upsampled image -> source row uy >> 1
            const bool ok = b_oct_ok[0] && (unsigned)uy < (unsigned)(2 * d.H);
            const unsigned v = ok ? b_fix[0] + (unsigned)(32 * (uy >> 1) * d.W) : 0xFFFFFFFFu;
            __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(lds + sb * PS_ST_U + PS_A_U + wave * 64);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst, 16, v, 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool ok = b_oct_ok[i] && (unsigned)(yb + b_rr[i]) < (unsigned)d.H;
                const unsigned v = ok ? b_fix[i] + (unsigned)(32 * yb * W) : 0xFFFFFFFFu;     // a row outside the image is out of range: the DMA writes zeros
                __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(lds + sb * PS_ST_U + PS_A_U + (wave + 4 * i) * 64);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst, 16, v, 0, 0, 0);
            }
        }
    };

    // ---- fragment read addresses (bytes from the start of the LDS object, stage 0): loop-invariant ----
    const int wm = wave >> 1, wc = wave & 1;         // 2 x 2 waves: 64 m x 32 c each
    const int q = l15 >> 2, p = l15 & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
    unsigned a_addr[2], b_addr[3][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int px = 8 * g + 4 * h + q;
        const int oct = wm * 8 + (p >> 1);
        a_addr[h] = lds0 + 16u * (unsigned)(oct * 64 + (px ^ (4 * (p >> 1)))) + 8u * (unsigned)(p & 1);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int pxs = px + s - 1;
            const bool pad = (s == 0 && (g % OPR) == 0 && h == 0 && q == 0) || (s == 2 && (g % OPR) == OPR - 1 && h == 1 && q == 3);
            const int octb = wc * 4 + (p >> 1);
            unsigned in_img;
            if constexpr (MODE == 2) {                // upsampled pixel (row rr, column ux) reads source slot rr * (W / 2) + (ux >> 1); two octets 8 slots apart
                const int rr = px / W, ux = (px % W) + s - 1;
                const int ls = rr * (W / 2) + ((ux < 0 ? 0 : ux) >> 1);
                in_img = 16u * (unsigned)(PS_A_U + octb * 32 + ((ls & 15) ^ (8 * (p >> 1)))) + 8u * (unsigned)(p & 1);
            } else {
                in_img = 16u * (unsigned)(PS_A_U + octb * 64 + ((pxs & 31) ^ (4 * (p >> 1)))) + 8u * (unsigned)(p & 1);
            }
            // padding lanes: the zero region minus the stage offset the immediates add (the region is addressed from stage 0 for every stage)
            const unsigned zero = 16u * (unsigned)(PS_NST * PS_ST_U) + 16u * (unsigned)(2 * SLOTS) * (unsigned)(p >> 1) + 8u * (unsigned)(p & 1);
            b_addr[s][h] = lds0 + (pad ? zero : in_img);
        }
    }
    // padding lanes must not move with the stage: keep one address set per stage buffer (the in-image lanes add the stage offset, the others do not)
    unsigned b_st[PS_NST][3][2];
#pragma unroll
    for (int sb = 0; sb < PS_NST; ++sb)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool pad = (s == 0 && (g % OPR) == 0 && h == 0 && q == 0) || (s == 2 && (g % OPR) == OPR - 1 && h == 1 && q == 3);
                b_st[sb][s][h] = b_addr[s][h] + (pad ? 0u : 16u * (unsigned)(sb * PS_ST_U));
            }

    f32x4 acc[4][2][3];                              // [m tile][c tile][tap s]
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) acc[mt][ct][sx] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](auto SBc) {
        constexpr int SB = decltype(SBc)::value;
        constexpr int SOFF = SB * PS_ST_U * 16;
        // all 16 dY reads + the first X fragment pair, one wait; then per (tap, channel tile): the NEXT pair's 4 reads are issued in front of the 12 MFMAs
        // of the current one and waited for behind them (192 matrix cycles cover the LDS round trip)
        s16x4 ar[4][2][2];                           // [m tile][part][h]   (immediates: m tile = 2 octets = 2048 B, part = 512 B)
        tr_read<SOFF + 0 * 2048>(ar[0][0][0], a_addr[0]); tr_read<SOFF + 0 * 2048>(ar[0][0][1], a_addr[1]);
        tr_read<SOFF + 0 * 2048 + 512>(ar[0][1][0], a_addr[0]); tr_read<SOFF + 0 * 2048 + 512>(ar[0][1][1], a_addr[1]);
        tr_read<SOFF + 1 * 2048>(ar[1][0][0], a_addr[0]); tr_read<SOFF + 1 * 2048>(ar[1][0][1], a_addr[1]);
        tr_read<SOFF + 1 * 2048 + 512>(ar[1][1][0], a_addr[0]); tr_read<SOFF + 1 * 2048 + 512>(ar[1][1][1], a_addr[1]);
        tr_read<SOFF + 2 * 2048>(ar[2][0][0], a_addr[0]); tr_read<SOFF + 2 * 2048>(ar[2][0][1], a_addr[1]);
        tr_read<SOFF + 2 * 2048 + 512>(ar[2][1][0], a_addr[0]); tr_read<SOFF + 2 * 2048 + 512>(ar[2][1][1], a_addr[1]);
        tr_read<SOFF + 3 * 2048>(ar[3][0][0], a_addr[0]); tr_read<SOFF + 3 * 2048>(ar[3][0][1], a_addr[1]);
        tr_read<SOFF + 3 * 2048 + 512>(ar[3][1][0], a_addr[0]); tr_read<SOFF + 3 * 2048 + 512>(ar[3][1][1], a_addr[1]);
        s16x4 br[2][2][2];                           // [buffer][part][h]
        auto b_reads = [&](int k, s16x4 (&dst)[2][2]) {      // combination k = (tap order o = k >> 1, channel tile ct = k & 1); centre tap first
            const int o = k >> 1, ct = k & 1;
            const int sx = o == 0 ? 1 : (o == 1 ? 0 : 2);
            if (ct == 0) {
                tr_read<0>(dst[0][0], b_st[SB][sx][0]); tr_read<0>(dst[0][1], b_st[SB][sx][1]);
                tr_read<B_PART>(dst[1][0], b_st[SB][sx][0]); tr_read<B_PART>(dst[1][1], b_st[SB][sx][1]);
            } else {
                tr_read<B_CT>(dst[0][0], b_st[SB][sx][0]); tr_read<B_CT>(dst[0][1], b_st[SB][sx][1]);
                tr_read<B_CT + B_PART>(dst[1][0], b_st[SB][sx][0]); tr_read<B_CT + B_PART>(dst[1][1], b_st[SB][sx][1]);
            }
        };
        b_reads(0, br[0]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) lgkm_wait4(ar[mt][0][0], ar[mt][0][1], ar[mt][1][0], ar[mt][1][1]);
        lgkm_wait4(br[0][0][0], br[0][0][1], br[0][1][0], br[0][1][1]);
        bf16x8 ah[4], al[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            ah[mt] = cat8(ar[mt][0][0], ar[mt][0][1]);
            al[mt] = cat8(ar[mt][1][0], ar[mt][1][1]);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int cur = k & 1;
            const int o = k >> 1, ct = k & 1;
            const int sx = o == 0 ? 1 : (o == 1 ? 0 : 2);
            if (k < 5) b_reads(k + 1, br[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 xh = cat8(br[cur][0][0], br[cur][0][1]), xl = cat8(br[cur][1][0], br[cur][1][1]);
            // same product order per accumulator as wgrad_k32_body: lo * hi, hi * lo, hi * hi, the four m tiles inside each product
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][ct][sx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mt], xh, acc[mt][ct][sx], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][ct][sx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], xl, acc[mt][ct][sx], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][ct][sx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], xh, acc[mt][ct][sx], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (k < 5) lgkm_wait4(br[cur ^ 1][0][0], br[cur ^ 1][0][1], br[cur ^ 1][1][0], br[cur ^ 1][1][1]);
        }
    };

    // one K-step: the DMA of step ks + 2 goes into the buffer step ks - 1 was read from (every wave left it at the last barrier); after the MFMAs this
    // wave's DMAs of step ks + 1 must have landed (all but the 6 issued a moment ago) and its LDS reads must be done before the barrier
    auto step = [&](int ks, auto SBc) {
        constexpr int SB = decltype(SBc)::value;
        const int ks_last = ks_end - 1;
        dma_stage(min(ks + 2, ks_last), (SB + 2) % PS_NST);          // (past the end: the last step again, into a buffer nobody reads any more)
        __builtin_amdgcn_sched_barrier(0);
        compute(SBc);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    if (ks_begin < ks_end) {
        const int ks_last = ks_end - 1;
        dma_stage(ks_begin, 0);
        dma_stage(min(ks_begin + 1, ks_last), 1);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA) : "memory");      // stage 0 (and the zero region's stores) done
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int ks = ks_begin;
        for (; ks + 3 <= ks_end; ks += 3) {
            step(ks, std::integral_constant<int, 0>{});
            step(ks + 1, std::integral_constant<int, 1>{});
            step(ks + 2, std::integral_constant<int, 2>{});
        }
        if (ks < ks_end) step(ks, std::integral_constant<int, 0>{});
        if (ks + 1 < ks_end) step(ks + 1, std::integral_constant<int, 1>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the unused DMAs of the last two steps
    }

    // ---- store: lane (g, l15) holds rows m = 4 g + {0..3} of column c = l15 of every 16 x 16 tile (as wgrad_k32_body) ----
    const int Ncols = d.C * 9;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int c = c0 + wc * 32 + ct * 16 + l15;
        if (c >= d.C) continue;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int m = m0 + wm * 64 + mt * 16 + 4 * g + v;
                if (m >= d.M) continue;
                if (gy > 1) {                      // split-K slab ws[z][r][m][c][3] (wgrad_group_reduce_kernel un-permutes)
                    float* __restrict__ o = d.ws + (int64_t)by * d.M * Ncols + (int64_t)r * d.M * d.C * 3 + ((int64_t)m * d.C + c) * 3;
#pragma unroll
                    for (int sx = 0; sx < 3; ++sx) o[sx] = acc[mt][ct][sx][v];
                } else {
                    float* __restrict__ o = d.dW + (int64_t)m * Ncols + c * 9 + r * 3;
#pragma unroll
                    for (int sx = 0; sx < 3; ++sx) o[sx] = d.accumulate ? (o[sx] + acc[mt][ct][sx][v]) : acc[mt][ct][sx][v];
                }
            }
    }
}

template <int W, int MODE>
__global__ __launch_bounds__(256, 2) void wgrad_ps_group_kernel(const vd_wgrad_job* __restrict__ jobs, int n_jobs) {
    __shared__ u32x4 lds[ps_lds_units(MODE)];       // ONE LDS object (a second one beside an LDS-DMA target makes hipcc wait vmcnt(0) before every ds_read)
    const vd_wgrad_job* __restrict__ jb = jobs + wgrad_find_job(jobs, n_jobs, blockIdx.x, false);
    const int lin = blockIdx.x - jb->first_block;
    if (lin >= jb->gx * jb->gy) return;            // padding blocks between jobs
    const vd_wgrad_desc d = jb->d;
    wgrad_ps_body<W, MODE>(d, jb->ks_per, jb->gx, jb->gy, lin, lds);
}

}  // namespace

// ---- entry points -------------------------------------------------------------------------------------------------------------------------
extern "C" int vd_presplit_pack(const float* x, void* y, int B, int C, int HW, int64_t x_bstride, int64_t y_bstride, void* stream) {
    VD_REQUIRE(x && y && B > 0 && C > 0 && HW > 0 && C % 8 == 0, "vd_presplit_pack: bad arguments (C must be a multiple of 8)");
    VD_REQUIRE((((uintptr_t)y) & 15) == 0 && (y_bstride & 3) == 0, "vd_presplit_pack: the pair image must be 16-byte aligned");
    hipLaunchKernelGGL(presplit_pack_kernel, dim3(vd_cdiv(HW, 256), C / 8, B), dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<u32x4*>(y), C / 8, HW,
                       x_bstride, y_bstride);
    VD_LAUNCH_CHECK("vd_presplit_pack");
    return 0;
}

extern "C" int vd_presplit_unpack(const void* y, float* x, int B, int C, int HW, int64_t y_bstride, int64_t x_bstride, void* stream) {
    VD_REQUIRE(x && y && B > 0 && C > 0 && HW > 0 && C % 8 == 0, "vd_presplit_unpack: bad arguments (C must be a multiple of 8)");
    VD_REQUIRE((((uintptr_t)y) & 15) == 0 && (y_bstride & 3) == 0, "vd_presplit_unpack: the pair image must be 16-byte aligned");
    hipLaunchKernelGGL(presplit_unpack_kernel, dim3(vd_cdiv(HW, 256), C / 8, B), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const u32x4*>(y), x, C / 8,
                       HW, y_bstride, x_bstride);
    VD_LAUNCH_CHECK("vd_presplit_unpack");
    return 0;
}

// 1 when vd_groupnorm_fwd_presplit has a kernel for this shape (register-resident octet-aligned groups: the 16x16 / 32x32 levels of the DDPM UNets)
extern "C" int vd_groupnorm_fwd_presplit_ok(int C, int HW, int G) {
    if (C <= 0 || G <= 0 || C % G || C % 8) return 0;
    const int cpg = C / G;
    if (HW == 1024) return (cpg == 4 || cpg == 8 || cpg == 12) ? 1 : 0;
    if (HW == 256) return (cpg == 4 || cpg == 8 || cpg == 12 || cpg == 16) ? 1 : 0;
    return 0;
}

extern "C" int vd_groupnorm_fwd_presplit(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int B, int C, int HW, int G,
                                         float eps, int apply_silu, int64_t x_bstride, int64_t y_bstride, void* stream) {
    VD_REQUIRE(x && gamma && beta && y && mean && rstd && B > 0, "vd_groupnorm_fwd_presplit: null pointer");
    VD_REQUIRE(vd_groupnorm_fwd_presplit_ok(C, HW, G), "vd_groupnorm_fwd_presplit: no kernel for C=%d HW=%d G=%d (vd_groupnorm_fwd_presplit_ok)", C, HW, G);
    VD_REQUIRE((((uintptr_t)y) & 15) == 0 && (y_bstride & 3) == 0, "vd_groupnorm_fwd_presplit: the pair image must be 16-byte aligned");
    if (const int rc = vd_gn_sticky("vd_groupnorm_fwd_presplit")) return rc;      // a poll timeout elsewhere: its NaN statistics must not be consumed here either
    const int cpg = C / G;
    hipStream_t st = (hipStream_t)stream;
    u32x4* yy = reinterpret_cast<u32x4*>(y);
#define VD_GN_PS(CPG_, NO_, PIT_, NTH_)                                                                                                          \
    hipLaunchKernelGGL((gn_fwd_ps_kernel<CPG_, NO_, PIT_, NTH_>), dim3(B * (C / (NO_ * 8))), dim3(NTH_), 0, st, x, gamma, beta, yy, mean, rstd, C, HW, G, \
                       eps, apply_silu, x_bstride, y_bstride)
    if (HW == 1024) {
        if (cpg == 4) VD_GN_PS(4, 1, 4, 256);
        else if (cpg == 8) VD_GN_PS(8, 1, 4, 256);
        else VD_GN_PS(12, 3, 2, 512);
    } else {
        if (cpg == 4) VD_GN_PS(4, 1, 1, 256);
        else if (cpg == 8) VD_GN_PS(8, 1, 1, 256);
        else if (cpg == 12) VD_GN_PS(12, 3, 1, 256);
        else VD_GN_PS(16, 2, 1, 256);
    }
#undef VD_GN_PS
    VD_LAUNCH_CHECK("vd_groupnorm_fwd_presplit");
    return 0;
}

// GroupNorm (+ SiLU) backward for the shapes vd_groupnorm_fwd_presplit_ok() accepts: dx as f32 (dx, nullable), as the pre-split image (dx_ps, nullable) or
// both; everything else as vd_groupnorm_bwd_fused (dgamma_ws / dbeta_ws rows per (image, channel), extra / extra2 residual gradients, rowsum of dx).
extern "C" int vd_groupnorm_bwd_presplit(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                         const float* extra, const float* extra2, float* dx, void* dx_ps, float* dgamma_ws, float* dbeta_ws, float* rowsum,
                                         int B, int C, int HW, int G, int apply_silu, int64_t dy_bstride, int64_t x_bstride, int64_t extra_bstride,
                                         int64_t extra2_bstride, int64_t dx_bstride, int64_t ps_bstride, int64_t rowsum_ld, void* stream) {
    VD_REQUIRE(dy && x && mean && rstd && gamma && beta && dgamma_ws && dbeta_ws && (dx || dx_ps) && B > 0, "vd_groupnorm_bwd_presplit: null pointer");
    VD_REQUIRE(vd_groupnorm_fwd_presplit_ok(C, HW, G), "vd_groupnorm_bwd_presplit: no kernel for C=%d HW=%d G=%d (vd_groupnorm_fwd_presplit_ok)", C, HW, G);
    VD_REQUIRE(!dx_ps || ((((uintptr_t)dx_ps) & 15) == 0 && (ps_bstride & 3) == 0), "vd_groupnorm_bwd_presplit: the pair image must be 16-byte aligned");
    VD_REQUIRE(!rowsum || rowsum_ld >= C, "vd_groupnorm_bwd_presplit: rowsum_ld < C");
    if (const int rc = vd_gn_sticky("vd_groupnorm_bwd_presplit")) return rc;
    const int cpg = C / G;
    hipStream_t st = (hipStream_t)stream;
    u32x4* pp = reinterpret_cast<u32x4*>(dx_ps);
#define VD_GNB_PS(CPG_, NO_, PIT_, NTH_)                                                                                                          \
    hipLaunchKernelGGL((gn_bwd_ps_kernel<CPG_, NO_, PIT_, NTH_>), dim3(B * (C / (NO_ * 8))), dim3(NTH_), 0, st, dy, x, mean, rstd, gamma, beta, extra,    \
                       extra2, dx, pp, dgamma_ws, dbeta_ws, C, HW, G, apply_silu, dy_bstride, x_bstride, extra_bstride, extra2_bstride, dx_bstride,     \
                       ps_bstride, rowsum, rowsum_ld)
    if (HW == 1024) {
        if (cpg == 4) VD_GNB_PS(4, 1, 4, 256);
        else if (cpg == 8) VD_GNB_PS(8, 1, 4, 256);
        else VD_GNB_PS(12, 3, 2, 512);
    } else {
        if (cpg == 4) VD_GNB_PS(4, 1, 1, 256);
        else if (cpg == 8) VD_GNB_PS(8, 1, 1, 256);
        else if (cpg == 12) VD_GNB_PS(12, 3, 1, 256);
        else VD_GNB_PS(16, 2, 1, 256);
    }
#undef VD_GNB_PS
    VD_LAUNCH_CHECK("vd_groupnorm_bwd_presplit");
    return 0;
}

// vd_gemm.hip's grouped launch (class 3000 + 4 W + 2 * upsample-fused): both operands pre-split
int vd_launch_wgrad_ps_group(const void* jobs, int n, int W, int up, int blocks, hipStream_t st) {
    const vd_wgrad_job* jb = reinterpret_cast<const vd_wgrad_job*>(jobs);
    // 4.5 KB of (unused) dynamic LDS on top of the kernel's 76 KB: ONE workgroup per CU instead of two (2 x 80.5 KB > 160 KB).  Two of them hold a CU's whole
    // LDS and register file for ~90 us at a time, and the backward pass on the main stream -- latency-bound 8x8 / 4x4 kernels, GroupNorm passes -- only gets a CU
    // when one retires; with one per CU every CU keeps half its registers and 79.5 KB of LDS for the main stream's workgroups all the time: enough for a 74-KB
    // workgroup of the whole-K 8x8 convolution (a 12-KB pad, the first version, left 72 KB: 17.07 against 17.005 ms per step, six interleaved pairs).  The weight
    // gradients themselves lose nothing measurable (they are power-bound beside the convolutions).  Against two per CU, same box, interleaved: -0.04 / -0.13 /
    // -0.07 ms per training step on three boxes with the 12-KB pad (profiles/r06_wgrad_occupancy_ab.txt).  VD_WGRAD_PS_LDS_PAD=0: two per CU.
    static const int pad = getenv("VD_WGRAD_PS_LDS_PAD") ? atoi(getenv("VD_WGRAD_PS_LDS_PAD")) : 4608;
#define VD_WG_PS(WW)                                                                                                \
    case WW:                                                                                                        \
        if (up) hipLaunchKernelGGL((wgrad_ps_group_kernel<WW, 2>), dim3(blocks), dim3(256), pad, st, jb, n);        \
        else hipLaunchKernelGGL((wgrad_ps_group_kernel<WW, 0>), dim3(blocks), dim3(256), pad, st, jb, n);           \
        return 0;
    switch (W) {
        VD_WG_PS(32) VD_WG_PS(16) VD_WG_PS(8)
        default: return -1;
    }
#undef VD_WG_PS
}
