// K4: attention softmax (key-major score matrix) and the tiny-token-count attention of the UNet mid block.
//
// The QK^T and PV contractions of the 16x16-token attention blocks run on the f32 MFMA through vd_gemm (vd_gemm.hip)
// in NCHW with no transposes:  St[j][i] = scale * sum_c k[c][j] q[c][i]  (A = k column-major, B = q plain),
// softmax over j per column i (this file),  out[c][i] = sum_j v[c][j] P[j][i].
// Replaces torch.baddbmm/softmax/bmm of diffusers' AttentionBlock (UNet2DModel, reached from loss.py:993).
#include "vd_common.h"
#include <stdlib.h>

namespace {

// One workgroup per (batch item, 64 columns); wave w owns rows j = w, w+4, ...  Online (max,sum) then one write pass.
__global__ __launch_bounds__(256) void softmax_col_fwd_kernel(float* __restrict__ S, int N) {
    __shared__ float smax[4][64], ssum[4][64];
    const int chunks = (N + 63) / 64;
    const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = ch * 64 + lane;
    float* __restrict__ Sb = S + (int64_t)b * N * N;
    float m = -INFINITY, s = 0.f;
    if (i < N) {
        for (int j = w; j < N; j += 4) {
            const float v = Sb[(int64_t)j * N + i];
            const float mn = fmaxf(m, v);
            s = s * __expf(m - mn) + __expf(v - mn);
            m = mn;
        }
    }
    smax[w][lane] = m;
    ssum[w][lane] = s;
    __syncthreads();
    float M = fmaxf(fmaxf(smax[0][lane], smax[1][lane]), fmaxf(smax[2][lane], smax[3][lane]));
    float Z = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) Z += (smax[k][lane] == -INFINITY) ? 0.f : ssum[k][lane] * __expf(smax[k][lane] - M);
    const float inv = 1.f / Z;
    if (i < N) {
        for (int j = w; j < N; j += 4) {
            const int64_t o = (int64_t)j * N + i;
            Sb[o] = __expf(Sb[o] - M) * inv;
        }
    }
}

// N <= 256: the wave's N/4 <= 64 rows of a column stay in registers -- one read, one exp and one write per score (the general
// kernel above re-reads the tile and evaluates three exponentials per element).
template <int R>   // rows per wave = ceil(N / 4) <= R
__global__ __launch_bounds__(256) void softmax_col_fwd_reg_kernel(float* __restrict__ S, int N) {
    __shared__ float smax[4][64], ssum[4][64];
    const int chunks = (N + 63) / 64;
    const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = ch * 64 + lane;
    const bool col = i < N;
    float* __restrict__ Sb = S + (int64_t)b * N * N + (col ? i : 0);
    float v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int j = w + 4 * k;
        v[k] = (col && j < N) ? Sb[(int64_t)j * N] : -INFINITY;
    }
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < R; ++k) m = fmaxf(m, v[k]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        v[k] = (m == -INFINITY) ? 0.f : __expf(v[k] - m);
        s += v[k];
    }
    smax[w][lane] = m;
    ssum[w][lane] = s;
    __syncthreads();
    const float M = fmaxf(fmaxf(smax[0][lane], smax[1][lane]), fmaxf(smax[2][lane], smax[3][lane]));
    float Z = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) Z += (smax[k][lane] == -INFINITY) ? 0.f : ssum[k][lane] * __expf(smax[k][lane] - M);
    const float f = (m == -INFINITY) ? 0.f : __expf(m - M) / Z;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int j = w + 4 * k;
        if (col && j < N) Sb[(int64_t)j * N] = v[k] * f;
    }
}

// dS = scale * P * (dP - sum_j P*dP), in place on dP; P and dP held in registers (N <= 256).
template <int R>
__global__ __launch_bounds__(256) void softmax_col_bwd_reg_kernel(const float* __restrict__ P, float* __restrict__ dP, int N, float scale) {
    __shared__ float sdot[4][64];
    const int chunks = (N + 63) / 64;
    const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = ch * 64 + lane;
    const bool col = i < N;
    const float* __restrict__ Pb = P + (int64_t)b * N * N + (col ? i : 0);
    float* __restrict__ Db = dP + (int64_t)b * N * N + (col ? i : 0);
    float p[R], g[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int j = w + 4 * k;
        const bool ok = col && j < N;
        p[k] = ok ? Pb[(int64_t)j * N] : 0.f;
        g[k] = ok ? Db[(int64_t)j * N] : 0.f;
    }
    float d = 0.f;
#pragma unroll
    for (int k = 0; k < R; ++k) d += p[k] * g[k];
    sdot[w][lane] = d;
    __syncthreads();
    const float dot = (sdot[0][lane] + sdot[1][lane]) + (sdot[2][lane] + sdot[3][lane]);
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int j = w + 4 * k;
        if (col && j < N) Db[(int64_t)j * N] = scale * p[k] * (g[k] - dot);
    }
}

// dS = scale * P * (dP - sum_j P*dP), in place on dP.
__global__ __launch_bounds__(256) void softmax_col_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, int N,
                                                              float scale) {
    __shared__ float sdot[4][64];
    const int chunks = (N + 63) / 64;
    const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = ch * 64 + lane;
    const float* __restrict__ Pb = P + (int64_t)b * N * N;
    float* __restrict__ Db = dP + (int64_t)b * N * N;
    float d = 0.f;
    if (i < N)
        for (int j = w; j < N; j += 4) d += Pb[(int64_t)j * N + i] * Db[(int64_t)j * N + i];
    sdot[w][lane] = d;
    __syncthreads();
    const float dot = (sdot[0][lane] + sdot[1][lane]) + (sdot[2][lane] + sdot[3][lane]);
    if (i < N)
        for (int j = w; j < N; j += 4) {
            const int64_t o = (int64_t)j * N + i;
            Db[o] = scale * Pb[o] * (Db[o] - dot);
        }
}

// Whole attention for N <= 64 tokens: one workgroup per batch item (mid block: N = 16, C = 256; < 0.1 % of FLOPs).
// STAGE (round 4): q, k, v (and dout in the backward) of the image are copied to LDS first (coalesced) when they fit 64 KB -- the mid block of config #2
// (C = 256, N = 16): every thread's channel loop then reads LDS instead of issuing 2 x C strided global loads one after the other (56 / 85 us -> see
// profiles/r04_attn_small.txt).
template <bool STAGE>
__global__ __launch_bounds__(256) void attn_small_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                             float* __restrict__ Pout, int C, int N, float scale,
                                                             int64_t qkv_bs, int64_t out_bs) {
    __shared__ float Ps[64 * 64];
    __shared__ float stage[STAGE ? 12288 : 1];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* __restrict__ q = qkv + (int64_t)b * qkv_bs;
    const float* __restrict__ k = q + (int64_t)C * N;
    const float* __restrict__ v = k + (int64_t)C * N;
    if (STAGE) {
        const int CN = C * N;                             // 3 C N <= 12288
        for (int e = tid; e < 3 * CN; e += 256) stage[e] = q[e];         // q | k | v are contiguous in qkv
        __syncthreads();
        q = stage;
        k = stage + CN;
        v = stage + 2 * CN;
    }
    for (int e = tid; e < N * N; e += 256) {
        const int j = e / N, i = e - j * N;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(k[c * N + j], q[c * N + i], s);
        Ps[e] = s * scale;
    }
    __syncthreads();
    if (tid < N) {
        const int i = tid;
        float m = -INFINITY;
        for (int j = 0; j < N; ++j) m = fmaxf(m, Ps[j * N + i]);
        float z = 0.f;
        for (int j = 0; j < N; ++j) {
            const float e = __expf(Ps[j * N + i] - m);
            Ps[j * N + i] = e;
            z += e;
        }
        const float inv = 1.f / z;
        for (int j = 0; j < N; ++j) Ps[j * N + i] *= inv;
    }
    __syncthreads();
    if (Pout)
        for (int e = tid; e < N * N; e += 256) Pout[(int64_t)b * N * N + e] = Ps[e];
    float* __restrict__ o = out + (int64_t)b * out_bs;
    for (int e = tid; e < C * N; e += 256) {
        const int c = e / N, i = e - c * N;
        float s = 0.f;
        for (int j = 0; j < N; ++j) s = fmaf(v[c * N + j], Ps[j * N + i], s);
        o[e] = s;
    }
}

template <bool STAGE>
__global__ __launch_bounds__(256) void attn_small_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                             const float* __restrict__ dout, float* __restrict__ dqkv, int C,
                                                             int N, float scale, int64_t qkv_bs, int64_t dout_bs,
                                                             int64_t dqkv_bs) {
    __shared__ float Ps[64 * 64];
    __shared__ float dS[64 * 64];
    __shared__ float dots[64];
    __shared__ float stage[STAGE ? 16384 : 1];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* __restrict__ q = qkv + (int64_t)b * qkv_bs;
    const float* __restrict__ k = q + (int64_t)C * N;
    const float* __restrict__ v = k + (int64_t)C * N;
    const float* __restrict__ dob = dout + (int64_t)b * dout_bs;
    if (STAGE) {
        const int CN = C * N;                             // 4 C N <= 16384
        for (int e = tid; e < 3 * CN; e += 256) stage[e] = q[e];
        for (int e = tid; e < CN; e += 256) stage[3 * CN + e] = dob[e];
        __syncthreads();
        q = stage;
        k = stage + CN;
        v = stage + 2 * CN;
        dob = stage + 3 * CN;
    }
    float* __restrict__ dq = dqkv + (int64_t)b * dqkv_bs;
    float* __restrict__ dk = dq + (int64_t)C * N;
    float* __restrict__ dv = dk + (int64_t)C * N;
    for (int e = tid; e < N * N; e += 256) {
        Ps[e] = P[(int64_t)b * N * N + e];
        const int j = e / N, i = e - j * N;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(v[c * N + j], dob[c * N + i], s);
        dS[e] = s;  // dP
    }
    __syncthreads();
    if (tid < N) {
        float d = 0.f;
        for (int j = 0; j < N; ++j) d += Ps[j * N + tid] * dS[j * N + tid];
        dots[tid] = d;
    }
    __syncthreads();
    for (int e = tid; e < N * N; e += 256) {
        const int i = e % N;
        dS[e] = scale * Ps[e] * (dS[e] - dots[i]);
    }
    __syncthreads();
    for (int e = tid; e < C * N; e += 256) {
        const int c = e / N, x = e - c * N;
        float sv = 0.f, sq = 0.f, sk = 0.f;
        for (int y = 0; y < N; ++y) {
            sv = fmaf(dob[c * N + y], Ps[x * N + y], sv);  // dv[c][j=x] = sum_i dout[c][i] P[j][i]
            sq = fmaf(k[c * N + y], dS[y * N + x], sq);    // dq[c][i=x] = sum_j k[c][j] dS[j][i]
            sk = fmaf(q[c * N + y], dS[x * N + y], sk);    // dk[c][j=x] = sum_i q[c][i] dS[j][i]
        }
        dv[e] = sv;
        dq[e] = sq;
        dk[e] = sk;
    }
}


// ---- K4 fused: flash-style attention core for N = 256 tokens, split-precision (bf16x3) contractions on the bf16 matrix cores -------
// Replaces the three launches  St = scale k^T q  ->  softmax over j  ->  o = v P  (and, backward, dP = v^T do -> dS -> dq = k dS)
// by one: the 256 x 256 score / probability matrix of a (batch, head) lives in the accumulator registers and never reaches HBM in the
// no-grad path (training writes P once for the backward pass; backward writes dS once for the dk product).
//
// Work split: one workgroup per (batch, head, 128 query columns i); wave w owns the 32 columns i0 + 32 w and ALL 256 keys j, so the
// softmax reduction over j is lane-local (128 values per lane) plus one exchange with lane ^ 32 -- no LDS, no barrier.
//   phase 1   X[j][i] = sum_c A1[c][j] B1[c][i]      (forward: A1 = k, B1 = q; backward: A1 = v, B1 = do)  8 accumulator tiles / wave
//   softmax   forward:  P = softmax_j(scale X);   backward:  dS = scale P (X - sum_j P X)  with the saved P loaded into the same layout
//   phase 2   Y[c][i] = sum_j A2[c][j] T[j][i]       (forward: A2 = v, T = P -> o;  backward: A2 = k, T = dS -> dq)
// The phase-1 accumulator is the phase-2 B operand as it stands (guide: "an accumulator tile as the next MFMA's operand"): registers
// 8s..8s+7 of tile jt are the k-fragment of keys  32 jt + 16 s + 8 (e >> 2) + 4 h + (e & 3)  (e = element, h = lane >> 5), so the
// A2 fragments are staged in that key order.  Every product is hi*hi + hi*lo + lo*hi over bf16 halves (f32 accumulation), the halves
// made where an operand is written to LDS (A1, B1, A2) or taken from the registers (T).
// One workgroup per CU (a wave may use the whole 512-register file): 128 + 128 accumulator registers + 48 staging registers.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8(const float (&v)[8], u32x4_t& hi, u32x4_t& lo) {
    bf16x8_t h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 t = (__bf16)v[j];
        h[j] = t;
        l[j] = (__bf16)(v[j] - (float)t);
    }
    hi = __builtin_bit_cast(u32x4_t, h);
    lo = __builtin_bit_cast(u32x4_t, l);
}

struct attn_core_args {
    const float* a1; const float* b1; const float* a2;   // [batch][channels][256] slices; channel stride 256
    float* y;                                            // phase-2 output, same form
    float* t_out;                                        // forward: P or null; backward: dS   ([batch*heads][256 j][256 i])
    const float* p_in;                                   // backward: saved P
    const float* o_in;                                   // backward: saved forward output o (for delta_i = sum_c do[c][i] o[c][i])
    int64_t a1_bs, b1_bs, a2_bs, y_bs, o_bs;             // batch strides (floats)
    int heads, d;                                        // head h = channels [h*d, (h+1)*d) of every slice
    float scale;
};

template <int DT, bool BWD>   // DT = 32-channel tiles of the head handled per phase-2 pass (d = 32 * DT * passes)
__global__ __launch_bounds__(256, 1) void attn_core_kernel(const attn_core_args g) {
    constexpr int N = 256, JT = 8;
    constexpr int BUF1 = 8 * 256 + 8 * 128;              // phase 1: A planes [chunk][part][octet][256 j] + B planes [..][128 i], units of 16 B
    constexpr int VPL = 258;                             // phase 2: plane stride (units); 258 = 2 mod 8 keeps the 4 planes a row is written to on distinct banks
    constexpr int BUF2 = 8 * VPL;
    static_assert(2 * BUF2 <= 2 * BUF1, "phase 2 reuses the phase-1 buffers");
    __shared__ u32x4_t lds[2 * BUF1];                    // 96 KB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;                       // the two column halves of a (batch, head) on one XCD: they share A1 / A2
        if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    }
    const int bh = bid >> 1, i0 = (bid & 1) * 128;
    const int b = bh / g.heads, hd = bh - b * g.heads;
    const int d = g.d;
    const float* __restrict__ A1 = g.a1 + (int64_t)b * g.a1_bs + (int64_t)hd * d * N;
    const float* __restrict__ B1 = g.b1 + (int64_t)b * g.b1_bs + (int64_t)hd * d * N + i0;
    const float* __restrict__ O1 = BWD ? g.o_in + (int64_t)b * g.o_bs + (int64_t)hd * d * N + i0 : nullptr;
    const float* __restrict__ A2 = g.a2 + (int64_t)b * g.a2_bs + (int64_t)hd * d * N;
    float* __restrict__ Yb = g.y + (int64_t)b * g.y_bs + (int64_t)hd * d * N + i0;

    // ---------------------------------------------------------------- phase 1
    // One wave per SIMD: nothing else hides a load's latency, so the loads run TWO stages ahead (two register sets, stages unrolled by
    // two) and a stage is one branch-free block  {issue loads of stage s + 2 | 48 MFMAs of stage s | split + LDS writes of stage s + 1}.
    f32x16 sacc[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int v = 0; v < 16; ++v) sacc[jt][v] = 0.f;

    float ra0[32], rb0[16], ro0[BWD ? 16 : 1], ra1[32], rb1[16], ro1[BWD ? 16 : 1];
    float dpart = 0.f;                                   // backward: this thread's share of delta for column bi
    const int bi = tid & 127, bo = tid >> 7;             // B1 item: column, octet (and octet + 2)
    const int nst1 = d / 32;
    auto load1 = [&](int st, float (&ra)[32], float (&rb)[16], float (&ro)[BWD ? 16 : 1]) {     // stage = 32 channels
        st = min(st, nst1 - 1);                          // past the end: reload the last stage (its copy is never read)
        const float* __restrict__ a = A1 + (int64_t)(st * 32) * N + tid;
#pragma unroll
        for (int c = 0; c < 32; ++c) ra[c] = a[(int64_t)c * N];
        const float* __restrict__ bq = B1 + (int64_t)(st * 32) * N + bi;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int c = 0; c < 8; ++c) rb[8 * e + c] = bq[(int64_t)((bo + 2 * e) * 8 + c) * N];
        if (BWD) {
            const float* __restrict__ oq = O1 + (int64_t)(st * 32) * N + bi;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int c = 0; c < 8; ++c) ro[8 * e + c] = oq[(int64_t)((bo + 2 * e) * 8 + c) * N];
        }
    };
    auto store1 = [&](int buf, const float (&ra)[32], const float (&rb)[16], const float (&ro)[BWD ? 16 : 1], bool real) {
        u32x4_t* __restrict__ As = lds + buf * BUF1;
        u32x4_t* __restrict__ Bs = As + 8 * 256;
#pragma unroll
        for (int o = 0; o < 4; ++o) {                    // octet o = chunk (o >> 1), k-octet (o & 1)
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = ra[8 * o + c];
            u32x4_t hi, lo;
            split8(v, hi, lo);
            As[(((o >> 1) * 2 + 0) * 2 + (o & 1)) * 256 + tid] = hi;
            As[(((o >> 1) * 2 + 1) * 2 + (o & 1)) * 256 + tid] = lo;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int o = bo + 2 * e;
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = rb[8 * e + c];
            if (BWD) {
                float t = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) t += rb[8 * e + c] * ro[8 * e + c];
                dpart += real ? t : 0.f;
            }
            u32x4_t hi, lo;
            split8(v, hi, lo);
            Bs[(((o >> 1) * 2 + 0) * 2 + (o & 1)) * 128 + bi] = hi;
            Bs[(((o >> 1) * 2 + 1) * 2 + (o & 1)) * 128 + bi] = lo;
        }
    };
    auto mfma1 = [&](int buf) {
        const u32x4_t* __restrict__ As = lds + buf * BUF1 + h * 256 + l31;
        const u32x4_t* __restrict__ Bs = lds + buf * BUF1 + 8 * 256 + h * 128 + 32 * wave + l31;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const bf16x8_t bhf = __builtin_bit_cast(bf16x8_t, Bs[((ch * 2 + 0) * 2) * 128]);
            const bf16x8_t blf = __builtin_bit_cast(bf16x8_t, Bs[((ch * 2 + 1) * 2) * 128]);
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, As[((ch * 2 + 0) * 2) * 256 + 32 * jt]);
                const bf16x8_t al = __builtin_bit_cast(bf16x8_t, As[((ch * 2 + 1) * 2) * 256 + 32 * jt]);
                sacc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhf, sacc[jt], 0, 0, 0);
                sacc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blf, sacc[jt], 0, 0, 0);
                sacc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhf, sacc[jt], 0, 0, 0);
            }
        }
    };
    load1(0, ra0, rb0, ro0);
    load1(1, ra1, rb1, ro1);
    store1(0, ra0, rb0, ro0, true);
    __syncthreads();
    for (int st = 0; st < nst1; st += 2) {
        load1(st + 2, ra0, rb0, ro0);
        mfma1(0);
        store1(1, ra1, rb1, ro1, st + 1 < nst1);
        __syncthreads();
        if (st + 1 >= nst1) break;
        load1(st + 3, ra1, rb1, ro1);
        mfma1(1);
        store1(0, ra0, rb0, ro0, st + 2 < nst1);
        __syncthreads();
    }

    // ---------------------------------------------------------------- softmax (forward) / its gradient (backward), in registers
    // lane (l31, h) of wave w holds column i = i0 + 32 w + l31, rows j = 32 jt + (v & 3) + 8 (v >> 2) + 4 h.
    // wave-uniform base (SGPR pair) + one 32-bit lane offset: the 128 rows of a lane are then constants added to that offset, not
    // 128 separate 64-bit addresses
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int64_t t_base = ((int64_t)bh * N) * N + i0 + 32 * wave_u;
    const unsigned t_lane = (unsigned)(l31 + 4 * h * N);
    float* __restrict__ t_out = g.t_out + t_base;
    if (!BWD) {
        float mx = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                sacc[jt][v] *= g.scale;
                mx = fmaxf(mx, sacc[jt][v]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                sacc[jt][v] = __expf(sacc[jt][v] - mx);
                sum += sacc[jt][v];
            }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int v = 0; v < 16; ++v) sacc[jt][v] *= inv;
    } else {
        // delta_i = sum_j P[j][i] dP[j][i] = sum_c do[c][i] o[c][i]  (o = v P): gathered while do was staged, so P is read ONCE
        float* __restrict__ dsum = reinterpret_cast<float*>(lds);
        dsum[tid] = dpart;                               // (column bi, octet half bo): the phase-1 buffers are free after its last barrier
        __syncthreads();
        const float delta = dsum[32 * wave + l31] + dsum[128 + 32 * wave + l31];
        const float* __restrict__ p_in = g.p_in + t_base;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
            for (int v = 0; v < 16; ++v)
                sacc[jt][v] = g.scale * p_in[t_lane + (unsigned)((32 * jt + (v & 3) + 8 * (v >> 2)) * N)] * (sacc[jt][v] - delta);
            asm volatile("" ::: "memory");               // one tile's 16 loads in flight at a time: 128 hoisted loads cost 128 registers
        }
    }
    // T leaves for HBM (P for the backward pass / dS for the dk product) and becomes the phase-2 B operand: bf16 (hi, lo) k-fragments,
    // made once, tile by tile (the f32 tile dies as its halves are made)
    u32x4_t thi[JT][2], tlo[JT][2];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        if (g.t_out) {
#pragma unroll
            for (int v = 0; v < 16; ++v) t_out[t_lane + (unsigned)((32 * jt + (v & 3) + 8 * (v >> 2)) * N)] = sacc[jt][v];
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float tv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) tv[e] = sacc[jt][8 * s2 + e];
            split8(tv, thi[jt][s2], tlo[jt][s2]);
        }
    }

    // ---------------------------------------------------------------- phase 2
    // A2 item (row c, u = (k-step of the stage, lane half)): keys 32 st + 16 (u >> 1) + 4 (u & 1) + {0..3} and + 8 + {0..3}
    constexpr int ROWS = 32 * DT, ITEMS = (ROWS * 4 + 255) / 256;
    const int vu = tid & 3, vc = tid >> 2;               // row vc + 64 it
    f32x4 rv[2][ITEMS][2];
    const int npass = d / ROWS;
    for (int pass = 0; pass < npass; ++pass) {
        const float* __restrict__ A2p = A2 + (int64_t)(pass * ROWS) * N;
        f32x16 oacc[DT];
#pragma unroll
        for (int ct = 0; ct < DT; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) oacc[ct][v] = 0.f;
        __syncthreads();                                 // every wave is done with the LDS of the previous phase / pass
#pragma unroll
        for (int st = -1; st < JT; ++st) {               // st = -1: prologue (loads of stages 0 and 1, LDS image of stage 0)
            // loads two stages ahead into the register set that stage st + 1's image has just been made from (st >= 0) / the first two
#pragma unroll
            for (int pre = (st < 0 ? 0 : st + 2); pre <= (st < 0 ? 1 : st + 2); ++pre) {
                if (pre < JT) {
#pragma unroll
                    for (int it = 0; it < ITEMS; ++it) {
                        const int c = min(vc + 64 * it, ROWS - 1);
                        const float* __restrict__ src = A2p + (int64_t)c * N + 32 * pre + 16 * (vu >> 1) + 4 * (vu & 1);
                        rv[pre & 1][it][0] = *reinterpret_cast<const f32x4*>(src);
                        rv[pre & 1][it][1] = *reinterpret_cast<const f32x4*>(src + 8);
                    }
                }
            }
            if (st >= 0) {
                const u32x4_t* __restrict__ Vs = lds + (st & 1) * BUF2 + l31;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8_t bhf = __builtin_bit_cast(bf16x8_t, thi[st][s2]), blf = __builtin_bit_cast(bf16x8_t, tlo[st][s2]);
#pragma unroll
                    for (int ct = 0; ct < DT; ++ct) {
                        const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, Vs[(0 * 4 + s2 * 2 + h) * VPL + 32 * ct]);
                        const bf16x8_t al = __builtin_bit_cast(bf16x8_t, Vs[(1 * 4 + s2 * 2 + h) * VPL + 32 * ct]);
                        oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhf, oacc[ct], 0, 0, 0);
                        oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blf, oacc[ct], 0, 0, 0);
                        oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhf, oacc[ct], 0, 0, 0);
                    }
                }
            }
            if (st + 1 < JT) {                           // LDS image of stage st + 1 (its loads were issued a full stage ago)
                u32x4_t* __restrict__ Vw = lds + ((st + 1) & 1) * BUF2;
#pragma unroll
                for (int it = 0; it < ITEMS; ++it) {
                    const int c = vc + 64 * it;
                    if (c < ROWS) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = rv[(st + 1) & 1][it][0][e];
                            v[4 + e] = rv[(st + 1) & 1][it][1][e];
                        }
                        u32x4_t hi, lo;
                        split8(v, hi, lo);
                        Vw[(0 * 4 + vu) * VPL + c] = hi;         // plane = part * 4 + kstep * 2 + h = part * 4 + u
                        Vw[(1 * 4 + vu) * VPL + c] = lo;
                    }
                }
            }
            __syncthreads();
        }
        float* __restrict__ Yp = Yb + (int64_t)(pass * ROWS) * N + 32 * wave_u;
#pragma unroll
        for (int ct = 0; ct < DT; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) Yp[t_lane + (unsigned)((32 * ct + (v & 3) + 8 * (v >> 2)) * N)] = oacc[ct][v];
    }
}

template <bool BWD>
static int launch_attn_core(const attn_core_args& a, int nbh, hipStream_t st) {
    const dim3 grid(2 * nbh);
    const int d = a.d;
    // d = 256 k, backward: eight channel tiles of phase-2 accumulators beside the 128 + 128 score / staging registers spill 101 VGPRs (396 bytes of
    // scratch per lane); two passes of four tiles (VD_ATTN_BWD_DT8=1: the one-pass form) re-stage k once more and spill 4
    constexpr int bwd_dt8 = 0;            // (the one-pass form spilled 101 VGPRs: profiles/HISTORY.md)
    if (d % 256 == 0 && BWD && !bwd_dt8) hipLaunchKernelGGL((attn_core_kernel<4, BWD>), grid, dim3(256), 0, st, a);
    else if (d % 256 == 0) hipLaunchKernelGGL((attn_core_kernel<8, BWD>), grid, dim3(256), 0, st, a);
    else if (d == 128) hipLaunchKernelGGL((attn_core_kernel<4, BWD>), grid, dim3(256), 0, st, a);
    else if (d == 64) hipLaunchKernelGGL((attn_core_kernel<2, BWD>), grid, dim3(256), 0, st, a);
    else if (d == 32) hipLaunchKernelGGL((attn_core_kernel<1, BWD>), grid, dim3(256), 0, st, a);
    else return VD_EINVAL;
    return 0;
}

}  // namespace

extern "C" int vd_softmax_col_fwd(float* S, int nb, int N, void* stream) {
    VD_REQUIRE(S && nb > 0 && N > 0, "vd_softmax_col_fwd: bad args");
    const dim3 grid(nb * ((N + 63) / 64));
    if (N <= 64) hipLaunchKernelGGL((softmax_col_fwd_reg_kernel<16>), grid, dim3(256), 0, (hipStream_t)stream, S, N);
    else if (N <= 256) hipLaunchKernelGGL((softmax_col_fwd_reg_kernel<64>), grid, dim3(256), 0, (hipStream_t)stream, S, N);
    else hipLaunchKernelGGL(softmax_col_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, S, N);
    VD_LAUNCH_CHECK("vd_softmax_col_fwd");
    return 0;
}

extern "C" int vd_softmax_col_bwd(const float* P, float* dP, int nb, int N, float scale, void* stream) {
    VD_REQUIRE(P && dP && nb > 0 && N > 0, "vd_softmax_col_bwd: bad args");
    const dim3 grid(nb * ((N + 63) / 64));
    if (N <= 64) hipLaunchKernelGGL((softmax_col_bwd_reg_kernel<16>), grid, dim3(256), 0, (hipStream_t)stream, P, dP, N, scale);
    else if (N <= 128) hipLaunchKernelGGL((softmax_col_bwd_reg_kernel<32>), grid, dim3(256), 0, (hipStream_t)stream, P, dP, N, scale);
    else if (N <= 256) hipLaunchKernelGGL((softmax_col_bwd_reg_kernel<64>), grid, dim3(256), 0, (hipStream_t)stream, P, dP, N, scale);
    else hipLaunchKernelGGL(softmax_col_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, P, dP, N, scale);
    VD_LAUNCH_CHECK("vd_softmax_col_bwd");
    return 0;
}

extern "C" int vd_attn_small_fwd(const float* qkv, float* out, float* P, int B, int C, int N, float scale, int64_t qkv_bstride,
                                 int64_t out_bstride, void* stream) {
    VD_REQUIRE(qkv && out && B > 0 && C > 0 && N > 0 && N <= 64, "vd_attn_small_fwd: needs N <= 64 (N=%d)", N);
    constexpr int nostage = 0;
    if (!nostage && 3 * C * N <= 12288)
        hipLaunchKernelGGL(attn_small_fwd_kernel<true>, dim3(B), dim3(256), 0, (hipStream_t)stream, qkv, out, P, C, N, scale, qkv_bstride, out_bstride);
    else
        hipLaunchKernelGGL(attn_small_fwd_kernel<false>, dim3(B), dim3(256), 0, (hipStream_t)stream, qkv, out, P, C, N, scale, qkv_bstride, out_bstride);
    VD_LAUNCH_CHECK("vd_attn_small_fwd");
    return 0;
}

extern "C" int vd_attn_small_bwd(const float* qkv, const float* P, const float* dout, float* dqkv, int B, int C, int N,
                                 float scale, int64_t qkv_bstride, int64_t dout_bstride, int64_t dqkv_bstride, void* stream) {
    VD_REQUIRE(qkv && P && dout && dqkv && B > 0 && C > 0 && N > 0 && N <= 64, "vd_attn_small_bwd: needs N <= 64 (N=%d)", N);
    constexpr int nostage = 0;
    if (!nostage && 4 * C * N <= 16384)
        hipLaunchKernelGGL(attn_small_bwd_kernel<true>, dim3(B), dim3(256), 0, (hipStream_t)stream, qkv, P, dout, dqkv, C, N, scale, qkv_bstride,
                           dout_bstride, dqkv_bstride);
    else
        hipLaunchKernelGGL(attn_small_bwd_kernel<false>, dim3(B), dim3(256), 0, (hipStream_t)stream, qkv, P, dout, dqkv, C, N, scale, qkv_bstride,
                           dout_bstride, dqkv_bstride);
    VD_LAUNCH_CHECK("vd_attn_small_bwd");
    return 0;
}

static bool attn_core_shape_ok(int B, int heads, int d, int N) {
    return B > 0 && heads > 0 && N == 256 && (d == 32 || d == 64 || d == 128 || (d > 0 && d % 256 == 0));
}

extern "C" int vd_attn_core_fwd(const float* qkv, float* out, float* P, int B, int heads, int head_dim, int N, float scale,
                                void* stream) {
    VD_REQUIRE(qkv && out, "vd_attn_core_fwd: null pointer");
    VD_REQUIRE(attn_core_shape_ok(B, heads, head_dim, N), "vd_attn_core_fwd: needs N == 256 tokens and head_dim in {32, 64, 128, 256 k} "
               "(B=%d heads=%d head_dim=%d N=%d)", B, heads, head_dim, N);
    VD_REQUIRE(((((uintptr_t)qkv) | ((uintptr_t)out)) & 15) == 0, "vd_attn_core_fwd: pointers must be 16-byte aligned");
    const int64_t C = (int64_t)heads * head_dim;
    attn_core_args a;
    a.a1 = qkv + C * N;          // k
    a.b1 = qkv;                  // q
    a.a2 = qkv + 2 * C * N;      // v
    a.y = out;
    a.t_out = P;
    a.p_in = nullptr;
    a.o_in = nullptr;
    a.o_bs = 0;
    a.a1_bs = a.b1_bs = a.a2_bs = 3 * C * N;
    a.y_bs = C * N;
    a.heads = heads;
    a.d = head_dim;
    a.scale = scale;
    VD_REQUIRE(launch_attn_core<false>(a, B * heads, (hipStream_t)stream) == 0, "vd_attn_core_fwd: unsupported head_dim %d", head_dim);
    VD_LAUNCH_CHECK("vd_attn_core_fwd");
    return 0;
}

extern "C" int vd_attn_core_bwd(const float* qkv, const float* P, const float* out, const float* dout, float* dS, float* dqkv, int B,
                                int heads, int head_dim, int N, float scale, void* stream) {
    VD_REQUIRE(qkv && P && out && dout && dS && dqkv, "vd_attn_core_bwd: null pointer");
    VD_REQUIRE(attn_core_shape_ok(B, heads, head_dim, N), "vd_attn_core_bwd: needs N == 256 tokens and head_dim in {32, 64, 128, 256 k} "
               "(B=%d heads=%d head_dim=%d N=%d)", B, heads, head_dim, N);
    VD_REQUIRE(((((uintptr_t)qkv) | ((uintptr_t)dout) | ((uintptr_t)dqkv)) & 15) == 0, "vd_attn_core_bwd: pointers must be 16-byte aligned");
    const int64_t C = (int64_t)heads * head_dim;
    attn_core_args a;
    a.a1 = qkv + 2 * C * N;      // v:  dP[j][i] = sum_c v[c][j] do[c][i]
    a.b1 = dout;
    a.a2 = qkv + C * N;          // k:  dq[c][i] = sum_j k[c][j] dS[j][i]
    a.y = dqkv;                  // the q slice of dqkv
    a.t_out = dS;
    a.p_in = P;
    a.o_in = out;
    a.o_bs = C * N;
    a.a1_bs = a.a2_bs = 3 * C * N;
    a.b1_bs = C * N;
    a.y_bs = 3 * C * N;
    a.heads = heads;
    a.d = head_dim;
    a.scale = scale;
    VD_REQUIRE(launch_attn_core<true>(a, B * heads, (hipStream_t)stream) == 0, "vd_attn_core_bwd: unsupported head_dim %d", head_dim);
    VD_LAUNCH_CHECK("vd_attn_core_bwd");
    return 0;
}
