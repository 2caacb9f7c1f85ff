// K4: attention softmax (key-major score matrix) and the tiny-token-count attention of the UNet mid block.
//
// The QK^T and PV contractions of the 16x16-token attention blocks run on the f32 MFMA through vd_gemm (vd_gemm.hip)
// in NCHW with no transposes:  St[j][i] = scale * sum_c k[c][j] q[c][i]  (A = k column-major, B = q plain),
// softmax over j per column i (this file),  out[c][i] = sum_j v[c][j] P[j][i].
// Replaces torch.baddbmm/softmax/bmm of diffusers' AttentionBlock (UNet2DModel, reached from loss.py:993).
#include "vd_common.h"

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
__global__ __launch_bounds__(256) void attn_small_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                             float* __restrict__ Pout, int C, int N, float scale,
                                                             int64_t qkv_bs, int64_t out_bs) {
    __shared__ float Ps[64 * 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* __restrict__ q = qkv + (int64_t)b * qkv_bs;
    const float* __restrict__ k = q + (int64_t)C * N;
    const float* __restrict__ v = k + (int64_t)C * N;
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

__global__ __launch_bounds__(256) void attn_small_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                             const float* __restrict__ dout, float* __restrict__ dqkv, int C,
                                                             int N, float scale, int64_t qkv_bs, int64_t dout_bs,
                                                             int64_t dqkv_bs) {
    __shared__ float Ps[64 * 64];
    __shared__ float dS[64 * 64];
    __shared__ float dots[64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* __restrict__ q = qkv + (int64_t)b * qkv_bs;
    const float* __restrict__ k = q + (int64_t)C * N;
    const float* __restrict__ v = k + (int64_t)C * N;
    const float* __restrict__ dob = dout + (int64_t)b * dout_bs;
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
    hipLaunchKernelGGL(attn_small_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, qkv, out, P, C, N, scale, qkv_bstride,
                       out_bstride);
    VD_LAUNCH_CHECK("vd_attn_small_fwd");
    return 0;
}

extern "C" int vd_attn_small_bwd(const float* qkv, const float* P, const float* dout, float* dqkv, int B, int C, int N,
                                 float scale, int64_t qkv_bstride, int64_t dout_bstride, int64_t dqkv_bstride, void* stream) {
    VD_REQUIRE(qkv && P && dout && dqkv && B > 0 && C > 0 && N > 0 && N <= 64, "vd_attn_small_bwd: needs N <= 64 (N=%d)", N);
    hipLaunchKernelGGL(attn_small_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, qkv, P, dout, dqkv, C, N, scale,
                       qkv_bstride, dout_bstride, dqkv_bstride);
    VD_LAUNCH_CHECK("vd_attn_small_bwd");
    return 0;
}
