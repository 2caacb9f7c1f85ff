// Implicit-GEMM convolution / GEMM / weight-gradient kernels on the f32-input MFMA of gfx950.
//
//   v_mfma_f32_32x32x2_f32: exact f32 (bitwise a k-ordered fmaf chain), 64 FLOP/clk/SIMD = 157.3 TFLOP/s chip peak,
//   the same peak as the f32 VALU but with one VGPR per operand per lane (MI355X_MICROARCH.md "Matrix cores").
//
// Tiling: a 256-thread workgroup (4 waves, 2x2) owns a (64*WM) x (64*WN) output tile; each wave owns WM x WN
// accumulators of 32x32.  The K dimension is staged through LDS in steps of 32 as 8 "k-groups" of 4 consecutive
// k values: LDS image  As[kgroup][row][4], Bs[kgroup][col][4]  (rows padded by one 16-B slot).  A lane reads its
// operand fragment with ONE ds_read_b128 per 32x32 tile per 8 k (lanes 0-31 take k-group 2g, lanes 32-63 take 2g+1,
// register t of the float4 feeds MFMA step t), which is conflict-free for both the 16-lane b128 read groups and the
// 8-lane b128 write groups.  Global->register loads of K-step t+1 are issued before the MFMAs of step t.
//
// Reference ops replaced: F.conv2d 3x3/1x1 (ResnetBlock2D, Downsample2D, Upsample2D, conv_in/out), F.linear
// (time embedding, attention projections), torch.bmm (attention) -- diffusers UNet2DModel reached from loss.py:993.
#include "vd_common.h"
#include <string.h>
#include <stdlib.h>

// vd_conv_k32p.hip: the persistent 16x16x32 split-precision 3x3 convolution (its own translation unit)
bool vd_conv3_k32p_eligible(const vd_gemm_desc& d);
int vd_launch_conv3_k32p(const vd_gemm_desc& d, int mode, hipStream_t st);
// vd_presplit.hip: the grouped 3x3 weight gradient with both operands pre-split (LDS-DMA + transposed reads)
int vd_launch_wgrad_ps_group(const void* jobs, int n, int W, int up, int blocks, hipStream_t st);
// vd_conv_sm.hip: the whole-K 16x16x32 split-precision 3x3 convolution of the 8x8 / 4x4 levels (round 6)
bool vd_conv3_sm_eligible(const vd_gemm_desc& d);
int vd_launch_conv3_sm(const vd_gemm_desc& d, hipStream_t st);
// vd_gemm_k32p.hip: the persistent 16x16x32 split-precision 1x1 convolution / plain product
bool vd_gemm1x1_k32p_pick(const vd_gemm_desc& d);
int vd_launch_gemm1x1_k32p(const vd_gemm_desc& d, hipStream_t st);

namespace {

constexpr int NT = 256;  // threads per workgroup
constexpr int BK = 32;   // k per LDS stage
constexpr int KG = BK / 4;

struct Pix {  // decomposition of one GEMM column n -> (batch item, output pixel)
    int b, oy, ox, p;
    bool valid;
};

__device__ __forceinline__ void kdecomp9(int k, int& c, int& r, int& s) {
    c = k / 9;
    int rs = k - c * 9;
    r = (rs * 11) >> 5;  // rs/3 for rs in [0,9)
    s = rs - r * 3;
}

// One element of the implicit im2col operand for column `px` and reduction index (c, r, s).
template <int BMODE>
__device__ __forceinline__ float conv_gather(const float* __restrict__ src, const vd_gemm_desc& d, const Pix& px,
                                             int64_t boff, int c, int r, int s, bool& ok) {
    int iy, ix;
    if (BMODE == VD_B_CONV3) {
        iy = px.oy + r - 1;
        ix = px.ox + s - 1;
        ok = (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
    } else if (BMODE == VD_B_CONV3_T) {
        iy = px.oy + 1 - r;
        ix = px.ox + 1 - s;
        ok = (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
    } else if (BMODE == VD_B_CONV3_S2) {
        iy = 2 * px.oy + r - d.pad;
        ix = 2 * px.ox + s - d.pad;
        ok = (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
    } else if (BMODE == VD_B_CONV3_UP) {
        int uy = px.oy + r - 1, ux = px.ox + s - 1;
        ok = (unsigned)uy < (unsigned)(2 * d.H) && (unsigned)ux < (unsigned)(2 * d.W);
        iy = uy >> 1;
        ix = ux >> 1;
    } else if (BMODE == VD_B_CONVG) {
        iy = px.oy * d.conv_stride + r - d.pad_h;
        ix = px.ox * d.conv_stride + s - d.pad_w;
        ok = (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
    } else {  // VD_B_CONV3_DIL
        int ty = px.oy - r, tx = px.ox - s;
        ok = ty >= 0 && tx >= 0 && !((ty | tx) & 1);
        iy = ty >> 1;
        ix = tx >> 1;
        ok = ok && iy < d.H && ix < d.W;
    }
    ok = ok && px.valid && c < d.C;
    // Branch-free: ALWAYS load (from offset 0 of the batch item when masked) and select afterwards.  A predicated
    // `ok ? src[..] : 0` makes hipcc branch around every load and wait vmcnt(0) per element (cdna guide §5, trap 4c).
    const int off = ok ? (c * d.H * d.W + iy * d.W + ix) : 0;
    return src[boff + off];            // caller zeroes masked elements later (deferred select)
}

// Batch offset with the optional second batch level (heads of multi-head attention are channel slices of one tensor):
// item i = outer * nb2 + inner  ->  outer * stride + inner * stride2.
__device__ __forceinline__ int64_t batch_off(const vd_gemm_desc& d, int i, int64_t stride, int64_t stride2) {
    if (d.nb2 <= 1) return (int64_t)i * stride;
    const int o = i / d.nb2;
    return (int64_t)o * stride + (int64_t)(i - o * d.nb2) * stride2;
}

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f32x4 sel4(bool ok, f32x4 v) {
    f32x4 z = zero4();
    return f32x4{ok ? v[0] : z[0], ok ? v[1] : z[1], ok ? v[2] : z[2], ok ? v[3] : z[3]};
}

// ---- MFMA over one LDS stage -------------------------------------------------------------------------------------
template <int WM, int WN, int LDA_, int LDB_>
__device__ __forceinline__ void mma_stage(const f32x4* __restrict__ As, const f32x4* __restrict__ Bs, int arow, int bcol,
                                          int h, f32x16 (&acc)[WM][WN]) {
#pragma unroll
    for (int g = 0; g < KG / 2; ++g) {
        f32x4 a[WM], b[WN];
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) a[mi] = As[(2 * g + h) * LDA_ + arow + mi * 32];
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) b[ni] = Bs[(2 * g + h) * LDB_ + bcol + ni * 32];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][t], b[ni][t], acc[mi][ni], 0, 0, 0);
    }
}

// ---- shared epilogue: D = alpha*acc + bias + rowadd + residual (+ D) -------------------------------------------------
// Optional terms sit behind WAVE-UNIFORM branches (kernel arguments), each covering a block of 8 unconditional loads
// from clamped addresses, so the loads of a block are in flight together (no per-element branch + wait).
template <int WM, int WN>
__device__ __forceinline__ void gemm_epilogue(const vd_gemm_desc& d, f32x16 (&acc)[WM][WN], int m0, int n0, int wm, int wn,
                                              int lane, int h) {
    const bool has_bias_m = d.bias != nullptr && !d.bias_on_n;
    const bool has_bias_n = d.bias != nullptr && d.bias_on_n;
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
        const int n = n0 + wn * 32 * WN + ni * 32 + (lane & 31);
        const bool nok = n < d.N;
        const int nc = nok ? n : d.N - 1;
        const int b = nc / d.NP, p = nc - b * d.NP;
        const int64_t dbase = d.d_trans ? (int64_t)nc * d.ldd : (batch_off(d, b, d.d_bstride, d.d_b2stride) + p);
        const int64_t dstr = d.d_trans ? 1 : d.ldd;
        float bn = 0.f;
        if (has_bias_n) bn = d.bias[nc];
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int mbase = m0 + wm * 32 * WM + mi * 32 + 4 * h + 16 * half;   // rows mbase + (u&3) + 8*(u>>2)
                float val[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) val[u] = d.alpha * acc[mi][ni][8 * half + u] + bn;
                if (has_bias_m) {
                    float t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = d.bias[min(mbase + (u & 3) + 8 * (u >> 2), d.M - 1)];
#pragma unroll
                    for (int u = 0; u < 8; ++u) val[u] += t[u];
                }
                if (d.rowadd != nullptr) {
                    float t[8];
                    const float* ra_ = d.rowadd + (int64_t)b * d.rowadd_bstride;
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = ra_[min(mbase + (u & 3) + 8 * (u >> 2), d.M - 1)];
#pragma unroll
                    for (int u = 0; u < 8; ++u) val[u] += t[u];
                }
                if (d.residual != nullptr) {
                    float t[8];
                    const float* rs_ = d.residual + (int64_t)b * d.res_bstride + p;
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = rs_[(int64_t)min(mbase + (u & 3) + 8 * (u >> 2), d.M - 1) * d.ldd];
#pragma unroll
                    for (int u = 0; u < 8; ++u) val[u] += t[u];
                }
                if (d.accumulate) {
                    float t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = d.D[dbase + (int64_t)min(mbase + (u & 3) + 8 * (u >> 2), d.M - 1) * dstr];
#pragma unroll
                    for (int u = 0; u < 8; ++u) val[u] += t[u];
                }
                if (d.act == 1) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) val[u] = fmaxf(val[u], 0.f);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int m = mbase + (u & 3) + 8 * (u >> 2);
                    if (nok && m < d.M) d.D[dbase + (int64_t)m * dstr] = val[u];
                }
                if (WM * WN >= 8) __builtin_amdgcn_sched_barrier(0);   // 128 accumulator registers: one 8-row block's loads in flight at a time
            }
        }
    }
}

// ---- generic GEMM / conv forward / dgrad -------------------------------------------------------------------------
template <int WM, int WN, int AMODE, int BMODE>
__global__ __launch_bounds__(NT, (WM * WN >= 4 ? 2 : 3)) void gemm_kernel(const vd_gemm_desc d) {      // (the 128 x 128 tile spilled 17-66 VGPRs at three workgroups per CU)
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int LDA_ = BM + 1, LDB_ = BN + 1;
    constexpr int A_F4 = BM * KG / NT;  // float4 slots per thread for A
    constexpr int B_F4 = BN * KG / NT;
    __shared__ f32x4 As[KG * LDA_];
    __shared__ f32x4 Bs[KG * LDB_];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5;
    const int tiles_m = (d.M + BM - 1) / BM;
    // XCD-aware remap: consecutive logical tiles (which share the activation tile) land on one XCD's L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    }
    const int tm = bid % tiles_m, tn = bid / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    const float* __restrict__ Ap = d.A;
    const float* __restrict__ Bp = d.B;
    if (d.a_bstride != 0) Ap += batch_off(d, n0 / d.NP, d.a_bstride, d.a_b2stride);

    const bool a_vec = (AMODE == VD_A_ROW) && ((d.lda & 3) == 0) && ((d.K & 3) == 0) &&
                       ((((uintptr_t)Ap) & 15) == 0);
    const bool b_vec = (BMODE == VD_B_KCONTIG) && ((d.ldb & 3) == 0) && ((d.K & 3) == 0) &&
                       ((d.b_bstride & 3) == 0) && ((((uintptr_t)Bp) & 15) == 0);

    // Column owned by this thread for the column-contiguous loaders (PLAIN / CONV*): fixed for the whole tile.
    Pix px;
    int64_t boff = 0;
    if (BMODE != VD_B_KCONTIG) {
        const int n = n0 + (tid % BN);
        px.valid = n < d.N;
        const int nn = px.valid ? n : 0;
        px.b = nn / d.NP;
        px.p = nn - px.b * d.NP;
        if (BMODE >= VD_B_CONV3) {
            px.oy = px.p / d.OW;
            px.ox = px.p - px.oy * d.OW;
        } else {
            px.oy = px.ox = 0;
        }
        boff = batch_off(d, px.b, d.b_bstride, d.b_b2stride);
    }

    // Global->register staging.  Loads are UNCONDITIONAL (masked lanes read a clamped, valid address) and their
    // validity bits are kept in amask/bmask; the zeroing select is applied in store_ab(), i.e. after the MFMAs of the
    // current K-step, so nothing consumes a load result (and forces an s_waitcnt) while the loads are being issued.
    f32x4 ra[A_F4], rb[B_F4];
    unsigned amask = 0, bmask = 0;      // bit (4*i + j) = element j of slot i is valid

    auto load_a = [&](int k0) {
        amask = 0;
        if (AMODE == VD_A_ROW) {
            if (a_vec) {                                       // wave-uniform
#pragma unroll
                for (int i = 0; i < A_F4; ++i) {
                    const int idx = tid + i * NT;
                    const int m = m0 + (idx >> 3), k = k0 + (idx & 7) * 4;
                    const bool ok = m < d.M && k < d.K;
                    ra[i] = *reinterpret_cast<const f32x4*>(Ap + (int64_t)(ok ? m : 0) * d.lda + (ok ? k : 0));
                    amask |= (ok ? 0xFu : 0u) << (4 * i);
                }
            } else {
#pragma unroll
                for (int i = 0; i < A_F4; ++i) {
                    const int idx = tid + i * NT;
                    const int m = m0 + (idx >> 3), k = k0 + (idx & 7) * 4;
                    const bool mok = m < d.M;
                    const float* p = Ap + (int64_t)(mok ? m : 0) * d.lda;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool ok = mok && (k + j < d.K);
                        ra[i][j] = p[ok ? k + j : 0];
                        amask |= (ok ? 1u : 0u) << (4 * i + j);
                    }
                }
            }
        } else {  // VD_A_COL: lanes run along m
            const int m = m0 + (tid % BM);
            const bool mok = m < d.M;
            const float* p = Ap + (mok ? m : 0);
            constexpr int KQ_STEP = NT / BM;
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const int k = k0 + ((tid / BM) + i * KQ_STEP) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = mok && (k + j < d.K);
                    ra[i][j] = p[(int64_t)(ok ? k + j : 0) * d.lda];
                    amask |= (ok ? 1u : 0u) << (4 * i + j);
                }
            }
        }
    };

    auto load_b = [&](int k0) {
        bmask = 0;
        if (BMODE == VD_B_KCONTIG) {
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int idx = tid + i * NT;
                const int n = n0 + (idx >> 3), k = k0 + (idx & 7) * 4;
                const bool nok = n < d.N;
                const int nn = nok ? n : 0;
                const int b = nn / d.NP, p = nn - b * d.NP;
                const float* q = Bp + batch_off(d, b, d.b_bstride, d.b_b2stride) + (int64_t)p * d.ldb;
                if (b_vec) {
                    const bool ok = nok && k < d.K;
                    rb[i] = *reinterpret_cast<const f32x4*>(q + (ok ? k : 0));
                    bmask |= (ok ? 0xFu : 0u) << (4 * i);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool ok = nok && (k + j < d.K);
                        rb[i][j] = q[ok ? k + j : 0];
                        bmask |= (ok ? 1u : 0u) << (4 * i + j);
                    }
                }
            }
        } else {
            constexpr int KQ_STEP = NT / BN;
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int k = k0 + ((tid / BN) + i * KQ_STEP) * 4;
                if (BMODE == VD_B_PLAIN) {
                    const float* q = Bp + boff + px.p;           // px.p = 0, boff = 0 when the column is out of range
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool ok = px.valid && (k + j < d.K);
                        rb[i][j] = q[(int64_t)(ok ? k + j : 0) * d.ldb];
                        bmask |= (ok ? 1u : 0u) << (4 * i + j);
                    }
                } else {
                    int c, r, s;
                    const int KW_ = (BMODE == VD_B_CONVG) ? d.kw : 3, KH_ = (BMODE == VD_B_CONVG) ? d.kh : 3;
                    if (BMODE == VD_B_CONVG) {
                        const int T_ = KH_ * KW_;
                        c = k / T_;
                        const int rs = k - c * T_;
                        r = rs / KW_;
                        s = rs - r * KW_;
                    } else {
                        kdecomp9(k, c, r, s);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bool ok;
                        rb[i][j] = conv_gather<BMODE>(Bp, d, px, boff, c, r, s, ok);
                        bmask |= (ok ? 1u : 0u) << (4 * i + j);
                        if (++s == KW_) {
                            s = 0;
                            if (++r == KH_) {
                                r = 0;
                                ++c;
                            }
                        }
                    }
                }
            }
        }
    };

    auto masked = [](f32x4 v, unsigned bits) {
        return f32x4{(bits & 1u) ? v[0] : 0.f, (bits & 2u) ? v[1] : 0.f, (bits & 4u) ? v[2] : 0.f, (bits & 8u) ? v[3] : 0.f};
    };
    auto store_ab = [&]() {
        if (AMODE == VD_A_ROW) {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const int idx = tid + i * NT;
                As[(idx & 7) * LDA_ + (idx >> 3)] = masked(ra[i], amask >> (4 * i));
            }
        } else {
            constexpr int KQ_STEP = NT / BM;
#pragma unroll
            for (int i = 0; i < A_F4; ++i) As[((tid / BM) + i * KQ_STEP) * LDA_ + (tid % BM)] = masked(ra[i], amask >> (4 * i));
        }
        if (BMODE == VD_B_KCONTIG) {
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int idx = tid + i * NT;
                Bs[(idx & 7) * LDB_ + (idx >> 3)] = masked(rb[i], bmask >> (4 * i));
            }
        } else {
            constexpr int KQ_STEP = NT / BN;
#pragma unroll
            for (int i = 0; i < B_F4; ++i) Bs[((tid / BN) + i * KQ_STEP) * LDB_ + (tid % BN)] = masked(rb[i], bmask >> (4 * i));
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mi][ni][v] = 0.f;

    const int wm = wave >> 1, wn = wave & 1;
    const int arow = wm * 32 * WM + (lane & 31);
    const int bcol = wn * 32 * WN + (lane & 31);

    const int ktiles = (d.K + BK - 1) / BK;
    load_a(0);
    load_b(0);
    store_ab();
    __syncthreads();
    if (VD_DBG(d) == 0) {
        for (int kt = 0; kt < ktiles; ++kt) {
            const bool more = kt + 1 < ktiles;
            if (more) {
                load_a((kt + 1) * BK);
                load_b((kt + 1) * BK);
            }
            mma_stage<WM, WN, LDA_, LDB_>(As, Bs, arow, bcol, h, acc);
            __syncthreads();
            if (more) store_ab();
            __syncthreads();
        }
    } else {  // timing-only ablations (results are invalid)
        for (int kt = 0; kt < ktiles; ++kt) {
            const bool more = kt + 1 < ktiles;
            if (more && !(VD_DBG(d) & 1)) {
                load_a((kt + 1) * BK);
                load_b((kt + 1) * BK);
            }
            if (!(VD_DBG(d) & 4)) mma_stage<WM, WN, LDA_, LDB_>(As, Bs, arow, bcol, h, acc);
            __syncthreads();
            if (more && !(VD_DBG(d) & 8)) store_ab();
            __syncthreads();
        }
        if (VD_DBG(d) & 2) {
            float s = 0.f;
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                    for (int v = 0; v < 16; ++v) s += acc[mi][ni][v];
            if (s == 123456.789f) d.D[0] = s;
            return;
        }
    }

    gemm_epilogue<WM, WN>(d, acc, m0, n0, wm, wn, lane, h);
}

// ---- patch-staged 3x3 convolution (forward / stride-1 dgrad / fused nearest-2x upsample) ---------------------------
// The im2col gather of gemm_kernel costs ~500 VALU instructions and 16 scalar loads per thread per 32-k step (9 taps
// re-load every input element).  Here one workgroup owns 128 output channels x (128/W) full-width output rows of one
// image and, per K-step of CK = 8 input channels (72 k), stages
//     As[72][128]            the weight slab (k-major, read by lanes along m: conflict-free ds_read_b32)
//     Ps[8][PR][PW]          the zero-padded input halo patch, each element loaded ONCE
// MFMA step (channel pair cp, tap (r,s)) reads  A = As[(2cp+h)*9 + 3r+s][m]  and  B = Ps[2cp+h][ty+r][x+s]  with
// lane-constant bases and IMMEDIATE offsets: no address arithmetic in the inner loop.  The two k-slots of the
// 32x32x2 MFMA (lane halves h) are the two channels of a pair.  Patch source offsets are computed once per tile.
constexpr int CK = 8;
constexpr int KSTEP = CK * 9;

// WN = 2: 128 x 128 tile, 3 workgroups / CU.  WN = 4: 128 x 256 tile (8 accumulators per wave, 2 workgroups / CU), used
// where it makes the grid an exact multiple of the 512 resident workgroups (128-channel layers at 32x32: 1024 tiles of
// 128 px on 768 slots leave a 1/3-occupied tail wave).
// MODE 3: CONV3 whose input is silu(GroupNorm(x)): the per-(image, channel) scale / shift pairs (vd_groupnorm_stats) sit in LDS and
// the transform is applied when the halo patch is written to LDS -- the normalised activation never exists in HBM (inference).
// MODE 4: stride-2 convolution (Downsample2D): output pixel (y, x) reads patch[2y + r][2x + s]; the patch of a tile is (2 TR + 1) x (2 W + 1)
// input pixels per image (zero beyond the image: pad (0,1,0,1), or symmetric padding with d.pad = 1).
template <int W, int MODE, int WN>  // MODE 0: CONV3, 1: CONV3_T (flipped taps), 2: CONV3_UP (source is half resolution), 3: GN+SiLU+CONV3, 4: CONV3_S2
__global__ __launch_bounds__(NT, (WN == 4 || W >= 128 || MODE == 4) ? 2 : 3) void conv3_patch_kernel(const vd_gemm_desc d, int ksteps_per_split) {
    constexpr int WM = 2, BM = 128;
    constexpr int CKK = CK;                                     // input channels per K-step
    constexpr int KSTEPK = CKK * 9;
    constexpr int NPIX = 64 * WN;                               // output pixels per tile
    constexpr int IMGS = (W * W >= NPIX) ? 1 : NPIX / (W * W);  // whole images per tile for the 8x8 / 4x4 layers
    constexpr int TR = (IMGS == 1) ? NPIX / W : W;              // output rows per image in the tile
    constexpr int PW = (MODE == 4) ? 2 * W + 1 : W + 2;         // halo patch per image
    constexpr int PR = (MODE == 4) ? 2 * TR + 1 : TR + 2;
    constexpr int PIMG = PR * PW;
    constexpr int PLANE = IMGS * PIMG;                          // patch floats per channel
    constexpr int LDA_ = BM + 1;
    constexpr int A_F4 = BM * KSTEPK / 4 / NT;                   // 9 float4 per thread
    constexpr int P_EL = (CKK * PLANE + NT - 1) / NT;            // patch elements per thread
    __shared__ float As[KSTEPK * LDA_];
    __shared__ float Ps[CKK * PLANE];
    __shared__ float Sab[(MODE == 3) ? 2 * 1024 : 2];          // (scale, shift) per input channel of this tile's image

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int tiles_m = (d.M + BM - 1) / BM;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    }
    const int tm = bid % tiles_m, tn = bid / tiles_m;
    const int m0 = tm * BM, n0 = tn * NPIX;
    int b0, y0, x0 = 0;
    if (IMGS == 1) {
        const int tiles_per_img = d.NP / NPIX;
        b0 = tn / tiles_per_img;
        // first output pixel of this tile.  W is the TILE width: for images wider than the tile (W = 64 / 128 on 64..256 px
        // rows) a tile is TR row segments starting at column x0, still NPIX consecutive pixels in row-major order.
        const int pix0 = (tn - b0 * tiles_per_img) * NPIX;
        y0 = pix0 / d.OW;
        x0 = pix0 - y0 * d.OW;
    } else {
        b0 = tn * IMGS;
        y0 = 0;
    }
    const float* __restrict__ Ap = d.A;
    const float* __restrict__ Xb = d.B + (int64_t)b0 * d.b_bstride;
    const int HWs = d.H * d.W;                                  // source plane (half resolution for MODE 2)
    const int nb_total = d.N / d.NP;

    // patch element -> source offset (channel 0 of the K-step), fixed for the whole tile
    int poff[P_EL];
    unsigned pmask = 0;
#pragma unroll
    for (int i = 0; i < P_EL; ++i) {
        const int e = tid + i * NT;
        const int c = e / PLANE, rem = e - c * PLANE;
        const int img = rem / PIMG, rem2 = rem - img * PIMG;
        const int py = rem2 / PW, px = rem2 - py * PW;
        int iy = y0 + py - 1, ix = x0 + px - 1;                 // coordinates in the (virtual, MODE 2: upsampled) input
        if (MODE == 4) {
            iy = 2 * y0 + py - d.pad;
            ix = 2 * x0 + px - d.pad;
        }
        bool ok = e < CKK * PLANE && (b0 + img) < nb_total;
        if (MODE == 2) {
            ok = ok && (unsigned)iy < (unsigned)(2 * d.H) && (unsigned)ix < (unsigned)(2 * d.W);
            iy >>= 1;
            ix >>= 1;
        } else {
            ok = ok && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
        }
        poff[i] = ok ? (int)(img * d.b_bstride) + c * HWs + iy * d.W + ix : 0;
        pmask |= (ok ? 1u : 0u) << i;
    }

    f32x4 ra[A_F4];
    float rp[P_EL];
    auto load_stage = [&](int c0) {                             // c0: first input channel of the K-step
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int idx = tid + i * NT;
            const int m = idx / (KSTEPK / 4), q = idx - m * (KSTEPK / 4);
            const int mm = min(m0 + m, d.M - 1);
            ra[i] = *reinterpret_cast<const f32x4*>(Ap + (int64_t)mm * d.lda + c0 * 9 + 4 * q);
        }
        const float* __restrict__ xs = Xb + (int64_t)c0 * HWs;
#pragma unroll
        for (int i = 0; i < P_EL; ++i) rp[i] = xs[poff[i]];
    };
    auto store_stage = [&](int c0) {                            // c0: first input channel of the stage being stored
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int idx = tid + i * NT;
            const int m = idx / (KSTEPK / 4), q = idx - m * (KSTEPK / 4);
            const bool ok = m0 + m < d.M;
#pragma unroll
            for (int j = 0; j < 4; ++j) As[(4 * q + j) * LDA_ + m] = ok ? ra[i][j] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < P_EL; ++i) {
            const int e = tid + i * NT;
            if (e < CKK * PLANE) {
                float v = rp[i];
                if (MODE == 3) {                                // same expression as gn_fwd_reg_kernel: z = x*ga + be ; z*sigmoid(z)
                    const int c = c0 + e / PLANE;
                    const float z = v * Sab[2 * c] + Sab[2 * c + 1];
                    v = z * sigmoidf_(z);
                }
                Ps[e] = ((pmask >> i) & 1u) ? v : 0.f;          // zero padding applies to the normalised activation
            }
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mi][ni][v] = 0.f;

    const int wm = wave >> 1, wn = wave & 1;
    // lane-constant LDS bases: A rows of this wave; patch position of this lane's pixel in each of its WN column groups
    const float* __restrict__ a_base = As + h * 9 * LDA_ + wm * 64 + (lane & 31);
    const float* __restrict__ p_base[WN];
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
        const int q = (wn * WN + ni) * 32 + (lane & 31);        // pixel within the tile
        const int img = q / (TR * W), r2 = q - img * (TR * W);
        const int ty = r2 / W, x = r2 - ty * W;
        p_base[ni] = Ps + h * PLANE + img * PIMG + (MODE == 4 ? 2 * ty * PW + 2 * x : ty * PW + x);
    }

    const int nsteps = d.C / CKK;
    const int ks_begin = blockIdx.y * ksteps_per_split;
    const int ks_end = min(nsteps, ks_begin + ksteps_per_split);
    load_stage(ks_begin * CKK);
    if (MODE == 3) {
        const float* __restrict__ ssb = d.gn_ss + (int64_t)b0 * 2 * d.C;
        for (int i = tid; i < 2 * d.C; i += NT) Sab[i] = ssb[i];
        __syncthreads();
    }
    store_stage(ks_begin * CKK);
    __syncthreads();
    for (int ks = ks_begin; ks < ks_end; ++ks) {
        const bool more = ks + 1 < ks_end;
        if (more && !(VD_DBG(d) & 1)) load_stage((ks + 1) * CKK);       // debug bits: timing-only ablations
        // software-pipelined operand fetch: the ds_reads of MFMA step u+1 are issued before the MFMAs of step u
        // (hipcc otherwise places each read right in front of its use and waits lgkmcnt(0): LDS latency per 4 MFMAs)
        auto fetch = [&](int u, float (&a)[WM], float (&bb)[WN]) {
            const int cp = u / 9, t = u - 9 * cp;
            const int r = t / 3, sx = t - 3 * r;
            const int pr = (MODE == 1) ? (2 - r) : r, ps = (MODE == 1) ? (2 - sx) : sx;
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) a[mi] = a_base[(2 * cp * 9 + t) * LDA_ + mi * 32];
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) bb[ni] = p_base[ni][2 * cp * PLANE + pr * PW + ps];
        };
        float a0[WM], b0[WN], a1[WM], b1[WN];
        fetch(0, a0, b0);
#pragma unroll
        for (int u = 0; u < (CKK / 2) * 9; u += 2) {
            fetch(u + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mi], b0[ni], acc[mi][ni], 0, 0, 0);
            if (u + 2 < (CKK / 2) * 9) fetch(u + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[mi], b1[ni], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
        if (more && !(VD_DBG(d) & 8)) store_stage((ks + 1) * CKK);
        __syncthreads();
    }
    if (gridDim.y == 1) {
        gemm_epilogue<WM, WN>(d, acc, m0, n0, wm, wn, lane, h);
        return;
    }
    // split-K: raw partial tile into slab z = blockIdx.y of ws[z][M][N]; splitk_epilogue_kernel finishes the job
    float* __restrict__ slab = d.ws + (int64_t)blockIdx.y * d.M * d.N;
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
        const int n = n0 + wn * 32 * WN + ni * 32 + (lane & 31);
        if (n >= d.N) continue;
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + wm * 64 + mi * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                if (m < d.M) slab[(int64_t)m * d.N + n] = acc[mi][ni][v];
            }
    }
}

// D = alpha * sum_z ws[z] (fixed order) + bias + rowadd + residual (+ D): the epilogue of a split-K launch.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const vd_gemm_desc d, int splits) {
    const int64_t total = (int64_t)d.M * d.N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / d.N), n = (int)(i - (int64_t)m * d.N);
        float s = d.ws[i];
        for (int z = 1; z < splits; ++z) s += d.ws[(int64_t)z * total + i];
        const int b = n / d.NP, p = n - b * d.NP;
        float val = d.alpha * s;
        if (d.bias) val += d.bias[d.bias_on_n ? n : m];
        if (d.rowadd) val += d.rowadd[(int64_t)b * d.rowadd_bstride + m];
        if (d.residual) val += d.residual[(int64_t)b * d.res_bstride + (int64_t)m * d.ldd + p];
        const int64_t off = (int64_t)b * d.d_bstride + (int64_t)m * d.ldd + p;
        if (d.accumulate) val += d.D[off];
        d.D[off] = val;
    }
}

// float4 version (N, NP, ldd and the batch strides multiples of 4, 16-byte aligned tensors): the 8x8 / 4x4 layers launch it 52 times per training step
__global__ __launch_bounds__(256) void splitk_epilogue4_kernel(const vd_gemm_desc d, int splits) {
    const int64_t total4 = ((int64_t)d.M * d.N) >> 2;
    const int64_t total = (int64_t)d.M * d.N;
    for (int64_t i4 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i4 < total4; i4 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = i4 << 2;
        const int m = (int)(i / d.N), n = (int)(i - (int64_t)m * d.N);
        f32x4 s = *reinterpret_cast<const f32x4*>(d.ws + i);
        for (int z = 1; z < splits; ++z) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(d.ws + (int64_t)z * total + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += p[j];
        }
        const int b = n / d.NP, p = n - b * d.NP;
        f32x4 val;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            val[j] = d.alpha * s[j];
            if (d.bias) val[j] += d.bias[d.bias_on_n ? n + j : m];
        }
        if (d.rowadd) {
            const float ra = d.rowadd[(int64_t)b * d.rowadd_bstride + m];
#pragma unroll
            for (int j = 0; j < 4; ++j) val[j] += ra;
        }
        if (d.residual) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(d.residual + (int64_t)b * d.res_bstride + (int64_t)m * d.ldd + p);
#pragma unroll
            for (int j = 0; j < 4; ++j) val[j] += r[j];
        }
        float* __restrict__ dp = d.D + (int64_t)b * d.d_bstride + (int64_t)m * d.ldd + p;
        if (d.accumulate) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(dp);
#pragma unroll
            for (int j = 0; j < 4; ++j) val[j] += o[j];
        }
        *reinterpret_cast<f32x4*>(dp) = val;
    }
}

// Launches the split-K epilogue of a vd_gemm problem (same expression order in both versions: identical results).
static void launch_splitk_epilogue(const vd_gemm_desc& d, int splits, hipStream_t st) {
    const int64_t total = (int64_t)d.M * d.N;
    const bool v4 = (d.N & 3) == 0 && (d.NP & 3) == 0 && (d.ldd & 3) == 0 && (d.d_bstride & 3) == 0 && (d.res_bstride & 3) == 0 &&
                    (((uintptr_t)d.D | (uintptr_t)d.residual | (uintptr_t)d.ws) & 15) == 0;
    if (v4) {
        const int64_t t4 = total >> 2;
        const int g2 = (int)((t4 + 255) / 256 < 2048 ? (t4 + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_epilogue4_kernel, dim3(g2), dim3(256), 0, st, d, splits);
    } else {
        const int g2 = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(g2), dim3(256), 0, st, d, splits);
    }
}

#include "vd_conv_bx3.inc"

// Eligibility of the patch-staged kernel for a vd_gemm problem.
static bool patch_eligible(const vd_gemm_desc& d) {
    if (d.a_mode != VD_A_ROW || d.a_bstride != 0 || d.act) return false;
    if (d.gn_ss && (d.b_mode != VD_B_CONV3 || (d.OW != 16 && d.OW != 32) || d.C > 1024)) return false;
    if (d.b_mode == VD_B_CONV3_S2) {            // stride 2: 32 -> 16, 16 -> 8, 8 -> 4 (full-width tiles)
        if ((d.OW != 4 && d.OW != 8 && d.OW != 16) || d.OH != d.OW || d.H != 2 * d.OH || d.W != 2 * d.OW || d.gn_ss) return false;
    } else if (d.b_mode != VD_B_CONV3 && d.b_mode != VD_B_CONV3_T && d.b_mode != VD_B_CONV3_UP) return false;
    if (d.OW != 4 && d.OW != 8 && d.OW != 16 && d.OW != 32 && d.OW != 64 && d.OW % 128 != 0) return false;
    if (d.OH != d.OW && d.OW < 16) return false;
    if (d.C % CK != 0 || d.OH * d.OW != d.NP || d.d_trans) return false;
    if (d.NP >= 128 ? (d.NP % 128 != 0) : (128 % d.NP != 0)) return false;
    if (d.K != d.C * 9 || (d.lda & 3) != 0 || (((uintptr_t)d.A) & 15) != 0) return false;
    if ((VD_DBG(d) & ~(16 | 1 | 8)) != 0 || (VD_DBG(d) != 0 && !(VD_DBG(d) & 16)) || d.tile != 0) return false;   // 16: ablations on this kernel
    if (d.b_mode == VD_B_CONV3_UP && d.OW == 4) return false;
    return d.M >= 64;
}

// Split the channel loop over workgroups when the tile grid alone cannot fill the chip (8x8 / 4x4 layers).
static void patch_plan(const vd_gemm_desc& d, int& splits, int& ks_per) {
    const int base = vd_cdiv(d.M, 128) * vd_cdiv(d.N, 128);
    const int nsteps = d.C / CK;
    splits = 1;
    if (base < 256) {
        splits = vd_cdiv(512, base);
        const int max_splits = nsteps / 4 > 0 ? nsteps / 4 : 1;      // >= 4 K-steps per split
        if (splits > max_splits) splits = max_splits;
    }
    ks_per = vd_cdiv(nsteps, splits);
    splits = vd_cdiv(nsteps, ks_per);
}

// 128 x 256 tiles where that makes the grid a multiple of the 512 co-resident workgroups (2 / CU).
static bool patch_wide(const vd_gemm_desc& d) {
    if (d.OW != 32 || d.NP % 256 != 0) return false;
    const int64_t wgs = (int64_t)vd_cdiv(d.M, 128) * (d.N / 256);
    return wgs >= 512 && wgs % 512 == 0;
}

static int launch_patch(const vd_gemm_desc& d, hipStream_t st) {
    int splits, ks_per;
    patch_plan(d, splits, ks_per);
    const int mode_ = d.b_mode == VD_B_CONV3 ? (d.gn_ss ? 3 : 0) : (d.b_mode == VD_B_CONV3_T ? 1 : (d.b_mode == VD_B_CONV3_S2 ? 4 : 2));
    if (splits == 1 && patch_wide(d)) {
        dim3 grid(vd_cdiv(d.M, 128) * (d.N / 256), 1);
        if (mode_ == 0) hipLaunchKernelGGL((conv3_patch_kernel<32, 0, 4>), grid, dim3(NT), 0, st, d, ks_per);
        else if (mode_ == 3) hipLaunchKernelGGL((conv3_patch_kernel<32, 3, 4>), grid, dim3(NT), 0, st, d, ks_per);
        else if (mode_ == 1) hipLaunchKernelGGL((conv3_patch_kernel<32, 1, 4>), grid, dim3(NT), 0, st, d, ks_per);
        else hipLaunchKernelGGL((conv3_patch_kernel<32, 2, 4>), grid, dim3(NT), 0, st, d, ks_per);
        return 0;
    }
    if (splits > 1 && d.ws == nullptr) {
        vd_set_error("vd_gemm: split-K workspace required (%d splits): query vd_gemm_ws_floats()", splits);
        return VD_EINVAL;
    }
    dim3 grid(vd_cdiv(d.M, 128) * vd_cdiv(d.N, 128), splits);
    const int mode = mode_;
    bool done = false;
#define VD_PATCH_CASE(WW, MD)                                                                            \
    if (!done && d.OW == WW && mode == MD) {                                                             \
        hipLaunchKernelGGL((conv3_patch_kernel<WW, MD, 2>), grid, dim3(NT), 0, st, d, ks_per);              \
        done = true;                                                                                     \
    }
#define VD_PATCH_WIDE(WW, MD)                                                                            \
    if (!done && (WW == 64 ? d.OW == 64 : (d.OW >= 128 && d.OW % 128 == 0)) && mode == MD) {             \
        hipLaunchKernelGGL((conv3_patch_kernel<WW, MD, 2>), grid, dim3(NT), 0, st, d, ks_per);              \
        done = true;                                                                                     \
    }
    VD_PATCH_WIDE(64, 0) VD_PATCH_WIDE(64, 1) VD_PATCH_WIDE(64, 2)
    VD_PATCH_WIDE(128, 0) VD_PATCH_WIDE(128, 1) VD_PATCH_WIDE(128, 2)
#undef VD_PATCH_WIDE
    VD_PATCH_CASE(32, 0) VD_PATCH_CASE(32, 1) VD_PATCH_CASE(32, 2) VD_PATCH_CASE(32, 3)
    VD_PATCH_CASE(16, 0) VD_PATCH_CASE(16, 1) VD_PATCH_CASE(16, 2) VD_PATCH_CASE(16, 3)
    VD_PATCH_CASE(8, 0) VD_PATCH_CASE(8, 1) VD_PATCH_CASE(8, 2)
    VD_PATCH_CASE(4, 0) VD_PATCH_CASE(4, 1)
    VD_PATCH_CASE(16, 4) VD_PATCH_CASE(8, 4) VD_PATCH_CASE(4, 4)
#undef VD_PATCH_CASE
    if (!done) return VD_EINVAL;
    if (splits > 1) launch_splitk_epilogue(d, splits, st);
    return 0;
}

// ---- direct 3x3 convolution for very few output channels (conv_out: 128 -> 3) ------------------------------------------
// An MFMA tile has >= 32 rows: with M = 3 output channels 90 % of the matrix work is padding (measured: 328 us for 0.9
// GFLOP).  Here a workgroup owns TH x TW = 256 output pixels of one image, one pixel per thread; per stage of 8 input
// channels the zero-padded halo patch goes through LDS once, weights are wave-uniform (scalar loads), every tap is
// one ds_read_b32 + MM FMAs.  HBM-bound: reads the input once (67 MB at B=128: ~20 us).
template <int TW, int MM>
__global__ __launch_bounds__(256) void conv3_smallm_kernel(const vd_gemm_desc d) {
    constexpr int TH = 256 / TW, PW = TW + 2, PH = TH + 2, CK = 8;
    constexpr int PLn = PH * PW;
    constexpr int P_EL = (CK * PLn + 255) / 256;
    __shared__ float Ps[CK * PLn];
    __shared__ f32x4 Ws[CK * 9];                     // weights of the stage: [channel][tap] -> the (<= 4) output channels
    const int tid = threadIdx.x;
    const int tiles_x = d.W / TW, tiles_y = d.H / TH;
    int bid = blockIdx.x;
    const int txi = bid % tiles_x;
    bid /= tiles_x;
    const int tyi = bid % tiles_y;
    const int b = bid / tiles_y;
    const int y0 = tyi * TH, x0 = txi * TW;
    const int ty = tid / TW, tx = tid - ty * TW;
    const int HWs = d.H * d.W;
    const float* __restrict__ xb = d.B + (int64_t)b * d.b_bstride;
    const float* __restrict__ Wp = d.A;

    int poff[P_EL];                                  // source offset inside one channel plane, -1: zero padding / no element
#pragma unroll
    for (int i = 0; i < P_EL; ++i) {
        const int e = tid + i * 256;
        const int rem = e % PLn;
        const int py = rem / PW, px = rem - py * PW;
        const int iy = y0 + py - 1, ix = x0 + px - 1;
        const bool ok = e < CK * PLn && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
        poff[i] = ok ? iy * d.W + ix : -1;
    }
    float acc[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[m] = 0.f;

    for (int c0 = 0; c0 < d.C; c0 += CK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < P_EL; ++i) {
            const int e = tid + i * 256;
            if (e < CK * PLn) {
                const int c = c0 + e / PLn;
                const bool ok = poff[i] >= 0 && c < d.C;
                const float v = xb[ok ? (int64_t)c * HWs + poff[i] : 0];
                Ps[e] = ok ? v : 0.f;
            }
        }
        if (tid < CK * 9) {
            const int c = c0 + tid / 9, t = tid - (tid / 9) * 9;
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < MM; ++m)
                if (m < d.M && c < d.C) w[m] = Wp[(int64_t)m * d.lda + (int64_t)c * 9 + t];
            Ws[tid] = w;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CK; ++c) {               // channels past C carry zero weights and a zero patch
            const float* __restrict__ pc = Ps + c * PLn + ty * PW + tx;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float v = pc[r * PW + q];
                    const f32x4 w = Ws[c * 9 + r * 3 + q];   // same address in every lane: LDS broadcast
#pragma unroll
                    for (int m = 0; m < MM; ++m) acc[m] = fmaf(w[m], v, acc[m]);
                }
        }
    }
    const int p = (y0 + ty) * d.W + x0 + tx;
#pragma unroll
    for (int m = 0; m < MM; ++m)
        if (m < d.M) {
            float v = acc[m] * d.alpha;
            if (d.bias) v += d.bias[m];
            d.D[(int64_t)b * d.d_bstride + (int64_t)m * d.ldd + p] = v;
        }
}

// The same convolution without LDS: a lane owns FOUR consecutive output pixels of a row for all (<= 4) output channels, the four waves of a
// workgroup split the input channels and their partial sums meet in LDS at the end.  Per channel a lane loads three rows of (1 + 4 + 1)
// pixels (the float4 is aligned, straight from global memory; the two edge pixels come from the neighbouring lanes' registers) and the 27
// weights arrive as scalar loads (wave-uniform address): 9 loads feed 36 * M FMAs, against 18 LDS reads per 9 * M FMAs in the kernel
// above, which is LDS-issue bound (85 us at B = 128 for a 17 us read of the input).  W % 4 == 0.
template <int MM, int NW>
__global__ __launch_bounds__(64 * NW) void conv3_fewout_kernel(const vd_gemm_desc d, int64_t quads) {
    __shared__ float red[NW - 1][MM][4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = d.W, H = d.H, W4 = W >> 2, HWs = H * W;
    const int64_t gq = (int64_t)blockIdx.x * 64 + lane;
    const bool live = gq < quads;
    const int64_t g = live ? gq : 0;
    const int x4 = (int)(g % W4);
    const int64_t rest = g / W4;
    const int y = (int)(rest % H), b = (int)(rest / H);
    const float* __restrict__ xb = d.B + (int64_t)b * d.b_bstride + y * W + 4 * x4;
    const bool rowok[3] = {y > 0, true, y < H - 1};
    const bool lf = x4 > 0, rt = x4 < W4 - 1;
    const bool xch = W4 <= 16 && (64 % W4) == 0;          // uniform (rows of up to 64 pixels; measured: 31.2 against 34.8 us at 32 x 32, B = 128 -- and 147 against 136 us
                                                          // with the 64-quad rows of 256 x 256 images, which keep the loads)
    int roff[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) roff[r] = rowok[r] ? (r - 1) * W : 0;
    float acc[MM][4];
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;
    const int cper = (d.C + NW - 1) / NW;
    const int c_begin = min(d.C, wave * cper), c_end = min(d.C, c_begin + cper);
    const float* __restrict__ Wp = d.A;
#pragma unroll 2
    for (int c = c_begin; c < c_end; ++c) {
        const float* __restrict__ p = xb + (int64_t)c * HWs;
        float e[3][6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float* __restrict__ q = p + roff[r];
            const f32x4 v = *reinterpret_cast<const f32x4*>(q);
            // the two edge pixels are the neighbouring lanes' (lane - 1: the quad to the left in the same row, whenever lf; lane + 1 likewise): a lane
            // exchange instead of two more loads per row
            // (rows that do not divide the 64 quads of a workgroup -- W = 48 -- have row neighbours in other workgroups: loads, as before)
            const float l = xch ? __shfl_up(v[3], 1) : q[lf ? -1 : 0], rr = xch ? __shfl_down(v[0], 1) : q[rt ? 4 : 3];
            e[r][0] = (rowok[r] && lf) ? l : 0.f;
            e[r][5] = (rowok[r] && rt) ? rr : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) e[r][1 + j] = rowok[r] ? v[j] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            if (m < d.M) {                                    // uniform
                const float* __restrict__ wm = Wp + (int64_t)m * d.lda + (int64_t)c * 9;      // wave-uniform: scalar loads
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int q3 = 0; q3 < 3; ++q3) {
                        const float w = wm[r * 3 + q3];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[m][j] = fmaf(w, e[r][q3 + j], acc[m][j]);
                    }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int m = 0; m < MM; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[wave - 1][m][j][lane] = acc[m][j];
    }
    __syncthreads();
    if (wave == 0 && live) {
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            if (m < d.M) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[m][j];
#pragma unroll
                    for (int k = 0; k < NW - 1; ++k) v += red[k][m][j][lane];                                            // fixed order
                    v *= d.alpha;
                    if (d.bias) v += d.bias[m];
                    o[j] = v;
                }
                *reinterpret_cast<f32x4*>(d.D + (int64_t)b * d.d_bstride + (int64_t)m * d.ldd + y * W + 4 * x4) = o;
            }
        }
    }
}

static bool fewout_eligible(const vd_gemm_desc& d) {
    return d.W % 4 == 0 && d.C >= 16 && (d.b_bstride & 3) == 0 && (d.d_bstride & 3) == 0 && (d.ldd & 3) == 0 &&
           ((((uintptr_t)d.B) | ((uintptr_t)d.D)) & 15) == 0 && (int64_t)d.H * d.W == d.NP;
}

static bool smallm_eligible(const vd_gemm_desc& d) {
    if (d.b_mode != VD_B_CONV3 || d.a_mode != VD_A_ROW || d.M > 4 || d.tile != 0 || d.debug != 0 || d.act) return false;
    if (d.rowadd || d.residual || d.d_trans || d.accumulate || d.bias_on_n || d.nb2 > 1 || d.a_bstride != 0 || d.gn_ss) return false;
    if (d.OH != d.H || d.OW != d.W) return false;
    if (d.W % 32 == 0) return d.H % 8 == 0;
    return d.W == 16 && d.H % 16 == 0;
}

static int launch_smallm(const vd_gemm_desc& d, hipStream_t st) {
    const int nb = d.N / d.NP;
    constexpr int fewout_off = 0;
    if (!fewout_off && fewout_eligible(d)) {
        const int64_t quads = (int64_t)nb * d.H * (d.W / 4);
        // eight waves per 256 pixels (round 6; four until then): 31 against 48 us for conv_out at B = 128 -- the loop is a latency chain (loads -> 100 FMAs per
        // channel) and four waves per workgroup left two per SIMD; sixteen: 34 us
        hipLaunchKernelGGL((conv3_fewout_kernel<4, 8>), dim3((unsigned)((quads + 63) / 64)), dim3(512), 0, st, d, quads);
        return 0;
    }
    if (d.W % 32 == 0) {
        dim3 grid((unsigned)(nb * (d.H / 8) * (d.W / 32)));
        hipLaunchKernelGGL((conv3_smallm_kernel<32, 4>), grid, dim3(256), 0, st, d);
    } else {
        dim3 grid((unsigned)(nb * (d.H / 16)));
        hipLaunchKernelGGL((conv3_smallm_kernel<16, 4>), grid, dim3(256), 0, st, d);
    }
    return 0;
}

// ---- plain GEMM with immediate-offset operands (1x1 convolutions, attention contractions) -----------------------------
// D[b][m][p] = sum_k A[m][k] * B[b][k][p]  for the VD_B_PLAIN operand (pixel-contiguous B).  Same structure as the
// patch-staged convolution: LDS images are k-major ( As[32][128], Bs[32][128] ), an MFMA step reads A and B with ONE
// ds_read_b32 each at lane-constant base + immediate offset; B (and a column-major A) are staged with float4 global
// loads and ds_write_b128, a row-major A (weights) with float4 loads and transposing ds_write_b32.
template <int AMODE>
__global__ __launch_bounds__(NT, 3) void gemm_plain_kernel(const vd_gemm_desc d) {
    constexpr int WM = 2, WN = 2, BM = 128, BN = 128;
    constexpr int LD_ = 128 + 4;                   // floats; rows stay 16-B aligned, lanes (consecutive n) conflict-free
    constexpr int LDA_ = (AMODE == VD_A_ROW) ? 129 : 132;   // row-major A is written by transposing b32 stores: odd stride
    constexpr int F4 = BM * BK / 4 / NT;           // 4 float4 per thread per operand
    __shared__ __attribute__((aligned(16))) float As[BK * LDA_];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LD_];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int tiles_m = (d.M + BM - 1) / BM;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    }
    const int tm = bid % tiles_m, tn = bid / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int b = n0 / d.NP, p0 = n0 - b * d.NP;   // BN | NP: a tile never straddles batch items
    const float* __restrict__ Ap = d.A + batch_off(d, b, d.a_bstride, d.a_b2stride);
    const float* __restrict__ Bp = d.B + batch_off(d, b, d.b_bstride, d.b_b2stride) + p0;

    f32x4 ra[F4], rb[F4];
    auto load_stage = [&](int k0) {
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            const int idx = tid + i * NT;
            if (AMODE == VD_A_ROW) {               // A[m][k]: float4 along k
                const int m = idx >> 3, q = idx & 7;
                ra[i] = *reinterpret_cast<const f32x4*>(Ap + (int64_t)min(m0 + m, d.M - 1) * d.lda + k0 + 4 * q);
            } else {                               // A[k][m]: float4 along m
                const int k = idx >> 5, q = idx & 31;
                ra[i] = *reinterpret_cast<const f32x4*>(Ap + (int64_t)(k0 + k) * d.lda + min(m0 + 4 * q, d.M - 4));
            }
            const int k = idx >> 5, q = idx & 31;  // B[k][n]: float4 along n
            rb[i] = *reinterpret_cast<const f32x4*>(Bp + (int64_t)(k0 + k) * d.ldb + 4 * q);
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            const int idx = tid + i * NT;
            if (AMODE == VD_A_ROW) {
                const int m = idx >> 3, q = idx & 7;
                const bool ok = m0 + m < d.M;
#pragma unroll
                for (int j = 0; j < 4; ++j) As[(4 * q + j) * LDA_ + m] = ok ? ra[i][j] : 0.f;
            } else {
                const int k = idx >> 5, q = idx & 31;
                // M is a multiple of 4: a float4 group of rows is entirely inside or entirely outside the matrix
                const bool ok = m0 + 4 * q < d.M;
                *reinterpret_cast<f32x4*>(&As[k * LDA_ + 4 * q]) = ok ? ra[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const int k = idx >> 5, q = idx & 31;
            *reinterpret_cast<f32x4*>(&Bs[k * LD_ + 4 * q]) = rb[i];
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mi][ni][v] = 0.f;

    const int wm = wave >> 1, wn = wave & 1;
    const float* __restrict__ a_base = As + h * LDA_ + wm * 64 + (lane & 31);
    const float* __restrict__ b_base = Bs + h * LD_ + wn * 64 + (lane & 31);

    const int ksteps = d.K / BK;
    load_stage(0);
    store_stage();
    __syncthreads();
    for (int ks = 0; ks < ksteps; ++ks) {
        const bool more = ks + 1 < ksteps;
        if (more) load_stage((ks + 1) * BK);
        auto fetch = [&](int t, float (&a)[WM], float (&bb)[WN]) {
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) a[mi] = a_base[2 * t * LDA_ + mi * 32];
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) bb[ni] = b_base[2 * t * LD_ + ni * 32];
        };
        float a0[WM], b0[WN], a1[WM], b1[WN];
        fetch(0, a0, b0);
#pragma unroll
        for (int t = 0; t < BK / 2; t += 2) {     // operands of step t+1 are in flight while step t's MFMAs issue
            fetch(t + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mi], b0[ni], acc[mi][ni], 0, 0, 0);
            if (t + 2 < BK / 2) fetch(t + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[mi], b1[ni], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
        if (more) store_stage();
        __syncthreads();
    }
    gemm_epilogue<WM, WN>(d, acc, m0, n0, wm, wn, lane, h);
}

static bool plain_eligible(const vd_gemm_desc& d) {
    if (d.b_mode != VD_B_PLAIN || d.d_trans || d.debug != 0 || d.tile != 0) return false;
    if (d.NP % 128 != 0 || d.N % 128 != 0 || d.K % BK != 0 || d.M < 64) return false;
    if ((d.ldb & 3) || (d.b_bstride & 3) || (((uintptr_t)d.B) & 15)) return false;
    if ((d.lda & 3) || (d.a_bstride & 3) || (((uintptr_t)d.A) & 15)) return false;
    if (d.a_mode == VD_A_COL && (d.M & 3)) return false;
    // enough tiles to fill the chip, otherwise the smaller generic tiles do better
    return (int64_t)vd_cdiv(d.M, 128) * (d.N / 128) >= 192;
}

// ---- weight gradient ---------------------------------------------------------------------------------------------
// D[m][n=(c,t)] = sum_{kk=(b,p)} dY[b][m][p] * gather(X)[b][c][p (+) t];  split-K over kk, slabs reduced afterwards.
template <int WM, int WN, int BMODE>
__global__ __launch_bounds__(NT, 3) void wgrad_kernel(const vd_wgrad_desc d, int kk_per_split) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int LDA_ = BM + 1, LDB_ = BN + 1;
    constexpr int A_F4 = BM * KG / NT;
    constexpr int B_F4 = BN * KG / NT;
    __shared__ f32x4 As[KG * LDA_];
    __shared__ f32x4 Bs[KG * LDB_];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int Ncols = d.C * d.T;
    const int tiles_m = (d.M + BM - 1) / BM;
    int bx = blockIdx.x, by = blockIdx.y;      // XCD-aware remap: one XCD owns a contiguous range of K-splits (see wgrad_patch_kernel)
    {
        const int T = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * blockIdx.y;
        if ((T & 7) == 0) {
            const int v = (lin & 7) * (T >> 3) + (lin >> 3);
            bx = v % gridDim.x;
            by = v / gridDim.x;
        }
    }
    const int tm = bx % tiles_m, tn = bx / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int Ktot = d.nb * d.NP;
    const int kk_begin = by * kk_per_split;
    const int kk_end = min(Ktot, kk_begin + kk_per_split);

    // columns owned by this thread in the B loader: n_i = n0 + (tid>>3) + 32*i ; k-group kq = tid & 7
    int nc[B_F4], nr[B_F4], ns[B_F4];
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
        const int n = n0 + (tid >> 3) + 32 * i;
        if (n < Ncols) {
            if (d.T == 9) {
                kdecomp9(n, nc[i], nr[i], ns[i]);
            } else {
                nc[i] = n;
                nr[i] = ns[i] = 0;
            }
        } else {
            nc[i] = -1;
            nr[i] = ns[i] = 0;
        }
    }
    const int kq = tid & 7;
    const int HW = d.H * d.W;

    f32x4 ra[A_F4], rb[B_F4];
    unsigned amask = 0, bmask = 0;     // validity bits; the zeroing select is deferred to store_ab() (after the MFMAs)
    auto load_ab = [&](int kk0) {
        amask = bmask = 0;
        // A: dY[b][m][p..p+3], lanes: (m = idx>>3, kq = idx&7).  Unconditional loads from clamped addresses.
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int idx = tid + i * NT;
            const int m = m0 + (idx >> 3), kk = kk0 + (idx & 7) * 4;
            const bool ok = m < d.M && kk < kk_end;
            const int kc = ok ? kk : 0;
            const int b = kc / d.NP, p = kc - b * d.NP;
            ra[i] = *reinterpret_cast<const f32x4*>(d.dY + (int64_t)b * d.dy_bstride + (int64_t)(ok ? m : 0) * d.NP + p);
            amask |= (ok ? 0xFu : 0u) << (4 * i);
        }
        // B: gather of X for 4 consecutive output pixels of one row
        const int kk = kk0 + kq * 4;
        const bool kvalid = kk < kk_end;
        const int kc = kvalid ? kk : 0;
        const int b = kc / d.NP;
        const int p = kc - b * d.NP;
        const int oy = p / d.OW, ox0 = p - oy * d.OW;
        const float* __restrict__ xb = d.X + (int64_t)b * d.x_bstride;
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const bool cok = kvalid && nc[i] >= 0;
            const float* __restrict__ xc = xb + (int64_t)(cok ? nc[i] : 0) * HW;
            if (BMODE == VD_B_PLAIN) {
                rb[i] = *reinterpret_cast<const f32x4*>(xc + p);
                bmask |= (cok ? 0xFu : 0u) << (4 * i);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ox = ox0 + j;
                    int iy, ix;
                    bool ok;
                    if (BMODE == VD_B_CONV3) {
                        iy = oy + nr[i] - 1;
                        ix = ox + ns[i] - 1;
                        ok = (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
                    } else if (BMODE == VD_B_CONV3_S2) {
                        iy = 2 * oy + nr[i] - d.pad;
                        ix = 2 * ox + ns[i] - d.pad;
                        ok = (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
                    } else {  // VD_B_CONV3_UP
                        const int uy = oy + nr[i] - 1, ux = ox + ns[i] - 1;
                        ok = (unsigned)uy < (unsigned)(2 * d.H) && (unsigned)ux < (unsigned)(2 * d.W);
                        iy = uy >> 1;
                        ix = ux >> 1;
                    }
                    ok = ok && cok;
                    rb[i][j] = xc[ok ? iy * d.W + ix : 0];
                    bmask |= (ok ? 1u : 0u) << (4 * i + j);
                }
            }
        }
    };
    auto masked = [](f32x4 v, unsigned bits) {
        return f32x4{(bits & 1u) ? v[0] : 0.f, (bits & 2u) ? v[1] : 0.f, (bits & 4u) ? v[2] : 0.f, (bits & 8u) ? v[3] : 0.f};
    };
    auto store_ab = [&]() {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int idx = tid + i * NT;
            As[(idx & 7) * LDA_ + (idx >> 3)] = masked(ra[i], amask >> (4 * i));
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) Bs[kq * LDB_ + (tid >> 3) + 32 * i] = masked(rb[i], bmask >> (4 * i));
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mi][ni][v] = 0.f;

    const int wm = wave >> 1, wn = wave & 1;
    const int arow = wm * 32 * WM + (lane & 31);
    const int bcol = wn * 32 * WN + (lane & 31);

    if (kk_begin < kk_end) {
        load_ab(kk_begin);
        store_ab();
        __syncthreads();
        for (int kk0 = kk_begin; kk0 < kk_end; kk0 += BK) {
            const bool more = kk0 + BK < kk_end;
            if (more) load_ab(kk0 + BK);
            mma_stage<WM, WN, LDA_, LDB_>(As, Bs, arow, bcol, h, acc);
            __syncthreads();
            if (more) store_ab();
            __syncthreads();
        }
    }

    float* __restrict__ out = (gridDim.y > 1) ? (d.ws + (int64_t)by * d.M * Ncols) : d.dW;
    const bool accum = (gridDim.y == 1) && d.accumulate;
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
        const int n = n0 + wn * 32 * WN + ni * 32 + (lane & 31);
        if (n >= Ncols) continue;
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + wm * 32 * WM + mi * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                if (m >= d.M) continue;
                const int64_t off = (int64_t)m * Ncols + n;
                out[off] = accum ? (out[off] + acc[mi][ni][v]) : acc[mi][ni][v];
            }
    }
}

// ---- patch-staged weight gradient (3x3, stride 1; optionally through the fused nearest-2x upsample) --------------
// One workgroup owns 128 output channels (rows, from dY) x 64 input channels x the 3 horizontal taps s of ONE tap row r
// and a contiguous range of K-steps; a K-step is 32 consecutive output pixels (ROWS = 32/W full rows of one image):
//     As[8][128][4]        dY in the b128 k-group layout of gemm_kernel (pixel groups of 4)
//     Bs[64][ROWS][W+2]    the X rows y+r-1 with a one-pixel halo, each element loaded ONCE; the three taps s are
//                          three LDS address offsets into the same patch (6 ds_read_b32 feed 12 MFMAs per channel group)
// Per 8 pixels a wave issues 2 ds_read_b128 + 6 ds_read_b32 for 24 MFMAs; no address arithmetic in the inner loop.
// Split-K partial slabs are reduced in fixed order by slab_reduce_kernel (deterministic).
template <int W, int MODE>  // MODE 0: CONV3, 2: CONV3_UP (X is the half-resolution source)
__global__ __launch_bounds__(NT, 3) void wgrad_patch_kernel(const vd_wgrad_desc d, int ksteps_per_split) {
    constexpr int ROWS = 32 / W;
    constexpr int PW = W + 2;
    constexpr int PLn = ROWS * PW;                 // patch floats per channel
    constexpr int CT = 64;                         // input channels per workgroup
    constexpr int LDA_ = 128 + 1;                  // float4 units
    constexpr int LDB_ = (PLn & 1) ? PLn : PLn + 1;
    constexpr int A_F4 = 128 * 32 / 4 / NT;        // 4
    constexpr int P_EL = (CT * PLn + NT - 1) / NT; // 9
    __shared__ f32x4 As[KG * LDA_];
    __shared__ float Bs[CT * LDB_];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int tiles_m = (d.M + 127) / 128;
    // XCD-aware remap: hardware deals linear workgroup ids round-robin over the 8 XCDs; make every XCD own a contiguous
    // range of K-splits so that the (m-tile, c-tile, r) workgroups re-reading one dY / X slice share that XCD's L2.
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int T = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * blockIdx.y;
        if ((T & 7) == 0) {
            const int v = (lin & 7) * (T >> 3) + (lin >> 3);
            bx = v % gridDim.x;
            by = v / gridDim.x;
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
    const int HWs = d.H * d.W;

    // patch element -> column part of the source offset + validity (fixed); the row term is added per K-step
    int pcol_off[P_EL];
    unsigned pmask = 0, prowbits = 0;              // bit i: element valid / element lies in patch row 1
#pragma unroll
    for (int i = 0; i < P_EL; ++i) {
        const int e = tid + i * NT;
        const int c = e / PLn, rem = e - c * PLn;
        const int rr = rem / PW, px = rem - rr * PW;
        int ix = px - 1;
        bool ok = e < CT * PLn && (c0 + c) < d.C;
        if (MODE == 2) {
            ok = ok && (unsigned)ix < (unsigned)(2 * d.W);
            ix >>= 1;
        } else {
            ok = ok && (unsigned)ix < (unsigned)d.W;
        }
        pcol_off[i] = ok ? ((c0 + c) * HWs + ix) : 0;
        pmask |= (ok ? 1u : 0u) << i;
        prowbits |= ((ROWS > 1 && rr > 0) ? 1u : 0u) << i;
    }

    f32x4 ra[A_F4];
    float rp[P_EL];
    unsigned rowmask = 0;                          // bit rr: source row of patch row rr is inside the image
    auto load_stage = [&](int ks) {
        const int b = ks / steps_per_img;
        const int y0 = (ks - b * steps_per_img) * ROWS;
        const float* __restrict__ dyb = d.dY + (int64_t)b * d.dy_bstride + y0 * W;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int idx = tid + i * NT;
            const int m = idx >> 3, q = idx & 7;
            const int mm = min(m0 + m, d.M - 1);
            ra[i] = *reinterpret_cast<const f32x4*>(dyb + (int64_t)mm * d.NP + 4 * q);
        }
        const float* __restrict__ xb = d.X + (int64_t)b * d.x_bstride;
        int rowoff[2] = {0, 0};
        rowmask = 0;
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            int iy = y0 + rr + r - 1;
            bool ok;
            if (MODE == 2) {
                ok = (unsigned)iy < (unsigned)(2 * d.H);
                iy >>= 1;
            } else {
                ok = (unsigned)iy < (unsigned)d.H;
            }
            rowoff[rr] = ok ? iy * d.W : 0;
            rowmask |= (ok ? 1u : 0u) << rr;
        }
#pragma unroll
        for (int i = 0; i < P_EL; ++i) {
            const int ro = ((prowbits >> i) & 1u) ? rowoff[1] : rowoff[0];
            rp[i] = xb[pcol_off[i] + ro];
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int idx = tid + i * NT;
            const int m = idx >> 3, q = idx & 7;
            const bool ok = m0 + m < d.M;
            As[q * LDA_ + m] = ok ? ra[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < P_EL; ++i) {
            const int e = tid + i * NT;
            if (e < CT * PLn) {
                const int c = e / PLn, rem = e - c * PLn;
                const bool ok = ((pmask >> i) & 1u) && ((rowmask >> ((prowbits >> i) & 1u)) & 1u);
                Bs[c * LDB_ + rem] = ok ? rp[i] : 0.f;
            }
        }
    };

    f32x16 acc[2][3];                              // [m group][tap s]
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int sx = 0; sx < 3; ++sx)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mi][sx][v] = 0.f;

    const int wm = wave >> 1, wc = wave & 1;       // 2 x 2 waves: 64 m x 32 c each
    const f32x4* __restrict__ a_base = As + h * LDA_ + wm * 64 + (lane & 31);
    const float* __restrict__ b_base = Bs + (wc * 32 + (lane & 31)) * LDB_ + 4 * h;

    if (ks_begin < ks_end) {
        load_stage(ks_begin);
        store_stage();
        __syncthreads();
        for (int ks = ks_begin; ks < ks_end; ++ks) {
            const bool more = ks + 1 < ks_end;
            if (more) load_stage(ks + 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {          // 8 pixels: k-slot h covers pixels 8g+4h .. 8g+4h+3
                const int rr = (8 * g) / W, x0 = 8 * g - rr * W;
                f32x4 a[2];
                float bw[6];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[mi] = a_base[2 * g * LDA_ + mi * 32];
#pragma unroll
                for (int j = 0; j < 6; ++j) bw[j] = b_base[rr * PW + x0 + j];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int sx = 0; sx < 3; ++sx)
                            acc[mi][sx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][t], bw[t + sx], acc[mi][sx], 0, 0, 0);
            }
            __syncthreads();
            if (more) store_stage();
            __syncthreads();
        }
    }

    const int Ncols = d.C * 9;
    const int c = c0 + wc * 32 + (lane & 31);
    if (gridDim.y > 1) {
        // split-K slab in the PERMUTED layout ws[z][r][m][c][3]: for one row m the 32 lanes (consecutive c) write 32 x 12 B
        // contiguous bytes, instead of 12-B pieces at a 36-B stride in the weight layout; slab_reduce_perm_kernel un-permutes.
        float* __restrict__ slab = d.ws + (int64_t)by * d.M * Ncols + (int64_t)r * d.M * d.C * 3;
        if (c < d.C) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int m = m0 + wm * 64 + mi * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                    if (m >= d.M) continue;
                    float* __restrict__ o = slab + ((int64_t)m * d.C + c) * 3;
#pragma unroll
                    for (int sx = 0; sx < 3; ++sx) o[sx] = acc[mi][sx][v];
                }
        }
        return;
    }
    float* __restrict__ out = d.dW;
    const bool accum = d.accumulate;
    if (c < d.C) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + wm * 64 + mi * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                if (m >= d.M) continue;
                const int64_t off = (int64_t)m * Ncols + c * 9 + r * 3;
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) out[off + sx] = accum ? (out[off + sx] + acc[mi][sx][v]) : acc[mi][sx][v];
            }
    }
}

// Epilogue shared by the two patch weight-gradient kernels: split-K slab in the permuted layout, or dW directly.
__device__ __forceinline__ void wgrad_patch_store(const vd_wgrad_desc& d, const f32x16 (&acc)[2][3], int r, int m0, int c0, int wm,
                                                  int wc, int lane, int h, int by, int nsplit) {
    const int Ncols = d.C * 9;
    const int c = c0 + wc * 32 + (lane & 31);
    if (c >= d.C) return;
    if (nsplit > 1) {
        float* __restrict__ slab = d.ws + (int64_t)by * d.M * Ncols + (int64_t)r * d.M * d.C * 3;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + wm * 64 + mi * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                if (m >= d.M) continue;
                float* __restrict__ o = slab + ((int64_t)m * d.C + c) * 3;
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) o[sx] = acc[mi][sx][v];
            }
        return;
    }
    float* __restrict__ out = d.dW;
    const bool accum = d.accumulate;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int m = m0 + wm * 64 + mi * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
            if (m >= d.M) continue;
            const int64_t off = (int64_t)m * Ncols + c * 9 + r * 3;
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) out[off + sx] = accum ? (out[off + sx] + acc[mi][sx][v]) : acc[mi][sx][v];
        }
}

#include "vd_wgrad_bx3.inc"

// Generalisation of wgrad_patch_kernel to the other image widths.  A K-step is still 32 output pixels = ROWS rows of
// TW = 32/ROWS pixels.  Rows are numbered globally (gr = image*OH + y, OH a power of two), so a step may span images
// (4x4 outputs: ROWS = 8 = two images); for OW >= 64 (ROWS == 1) a step is one 32-pixel segment of a row and the halo
// columns come from the neighbouring segments.  Same LDS images, MFMA schedule and epilogue as wgrad_patch_kernel; the
// source offsets of the patch elements are recomputed per K-step (a dozen integer ops per element against 96 MFMAs).
// 2 workgroups/CU: at 3 the per-step address state spills (26-41 VGPRs) and the kernel drops from 115 to 85 TF.
// KPIX = output pixels per K-step (32, or 64: twice the MFMAs between barriers).
template <int ROWS, int MODE, int KPIX>  // MODE 0: CONV3, 2: CONV3_UP (X is the half-resolution source)
__global__ __launch_bounds__(NT, 2) void wgrad_patch_gen_kernel(const vd_wgrad_desc d, int ksteps_per_split, int oh_shift,
                                                                 int segs_shift) {
    constexpr int TW = KPIX / ROWS;
    constexpr int KGX = KPIX / 4;                  // k-groups (float4 of pixels) per K-step
    constexpr int PW = TW + 2;
    constexpr int PLn = ROWS * PW;
    constexpr int CT = 64;
    constexpr int LDA_ = 128 + 1;
    constexpr int LDB_ = (PLn & 1) ? PLn : PLn + 1;
    constexpr int A_F4 = 128 * KPIX / 4 / NT;
    constexpr int P_EL = (CT * PLn + NT - 1) / NT;
    __shared__ f32x4 As[KGX * LDA_];
    __shared__ float Bs[CT * LDB_];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int tiles_m = (d.M + 127) / 128;
    int bx = blockIdx.x, by = blockIdx.y;      // XCD-aware remap (see wgrad_patch_kernel)
    {
        const int T = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * blockIdx.y;
        if ((T & 7) == 0) {
            const int v = (lin & 7) * (T >> 3) + (lin >> 3);
            bx = v % gridDim.x;
            by = v / gridDim.x;
        }
    }
    const int r = bx % 3;
    const int rest = bx / 3;
    const int tm = rest % tiles_m, tc = rest / tiles_m;
    const int m0 = tm * 128, c0 = tc * CT;
    const int ks_total = (d.nb * d.NP) / KPIX;
    const int ks_begin = by * ksteps_per_split;
    const int ks_end = min(ks_total, ks_begin + ksteps_per_split);
    const int HWs = d.H * d.W;
    const int oh_mask = d.OH - 1, seg_mask = (1 << segs_shift) - 1;
    const int xbs32 = (int)d.x_bstride;

    // per patch element ONE packed register: LDS slot (13 bits) | channel (6) | patch row (4) | patch column (7) | valid ch | exists
    unsigned pk[P_EL];
#pragma unroll
    for (int i = 0; i < P_EL; ++i) {
        const int e = tid + i * NT;
        const int c = e / PLn, rem = e - c * PLn;
        const int rr = rem / PW, px = rem - rr * PW;
        const bool in = e < CT * PLn;
        pk[i] = in ? ((unsigned)(c * LDB_ + rem) | ((unsigned)c << 13) | ((unsigned)rr << 19) | ((unsigned)px << 23) |
                      (((c0 + c) < d.C ? 1u : 0u) << 30) | (1u << 31))
                   : 0u;
    }
    const int qa = tid & (KGX - 1);                // this thread's k-group (4 pixels) in the dY loader
    const int rrA = (4 * qa) / TW, xA = (4 * qa) % TW;

    f32x4 ra[A_F4];
    float rp[P_EL];
    unsigned okmask = 0;
    auto load_stage = [&](int ks) {
        const int xs = ROWS == 1 ? (ks & seg_mask) * KPIX : 0;
        const int gr0 = ROWS == 1 ? (ks >> segs_shift) : ks * ROWS;
        {
            const int gr = gr0 + rrA;
            const int b = gr >> oh_shift, y = gr & oh_mask;
            const float* __restrict__ dyp = d.dY + (int64_t)b * d.dy_bstride + y * d.OW + xs + xA;
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const int m = (tid + i * NT) / KGX;
                const int mm = min(m0 + m, d.M - 1);
                ra[i] = *reinterpret_cast<const f32x4*>(dyp + (int64_t)mm * d.NP);
            }
        }
        okmask = 0;
        const int b0 = gr0 >> oh_shift;
        const float* __restrict__ xb = d.X + (int64_t)b0 * d.x_bstride;
#pragma unroll
        for (int i = 0; i < P_EL; ++i) {
            const unsigned u = pk[i];
            const int gr = gr0 + (int)((u >> 19) & 15u);
            const int b = gr >> oh_shift, y = gr & oh_mask;
            int iy = y + r - 1, ix = xs + (int)((u >> 23) & 127u) - 1;
            bool ok;
            if (MODE == 2) {
                ok = (unsigned)iy < (unsigned)(2 * d.H) && (unsigned)ix < (unsigned)(2 * d.W);
                iy >>= 1;
                ix >>= 1;
            } else {
                ok = (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
            }
            ok = ok && ((u >> 30) & 1u);
            const int off = ok ? ((b - b0) * xbs32 + (c0 + (int)((u >> 13) & 63u)) * HWs + iy * d.W + ix) : 0;
            rp[i] = xb[off];                           // uniform 64-bit base + 32-bit lane offset
            okmask |= (ok ? 1u : 0u) << i;
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int idx = tid + i * NT;
            const int m = idx / KGX, q = idx & (KGX - 1);
            const bool ok = m0 + m < d.M;
            As[q * LDA_ + m] = ok ? ra[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < P_EL; ++i)
            if ((pk[i] >> 31) & 1u) Bs[pk[i] & 8191u] = ((okmask >> i) & 1u) ? rp[i] : 0.f;
    };

    f32x16 acc[2][3];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int sx = 0; sx < 3; ++sx)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mi][sx][v] = 0.f;

    const int wm = wave >> 1, wc = wave & 1;
    const f32x4* __restrict__ a_base = As + h * LDA_ + wm * 64 + (lane & 31);
    // k-slot h covers pixels 8g+4h..8g+4h+3: the next 4 pixels of the row, or (4-pixel rows) the next row
    const float* __restrict__ b_base = Bs + (wc * 32 + (lane & 31)) * LDB_ + (TW == 4 ? h * PW : 4 * h);

    if (ks_begin < ks_end) {
        load_stage(ks_begin);
        store_stage();
        __syncthreads();
        for (int ks = ks_begin; ks < ks_end; ++ks) {
            const bool more = ks + 1 < ks_end;
            if (more) load_stage(ks + 1);
#pragma unroll
            for (int g = 0; g < KPIX / 8; ++g) {
                const int boff = TW == 4 ? (2 * g) * PW : ((8 * g) / TW) * PW + (8 * g) % TW;
                f32x4 a[2];
                float bw[6];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[mi] = a_base[2 * g * LDA_ + mi * 32];
#pragma unroll
                for (int j = 0; j < 6; ++j) bw[j] = b_base[boff + j];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int sx = 0; sx < 3; ++sx)
                            acc[mi][sx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][t], bw[t + sx], acc[mi][sx], 0, 0, 0);
            }
            __syncthreads();
            if (more) store_stage();
            __syncthreads();
        }
    }
    wgrad_patch_store(d, acc, r, m0, c0, wm, wc, lane, h, by, gridDim.y);
}


static int ilog2_exact(int v) {  // log2 of a power of two, -1 otherwise
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

// 0: not eligible; 1: wgrad_patch_kernel (OW 16 / 32); 2: wgrad_patch_gen_kernel (OW 4 / 8 / multiples of 32 from 64 up)
static int wgrad_patch_kind(const vd_wgrad_desc& d) {
    if (d.T == 9 && d.mode == VD_B_CONV3_S2 && d.math == 1) {    // round 4: the stride-2 Downsample2D convolution on the split-precision kernel (8x8 / 16x16 outputs)
        if (d.NP != d.OH * d.OW || d.M < 64 || d.C < 64 || d.tile != 0) return -1;
        // 8x8 / 16x16 / 32x32 outputs, or 32-pixel row segments of wider ones (config #4's 128x128 / 64x64 Downsample2D outputs)
        if (d.OH == d.OW && (d.OW == 8 || d.OW == 16 || d.OW == 32 || (d.OW >= 64 && d.OW % 32 == 0)) && d.H == 2 * d.OH && d.W == 2 * d.OW &&
            (d.pad == 0 || d.pad == 1) && (d.x_bstride & 3) == 0 && ((((uintptr_t)d.X) & 15) == 0) && (int64_t)d.C * d.H * d.W < (1ll << 29))
            return 4;
        return -1;
    }
    if (d.T != 9 || (d.mode != VD_B_CONV3 && d.mode != VD_B_CONV3_UP)) return 0;
    if (d.NP != d.OH * d.OW || d.M < 64 || d.C < 64 || d.tile != 0) return 0;
    if (d.math == 1) {      // split-precision kernel (explicit request): stride-1 3x3 at 8x8 / 16x16 / 32x32
        const bool up = d.mode == VD_B_CONV3_UP;
        if ((d.mode == VD_B_CONV3 || up) && d.OH == d.OW && (d.OW == 8 || d.OW == 16 || d.OW == 32) && d.H * (up ? 2 : 1) == d.OH &&
            d.W * (up ? 2 : 1) == d.OW && (d.x_bstride & 3) == 0 && ((((uintptr_t)d.X) & 15) == 0))
            return 4;
        if (d.mode == VD_B_CONV3 && d.OW == 4 && d.OH == 4 && d.H == 4 && d.W == 4 && (d.x_bstride & 3) == 0 && (d.dy_bstride & 3) == 0 &&
            ((((uintptr_t)d.X) & 15) == 0))
            return 4;                           // 4x4 outputs: two whole images per K-step
        if ((d.mode == VD_B_CONV3 || up) && d.OW >= 64 && d.OW % 32 == 0 && d.H * (up ? 2 : 1) == d.OH && d.W * (up ? 2 : 1) == d.OW &&
            (d.x_bstride & 3) == 0 && ((((uintptr_t)d.X) & 15) == 0))
            return 4;                           // wide images: 32-pixel row segments
        return -1;
    }
    constexpr bool k64 = false;          // (64-pixel K-steps for the 16 / 32 px layers: measured neutral, profiles/HISTORY.md)
    if (k64 && (d.OW == 16 || d.OW == 32) && ilog2_exact(d.OH) >= 0 && ((int64_t)d.nb * d.NP) % 64 == 0) return 3;
    if (d.OW == 16 || d.OW == 32) return d.OH % (32 / d.OW) == 0 ? 1 : 0;
    if (ilog2_exact(d.OH) < 0 || ((int64_t)d.nb * d.NP) % 32 != 0) return 0;
    if (d.x_bstride * (int64_t)(8 / d.OH + 2) >= (1ll << 31)) return 0;      // 32-bit in-step offsets
    // 4x4 / 8x8 outputs: K = nb*NP is so short that both kernels are prologue/slab bound; measured on MI355X the generic
    // kernel is as fast or faster there (75 vs 71 TF at 8x8, 48 vs 32 TF at 4x4), so the patch variant is opt-in.
    constexpr bool small_patch = false;
    if (d.OW == 4 || d.OW == 8) return small_patch ? 2 : 0;
    if (d.OW >= 64 && d.OW % 32 == 0 && ilog2_exact(d.OW / 32) >= 0) return 2;
    return 0;
}
static bool wgrad_patch_eligible(const vd_wgrad_desc& d) { return wgrad_patch_kind(d) > 0; }

static void wgrad_patch_plan(const vd_wgrad_desc& d, int& splits, int& ks_per) {
    const int kpix = wgrad_patch_kind(d) == 3 ? 64 : 32;
    const int ks_total = (int)(((int64_t)d.nb * d.NP + kpix - 1) / kpix);      // K-step = 32 (64) output pixels (4x4: two images, the last may be half empty)
    const int base = vd_cdiv(d.M, 128) * vd_cdiv(d.C, 64) * 3;
    splits = d.splits;
    if (splits <= 0) {  // ~3 workgroups per CU, at least 8 K-steps per split
        constexpr int target = 768;
        splits = vd_cdiv(target, base);
        const int max_splits = ks_total / 8 > 0 ? ks_total / 8 : 1;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    ks_per = vd_cdiv(ks_total, splits);
    splits = vd_cdiv(ks_total, ks_per);
}

// dW[i] (+)= sum_z ws[z][i], fixed order.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                          int64_t n4, int splits, int accumulate) {
    const f32x4* __restrict__ w4 = reinterpret_cast<const f32x4*>(ws);
    f32x4* __restrict__ o4 = reinterpret_cast<f32x4*>(out);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 s = w4[i];
        for (int z = 1; z < splits; ++z) s += w4[(int64_t)z * n4 + i];
        if (accumulate) s += o4[i];
        o4[i] = s;
    }
}
// Reduce of the permuted wgrad slabs ws[z][r][m][c][3] -> dW[m][c][r][3]: slab-order iteration (the splits x larger read
// side is coalesced), fixed summation order.
__global__ __launch_bounds__(256) void slab_reduce_perm_kernel(const float* __restrict__ ws, float* __restrict__ out, int M,
                                                               int C, int splits, int accumulate) {
    const int64_t n = (int64_t)M * C * 9;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = ws[i];
        for (int z = 1; z < splits; ++z) s += ws[(int64_t)z * n + i];
        const int sx = (int)(i % 3);
        const int64_t t = i / 3;
        const int c = (int)(t % C);
        const int64_t t2 = t / C;
        const int m = (int)(t2 % M), r = (int)(t2 / M);
        const int64_t o = ((int64_t)m * C + c) * 9 + r * 3 + sx;
        out[o] = accumulate ? (out[o] + s) : s;
    }
}

__global__ __launch_bounds__(256) void slab_reduce_scalar_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                                 int64_t n, int splits, int accumulate) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = ws[i];
        for (int z = 1; z < splits; ++z) s += ws[(int64_t)z * n + i];
        if (accumulate) s += out[i];
        out[i] = s;
    }
}

__global__ __launch_bounds__(256) void wtranspose_kernel(const float* __restrict__ W, float* __restrict__ Wt, int M, int C,
                                                         int T) {
    const int64_t total = (int64_t)M * C * T;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // i indexes Wt[c][m][t]
        const int t = (int)(i % T);
        const int64_t cm = i / T;
        const int m = (int)(cm % M), c = (int)(cm / M);
        Wt[i] = W[((int64_t)m * C + c) * T + t];
    }
}

__global__ __launch_bounds__(256) void sumpool2x2_kernel(const float* __restrict__ dU, float* __restrict__ dX, int B, int C,
                                                         int H, int W, int64_t du_bs, int64_t dx_bs, int accumulate) {
    const int64_t per = (int64_t)C * H * W, total = per * B;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const int64_t r = i - (int64_t)b * per;
        const int x = (int)(r % W);
        const int64_t cy = r / W;
        const int y = (int)(cy % H);
        const int64_t c = cy / H;
        const float* u = dU + (int64_t)b * du_bs + (c * 2 * H + 2 * y) * (2 * W) + 2 * x;
        float s = (u[0] + u[1]) + (u[2 * W] + u[2 * W + 1]);
        float* o = dX + (int64_t)b * dx_bs + r;
        *o = accumulate ? (*o + s) : s;
    }
}

// col2im of the stride-2 dgrad: G[b][(c,r,s)][oy][ox] = sum_m W[m][c][r][s] dY[b][m][oy][ox] (a plain GEMM, no structural
// zeros) is gathered into dX[b][c][y][x] = sum_{r,s : y-r, x-s even} G[b][(c,r,s)][(y-r)/2][(x-s)/2]; fixed tap order.
__global__ __launch_bounds__(256) void col2im_s2_kernel(const float* __restrict__ G, float* __restrict__ dX, int B, int C,
                                                        int H, int W, int OH, int OW, int pad, int64_t g_bs, int64_t dx_bs) {
    const int64_t per = (int64_t)C * H * W, total = per * B;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const int64_t q = i - (int64_t)b * per;
        const int x = (int)(q % W);
        const int64_t cy = q / W;
        const int y = (int)(cy % H);
        const int c = (int)(cy / H);
        const float* g = G + (int64_t)b * g_bs + (int64_t)c * 9 * OH * OW;
        const int yp = y + pad, xp = x + pad;
        float acc = 0.f;
        for (int r = yp & 1; r < 3; r += 2) {
            const int iy = (yp - r) >> 1;
            if (yp < r || iy >= OH) continue;
            for (int s = xp & 1; s < 3; s += 2) {
                const int ix = (xp - s) >> 1;
                if (xp < s || ix >= OW) continue;
                acc += g[(r * 3 + s) * OH * OW + iy * OW + ix];
            }
        }
        dX[(int64_t)b * dx_bs + q] = acc;
    }
}

// Long rows (P >= 8192 floats: the 128x128 / 256x256 feature maps of config #4): the FOUR waves of a workgroup share one row (a quarter each, four
// float4 loads in flight per lane), partial sums combined in fixed order -- one wave per 256 KB row left the launch at ~2.7 TB/s with 1024 waves.
__global__ __launch_bounds__(256) void rowsum_long_kernel(const float* __restrict__ X, float* __restrict__ ws, int M, int P, int64_t x_bs, int64_t ws_ld) {
    __shared__ float part[4];
    const int row = blockIdx.x, b = row / M, m = row - b * M;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int Q = P >> 4;                                    // float4 per wave (P % 16 == 0)
    const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(X + (int64_t)b * x_bs + (int64_t)m * P) + (int64_t)w * Q;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = lane;
    for (; i + 192 < Q; i += 256) {
        const f32x4 a = x4[i], c = x4[i + 64], e = x4[i + 128], f = x4[i + 192];
        s0 += (a[0] + a[1]) + (a[2] + a[3]);
        s1 += (c[0] + c[1]) + (c[2] + c[3]);
        s2 += (e[0] + e[1]) + (e[2] + e[3]);
        s3 += (f[0] + f[1]) + (f[2] + f[3]);
    }
    for (; i < Q; i += 64) {
        const f32x4 a = x4[i];
        s0 += (a[0] + a[1]) + (a[2] + a[3]);
    }
    const float s = wave_sum((s0 + s1) + (s2 + s3));
    if (lane == 0) part[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) ws[(int64_t)b * ws_ld + m] = (part[0] + part[1]) + (part[2] + part[3]);
}

// ws[b][m] = sum_p X[b][m][p] ; one wave per (b, m) row.
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ X, float* __restrict__ ws, int B, int M, int P,
                                                     int64_t x_bs, int64_t ws_ld) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * M) return;
    const int b = row / M, m = row - b * M;
    const float* __restrict__ x = X + (int64_t)b * x_bs + (int64_t)m * P;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    if ((P & 3) == 0 && ((((uintptr_t)x) & 15) == 0)) {
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
        const int Q = P >> 2;
        int i = lane;
        for (; i + 192 < Q; i += 256) {                             // four independent 16-byte loads in flight per lane (a 32x32 row is exactly one round);
            const f32x4 v0 = x4[i], v1 = x4[i + 64], v2 = x4[i + 128], v3 = x4[i + 192];      // the additions keep the order of the one-load loop
            s += (v0[0] + v0[1]) + (v0[2] + v0[3]);
            s += (v1[0] + v1[1]) + (v1[2] + v1[3]);
            s += (v2[0] + v2[1]) + (v2[2] + v2[3]);
            s += (v3[0] + v3[1]) + (v3[2] + v3[3]);
        }
        for (; i < Q; i += 64) {
            f32x4 v = x4[i];
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
    } else {
        for (int i = lane; i < P; i += 64) s += x[i];
    }
    s = wave_sum(s);
    if (lane == 0) ws[(int64_t)b * ws_ld + m] = s;
}

// out[c] (+)= sum_b ws[b][c].  64 columns per workgroup; wave w sums rows b = w, w+4, ... with 8 independent loads in
// flight, then the four partials are combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ ws, float* __restrict__ out, int B, int C,
                                                     int64_t ld, int accumulate) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < C) {
        int b = w;
        for (; b + 28 < B; b += 32) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = ws[(int64_t)(b + 4 * u) * ld + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; b < B; b += 4) s += ws[(int64_t)b * ld + c];
    }
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < C) {
        const float t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        out[c] = accumulate ? (out[c] + t) : t;
    }
}

// ---- direct 3x3 weight gradient when one side has <= 4 channels (conv_in 3 -> 128, conv_out 128 -> 3) -------------------------
// An MFMA tile would be > 90 % padding (measured: 231 us for 0.9 GFLOP on the generic kernel); the work is one pass over the BIG
// operand (67 MB at B = 128).  A workgroup owns one image and 32 channels of the big side; the small side's image (<= 4 channels,
// zero halo) sits in LDS.  Lane = column x, 8 channel slots per workgroup: a thread reads big[c][y][x] once (coalesced rows) and
// updates its 9 x CS accumulators from LDS; a 32-lane shuffle tree finishes each channel.  Images larger than 32 x 32 are cut
// into 32 x 32 pixel tiles (one workgroup each).  Per-(image, tile) partials go to ws[part][M*C*9] in the weight layout and are
// summed in fixed order by colsum_kernel (deterministic).
//   BIG_IS_X:  big = X (C channels), small = dY (M <= 4):  dW[m][c][r][s] += X[c][y][x] * dY[m][y - r + 1][x - s + 1]
//   otherwise: big = dY (M channels), small = X (C <= 4):  dW[m][c][r][s] += dY[m][y][x] * X[c][y + r - 1][x + s - 1]
template <bool BIG_IS_X>
__global__ __launch_bounds__(256) void wgrad_small_kernel(const vd_wgrad_desc d) {
    constexpr int MAXS = 4, PWM = 34;
    __shared__ float S[MAXS * PWM * PWM];
    const int CB = BIG_IS_X ? d.C : d.M, CS = BIG_IS_X ? d.M : d.C;
    const int H = d.OH, W = d.OW, HW = H * W;
    constexpr int PW = PWM, PHW = PWM * PWM;                      // a 32 x 32 pixel tile of the image + halo
    const int cb_blocks = (CB + 31) / 32, tiles_x = (W + 31) / 32, tiles = tiles_x * ((H + 31) / 32);
    int rest = blockIdx.x;
    const int cb0 = (rest % cb_blocks) * 32;
    rest /= cb_blocks;
    const int tile = rest % tiles, b = rest / tiles;
    const int y0 = (tile / tiles_x) * 32, x0 = (tile - (tile / tiles_x) * tiles_x) * 32;
    const int TH = min(32, H - y0), TW = min(32, W - x0);
    const float* __restrict__ big = (BIG_IS_X ? d.X + (int64_t)b * d.x_bstride : d.dY + (int64_t)b * d.dy_bstride);
    const float* __restrict__ sml = (BIG_IS_X ? d.dY + (int64_t)b * d.dy_bstride : d.X + (int64_t)b * d.x_bstride);
    for (int i = threadIdx.x; i < CS * PHW; i += 256) {
        const int cs = i / PHW, rem = i - cs * PHW;
        const int yy = y0 + rem / PW - 1, xx = x0 + rem - (rem / PW) * PW - 1;
        const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        S[i] = ok ? sml[(int64_t)cs * HW + yy * W + xx] : 0.f;
    }
    __syncthreads();
    const int x = threadIdx.x & 31, slot = threadIdx.x >> 5;
    const int Ncols = d.C * 9;
    float* __restrict__ part = d.ws + ((int64_t)b * tiles + tile) * d.M * Ncols;
    // the four big channels of a 32-lane group share every read of the small operand's patch: 27 LDS reads feed 4 x 27 FMAs per pixel
    // (one channel at a time, the kernel was LDS-issue bound: 104 us for a 15 us read of the big operand)
    float acc[4][MAXS][9];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int cs = 0; cs < MAXS; ++cs)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[k][cs][t] = 0.f;
    if (x < TW) {
        const float* __restrict__ src[4];
        bool live[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cb = cb0 + slot + 8 * k;                   // uniform per 32-lane group
            live[k] = cb < CB;
            src[k] = big + (int64_t)(live[k] ? cb : 0) * HW + (int64_t)y0 * W + x0 + x;
        }
        for (int y = 0; y < TH; ++y) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = src[k][y * W];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = live[k] ? v[k] : 0.f;
#pragma unroll
            for (int cs = 0; cs < MAXS; ++cs) {
                if (cs < CS) {
                    const float* __restrict__ sp = S + cs * PHW + (y + 1) * PW + (x + 1);
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int r = t / 3, sx = t - 3 * r;
                        const int dy = BIG_IS_X ? 1 - r : r - 1, dx = BIG_IS_X ? 1 - sx : sx - 1;
                        const float sv = sp[dy * PW + dx];
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k][cs][t] = fmaf(v[k], sv, acc[k][cs][t]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int cb = cb0 + slot + 8 * k;
        if (cb >= CB) continue;
#pragma unroll
        for (int cs = 0; cs < MAXS; ++cs) {
            if (cs < CS) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    float v = acc[k][cs][t];
#pragma unroll
                    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 32);
                    if (x == 0) {
                        const int m = BIG_IS_X ? cs : cb, c = BIG_IS_X ? cb : cs;
                        part[(int64_t)m * Ncols + c * 9 + t] = v;
                    }
                }
            }
        }
    }
}

static bool wgrad_small_eligible(const vd_wgrad_desc& d) {
    if (d.T != 9 || d.mode != VD_B_CONV3 || d.tile != 0 || d.math != 0) return false;
    if (d.H != d.OH || d.W != d.OW || d.OH * d.OW != d.NP) return false;
    const bool small_m = d.M <= 4 && d.C >= 32, small_c = d.C <= 4 && d.M >= 32;
    return (small_m || small_c) && d.nb >= 1;
}
static int wgrad_small_parts(const vd_wgrad_desc& d) { return d.nb * vd_cdiv(d.OW, 32) * vd_cdiv(d.OH, 32); }

// Many column sums in ONE launch (the ~180 bias / GroupNorm-parameter gradient reductions of a backward pass): workgroup i
// reads its job from a device table {ws address, out address, columns (<= 64), ld} and does exactly what colsum_kernel does
// for one 64-column chunk (same summation order: results are bit-identical to the separate launches); always accumulates.
__global__ __launch_bounds__(256) void colsum_seg_kernel(const int64_t* __restrict__ table, int B) {
    __shared__ float part[4][64];
    const int64_t* __restrict__ t = table + 4 * (int64_t)blockIdx.x;
    const float* __restrict__ ws = reinterpret_cast<const float*>(t[0]);
    float* __restrict__ out = reinterpret_cast<float*>(t[1]);
    const int C = (int)t[2];
    const int64_t ld = t[3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane;
    float s = 0.f;
    if (c < C) {
        int b = w;
        for (; b + 28 < B; b += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ws[(int64_t)(b + 4 * u) * ld + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < B; b += 4) s += ws[(int64_t)b * ld + c];
    }
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < C) out[c] += (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

template <int WM, int WN>
int launch_gemm_t(const vd_gemm_desc& d, hipStream_t st) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    const int grid = vd_cdiv(d.M, BM) * vd_cdiv(d.N, BN);
#define VD_GEMM_CASE(AM, BMD)                                                                   \
    if (d.a_mode == AM && d.b_mode == BMD) {                                                    \
        hipLaunchKernelGGL((gemm_kernel<WM, WN, AM, BMD>), dim3(grid), dim3(NT), 0, st, d);     \
        return 0;                                                                               \
    }
    VD_GEMM_CASE(VD_A_ROW, VD_B_PLAIN)
    VD_GEMM_CASE(VD_A_ROW, VD_B_KCONTIG)
    VD_GEMM_CASE(VD_A_ROW, VD_B_CONV3)
    VD_GEMM_CASE(VD_A_ROW, VD_B_CONV3_T)
    VD_GEMM_CASE(VD_A_ROW, VD_B_CONV3_S2)
    VD_GEMM_CASE(VD_A_ROW, VD_B_CONV3_UP)
    VD_GEMM_CASE(VD_A_ROW, VD_B_CONV3_DIL)
    VD_GEMM_CASE(VD_A_ROW, VD_B_CONVG)
    VD_GEMM_CASE(VD_A_COL, VD_B_PLAIN)
    VD_GEMM_CASE(VD_A_COL, VD_B_KCONTIG)
#undef VD_GEMM_CASE
    vd_set_error("vd_gemm: unsupported (a_mode=%d, b_mode=%d)", d.a_mode, d.b_mode);
    return VD_EINVAL;
}

// Largest tile that still fills the 256 CUs (a launch needs >> 256 workgroups, cdna guide G11).
int pick_tile(int M, int N, int max_bn) {
    auto wgs = [&](int bm, int bn) { return (int64_t)vd_cdiv(M, bm) * vd_cdiv(N, bn); };
    if (max_bn >= 128 && M > 64 && wgs(128, 128) >= 384) return 1;
    if (max_bn >= 128 && wgs(64, 128) >= 256) return 2;
    if (max_bn >= 128 && M <= 64 && wgs(64, 128) >= 128) return 2;
    return 3;
}

template <int WM, int WN>
int launch_wgrad_t(const vd_wgrad_desc& d, int splits, int kk_per, hipStream_t st) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    dim3 grid(vd_cdiv(d.M, BM) * vd_cdiv(d.C * d.T, BN), splits);
#define VD_WG_CASE(BMD)                                                                          \
    if (d.mode == BMD) {                                                                         \
        hipLaunchKernelGGL((wgrad_kernel<WM, WN, BMD>), grid, dim3(NT), 0, st, d, kk_per);       \
        return 0;                                                                                \
    }
    VD_WG_CASE(VD_B_PLAIN)
    VD_WG_CASE(VD_B_CONV3)
    VD_WG_CASE(VD_B_CONV3_S2)
    VD_WG_CASE(VD_B_CONV3_UP)
#undef VD_WG_CASE
    vd_set_error("vd_conv_wgrad: unsupported mode %d", d.mode);
    return VD_EINVAL;
}

}  // namespace

#ifdef VD_WG_STAMPS
extern "C" int vd_wg_stamps_set(void* buf) {          // diagnostic build only: per-wave stamp buffer of wgrad_bx3_body ([blocks][4 waves][12] u64), or NULL
    unsigned long long* p = (unsigned long long*)buf;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wg_stamps), &p, sizeof(p));
}
#endif

extern "C" int64_t vd_gemm_ws_floats(const vd_gemm_desc* desc) {
    if (desc && desc->a_packed && !bx3_eligible(*desc) && gemm_bx3_eligible(*desc) && desc->math != 2) {
        if (vd_gemm_tile(desc) != 9) return 0;
        int splits, st_per;
        gemm_bx3_plan(*desc, splits, st_per);
        return splits > 1 ? (int64_t)splits * desc->M * desc->N : 0;
    }
    if (desc && desc->a_packed) {
        if (!bx3_eligible(*desc)) return 0;
        if (vd_conv3_sm_eligible(*desc)) return 0;               // whole K per workgroup: no slabs
        int splits, c_per;
        bx3_plan(*desc, splits, c_per);
        return splits > 1 ? (int64_t)splits * desc->M * desc->N : 0;
    }
    if (!desc || !patch_eligible(*desc)) return 0;
    int splits, ks_per;
    patch_plan(*desc, splits, ks_per);
    return splits > 1 ? (int64_t)splits * desc->M * desc->N : 0;
}

extern "C" int vd_gemm_tile(const vd_gemm_desc* desc) {
    if (!desc) return 0;
    const vd_gemm_desc& d = *desc;
    if (d.a_packed && d.math == 2) {      // opt-in f16 operands (vd_conv3_pack_weights_f16_multi): only the persistent 16x16x32 kernels read them
        if (bx3_eligible(d) && !bx3_big_split(d) && k32p_pick(d)) return 18;
        if (gemm_bx3_eligible(d) && vd_gemm1x1_k32p_pick(d)) return 19;
        return -1;
    }
    if (d.a_packed) {
        if (bx3_eligible(d)) {
            int splits, c_per;
            bx3_plan(d, splits, c_per);
            static const int big_off = getenv("VD_BX3_BIG_OFF") ? atoi(getenv("VD_BX3_BIG_OFF")) : 0;
            if (vd_conv3_sm_eligible(d)) return 20;                    // 20: conv3_sm_kernel (8x8 / 4x4 levels: 64-channel x 2 | 4-image tiles, whole K, no split)
            if (bx3_big_split(d)) return 16;                           // 16: 8x8 layers, 128 x 256 tiles with the channel loop split
            if (k32p_pick(d)) return 18;                               // 18: conv3_k32p_kernel (persistent 16x16x32 kernel, any image of 8 x 32 segments)
            const int big = (big_off || d.b_mode == VD_B_CONV3_S2) ? 0 : bx3_big_tile(d, splits);
            constexpr int keep_huge = 0;
            constexpr int k32_up32 = 0;
            if (big >= 1 && !(big == 2 && (keep_huge || (d.b_mode == VD_B_CONV3_UP && !k32_up32))) && conv3_k32_eligible(d)) return 17;      // 17: conv3_k32_kernel (16x16x32 MFMA, 128 x 256 tile)
            return big == 2 ? 15 : (big == 1 ? 12 : 8);                // 12 / 15: the 128 x 256 / 128 x 512 tile, eight waves
        }
        if (!gemm_bx3_eligible(d)) return -1;
        if (vd_gemm1x1_k32p_pick(d)) return 19;                 // 19: gemm1x1_k32p_kernel (persistent 16x16x32 kernel, 128 x 256 tiles, LDS-DMA weights)
        // >= 2 tiles per resident workgroup (512 slots): the persistent variant walks them with the next tile's loads in flight
        static const int gbig_off = getenv("VD_GEMM_BX3_BIG_OFF") ? atoi(getenv("VD_GEMM_BX3_BIG_OFF")) : 0;
        if (!gbig_off && gemm_bx3_big_tile(d)) return 13;       // 13: the 128 x 256 tile, eight waves
        constexpr int persist = 1;
        return (persist && vd_cdiv(d.M, 128) * (d.N / 128) >= 1024) ? 11 : 9;
    }
    if (d.math == 1) return gemm_bx3_act_eligible(d) ? 10 : -1;
    if (smallm_eligible(d)) return 7;                        // direct convolution for <= 4 output channels
    if (patch_eligible(d)) {
        int splits, ks_per;
        patch_plan(d, splits, ks_per);
        return (splits == 1 && patch_wide(d)) ? 6 : 4;      // 6: the 128 x 256 tile variant of the patch kernel
    }
    if (plain_eligible(d)) return 5;
    int max_bn = 128;
    if (d.a_bstride != 0 && d.NP % 128 != 0) max_bn = 64;
    int tile = d.tile ? d.tile : pick_tile(d.M, d.N, max_bn);
    if (max_bn < 128 && tile != 3) tile = 3;
    return tile;
}

extern "C" int vd_gemm(const vd_gemm_desc* desc, void* stream) {
    VD_REQUIRE(desc != nullptr, "vd_gemm: null desc");
    vd_gemm_desc d = *desc;
    VD_REQUIRE(d.A && d.B && d.D, "vd_gemm: null operand");
    VD_REQUIRE(d.b_presplit == 0 || (d.b_presplit == 1 && vd_gemm_tile(&d) == 18),
               "vd_gemm: a pre-split B operand (b_presplit = %d) is read by the persistent 16x16x32 convolution only (vd_gemm_tile() == 18; this problem: %d)",
               d.b_presplit, d.b_presplit == 1 ? vd_gemm_tile(&d) : 0);
#ifndef VD_ABLATION
    VD_REQUIRE(d.debug == 0, "vd_gemm: desc.debug = %d -- timing-only ablation bits exist in `make ABLATION=1` builds only", d.debug);
#endif
    VD_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0 && d.NP > 0 && d.N % d.NP == 0, "vd_gemm: bad dims M=%d N=%d K=%d NP=%d", d.M,
               d.N, d.K, d.NP);
    if (d.b_mode >= VD_B_CONV3) {
        VD_REQUIRE(d.C > 0 && d.H > 0 && d.W > 0 && d.OH * d.OW == d.NP, "vd_gemm: bad conv dims");
        if (d.b_mode == VD_B_CONVG)
            VD_REQUIRE(d.kh > 0 && d.kw > 0 && d.K == d.C * d.kh * d.kw && d.conv_stride >= 1 && d.conv_stride <= 4 && d.pad_h >= 0 && d.pad_w >= 0 &&
                           d.OH == (d.H + 2 * d.pad_h - d.kh) / d.conv_stride + 1 && d.OW == (d.W + 2 * d.pad_w - d.kw) / d.conv_stride + 1 &&
                           !d.a_packed && d.math == 0 && d.a_mode == VD_A_ROW,
                       "vd_gemm: VD_B_CONVG needs kh, kw > 0, K = C*kh*kw, conv_stride 1 .. 4, OH / OW = (H + 2 pad - k) / stride + 1, row-major exact-f32 A");
        else
            VD_REQUIRE(d.K == d.C * 9, "vd_gemm: conv K must be C*9");
    }
    VD_REQUIRE(d.act == 0 || (d.act == 1 && !d.a_packed && d.math == 0), "vd_gemm: act = 1 (ReLU) is honoured by the exact-f32 kernels only");
    if (false) {
    }
    if (d.a_bstride != 0) VD_REQUIRE(d.NP % 64 == 0, "vd_gemm: per-batch A needs NP %% 64 == 0 (NP=%d)", d.NP);
    VD_REQUIRE(!(d.d_trans && (d.residual || d.rowadd)), "vd_gemm: d_trans excludes residual/rowadd");
    if (d.nb2 > 1)
        VD_REQUIRE(!d.residual && !d.rowadd && !d.d_trans && d.b_mode <= VD_B_KCONTIG && (d.N / d.NP) % d.nb2 == 0,
                   "vd_gemm: two-level batch (nb2=%d) needs plain operands, no residual/rowadd, nb %% nb2 == 0", d.nb2);
    const int tile = vd_gemm_tile(&d);
    VD_REQUIRE(tile != -1, "vd_gemm: a_packed (split-precision bf16) needs a 3x3 convolution with 8x8 / 16x16 / 32x32 outputs, "
                           "C %% 16 == 0, M >= 64, or a VD_B_PLAIN product with shared A, NP %% 128 == 0, K %% 16 == 0, M >= 64; "
                           "a_packed_mpad = M rounded up to 128; math = 1 needs per-batch A, PLAIN / KCONTIG B, NP %% 128 == 0, K %% 16 == 0, K >= 32, M >= 64");
    VD_REQUIRE(!d.gn_ss || tile == 4 || tile == 6 || tile == 8 || tile == 12 || tile == 15 || tile == 17 || tile == 18,
               "vd_gemm: gn_ss (GroupNorm folded into the loader) needs the patch-staged 3x3 kernel (OW 16/32, C %% 8 == 0, M >= 64)");
    VD_REQUIRE(!d.pool2 || tile == 8 || tile == 12 || tile == 17 || tile == 18, "vd_gemm: pool2 needs the split-precision 3x3 kernel (VD_B_CONV3_T with a_packed)");
    VD_REQUIRE(!d.act_out || (tile == 18 && d.gn_ss), "vd_gemm: act_out is written by the persistent 16x16x32 3x3 convolution with gn_ss only (vd_gemm_tile() == 18)");
    VD_REQUIRE(!d.gn_part || ((tile == 17 || tile == 18) && !d.pool2),
               "vd_gemm: gn_part is written by the 16x16x32 split-precision 3x3 kernels only (vd_gemm_tile() == 17 / 18)");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (tile) {
        case 1: rc = launch_gemm_t<2, 2>(d, st); break;
        case 2: rc = launch_gemm_t<1, 2>(d, st); break;
        case 3: rc = launch_gemm_t<1, 1>(d, st); break;
        case 4:
        case 6: rc = launch_patch(d, st); break;
        case 7: rc = launch_smallm(d, st); break;
        case 8: case 12: case 15: case 16: case 17: case 18: rc = launch_bx3(d, st); break;
        case 19: rc = vd_launch_gemm1x1_k32p(d, st) == 0 ? 0 : VD_EINVAL; break;
        case 20: rc = vd_launch_conv3_sm(d, st) == 0 ? 0 : VD_EINVAL; break;
        case 10: launch_gemm_bx3_act(d, st); rc = 0; break;
        case 13:
            hipLaunchKernelGGL(gemm_bx3_kernel<512>, dim3(vd_cdiv(d.M, 128) * (d.N / 256)), dim3(512), 0, st, d, 1 << 30);
            rc = 0;
            break;
        case 9: {
            int splits, st_per;
            gemm_bx3_plan(d, splits, st_per);
            VD_REQUIRE(splits == 1 || d.ws, "vd_gemm: split-K 1x1 launch needs the workspace (vd_gemm_ws_floats)");
            hipLaunchKernelGGL(gemm_bx3_kernel<256>, dim3(vd_cdiv(d.M, 128) * (d.N / 128), splits), dim3(NT), 0, st, d, st_per);
            if (splits > 1) launch_splitk_epilogue(d, splits, st);
            rc = 0;
            break;
        }
        case 11:
            hipLaunchKernelGGL(gemm_bx3_persist_kernel, dim3(512), dim3(NT), 0, st, d, vd_cdiv(d.M, 128) * (d.N / 128));
            rc = 0;
            break;
        case 5: {
            const int grid = vd_cdiv(d.M, 128) * (d.N / 128);
            if (d.a_mode == VD_A_ROW)
                hipLaunchKernelGGL((gemm_plain_kernel<VD_A_ROW>), dim3(grid), dim3(NT), 0, st, d);
            else
                hipLaunchKernelGGL((gemm_plain_kernel<VD_A_COL>), dim3(grid), dim3(NT), 0, st, d);
            rc = 0;
            break;
        }
        default: vd_set_error("vd_gemm: bad tile %d", tile); return VD_EINVAL;
    }
    if (rc) return rc;
    VD_LAUNCH_CHECK("vd_gemm");
    return 0;
}

extern "C" int64_t vd_conv3_packed_bytes(int M, int C, int taps) {
    if (M <= 0 || C <= 0 || C % XC != 0 || (taps != 9 && taps != 1)) return 0;
    return (int64_t)((M + 127) / 128 * 128) * C * taps * 4;
}

extern "C" int vd_conv3_pack_weights(const float* W, void* packed, int M, int C, int taps, int64_t row_stride, int64_t chan_stride,
                                     void* stream) {
    VD_REQUIRE(W && packed && M > 0 && C > 0 && C % XC == 0 && (taps == 9 || taps == 1),
               "vd_conv3_pack_weights: bad arguments (M=%d C=%d taps=%d; C %% 16 == 0, taps 9 or 1)", M, C, taps);
    VD_REQUIRE((((uintptr_t)packed) & 15) == 0, "vd_conv3_pack_weights: packed must be 16-byte aligned");
    const int Mpad = (M + 127) / 128 * 128;
    const int total = (C / XC) * 2 * Mpad;
    hipLaunchKernelGGL(conv3_pack_kernel, dim3(vd_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, W,
                       reinterpret_cast<u32x4*>(packed), M, C, Mpad, row_stride, chan_stride, taps);
    VD_LAUNCH_CHECK("vd_conv3_pack_weights");
    return 0;
}

extern "C" int vd_conv3_pack_weights_multi(const int64_t* table, int n_jobs, int64_t total_blocks, void* stream) {
    VD_REQUIRE(table && n_jobs > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "vd_conv3_pack_weights_multi: bad arguments");
    hipLaunchKernelGGL(conv3_pack_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, table, n_jobs);
    VD_LAUNCH_CHECK("vd_conv3_pack_weights_multi");
    return 0;
}

static void wgrad_plan(const vd_wgrad_desc& d, int& tile, int& splits, int& kk_per) {
    if (wgrad_small_eligible(d)) {          // direct kernel: one partial image of dW per batch item
        tile = 6;
        splits = wgrad_small_parts(d) > 1 ? wgrad_small_parts(d) : 2;       // one partial dW per (image, 32 x 32 tile); >= 2: workspace always requested
        kk_per = 1;
        return;
    }
    if (wgrad1x1_bx3_eligible(d)) {
        tile = 5;
        wgrad1x1_bx3_plan(d, splits, kk_per);
        return;
    }
    if (wgrad_patch_eligible(d)) {
        tile = 4;
        wgrad_patch_plan(d, splits, kk_per);
        return;
    }
    const int Ncols = d.C * d.T, Ktot = d.nb * d.NP;
    tile = d.tile;
    if (!tile) {
        tile = (d.M > 64 && Ncols > 64) ? 1 : 3;
        // few 128x128 tiles would need many K-splits to fill the chip, and every split writes a full slab: prefer the
        // 64x64 tile (4x the tiles, 1/4 of the splits and of the slab traffic) for small weight matrices
        if (tile == 1 && vd_cdiv(d.M, 128) * vd_cdiv(Ncols, 128) < 32) tile = 3;
    }
    const int bm = tile == 1 ? 128 : 64, bn = tile == 3 ? 64 : 128;
    const int tiles = vd_cdiv(d.M, bm) * vd_cdiv(Ncols, bn);
    splits = d.splits;
    if (splits <= 0) {  // one full wave of 3 workgroups per CU (768 slots; measured: 528 or 576 WGs leave a half-empty tail)
        constexpr int target = 768;
        splits = tiles >= target ? 1 : target / tiles;
        const int max_splits = Ktot / (BK * 8) > 0 ? Ktot / (BK * 8) : 1;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    kk_per = vd_cdiv(Ktot, splits);
    kk_per = ((kk_per + BK - 1) / BK) * BK;
    splits = vd_cdiv(Ktot, kk_per);
}

extern "C" int vd_conv_wgrad_plan(const vd_wgrad_desc* desc, int* tile, int* splits) {
    if (!desc || !tile || !splits) return VD_EINVAL;
    int kk_per;
    wgrad_plan(*desc, *tile, *splits, kk_per);
    return 0;
}

extern "C" int vd_conv_wgrad(const vd_wgrad_desc* desc, void* stream) {
    VD_REQUIRE(desc != nullptr, "vd_conv_wgrad: null desc");
    vd_wgrad_desc d = *desc;
    VD_REQUIRE(d.dY && d.X && d.dW, "vd_conv_wgrad: null operand");
    VD_REQUIRE(d.presplit == 0, "vd_conv_wgrad: pre-split operands (presplit = %d) are taken by the grouped launches only (vd_conv_wgrad_group_*)", d.presplit);
    VD_REQUIRE(d.T == 9 || d.T == 1, "vd_conv_wgrad: T must be 1 or 9");
    VD_REQUIRE((d.T == 1) == (d.mode == VD_B_PLAIN), "vd_conv_wgrad: T/mode mismatch");
    VD_REQUIRE(d.NP == d.OH * d.OW && d.NP % 4 == 0 && d.OW % 4 == 0, "vd_conv_wgrad: NP/OW must be multiples of 4");
    VD_REQUIRE((d.dy_bstride & 3) == 0 && ((((uintptr_t)d.dY) & 15) == 0), "vd_conv_wgrad: dY must be 16-B aligned");
    if (d.mode == VD_B_PLAIN)
        VD_REQUIRE((d.x_bstride & 3) == 0 && ((((uintptr_t)d.X) & 15) == 0) && d.H * d.W == d.NP,
                   "vd_conv_wgrad: 1x1 X alignment");
    VD_REQUIRE(d.math == 0 || wgrad_patch_kind(d) == 4 || wgrad1x1_bx3_eligible(d),
               "vd_conv_wgrad: math = 1 (split-precision bf16) needs a stride-1 3x3 convolution with 8x8 / 16x16 / 32x32 outputs, a stride-2 one with 8x8 / 16x16 outputs or a "
               "1x1 convolution with NP %% 8 == 0; M >= 64, C >= 64, 16-byte aligned operands");
    const int Ncols = d.C * d.T;
    int tile, splits, kk_per;
    wgrad_plan(d, tile, splits, kk_per);
    VD_REQUIRE(splits == 1 || d.ws != nullptr, "vd_conv_wgrad: workspace required for %d splits", splits);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (tile) {
        case 4: {
            dim3 grid(vd_cdiv(d.M, 128) * vd_cdiv(d.C, 64) * 3, splits);
            rc = 0;
            if (wgrad_patch_kind(d) == 4) {
                const bool up = d.mode == VD_B_CONV3_UP;
#define VD_WBX3(WW)                                                                                     \
    do {                                                                                                \
        if (wgrad_k32_enabled()) {                                                                      \
            if (up) hipLaunchKernelGGL((wgrad_k32_kernel<WW, 2>), grid, dim3(NT), 0, st, d, kk_per);    \
            else hipLaunchKernelGGL((wgrad_k32_kernel<WW, 0>), grid, dim3(NT), 0, st, d, kk_per);       \
        } else if (up) hipLaunchKernelGGL((wgrad_bx3_kernel<WW, 2>), grid, dim3(NT), 0, st, d, kk_per); \
        else hipLaunchKernelGGL((wgrad_bx3_kernel<WW, 0>), grid, dim3(NT), 0, st, d, kk_per);           \
    } while (0)
                if (d.mode == VD_B_CONV3_S2) {
                    if (d.OW >= 64) hipLaunchKernelGGL((wgrad_bx3_kernel<32, 4, true>), grid, dim3(NT), 0, st, d, kk_per);
                    else if (d.OW == 32) hipLaunchKernelGGL((wgrad_bx3_kernel<32, 4>), grid, dim3(NT), 0, st, d, kk_per);
                    else if (d.OW == 16) hipLaunchKernelGGL((wgrad_bx3_kernel<16, 4>), grid, dim3(NT), 0, st, d, kk_per);
                    else hipLaunchKernelGGL((wgrad_bx3_kernel<8, 4>), grid, dim3(NT), 0, st, d, kk_per);
                } else if (d.OW >= 64 && up) hipLaunchKernelGGL((wgrad_bx3_kernel<32, 2, true>), grid, dim3(NT), 0, st, d, kk_per);
                else if (d.OW >= 64) hipLaunchKernelGGL((wgrad_bx3_kernel<32, 0, true>), grid, dim3(NT), 0, st, d, kk_per);
                else if (d.OW == 4) hipLaunchKernelGGL((wgrad_bx3_kernel<4, 0>), grid, dim3(NT), 0, st, d, kk_per);
                else if (d.OW == 32) VD_WBX3(32);
                else if (d.OW == 16) VD_WBX3(16);
                else VD_WBX3(8);
#undef VD_WBX3
            } else if (wgrad_patch_kind(d) == 3) {
                const int ohs = ilog2_exact(d.OH);
                const bool up = d.mode == VD_B_CONV3_UP;
                if (d.OW == 32) {
                    if (up) hipLaunchKernelGGL((wgrad_patch_gen_kernel<2, 2, 64>), grid, dim3(NT), 0, st, d, kk_per, ohs, 0);
                    else hipLaunchKernelGGL((wgrad_patch_gen_kernel<2, 0, 64>), grid, dim3(NT), 0, st, d, kk_per, ohs, 0);
                } else {
                    if (up) hipLaunchKernelGGL((wgrad_patch_gen_kernel<4, 2, 64>), grid, dim3(NT), 0, st, d, kk_per, ohs, 0);
                    else hipLaunchKernelGGL((wgrad_patch_gen_kernel<4, 0, 64>), grid, dim3(NT), 0, st, d, kk_per, ohs, 0);
                }
            } else if (wgrad_patch_kind(d) == 2) {
                const int ohs = ilog2_exact(d.OH), sgs = d.OW >= 32 ? ilog2_exact(d.OW / 32) : 0;
                const bool up = d.mode == VD_B_CONV3_UP;
#define VD_WPG(R_)                                                                                               \
    do {                                                                                                         \
        if (up)                                                                                                  \
            hipLaunchKernelGGL((wgrad_patch_gen_kernel<R_, 2, 32>), grid, dim3(NT), 0, st, d, kk_per, ohs, sgs);     \
        else                                                                                                     \
            hipLaunchKernelGGL((wgrad_patch_gen_kernel<R_, 0, 32>), grid, dim3(NT), 0, st, d, kk_per, ohs, sgs);     \
    } while (0)
                if (d.OW == 4)
                    VD_WPG(8);
                else if (d.OW == 8)
                    VD_WPG(4);
                else
                    VD_WPG(1);
#undef VD_WPG
            } else if (d.OW == 32 && d.mode == VD_B_CONV3)
                hipLaunchKernelGGL((wgrad_patch_kernel<32, 0>), grid, dim3(NT), 0, st, d, kk_per);
            else if (d.OW == 32)
                hipLaunchKernelGGL((wgrad_patch_kernel<32, 2>), grid, dim3(NT), 0, st, d, kk_per);
            else if (d.mode == VD_B_CONV3)
                hipLaunchKernelGGL((wgrad_patch_kernel<16, 0>), grid, dim3(NT), 0, st, d, kk_per);
            else
                hipLaunchKernelGGL((wgrad_patch_kernel<16, 2>), grid, dim3(NT), 0, st, d, kk_per);
            break;
        }
        case 6: {
            const int parts = wgrad_small_parts(d);
            const dim3 grid(parts * vd_cdiv(d.M <= 4 ? d.C : d.M, 32));
            if (d.M <= 4) hipLaunchKernelGGL((wgrad_small_kernel<true>), grid, dim3(256), 0, st, d);
            else hipLaunchKernelGGL((wgrad_small_kernel<false>), grid, dim3(256), 0, st, d);
            VD_LAUNCH_CHECK("vd_conv_wgrad/direct");
            hipLaunchKernelGGL(colsum_kernel, dim3(vd_cdiv(d.M * Ncols, 64)), dim3(256), 0, st, d.ws, d.dW, parts, d.M * Ncols,
                               (int64_t)d.M * Ncols, d.accumulate);
            VD_LAUNCH_CHECK("vd_conv_wgrad/direct-reduce");
            return 0;
        }
        case 5:
            hipLaunchKernelGGL(wgrad1x1_bx3_kernel, dim3(vd_cdiv(d.M, 128) * vd_cdiv(d.C, 128), splits), dim3(NT), 0, st, d, kk_per);
            rc = 0;
            break;
        case 1: rc = launch_wgrad_t<2, 2>(d, splits, kk_per, st); break;
        case 2: rc = launch_wgrad_t<1, 2>(d, splits, kk_per, st); break;
        case 3: rc = launch_wgrad_t<1, 1>(d, splits, kk_per, st); break;
        default: vd_set_error("vd_conv_wgrad: bad tile %d", tile); return VD_EINVAL;
    }
    if (rc) return rc;
    VD_LAUNCH_CHECK("vd_conv_wgrad");
    if (splits > 1 && tile == 4) {
        const int64_t n = (int64_t)d.M * Ncols;
        const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(slab_reduce_perm_kernel, dim3(grid), dim3(256), 0, st, d.ws, d.dW, d.M, d.C, splits, d.accumulate);
        VD_LAUNCH_CHECK("vd_conv_wgrad/reduce");
    } else if (splits > 1) {
        const int64_t n = (int64_t)d.M * Ncols;
        if ((n & 3) == 0 && ((((uintptr_t)d.dW) & 15) == 0) && ((((uintptr_t)d.ws) & 15) == 0)) {
            const int grid = (int)((n / 4 + 255) / 256 < 2048 ? (n / 4 + 255) / 256 : 2048);
            hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid), dim3(256), 0, st, d.ws, d.dW, n / 4, splits, d.accumulate);
        } else {
            const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
            hipLaunchKernelGGL(slab_reduce_scalar_kernel, dim3(grid), dim3(256), 0, st, d.ws, d.dW, n, splits, d.accumulate);
        }
        VD_LAUNCH_CHECK("vd_conv_wgrad/reduce");
    }
    return 0;
}

// ---- grouped weight gradients -----------------------------------------------------------------------------------------------
// Class of a split-precision weight gradient = the kernel instantiation it runs on; only jobs of one class share a launch.
//   3x3: 4 * W + 2 * (CONV3_UP) + (wide image)   (W = 32 / 16 / 8 / 4),   stride-2 3x3: 2000 + W (W = 16 / 8),   1x1: 1000,   0: not groupable
//   both operands PRE-SPLIT (d.presplit == 3, round 5): 3000 + 4 * W + 2 * (CONV3_UP)   (3x3 at 8x8 / 16x16 / 32x32 outputs)
static int wgrad_group_class(const vd_wgrad_desc& d) {
    if (d.math != 1 || d.splits != 0 || d.tile != 0) return 0;
    if (d.presplit != 0) {
        const bool up = d.mode == VD_B_CONV3_UP;
        if (d.presplit != 3 || d.T != 9 || (d.mode != VD_B_CONV3 && !up) || (d.OW != 8 && d.OW != 16 && d.OW != 32) || d.OH != d.OW ||
            d.H * (up ? 2 : 1) != d.OH || d.W * (up ? 2 : 1) != d.OW ||
            (d.M & 7) || (d.C & 7) || d.M < 64 || d.C < 64 || (d.dy_bstride & 3) || (d.x_bstride & 3) || ((((uintptr_t)d.dY) | ((uintptr_t)d.X)) & 15) ||
            (int64_t)d.M * d.NP * 4 >= (1ll << 32) || (int64_t)d.C * d.NP * 4 >= (1ll << 32))
            return 0;
        return 3000 + 4 * d.OW + (up ? 2 : 0);
    }
    if (wgrad1x1_bx3_eligible(d)) return 1000;
    if (d.T != 9 || wgrad_patch_kind(d) != 4) return 0;
    if (d.mode == VD_B_CONV3_S2) return d.OW >= 64 ? 2000 + 33 : 2000 + d.OW;       // stride 2: 2000 + W (8 / 16 / 32), 2033 = 32-pixel segments of wide outputs
    const int up = d.mode == VD_B_CONV3_UP ? 2 : 0;
    if (d.OW >= 64) return 4 * 32 + up + 1;
    const int cls = 4 * d.OW + up;
    // only the classes vd_conv_wgrad_group_launch has a kernel for (4x4 through the fused upsample -- a 2x2 input -- has none: single-layer path)
    switch (cls) {
        case 4 * 32 + 0: case 4 * 32 + 2: case 4 * 16 + 0: case 4 * 16 + 2: case 4 * 8 + 0: case 4 * 8 + 2: case 4 * 4 + 0: return cls;
        default: return 0;
    }
}

extern "C" int vd_conv_wgrad_group_class(const vd_wgrad_desc* desc) { return desc ? wgrad_group_class(*desc) : 0; }
extern "C" int64_t vd_conv_wgrad_group_job_bytes(void) { return (int64_t)sizeof(vd_wgrad_job); }
// Which kernel vd_conv_wgrad_group_launch runs for a class: 9 = wgrad9_group_kernel (all nine taps per workgroup), 32 = wgrad_k32_group_kernel
// (16x16x32 one-tap-row kernel, the default where it applies), 0 = wgrad_bx3_group_kernel / wgrad1x1_bx3_group_kernel (profiling names, tests).
extern "C" int vd_conv_wgrad_group_variant(int cls) {
    if (cls >= 3000) return 3000;                   // wgrad_ps_group_kernel (pre-split operands)
    if (wgrad9_class(cls)) return 9;
    if (cls == 1000) return wgrad1x1_wide_enabled() ? 256 : 0;
    if (cls < 1000 && cls != 4 * 4 + 0 && !(cls & 1) && wgrad_k32_enabled()) return 32;
    return 0;
}

// Plan: fills the HOST image of the device job table (n * vd_conv_wgrad_group_job_bytes() bytes) with ws OFFSETS (floats) in d.ws,
// so the image does not depend on where the workspace lives; vd_conv_wgrad_group_launch adds the base.  Every workgroup of the
// grid gets about the same number of K-steps: sum_j tiles_j * ksteps_j / target.
extern "C" int vd_conv_wgrad_group_plan(const vd_wgrad_desc* descs, int n, void* table_out, int64_t* ws_floats, int* blocks, int* rblocks) {
    VD_REQUIRE(descs && n > 0 && table_out && ws_floats && blocks && rblocks, "vd_conv_wgrad_group_plan: bad args");
    const int cls = wgrad_group_class(descs[0]);
    VD_REQUIRE(cls != 0, "vd_conv_wgrad_group_plan: job 0 is not a split-precision (math = 1) 3x3 / 1x1 weight gradient");
    vd_wgrad_job* jobs = reinterpret_cast<vd_wgrad_job*>(table_out);
    const bool one = cls == 1000;
    const bool nine = cls < 3000 && wgrad9_class(cls);           // one workgroup (512 threads, one per CU) per tile produces all nine taps
    const bool wide1 = one && wgrad1x1_wide_enabled();           // 1x1: BM x 256 tiles, 32-pixel K-steps, one 512-thread workgroup per CU
    auto job_base = [&](const vd_wgrad_desc& d) -> int64_t {
        if (wide1) return (int64_t)vd_cdiv(d.M, wgrad1x1_wide_bm(d)) * vd_cdiv(d.C, 256);
        return one ? (int64_t)vd_cdiv(d.M, 128) * vd_cdiv(d.C, 128) : (int64_t)vd_cdiv(d.M, 128) * vd_cdiv(d.C, 64) * (nine ? 1 : 3);
    };
    auto job_ks = [&](const vd_wgrad_desc& d) -> int64_t {
        if (wide1) return ((int64_t)d.nb * (d.NP >> 3) + 3) >> 2;
        return one ? (((int64_t)d.nb * (d.NP >> 3) + 7) >> 3) : (((int64_t)d.nb * d.NP + 31) / 32);
    };
    int64_t work = 0;
    for (int j = 0; j < n; ++j) {
        const vd_wgrad_desc& d = descs[j];
        VD_REQUIRE(d.dY && d.X && d.dW && wgrad_group_class(d) == cls, "vd_conv_wgrad_group_plan: job %d is of another kernel class (%d vs %d)", j,
                   wgrad_group_class(d), cls);
        work += job_base(d) * job_ks(d);
    }
    // K-steps per workgroup: enough workgroups to fill the chip (768 / 512 slots), but no K range longer than `cap` steps: the tiles
    // that share a K range (3 tap rows x C/64 for dY, x M/128 for X) run side by side on one XCD and re-read it through that XCD's
    // L2 only while they stay within a few hundred K-steps of each other.  Measured on MI355X (config #2, B = 128): 3x3 at 32x32
    // 1.93 -> 1.46 ms and at 16x16 1.71 -> 1.29 ms going from ~650 to 128 steps (32 pixels each) per workgroup; 1x1 (64-pixel steps) 1.0 -> 0.76 ms
    // at 32 steps; still shorter ranges (more slabs) lose again.
    static const int t3 = getenv("VD_WGRAD_GROUP_TARGET") ? atoi(getenv("VD_WGRAD_GROUP_TARGET")) : 768;
    static const int t1 = getenv("VD_W1X1_GROUP_TARGET") ? atoi(getenv("VD_W1X1_GROUP_TARGET")) : 512;
    static const int cap3 = getenv("VD_WGRAD_GROUP_KCAP") ? atoi(getenv("VD_WGRAD_GROUP_KCAP")) : 128;
    static const int cap1 = getenv("VD_W1X1_GROUP_KCAP") ? atoi(getenv("VD_W1X1_GROUP_KCAP")) : 32;
    static const int t9 = getenv("VD_WGRAD9_TARGET") ? atoi(getenv("VD_WGRAD9_TARGET")) : 256;
    static const int cap9 = getenv("VD_WGRAD9_KCAP") ? atoi(getenv("VD_WGRAD9_KCAP")) : 128;
    static const int tw = getenv("VD_W1X1_WIDE_TARGET") ? atoi(getenv("VD_W1X1_WIDE_TARGET")) : 256;
    static const int capw = getenv("VD_W1X1_WIDE_KCAP") ? atoi(getenv("VD_W1X1_WIDE_KCAP")) : 64;
    const bool k32 = cls >= 3000 || (!one && !nine && cls != 4 * 4 + 0 && !(cls & 1) && wgrad_k32_enabled());      // two workgroups per CU: 512 resident slots
    // (512 = the resident slots; 448 / 384 measured 0.15 ms per config-#2 step faster, same box, three interleaved rounds: the small classes get
    // longer K ranges and fewer slabs, the large ones are capped at 128 steps either way -- profiles/r04_wgrad_k32_target.txt)
    static const int t32 = getenv("VD_WGRAD_K32_TARGET") ? atoi(getenv("VD_WGRAD_K32_TARGET")) : 448;
    const int target = wide1 ? tw : (one ? t1 : (nine ? t9 : (k32 ? t32 : t3)));
    const int min_ks = one ? (wide1 ? 8 : 4) : 8;
    int64_t per = (work + target - 1) / target;                  // K-steps per workgroup
    const int cap = wide1 ? capw : (one ? cap1 : (nine ? cap9 : cap3));
    if (per > cap) per = cap;
    // (A round-quantised choice of the K-range length measured neutral and is gone: profiles/HISTORY.md "VD_WGRAD_QUANT".)
    if (per < min_ks) per = min_ks;
    int64_t off = 0;
    int blk = 0, rblk = 0;
    for (int j = 0; j < n; ++j) {
        vd_wgrad_job& jb = jobs[j];
        memset(&jb, 0, sizeof(jb));
        jb.d = descs[j];
        const vd_wgrad_desc& d = descs[j];
        const int base = (int)job_base(d);
        const int ks = (int)job_ks(d);
        int splits = (int)((ks + per - 1) / per);
        if (splits < 1) splits = 1;
        const int ks_per = vd_cdiv(ks, splits);
        splits = vd_cdiv(ks, ks_per);
        jb.gx = base;
        jb.gy = splits;
        jb.ks_per = ks_per;
        jb.first_block = blk;
        blk += (base * splits + 7) / 8 * 8;
        const int64_t nel = (int64_t)d.M * d.C * d.T;
        jb.first_rblock = rblk;
        jb.rblocks = 0;
        jb.d.ws = reinterpret_cast<float*>((uintptr_t)0);
        if (splits > 1) {
            jb.rblocks = (int)((nel + 255) / 256 < 1024 ? (nel + 255) / 256 : 1024);
            jb.d.ws = reinterpret_cast<float*>((uintptr_t)(off * sizeof(float)));      // offset, rebased at launch
            off += (int64_t)splits * nel;
            off = (off + 3) / 4 * 4;
        }
        rblk += jb.rblocks;
    }
    *ws_floats = off;
    *blocks = blk;
    *rblocks = rblk;
    return cls;
}

__global__ __launch_bounds__(64) void wgrad_group_rebase_kernel(vd_wgrad_job* __restrict__ jobs, int n, float* ws) {
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j < n && jobs[j].gy > 1) jobs[j].d.ws = ws + ((uintptr_t)jobs[j].d.ws) / sizeof(float);
}

// dev_table: the planned image copied to the device, ALREADY rebased onto ws (vd_conv_wgrad_group_rebase, once per upload).
extern "C" int vd_conv_wgrad_group_rebase(void* dev_table, int n, float* ws, void* stream) {
    VD_REQUIRE(dev_table && n > 0, "vd_conv_wgrad_group_rebase: bad args");
    hipLaunchKernelGGL(wgrad_group_rebase_kernel, dim3(vd_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream,
                       reinterpret_cast<vd_wgrad_job*>(dev_table), n, ws);
    VD_LAUNCH_CHECK("vd_conv_wgrad_group_rebase");
    return 0;
}

extern "C" int vd_conv_wgrad_group_launch(const void* dev_table, int n, int cls, int blocks, int rblocks, void* stream) {
    VD_REQUIRE(dev_table && n > 0 && blocks > 0 && rblocks >= 0, "vd_conv_wgrad_group_launch: bad args");
    const vd_wgrad_job* jobs = reinterpret_cast<const vd_wgrad_job*>(dev_table);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(blocks);
    switch (cls) {
        case 1000:
            if (wgrad1x1_wide_enabled()) {
                static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad1x1_wide_group_kernel),
                                                                   hipFuncAttributeMaxDynamicSharedMemorySize, W1X1_WIDE_LDS);
                VD_REQUIRE(attr == hipSuccess, "vd_conv_wgrad_group_launch: cannot reserve %d bytes of LDS", W1X1_WIDE_LDS);
                hipLaunchKernelGGL(wgrad1x1_wide_group_kernel, grid, dim3(512), W1X1_WIDE_LDS, st, jobs, n);
            } else {
                hipLaunchKernelGGL(wgrad1x1_bx3_group_kernel, grid, dim3(NT), 0, st, jobs, n);
            }
            break;
#define VD_WG_K32(WW, MD)                                                                                   \
    if (wgrad_k32_enabled()) hipLaunchKernelGGL((wgrad_k32_group_kernel<WW, MD>), grid, dim3(NT), 0, st, jobs, n); \
    else hipLaunchKernelGGL((wgrad_bx3_group_kernel<WW, MD>), grid, dim3(NT), 0, st, jobs, n);
        case 4 * 32 + 0:
            if (wgrad9_class(cls)) hipLaunchKernelGGL((wgrad9_group_kernel<32>), grid, dim3(512), 0, st, jobs, n, wgrad9_flags());
            else { VD_WG_K32(32, 0) }
            break;
        case 4 * 32 + 2: VD_WG_K32(32, 2) break;
        case 4 * 32 + 1: hipLaunchKernelGGL((wgrad_bx3_group_kernel<32, 0, true>), grid, dim3(NT), 0, st, jobs, n); break;
        case 4 * 32 + 3: hipLaunchKernelGGL((wgrad_bx3_group_kernel<32, 2, true>), grid, dim3(NT), 0, st, jobs, n); break;
        case 4 * 16 + 0:
            if (wgrad9_class(cls)) hipLaunchKernelGGL((wgrad9_group_kernel<16>), grid, dim3(512), 0, st, jobs, n, wgrad9_flags());
            else { VD_WG_K32(16, 0) }
            break;
        case 4 * 16 + 2: VD_WG_K32(16, 2) break;
        case 4 * 8 + 0: VD_WG_K32(8, 0) break;
        case 4 * 8 + 2: VD_WG_K32(8, 2) break;
#undef VD_WG_K32
        case 4 * 4 + 0: hipLaunchKernelGGL((wgrad_bx3_group_kernel<4, 0>), grid, dim3(NT), 0, st, jobs, n); break;
        case 3000 + 4 * 32: case 3000 + 4 * 16: case 3000 + 4 * 8: case 3000 + 4 * 32 + 2: case 3000 + 4 * 16 + 2: case 3000 + 4 * 8 + 2:
            VD_REQUIRE(vd_launch_wgrad_ps_group(dev_table, n, (cls - 3000) / 4, (cls - 3000) & 2, blocks, st) == 0,
                       "vd_conv_wgrad_group_launch: no pre-split kernel for class %d", cls);
            break;
        case 2000 + 33: hipLaunchKernelGGL((wgrad_bx3_group_kernel<32, 4, true>), grid, dim3(NT), 0, st, jobs, n); break;
        case 2000 + 32: hipLaunchKernelGGL((wgrad_bx3_group_kernel<32, 4>), grid, dim3(NT), 0, st, jobs, n); break;
        case 2000 + 16: hipLaunchKernelGGL((wgrad_bx3_group_kernel<16, 4>), grid, dim3(NT), 0, st, jobs, n); break;
        case 2000 + 8: hipLaunchKernelGGL((wgrad_bx3_group_kernel<8, 4>), grid, dim3(NT), 0, st, jobs, n); break;
        default: vd_set_error("vd_conv_wgrad_group_launch: unknown kernel class %d", cls); return VD_EINVAL;
    }
    VD_LAUNCH_CHECK("vd_conv_wgrad_group_launch");
    if (rblocks > 0) {
        hipLaunchKernelGGL(wgrad_group_reduce_kernel, dim3(rblocks), dim3(256), 0, st, jobs, n);
        VD_LAUNCH_CHECK("vd_conv_wgrad_group_launch/reduce");
    }
    return 0;
}

// Workspace floats vd_conv_wgrad needs for the given problem (0 when no split is chosen).
extern "C" int64_t vd_conv_wgrad_ws_floats(const vd_wgrad_desc* desc) {
    if (!desc) return 0;
    int tile, splits, kk_per;
    wgrad_plan(*desc, tile, splits, kk_per);
    return splits > 1 ? (int64_t)splits * desc->M * desc->C * desc->T : 0;
}

extern "C" int vd_weight_transpose(const float* W, float* Wt, int M, int C, int T, void* stream) {
    VD_REQUIRE(W && Wt && M > 0 && C > 0 && T > 0, "vd_weight_transpose: bad args");
    const int64_t total = (int64_t)M * C * T;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(wtranspose_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, Wt, M, C, T);
    VD_LAUNCH_CHECK("vd_weight_transpose");
    return 0;
}

extern "C" int vd_sumpool2x2(const float* dU, float* dX, int B, int C, int H, int W, int64_t du_bstride, int64_t dx_bstride,
                             int accumulate, void* stream) {
    VD_REQUIRE(dU && dX && B > 0 && C > 0 && H > 0 && W > 0, "vd_sumpool2x2: bad args");
    const int64_t total = (int64_t)B * C * H * W;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(sumpool2x2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dU, dX, B, C, H, W, du_bstride,
                       dx_bstride, accumulate);
    VD_LAUNCH_CHECK("vd_sumpool2x2");
    return 0;
}

extern "C" int vd_col2im_s2(const float* G, float* dX, int B, int C, int H, int W, int OH, int OW, int pad, int64_t g_bstride,
                            int64_t dx_bstride, void* stream) {
    VD_REQUIRE(G && dX && B > 0 && C > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && (pad == 0 || pad == 1), "vd_col2im_s2: bad args");
    const int64_t total = (int64_t)B * C * H * W;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(col2im_s2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, G, dX, B, C, H, W, OH, OW, pad, g_bstride,
                       dx_bstride);
    VD_LAUNCH_CHECK("vd_col2im_s2");
    return 0;
}

extern "C" int vd_rowsum(const float* X, float* ws, int B, int M, int P, int64_t x_bstride, int64_t ws_ld, void* stream) {
    VD_REQUIRE(X && ws && B > 0 && M > 0 && P > 0 && ws_ld >= M, "vd_rowsum: bad args");
    if (P >= 8192 && (P & 15) == 0 && (x_bstride & 3) == 0 && ((((uintptr_t)X) & 15) == 0))
        hipLaunchKernelGGL(rowsum_long_kernel, dim3(B * M), dim3(256), 0, (hipStream_t)stream, X, ws, M, P, x_bstride, ws_ld);
    else
        hipLaunchKernelGGL(rowsum_kernel, dim3(vd_cdiv((int64_t)B * M, 4)), dim3(256), 0, (hipStream_t)stream, X, ws, B, M, P,
                           x_bstride, ws_ld);
    VD_LAUNCH_CHECK("vd_rowsum");
    return 0;
}

extern "C" int vd_colsum_segmented(const int64_t* table, int n_jobs, int B, void* stream) {
    VD_REQUIRE(table && n_jobs > 0 && B > 0, "vd_colsum_segmented: bad args");
    hipLaunchKernelGGL(colsum_seg_kernel, dim3(n_jobs), dim3(256), 0, (hipStream_t)stream, table, B);
    VD_LAUNCH_CHECK("vd_colsum_segmented");
    return 0;
}

extern "C" int vd_colsum(const float* ws, float* out, int B, int C, int64_t ld, int accumulate, void* stream) {
    VD_REQUIRE(ws && out && B > 0 && C > 0, "vd_colsum: bad args");
    hipLaunchKernelGGL(colsum_kernel, dim3(vd_cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, ws, out, B, C, ld, accumulate);
    VD_LAUNCH_CHECK("vd_colsum");
    return 0;
}
