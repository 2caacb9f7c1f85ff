// K4 flash: multi-head spatial self-attention for MORE than 256 tokens (head_dim 32; the 32x32-token level of the latent UNet,
// BASELINE config #5: 14 heads x 1024 tokens) without the N x N score matrix in HBM (round 4).
//
// Before: St = scale k^T q (vd_gemm) -> softmax over j (vd_softmax_col_fwd) -> o = v P (vd_gemm), and five more N x N passes in the
// backward: 5.2 GB of score-matrix traffic per attention block at B = 8 (9.8 ms of config #5's 36 ms step).  Here the scores of a
// (batch, head) exist only in accumulator registers, 256 rows at a time:
//   forward   online softmax over key blocks (running max m, running sum l, o rescaled by exp(m_old - m_new)); saves lse = m + log l
//   dq pass   recomputes P = exp(scale s - lse_i) per key block, dS = scale P (v^T do - delta_i), dq += k dS; writes delta_i = sum_c do o
//   dk/dv     the transposed problem: a workgroup owns 128 KEYS and walks the query blocks; lse / delta are per ROW there (LDS broadcast)
// Same register geometry as attn_core_kernel (vd_attn.hip): one workgroup per (batch, head, 128 columns), wave w owns 32 columns and all
// 256 rows of the current block (8 accumulator tiles of the 32x32x16 MFMA), the phase-1 accumulator is the phase-2 B operand as it stands.
// Every product is hi*hi + hi*lo + lo*hi over bf16 halves with f32 accumulation, the halves made where an operand is written to LDS.
// Replaces torch.baddbmm / softmax / bmm of diffusers' AttentionBlock (UNet2DModel of `LDM-CELEBA-HQ-256`, reference model.py:706-776,
// reached from loss.py:993).
#include "vd_common.h"

namespace {

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

struct flash_args {
    const float* q; const float* k; const float* v;      // [batch][heads * 32][N] slices of qkv (batch stride qkv_bs, channel stride N)
    const float* o; const float* dout;                   // backward: saved forward output / its gradient, [batch][heads * 32][N]
    float* out;                                          // forward: o
    float* lse;                                          // [batch * heads][N]   forward: written; backward: read
    float* delta;                                        // [batch * heads][N]   dq pass: written; dk/dv pass: read
    float* dq; float* dk; float* dv;                     // slices of dqkv (batch stride dqkv_bs)
    int64_t qkv_bs, o_bs, do_bs, out_bs, dqkv_bs;
    int heads, N;
    float scale;
};

constexpr int D = 32;                                    // head_dim
constexpr int RB = 256;                                  // rows per block
constexpr int JT = 8;                                    // 32-row accumulator tiles per block
constexpr int A1U = 8 * RB;                              // phase-1 A image: [chunk][part][k-octet][256 rows], units of 16 B (32 KB)
constexpr int PS = 36;                                   // phase-2 plane stride (units): 36 = 4 mod 16 spreads a quad's writes over the banks
constexpr int A2U = JT * 8 * PS;                         // phase-2 A image: [stage][part][u = k-step * 2 + lane half][32 channels] (36 KB)

// MODE 0: forward (columns = queries; rows = keys: A1 = k, A2 = v)
// MODE 1: dq     (columns = queries; rows = keys: A1 = k and v, A2 = k)
// MODE 2: dk, dv (columns = keys;    rows = queries: A1 = q and do, A2 = do and q)
template <int MODE>
__global__ __launch_bounds__(256, 1) void attn_flash_kernel(const flash_args g) {
    constexpr bool TWO = MODE != 0;                      // a second phase-1 product (X = v^T do and its transpose)
    __shared__ u32x4_t lds[A1U * (TWO ? 2 : 1) + A2U * (MODE == 2 ? 2 : 1) + 128];
    u32x4_t* const A1a = lds;
    u32x4_t* const A1b = lds + (TWO ? A1U : 0);
    u32x4_t* const A2a = lds + A1U * (TWO ? 2 : 1);
    u32x4_t* const A2b = A2a + (MODE == 2 ? A2U : 0);
    float* const rowv = reinterpret_cast<float*>(A2a + A2U * (MODE == 2 ? 2 : 1));       // 512 floats: lse / delta of the row block (MODE 2), delta partials (MODE 1)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int N = g.N;
    const int cblocks = N / 128;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;                       // the column blocks of a (batch, head) on one XCD: they walk the same rows
        if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    }
    const int bh = bid / cblocks, i0 = (bid - bh * cblocks) * 128;
    const int b = bh / g.heads, hd = bh - b * g.heads;
    const int64_t hoff = (int64_t)hd * D * N;
    const float* __restrict__ Q = g.q + (int64_t)b * g.qkv_bs + hoff;
    const float* __restrict__ K = g.k + (int64_t)b * g.qkv_bs + hoff;
    const float* __restrict__ V = g.v + (int64_t)b * g.qkv_bs + hoff;
    const float* __restrict__ DO = MODE ? g.dout + (int64_t)b * g.do_bs + hoff : nullptr;
    // column-side operands (fixed for the workgroup) and row-side operands (walked block by block)
    const float* __restrict__ B1a = (MODE == 2 ? K : Q) + i0;
    const float* __restrict__ B1b = MODE == 0 ? nullptr : (MODE == 1 ? DO : V) + i0;
    const float* __restrict__ R1a = MODE == 2 ? Q : K;
    const float* __restrict__ R1b = MODE == 0 ? nullptr : (MODE == 1 ? V : DO);
    const float* __restrict__ R2a = MODE == 0 ? V : (MODE == 1 ? K : DO);
    const float* __restrict__ R2b = MODE == 2 ? Q : nullptr;

    // ---------------------------------------------------------------- prologue: column-side fragments into registers
    // thread (column bi, octet pair bo): channels 8 bo .. 8 bo + 7 and 8 (bo + 2) .. ; image [chunk][part][k-octet][128 columns] in the A1 buffers
    const int bi = tid & 127, bo = tid >> 7;
    float dpart = 0.f;
    {
        float rb[16], rc[16], ro[16];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                rb[8 * e + c] = B1a[(int64_t)((bo + 2 * e) * 8 + c) * N + bi];
                if constexpr (TWO) rc[8 * e + c] = B1b[(int64_t)((bo + 2 * e) * 8 + c) * N + bi];
                if constexpr (MODE == 1) ro[8 * e + c] = g.o[(int64_t)b * g.o_bs + hoff + i0 + (int64_t)((bo + 2 * e) * 8 + c) * N + bi];
            }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int o = bo + 2 * e;
            float va[8], vb[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                va[c] = rb[8 * e + c];
                if constexpr (TWO) vb[c] = rc[8 * e + c];
                if constexpr (MODE == 1) dpart += rc[8 * e + c] * ro[8 * e + c];
            }
            u32x4_t hi, lo;
            split8(va, hi, lo);
            A1a[(((o >> 1) * 2 + 0) * 2 + (o & 1)) * 128 + bi] = hi;
            A1a[(((o >> 1) * 2 + 1) * 2 + (o & 1)) * 128 + bi] = lo;
            if constexpr (TWO) {
                split8(vb, hi, lo);
                A1a[1024 + (((o >> 1) * 2 + 0) * 2 + (o & 1)) * 128 + bi] = hi;
                A1a[1024 + (((o >> 1) * 2 + 1) * 2 + (o & 1)) * 128 + bi] = lo;
            }
        }
        if constexpr (MODE == 1) rowv[tid] = dpart;
    }
    __syncthreads();
    u32x4_t bah[2], bal[2], bbh[2], bbl[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const int at = h * 128 + 32 * wave + l31;
        bah[ch] = A1a[((ch * 2 + 0) * 2) * 128 + at];
        bal[ch] = A1a[((ch * 2 + 1) * 2) * 128 + at];
        if constexpr (TWO) {
            bbh[ch] = A1a[1024 + ((ch * 2 + 0) * 2) * 128 + at];
            bbl[ch] = A1a[1024 + ((ch * 2 + 1) * 2) * 128 + at];
        }
    }
    float delta_c = 0.f;                                 // MODE 1: delta of this lane's column
    if constexpr (MODE == 1) {
        delta_c = rowv[32 * wave + l31] + rowv[128 + 32 * wave + l31];
        if (h == 0) g.delta[(int64_t)bh * N + i0 + 32 * wave + l31] = delta_c;
    }
    float lse_c = 0.f;                                   // MODE 1: lse of this lane's column
    if constexpr (MODE == 1) lse_c = g.lse[(int64_t)bh * N + i0 + 32 * wave + l31];
    __syncthreads();                                     // the fragments are in registers: the A1 buffers may be overwritten

    f32x16 oa, ob;                                       // phase-2 accumulators [32 channels][32 columns]: o | dq | dv (oa), dk (ob)
#pragma unroll
    for (int v = 0; v < 16; ++v) oa[v] = 0.f, ob[v] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;                // MODE 0: online softmax state of this lane's column

    const int vu = tid & 3, vc = (tid >> 2) & 31, vst = tid >> 7;          // phase-2 item: (u, channel, stage vst + 2 it)
    const int nrb = N / RB;
    // row-block operands in registers: every thread owns row r0 + tid of the phase-1 operands (32 channels) and four phase-2 items (8 keys of
    // one channel each).  The loads of block rbk + 1 are issued before block rbk's MFMAs (one wave per SIMD: nothing else hides them).
    float ra[32], rb2[TWO ? 32 : 1];
    f32x4 rv[4][2], rw[MODE == 2 ? 4 : 1][2];
    float lse_r = 0.f, delta_r = 0.f;
    auto load_block = [&](int rbk) {
        const int r0 = rbk * RB;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            ra[c] = R1a[(int64_t)c * N + r0 + tid];
            if constexpr (TWO) rb2[c] = R1b[(int64_t)c * N + r0 + tid];
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int64_t off = (int64_t)vc * N + r0 + 32 * (vst + 2 * it) + 16 * (vu >> 1) + 4 * (vu & 1);
            rv[it][0] = *reinterpret_cast<const f32x4*>(R2a + off);
            rv[it][1] = *reinterpret_cast<const f32x4*>(R2a + off + 8);
            if constexpr (MODE == 2) {
                rw[it][0] = *reinterpret_cast<const f32x4*>(R2b + off);
                rw[it][1] = *reinterpret_cast<const f32x4*>(R2b + off + 8);
            }
        }
        if constexpr (MODE == 2) {
            lse_r = g.lse[(int64_t)bh * N + r0 + tid];
            delta_r = g.delta[(int64_t)bh * N + r0 + tid];
        }
    };
    auto store_block = [&]() {
#pragma unroll
        for (int o = 0; o < 4; ++o) {                    // octet o = chunk (o >> 1), k-octet (o & 1)
            float va[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) va[c] = ra[8 * o + c];
            u32x4_t hi, lo;
            split8(va, hi, lo);
            A1a[(((o >> 1) * 2 + 0) * 2 + (o & 1)) * RB + tid] = hi;
            A1a[(((o >> 1) * 2 + 1) * 2 + (o & 1)) * RB + tid] = lo;
            if constexpr (TWO) {
#pragma unroll
                for (int c = 0; c < 8; ++c) va[c] = rb2[8 * o + c];
                split8(va, hi, lo);
                A1b[(((o >> 1) * 2 + 0) * 2 + (o & 1)) * RB + tid] = hi;
                A1b[(((o >> 1) * 2 + 1) * 2 + (o & 1)) * RB + tid] = lo;
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int st = vst + 2 * it;
            float va[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) va[e] = rv[it][0][e], va[4 + e] = rv[it][1][e];
            u32x4_t hi, lo;
            split8(va, hi, lo);
            A2a[((st * 2 + 0) * 4 + vu) * PS + vc] = hi;
            A2a[((st * 2 + 1) * 4 + vu) * PS + vc] = lo;
            if constexpr (MODE == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) va[e] = rw[it][0][e], va[4 + e] = rw[it][1][e];
                split8(va, hi, lo);
                A2b[((st * 2 + 0) * 4 + vu) * PS + vc] = hi;
                A2b[((st * 2 + 1) * 4 + vu) * PS + vc] = lo;
            }
        }
        if constexpr (MODE == 2) {
            rowv[tid] = lse_r;
            rowv[256 + tid] = delta_r;
        }
    };
    // acc[c][col] += sum over the 32 rows of tile st of A2[c][row] T[row][col]: T's bf16 halves are made from the f32 tile as it dies
    auto phase2 = [&](f32x16& acc, const u32x4_t* __restrict__ img, const f32x16& t, int st) {
        const u32x4_t* __restrict__ Vs = img + l31;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float tv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) tv[e] = t[8 * s2 + e];
            u32x4_t thi, tlo;
            split8(tv, thi, tlo);
            const bf16x8_t bhf = __builtin_bit_cast(bf16x8_t, thi), blf = __builtin_bit_cast(bf16x8_t, tlo);
            const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, Vs[((st * 2 + 0) * 4 + s2 * 2 + h) * PS]);
            const bf16x8_t al = __builtin_bit_cast(bf16x8_t, Vs[((st * 2 + 1) * 4 + s2 * 2 + h) * PS]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhf, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blf, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhf, acc, 0, 0, 0);
        }
    };
    // one 32-row tile of a phase-1 product over the 32 channels
    auto phase1 = [&](f32x16& acc, const u32x4_t* __restrict__ img, const u32x4_t (&fh)[2], const u32x4_t (&fl)[2], int jt) {
        const u32x4_t* __restrict__ As = img + h * RB + l31 + 32 * jt;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const bf16x8_t bhf = __builtin_bit_cast(bf16x8_t, fh[ch]), blf = __builtin_bit_cast(bf16x8_t, fl[ch]);
            const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, As[((ch * 2 + 0) * 2) * RB]);
            const bf16x8_t al = __builtin_bit_cast(bf16x8_t, As[((ch * 2 + 1) * 2) * RB]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhf, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blf, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhf, acc, 0, 0, 0);
        }
    };

    load_block(0);
    for (int rbk = 0; rbk < nrb; ++rbk) {
        store_block();
        __syncthreads();
        if (rbk + 1 < nrb) load_block(rbk + 1);

        // lane (l31, h) of wave w: column 32 w + l31, rows 32 jt + (v & 3) + 8 (v >> 2) + 4 h
        if constexpr (MODE == 0) {
            // ---- phase 1 over the whole block (the softmax needs the block maximum before any exponential) ----
            f32x16 sacc[JT];
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
                for (int v = 0; v < 16; ++v) sacc[jt][v] = 0.f;
                phase1(sacc[jt], A1a, bah, bal, jt);
            }
            float bm = -INFINITY;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    sacc[jt][v] *= g.scale;
                    bm = fmaxf(bm, sacc[jt][v]);
                }
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
            const float m_new = fmaxf(m_run, bm);
            const float alpha = __expf(m_run - m_new);   // first block: exp(-inf) = 0
            float sum = 0.f;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    sacc[jt][v] = __expf(sacc[jt][v] - m_new);
                    sum += sacc[jt][v];
                }
            sum += __shfl_xor(sum, 32, 64);
            l_run = l_run * alpha + sum;
            m_run = m_new;
#pragma unroll
            for (int v = 0; v < 16; ++v) oa[v] *= alpha;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) phase2(oa, A2a, sacc[jt], jt);
        } else {
            // ---- tile by tile: S and X of 32 rows, the softmax gradient, and the phase-2 products they feed (no 256-row matrix is ever live) ----
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                f32x16 sa, xa;
#pragma unroll
                for (int v = 0; v < 16; ++v) sa[v] = 0.f, xa[v] = 0.f;
                phase1(sa, A1a, bah, bal, jt);
                phase1(xa, A1b, bbh, bbl, jt);
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const float p = __expf(g.scale * sa[v] - lse_c);
                        sa[v] = g.scale * p * (xa[v] - delta_c);                    // dS
                    }
                    phase2(oa, A2a, sa, jt);                                        // dq += k dS
                } else {
                    const f32x4* __restrict__ lr = reinterpret_cast<const f32x4*>(rowv) + h;        // rows 4 h + {0..3} of an 8-row group
                    const f32x4* __restrict__ dr = reinterpret_cast<const f32x4*>(rowv + 256) + h;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 l4 = lr[8 * jt + 2 * q4], d4 = dr[8 * jt + 2 * q4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int v = 4 * q4 + e;
                            const float p = __expf(g.scale * sa[v] - l4[e]);
                            sa[v] = p;                                              // P^T
                            xa[v] = g.scale * p * (xa[v] - d4[e]);                  // dS^T
                        }
                    }
                    phase2(oa, A2a, sa, jt);                                        // dv += do P^T
                    phase2(ob, A2b, xa, jt);                                        // dk += q dS^T
                }
            }
        }
        __syncthreads();                                 // every wave is done with this block's LDS images
    }

    // ---------------------------------------------------------------- epilogue: [32 channels][32 columns] tiles, rows (v & 3) + 8 (v >> 2) + 4 h
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned t_lane = (unsigned)(l31 + 4 * h * N);
    if constexpr (MODE == 0) {
        const float inv = 1.0f / l_run;
        float* __restrict__ Yp = g.out + (int64_t)b * g.out_bs + hoff + i0 + 32 * wave_u;
#pragma unroll
        for (int v = 0; v < 16; ++v) Yp[t_lane + (unsigned)(((v & 3) + 8 * (v >> 2)) * N)] = oa[v] * inv;
        if (g.lse && h == 0) g.lse[(int64_t)bh * N + i0 + 32 * wave_u + l31] = m_run + __logf(l_run);
    } else if constexpr (MODE == 1) {
        float* __restrict__ Yp = g.dq + (int64_t)b * g.dqkv_bs + hoff + i0 + 32 * wave_u;
#pragma unroll
        for (int v = 0; v < 16; ++v) Yp[t_lane + (unsigned)(((v & 3) + 8 * (v >> 2)) * N)] = oa[v];
    } else {
        float* __restrict__ Yv = g.dv + (int64_t)b * g.dqkv_bs + hoff + i0 + 32 * wave_u;
        float* __restrict__ Yk = g.dk + (int64_t)b * g.dqkv_bs + hoff + i0 + 32 * wave_u;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            Yv[t_lane + (unsigned)(((v & 3) + 8 * (v >> 2)) * N)] = oa[v];
            Yk[t_lane + (unsigned)(((v & 3) + 8 * (v >> 2)) * N)] = ob[v];
        }
    }
}

bool flash_shape_ok(int B, int heads, int d, int N) { return B > 0 && heads > 0 && d == D && N >= 256 && N % 256 == 0 && N <= (1 << 16); }

}  // namespace

extern "C" int vd_attn_flash_fwd(const float* qkv, float* out, float* lse, int B, int heads, int head_dim, int N, float scale,
                                 int64_t qkv_bstride, int64_t out_bstride, void* stream) {
    VD_REQUIRE(qkv && out, "vd_attn_flash_fwd: null pointer");
    VD_REQUIRE(flash_shape_ok(B, heads, head_dim, N), "vd_attn_flash_fwd: needs head_dim == 32 and a multiple of 256 tokens (B=%d heads=%d d=%d N=%d)",
               B, heads, head_dim, N);
    VD_REQUIRE(((((uintptr_t)qkv) | ((uintptr_t)out)) & 15) == 0 && (qkv_bstride & 3) == 0, "vd_attn_flash_fwd: pointers / batch stride must be 16-byte aligned");
    flash_args a{};
    const int64_t CN = (int64_t)heads * head_dim * N;
    a.q = qkv, a.k = qkv + CN, a.v = qkv + 2 * CN;
    a.out = out, a.lse = lse;
    a.qkv_bs = qkv_bstride, a.out_bs = out_bstride;
    a.heads = heads, a.N = N, a.scale = scale;
    hipLaunchKernelGGL((attn_flash_kernel<0>), dim3(B * heads * (N / 128)), dim3(256), 0, (hipStream_t)stream, a);
    VD_LAUNCH_CHECK("vd_attn_flash_fwd");
    return 0;
}

extern "C" int vd_attn_flash_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* delta, float* dqkv, int B,
                                 int heads, int head_dim, int N, float scale, int64_t qkv_bstride, int64_t out_bstride, int64_t dout_bstride,
                                 int64_t dqkv_bstride, void* stream) {
    VD_REQUIRE(qkv && out && dout && lse && delta && dqkv, "vd_attn_flash_bwd: null pointer");
    VD_REQUIRE(flash_shape_ok(B, heads, head_dim, N), "vd_attn_flash_bwd: needs head_dim == 32 and a multiple of 256 tokens (B=%d heads=%d d=%d N=%d)",
               B, heads, head_dim, N);
    VD_REQUIRE(((((uintptr_t)qkv) | ((uintptr_t)dout) | ((uintptr_t)dqkv) | ((uintptr_t)out)) & 15) == 0 && ((qkv_bstride | dout_bstride | dqkv_bstride | out_bstride) & 3) == 0,
               "vd_attn_flash_bwd: pointers / batch strides must be 16-byte aligned");
    flash_args a{};
    const int64_t CN = (int64_t)heads * head_dim * N;
    a.q = qkv, a.k = qkv + CN, a.v = qkv + 2 * CN;
    a.o = out, a.dout = dout;
    a.lse = const_cast<float*>(lse), a.delta = delta;
    a.dq = dqkv, a.dk = dqkv + CN, a.dv = dqkv + 2 * CN;
    a.qkv_bs = qkv_bstride, a.o_bs = out_bstride, a.do_bs = dout_bstride, a.dqkv_bs = dqkv_bstride;
    a.heads = heads, a.N = N, a.scale = scale;
    const dim3 grid(B * heads * (N / 128));
    hipLaunchKernelGGL((attn_flash_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, a);         // dq, delta
    hipLaunchKernelGGL((attn_flash_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, a);         // dk, dv (reads delta)
    VD_LAUNCH_CHECK("vd_attn_flash_bwd");
    return 0;
}
