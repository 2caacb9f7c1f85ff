// ---- split-precision 1x1 convolution / plain product on v_mfma_f32_16x16x32_bf16, persistent tile walk (round 4) -----------------------
//     D[b][m][p] = alpha * sum_k W[m][k] * X[b][k][p] (+ bias[m]) (+ residual) (+ D)          (VD_B_PLAIN with a_packed: 1x1 convolutions,
//     attention projections, their input gradients W^T dY through the transposed packed operand)
// The structure of vd_conv_k32p.hip with one tap and no halo: a workgroup of eight 64 x 64 waves owns 128 output channels x 256 consecutive
// pixels, one workgroup per CU walks the tiles of its XCD's range, a stage is ONE chunk pair (32 input channels):
//     As[buf][c2][part][q][128 m]   16 KB x 2   the packed weights, global -> LDS by LDS-DMA (lane-linear image: a straight copy of 8 runs of 2 KB)
//     Ps[buf][c2][part][q][272 u]   34 KB x 2   the activations, split into bf16 (hi, lo) where they are written (round 6: one 16-byte load per channel and
//                                               pixel quad; a plane is four slices of 68 units, pixel n at (n & 3) * 68 + (n >> 2))
// Both images are double-buffered, so a stage costs ONE barrier: the DMA of stage s+1 and the global loads of the activations of stage s+2 (two
// register sets) are issued before the 48 MFMAs of stage s, the conversion + LDS write of stage s+1's activations after them, then the counted
// wait for the DMA and the barrier.  Fragment
// reads are pipelined by hand (next pixel tile's pair before the current tile's 12 MFMAs).  The stage pipeline continues into the next tile.
// Why: round 3's kernels for this family (gemm_bx3_kernel / _persist, 32x32x16 MFMA, 128 x 128 tiles, register-staged weights, two barriers per
// 64-channel stage) ran at 0.19 matrix-pipe utilisation -- 69.6 us for the 256 -> 768 projection at 16x16 (B = 128) whose HBM floor is 21 us and
// whose MFMA time at the sustained split-precision rate is 19 us (profiles/r03_shape_probe.txt, r03_pmc_mfma.json).
// Reference work replaced: diffusers ResnetBlock2D.conv_shortcut and AttentionBlock query / key / value / proj_attn (reference loss.py:993).
#include "vd_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ u32x4 to_f16x8(const float (&v)[8]) {
    f16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)v[j];
    return __builtin_bit_cast(u32x4, h);
}

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

struct g32p_args {
    vd_gemm_desc d;
    int n_tiles;
    int tiles_m;
#ifdef VD_G32P_VARIANTS
    int flags;                    // diagnostic build (tools/build_k32p_diag.sh) only: timing-only ablations, VD_G32P_FLAGS
#endif
};

// F16 (vd_gemm_desc.math = 2, opt-in mixed precision): single f16 planes for both operands (packed weights: vd_conv3_pack_weights_f16_multi), one
// v_mfma_f32_16x16x32_f16 per product term.  NPART = planes per operand.
// BM (round 5): 128 output channels x 256 pixels, or 256 x 128 -- the same 32 K accumulators, LDS bytes and MFMAs per stage, but HALF the activation
// items to load, split and store per MFMA (the weights, which arrive by LDS-DMA and cost no VALU, double instead).  The kernel is bound by the
// VALU issue of that split, repeated for every m-tile of a pixel tile (6 x for the 256 -> 768 projection); with BM = 256 it is repeated 3 x, and
// layers of 256 output channels convert every activation exactly once.  Taken when M % 256 == 0.
template <bool F16, int BM = 128>
__global__ __launch_bounds__(512, 2) void gemm1x1_k32p_kernel(const g32p_args a) {
    const vd_gemm_desc& d = a.d;
    constexpr int NPIX = 128 * 256 / BM, NTH = 512;
    constexpr int WN = NPIX / 64;                                 // waves along the pixels (4 | 2); 8 / WN along the channels
    constexpr int NPART = F16 ? 1 : 2;
    constexpr int A_UNITS = 2 * NPART * 2 * BM;                   // 1024 units = 16 KB per stage: [c2][part][q][m] (f16: 8 KB; BM = 256: 32 KB)
    constexpr int A_IT = A_UNITS / NTH;                           // 2 (1; BM = 256: 4)
    // activations (round 6): a thread loads CH_T channels x 4 CONSECUTIVE pixels (one 16-byte load per channel: 1 KB per wave instruction instead of the
    // 256 B of round 5's dword loads -- the CU's vector-memory address path takes one wave instruction per ~16-25 cycles whatever its width, and 128 of them per
    // 1536-cycle stage made the loop issue-bound) and writes its CH_T-channel slice of the four pixels' units.  A plane keeps pixel n at unit
    // (n & 3) * PS4 + (n >> 2): the four pixels of a thread go to four slices PS4 = NPIX / 4 + 4 units apart, so a write instruction covers consecutive
    // units and a fragment read (16 consecutive pixels) four 64-byte runs on disjoint banks (272 | 144 dwords apart).
    constexpr int CH_T = NPIX == 256 ? 4 : 2;                     // 4 octets x (8 / CH_T) channel groups x NPIX / 4 pixel quads = 512 threads
    constexpr int PS4 = NPIX / 4 + 4;
    constexpr int PL = 4 * PS4;                                   // units per plane
    constexpr int P_UNITS = 4 * NPART * PL;                       // [c2][part][q][plane]: 34 KB per stage (BM = 256: 18 KB)
    __shared__ u32x4 lds[2 * A_UNITS + 2 * P_UNITS];              // ONE LDS object (see vd_conv_k32p.hip)
    u32x4* const As = lds;
    u32x4* const Ps = lds + 2 * A_UNITS;

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef VD_G32P_VARIANTS    // timing-only ablations (WRONG results): 2 no epilogue stores, 4 no activation loads, 8 no weight DMA, 16 no MFMAs, 32 no split + LDS write
    const bool fl_nostore = a.flags & 2, fl_noload = a.flags & 4, fl_nodma = a.flags & 8, fl_nomfma = a.flags & 16, fl_nowrite = a.flags & 32;
#else
    constexpr bool fl_nostore = false, fl_noload = false, fl_nodma = false, fl_nomfma = false, fl_nowrite = false;
#endif

    const int G8 = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = a.n_tiles >> 3, r8 = a.n_tiles & 7;
    const int xs = xcd * q8 + (xcd < r8 ? xcd : r8), xn = q8 + (xcd < r8 ? 1 : 0);
    if (slot >= xn) return;
    int id = xs + slot;
    const int id_end = xs + xn;

    const int Mpad = d.a_packed_mpad;
    const __amdgpu_buffer_rsrc_t ars =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(reinterpret_cast<const u32x4*>(d.a_packed)), 0, 0xFFFFFFF0, 0x00020000);
    const unsigned ldb4 = 4u * (unsigned)d.ldb;

    // weights: unit u = tid + i * 512 of a stage: run = u >> 7 = c2 * 4 + part * 2 + q, m = u & 127; global unit (cp * 8 + run) * Mpad + m0 + m
    unsigned aoff[4];             // (A_IT <= 4; a constant size: see vd_conv_k32p.hip)
#pragma unroll
    for (int i = 0; i < A_IT; ++i) aoff[i] = 16u * (unsigned)(((tid + i * NTH) / BM) * Mpad + ((tid + i * NTH) % BM));
    auto dma_a = [&](int m0_, int cp, int buf) {
        if (fl_nodma) return;
        const unsigned so = 16u * (unsigned)(cp * 4 * NPART * Mpad + m0_);                    // wave-uniform
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(As + buf * A_UNITS + i * NTH + wave * 64);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, dst, 16, aoff[i], so, 0, 0);
        }
    };
    // activations: thread = (octet cq = wave >> 1, channel group sub, pixel quad): channels cp * 32 + cq * 8 + sub * CH_T + k, pixels n0 + 4 quad + {0..3}
    const int cq = wave >> 1;
    const int quad = tid % (NPIX / 4), sub = (tid / (NPIX / 4)) % (8 / CH_T);
    unsigned poff;
    const int pdst = ((cq >> 1) * 2 * NPART + (cq & 1)) * PL + quad;          // plane (c2, part 0, q), slice 0; the lo plane: + 2 planes; pixel pp: + pp * PS4
    auto set_tile_px = [&](int n0_) {
        const int n = n0_ + 4 * quad;                             // NP % 4 == 0: a quad never straddles images
        const int b = n / d.NP, p = n - b * d.NP;
        poff = 4u * (unsigned)((int64_t)b * d.b_bstride + p) + (unsigned)(cq * 8 + sub * CH_T) * ldb4;
    };
    // Two register sets: the loads of stage s + 2 are issued at the start of stage s and converted at the end of stage s + 1 (two stages of MFMAs,
    // ~1.6 us, between a load and its use: a single stage does not cover an HBM round trip under load -- measured 2.4 us per 0.8-us stage).
    f32x4 rp[2][CH_T];
    // The activation loads are inline asm with hand-counted waits: hipcc's own counter model waits vmcnt(0) where a register set loaded a stage
    // earlier is consumed (it does not see that younger operations may stay in flight), which would drain the next stage's loads every stage
    // (cdna_hip_programming.md §5.7 item 1).  wait_p ties the wait to the destination registers so that no use can be scheduled above it.
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const uint64_t xaddr = reinterpret_cast<uint64_t>(d.B);
    const i32x4 xdesc = {(int)(uint32_t)xaddr, (int)(uint32_t)((xaddr >> 32) & 0xFFFF), (int)0xFFFFFFF0, 0x00020000};   // raw buffer over the whole tensor, no stride
    auto load_p = [&](int cp, f32x4 (&r)[CH_T]) {
        if (fl_noload) return;
#pragma unroll
        for (int k = 0; k < CH_T; ++k) {
            const unsigned so = (unsigned)(cp * 32 + k) * ldb4;                                 // wave-uniform
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(r[k]) : "v"(poff), "s"(xdesc), "s"(so) : "memory");
        }
    };
    auto wait_p = [&](f32x4 (&r)[CH_T], auto LEFT) {             // all but the LEFT youngest vector-memory operations have completed
        constexpr int left = decltype(LEFT)::value;
        if constexpr (CH_T == 4) {
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[CH_T - 1]) : "n"(left) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r[0]), "+v"(r[CH_T - 1]) : "n"(left) : "memory");
        }
    };
    auto write_p = [&](int buf, const f32x4 (&r)[CH_T]) {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16xc __attribute__((ext_vector_type(CH_T)));
        typedef _Float16 f16xc __attribute__((ext_vector_type(CH_T)));
        char* const base = reinterpret_cast<char*>(Ps + buf * P_UNITS + pdst) + sub * CH_T * 2;
        if (fl_nowrite) return;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            if constexpr (F16) {
                f16xc h;
#pragma unroll
                for (int k = 0; k < CH_T; ++k) h[k] = (_Float16)r[k][pp];
                if constexpr (CH_T == 4) *reinterpret_cast<u32x2*>(base + pp * PS4 * 16) = __builtin_bit_cast(u32x2, h);
                else *reinterpret_cast<unsigned*>(base + pp * PS4 * 16) = __builtin_bit_cast(unsigned, h);
            } else {
                bf16xc h, l;
#pragma unroll
                for (int k = 0; k < CH_T; ++k) {
                    const __bf16 t = (__bf16)r[k][pp];
                    h[k] = t;
                    l[k] = (__bf16)(r[k][pp] - (float)t);
                }
                if constexpr (CH_T == 4) {
                    *reinterpret_cast<u32x2*>(base + pp * PS4 * 16) = __builtin_bit_cast(u32x2, h);
                    *reinterpret_cast<u32x2*>(base + (pp * PS4 + 2 * PL) * 16) = __builtin_bit_cast(u32x2, l);
                } else {
                    *reinterpret_cast<unsigned*>(base + pp * PS4 * 16) = __builtin_bit_cast(unsigned, h);
                    *reinterpret_cast<unsigned*>(base + (pp * PS4 + 2 * PL) * 16) = __builtin_bit_cast(unsigned, l);
                }
            }
        }
    };

    f32x4 acc[4][4];                                              // [pixel tile ni][channel tile mi]
    const int wm = wave / WN, wn = wave % WN;
    const int c2 = g >> 1, q = g & 1;
    const u32x4* __restrict__ a_base = As + (c2 * 2 * NPART + q) * BM + wm * 64 + l15;
    const u32x4* __restrict__ p_base = Ps + (c2 * 2 * NPART + q) * PL + (l15 & 3) * PS4 + wn * 16 + (l15 >> 2);      // pixel wn * 64 + ni * 16 + l15: + ni * 4

    // 48 MFMAs of one stage; the next pixel tile's fragments are read before the current tile's 12 MFMAs (see vd_conv_k32p.hip mfma_row_pipe)
    auto mfma_stage = [&](int buf) {
        if (fl_nomfma) return;
        const u32x4* __restrict__ a_cur = a_base + buf * A_UNITS;
        const u32x4* __restrict__ p_cur = p_base + buf * P_UNITS;
        if constexpr (F16) {
            f16x8 wh[4], xh[2];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) wh[mi] = __builtin_bit_cast(f16x8, a_cur[mi * 16]);
            xh[0] = __builtin_bit_cast(f16x8, p_cur[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int cur = ni & 1, nxt = cur ^ 1;
                if (ni < 3) {
                    xh[nxt] = __builtin_bit_cast(f16x8, p_cur[(ni + 1) * 4]);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[cur], wh[mi], acc[ni][mi], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
        bf16x8 wh[4], wl[4], xh[2], xl[2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            wh[mi] = __builtin_bit_cast(bf16x8, a_cur[mi * 16]);
            wl[mi] = __builtin_bit_cast(bf16x8, a_cur[2 * BM + mi * 16]);
        }
        xh[0] = __builtin_bit_cast(bf16x8, p_cur[0]);
        xl[0] = __builtin_bit_cast(bf16x8, p_cur[2 * PL]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int cur = ni & 1, nxt = cur ^ 1;
            if (ni < 3) {
                xh[nxt] = __builtin_bit_cast(bf16x8, p_cur[(ni + 1) * 4]);
                xl[nxt] = __builtin_bit_cast(bf16x8, p_cur[2 * PL + (ni + 1) * 4]);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[cur], wl[mi], acc[ni][mi], 0, 0, 0);
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[cur], wh[mi], acc[ni][mi], 0, 0, 0);
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[cur], wh[mi], acc[ni][mi], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // lane (g, l15) holds pixels n0 + wn*64 + ni*16 + g*4 + {0..3} of channel m0 + wm*64 + mi*16 + l15
    auto epilogue = [&](int m0_, int n0_) {
        int64_t po[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0_ + wn * 64 + ni * 16 + g * 4;        // NP % 4 == 0: a float4 never straddles images
            const int b = n / d.NP, p = n - b * d.NP;
            po[ni] = (int64_t)b * d.d_bstride + p;
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0_ + wm * 64 + mi * 16 + l15;
            const int mc = m < d.M ? m : d.M - 1;
            const float add = d.bias != nullptr ? d.bias[mc] : 0.f;
            f32x4 val[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) val[ni] = d.alpha * acc[ni][mi] + add;
            if (d.residual != nullptr) {
                f32x4 t[4];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int n = n0_ + wn * 64 + ni * 16 + g * 4;
                    const int b = n / d.NP, p = n - b * d.NP;
                    t[ni] = *reinterpret_cast<const f32x4*>(d.residual + (int64_t)b * d.res_bstride + (int64_t)mc * d.ldd + p);
                }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) val[ni] += t[ni];
            }
            float* __restrict__ dst = d.D + (int64_t)mc * d.ldd;
            if (d.accumulate) {
                f32x4 t[4];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) t[ni] = *reinterpret_cast<const f32x4*>(dst + po[ni]);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) val[ni] += t[ni];
            }
            if (m < d.M && !(fl_nostore && val[0][0] != 12345.f)) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) *reinterpret_cast<f32x4*>(dst + po[ni]) = val[ni];
            }
        }
    };

    const int nst = d.K / 32;                                     // stages (chunk pairs) per tile
    auto tile_of = [&](int t, int& m0_, int& n0_) {
        m0_ = (t % a.tiles_m) * BM;
        n0_ = (t / a.tiles_m) * NPIX;
    };
    int m0, n0, m0n = 0, n0n = 0;
    tile_of(id, m0, n0);
    bool has_next = id + G8 < id_end;
    if (has_next) tile_of(id + G8, m0n, n0n);

    // Stage s of a tile (nst is even, so s & 1 is also the parity of the global stage count):
    //   invariant at its start: As[s & 1] / Ps[s & 1] hold stage s; rp[(s + 1) & 1] holds the in-flight loads of stage s + 1
    //   DMA A(s + 1) -> As[(s + 1) & 1];  loads X(s + 2) -> rp[s & 1];  48 MFMAs;  convert rp[(s + 1) & 1] -> Ps[(s + 1) & 1];
    //   wait for the DMA (the CH_T younger activation loads stay in flight across the barrier);  barrier
    // "s + 1" / "s + 2" run into the next tile of this workgroup at a tile's end (its m-tile for the weights, its pixels for the activations).
    auto stage = [&](int cp, auto PAR, int m0_, int m0n_, int n0n_, bool has_next_) {
        constexpr int par = decltype(PAR)::value;
        const bool e1 = cp + 1 < nst || has_next_, e2 = cp + 2 < nst || has_next_;
        if (e1) {
            if (cp + 1 < nst) dma_a(m0_, cp + 1, par ^ 1);
            else dma_a(m0n_, 0, par ^ 1);
        }
        if (e2 && cp + 2 == nst) set_tile_px(n0n_);              // this tile's activation loads are all issued: switch to the next tile's pixels
        // ALWAYS CH_T loads per stage (past the end of the work: the last stage's again, never used): the hand-counted wait below is then the same
        // instruction on every path -- two asm statements on two branches would make hipcc merge their register operands with copies it may
        // place in front of the wait, i.e. copies of registers whose loads are still in flight
        load_p(e2 ? (cp + 2 < nst ? cp + 2 : cp + 2 - nst) : nst - 1, rp[par]);
        __builtin_amdgcn_sched_barrier(0);
        mfma_stage(par);
        // stage s + 1's activations (loaded a stage ago) and its DMA (issued before this stage's 16 loads) have landed
        wait_p(rp[par ^ 1], std::integral_constant<int, CH_T>{});
        if (e1) write_p(par ^ 1, rp[par ^ 1]);
        // this wave's LDS writes are done.  Raw s_barrier: __syncthreads() would drain the vector-memory queue (vmcnt(0)) and with it the
        // two-stage lead of the activation loads.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // prologue: stages 0 and 1 of the first tile
    set_tile_px(n0);
    dma_a(m0, 0, 0);
    load_p(0, rp[0]);
    load_p(1, rp[1]);                                              // (nst >= 2: K % 64 == 0)
    wait_p(rp[0], std::integral_constant<int, CH_T>{});
    write_p(0, rp[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    while (true) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int cp = 0; cp < nst; cp += 2) {
            stage(cp, std::integral_constant<int, 0>{}, m0, m0n, n0n, has_next);
            stage(cp + 1, std::integral_constant<int, 1>{}, m0, m0n, n0n, has_next);
        }
        epilogue(m0, n0);
        if (!has_next) break;
        id += G8;
        m0 = m0n, n0 = n0n;
        has_next = id + G8 < id_end;
        if (has_next) tile_of(id + G8, m0n, n0n);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the unused loads of the last two stages
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

}  // namespace

// vd_gemm.hip asks: can the persistent 16x16x32 1x1 kernel take this VD_B_PLAIN / a_packed problem, and is its grid worth it?
bool vd_gemm1x1_k32p_pick(const vd_gemm_desc& d) {
    static const int off = env_int("VD_G32P_OFF", 0);
    if (off || !d.a_packed || d.b_mode != VD_B_PLAIN) return false;
    if (d.K % 64 != 0 || d.N % 256 != 0 || d.NP % 4 != 0 || d.M < 64 || d.a_packed_mpad < d.M || (d.a_packed_mpad & 127)) return false;
    if (d.bias_on_n || d.d_trans || d.nb2 > 1 || d.a_bstride != 0 || d.rowadd || d.gn_ss || d.gn_part || d.debug || d.tile || d.act || d.pool2) return false;
    if ((d.ldd & 3) || (d.d_bstride & 3) || (((uintptr_t)d.D) & 15) || (((uintptr_t)d.a_packed) & 15)) return false;
    if ((d.ldb & 3) || (d.b_bstride & 3) || (((uintptr_t)d.B) & 15)) return false;      // 16-byte activation loads (four consecutive pixels of a channel)
    if (d.residual && ((d.res_bstride & 3) || (((uintptr_t)d.residual) & 15))) return false;
    const int64_t nb = d.N / d.NP;
    if (nb * d.b_bstride * 4 >= (1ll << 32) || (int64_t)d.K * d.ldb * 4 >= (1ll << 32)) return false;      // 32-bit buffer offsets
    const int nt = vd_cdiv(d.M, 128) * (d.N / 256);               // (the 256 x 128 tiles of M % 256 == 0 problems: the same count)
    const int rounds = vd_cdiv(nt, 256);
    constexpr int min_tiles = 192;          // (lower thresholds measured neutral: profiles/r04_g32p_min_tiles.txt)
    if (nt < min_tiles) return false;
    return nt < 192 || 4 * nt >= 3 * rounds * 256;                // every round of 256 workgroup slots at least 75 % full
}

int vd_launch_gemm1x1_k32p(const vd_gemm_desc& d, hipStream_t st) {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return -1;
        n_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
        n_cu &= ~7;
        if (n_cu < 8) n_cu = 8;
    }
    static const int bm256 = env_int("VD_G32P_BM256", 1);         // A/B switch: 0 = the 128 x 256 tile everywhere (round 4)
    const bool big_m = bm256 && d.math != 2 && d.M % 256 == 0 && d.N % 128 == 0;
    g32p_args a;
#ifdef VD_G32P_VARIANTS
    a.flags = env_int("VD_G32P_FLAGS", 0);
#endif
    a.d = d;
    a.tiles_m = vd_cdiv(d.M, big_m ? 256 : 128);
    a.n_tiles = a.tiles_m * (d.N / (big_m ? 128 : 256));
    const int grid = a.n_tiles < n_cu ? ((a.n_tiles + 7) & ~7) : n_cu;
    if (d.math == 2) hipLaunchKernelGGL((gemm1x1_k32p_kernel<true, 128>), dim3(grid), dim3(512), 0, st, a);
    else if (big_m) hipLaunchKernelGGL((gemm1x1_k32p_kernel<false, 256>), dim3(grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((gemm1x1_k32p_kernel<false, 128>), dim3(grid), dim3(512), 0, st, a);
    return 0;
}
