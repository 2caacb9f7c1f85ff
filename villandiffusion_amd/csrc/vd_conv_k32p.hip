// ---- split-precision 3x3 convolution on v_mfma_f32_16x16x32_bf16, PERSISTENT tile walk (round 4) ---------------------------------------
// Same arithmetic, packed-weight layout, LDS images and MFMA loop as conv3_k32_kernel (vd_conv_k32.inc, round 3); what changes is everything
// AROUND the loop.  Round 3's kernel is one 128-channel x 256-pixel tile per workgroup, one workgroup per CU (143 KB of LDS): nothing on a CU
// overlaps a tile's prologue (first weight stage + halo patch out of HBM, ~3 us) or its epilogue (128 KB of stores per workgroup, all 256
// workgroups of a round at once: 33 MB, ~6 us of HBM writes with the matrix pipes idle).  Measured (profiles/r03_conv_k32_ab.txt, B = 128, 32x32,
// 128 / 256 / 384 input channels): 103 / 180 / 242 us = about 33 us per LAUNCH that does not scale with the channel loop (two rounds of 256
// tiles), i.e. a third of the 128-channel layers, while the loop itself already runs at ~1.65 PFLOP/s executed.  Here:
//   * ONE workgroup per CU walks tiles id0, id0 + 32, ... of its XCD's contiguous tile range (blockIdx % 8 = XCD under round-robin dispatch: speed
//     only); the stage pipeline (weights two stages ahead, patch one chunk pair ahead) simply continues INTO the next tile: its first weight
//     stages and its first halo patch are fetched and converted beside the last MFMAs of the current tile, the epilogue's stores drain beside
//     the next tile's loop.  Only the first prologue and the last epilogue of a launch stay exposed.
//   * the tile is TR x TW pixels (TW = 16: the whole 16x16 image; TW = 32: 8 rows x 32 columns) at ANY position of an OH x OW image with
//     OW % TW == 0, OH % TR == 0: the 64x64 ... 256x256 levels of BASELINE configs #4 / #5 run on this kernel too (round 3: images wider than
//     32 pixels fell back to the 32x32x16 kernel's 128 x 128 tiles).
//   * DMA = true: the packed weights go global -> LDS by LDS-DMA (buffer_load ... lds, 16 bytes per lane, the As stage image is lane-linear
//     already): no staging registers, no ds_write_b128 pass (48 KB per stage per workgroup through the VGPR -> LDS path), counted vmcnt waits
//     and raw s_barrier by hand (cdna_hip_programming.md §5 "Pipelining across barriers").
// Reference work replaced: diffusers ResnetBlock2D / Upsample2D 3x3 convolutions fwd + input gradient (reference loss.py:993 -> UNet2DModel).
#include "vd_common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// F16 kernels (round 4, opt-in mixed precision: vd_gemm_desc.math = 2): ONE f16 product per term instead of the three bf16 ones.
__device__ __forceinline__ u32x4 to_f16x8(const float (&v)[8]) {
    f16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)v[j];           // v_cvt_f16_f32: round to nearest even
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

struct k32p_args {
    vd_gemm_desc d;
    int n_tiles;      // tiles_m * tiles_n
    int tiles_m;
    int tiles_x;      // OW / TW
    int tiles_img;    // (OH / TR) * (OW / TW)
    int flags;        // bit 0: no stagger of the wave halves (A/B)
    unsigned long long* stamps;   // diagnostic build (-DVD_K32P_STAMPS) only: [grid][16] s_memrealtime / s_memtime stamps per workgroup
};

// F16 (math = 2): the operands are single f16 planes -- packed weights [chunk][tap][q][Mpad] (vd_conv3_pack_weights_f16), patch planes [c2][q] --
// and a product term is ONE v_mfma_f32_16x16x32_f16.  NPART = planes per operand (2: bf16 hi / lo, 1: f16); every index below is written in it.
// PS (round 5, vd_gemm_desc.b_presplit): the input is a PRE-SPLIT image (vd_presplit.hip: unit (octet, pixel, part) = the 8 channels of a pixel as bf16
// hi / lo, 16 bytes, at ((octet * HW + pixel) * 2 + part) * 16) written by its producer -- a patch item is then two 16-byte loads that go to LDS as they
// are: no conversion (24 VALU instructions per item), 2 instead of 8 loads per item.  Same values, same LDS image, same MFMA order: same bits.
template <int TW, int MODE, bool DMA, bool PIPE, bool F16, bool PS = false>   // MODE 0: CONV3, 1: CONV3_T (flipped taps), 2: CONV3_UP, 3: CONV3 of silu(GroupNorm(x))
__global__ __launch_bounds__(512, 2) void conv3_k32p_kernel(const k32p_args a) {
    static_assert(!PS || (!F16 && MODE != 3), "pre-split inputs: split-precision arithmetic, no folded GroupNorm");
    const vd_gemm_desc& d = a.d;
    constexpr int BM = 128, NPIX = 256, NTH = 512;
    constexpr int NPART = F16 ? 1 : 2;
    constexpr int RUNS = 3 * NPART * 2;                            // 2 KB runs of one chunk's tap row: (s, part, q)
    constexpr int TR = NPIX / TW;
    constexpr int PW = TW + 2, PR = TR + 2, PIMG = PR * PW;
    constexpr int PLANE = (PIMG + 15) / 16 * 16;
    constexpr int A_HALF = RUNS * BM;                              // 1536 (768) units: one tap row of ONE chunk
    constexpr int A_UNITS = 2 * A_HALF;                           // 3072 units = 48 KB per buffer (f16: 24 KB)
    constexpr int A_IT = A_UNITS / NTH;                           // 6 (3)
    constexpr int P_IT = (PIMG + 127) / 128;                      // 3
    constexpr int P_LOADS = PS ? 2 * P_IT : 8 * P_IT;             // vector-memory loads of one patch per thread (counted vmcnt waits below)
    constexpr int RED_UNITS = 8 * 64 * 2 * 4 / 16;                // gn_part scratch: [wave][64 channels][2] floats = 4 KB
    // ONE LDS object (a second __shared__ array beside an LDS-DMA target makes hipcc wait vmcnt(0) before every ds_read: guide §5 item 4a)
    __shared__ u32x4 lds[2 * A_UNITS + 4 * NPART * PLANE + RED_UNITS];
    u32x4* const As = lds;
    u32x4* const Ps = lds + 2 * A_UNITS;
    float* const red = reinterpret_cast<float*>(lds + 2 * A_UNITS + 4 * NPART * PLANE);

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef VD_K32P_STAMPS      // where a tile's time goes (guide §7 "In-kernel stamps"): never in the shipped library
    int n_stamp = 0;
    auto stamp = [&]() {
        if (a.stamps != nullptr && tid == 0 && n_stamp < 15) {
            a.stamps[blockIdx.x * 32 + n_stamp] = __builtin_amdgcn_s_memrealtime();
            a.stamps[blockIdx.x * 32 + 16 + n_stamp] = __builtin_amdgcn_s_memtime();
            ++n_stamp;
        }
    };
#define VD_STAMP() stamp()
#else
#define VD_STAMP()
#endif
    VD_STAMP();                                                   // 0: workgroup start
#ifdef VD_K32P_STAMPS      // per-wave segment sums of the stage pipeline (shader cycles): [issue, mfma, patch switch, vmcnt wait, barrier, epilogue];
                           // every tick is one s_memtime round trip on the wave's critical path (~100-200 cycles): read differences, not absolutes
    unsigned long long fs[6] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull}, ft_prev = 0ull;
    auto ftick = [&](int k) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        if (k >= 0) fs[k] += t - ft_prev;
        ft_prev = t;
    };
#define VD_FTICK(k)                                 \
    do {                                            \
        __builtin_amdgcn_sched_barrier(0);          \
        ftick(k);                                   \
        __builtin_amdgcn_sched_barrier(0);          \
    } while (0)
#else
#define VD_FTICK(k)
#endif
#ifdef VD_K32P_VARIANTS    // timing-only ablations (WRONG results) of the diagnostic build (tools/build_k32p_diag.sh): VD_K32P_FLAGS bits 2 / 4 / 8 / 16
    const bool fl_nopatch = a.flags & 2, fl_nodma = a.flags & 4, fl_noepi = a.flags & 8, fl_nomfma = a.flags & 16;
#else
    constexpr bool fl_nopatch = false, fl_nodma = false, fl_noepi = false, fl_nomfma = false;
#endif

    // ---- this workgroup's tile list: XCD x = blockIdx % 8 owns the contiguous range [xs, xs + xn), slot j = blockIdx / 8 walks xs + j, + G/8, ...
    const int G8 = gridDim.x >> 3;                                // workgroups per XCD (gridDim.x % 8 == 0)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = a.n_tiles >> 3, r8 = a.n_tiles & 7;
    const int xs = xcd * q8 + (xcd < r8 ? xcd : r8), xn = q8 + (xcd < r8 ? 1 : 0);
    if (slot >= xn) return;
    int id = xs + slot;
    const int id_end = xs + xn;

    const u32x4* __restrict__ Apk = reinterpret_cast<const u32x4*>(d.a_packed);
    const int Mpad = d.a_packed_mpad;
    const int HWs = d.H * d.W;                                    // source plane (MODE 2: the half-resolution input)
    const int OWi = d.OW, OHi = d.OH;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(Apk), 0, 0xFFFFFFF0, 0x00020000);

    // tile decode (wave-uniform)
    int m0, b0, y0, x0, tix;                                      // current tile
    auto decode = [&](int t, int& m0_, int& b0_, int& y0_, int& x0_, int& tix_) {
        const int tm = t % a.tiles_m, tn = t / a.tiles_m;
        b0_ = tn / a.tiles_img;
        tix_ = tn - b0_ * a.tiles_img;
        const int ty = tix_ / a.tiles_x, tx = tix_ - ty * a.tiles_x;
        m0_ = tm * BM;
        y0_ = ty * TR;
        x0_ = tx * TW;
    };
    decode(id, m0, b0, y0, x0, tix);

    // patch item (cq, pixel): waves 2 cq and 2 cq + 1 stage the k-octet cq of the chunk pair, thread t & 127 the patch pixels (t & 127) + 128 i
    const int cq = wave >> 1;
    int ppy[P_IT], ppx[P_IT], pdst[P_IT];
    unsigned pvalid = 0;
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
        const int rem = (tid & 127) + 128 * i;
        ppy[i] = rem / PW;
        ppx[i] = rem - ppy[i] * PW;
        pvalid |= (rem < PIMG ? 1u : 0u) << i;
        pdst[i] = ((cq >> 1) * 2 * NPART + (cq & 1)) * PLANE + rem;      // Ps[c2][part = 0][q][pixel]; the lo part sits 2 * PLANE further
    }
    unsigned poff[P_IT];
    unsigned poff_lo[PS ? P_IT : 1];                              // PS: the lo unit's offset (a masked item must stay out of range: 0xFFFFFFFF + 16 would wrap)
    unsigned pmask = 0;
    __amdgpu_buffer_rsrc_t xrs;
    __amdgpu_buffer_rsrc_t ors;                                  // MODE 3, d.act_out: the normalised activation's image of the tile's batch item
    unsigned pinner = 0;                                          // ... items that are pixels OF the tile (not its halo): written by this workgroup
    bool act_store = false;
    const float* __restrict__ ssg = nullptr;
    auto set_patch_tile = [&](int b0_, int y0_, int x0_, int m0_) {       // per-tile source offsets of the patch items (fixed for the tile's chunk pairs)
        pmask = 0;
#pragma unroll
        for (int i = 0; i < P_IT; ++i) {
            int iy = y0_ + ppy[i] - 1, ix = x0_ + ppx[i] - 1;
            bool ok = (pvalid >> i) & 1u;
            ok = ok && (unsigned)iy < (unsigned)OHi && (unsigned)ix < (unsigned)OWi;
            if (MODE == 2) {
                iy >>= 1;
                ix >>= 1;
            }
            if constexpr (PS) {
                poff[i] = ok ? 32u * (unsigned)(cq * HWs + iy * d.W + ix) : 0xFFFFFFFFu;
                poff_lo[i] = ok ? poff[i] + 16u : 0xFFFFFFFFu;
            } else {
                poff[i] = ok ? 4u * (unsigned)(cq * 8 * HWs + iy * d.W + ix) : 0xFFFFFFFFu;
            }
            pmask |= (ok ? 1u : 0u) << i;
        }
        act_store = MODE == 3 && d.act_out != nullptr && m0_ == 0;      // every element once: the workgroups of the first channel tile
        if (act_store) {
            pinner = 0;
#pragma unroll
            for (int i = 0; i < P_IT; ++i) {
                const bool in = ((pmask >> i) & 1u) && ppy[i] >= 1 && ppy[i] <= TR && ppx[i] >= 1 && ppx[i] <= TW;
                pinner |= (in ? 1u : 0u) << i;
            }
            ors = __builtin_amdgcn_make_buffer_rsrc(d.act_out + (int64_t)b0_ * d.act_bstride, 0, 0xFFFFFFF0, 0x00020000);
        }
        xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.B + (int64_t)b0_ * d.b_bstride), 0, 0xFFFFFFF0, 0x00020000);
        if (MODE == 3) ssg = d.gn_ss + (int64_t)b0_ * 2 * d.C + 16 * cq;
    };

    u32x4 ra[DMA ? 1 : A_IT];
    float rp[PS ? 1 : P_IT][8];
    u32x4 cph[P_IT], cpl[P_IT];
    unsigned aoff[6];             // (A_IT <= 6; sized by a constant: with `aoff[A_IT]`, A_IT depending on F16, hipcc 7.2 emits no host stubs for this template)
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int run = (tid + i * NTH) >> 7;                     // 0..23 = c2 * 12 + (s, part, q); chunk c2 = 1 is 36 runs further in the packed operand
        aoff[i] = 16u * (unsigned)(((run % RUNS) + 3 * RUNS * (run / RUNS)) * Mpad + (tid & 127));
    }
    auto load_a = [&](int m0_, int cp, int r, int buf) {         // stage (m-tile, chunk pair, tap row): registers, or straight into As[buf] (DMA)
        const unsigned so = 16u * (unsigned)((cp * 6 * RUNS + r * RUNS) * Mpad + m0_);        // wave-uniform
        if constexpr (DMA) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                __attribute__((address_space(3))) void* dst =
                    (__attribute__((address_space(3))) void*)(As + buf * A_UNITS + i * NTH + wave * 64);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, dst, 16, aoff[i], so, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) ra[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, aoff[i], so, 0));
        }
    };
    auto store_a = [&](int buf) {
        if constexpr (!DMA) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) As[buf * A_UNITS + tid + i * NTH] = ra[i];
        }
    };
    auto load_p = [&](int cp) {
        if constexpr (PS) {                                       // the (hi, lo) units of the item, as they will sit in LDS
            const unsigned so = 32u * (unsigned)(cp * 4 * HWs);                                  // wave-uniform: chunk pair = 4 octets
#pragma unroll
            for (int i = 0; i < P_IT; ++i) {
                cph[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, poff[i], so, 0));
                cpl[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, poff_lo[i], so, 0));
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned so = 4u * (unsigned)((cp * 32 + j) * HWs);                           // wave-uniform
#pragma unroll
            for (int i = 0; i < P_IT; ++i) rp[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, poff[i], so, 0));
        }
    };
    auto convert_p = [&](int cp) {
        if constexpr (PS) return;
        float ss[16];
        if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 16; ++j) ss[j] = ssg[cp * 64 + j];
        }
#pragma unroll
        for (int i = 0; i < P_IT; ++i) {
            float v[8];
            const bool ok = (pmask >> i) & 1u;
            if (MODE == 3) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float z = rp[i][j] * ss[2 * j] + ss[2 * j + 1];
                    rp[i][j] = z * sigmoidf_(z);
                }
                if (act_store && ((pinner >> i) & 1u)) {          // the training forward keeps silu(gn(x)) for the weight gradient: same offsets as the loads
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rp[i][j]), ors, poff[i], 4u * (unsigned)((cp * 32 + j) * HWs), 0);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (MODE != 3 || ok) ? rp[i][j] : 0.f;
            if constexpr (F16) cph[i] = to_f16x8(v);
            else split8(v, cph[i], cpl[i]);
        }
    };
    auto write_p = [&]() {
#pragma unroll
        for (int i = 0; i < P_IT; ++i) {
            if ((pvalid >> i) & 1u) {
                Ps[pdst[i]] = cph[i];
                if constexpr (!F16) Ps[pdst[i] + 2 * PLANE] = cpl[i];
            }
        }
    };

    f32x4 acc[4][4];                                              // [pixel tile ni][channel tile mi]
    const int wm = wave >> 2, wn = wave & 3;
    const int c2 = g >> 1, q = g & 1;
    const u32x4* __restrict__ a_base = As + c2 * A_HALF + q * BM + wm * 64 + l15;
    const u32x4* __restrict__ p_base[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int qx = wn * 64 + ni * 16 + l15;
        const int ty = qx / TW, x = qx - ty * TW;
        p_base[ni] = Ps + (c2 * 2 * NPART + q) * PLANE + ty * PW + x;
    }

    auto mfma_row = [&](int r, int buf) {
        const int pr = (MODE == 1) ? 2 - r : r;
        const u32x4* __restrict__ a_cur = a_base + buf * A_UNITS;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int ps = (MODE == 1) ? 2 - s : s;
            bf16x8 wh[4], wl[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                wh[mi] = __builtin_bit_cast(bf16x8, a_cur[(s * 4 + 0) * BM + mi * 16]);
                wl[mi] = __builtin_bit_cast(bf16x8, a_cur[(s * 4 + 2) * BM + mi * 16]);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const bf16x8 xh = __builtin_bit_cast(bf16x8, p_base[ni][pr * PW + ps]);
                const bf16x8 xl = __builtin_bit_cast(bf16x8, p_base[ni][2 * PLANE + pr * PW + ps]);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wl[mi], acc[ni][mi], 0, 0, 0);
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, wh[mi], acc[ni][mi], 0, 0, 0);
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wh[mi], acc[ni][mi], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                    // one tap's fragments live at a time
        }
    };

    // The same 144 MFMAs with the fragment reads software-pipelined by hand (PIPE): hipcc issues a pixel-tile's two ds_read_b128 right in front of
    // the 12 MFMAs that consume them and waits lgkmcnt(0) -- one exposed LDS round trip per 12 MFMAs and wave (in-kernel stamps: a stage takes
    // ~7000 shader cycles where its MFMAs need 4608).  Here the next pixel tile's fragments are read BEFORE the current tile's MFMAs (second
    // register pair), and the next tap's weight fragments replace the current ones one channel tile at a time inside the tap's last pixel tile,
    // right behind the last MFMAs that read them; sched_group_barriers pin that order.  Same MFMA order per accumulator: same bits.
    auto mfma_row_pipe = [&](int r, int buf) {
        const int pr = (MODE == 1) ? 2 - r : r;
        const u32x4* __restrict__ a_cur = a_base + buf * A_UNITS;
        bf16x8 wh[4], wl[4], xh[2], xl[2];
        auto tap_col = [&](int s) { return (MODE == 1) ? 2 - s : s; };
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            wh[mi] = __builtin_bit_cast(bf16x8, a_cur[(0 * 4 + 0) * BM + mi * 16]);
            wl[mi] = __builtin_bit_cast(bf16x8, a_cur[(0 * 4 + 2) * BM + mi * 16]);
        }
        xh[0] = __builtin_bit_cast(bf16x8, p_base[0][pr * PW + tap_col(0)]);
        xl[0] = __builtin_bit_cast(bf16x8, p_base[0][2 * PLANE + pr * PW + tap_col(0)]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int cur = (s * 4 + ni) & 1, nxt = cur ^ 1;
                const bool last = (s == 2 && ni == 3);
                if (!last) {                                      // the next pixel tile (of this tap, or the first of the next tap)
                    const int ns = (ni == 3) ? s + 1 : s, nn = (ni == 3) ? 0 : ni + 1;
                    xh[nxt] = __builtin_bit_cast(bf16x8, p_base[nn][pr * PW + tap_col(ns)]);
                    xl[nxt] = __builtin_bit_cast(bf16x8, p_base[nn][2 * PLANE + pr * PW + tap_col(ns)]);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[cur], wl[mi], acc[ni][mi], 0, 0, 0);
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[cur], wh[mi], acc[ni][mi], 0, 0, 0);
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[cur], wh[mi], acc[ni][mi], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    if (ni == 3 && s < 2) {                       // this channel tile's weights are dead: fetch the next tap's
                        wh[mi] = __builtin_bit_cast(bf16x8, a_cur[((s + 1) * 4 + 0) * BM + mi * 16]);
                        wl[mi] = __builtin_bit_cast(bf16x8, a_cur[((s + 1) * 4 + 2) * BM + mi * 16]);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // F16: the same pipeline with one fragment per operand and one MFMA per (pixel tile, channel tile)
    auto mfma_row_f16 = [&](int r, int buf) {
        const int pr = (MODE == 1) ? 2 - r : r;
        const u32x4* __restrict__ a_cur = a_base + buf * A_UNITS;
        f16x8 wh[4], xh[2];
        auto tap_col = [&](int s) { return (MODE == 1) ? 2 - s : s; };
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) wh[mi] = __builtin_bit_cast(f16x8, a_cur[mi * 16]);
        xh[0] = __builtin_bit_cast(f16x8, p_base[0][pr * PW + tap_col(0)]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int cur = (s * 4 + ni) & 1, nxt = cur ^ 1;
                if (!(s == 2 && ni == 3)) {
                    const int ns = (ni == 3) ? s + 1 : s, nn = (ni == 3) ? 0 : ni + 1;
                    xh[nxt] = __builtin_bit_cast(f16x8, p_base[nn][pr * PW + tap_col(ns)]);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[cur], wh[mi], acc[ni][mi], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (ni == 3 && s < 2) {
                        wh[mi] = __builtin_bit_cast(f16x8, a_cur[((s + 1) * 2) * BM + mi * 16]);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- epilogue of one tile: lane (g, l15) holds pixels wn*64 + ni*16 + g*4 + {0..3} of channel m0 + wm*64 + mi*16 + l15 ----
    auto epilogue = [&](int m0_, int b0_, int y0_, int x0_, int tix_) {
        float* __restrict__ Db = d.D + (int64_t)b0_ * d.d_bstride;
        if (MODE == 1 && d.pool2) {
            // input gradient of the upsample convolution: dx[y][x] = sum of the 2x2 block of dU (the full-resolution dU never exists).
            // TW = 32: pixel tiles ni / ni + 2 of a wave are the same columns of rows 2 wn, 2 wn + 1; TW = 16: tile ni is row 4 wn + ni.
            const int OW2 = OWi >> 1;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int m = m0_ + wm * 64 + mi * 16 + l15;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int na = (TW == 32) ? k : 2 * k, nb_ = (TW == 32) ? k + 2 : 2 * k + 1;
                    const f32x4 t = acc[na][mi] + acc[nb_][mi];
                    const f32x2 o = f32x2{d.alpha * (t[0] + t[1]), d.alpha * (t[2] + t[3])};
                    const int oy = (TW == 32) ? (y0_ >> 1) + wn : (y0_ >> 1) + 2 * wn + k;
                    const int ox = (x0_ >> 1) + ((TW == 32) ? k * 8 + g * 2 : g * 2);
                    if (m < d.M) *reinterpret_cast<f32x2*>(Db + (int64_t)m * d.ldd + oy * OW2 + ox) = o;
                }
            }
            return;
        }
        // pixel offset of the lane's first float4 inside the image: tile row (wn*64 + ni*16 + g*4) / TW, column ... % TW
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0_ + wm * 64 + mi * 16 + l15;
            const int mc = m < d.M ? m : d.M - 1;
            float add = 0.f;
            if (d.bias != nullptr) add = d.bias[mc];
            if (d.rowadd != nullptr) add += d.rowadd[(int64_t)b0_ * d.rowadd_bstride + mc];
            f32x4 val[4];
            int po[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                val[ni] = d.alpha * acc[ni][mi] + add;
                const int qx = wn * 64 + ni * 16 + g * 4;
                po[ni] = (y0_ + qx / TW) * OWi + x0_ + (qx % TW);
            }
            if (d.residual != nullptr) {
                const float* __restrict__ rs_ = d.residual + (int64_t)b0_ * d.res_bstride + (int64_t)mc * d.ldd;
                f32x4 t[4];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) t[ni] = *reinterpret_cast<const f32x4*>(rs_ + po[ni]);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) val[ni] += t[ni];
            }
            float* __restrict__ dst = Db + (int64_t)mc * d.ldd;
            if (d.accumulate) {
                f32x4 t[4];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) t[ni] = *reinterpret_cast<const f32x4*>(dst + po[ni]);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) val[ni] += t[ni];
            }
            if (m < d.M) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) *reinterpret_cast<f32x4*>(dst + po[ni]) = val[ni];
            }
            if (d.gn_part != nullptr) {                           // (sum, sum of squares) of this channel over the wave's 64 pixels, fixed order
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        s1 += val[ni][j];
                        s2 += val[ni][j] * val[ni][j];
                    }
                s1 += __shfl_xor(s1, 16, 64);
                s2 += __shfl_xor(s2, 16, 64);
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (g == 0) {
                    red[(wave * 64 + mi * 16 + l15) * 2] = s1;
                    red[(wave * 64 + mi * 16 + l15) * 2 + 1] = s2;
                }
            }
        }
        if (d.gn_part != nullptr) {                               // the four pixel quarters (waves wn = 0..3 of a channel half), in wave order
            __syncthreads();
            if (tid < BM && m0_ + tid < d.M) {
                const int wm_ = tid >> 6, cl = tid & 63;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) {
                    s1 += red[((wm_ * 4 + w4) * 64 + cl) * 2];
                    s2 += red[((wm_ * 4 + w4) * 64 + cl) * 2 + 1];
                }
                float* __restrict__ o = d.gn_part + (((int64_t)b0_ * a.tiles_img + tix_) * d.M + m0_ + tid) * 2;
                o[0] = s1;
                o[1] = s2;
            }
        }
    };

    // ---- the stage pipeline.  Stage = (tile, chunk pair cp, tap row r); G(s): stage s's weight loads (registers or DMA), W(s): registers -> As
    // (register staging only), M(s): its 144 MFMAs.  Register staging: the two waves of a SIMD (w, w + 4) run G / M / W in different orders
    // between two barriers (stagger, MI355X_MICROARCH.md "Two waves per SIMD" item 9):  waves 0-3: G(s+1) M(s) W(s+1) | waves 4-7: W(s+1) G(s+2) M(s).
    // DMA: every wave issues G(s+1) (into the other buffer) before M(s) and waits for its own DMAs with a counted vmcnt before the barrier.
    const int npairs = d.C / 32;
    const bool late = !DMA && wave >= 4 && !(a.flags & 1);        // wave-uniform
    auto stage_m0 = [&](int cp, int r, int ahead, int m0c, int m0n, int& scp, int& sr, bool& exists, bool has_next) -> int {
        // the stage `ahead` (1 or 2) after (cp, r) of the current tile: its (cp, r), its m-tile, and whether it exists at all
        int lin = cp * 3 + r + ahead;
        const int T1 = 3 * npairs;
        if (lin < T1) {
            scp = lin / 3;
            sr = lin - 3 * scp;
            exists = true;
            return m0c;
        }
        lin -= T1;                                                // into the next tile (T1 >= 3 > ahead: never beyond it)
        scp = lin / 3;
        sr = lin - 3 * scp;
        exists = has_next;
        return m0n;
    };

    bool has_next = id + G8 < id_end;
    int m0n = m0, b0n = b0, y0n = y0, x0n = x0, tixn = tix;
    if (has_next) decode(id + G8, m0n, b0n, y0n, x0n, tixn);

    set_patch_tile(b0, y0, x0, m0);
    load_a(m0, 0, 0, 0);
    load_p(0);
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P_LOADS) : "memory");        // the DMA stage (issued first) has landed
    store_a(0);
    convert_p(0);
    write_p();
    {
        int scp, sr;
        bool ex;
        const int sm = stage_m0(0, 0, 1, m0, m0n, scp, sr, ex, has_next);
        if (late && ex) load_a(sm, scp, sr, 1);                   // G(1) of the late half
    }
    __syncthreads();
    VD_STAMP();                                                   // 1: prologue done
    VD_FTICK(-1);

    int t = 0;                                                    // stage parity across tiles
    while (true) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int cp = 0; cp < npairs; ++cp) {
            const bool more = cp + 1 < npairs;
            const bool pnext = more || has_next;                  // a patch follows this chunk pair (next pair, or the next tile's first)
#pragma unroll
            for (int r = 0; r < 3; ++r, ++t) {
                const int buf = t & 1;
                int scp1, sr1, scp2, sr2;
                bool ex1, ex2;
                const int sm1 = stage_m0(cp, r, 1, m0, m0n, scp1, sr1, ex1, has_next);
                const int sm2 = stage_m0(cp, r, 2, m0, m0n, scp2, sr2, ex2, has_next);
                if constexpr (DMA) {
                    if (ex1 && !fl_nodma) load_a(sm1, scp1, sr1, buf ^ 1);     // As[buf ^ 1] was last read by M(s-1): every wave has left the barrier behind it
                } else if (late) {
                    if (ex1) store_a(buf ^ 1);                    // W(s+1)
                    if (ex2) load_a(sm2, scp2, sr2, 0);           // G(s+2)
                } else if (ex1) {
                    load_a(sm1, scp1, sr1, 0);                    // G(s+1)
                }
                if (r == 1 && pnext && !fl_nopatch) {
                    if (more) {
                        load_p(cp + 1);
                    } else {                                      // the next tile's first patch
                        set_patch_tile(b0n, y0n, x0n, m0n);
                        load_p(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                VD_FTICK(0);
                if (r == 2 && pnext && !fl_nopatch) {             // VALU work beside the MFMAs of this tap row
                    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_IT) : "memory");   // the patch loads (older than this stage's DMAs)
                    convert_p(more ? cp + 1 : 0);
                }
                if (!fl_nomfma) {
                    if constexpr (F16) mfma_row_f16(r, buf);
                    else if constexpr (PIPE) mfma_row_pipe(r, buf);
                    else mfma_row(r, buf);
                }
                VD_FTICK(1);
                if (!DMA && !late && ex1) store_a(buf ^ 1);       // W(s+1)
                if (r == 2 && pnext && !fl_nopatch) {
                    __syncthreads();                              // every wave has finished reading the patch
                    write_p();
                }
                VD_FTICK(2);
                if constexpr (DMA) {
                    // this stage's DMAs (issued before any patch load of this stage) must have landed before the barrier that releases M(s+1)
                    if (r == 1 && pnext && !fl_nopatch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P_LOADS) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                VD_FTICK(3);
                __syncthreads();
                VD_FTICK(4);
            }
        }
        VD_STAMP();                                               // 2, 4, ...: a tile's channel loop done
        if (!fl_noepi) epilogue(m0, b0, y0, x0, tix);
        VD_STAMP();                                               // 3, 5, ...: its epilogue issued
        VD_FTICK(5);
        if (!has_next) break;
        id += G8;
        m0 = m0n, b0 = b0n, y0 = y0n, x0 = x0n, tix = tixn;
        has_next = id + G8 < id_end;
        if (has_next) decode(id + G8, m0n, b0n, y0n, x0n, tixn);
    }
#ifdef VD_K32P_STAMPS
    if (a.stamps != nullptr && lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) a.stamps[256 * 32 + (blockIdx.x * 8 + wave) * 8 + k] = fs[k];
    }
#endif
}

// f16 packed weights (math = 2): unit ((cc * T + t) * 2 + q) * Mpad + m = the 8 channels cc * 16 + q * 8 + j of tap t, row m, as f16 (one plane:
// half the bytes of the bf16 (hi, lo) operand).  Job table as vd_conv3_pack_weights_multi: {src, dst, M, C, row_stride, chan_stride, first block,
// taps}; one thread per (m, chunk, q), lanes along m.
template <int T>
__device__ __forceinline__ void pack_f16_one(const float* __restrict__ src, u32x4* __restrict__ dst, int M, int C, int Mpad, int64_t rs, int64_t cs,
                                             int local) {
    if (local >= Mpad * (C / 16) * 2) return;
    const int m = local % Mpad;
    const int rest = local / Mpad;
    const int q = rest & 1, cc = rest >> 1;
    const float* __restrict__ s0 = src + (int64_t)(m < M ? m : 0) * rs + (int64_t)(cc * 16 + q * 8) * cs;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = (m < M) ? s0[(int64_t)j * cs + t] : 0.f;
        dst[((int64_t)(cc * T + t) * 2 + q) * Mpad + m] = to_f16x8(w);
    }
}

__global__ __launch_bounds__(256) void pack_f16_multi_kernel(const int64_t* __restrict__ table, int n_jobs) {
    int lo = 0, hi = n_jobs - 1;
    const int64_t blk = blockIdx.x;
    while (lo < hi) {                                            // last job whose first block <= blk (block-uniform)
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 8 + 6] <= blk) lo = mid;
        else hi = mid - 1;
    }
    const int64_t* __restrict__ jb = table + lo * 8;
    const int M = (int)jb[2], C = (int)jb[3];
    const int local = (int)(blk - jb[6]) * 256 + threadIdx.x;
    if (jb[7] == 1)
        pack_f16_one<1>(reinterpret_cast<const float*>(jb[0]), reinterpret_cast<u32x4*>(jb[1]), M, C, (M + 127) / 128 * 128, jb[4], jb[5], local);
    else
        pack_f16_one<9>(reinterpret_cast<const float*>(jb[0]), reinterpret_cast<u32x4*>(jb[1]), M, C, (M + 127) / 128 * 128, jb[4], jb[5], local);
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

}  // namespace

extern "C" int vd_conv3_pack_weights_f16_multi(const int64_t* table, int n_jobs, int64_t total_blocks, void* stream) {
    VD_REQUIRE(table && n_jobs > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "vd_conv3_pack_weights_f16_multi: bad arguments");
    hipLaunchKernelGGL(pack_f16_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, table, n_jobs);
    VD_LAUNCH_CHECK("vd_conv3_pack_weights_f16_multi");
    return 0;
}

// What launch_bx3 (vd_gemm.hip) asks: can the persistent 16x16x32 kernel take this problem?  Output tiles of 256 pixels (16 x 16, or 8 rows x 32
// columns of any image whose sides divide), whole chunk pairs, aligned float4 epilogue, and enough tiles to give every CU at least one.
bool vd_conv3_k32p_eligible(const vd_gemm_desc& d) {
    static const int off = env_int("VD_K32P_OFF", 0);
    if (off || d.C % 32 != 0 || d.bias_on_n || d.d_trans || d.nb2 > 1) return false;
    if (d.b_presplit && (d.b_presplit != 1 || d.gn_ss || d.math == 2 || (d.b_bstride & 3) || (((uintptr_t)d.B) & 15))) return false;
    const int TW = d.OW == 16 ? 16 : 32, TR = 256 / TW;
    if (d.OW % TW != 0 || d.OH % TR != 0 || d.OH * d.OW != d.NP) return false;
    if (d.OW == 16 && d.OH != 16) return false;
    if (d.b_mode == VD_B_CONV3_UP && (d.H * 2 != d.OH || d.W * 2 != d.OW)) return false;
    if (d.gn_ss && d.b_mode != VD_B_CONV3) return false;
    if (d.act_out && (!d.gn_ss || (d.act_bstride & 3) || d.math == 2)) return false;
    const int ldd_mult = d.pool2 ? 2 : 4;
    if ((d.ldd % ldd_mult) || (d.d_bstride % ldd_mult) || (((uintptr_t)d.D) & (4 * ldd_mult - 1))) return false;
    if (d.residual && ((d.res_bstride & 3) || (((uintptr_t)d.residual) & 15))) return false;
    if (d.pool2 && ((d.OW & 1) || (d.OH & 1))) return false;
    if ((int64_t)d.C * d.H * d.W * 4 >= (1ll << 32)) return false;           // 32-bit buffer offsets inside one image
    return true;
}

int vd_launch_conv3_k32p(const vd_gemm_desc& d, int mode, hipStream_t st) {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return -1;
        n_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
        n_cu &= ~7;                                               // whole XCD octets
        if (n_cu < 8) n_cu = 8;
    }
#ifdef VD_K32P_VARIANTS
    static const int flags = env_int("VD_K32P_FLAGS", 0);       // (bit 0: no stagger of the wave halves -- register-staged variants only)
#else
    constexpr int flags = 0;
#endif
#ifdef VD_K32P_VARIANTS
    static const int dma = env_int("VD_K32P_DMA", 1);
    static const int pipe = env_int("VD_K32P_PIPE", 1);
#endif
    k32p_args a;
    a.d = d;
    const int TW = d.OW == 16 ? 16 : 32, TR = 256 / TW;
    a.tiles_m = vd_cdiv(d.M, 128);
    a.tiles_x = d.OW / TW;
    a.tiles_img = a.tiles_x * (d.OH / TR);
    a.n_tiles = a.tiles_m * (d.N / d.NP) * a.tiles_img;
    a.flags = flags;
    a.stamps = nullptr;
#ifdef VD_K32P_STAMPS
    a.stamps = reinterpret_cast<unsigned long long*>(d.ws);       // the diagnostic build borrows the (unused) split-K workspace pointer
#endif
    int grid = a.n_tiles < n_cu ? ((a.n_tiles + 7) & ~7) : n_cu;
    // Shipped: LDS-DMA weight stages + hand-pipelined fragment reads (profiles/r04_k32p_ab.txt: +5-10 % per kernel over round 3's kernel at 16x16 /
    // 32x32, +20-28 % on the wide levels; the register-staged and the compiler-scheduled variants were neutral / slower).  -DVD_K32P_VARIANTS
    // builds the other three for A/B runs (VD_K32P_DMA=0 / VD_K32P_PIPE=0).
#ifdef VD_K32P_VARIANTS
#define VD_K32P_CASE(WW, MD)                                                                                         \
    if (TW == WW && mode == MD) {                                                                                    \
        if (dma && pipe) hipLaunchKernelGGL((conv3_k32p_kernel<WW, MD, true, true, false>), dim3(grid), dim3(512), 0, st, a);        \
        else if (dma) hipLaunchKernelGGL((conv3_k32p_kernel<WW, MD, true, false, false>), dim3(grid), dim3(512), 0, st, a);          \
        else if (pipe) hipLaunchKernelGGL((conv3_k32p_kernel<WW, MD, false, true, false>), dim3(grid), dim3(512), 0, st, a);         \
        else hipLaunchKernelGGL((conv3_k32p_kernel<WW, MD, false, false, false>), dim3(grid), dim3(512), 0, st, a);                  \
        return 0;                                                                                                    \
    }
#else
#define VD_K32P_CASE(WW, MD)                                                                                         \
    if (TW == WW && mode == MD) {                                                                                    \
        if (d.math == 2) hipLaunchKernelGGL((conv3_k32p_kernel<WW, MD, true, true, true>), dim3(grid), dim3(512), 0, st, a);   \
        else hipLaunchKernelGGL((conv3_k32p_kernel<WW, MD, true, true, false>), dim3(grid), dim3(512), 0, st, a);    \
        return 0;                                                                                                    \
    }
#endif
#define VD_K32P_PS(WW, MD)                                                                                           \
    if (d.b_presplit && TW == WW && mode == MD) {                                                                    \
        hipLaunchKernelGGL((conv3_k32p_kernel<WW, MD, true, true, false, true>), dim3(grid), dim3(512), 0, st, a);   \
        return 0;                                                                                                    \
    }
    VD_K32P_PS(32, 0) VD_K32P_PS(32, 1) VD_K32P_PS(32, 2) VD_K32P_PS(16, 0) VD_K32P_PS(16, 1) VD_K32P_PS(16, 2)
#undef VD_K32P_PS
    if (d.b_presplit) return -1;
    VD_K32P_CASE(32, 0) VD_K32P_CASE(32, 1) VD_K32P_CASE(32, 2) VD_K32P_CASE(32, 3)
    VD_K32P_CASE(16, 0) VD_K32P_CASE(16, 1) VD_K32P_CASE(16, 2) VD_K32P_CASE(16, 3)
#undef VD_K32P_CASE
    return -1;
}
