// ---- split-precision 3x3 convolution for the 8x8 / 4x4 levels: whole channel loop per workgroup, no split-K (round 6) ---------------------------
// The 8x8 / 4x4 layers of the DDPM UNets (256 channels, B = 128: 8192 / 2048 output pixels) do not fill the chip with 128 x 256 (or 128 x 128) tiles:
// rounds 2-5 split the channel loop over workgroups instead (conv3_bx3_kernel<8 | 4, ..>: 4 - 8 partial slabs per tile + a splitk_epilogue launch
// per convolution), which made these launches 47-51 us (8x8) and 24-27 us (4x4) plus 7 us of epilogue for 21 / 5 us of matrix work, fetched 2.6-5.6 x
// their algorithmic bytes and put 65 reduction launches into every training step (profiles/r05_pmc_traffic.json, r05_bench_detail.json).  Here the
// tile shrinks until M x N alone covers the chip and a workgroup keeps the WHOLE K:
//     64 output channels x IMGS whole images (8x8: 2 images = 128 pixels -> 4 x 64 = 256 tiles at M = 256, B = 128;  4x4: 4 images = 64 pixels -> 128 tiles),
//     four waves of 32 channels x (64 | 32) pixels on v_mfma_f32_16x16x32_bf16, the stage pipeline of conv3_k32p_kernel (vd_conv_k32p.hip): a stage = one
//     tap row of a chunk pair (32 input channels), packed weights global -> LDS by LDS-DMA two buffers deep (24 KB per stage), the halo patch of the
//     chunk pair (IMGS x (W + 2)^2 pixels, f32 -> bf16 (hi, lo) where it is written) loaded one chunk pair ahead, hand-pipelined fragment reads.
// Same arithmetic as every split-precision kernel (x = hi + lo, three MFMAs per product term, f32 accumulation); the order of the additions differs
// from the split kernels' (one accumulation chain over all channels instead of 4 - 8 partial sums), so results agree to the path's tolerance, not
// bit for bit.  Forward (VD_B_CONV3) and flipped-tap input gradient (VD_B_CONV3_T: the caller passes the transposed packed operand), f32 NCHW input.
// Reference work replaced: diffusers ResnetBlock2D conv1 / conv2 and their input gradients at the two coarsest levels (reference loss.py:993).
#include "vd_common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
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

struct sm_args {
    vd_gemm_desc d;
    int n_tiles;      // tiles_m * tiles_n
    int tiles_n;      // ceil(B / IMGS)
    int nb;           // images
#ifdef VD_SM_STAMPS
    unsigned long long* stamps;   // diagnostic build (tools/build_k32p_diag.sh) only: per-wave segment sums in shader cycles, [grid][4 waves][8]
#endif
};

// KH: chunk pairs (32 input channels each) per stage.  Shipped: 1 (KH = 2 halves the number of barriers and was measured 3-25 % slower).
template <int W, int MODE, int IMGS, int KH>   // MODE 0: VD_B_CONV3, 1: VD_B_CONV3_T (flipped taps)
__global__ __launch_bounds__(256, 2) void conv3_sm_kernel(const sm_args a) {
    const vd_gemm_desc& d = a.d;
    constexpr int BM = 64, NTH = 256, HW = W * W, NPIX = IMGS * HW;
    constexpr int RUNS = 12;                                      // 2 KB (128-row) runs of one chunk's tap row in the packed operand: (s, part, q)
    constexpr int PW = W + 2, PIMG1 = PW * PW, PTOT = IMGS * PIMG1;
    constexpr int PLANE = (PTOT + 15) / 16 * 16;
    constexpr int A_HALF = RUNS * BM;                             // 768 units: one tap row of ONE chunk, 64 rows
    constexpr int A_PAIR = 2 * A_HALF;                            // 1536 units = 24 KB: one tap row of a chunk pair
    constexpr int A_UNITS = KH * A_PAIR;                          // per buffer
    constexpr int A_IT = A_UNITS / NTH;                           // 6 KH LDS-DMA pieces of 1 KB per wave and stage
    constexpr int NI = NPIX / 32;                                 // pixel tiles of 16 per wave: 4 | 2
    static_assert(NPIX % 32 == 0 && 16 % W == 0 && HW % 16 == 0, "16-pixel fragments are whole rows of one image");
    static_assert(KH == 1 || KH == 2, "one or two chunk pairs per stage");
    // ONE LDS object (a second __shared__ array beside an LDS-DMA target makes hipcc wait vmcnt(0) before every ds_read: vd_conv_k32p.hip)
    __shared__ u32x4 lds[2 * A_UNITS + KH * 8 * PLANE];
    u32x4* const As = lds;
    u32x4* const Ps = lds + 2 * A_UNITS;

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef VD_SM_STAMPS      // per-wave segment sums: [issue, patch convert, mfma, patch switch, vmcnt wait, barrier, epilogue, total]; every tick is an s_memtime round trip
    unsigned long long fs[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}, ft_prev = 0ull, ft_first = 0ull;
    auto ftick = [&](int k) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        if (k >= 0) fs[k] += t - ft_prev;
        else ft_first = t;
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

    // tile: the workgroups of an XCD (blockIdx % 8 under round-robin dispatch: speed only) take a contiguous range of the m-major tile order, i.e. few m-tiles
    int t = blockIdx.x;
    if ((gridDim.x & 7) == 0) t = (t & 7) * (gridDim.x >> 3) + (t >> 3);
    const int tm = t / a.tiles_n, tn = t - tm * a.tiles_n;
    const int m0 = tm * BM, b0 = tn * IMGS;

    const u32x4* __restrict__ Apk = reinterpret_cast<const u32x4*>(d.a_packed);
    const int Mpad = d.a_packed_mpad;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(Apk), 0, 0xFFFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.B + (int64_t)b0 * d.b_bstride), 0, 0xFFFFFFF0, 0x00020000);

    // ---- weights: piece i of a stage = run 4 i + wave (c2 * 12 + (s, part, q)), rows m0 .. m0 + 63: 64 consecutive 16-byte units, lane-linear ----
    unsigned aoff[12];            // (A_IT <= 12; a constant size: with `aoff[A_IT]`, A_IT depending on KH, hipcc 7.2 emits no host stub for the template -- vd_conv_k32p.hip)
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int run = 4 * i + wave;                             // 0 .. 24 KH - 1 = (chunk pair kh, c2, (s, part, q)); chunk c2 = 1 is 36 runs further, pair kh = 1: 72
        aoff[i] = 16u * (unsigned)(((run % RUNS) + 3 * RUNS * ((run % 24) / RUNS) + 6 * RUNS * (run / 24)) * Mpad + lane);
    }
    auto load_a = [&](int cg, int r, int buf) {                  // stage (chunk group cg = KH pairs, tap row r)
        const unsigned so = 16u * (unsigned)((cg * KH * 6 * RUNS + r * RUNS) * Mpad + m0);     // wave-uniform
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(As + buf * A_UNITS + (4 * i + wave) * 64);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, dst, 16, aoff[i], so, 0, 0);
        }
    };

    // ---- patch: wave cq stages k-octet cq of a chunk pair.  A lane loads FOUR consecutive pixels of an image row for each of the octet's 8 channels (8 x
    // 16-byte loads: W = 8: item (image, row, half row) on lanes 0-31; W = 4: item (image, row) on lanes 0-15) and writes those four pixels' (hi, lo) units.
    // First version: one dword per (pixel, channel) like conv3_k32p_kernel -- 32 | 24 wave instructions per wave and chunk pair beside 18 | 9 LDS-DMA pieces,
    // and the CU's vector-memory address path (one wave instruction per ~23 cycles) took longer than the stage's MFMAs.  The halo ring of every image
    // is zeroed once: only interior pixels are ever written.
    const int cq = wave;
    constexpr int P_LANES = IMGS * W * (W / 4);                   // 32 | 16 items per octet
    const bool p_act = lane < P_LANES;
    const int p_xh = (W == 8) ? (lane & 1) : 0, p_y = (W == 8) ? ((lane >> 1) & 7) : (lane & 3), p_img = (W == 8) ? ((lane >> 4) & (IMGS - 1)) : ((lane >> 2) & 3);
    const bool p_ok = p_act && b0 + p_img < a.nb;
    const unsigned poff = p_ok ? 4u * (unsigned)((int64_t)p_img * d.b_bstride + cq * 8 * HW + p_y * W + p_xh * 4) : 0xFFFFFFFFu;      // out of range: zeros
    const int pdst = ((cq >> 1) * 4 + (cq & 1)) * PLANE + p_img * PIMG1 + (p_y + 1) * PW + p_xh * 4 + 1;      // Ps[kh][c2][part = 0][q][pixel]; lo: + 2 PLANE
    constexpr int P_LOADS = 8;                                    // vector-memory loads per wave and chunk pair
    f32x4 rp[KH][8];
    u32x4 cph[KH][4], cpl[KH][4];
    auto load_p = [&](int cg, int kh) {                           // chunk pair kh of chunk group cg
        if (p_act) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned so = 4u * (unsigned)(((cg * KH + kh) * 32 + j) * HW);            // wave-uniform
                rp[kh][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, poff, so, 0));
            }
        }
    };
    auto convert_p = [&]() {
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = rp[kh][j][k];
                split8(v, cph[kh][k], cpl[kh][k]);
            }
    };
    auto write_p = [&]() {
        if (p_act) {
#pragma unroll
            for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    Ps[kh * 8 * PLANE + pdst + k] = cph[kh][k];
                    Ps[kh * 8 * PLANE + pdst + k + 2 * PLANE] = cpl[kh][k];
                }
        }
    };
    for (int i = tid; i < KH * 8 * PLANE; i += NTH) Ps[i] = u32x4{0u, 0u, 0u, 0u};       // (the halo ring; made visible by the barrier in front of the first write_p)

    // ---- fragments: wave (wm, wn) owns channels wm * 32 .. + 31 and pixels wn * NPIX / 2 .. ; 16 consecutive tile pixels = whole rows of ONE image ----
    f32x4 acc[NI][2];
    const int wm = wave >> 1, wn = wave & 1;
    const int c2 = g >> 1, q = g & 1;
    const u32x4* __restrict__ a_base = As + c2 * A_HALF + q * BM + wm * 32 + l15;
    const u32x4* __restrict__ p_base[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int qx = wn * (NPIX / 2) + ni * 16 + l15;
        const int img = qx / HW, rr = qx - img * HW;
        p_base[ni] = Ps + (c2 * 4 + q) * PLANE + img * PIMG1 + (rr / W) * PW + (rr % W);
    }

    // one tap row of one chunk pair: 3 taps x NI pixel tiles x 2 channel tiles x 3 products.  One wave per SIMD: nothing but this wave's own reads-ahead
    // hides the LDS round trip, and a pixel tile is only 6 MFMAs (96 cycles) -- so pixel pairs are read XD tiles ahead (ring of XD + 1 register pairs)
    // and the NEXT tap's four weight fragments into a second register set at the start of the current tap (first version: one tile of lead and the
    // weights replaced inside the tap's last tile: 37 cycles per MFMA).
    constexpr int XD = 3, XR = XD + 1, NT3 = 3 * NI;
    auto mfma_row = [&](int r, int buf, int kh) {
        const int pr = (MODE == 1) ? 2 - r : r;
        const u32x4* __restrict__ a_cur = a_base + buf * A_UNITS + kh * A_PAIR;
        const int pko = kh * 8 * PLANE;                           // chunk pair kh of the patch
        bf16x8 wh[2][2], wl[2][2], xh[XR], xl[XR];
        auto tap_col = [&](int s) { return (MODE == 1) ? 2 - s : s; };
        auto load_w = [&](int s, int set) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                wh[set][mi] = __builtin_bit_cast(bf16x8, a_cur[(s * 4 + 0) * BM + mi * 16]);
                wl[set][mi] = __builtin_bit_cast(bf16x8, a_cur[(s * 4 + 2) * BM + mi * 16]);
            }
        };
        auto load_x = [&](int k) {                               // pixel tile k = s * NI + ni of the stage
            const int s = k / NI, ni = k - s * NI;
            xh[k % XR] = __builtin_bit_cast(bf16x8, p_base[ni][pko + pr * PW + tap_col(s)]);
            xl[k % XR] = __builtin_bit_cast(bf16x8, p_base[ni][pko + 2 * PLANE + pr * PW + tap_col(s)]);
        };
        load_w(0, 0);
#pragma unroll
        for (int k = 0; k < XD; ++k) load_x(k);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < NT3; ++k) {
            const int s = k / NI, ni = k - s * NI, cur = k % XR, ws = s & 1;
            if (k + XD < NT3) {
                load_x(k + XD);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            if (ni == 0 && s < 2) {
                load_w(s + 1, ws ^ 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[cur], wl[ws][mi], acc[ni][mi], 0, 0, 0);
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[cur], wh[ws][mi], acc[ni][mi], 0, 0, 0);
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[cur], wh[ws][mi], acc[ni][mi], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- the stage pipeline (vd_conv_k32p.hip, LDS-DMA form): stage = (chunk group cg of KH pairs, tap row r); the DMA of stage s + 1 goes out before the
    // MFMAs of stage s; the patch of group cg + 1 is loaded during (cg, 0 .. KH - 1) -- one chunk pair per stage --, converted in front of (cg, 2) and
    // written behind it ----
    const int ngroups = d.C / (32 * KH);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    load_a(0, 0, 0);
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) load_p(0, kh);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    convert_p();
    __syncthreads();                                              // the zeroed patch is complete
    write_p();
    __syncthreads();
    int st = 0;
    VD_FTICK(-1);
    for (int cg = 0; cg < ngroups; ++cg) {
        const bool more = cg + 1 < ngroups;
#pragma unroll
        for (int r = 0; r < 3; ++r, ++st) {
            const int buf = st & 1;
            const bool ex1 = r < 2 || more;                       // a stage follows
            if (ex1) load_a(r < 2 ? cg : cg + 1, r < 2 ? r + 1 : 0, buf ^ 1);      // As[buf ^ 1] was last read before the barrier behind this wave
            const bool ld = more && (KH == 2 ? r < 2 : r == 1);   // this stage loads one chunk pair of the next group's patch
            if (ld) load_p(cg + 1, KH == 2 ? r : 0);
            __builtin_amdgcn_sched_barrier(0);
            VD_FTICK(0);
            if (r == 2 && more) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_IT) : "memory");       // the patch loads (older than this stage's DMAs)
                convert_p();
            }
            VD_FTICK(1);
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) mfma_row(r, buf, kh);
            VD_FTICK(2);
            if (r == 2 && more) {
                __syncthreads();                                  // every wave has finished reading the patch
                write_p();
            }
            VD_FTICK(3);
            // this stage's DMAs (issued before any patch load of this stage) must have landed before the barrier that releases the next stage
            if (ld) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P_LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            VD_FTICK(4);
            __syncthreads();
            VD_FTICK(5);
        }
    }

    // ---- epilogue: lane (g, l15) holds tile pixels wn * NPIX / 2 + ni * 16 + g * 4 + {0..3} (one row segment of one image) of channel m0 + wm * 32 + mi * 16 + l15 ----
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + wm * 32 + mi * 16 + l15;
        const int mc = m < d.M ? m : d.M - 1;
        const float badd = d.bias != nullptr ? d.bias[mc] : 0.f;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int qx = wn * (NPIX / 2) + ni * 16 + g * 4;
            const int img = qx / HW, p = qx - img * HW;
            const int b = b0 + img;
            if (b >= a.nb || m >= d.M) continue;
            float add = badd;
            if (d.rowadd != nullptr) add += d.rowadd[(int64_t)b * d.rowadd_bstride + mc];
            f32x4 val = d.alpha * acc[ni][mi] + add;
            if (d.residual != nullptr) val += *reinterpret_cast<const f32x4*>(d.residual + (int64_t)b * d.res_bstride + (int64_t)mc * d.ldd + p);
            float* __restrict__ dst = d.D + (int64_t)b * d.d_bstride + (int64_t)mc * d.ldd + p;
            if (d.accumulate) val += *reinterpret_cast<const f32x4*>(dst);
            *reinterpret_cast<f32x4*>(dst) = val;
        }
    }
#ifdef VD_SM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VD_FTICK(6);
    if (a.stamps != nullptr && lane == 0) {
        fs[7] = ft_prev - ft_first;
#pragma unroll
        for (int k = 0; k < 8; ++k) a.stamps[(blockIdx.x * 4 + wave) * 8 + k] = fs[k];
    }
#endif
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

}  // namespace

// vd_gemm.hip asks: is this one of the 8x8 / 4x4 stride-1 problems the whole-K kernel takes?  (vd_gemm_tile() == 20)
bool vd_conv3_sm_eligible(const vd_gemm_desc& d) {
    static const int off = env_int("VD_CONV_SM_OFF", 0);          // 1: rounds 2-5's split-K kernels (A/B switch)
    if (off || !d.a_packed || d.math == 2 || d.b_presplit) return false;
    if (d.b_mode != VD_B_CONV3 && d.b_mode != VD_B_CONV3_T) return false;
    if (d.OH != d.OW || (d.OW != 8 && d.OW != 4) || d.H != d.OH || d.W != d.OW || d.OH * d.OW != d.NP) return false;
    if (d.C % 32 != 0 || d.K != d.C * 9 || d.M < 64 || d.a_packed_mpad < d.M || (d.a_packed_mpad & 127)) return false;
    if (d.bias_on_n || d.d_trans || d.nb2 > 1 || d.gn_ss || d.gn_part || d.act_out || d.pool2 || d.debug || d.tile || d.act) return false;
    if ((d.ldd & 3) || (d.d_bstride & 3) || (((uintptr_t)d.D) & 15) || (((uintptr_t)d.a_packed) & 15)) return false;
    if (d.residual && ((d.res_bstride & 3) || (((uintptr_t)d.residual) & 15))) return false;
    if ((d.b_bstride & 3) || (((uintptr_t)d.B) & 15)) return false;                // 16-byte row loads
    const int imgs = d.OW == 8 ? 2 : 4;
    if ((int64_t)imgs * d.b_bstride * 4 >= (1ll << 31)) return false;             // 32-bit buffer offsets inside one tile's images
    const int nb = d.N / d.NP;
    // 8x8: from 64 tiles on; 4x4 (36 MFMAs per wave and stage behind the same 24 KB of weights): only where the grid fills the chip (M = 512 at B = 128) --
    // below that the split kernels are as fast or faster; two-image tiles (256 tiles at M = 256) win alone and lose inside the step (profiles/r06_conv_sm_ab.txt)
    return vd_cdiv(d.M, 64) * vd_cdiv(nb, imgs) >= (d.OW == 8 ? 64 : 256);
}

int vd_launch_conv3_sm(const vd_gemm_desc& d, hipStream_t st) {
    sm_args a;
    a.d = d;
#ifdef VD_SM_STAMPS
    a.stamps = reinterpret_cast<unsigned long long*>(d.ws);       // the diagnostic build borrows the (unused) split-K workspace pointer
#endif
    a.nb = d.N / d.NP;
    const int imgs = d.OW == 8 ? 2 : 4;
    a.tiles_n = vd_cdiv(a.nb, imgs);
    a.n_tiles = vd_cdiv(d.M, 64) * a.tiles_n;
    const int mode = d.b_mode == VD_B_CONV3_T ? 1 : 0;
    // (KH = 2 -- two chunk pairs per stage, half the barriers -- and one 8x8 image per tile -- 512 tiles, two workgroups per CU -- measured slower:
    // profiles/r06_conv_sm_ab.txt)
#define VD_SM_CASE(WW, MD, IM)                                                                                   \
    if (d.OW == WW && mode == MD && imgs == IM) {                                                                \
        hipLaunchKernelGGL((conv3_sm_kernel<WW, MD, IM, 1>), dim3(a.n_tiles), dim3(256), 0, st, a);              \
        return 0;                                                                                                \
    }
    VD_SM_CASE(8, 0, 2) VD_SM_CASE(8, 1, 2) VD_SM_CASE(4, 0, 4) VD_SM_CASE(4, 1, 4)
#undef VD_SM_CASE
    return -1;
}
