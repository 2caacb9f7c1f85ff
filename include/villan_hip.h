/*
 * villan_hip.h -- C ABI of the MI355X (gfx950) backdoored-diffusion hot path.
 *
 * The reference (IBM/VillanDiffusion) has no FFI: its hot path is Python calling
 * torch/cuDNN through an un-vendored diffusers fork.  This header is therefore the
 * boundary *below* the reference's Python surface (last row): each
 * entry point replaces the torch op sequence named in its comment (reference
 * file:line of the call site that triggers it).  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory (fp32 unless
 *     stated), owned by the caller, 16-byte aligned;
 *   - image tensors are NCHW with contiguous CHW and an explicit batch stride
 *     ("bstride", in elements) so channel-slices of a wider buffer can be passed
 *     without a copy (zero-copy skip concatenation);
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*), never
 *     synchronise, allocate nothing; thread-safe w.r.t. distinct streams;
 *   - return 0 on success, otherwise a hipError_t / negative VD_E* code;
 *     vd_last_error() gives a message.  Nothing throws.
 */
#ifndef VILLAN_HIP_H
#define VILLAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VD_ABI_VERSION 11
#define VD_EINVAL (-22)
#define VD_ETIMEDOUT (-110) /* an EARLIER asynchronous launch reported a bounded-poll timeout (see vd_async_errors) */

int vd_abi_version(void);
const char* vd_last_error(void);
/* 0 when a gfx950-capable device is visible to this process. */
int vd_device_ok(void);
/* Errors that kernels report asynchronously through a word of pinned host memory (today: poll timeouts of the one-launch chunked
 * GroupNorm, whose statistics are then NaN).  Returns the count since the last clear without synchronising; while it is non-zero every
 * vd_groupnorm_* call fails with VD_ETIMEDOUT (sticky).  clear != 0 resets it. */
int vd_async_errors(int clear);

/* ------------------------------------------------------------------------------------------
 * K2/K4/K5/K6/K7 -- implicit-GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32, exact f32).
 *   D[b][m][p] = alpha * sum_k A[m][k] * Bop[k][(b,p)]  (+bias +rowadd +residual)
 * Replaces F.conv2d / F.linear / torch.bmm inside UNet2DModel (reached via loss.py:993 and
 * the pipeline call VillanDiffusion.py:579, model.py:519).
 * ------------------------------------------------------------------------------------------ */
enum vd_a_mode { VD_A_ROW = 0,      /* A[m*lda + k]  (k contiguous: conv/linear weights)      */
                 VD_A_COL = 1 };    /* A[k*lda + m]  (m contiguous)                            */
enum vd_b_mode { VD_B_PLAIN = 0,    /* B[b*bs + k*ldb + p]   (1x1 conv, plain matrices)        */
                 VD_B_KCONTIG = 1,  /* B[b*bs + p*ldb + k]                                     */
                 VD_B_CONV3 = 2,    /* 3x3 pad 1 gather, k = c*9 + r*3 + s                     */
                 VD_B_CONV3_T = 3,  /* 3x3 pad 1 with flipped taps (stride-1 dgrad)            */
                 VD_B_CONV3_S2 = 4, /* pad (0,1,0,1) + 3x3 stride 2 (Downsample2D)             */
                 VD_B_CONV3_UP = 5, /* nearest x2 folded into the 3x3 pad 1 gather (Upsample2D) */
                 VD_B_CONV3_DIL = 6,/* dgrad of CONV3_S2 (zero-dilated gather)                 */
                 VD_B_CONVG = 7     /* general kh x kw convolution, stride conv_stride, zero padding (pad_h, pad_w),
                                       k = (c*kh + r)*kw + s: the InceptionV3 convolutions of the FID measure
                                       (1x7 / 7x1 / 5x5 / 3x3 stride 2; reference fid_score.py:91-148 -> pytorch-fid) */ };

typedef struct vd_gemm_desc {
    const float* A;
    const float* B;
    float* D;
    const float* bias;       /* [M] (or [N] when bias_on_n), nullable                           */
    const float* rowadd;     /* rowadd[b*rowadd_bstride + m] broadcast over p (temb), nullable  */
    const float* residual;   /* residual[b*res_bstride + m*ldd + p], nullable                   */
    int32_t M, N, K;         /* N = nb * NP                                                     */
    int32_t a_mode, b_mode;
    int32_t NP;              /* columns per batch item (OH*OW for conv)                         */
    int32_t C, H, W;         /* conv source dims (per batch item)                               */
    int32_t OH, OW;          /* conv output dims                                                */
    int32_t bias_on_n;
    int32_t d_trans;         /* 1: D[n*ldd + m] (n-major store)                                 */
    int32_t accumulate;      /* 1: D += result                                                  */
    int32_t tile;            /* 0 auto, 1: 128x128, 2: 64x128, 3: 64x64                         */
    int32_t debug;           /* MUST be 0: the release library fails with VD_EINVAL otherwise (ABI 11).  Only a diagnostic
                                `make ABLATION=1` build honours timing-only ablation bits here (results invalid): 1 = no
                                global loads after the first K-step, 2 = no epilogue, 4 = no MFMA, 8 = no LDS stores      */
    float alpha;
    int64_t lda, a_bstride;  /* a_bstride != 0: per-batch A (requires tile_n | NP)              */
    int64_t ldb, b_bstride;
    int64_t ldd, d_bstride;
    int64_t res_bstride, rowadd_bstride;
    float* ws;               /* split-K workspace (vd_gemm_ws_floats() floats), nullable when that is 0        */
    int32_t pad;             /* VD_B_CONV3_S2 only: 0 = zero pad (0,1,0,1) (Downsample2D padding=0, the DDPM UNets),
                                1 = symmetric padding 1 (Downsample2D padding=1, the LDM / NCSN++ UNets)        */
    int32_t nb2;             /* > 1: two-level batch, item i = outer*nb2 + inner at outer*X_bstride + inner*X_b2stride
                                (the heads of multi-head attention are channel slices of one q/k/v tensor); needs
                                rowadd == residual == NULL                                                       */
    int64_t a_b2stride, b_b2stride, d_b2stride;
    const float* gn_ss;      /* VD_B_CONV3 only, nullable: per-(batch item, input channel) GroupNorm scale / shift pairs
                                [nb][C][2] from vd_groupnorm_stats; the convolution then reads silu(x*scale + shift) instead
                                of x (GroupNorm + SiLU folded into the patch loader: inference path, nothing is saved).
                                Needs the patch-staged kernel (OW >= 16, C % 8 == 0, C <= 1024, M >= 64)              */
    const void* a_packed;    /* nullable.  3x3 convolutions (VD_B_CONV3 / _T at 4x4 ... 32x32, _UP at 8x8 ... 32x32, any of them on 64-wide or
                                128k-wide outputs; C % 16 == 0,
                                M >= 64) and VD_B_PLAIN products with a shared A (1x1 convolutions: NP % 128 == 0 or 128 % NP == 0, N % 128 == 0, K % 16 == 0,
                                M >= 64) only: the weights pre-split into bf16 (hi, lo) pairs by vd_conv3_pack_weights.  The
                                contraction then runs as three bf16 MFMAs per product term (hi*hi + hi*lo + lo*hi, f32
                                accumulation; ~1e-5 relative to the exact-f32 kernel) instead of on the f32 MFMA.  A is still
                                required (shape checks) but not read.  Problems outside that set fail with VD_EINVAL.   */
    int32_t a_packed_mpad;   /* row count the packed operand was built with (M rounded up to 128)                      */
    int32_t math;            /* 2: a_packed holds f16 operands (vd_conv3_pack_weights_f16_multi): ONE f16 MFMA per product term, f32 accumulation
                                (~2e-4 relative per contraction) -- only problems the persistent 16x16x32 kernels take (vd_gemm_tile() 18 / 19),
                                else VD_EINVAL.
                                0: exact f32 MFMA (or a_packed).  1: split-precision product of two ACTIVATION matrices (attention
                                scores / values and their gradients): per-batch A (a_bstride != 0), VD_B_PLAIN or VD_B_KCONTIG,
                                NP % 128 == 0, K % 16 == 0, K >= 32, M >= 64, 16-byte aligned operands and strides; both operands
                                are split into bf16 (hi, lo) inside the kernel.  Anything else fails with VD_EINVAL.           */
    int32_t pool2;           /* 1: VD_B_CONV3_T with a_packed at 16x16 / 32x32 outputs only (the input gradient of an Upsample2D convolution):
                                the epilogue adds each 2x2 block of output pixels and writes D at HALF resolution (ldd = (OH/2)*(OW/2)),
                                i.e. conv-transpose followed by the adjoint of the nearest-2x upsample, without the full-resolution tensor.
                                Needs an unsplit grid (M/128 * N/128 >= 256 tiles), no bias / rowadd / residual / accumulate; else VD_EINVAL */
    int32_t kh, kw;          /* VD_B_CONVG only: kernel height / width (K = C*kh*kw)                                    */
    int32_t conv_stride;     /* VD_B_CONVG only: 1 .. 4 (both directions)                                               */
    int32_t pad_h, pad_w;    /* VD_B_CONVG only: zero padding on each side                                              */
    int32_t act;             /* 0: none; 1: D = max(D, 0) after every other epilogue term (BasicConv2d = conv + folded BatchNorm + ReLU).
                                Honoured by the exact-f32 gather / plain kernels only (a_packed == NULL, math == 0); else VD_EINVAL */
    float* gn_part;          /* optional OUTPUT of the 16x16x32 split-precision 3x3 convolution (vd_gemm_tile() == 17, else VD_EINVAL):
                                [B][NP / 256][M][2] = (sum, sum of squares) of the FINAL result (after bias / rowadd / residual) of channel m
                                over each 256-pixel tile -- the GroupNorm that follows (ResnetBlock2D: conv1 -> norm2) gets its statistics
                                from vd_groupnorm_stats_from_partials() instead of a read of the whole tensor.  Fixed-order sums.           */
    float* act_out;          /* optional OUTPUT of the persistent 16x16x32 3x3 convolution with gn_ss (vd_gemm_tile() == 18, else VD_EINVAL): the
                                normalised activation silu(x * scale + shift) the loader computes anyway, written once per element
                                ([nb][C][H][W], batch stride act_bstride) by the workgroups of the first channel tile.  The TRAINING forward
                                saves it for the weight gradient instead of running a separate GroupNorm + SiLU pass (round 4).           */
    int64_t act_bstride;
    int32_t b_presplit;      /* ABI 11.  1: B is a PRE-SPLIT image (vd_presplit_* below: bf16 (hi, lo) pairs, 8 channels of a pixel per 16-byte
                                unit) instead of f32 NCHW -- same bytes, same batch stride, same channel-octet offsets.  Only the persistent
                                16x16x32 convolution (vd_gemm_tile() == 18) with a_packed, math 0, VD_B_CONV3 / VD_B_CONV3_T and no gn_ss takes
                                it: the patch loader then moves (hi, lo) units without converting.  Anything else: VD_EINVAL.      */
    int32_t reserved_;
} vd_gemm_desc;

int vd_gemm(const vd_gemm_desc* desc, void* stream);
/* Floats of split-K workspace vd_gemm needs for this problem (0 for most shapes; the patch-staged convolution splits
 * its channel loop over workgroups for the 8x8 / 4x4 layers, partial slabs are reduced in fixed order). */
int64_t vd_gemm_ws_floats(const vd_gemm_desc* desc);
/* Kernel vd_gemm will use for this problem: 1: 128x128, 2: 64x128, 3: 64x64 gather tiles, 4 / 6: patch-staged 3x3
 * convolution kernel with 128x128 / 128x256 tiles, 5: plain GEMM kernel, 7: direct 3x3 convolution for <= 4 output
 * channels, 8 / 9 / 10: split-precision bf16 3x3 convolution / plain product (a_packed) / activation product (math = 1), 11: the persistent
 * variant of 9 (grids of >= 1024 tiles), 12 / 15 / 16: the 128 x 256 / 128 x 512 / split 128 x 256 tiles of 8, 13: the 128 x 256 tile of 9,
 * 17: the 16x16x32-MFMA 3x3 convolution (32-channel K-steps, 128 x 256 tile), 18: its persistent variant (LDS-DMA weight stages, 8 x 32 pixel
 * segments of images of any size), 19: the persistent 16x16x32 kernel for 9's problems, 20 (round 6): the whole-K 16x16x32 kernel of the 8x8 level
 * (and of 4x4 grids that fill the chip): 64 channels x 2 | 4 whole images per workgroup, no split-K workspace, f32 input, VD_B_CONV3 / VD_B_CONV3_T,
 * bias / rowadd / residual / accumulate; its sums run over all channels in one chain, so it agrees with the split kernels (8 / 16) to the path's
 * tolerance, not bit for bit, -1: a_packed given for an unsupported problem (profiling / tests). */
int vd_gemm_tile(const vd_gemm_desc* desc);

/* Weight gradient of a 3x3 / 1x1 convolution (K2/K5/K6/K7 backward, deterministic split-K):
 *   dW[m][c*T + t] (+)= sum_{b,p} dY[b][m][p] * gather(X)[b][c][p (+) t]
 * mode in {VD_B_PLAIN (1x1), VD_B_CONV3, VD_B_CONV3_S2, VD_B_CONV3_UP}.
 * ws must hold splits*M*C*T floats when splits > 1.  Replaces autograd's conv2d weight grad. */
typedef struct vd_wgrad_desc {
    const float* dY;
    const float* X;
    float* dW;
    float* ws;
    int32_t M, C, T;          /* T = 9 or 1                                                     */
    int32_t nb, NP;           /* batch, output pixels per item (OH*OW)                          */
    int32_t H, W, OH, OW;     /* X spatial dims, dY spatial dims                                */
    int32_t mode, splits, accumulate, tile;
    int64_t dy_bstride, x_bstride;
    int32_t pad;              /* VD_B_CONV3_S2 only, as in vd_gemm_desc                         */
    int32_t math;             /* 0: exact f32 MFMA.  1: split-precision bf16 MFMA (hi*hi + hi*lo + lo*hi, f32 accumulation,
                                 ~1e-5 relative): VD_B_CONV3 / VD_B_CONV3_UP with 8x8 / 16x16 / 32x32 outputs, either of them on images whose
                                 width is a multiple of 32 from 64 up, VD_B_CONV3 at 4x4, or VD_B_PLAIN (1x1) with
                                 NP % 8 == 0; M >= 64, C >= 64; otherwise VD_EINVAL */
    int32_t presplit;         /* ABI 11, math == 1 only.  Bit 0: X, bit 1: dY is a PRE-SPLIT image (vd_presplit_* below) instead of f32 NCHW.
                                 3 (both): the kernel fetches both operands by LDS-DMA and reads them through ds_read_b64_tr_b16 -- no
                                 conversion, no staging registers (VD_B_CONV3 / VD_B_CONV3_UP at 8x8 / 16x16 / 32x32 outputs, M % 8 == C % 8 == 0,
                                 grouped launches only).  Any other non-zero value, or a problem outside that set: VD_EINVAL.               */
    int32_t reserved_;
} vd_wgrad_desc;

int vd_conv_wgrad(const vd_wgrad_desc* desc, void* stream);
/* Floats of workspace vd_conv_wgrad will use for this problem (0 = no split-K chosen). */
int64_t vd_conv_wgrad_ws_floats(const vd_wgrad_desc* desc);
/* Tile and split count vd_conv_wgrad will use (profiling / tests). */
int vd_conv_wgrad_plan(const vd_wgrad_desc* desc, int* tile, int* splits);

/* ------------------------------------------------------------------------------------------
 * PRE-SPLIT activation images (round 5, ABI 11) -- csrc/vd_presplit.hip.
 * The split-precision kernels contract f32 operands as bf16 (hi, lo) pairs; a pre-split image holds those pairs as its PRODUCER
 * wrote them, so that no consumer converts: unit (o, p, part) = 16 bytes = channels 8o .. 8o+7 of pixel p as bf16 (part 0 = hi =
 * bf16(x), part 1 = lo = bf16(x - hi)) at byte ((o * HW + p) * 2 + part) * 16 of the image.  Same bytes, batch stride and
 * channel-octet offsets as the f32 [C][HW] image it replaces; C % 8 == 0.  Consumers: vd_gemm_desc.b_presplit (3x3 convolution
 * forward / input gradient), vd_wgrad_desc.presplit (grouped 3x3 weight gradient: LDS-DMA + transposed LDS reads).
 * Replaces the F.group_norm + F.silu -> F.conv2d hand-over inside diffusers ResnetBlock2D (reference loss.py:993).
 * ------------------------------------------------------------------------------------------ */
/* f32 NCHW -> pre-split image and back (x = hi + lo: 16 significant bits); strides in floats. */
int vd_presplit_pack(const float* x, void* y, int B, int C, int HW, int64_t x_bstride, int64_t y_bstride, void* stream);
int vd_presplit_unpack(const void* y, float* x, int B, int C, int HW, int64_t y_bstride, int64_t x_bstride, void* stream);
/* GroupNorm (+ SiLU) forward whose output IS the pre-split image (one read of x, one write of the pairs; mean / rstd as
 * vd_groupnorm_fwd).  _ok: 1 when a kernel exists for (C, HW, G) -- the 16x16 / 32x32 levels with 4 .. 16 channels per group. */
int vd_groupnorm_fwd_presplit_ok(int C, int HW, int G);
int vd_groupnorm_fwd_presplit(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int B, int C, int HW, int G,
                              float eps, int apply_silu, int64_t x_bstride, int64_t y_bstride, void* stream);
/* GroupNorm (+ SiLU) backward (same shapes) as vd_groupnorm_bwd_fused -- dgamma_ws / dbeta_ws rows per (image, channel), residual gradients extra /
 * extra2 added into dx, rowsum[b][c] = sum_p dx -- with dx written as f32 (dx), as the pre-split image (dx_ps: the operand of the producing
 * convolution's input AND weight gradient), or both; at least one of the two must be given. */
int vd_groupnorm_bwd_presplit(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                              const float* extra, const float* extra2, float* dx, void* dx_ps, float* dgamma_ws, float* dbeta_ws, float* rowsum,
                              int B, int C, int HW, int G, int apply_silu, int64_t dy_bstride, int64_t x_bstride, int64_t extra_bstride,
                              int64_t extra2_bstride, int64_t dx_bstride, int64_t ps_bstride, int64_t rowsum_ld, void* stream);

/* GROUPED weight gradients: several split-precision (math = 1) weight gradients of one kernel class in ONE launch pair (compute +
 * fixed-order slab reduction).  A weight gradient has a small output and a huge reduction length (K = batch * pixels), so a launch
 * that must fill 256 CUs alone splits K over ~32 workgroups per tile and moves 32 partial copies of dW through memory; sharing the
 * grid between the convolutions of a whole gradient bucket keeps the chip full with ~3 splits per tile.  (The reference's autograd
 * computes one weight gradient per convolution, VillanDiffusion.py:1161 accelerator.backward; the result is the same sum.)
 *   class  = vd_conv_wgrad_group_class(desc): 0 = not groupable (vd_conv_wgrad), equal values may share a launch;
 *   plan   : writes the host image of the device job table (n * vd_conv_wgrad_group_job_bytes() bytes; slab pointers held as OFFSETS),
 *            the workspace floats, the compute and reduce grid sizes; returns the class (> 0) or VD_EINVAL (< 0 never; 0 = error);
 *   rebase : once per uploaded table: turns the slab offsets into pointers into `ws`;
 *   launch : the two launches.  Deterministic (fixed reduction order); results agree with vd_conv_wgrad to summation order. */
int vd_conv_wgrad_group_class(const vd_wgrad_desc* desc);
int64_t vd_conv_wgrad_group_job_bytes(void);
/* Kernel family a class runs on: 9 = all nine taps per workgroup (plain 3x3 at 16x16 / 32x32), 32 = opt-in 16x16x32 one-tap-row kernel, 0 = default. */
int vd_conv_wgrad_group_variant(int cls);
int vd_conv_wgrad_group_plan(const vd_wgrad_desc* descs, int n, void* table_out, int64_t* ws_floats, int* blocks, int* rblocks);
int vd_conv_wgrad_group_rebase(void* dev_table, int n, float* ws, void* stream);
int vd_conv_wgrad_group_launch(const void* dev_table, int n, int cls, int blocks, int rblocks, void* stream);

/* Split-precision operand for vd_gemm_desc.a_packed.  taps = 9: element (m, c, t) of the logical [M][C][9] matrix of a 3x3
 * convolution is read from W[m*row_stride + c*chan_stride + t] (plain weights: row_stride = C*9, chan_stride = 9; the transposed
 * operand of the stride-1 dgrad, A'[c_in][m_out] = W[m_out][c_in]: row_stride = 9, chan_stride = C_in*9 with M = C_in, C = M_out).
 * taps = 1: element (m, c) of a plain [M][C] matrix (1x1 convolution, attention projection) from W[m*row_stride + c*chan_stride]
 * (row-major: (C, 1); its transpose: (1, M_out-row length)).  Stored as bf16 hi = bf16(w), lo = bf16(w - hi) in the kernel's
 * fragment order.  `packed` holds vd_conv3_packed_bytes(M, C, taps) bytes, 16-byte aligned; C % 16 == 0. */
int64_t vd_conv3_packed_bytes(int M, int C, int taps);
int vd_conv3_pack_weights(const float* W, void* packed, int M, int C, int taps, int64_t row_stride, int64_t chan_stride, void* stream);
/* The same for n_jobs operands in one launch (all convolutions of a network after an optimizer step).  table: DEVICE array of
 * int64 [n_jobs][8] = {W address, packed address, M, C, row_stride, chan_stride, first workgroup of the job, taps}; job j owns
 * ceil(Mpad_j * C_j / 8 / 256) workgroups of 256 threads, first-workgroup numbers ascending from 0; total_blocks = their sum. */
int vd_conv3_pack_weights_multi(const int64_t* table, int n_jobs, int64_t total_blocks, void* stream);
/* The same job table, f16 operands (round 4, opt-in mixed precision -- the reference's GPU arithmetic is fp16 autocast, VillanDiffusion.py:260-264):
 * unit ((cc * taps + t) * 2 + q) * Mpad + m = channels cc*16 + q*8 + j of tap t, row m, rounded to f16; a job's image holds
 * vd_conv3_packed_bytes(M, C, taps) / 2 bytes.  Consumed by vd_gemm with vd_gemm_desc.math = 2. */
int vd_conv3_pack_weights_f16_multi(const int64_t* table, int n_jobs, int64_t total_blocks, void* stream);

/* W[M][C][T] -> Wt[C][M][T]  (operand for the dgrad GEMMs). */
int vd_weight_transpose(const float* W, float* Wt, int M, int C, int T, void* stream);
/* dX[b][c][y][x] (+)= sum_{2x2} dU[b][c][2y+i][2x+j]   (Upsample2D backward). */
int vd_sumpool2x2(const float* dU, float* dX, int B, int C, int H, int W, int64_t du_bstride, int64_t dx_bstride,
                  int accumulate, void* stream);
/* Stride-2 conv dgrad, second half: dX[b][c][y][x] = sum_{r,s: y+pad-r, x+pad-s even}
 * G[b][(c*9+r*3+s)][(y+pad-r)/2][(x+pad-s)/2], where G[b] = W2d^T[C*9, M] . dY[b][M, OH*OW] comes from vd_gemm (VD_A_COL,
 * VD_B_PLAIN) -- the stride-2 Downsample2D conv of the reference UNet (pad as in vd_gemm_desc). */
int vd_col2im_s2(const float* G, float* dX, int B, int C, int H, int W, int OH, int OW, int pad, int64_t g_bstride,
                 int64_t dx_bstride, void* stream);
/* ws[b*ws_ld + m] = sum_p X[b][m][p]  (bias / temb-projection gradients), then vd_colsum over b. */
int vd_rowsum(const float* X, float* ws, int B, int M, int P, int64_t x_bstride, int64_t ws_ld, void* stream);
/* out[c] (+)= sum_b ws[b*ld + c]  -- fixed order, deterministic. */
int vd_colsum(const float* ws, float* out, int B, int C, int64_t ld, int accumulate, void* stream);
/* n_jobs column sums in one launch: job i = table[4i..4i+3] = {address of ws (+ first column), address of out (+ first
 * column), number of columns (<= 64), ld}; out[c] += sum_b ws[b*ld + c], same order as vd_colsum (bit-identical).
 * `table` is a DEVICE array (the job list of a backward pass is the same every step: build once, reuse). */
int vd_colsum_segmented(const int64_t* table, int n_jobs, int B, void* stream);

/* ------------------------------------------------------------------------------------------
 * K1 -- GroupNorm (+SiLU) forward/backward.  Replaces F.group_norm + F.silu of
 * ResnetBlock2D.norm{1,2}, AttentionBlock.group_norm, conv_norm_out.
 * ------------------------------------------------------------------------------------------ */
/* ws: vd_groupnorm_ws_floats() floats of scratch (16-byte aligned), or NULL.  Groups larger than 28 K elements (256x256 images) are cut
 * into chunks handled by separate workgroups when ws is given (fixed-order combination of the chunk statistics).  Outside a HIP-graph
 * capture the chunks of a group exchange their statistics inside ONE launch (bounded polling of (value, launch-epoch) words in ws; a
 * chunk stays in registers between the statistics and the apply step); inside a capture, or with VD_GN_CHUNK1_OFF=1, a statistics
 * launch and an apply launch. */
int64_t vd_groupnorm_ws_floats(int B, int C, int HW, int G);
int vd_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                     int B, int C, int HW, int G, float eps, int apply_silu, int64_t x_bstride, int64_t y_bstride,
                     float* ws, void* stream);
/* Statistics only: mean / rstd per (b, group) and ss[b][c] = {gamma_c*rstd, beta_c - mean*gamma_c*rstd} for vd_gemm_desc.gn_ss
 * (one read of x, no y).  Groups of at most 12 K elements. */
int vd_groupnorm_stats(const float* x, const float* gamma, const float* beta, float* ss, float* mean, float* rstd,
                       int B, int C, int HW, int G, float eps, int64_t x_bstride, void* stream);
/* vd_groupnorm_stats from the per-tile channel sums a convolution left in vd_gemm_desc.gn_part ([B][tiles][C][2], HW pixels per image in
 * `tiles` tiles): same outputs, no pass over x.  Sums are combined in double precision in a fixed order. */
int vd_groupnorm_stats_from_partials(const float* part, int tiles, const float* gamma, const float* beta, float* ss, float* mean,
                                     float* rstd, int B, int C, int HW, int G, float eps, void* stream);
/* dx = GN'(dy) (+ extra); dgamma_ws/dbeta_ws are [B][C] partials (reduce with vd_colsum). */
int vd_groupnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                     const float* beta, const float* extra, float* dx, float* dgamma_ws, float* dbeta_ws,
                     int B, int C, int HW, int G, int apply_silu, int64_t dy_bstride, int64_t x_bstride,
                     int64_t extra_bstride, int64_t dx_bstride, float* ws, void* stream);
/* The same with a second residual gradient (extra2: the gradient a skip connection carries into x -- replaces the
 * vd_add_strided pass after the block, VillanDiffusion's UNet concatenates skips: SURVEY 3.4) and, when rowsum != NULL,
 * rowsum[b*rowsum_ld + c] = sum_p dx[b][c][p]: the per-image bias-gradient rows of the layer that produced x (autograd's
 * conv2d bias gradient is the sum of its dY = this dx over batch and pixels; the per-image rows also are the
 * time_emb_proj output gradient of a ResnetBlock2D).  NULL extra / extra2 / rowsum are skipped. */
int vd_groupnorm_bwd_fused(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, const float* extra, const float* extra2, float* dx, float* dgamma_ws,
                           float* dbeta_ws, float* rowsum, int B, int C, int HW, int G, int apply_silu,
                           int64_t dy_bstride, int64_t x_bstride, int64_t extra_bstride, int64_t extra2_bstride,
                           int64_t dx_bstride, int64_t rowsum_ld, float* ws, void* stream);

/* ------------------------------------------------------------------------------------------
 * K4 -- attention softmax.  S is [nb][N][N] stored key-major: S[b][j][i] = k_j . q_i * scale;
 * softmax runs over j for every column i.  Replaces torch.softmax in AttentionBlock.
 * ------------------------------------------------------------------------------------------ */
int vd_softmax_col_fwd(float* S, int nb, int N, void* stream);                       /* in place */
int vd_softmax_col_bwd(const float* P, float* dP, int nb, int N, float scale, void* stream); /* dP -> dS in place */
/* Whole single-head attention for tiny token counts (N <= 64): qkv is [B][3C][N]. */
int vd_attn_small_fwd(const float* qkv, float* out, float* P, int B, int C, int N, float scale,
                      int64_t qkv_bstride, int64_t out_bstride, void* stream);
int vd_attn_small_bwd(const float* qkv, const float* P, const float* dout, float* dqkv, int B, int C, int N,
                      float scale, int64_t qkv_bstride, int64_t dout_bstride, int64_t dqkv_bstride, void* stream);

/* K4 fused (diffusers AttentionBlock reached from loss.py:993): the whole attention core of N == 256 tokens in one
 * launch, scores / probabilities in registers, split-precision (bf16 hi/lo, f32 accumulation) contractions on the bf16 MFMA.
 *   qkv [B][3C][N] (q | k | v, C = heads * head_dim, a head = a channel slice), out [B][C][N]:
 *   out[b][h d + c][i] = sum_j v[c][j] P[j][i],  P[.][i] = softmax_j(scale * sum_c k[c][j] q[c][i]).
 * P == NULL: nothing but `out` is written (no-grad path).  P != NULL: [B*heads][N][N] (P[j][i]) is written once for the backward pass.
 * head_dim in {32, 64, 128} or a multiple of 256; anything else / N != 256 -> VD_EINVAL (callers keep the 3-launch path).
 * vd_attn_core_bwd: dP = v^T dout, dS = scale P (dP - delta) with delta_i = sum_j P dP = sum_c dout[c][i] out[c][i] (the saved forward
 * output, so P is read once), dS written to [B*heads][N][N], and dq = k dS written into the q slice of dqkv [B][3C][N] -- one launch;
 * dk = q dS^T and dv = dout P^T are plain products of dS / P (vd_gemm). */
int vd_attn_core_fwd(const float* qkv, float* out, float* P, int B, int heads, int head_dim, int N, float scale, void* stream);
int vd_attn_core_bwd(const float* qkv, const float* P, const float* out, const float* dout, float* dS, float* dqkv, int B, int heads,
                     int head_dim, int N, float scale, void* stream);

/* K4 flash (round 4; same reference code as above, multi-head blocks with MORE than 256 tokens -- the 32x32-token level of the
 * `LDM-CELEBA-HQ-256` UNet, reference model.py:706-776): the N x N score matrix never reaches HBM.
 *   head_dim == 32, N a multiple of 256 (N >= 256); qkv [B][3C][N] with batch stride qkv_bstride, out [B][C][N] (out_bstride).
 * vd_attn_flash_fwd: online softmax over 256-key blocks; lse [B*heads][N] = log sum_j exp(scale s_ji) is written when non-NULL
 *   (the backward pass needs it; NULL in the no-grad path).
 * vd_attn_flash_bwd: two launches -- (1) dq and delta_i = sum_c dout[c][i] out[c][i] (delta: [B*heads][N] scratch, written), with
 *   P = exp(scale s - lse_i) recomputed per key block; (2) dk and dv, the transposed walk (a workgroup owns 128 keys).  All three
 *   slices of dqkv [B][3C][N] are written (no accumulation).  Other shapes -> VD_EINVAL. */
int vd_attn_flash_fwd(const float* qkv, float* out, float* lse, int B, int heads, int head_dim, int N, float scale, int64_t qkv_bstride,
                      int64_t out_bstride, void* stream);
int vd_attn_flash_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* delta, float* dqkv, int B, int heads,
                      int head_dim, int N, float scale, int64_t qkv_bstride, int64_t out_bstride, int64_t dout_bstride,
                      int64_t dqkv_bstride, void* stream);

/* ------------------------------------------------------------------------------------------
 * K3 -- timestep embedding + small elementwise helpers.
 * ------------------------------------------------------------------------------------------ */
/* emb[b][0:half]=sin(t_b*freqs_i), [half:2*half]=cos(t_b*freqs_i); flip_sin_to_cos swaps the halves.
 * freqs = exp(-ln(1e4)*i/(half-freq_shift)) is built by the host with the upstream fp32 op sequence. */
int vd_timestep_embedding(const float* t, const float* freqs, float* emb, int B, int half, int flip_sin_to_cos,
                          void* stream);
int vd_silu_fwd(const float* x, float* y, int64_t n, void* stream);
int vd_silu_bwd(const float* dy, const float* x, float* dx, int64_t n, int accumulate, void* stream);
/* dst[b][c][p] (+)= src[b][c][p] over [B][C*P] with batch strides. */
int vd_add_strided(float* dst, const float* src, int B, int64_t inner, int64_t dst_bstride, int64_t src_bstride,
                   int accumulate, void* stream);
/* x[i] *= alpha (alpha==0 stores exact zeros: gradient-buffer clear). */
int vd_scale(float* x, int64_t n, float alpha, void* stream);
/* out = sum_i coef[i] * src[i], n_src <= 6 (multistep sampler updates).  srcs / coefs are HOST arrays. */
int vd_lincomb(float* out, const float* const* srcs, const float* coefs, int n_src, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * K8 -- q-sample + backdoor shift + target + MSE (loss.py:909-939, 978-1006).
 *   x_t = a[t]*x0 + s[t]*eps + step[t]*R ;  y = coef[t]*R + eps     (VP/LDM: a=sqrt(abar), s=sqrt(1-abar);
 *                                                                    VE: a=1, s=sigma)
 * tables are device fp32 arrays indexed by the int64 timestep.
 * ------------------------------------------------------------------------------------------ */
int vd_qsample_backdoor(const float* x0, const float* R, const float* eps, const int64_t* t,
                        const float* tab_a, const float* tab_s, const float* tab_step, const float* tab_coef,
                        float* x_t, float* y, int B, int64_t chw, void* stream);
/* loss = mean((y - pred*pscale[b])^2); dpred = 2*(pred*pscale-y)*pscale/n * gscale.  partial: >= 1024 floats. */
int vd_mse_fwd_bwd(const float* pred, const float* y, const float* pscale, float* dpred, float* loss,
                   float* partial, int B, int64_t chw, float gscale, void* stream);
/* The same with the reference's selectable norm (loss.py:849-858, picked at loss.py:994,1003): kind VD_LOSS_L2 = F.mse_loss,
 * VD_LOSS_L1 = F.l1_loss (gradient sign(d), 0 at d == 0), VD_LOSS_HUBER = F.smooth_l1_loss (beta 1); all reduced by the mean. */
#define VD_LOSS_L2 0
#define VD_LOSS_L1 1
#define VD_LOSS_HUBER 2
int vd_loss_fwd_bwd(const float* pred, const float* y, const float* pscale, float* dpred, float* loss,
                    float* partial, int B, int64_t chw, float gscale, int kind, void* stream);

/* ------------------------------------------------------------------------------------------
 * K9 -- global grad-norm clip + Adam on flat buffers (VillanDiffusion.py:445,1165-1169).
 * ------------------------------------------------------------------------------------------ */
int vd_l2norm_sq(const float* g, int64_t n, float* partial, float* out_sq, void* stream);
/* clip = min(1, max_norm/(sqrt(*norm_sq)*inv_scale + 1e-6)); g' = g*inv_scale*clip; torch.optim.Adam update.  With norm_sq given and
 * *norm_sq not finite the kernel leaves p / m / v untouched (GradScaler.step's overflow skip) and adds 1 to *skipped (device word, may be
 * NULL): the host reads that counter lazily and refuses to go on (default arithmetic) or halves its loss scale (f16 mode). */
int vd_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* norm_sq, float max_norm,
                 float inv_scale, float lr, float beta1, float beta2, float eps, int step, unsigned* skipped, void* stream);

/* ------------------------------------------------------------------------------------------
 * K10 -- sampler steps (diffusers *Scheduler.step, via pipeline(...) VillanDiffusion.py:579).
 *   x0 = (x - c_eps*eps) / c_div ; clamp(+-clip) when clip > 0 ; out = c_x0*x0 + c_x*x + c_e*eps + c_z*z
 * every product/sum individually rounded (no FMA contraction) so results match the torch op sequence bit for bit.
 * z: noise pointer, or NULL with (seed, offset) for the on-device Philox4x32-10 + Box-Muller stream (c_z != 0).
 * ------------------------------------------------------------------------------------------ */
int vd_sched_step(const float* x, const float* eps, const float* z, float* out, float* x0_out, int64_t n,
                  float c_eps, float c_div, float clip, float c_x0, float c_x, float c_e, float c_z,
                  uint64_t seed, uint64_t offset, void* stream);
/* out[b] = sqrt(sum_i x[b][i]^2): per-sample L2 norms (ScoreSDE-VE corrector step size, diffusers step_correct). */
int vd_batch_l2norm(const float* x, float* out, int B, int64_t inner, void* stream);
/* out = clamp(x*mul + add, lo, hi), optionally NCHW -> NHWC (pipeline post-processing). */
int vd_postprocess(const float* x, float* out, int B, int C, int HW, float mul, float add, float lo, float hi,
                   int to_nhwc, void* stream);
/* out[n] = mean over the SSIM map of image pair n (a, b: [N, C, H, W]; win: the K x K gaussian window, K odd <= 32): torchmetrics'
 * StructuralSimilarityIndexMeasure(data_range) as measure() uses it (reference VillanDiffusion.py:1001-1007): reflect padding whose border is
 * cropped from the map; c1 = (0.01 data_range)^2, c2 = (0.03 data_range)^2. */
int vd_ssim(const float* a, const float* b, const float* win, float* out, int N, int C, int H, int W, int K, float c1, float c2,
            void* stream);
/* VQ-VAE quantiser (diffusers VectorQuantizer.forward, reached through VQModel.decode: reference loss.py:951-962,
 * model.py:713): zq[b][:, p] = codebook[argmin_e |z[b][:, p] - e|^2], idx[b*HW + p] = that e (may be NULL).
 * z / zq: [B, D, HW] with batch strides, codebook [n_e, D], D <= 16. */
int vd_vq_nearest(const float* z, const float* codebook, float* zq, int64_t* idx, int B, int D, int HW, int n_e,
                  int64_t z_bstride, int64_t q_bstride, void* stream);
/* NCSN++ (SDE-VE score network, reference model.py:839-894) helpers.
 * vd_fir_resample2: diffusers upsample_2d / downsample_2d with the (1,3,3,1) FIR kernel, factor 2, on `planes` images of
 *   H x W; out = scale * resample(x) (+ out).  The adjoints are the same kernels: d(up)^T g = 4 down(g), d(down)^T g = up(g)/4.
 * vd_fourier_embedding: GaussianFourierProjection(log=True): emb[b] = [sin(log t_b W 2pi), cos(log t_b W 2pi)].
 * vd_rowscale: out[b][:] = x[b][:] * s[b] (divide=0) or / s[b] (divide=1)  (UNet2DModel's final `sample / timesteps`). */
int vd_fir_resample2(const float* x, float* out, int64_t planes, int H, int W, int up, float scale, int accumulate, void* stream);
int vd_fourier_embedding(const float* t, const float* W, float* emb, int B, int half, void* stream);
int vd_rowscale(const float* x, const float* s, float* out, int B, int64_t inner, int divide, void* stream);
/* z ~ N(0,1) from Philox4x32-10 (throughput mode noise). */
int vd_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);

/* ------------------------------------------------------------------------------------------
 * K11 -- trigger stamping on the GPU (dataset.py:475-545 + util.py:119-147).
 * img: uint8 NHWC dataset resident in HBM; idx[b] (nullable = identity) selects the sample of batch slot b (fused
 * gather); flags: bit0 = poisoned, bit1 = horizontal flip.
 *   x = norm(img) ; clean: pixel_values=0, target=x ; poisoned: pixel_values=mask?x:trigger, target=target
 * ------------------------------------------------------------------------------------------ */
int vd_poison_batch(const uint8_t* img, const int64_t* idx, const uint8_t* flags, const float* trigger, const float* target,
                    float* pixel_values, float* tgt_out, float* image_out, int B, int C, int H, int W, float vmin,
                    float vmax, int R_trigger_only, void* stream);

/* ------------------------------------------------------------------------------------------
 * FID feature extractor (reference fid_score.py:91-148 -> pytorch-fid InceptionV3): the convolutions run
 * through vd_gemm (VD_B_CONVG / VD_B_PLAIN with act = 1 over BatchNorm-folded weights); these are the remaining ops.
 * ------------------------------------------------------------------------------------------ */
/* 3x3 pooling of NCHW planes: mode 0 = F.max_pool2d, 1 = F.avg_pool2d(count_include_pad=False); stride 1 | 2, pad 0 | 1.
 * y has (H + 2 pad - 3) / stride + 1 rows; bstrides in elements (channel slices of wider buffers are allowed). */
int vd_pool3(const float* x, float* y, int B, int C, int H, int W, int stride, int pad, int mode, int64_t x_bstride, int64_t y_bstride,
             void* stream);
/* y = mul * F.interpolate(x, (OH, OW), mode="bilinear", align_corners=False) + add over `planes` contiguous H x W planes. */
int vd_resize_bilinear(const float* x, float* y, int64_t planes, int H, int W, int OH, int OW, float mul, float add, void* stream);

/* ------------------------------------------------------------------------------------------
 * LPIPS (reference VillanDiffusion.py:892 -> lpips.LPIPS(net='alex')): the AlexNet convolutions run through vd_gemm
 * (VD_B_CONVG with act = 1) and vd_pool3; these are the remaining ops.
 * ------------------------------------------------------------------------------------------ */
/* y[b][c][:] = x[b][c][:] * mul[c] + add[c] over contiguous [B][C][HW] tensors (the input ScalingLayer). */
int vd_channel_affine(const float* x, const float* mul, const float* add, float* y, int B, int C, int HW, void* stream);
/* One tap of the metric on contiguous [N][C][HW] feature maps: out[n] (+)= mean_p sum_c w[c] * (unit(f0)[n][c][p] - unit(f1)[n][c][p])^2,
 * unit(f) = f / (sqrt(sum_c f^2) + 1e-10). */
int vd_lpips_layer(const float* f0, const float* f1, const float* w, float* out, int N, int C, int HW, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VILLAN_HIP_H */
