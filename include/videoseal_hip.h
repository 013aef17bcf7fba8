/*
 * videoseal_hip.h -- C-ABI of libvideoseal_hip.so: the MI355X (gfx950) kernels behind the
 * VideoSeal embed -> (augment) -> extract hot path.
 *
 * The reference (facebookresearch/videoseal) has no FFI/plugin seam: callers bind to a Python
 * nn.Module.  This header is therefore the boundary a maintainer would add:
 * plain pointers + sizes, no torch types, no allocation, no ownership transfer, no hidden sync.
 * Every entry point is stream-ordered (`stream` is a hipStream_t passed as void*), returns
 * VS_OK (0) or a negative VS_ERR_* code, and never touches host memory behind the pointers.
 * INTEGRATION.md shows the ctypes stub that binds it (videoseal_amd/native.py is that stub).
 *
 * Layout conventions
 *   full-resolution frames   : NCHW float32 in [0,1]  (the reference API contract, wam.py:134-204)
 *   network activations      : NHWC float32, channel stride `ld` (floats, multiple of 4); channels in
 *                              [C, ld) are zero and every producer keeps them zero
 *   conv / linear weights    : packed row-major [N][Ktot], Ktot = KH*KW*CinP, CinP = Cin rounded up to 16,
 *                              k = (ky*KW + kx)*CinP + c   (videoseal_amd/packing.py builds them, BN folded)
 */
#ifndef VIDEOSEAL_HIP_H
#define VIDEOSEAL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VS_OK 0
#define VS_ERR_BAD_ARG (-1)      /* null pointer, non-positive size, misaligned ld            */
#define VS_ERR_UNSUPPORTED (-2)  /* configuration outside what the kernels implement          */
#define VS_ERR_LAUNCH (-3)       /* hipGetLastError() != hipSuccess after the launch          */

#define VS_ACT_NONE 0
#define VS_ACT_RELU 1
#define VS_ACT_GELU 2            /* exact erf GELU (nn.GELU default)                          */
#define VS_ACT_TANH 3
#define VS_ACT_SILU 4            /* x * sigmoid(x) (legacy videoseal_0.0 U-Net, common.py:118-119)  */

#define VS_PAD_ZERO 0
#define VS_PAD_REFLECT 1

#define VS_VIDEO_REPEAT 0        /* videoseal.py:92-94  */
#define VS_VIDEO_ALTERNATE 1     /* videoseal.py:95-100 */
#define VS_VIDEO_INTERPOLATE 2   /* videoseal.py:101-117 */

/* library / device introspection */
int vs_version(void);                       /* ABI version, currently 3 (round 2: vs_conv_desc_t / vs_model_cfg_t grew the arithmetic and planes fields; round 6: grn_part / grn_gamma / grn_nchunk) */
const char* vs_arch(void);                  /* "gfx950" */
/* Development switch for tests / tools.  PROCESS-GLOBAL (one atomic per key: setting it is thread-safe, but a value set by one thread is seen
 * by the resize_pre / embed_tail launches of EVERY thread and stream -- e.g. the streaming overlap stream -- until it is reset; not a per-call
 * option, not for production hosts).  Launch paths never call getenv.  key 0 = resize_pre form (1 = the 32 x 8 tile kernel), 1 = resize_pre
 * strip height in output rows, 2 = embed_tail strip height in rows, 3 = JPEG round-trip form (1 = the one-pixel-per-lane kernels), 4 = Crop -> Resize -> colour form (1 = the 32 x 8 tile kernel), 5 = vs_to_planes_affine form (1 = one row per wave), 6 = output rows per strip of the one-row depthwise kernel (1 / 2), 7 = vs_layernorm_act form (1 = one wave per row), 8 = K-slice epilogue form (1 = the first kernel, one load in flight per thread); value 0 = the
 * library's choice.  The environment default
 * VIDEOSEAL_RESIZE=tile is latched ONCE per process at the first launch: toggling the variable afterwards has no effect, use this call. */
int vs_debug_set(int key, int value);
const char* vs_error_string(int code);
int vs_sizeof_conv_desc(void);              /* sizeof(vs_conv_desc_t): lets a binding verify its struct mirror */
int vs_sizeof_tail_desc(void);              /* sizeof(vs_tail_desc_t) */

/*
 * Implicit-GEMM convolution on the matrix cores, fp32 in / fp32 out.  Three arithmetic back-ends with
 * fp32-rounding-level accuracy: v_mfma_f32_32x32x2_f32 (exact fmaf chain, 157 TF), the "3 x bf16" split
 * (every operand = sum of three bf16 terms, six v_mfma_f32_32x32x16_bf16 per K chunk, 2.67x faster) and the "2 x f16" split
 * (arith = 2: two round-to-nearest f16 terms with power-of-two range scaling, three v_mfma_f32_32x32x16_f16 per K chunk):
 *   out[m, n] = epilogue( sum_k A[m,k] * wt[n,k] )    m = (b, oy, ox), n = output channel
 * Replaces every dense conv / nn.Linear of the path:
 *   unet.py:24-39 (ResnetBlock 3x3 + folded BN + ReLU, fused 1x1 res_conv via phase 2),
 *   unet.py:74-76 (3x3 stride-2 down), common.py:45-52 (reflect-pad 3x3 of Upsample),
 *   convnext.py:108-119 (4x4 s4 stem, 2x2 s2 downsample), convnext.py:32-35,47-51 (pwconv1/2).
 * Epilogue order: v = acc + bias; v = act(v); [phase 2: v += bias2 + in2 (1x1) wt2]; v += res; store.
 */
typedef struct vs_conv_desc {
  const float* in;          /* input activations                                                   */
  int64_t in_sb, in_sy, in_sx; /* strides in floats: frame, row, pixel                             */
  int32_t B, H, W, Cin;     /* Cin = floats read per tap (multiple of 4)                           */
  int32_t KH, KW, SH, SW, PH, PW, pad_mode;
  int32_t Ho, Wo;
  const float* wt;          /* [N][KH*KW*CinP]                                                     */
  int32_t CinP, N;
  const float* a_scale;     /* optional A transform a' = a*a_scale[b*a_scale_ld + c] + a_shift[c]; both  */
                            /*   arrays must be readable up to CinP entries per row (16-float chunks)     */
  int64_t a_scale_ld;       /*   (GRN apply, common.py:166-169); KH=KW=1 only                      */
  const float* a_shift;
  const float* bias;        /* [N] or NULL                                                         */
  int32_t act;
  int32_t n_store;          /* columns written (>= N, <= out_ld - out_coff); extra columns get 0   */
  const float* res;         /* optional residual [M][res_ld], added after act                      */
  int64_t res_ld;
  const float* in2;         /* optional phase 2: pointwise conv of a second tensor [M][in2_ld]     */
  int64_t in2_ld;
  int32_t Cin2, Cin2P;
  const float* wt2;         /* [N][Cin2P]                                                          */
  const float* bias2;
  float* out;               /* [M][out_ld], written at column offset out_coff                      */
  int64_t out_ld;
  int32_t out_coff;
  int32_t tile_hint;        /* low nibble: 0 = auto, 1 = 128x128, 2 = 128x64, 3 = 256x32, 4 = 128x192, 5 = 128x96,   */
                            /* (6..9: retired);                                                                      */
                            /* 10 = 128x32, 11 = 128x64, 12 = 128x128: 3x3 stride-1 'patch' kernel (8x16-pixel tile);  */
                            /* 13 = 64x64, 14 = 64x128 (generic kernel, small-M layers);                              */
                            /* 15 = 128x128, 16 = 128x192, 19 = 128x64, 21 = 256x64: wave-specialised patch kernel (needs wt_blk); */
                            /* 20: persistent 3x3 kernel for 16-input-channel layers with <= 32 outputs (weights in registers);          */
                            /* 17 = 128x128, 18 = 128x192: wave-specialised 1x1 GEMM (dense rows, Cin % 32 == 0, wt_blk);  */
                            /* 22 = 256x192, 23 = 256x128: all-DMA 3x3 kernel on pre-split planes (arith 2, in_pl, H % 16 == W % 16 == 0);  */
                            /* 24 = 256x192, 25 = 256x128: all-DMA 1x1 GEMM on pre-split planes (arith 2, in_pl; split_k, sumsq_part, res);  */
                            /* 26 = 128x96: the wave-specialised 1x1 GEMM with four consumer waves stacked over the rows (arith 2; round 5); */
                            /* 27 = 256x256: the all-DMA planes GEMM with ONE wave per SIMD (128 x 128 per wave, four 32 KiB stages; round 6:  */
                            /*      15 % fewer operand bytes per FLOP than tile 24 for the GEMMs that are LDS-DMA-bound -- ChunkySeal);          */
                            /* round 6: the 96-accumulator register tiles -- 18 and 24 -- are built without the tanh epilogue: an explicit request */
                            /*      for VS_ACT_TANH there with split_k <= 1 answers VS_ERR_UNSUPPORTED (tile 0 = auto never picks them for tanh;   */
                            /*      17 / 25 / 26 / 27 and every K-sliced launch carry it);                                                          */
                            /* | VS_CONV_TILE_HI: tile code + 16;                                                     */
                            /* | VS_CONV_FORCE_F32: v_mfma_f32_32x32x2_f32 path; | VS_CONV_FORCE_SPLIT */
  const void* wt_split;     /* optional [P][N][Ktot] 16-bit planes (P = 3 bf16 / 2 f16, see arith): wt split into P terms; when set */
  const void* wt2_split;    /*   (and wt2_split for phase 2) the split MFMA path is used                  */
  const void* wt_blk;       /* optional: the same split weights in LDS-image order [ceil(N/32)][Ktot/16][P][1 KiB]  */
  const void* wt2_blk;      /*   (slot of (row r, k-half h) inside a block = 2r + (h ^ ((r>>3)&1)), rows >= N zero);  */
                            /*   enables the producer/consumer kernels, tile codes 6..9, 15..18                        */
  float* splitk_ws;         /* split_k > 1 (tile codes 17, 18 only): workspace [split_k][M][splitk_ld] floats for the */
  int64_t splitk_ld;        /*   partial sums of the K slices; splitk_ld >= N.  Summed in slice order (deterministic)  */
  int32_t split_k;          /*   0 / 1 = no K split                                                                    */
  int32_t arith;            /* arithmetic of the split path: 0 / 3 = "3 x bf16" (wt_split / wt_blk hold 3 bf16 planes, 6 products),   */
                            /*   2 = "2 x f16" (2 f16 planes of wt * w_mul, 3 products; a_mul / acc_mul / acc_mul2 below)          */
  float* sumsq_part;        /* optional (1x1, no phase 2 / residual / K split): [ceil(M/32)][N] sums of squares of the stored  */
                            /*   values per 32-row group and column = GRN's ||x||^2 partials (common.py:166), see              */
                            /*   vs_grn_scale_from_partials                                                                    */
  const void* in_pl;        /* tile codes 22 / 23 (arith 2): the operands as PRE-SPLIT planes, f16 hi / lo of value * a_mul, layout        */
  const void* in2_pl;       /*   [2 planes][C/16][B*H*W][16] (vs_to_planes, or a previous launch's out_pl); `in` / `in2` may then be NULL  */
  void* out_pl;             /*   optional: the result written as planes for the next launch (N % 16 == 0); `out` may then be NULL          */
  float a_mul;              /* arith = 2: power of two the activations (in and in2) are multiplied with before the f16 split       */
  float acc_mul;            /*   = 1 / (a_mul * w_mul): the accumulator of phase 1 is multiplied with it before bias / activation  */
  float acc_mul2;           /*   = 1 / (a_mul * w2_mul): the same for the products of the second phase (in2 x wt2)                 */
  int32_t grn_nchunk;       /* ABI v3 (round 6), tile codes 17 / 18 / 26 with a_scale only: the GRN finish folded into the GEMM.  grn_part =   */
  const float* grn_part;    /*   the ||h||^2 partials [B][grn_nchunk][CinP] a previous launch wrote through sumsq_part (grn_nchunk = H*W/32   */
  const float* grn_gamma;   /*   <= 16), grn_gamma = GRN's gamma [CinP]: the kernel derives scale = 1 + gamma * Gx / (mean_c Gx + 1e-6) itself */
                            /*   (bit-identical to vs_grn_scale_from_partials) and a_scale is NOT read -- it must still be a valid pointer;   */
                            /*   every other tile code answers VS_ERR_UNSUPPORTED when grn_part is set                                        */
  int32_t sumsq_hw;         /* ABI v3, tile codes 24 / 25 with sumsq_part only: rows per frame when that is NOT a multiple of 32 (>= 32): sumsq_part */
  int32_t reserved_;        /*   is then [ceil(M/32)][2][N] -- per 32-row group the sums of its rows in the frame of its first row | in the next      */
                            /*   frame (vs_grn_scale_from_straddle_partials); 0 = the [M/32][N] form                                               */
} vs_conv_desc_t;
#define VS_CONV_FORCE_F32 0x10
#define VS_CONV_FORCE_SPLIT 0x20
#define VS_CONV_TILE_HI 0x40
#define VS_CONV_PRE 0x80          /* wave-specialised 3x3 kernel (tile codes 15 / 16) only: a_scale is a border-class table [frames][9][N]     */
                                  /*   (frame stride a_scale_ld, 0 = shared): v = act(acc + table[frame][class(y,x)][n] + bias), see vs_msg_pre */
int vs_conv_gemm(const vs_conv_desc_t* d, void* stream);
/* fp32 NHWC rows [rows][ld] -> the operand planes of tile codes 22 / 23: [2][C/16][rows][16] f16, hi = f16(x * a_mul), lo = f16(x * a_mul - hi).
 * C % 16 == 0; `planes` holds 2 * rows * C f16 values. */
int vs_to_planes(const float* x, int64_t rows, int C, int64_t ld, float a_mul, void* planes, void* stream);
/* The same with the GRN apply of common.py:166-169 in front of the split: x[r][c] * scale[r / rows_per_frame][c] + shift[c] (scale == NULL:
 * plain conversion).  scale / shift must be readable up to C entries per row. */
int vs_to_planes_affine(const float* x, int64_t rows, int C, int64_t ld, float a_mul, const float* scale, int64_t scale_ld,
                        const float* shift, int rows_per_frame, void* planes, void* stream);

/* LayerNorm over the channel dim of [rows][ld] (+ optional activation).  common.py:131-155 (both data formats). */
int vs_layernorm_act(const float* x, int64_t rows, int C, int64_t ld, const float* w, const float* b, float eps,
                     int act, float* out, int64_t out_ld, void* stream);
/* (round 6) The LayerNorm in front of a 2 x 2 / stride-2 down-sampling conv (convnext.py:109-117) written as that conv's patch matrix:
 * out [B][H/2][W/2][4 C], tap-major (ky, kx) then channel -- the conv is then a 1x1 vs_conv_gemm over rows of K = 4 C with the same packed weights.
 * C % 4 == 0 and 16-byte aligned operands, else VS_ERR_UNSUPPORTED (keep vs_layernorm_act + the strided conv). */
int vs_layernorm_patch2x2(const float* x, int B, int H, int W, int C, int64_t ld, const float* w, const float* b, float eps, float* out, void* stream);
/* (round 6) The ConvNeXt stem in one kernel: 4 x 4 patchify conv (3 -> CO channels, stride 4 or 2; convnext.py:100-104) + its LayerNorm, exact fp32
 * multiply-adds on the vector ALUs.  x: NHWC frames with 4 floats per pixel (rgb + zero lane), wt: the conv's packed rows [CO][4 ky][16 = 4 kx x 4],
 * out [B][Ho][Wo][out_ld].  CO in {64, 96, 128}, out_ld == CO, 16-byte aligned operands; else VS_ERR_UNSUPPORTED (keep vs_conv_gemm + vs_layernorm_act). */
int vs_stem_conv_ln(const float* x, int B, int H, int W, int stride, const float* wt, const float* bias, const float* lnw, const float* lnb, float eps,
                    int CO, float* out, int64_t out_ld, void* stream);

/* ChanRMSNorm over the channel dim of [rows][ld] (common.py:172-179: F.normalize(x, dim=1) * sqrt(C) * gamma, i.e.
 * x / max(||x||_2, 1e-12) * sqrt(C) * gamma[c]) + activation (+ add[row][c]: the ResnetBlock's res_conv branch, unet.py:38-39).
 * The U-Net norm of the legacy videoseal_0.0 card.  C % 4 == 0. */
int vs_rmsnorm_act(const float* x, int64_t rows, int C, int64_t ld, const float* gamma, int act, const float* add, int64_t add_ld,
                   float* out, int64_t out_ld, void* stream);

/* Multi-head self-attention of the SAM-style ViT extractor (vit.py:302-360 + 436-470), fp32:
 *   qkv  [frames*H*W][3*heads*hd]   rows = tokens (frame, y, x); columns [q | k | v], each [head][hd]   (vit.py:345-347)
 *   out  [frames*H*W][heads*hd]     column = head*hd + c                                               (vit.py:357)
 *   S[i][j] = (q_i * hd^-1/2) . k_j + q_i . rel_h[y_i - y_j + T_h - 1] + q_i . rel_w[x_i - x_j + T_w - 1];  out_i = softmax_j(S) v
 * window = 0: global attention over the H x W grid of a frame (T_h = H, T_w = W); window > 0: non-overlapping window x window
 * token windows (vit.py:363-402; H and W must be multiples of window), T_h = T_w = window.  rel_h / rel_w: [2*T-1][hd] or NULL.
 * hd in {16, 32, 64}; tokens per attention group <= 256. */
int vs_vit_attention(const float* qkv, int frames, int H, int W, int heads, int hd, int window, const float* rel_h, const float* rel_w,
                     float* out, void* stream);

/* ConvNeXt-V2 block front: depthwise 7x7 (pad 3, bias) fused with LayerNorm(C).  convnext.py:43-46.
 * wdw is packed [49][C]. */
int vs_dwconv7_ln(const float* x, int B, int H, int W, int C, int64_t ld, const float* wdw, const float* bdw,
                  const float* lnw, const float* lnb, float eps, float* out, int64_t out_ld, void* stream);
/* The same with the LayerNorm output written as the operand planes of vs_conv_gemm tile codes 24 / 25 (pwconv1 then reads them by
 * LDS-DMA): [2][planes_C/16][B*H*W][16] f16, hi / lo of value * a_mul; planes_C >= C is the consumer's padded K (CinP), channels >= C zero. */
int vs_dwconv7_ln_planes(const float* x, int B, int H, int W, int C, int64_t ld, const float* wdw, const float* bdw,
                         const float* lnw, const float* lnb, float eps, float a_mul, int planes_C, void* planes, void* stream);

/* GRN statistics: scale[b][c] = 1 + gamma[c] * Gx[b][c] / (mean_c Gx[b][.] + 1e-6), Gx = ||h[b,:,c]||_2.
 * common.py:166-168.  `partial` is workspace of nchunk*B*C floats, nchunk = ceil(HW/64). */
/* The same from the partial sums a vs_conv_gemm launch left in sumsq_part ([B][HW/32][C], HW % 32 == 0): no second pass
 * over the activations. */
int vs_grn_scale_from_partials(const float* partial, int B, int HW, int C, const float* gamma, float* scale, int64_t scale_ld,
                               void* stream);
/* The same for frames whose HW (>= 32) is not a multiple of 32 (ChunkySeal's 31 x 31): `partial` is the [ceil(B*HW/32)][2][C] form a tile-24 / 25
 * launch writes with vs_conv_desc_t::sumsq_hw = HW (per 32-row group: sums of its rows in the frame of its first row | in the next frame). */
int vs_grn_scale_from_straddle_partials(const float* partial, int B, int HW, int C, const float* gamma, float* scale, int64_t scale_ld,
                                        void* stream);
/* GRN apply as a separate in-place pass: h = h * scale[b] + beta (common.py:168 minus the residual `+ x`, which the caller's
 * pwconv2 input convention already folds into scale = 1 + gamma*Nx).  beta must be readable up to the next multiple of 4. */
int vs_grn_apply(float* h, int B, int HW, int C, int64_t ld, const float* scale, int64_t scale_ld, const float* beta, void* stream);
int vs_grn_scale(const float* h, int B, int HW, int C, int64_t ld, const float* gamma, float* partial,
                 float* scale, void* stream);

/* BatchNorm2d with BATCH statistics over the rows of an NHWC tensor -- the U-Net's ResnetBlocks under model.train()
 * (unet.py:26,30; nn.BatchNorm2d training branch: biased variance for the normalisation, running statistics updated in place
 * with `momentum` and the UNBIASED variance; either running pointer may be NULL).  Outputs the folded per-channel
 * scale = gamma / sqrt(var + eps) and shift = beta - mean * scale, to be applied with vs_scale_shift_act.  Deterministic
 * (fp64 partial sums in a fixed order).  `partial` is workspace of vs_bn_partial_doubles(rows, ld) doubles. */
int64_t vs_bn_partial_doubles(int64_t rows, int64_t ld);
int vs_bn_batch_stats(const float* x, int64_t rows, int C, int64_t ld, const float* gamma, const float* beta, float eps,
                      float momentum, float* running_mean, float* running_var, double* partial, float* scale, float* shift,
                      void* stream);
/* The same in two halves for nn.SyncBatchNorm (train.py:438-440 converts the model when training is distributed; torch's
 * SyncBatchNorm normalises with the statistics of the GLOBAL batch and updates the running statistics with the global unbiased variance):
 * vs_bn_partial_sums writes sums = [sum x (ld) | sum x^2 (ld) | rows] = 2*ld + 1 doubles for the local rows; the host all-reduces that
 * vector over the ranks (RCCL, fp64 sum) and vs_bn_finish_sums turns it into scale / shift and the running-statistics update.  Without
 * the all-reduce the pair is bit-identical to vs_bn_batch_stats (same summation order). */
int vs_bn_partial_sums(const float* x, int64_t rows, int C, int64_t ld, double* partial, double* sums, void* stream);
int vs_bn_finish_sums(const double* sums, int C, int64_t ld, const float* gamma, const float* beta, float eps, float momentum,
                      float* running_mean, float* running_var, float* scale, float* shift, void* stream);
/* out = act(x * scale[c] + shift[c]) (+ add): the normalisation + ReLU (+ res_conv branch, unet.py:38-39) of a train-mode ResnetBlock.
 * C % 4 == 0. */
int vs_scale_shift_act(const float* x, int64_t rows, int C, int64_t ld, const float* scale, const float* shift, int act,
                       const float* add, int64_t add_ld, float* out, int64_t out_ld, void* stream);

/* ---- backward building blocks of the ConvNeXt-V2 extractor (csrc/bwd_ops.hip; SURVEY.md 8(f)1: the detector fine-tuning step of
 * train.py:517-523, 626-643).  Backward-DATA products are vs_conv_gemm launches on transposed weights; these are the rest.  All NHWC
 * fp32, leading dimensions multiples of 4, reductions deterministic (fixed order, fp64 across chunks).
 * vs_gemm_wgrad: dw[n][k] = sum_rows dy[row][n] * x[row][k] (dense [N][K]); partial = vs_gemm_wgrad_partial_floats floats.
 * vs_dwconv7: depthwise 7x7 pad 3 (+ bias) (+ add), w = [49][ld]; flip = 1 uses tap 48 - t: the backward-data pass (convnext.py:43).
 * vs_dwconv7_wgrad: dw[49][ld] of the same; partial = vs_dwconv7_wgrad_partial_floats.
 * vs_layernorm_bwd: LayerNorm over C (biased variance): dx, d weight, d bias; stats = 2 * rows floats (mean, rstd out); partial =
 *   vs_colreduce_partial_floats(1, rows, ld).
 * vs_gelu_grn_bwd: h3 = gamma * (h2 * Nx) + beta + h2, h2 = gelu(h1), Nx = ||h2||_2(H,W) / (mean_c + 1e-6) (common.py:158-169):
 *   dh1, d gamma, d beta from d3 = dL/dh3; coef = 6 * B * ld floats, partial = vs_colreduce_partial_floats(B, HW, ld).
 * vs_patchify / vs_unpatch: patch matrix of a P x P stride-P conv in the k order of the packed weights, and its adjoint.
 * vs_col2im3x3_reflect: adjoint of vs_im2col3x3 with reflection padding.  vs_colmean / vs_pool_gelu_bwd / vs_matmul_small: the head
 *   (pixel_decoder.py:61-83).  vs_bce_logits: decoding loss (videosealloss.py:150-156) and d preds (column 0 gets 0). */
int64_t vs_gemm_wgrad_partial_floats(int64_t rows, int N, int K);
int vs_gemm_wgrad(const float* dy, int64_t dy_ld, int N, const float* x, int64_t x_ld, int K, int64_t rows, float* partial, float* dw,
                  void* stream);
/* vs_conv3x3_wgrad: weight gradient of a 3x3 conv (padding 1: VS_PAD_ZERO with stride 1 | 2, unet.py:21-27, 52; VS_PAD_REFLECT with stride 1,
 * the Upsample conv of unet.py:170-197) straight from the NHWC image: dw[n][tap * ld + c] -- the result of vs_im2col3x3(_strided) +
 * vs_gemm_wgrad without the patch matrix.  Supported shapes (vs_conv3x3_wgrad_supported): N >= 64 and ld >= 64 (matrix cores), or
 * N <= 32, ld <= 96 with N % 4 == 0 (the thin outer levels).  partial: vs_conv3x3_wgrad_partial_floats(N, ld, B, H, W, stride) floats.
 * vs_pad_embed1 / vs_reflect_fold1: the backward-DATA pass of the reflection-padded conv = a zero-padded conv of the gradient placed in
 * a zero canvas [B, H + 2, W + 2] (vs_pad_embed1), folded back onto [B, H, W] (vs_reflect_fold1: adjoint of reflection padding 1). */
int vs_conv3x3_wgrad_supported(int N, int64_t ld, int stride);
int64_t vs_conv3x3_wgrad_partial_floats(int N, int64_t ld, int B, int H, int W, int stride);
int vs_conv3x3_wgrad(const float* dy, int64_t dy_ld, int N, const float* x, int64_t ld, int B, int H, int W, int stride, int pad_mode,
                     float* partial, float* dw, void* stream);
int vs_pad_embed1(const float* dy, int B, int H, int W, int64_t ld, float* out, void* stream);
/* ---- backward of the SAM-style ViT extractor of the legacy card (csrc/bwd_vit.hip; vit.py:146-193, 302-360, 436-470).
 * vs_gelu_bwd: dz = dy * gelu'(z) (rows of C values; pad columns of dz are zeroed).
 * vs_vit_attention_bwd: adjoint of vs_vit_attention.  qkv / out: the forward's operand and result, dout: gradient of out; dqkv receives
 * (dq | dk | dv) in qkv's layout; drel_h / drel_w [2 * T - 1][hd] the gradients of the two relative-position tables (NULL with rel_h ==
 * NULL).  scratch: vs_vit_attention_bwd_scratch_floats(...) floats.  Deterministic (no atomics). */
int vs_gelu_bwd(const float* z, int64_t ld, const float* dy, int64_t dy_ld, int64_t rows, int C, float* dz, int64_t dz_ld, void* stream);
/* vs_act_bwd: dz = dy * act'(z) for VS_ACT_RELU | GELU | TANH | SILU (z = the pre-activation value).
 * vs_rmsnorm_act_bwd: adjoint of vs_rmsnorm_act (ChanRMSNorm + activation of the legacy U-Net, common.py:172-179): dx, and term[r][c] whose
 * column sums over the rows are d gamma. */
int vs_act_bwd(const float* z, int64_t ld, const float* dy, int64_t dy_ld, int64_t rows, int C, int act, float* dz, int64_t dz_ld, void* stream);
int vs_rmsnorm_act_bwd(const float* x, int64_t rows, int C, int64_t ld, const float* gamma, int act, const float* dy, int64_t dy_ld, float* dx,
                       int64_t dx_ld, float* term, int64_t term_ld, void* stream);
int64_t vs_vit_attention_bwd_scratch_floats(int frames, int H, int W, int heads, int window);
int vs_vit_attention_bwd(const float* qkv, const float* out, const float* dout, int frames, int H, int W, int heads, int hd, int window,
                         const float* rel_h, const float* rel_w, float* dqkv, float* scratch, float* drel_h, float* drel_w, void* stream);
int vs_reflect_fold1(const float* dxp, int B, int H, int W, int64_t ld, float* out, void* stream);
int vs_dwconv7(const float* x, int B, int H, int W, int C, int64_t ld, const float* w, const float* bias, int flip, const float* add,
               int64_t add_ld, float* out, int64_t out_ld, void* stream);
int64_t vs_dwconv7_wgrad_partial_floats(int B, int H, int64_t ld);
int vs_dwconv7_wgrad(const float* x, int64_t ld, const float* dy, int64_t dy_ld, int B, int H, int W, int C, float* partial, float* dw,
                     void* stream);
int64_t vs_colreduce_partial_floats(int B, int64_t HW, int64_t ld);
int vs_layernorm_bwd(const float* x, int64_t ld, const float* dy, int64_t dy_ld, const float* w, int64_t rows, int C, float eps, float* dx,
                     int64_t dx_ld, float* stats, float* partial, float* dw, float* db, void* stream);
int vs_gelu_grn_bwd(const float* h1, int64_t ld, const float* d3, int64_t d3_ld, const float* gamma, int B, int HW, int C, float* partial,
                    float* coef, float* dh1, int64_t dh1_ld, float* dgamma, float* dbeta, void* stream);
int vs_patchify(const float* x, int B, int H, int W, int64_t pld, int P, float* cols, void* stream);
int vs_unpatch(const float* dcols, int B, int H, int W, int64_t pld, int P, float* dx, void* stream);
/* the same for a P x P conv with stride S <= P (overlapping patches: ChunkySeal's 4 x 4 stride-2 stem, convnext.py:109) */
int vs_patchify_s(const float* x, int B, int H, int W, int64_t pld, int P, int S, float* cols, void* stream);
int vs_unpatch_s(const float* dcols, int B, int H, int W, int64_t pld, int P, int S, float* dx, void* stream);
int vs_col2im3x3_reflect(const float* dcols, int B, int H, int W, int64_t ld, float* dx, void* stream);
int vs_colmean(const float* x, int B, int HW, int64_t ld, float* out, void* stream);
int vs_pool_gelu_bwd(const float* z, int64_t ld, const float* dpooled, int64_t dp_ld, int B, int HW, int C, float* dz, int64_t dz_ld,
                     void* stream);
int vs_matmul_small(const float* A, int64_t lda, const float* Bm, int64_t ldb, int M, int N, int K, float* C, int64_t ldc, void* stream);
int vs_bce_logits(const float* preds, const int32_t* msgs, int msg_rows, int B, int k, float temperature, float gscale, float* dpreds,
                  float* loss, void* stream);

/* ---- backward building blocks of the U-Net embedder under model.train() (csrc/bwd_unet.hip; the generator side of train.py:626-643).
 * vs_bn_mean_rstd: batch statistics from the (all-reduced) moment vector of vs_bn_partial_sums.  vs_bn_relu_bwd_sums / _apply: BatchNorm
 *   on batch statistics (+ the ReLU that follows it) backward in two halves -- sums = [sum g xhat | sum g | rows] (2 * 4 ceil(C/4) + 1
 *   doubles) is what SyncBatchNorm all-reduces between them; dgamma / dbeta are the local sums.  vs_dilate2 + the forward conv on flipped
 *   weights = backward-data of the stride-2 down convs; vs_im2col3x3_strided: their patch matrix (zero padding 1) for vs_gemm_wgrad.
 * vs_upcat2x_bwd: adjoint of vs_upcat2x.  vs_msg_table_grad: the message embedding table.  vs_outc_tanh_bwd: adjoint of vs_outc_tanh
 *   (dv = [rows][4] dense, for the weight / bias gradients). */
int vs_bn_mean_rstd(const double* sums, int C, int64_t ld, float eps, float* mean, float* rstd, void* stream);
int64_t vs_bn_bwd_partial_floats(int64_t rows, int64_t ld);
int vs_bn_relu_bwd_sums(const float* raw, int64_t ld, const float* dy, int64_t dy_ld, const float* mean, const float* rstd, const float* scale,
                        const float* shift, int relu, int64_t rows, int C, float* partial, double* sums, float* dgamma, float* dbeta,
                        void* stream);
int vs_bn_relu_bwd_apply(const float* raw, int64_t ld, const float* dy, int64_t dy_ld, const float* mean, const float* rstd, const float* scale,
                         const float* shift, int relu, const double* sums, int64_t rows, int C, float* dx, int64_t dx_ld, void* stream);
int vs_dilate2(const float* dy, int B, int Ho, int Wo, int64_t ld, int H, int W, float* out, void* stream);
int vs_im2col3x3_strided(const float* x, int B, int H, int W, int64_t ld, int stride, float* cols, void* stream);
int vs_upcat2x_bwd(const float* dhi, int64_t hi_ld, int B, int H, int W, int C1, int C2, float skip_scale, float* dx, int64_t ld1,
                   float* dskip, int64_t ld2, void* stream);
int vs_msg_table_grad(const float* dlat, const int32_t* msgs, int Bm, int nbits, int hidden, float* dtable, void* stream);
int vs_relu_bwd(const float* z, int64_t ld, const float* dy, int64_t dy_ld, int64_t rows, int C, float* dz, int64_t dz_ld, void* stream);
int vs_outc_tanh_bwd(const float* delta, const float* ddelta, int64_t rows_per_frame, int B, int C, const float* w, int Cout, int use_tanh,
                     float* dx, int64_t dx_ld, float* dv, void* stream);

/* ---- ResnetBlock of the thin full-resolution U-Net levels in one launch (csrc/resblock_thin.hip; unet.py:24-39 with eval BatchNorm folded
 * into the convolutions, ReLU):   out = relu(conv3x3_1(t) + b1) + (conv1x1_res(x) + br),   t = relu(conv3x3_0(x) + b0)
 * x: NHWC fp32 [B][H][W][x_ld], Cin channels read per pixel (multiple of 4); Cout = 16 mid and output channels with Cin <= 16 (VideoSeal 1.0: `inc`,
 * last `ups` block) or 32 with Cin <= 32 (the 128^2 level; PixelSeal's `inc`; 2 x f16 arithmetic only: the weights live in LDS); t stays on chip.
 * Weights as the operand planes of vs_split_block for K = 9 * Cout resp. Cout ([P][Cout][K] 16-bit patterns, P = 2 f16 terms
 * of w * w_mul / 3 bf16 terms; channels >= Cin zero); acc_mul* = 1 / (a_mul * w_mul) of each convolution (arith 2).  Replaces two vs_conv_gemm
 * launches (and the HBM round trip of t) where vs_resblock_thin_supported(x_ld, mid, out) says so. */
typedef struct vs_resblock_thin_desc {
  const float* x; int64_t x_ld;
  int32_t B, H, W, Cin;
  const void *w0_split, *w1_split, *wr_split;
  const float *b0, *b1, *br;                /* [16] each or NULL                                                  */
  int32_t arith;                            /* 2 = 2 x f16, 3 = 3 x bf16 (16 channels only)                       */
  int32_t Cout;                             /* mid = output channels: 16 (or 0) / 32; weights then [P][Cout][9 * K], K = 16 / 32  */
  float a_mul, acc_mul0, acc_mul1, acc_mulr;
  float* out; int64_t out_ld;
} vs_resblock_thin_desc_t;
int vs_resblock_thin_supported(int cin_ld, int cmid, int cout);
int vs_resblock_thin(const vs_resblock_thin_desc_t* d, void* stream);

/* *flag |= 1 when x[0 .. n) holds inf / NaN: the always-on guard of the 2 x f16 arithmetic at the network boundary (engine.py reads the flag
 * synchronously on the first call with new weights and asynchronously afterwards). */
int vs_check_finite(const float* x, int64_t n, int* flag, void* stream);

/* *bits = max(*bits, bit pattern of max |x[0 .. n)|) (non-negative floats order like their bit patterns; inf / NaN sort above every finite
 * value).  The calibration pass of the per-layer arithmetic choice (engine.py `_calibrate_extractor`): after the range guard has tripped, the
 * extractor runs once on the exact 3 x bf16 split recording the largest operand of every dense layer; only the layers whose operand leaves the
 * f16 range of the 2 x f16 split stay on the exact split (no reference counterpart: the reference computes in fp32 on ATen). */
int vs_absmax(const float* x, int64_t n, unsigned* bits, void* stream);

/* ---- ConvNeXt-V2 block body pwconv1 -> GELU -> GRN -> pwconv2 + residual with the 4C-wide tensor kept on chip (csrc/convnext_fused.hip;
 * convnext.py:47-56, common.py:158-169), C = 96 / 192 (stages 0 / 1 of the VideoSeal extractor), 2 x f16 arithmetic.  Two launches per block:
 * stats = 1 writes the GRN partial sums [rows / 32][4C] (finished by vs_grn_scale_from_partials), stats = 0 recomputes pwconv1 + GELU, applies
 * scale [B][scale_ld] / beta and runs pwconv2 (+ bias2 + res -> out; out may alias res).  tn_planes: the f16 operand planes of the LayerNorm output
 * (vs_dwconv7_ln_planes); wimg: vs_cnx_block_image_bytes(C) bytes packed by the host (engine.pack_cnx_block: per 32 h-channels the W1 rows, the
 * W2 columns in the k order the accumulator layout dictates, bias1 / beta).  stats bit 1 (value 2) selects the serial kernel instead of the
 * software-pipelined one (identical results; for tests and tools). */
int vs_cnx_block_supported(int C, int64_t rows, int HW);
int64_t vs_cnx_block_image_bytes(int C);
int vs_cnx_block(const void* tn_planes, const void* wimg, int C, int64_t rows, int HW, int stats, float acc_mul1, float acc_mul2, const float* scale,
                 int64_t scale_ld, const float* bias2, const float* res, int64_t res_ld, float* out, int64_t out_ld, float* part32, void* stream);

/* ---- weight operand packing on the device (csrc/pack.hip): packed fp32 weights [N][K] (k = (tap, channel) with K / ntaps a multiple of 16) ->
 * the 16-bit planes [P][N][K] and their LDS-image order [ceil(N/32)][K/16][P][64][8] the split kernels read (P = 3 bf16 truncation terms for
 * arith 3; P = 2 f16 terms of w * w_mul for arith 2).  One launch per layer instead of a dozen ATen ops. */
int vs_split_block(const float* wt, int N, int64_t K, int ntaps, int arith, float w_mul, void* split, void* blk, void* stream);
/* vs_pack_conv: fp32 conv weights [Co][Ci][KH][KW] -> the packed matrix [rows][KH*KW*cinp] (k = (ky * KW + kx) * cinp + c, pad channels zero) that
 * vs_conv_gemm / vs_split_block read.  transpose = 0: rows = Co, channels = Ci (the forward weights); transpose = 1: rows = Ci, channels = Co,
 * taps flipped -- the backward-data weights of the same layer (for 1x1: the transposed GEMM matrix). */
int vs_pack_conv(const float* w, int Co, int Ci, int KH, int KW, int cinp, int transpose, float* out, void* stream);

/* ---- adjoints of the full-resolution shell and of the augmentations between embed and detect (csrc/bwd_shell.hip): d(loss)/d(imgs_w) and
 * d(loss)/d(imgs_aug) -> d(delta), the gradient the U-Net backward starts from (train.py:626-643).  Gather form, deterministic.
 * vs_resize_nchw_bwd: transpose of vs_resize_nchw (dy [planes][oh][ow] -> dx [planes][H][W]; tmp = planes * oh * W floats); also the transpose of
 *   the watermark's up-resize inside vs_embed_tail (wam.py:99-101) when called with (H, W) = (S_h, S_w).
 * vs_embed_tail_bwd: blend / attenuation(imgs, imgs_w) / clamp of wam.py:103-113 (hmap_full = the heat-map of vs_jnd_heatmap or NULL;
 *   preds = the forward's preds_w): g_full [F][Cd][H][W] from d_imgs_w [F][3][H][W] (and/or d_preds_w).
 * vs_tail_key_reduce: key-frame expansion adjoint (videoseal.py:80-118) times the low-resolution heat-map (or NULL): d_delta [total_key][Cd][S][S].
 * vs_aug_crop_flip_bwd / vs_mask_mul / vs_aug_color_bwd / vs_clamp01_bwd: Crop, HorizontalFlip, the mask blend of augmenter.py:175, Brightness /
 *   Contrast / Saturation / Grayscale / Hue (op codes of vs_aug_color), the clamp in front of JPEG's straight-through estimator.
 * vs_nhwc_to_nchw_scaled: the extractor's input gradient (NHWC) back to frame planes, times d(x * 2 - 1) / dx.
 * vs_percep_mse / vs_percep_mse_grad: the "mse" / "yuv" perceptual term (perceptual.py:20-28, yuvloss.py:11-27) and upstream * d loss / d imgs_w;
 *   partial = vs_percep_partial_doubles(F, H, W) doubles. */
int vs_resize_nchw_bwd(const float* dy, float* dx, int planes, int H, int W, int oh, int ow, int antialias, float* tmp, void* stream);
int vs_embed_tail_bwd(const float* imgs, const float* preds, const float* hmap_full, const float* d_imgs_w, const float* d_preds_w, int F, int H, int W,
                      int Cd, int clamp, float scaling_i, float scaling_w, float* g_full, void* stream);
int vs_tail_key_reduce(const float* g_low, const float* hmap_low, int F, int Cd, int S_h, int S_w, int step, int video_mode, int total_key,
                       float* d_delta, void* stream);
int vs_aug_crop_flip_bwd(const float* dy, float* dx, int planes, int H, int W, int i0, int j0, int h, int w, int flip, void* stream);
int vs_mask_mul(const float* dy, const float* mask, float* dx, int F, int C, int H, int W, int complement, void* stream);
int64_t vs_aug_color_bwd_scratch_floats(int F, int H, int W);
int vs_aug_color_bwd(const float* x, const float* dy, float* dx, int F, int H, int W, int op, float factor, const float* means, float* scratch,
                     void* stream);
int vs_clamp01_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);
/* vs_gaussian_blur_bwd: adjoint of vs_gaussian_blur (reflection padding folded back); tmp: planes * H * W floats.
 * vs_aug_warp_bwd: adjoint of vs_aug_warp (Rotate: nearest, Perspective: bilinear; zero padding), gather form, deterministic.  inv[9]: the
 * row-major 3 x 3 map from input pixel-centre coordinates (x + 0.5, y + 0.5, 1) to homogeneous output pixel-centre coordinates -- the inverse
 * of the sampling map, used only to bound the search (every candidate is re-sampled with the forward arithmetic).
 * vs_aug_color_bwd now covers the hue op as well (chain rule through torchvision's RGB -> HSV -> RGB with autograd's conventions). */
/* vs_aug_gather_frames_bwd: adjoint of vs_aug_gather_frames; start [n_src + 1] / outs: per source frame the outputs that copied it (CSR, ascending).
 * vs_aug_window_average_bwd: adjoint of vs_aug_window_average. */
int vs_aug_gather_frames_bwd(const float* dy, const int32_t* start, const int32_t* outs, float* dx, int n_src, int64_t frame_floats, void* stream);
int vs_aug_window_average_bwd(const float* dy, float* dx, int F, int64_t frame_floats, int half_window, float alpha, void* stream);
int vs_gaussian_blur_bwd(const float* dy, float* tmp, float* dx, int planes, int H, int W, int k, float sigma, void* stream);
int vs_aug_warp_bwd(const float* dy, float* dx, int planes, int H, int W, int oh, int ow, int kind, const float* coeffs, int bilinear,
                    const float* inv, void* stream);
int vs_nhwc_to_nchw_scaled(const float* src, int F, int H, int W, int C, int64_t ld, float mul, float* dst, void* stream);
int64_t vs_percep_partial_doubles(int F, int H, int W);
int vs_percep_mse(const float* imgs, const float* imgs_w, int F, int H, int W, int yuv, double* partial, float* loss, void* stream);
int vs_percep_mse_grad(const float* imgs, const float* imgs_w, int F, int H, int W, int yuv, float upstream, float* d_imgs_w, void* stream);

/* Bilinear x2 (align_corners=False) of cat(x, skip*skip_scale) along channels.  unet.py:186-191 + common.py:46. */
int vs_upcat2x(const float* x, int C1, int64_t ld1, const float* skip, int C2, int64_t ld2, float skip_scale,
               int B, int H, int W, float* out, int64_t out_ld, void* stream);

/* The Upsample group of the U-Net (common.py:45-52: bilinear x2 -> ReflectionPad2d(1) -> Conv3x3 (no bias) -> LayerNorm(C) -> act) WITHOUT
 * the up-sampled tensor.  Interpolation, padding and the tap shift are linear over space and commute with the conv's channel mixing:
 *   conv3x3(pad(up(v)))[Y,X,c] = sum_t up(z_t)[refl(Y+ky-1), refl(X+kx-1), c],   z[b][y][x][t*Co + c] = sum_ci W[c][ci][t] v[b][y][x][ci]
 * z is one vs_conv_gemm launch on the LOW-resolution map (1x1, N = 9*Co rows ordered (tap, channel): a quarter of the conv's MACs);
 * vs_upconv_gather_ln does the 9-tap x 4-neighbour gather, the LayerNorm over Co (biased variance, eps) and the activation, and writes
 * [B][2H][2W][out_ld].  vs_upconv_supported(Co): Co % 16 == 0 and Co / min(16, Co/4) a power of two (every released card).
 * vs_cat2_scale: out[r] = [x[r][0:C1] | skip[r][0:C2] * skip_scale] -- unet.py:186-187 at the low resolution; x == NULL when the
 * producer of x already wrote columns [0, C1) of `out`. */
int vs_upconv_supported(int Co);
int vs_upconv_gather_ln(const float* z, int64_t z_ld, int B, int H, int W, int Co, const float* lnw, const float* lnb, float eps,
                        int act, float* out, int64_t out_ld, void* stream);
int vs_cat2_scale(const float* x, int C1, int64_t ld1, const float* skip, int C2, int64_t ld2, float skip_scale, int64_t rows,
                  float* out, int64_t out_ld, void* stream);

/* The same Upsample group in ONE kernel for the thin levels (Co = 16 or 32, C1 % 16 == 0, C2 % 16 == 0, C1 + C2 <= 256): the nine-tap
 * GEMM of an 8 x 8-cell tile (+ halo) runs on the matrix cores inside the workgroup and z stays in LDS; inputs are x and skip
 * themselves (no concat buffer).  wt_split = the [P][9*Co][C1+C2] 16-bit planes of the (tap, channel)-ordered weight; arith / a_mul / acc_mul
 * as in vs_conv_desc_t -- arith 0 or 3: three bf16 planes, exact; arith 2: two f16 planes of wt * w_mul, acc_mul = 1 / (a_mul * w_mul). */
int vs_upconv_fused_supported(int C1, int C2, int Co);
int vs_upconv_fused_preferred(int C1, int C2, int Co);   /* supported AND measured faster than GEMM + gather (Co = 16 levels) */
int vs_upconv_fused(const float* x, int C1, int64_t ld1, const float* skip, int C2, int64_t ld2, float skip_scale, const void* wt_split,
                    int B, int H, int W, int Co, const float* lnw, const float* lnb, float eps, int act, float* out, int64_t out_ld,
                    int arith, float a_mul, float acc_mul, void* stream);

/* Patch matrix of a 3x3 / stride 1 / pad 1 conv (zero or reflect padding) on a small NHWC map: out[m][t*ld + c], m = (b, y, x),
 * t = ky*3 + kx -- the K order of the packed conv weights, so the conv becomes vs_conv_gemm with KH = KW = 1, Cin = 9*ld on `out`
 * (used for the pixel decoder's conv on the 8x8 map, pixel_decoder.py:44-48, whose 2048 rows cannot fill the chip as an implicit GEMM). */
int vs_im2col3x3(const float* x, int B, int H, int W, int64_t ld, int pad_mode, float* out, void* stream);

/* Message latent: lat[b][c] = sum_k table[2k + msg[b][k]][c].  msg_processor.py:88-98.  msgs are int32 0/1. */
int vs_msg_latent(const float* table, const int32_t* msgs, int Bm, int nbits, int hidden, float* lat, void* stream);
/* Broadcast lat[b or 0][0:hidden] over H*W pixels into channels [coff, coff+hidden) of dst.  msg_processor.py:96-115. */
int vs_broadcast_channels(const float* lat, int Bm, int hidden, float* dst, int B, int HW, int64_t ld, int coff,
                          void* stream);

/* First bottleneck block of the U-Net (unet.py:183-185): its input is [latent | message], and the message channels are CONSTANT over
 * space (msg_processor.py:96-115), so conv3x3 over them is a per-frame table of NINE border classes (zero padding: first / last rows
 * and columns miss taps).  P[bm][tap*N + n] = sum_c W[n][tap][msg c] * latent[bm][c] comes from one small vs_conv_gemm; vs_msg_pre
 * reduces it to table[bm][cy*3 + cx][n] = sum_{valid ky, kx} P[bm][(ky*3+kx)*N + n]  (cy / cx: 0 first, 1 interior, 2 last row / column).
 * The 3x3 conv over the latent channels alone then adds table[frame][class(y, x)][n] before bias + activation (VS_CONV_PRE with
 * a_scale = table, a_scale_ld = 9*N, or 0 when one message serves every frame): K shrinks from 9*(lat+msg) to 9*lat. */
int vs_msg_pre(const float* P, int Bm, int N, float* table, void* stream);

/* 1x1 conv C -> Cout (Cout <= 4) + bias + optional tanh, NHWC in, planar [B][Cout][HW] out.  unet.py:166,194-196. */
int vs_outc_tanh(const float* x, int64_t rows_per_frame, int B, int C, int64_t ld, const float* w, const float* bias,
                 int Cout, int use_tanh, float* out, void* stream);

/* Mean over HW then Linear.  pixel_decoder.py:76-79.  w is [N][C] row-major (native nn.Linear layout). */
int vs_pool_linear(const float* x, int B, int HW, int C, int64_t ld, const float* w, const float* bias, int N,
                   float* out, void* stream);

/*
 * Resize (bilinear, align_corners=False, antialias on/off -- ATen's separable triangle filter,
 * wam.py:161-164, 222-225; videoseal.py:303-306) of NCHW frames fused with the layout change and
 * the pre-processing of the consumers:
 *   dst_rgb [B][oh][ow][4]  = (r,g,b,0) * mul + add             (detector: x*2-1, extractor.py:164;
 *                                                                 low-res JND: raw, mul=1 add=0)
 *   dst_y   [Bk][oh][ow][4] = ((yr*r + yg*g + yb*b) * 2 - 1, 0,0,0) for frames f % y_step == 0
 *                             (data/transforms.py:23-27 row 0 + embedder.py:163)
 * Either destination may be NULL.  ymat3 is a HOST array of 3 floats (row 0 of rgb2yuv.M) or NULL, in
 * which case dst_y receives (r,g,b,0)*2-1 of the key frames (RGB embedders such as ChunkySeal's).
 */
int vs_resize_pre(const float* src, int B, int C, int H, int W, int oh, int ow, int antialias, float* dst_rgb, float mul,
                  float add, float* dst_y, int y_step, const float* ymat3, void* stream);
/* The same for uint8 RGB24 frames [B][H][W][3] as they come off the ffmpeg pipe of inference_streaming.py:57-66:
 * every sample is converted as float(u) / 255.0f, i.e. `torch.tensor(clip, dtype=float32).permute(0,3,1,2) / 255.0`
 * (inference_streaming.py:26, 120) fused into the resize -- no fp32 copy of the clip is ever materialised. */
int vs_resize_pre_u8(const unsigned char* src, int B, int H, int W, int oh, int ow, int antialias, float* dst_rgb, float mul,
                     float add, float* dst_y, int y_step, const float* ymat3, void* stream);

/* JND heat-map (jnd.py:63-108, in_channels=1/out_channels=1) of an RGB image addressed by strides
 * (floats): frame, channel, row, pixel.  taps43 is a HOST array (copied into the kernel arguments):
 * [0..24] 5x5 lum, [25..33] sobel x, [34..42] sobel y -- the values of attenuation.conv_{lum,x,y}.weight. */
int vs_jnd_heatmap(const float* img, int B, int H, int W, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                   const float* taps43, float* hmap, void* stream);

/*
 * Embed tail (wam.py:182-197, videoseal.py:316-344): for every full-resolution pixel
 *   d      = resize(video_mode(delta))             delta: [Fk][Cd][S][S], Cd in {1,3}
 *   d     *= hmap_lowres (before the resize) or JND(imgs) (after it)       (hmap_lowres may be NULL,
 *                                                                           attenuate=0 disables JND)
 *   out    = clamp(scaling_i*imgs + scaling_w*d, 0, 1)
 * attenuate = 2 (full-resolution JND only) is the training forward's order of operations (wam.py:103-113, jnd.py:110-114):
 *   v = scaling_i*imgs + scaling_w*d;  out = clamp(imgs + JND(imgs) * (v - imgs), 0, 1);  preds_w = the UN-attenuated d.
 * imgs/out: NCHW [F][3][H][W]; preds_w (optional) [F][Cd][H][W].
 * io_u8 = 1: imgs and out are uint8 RGB24 [F][H][W][3]; read as float(u)/255.0f, written as (unsigned char)(v*255.0f)
 * = `(imgs_w * 255.0).byte().permute(0,2,3,1)` of inference_streaming.py:31 (needs clamp = 1).
 */
typedef struct vs_tail_desc {
  const void* imgs; void* out; float* preds_w;
  const float* delta; const float* hmap_lowres; const float* taps43;   /* taps43: HOST pointer */
  int32_t F, H, W, S_h, S_w, Cd;
  int32_t step, video_mode, total_key;      /* key-frame expansion                         */
  int32_t attenuate, clamp, antialias;
  float scaling_i, scaling_w;
  int32_t io_u8;
  int32_t variant;          /* 0 = default (row-streaming kernel where it applies, VIDEOSEAL_TAIL overrides); 1 = 43-tap JND on 16-row tiles,   */
                            /* 2 = separable stencils on 16-row tiles, 3 = 8-row tiles with the pixels parked in LDS, 4 = row-streaming:        */
                            /* same values from every form (tests/test_gpu_kernels.py), A/B handle for tools/bench_shell.py                      */
} vs_tail_desc_t;
int vs_embed_tail(const vs_tail_desc_t* d, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Model-level entry points (the granularity a non-Python host would bind): an opaque model built from the card numbers
 * (utils/cfg.py:88-154, embedder.py:243-262, extractor.py:189-208) and the reference state_dict (cfg.py:147-150:
 * checkpoint['model'], HOST fp32 tensors by their reference names, loaded like strict=False: unknown names are ignored,
 * required ones must be present).  vs_model_create folds eval-BatchNorm, packs / splits / blocks the weights and uploads
 * them once; afterwards the model is immutable (thread-safe across streams) and embed / detect allocate nothing: the
 * caller passes a 256-byte-aligned device workspace of vs_model_workspace_bytes(...) bytes.
 *   vs_model_embed  = Wam.embed / Videoseal.embed on ONE chunk of device-resident frames (wam.py:134-204,
 *                     videoseal.py:258-350): resize -> Y -> U-Net -> (low-res | full-res) JND -> blend -> clamp.
 *                     imgs / imgs_w: NCHW fp32 [frames][3][H][W]; msgs: int32 0/1 [n_msgs][nbits], n_msgs = 1 (video:
 *                     one message for the clip) or ceil(frames/step) (image mode: step = 1); preds_w optional.
 *   vs_model_detect = Wam.detect / Videoseal.detect (wam.py:206-234): resize -> ConvNeXt-V2 -> logits [frames][1+nbits].
 * io_u8 = 1: imgs / imgs_w are uint8 RGB24 [frames][H][W][3] (inference_streaming.py:26,31), conversions fused.
 * Chunking over long clips, frame aggregation (videoseal.py:390-428) and message generation stay with the caller. */
typedef struct vs_model vs_model_t;
typedef struct vs_model_cfg {
  int32_t nbits, hidden, img_size;          /* args.nbits, nbits * hidden_size_multiplier, args.img_size_proc             */
  int32_t in_ch, out_ch, yuv;               /* embedder I/O channels; yuv = 'yuv' in the embedder name (embedder.py:281) */
  int32_t nlev;                             /* U-Net down/up levels = len(z_channels_mults) - 1                           */
  int32_t zc[8];                            /* z_channels * mult, nlev + 1 entries                                        */
  int32_t num_blocks, last_tanh;            /* bottleneck ResnetBlocks; tanh on the output                                */
  int32_t depths[4], dims[4], stem_stride;  /* ConvNeXt-V2 (dims already scaled by sqrt(nbits/128) where the card asks)   */
  int32_t attenuate, clamp;                 /* JND attenuation present; clamp imgs_w to [0,1]                             */
  float scaling_w, scaling_i;               /* blender (mutable by evals: pass the current values at create time)         */
  int32_t arith, reserved_;                 /* arithmetic of the dense layers (vs_conv_desc_t::arith): 2 = 2 x f16, 3 = 3 x bf16,   */
                                            /*   0 = the library default (VS_DEFAULT_ARITH, overridable with VIDEOSEAL_CONV=f16x2|bf16x3) */
} vs_model_cfg_t;
typedef struct vs_tensor { const char* name; const float* data; int64_t numel; } vs_tensor_t;
int vs_model_create(const vs_model_cfg_t* cfg, const vs_tensor_t* tensors, int ntensors, vs_model_t** out);
void vs_model_destroy(vs_model_t* m);
int64_t vs_model_workspace_bytes(const vs_model_t* m, int frames, int H, int W, int step);
int vs_model_embed(vs_model_t* m, const void* imgs, const int32_t* msgs, int n_msgs, int frames, int H, int W, int step,
                   int video_mode, int lowres_attenuation, int antialias, int io_u8, void* imgs_w, float* preds_w, void* ws,
                   int64_t ws_bytes, void* stream);
int vs_model_detect(vs_model_t* m, const void* imgs, int frames, int H, int W, int antialias, int io_u8, float* logits, void* ws,
                    int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Augmentations (videoseal/augmentation/valuemetric.py, geometric.py, utils/image.py).  Frames are NCHW fp32 [F][3][H][W]
 * (masks: [F][1][H][W] through the `planes` entry points).  All are forward-only (the reference's JPEG / median use a
 * straight-through estimator whose forward value is exactly the codec / filter output).
 */
#define VS_COLOR_BRIGHTNESS 0   /* valuemetric.py:99-117  torchvision adjust_brightness: clamp(f*x)                      */
#define VS_COLOR_CONTRAST 1     /* valuemetric.py:120-137 adjust_contrast: blend with the per-frame mean of gray          */
#define VS_COLOR_SATURATION 2   /* valuemetric.py:139-155 adjust_saturation: blend with gray (0.2989, 0.587, 0.114)       */
#define VS_COLOR_HUE 3          /* valuemetric.py:157-174 adjust_hue: RGB -> HSV, h = (h + f) mod 1, HSV -> RGB            */
#define VS_COLOR_GRAYSCALE 4    /* valuemetric.py:196-208 0.299 R + 0.587 G + 0.114 B broadcast to 3 channels            */
int64_t vs_aug_color_scratch_floats(int F, int H, int W);
int vs_aug_color(const float* src, float* dst, int F, int H, int W, int op, float factor, float* scratch, void* stream);
/* A run of n (1..6) colour ops in ONE pass (round 5; the validation chains of augmentation/__init__.py:107-123 apply Brightness -> Contrast ->
 * Saturation -> Hue back to back): `ops` / `factors` are HOST arrays, every op is the expression of vs_aug_color applied in order on the
 * pixel's registers -> bit-identical to the separate calls.  VS_COLOR_CONTRAST needs the mean of ITS input: only as ops[0] (scratch as for
 * vs_aug_color); cut longer sequences there. */
int vs_aug_color_chain(const float* src, float* dst, int F, int H, int W, int n, const int* ops, const float* factors, float* scratch, void* stream);
/* Crop (window i0, j0, ch, cw inside the H x W frames) -> bilinear Resize to oh x ow -> n (0..6) colour ops without VS_COLOR_CONTRAST, one kernel,
 * 3-channel frames: the cropped clip is never written, the source window of a 32 x 8 output tile is staged in LDS once; same taps in the same
 * order as vs_resize_nchw on the cropped tensor -> bit-identical to vs_aug_crop_flip + vs_resize_nchw + vs_aug_color.  VS_ERR_UNSUPPORTED when
 * the window does not fit 60 KB of LDS (down-scaling beyond ~5 x): use the separate calls. */
int vs_aug_crop_resize_color(const float* src, float* dst, int F, int H, int W, int i0, int j0, int ch, int cw, int oh, int ow, int antialias,
                             int n, const int* ops, const float* factors, void* stream);
/* watermark masking of the training forward (augmenter.py:171-176): dst = imgs_w * m + imgs * (1 - m), m [F][1][H][W]          */
int vs_aug_mask_blend(const float* imgs_w, const float* imgs, const float* mask, float* dst, int F, int C, int H, int W, void* stream);
/* GaussianNoise (valuemetric.py:176-194): dst = x + noise * std over n floats; `noise` is the caller's torch.randn_like draw    */
int vs_aug_add_scaled(const float* x, const float* noise, float std, float* dst, int64_t n, void* stream);
/* DropFrame / SpeedChange (augmentation/video.py:491-526, 263-313): dst[f] = src[idx[f]] for n_out whole frames; idx int32 on device */
int vs_aug_gather_frames(const float* src, const int32_t* idx, float* dst, int n_out, int64_t frame_floats, void* stream);
/* video.py:411-486 WindowAveraging: (1 - alpha) * f[i] + alpha * mean of the frames within half_window of i */
int vs_aug_window_average(const float* src, float* dst, int F, int64_t frame_floats, int half_window, float alpha, void* stream);
/* crop (geometric.py:94-124, zero fill outside) and/or horizontal flip (geometric.py:186-196) of `planes` H x W planes   */
int vs_aug_crop_flip(const float* src, float* dst, int planes, int H, int W, int i0, int j0, int h, int w, int flip, void* stream);
/* bilinear resize NCHW -> NCHW, align_corners=False, antialias on/off (geometric.py:62-91; augmenter.py:147-150)        */
int vs_resize_nchw(const float* src, float* dst, int planes, int H, int W, int oh, int ow, int antialias, void* stream);
/* Rotate / Perspective (geometric.py:28-59, 127-183 -> torchvision F.rotate (nearest) / F.perspective (bilinear)): sampling grid +
 * grid_sample(padding zeros, align_corners=False) in one kernel, `planes` H x W planes -> oh x ow.
 *   kind 0 (affine):      coeffs[6] (HOST) = inverse affine matrix rows divided by (0.5*W) resp. (0.5*H)  (_gen_affine_grid)
 *   kind 1 (perspective): coeffs[8] (HOST) = the 8 coefficients of _get_perspective_coeffs(startpoints, endpoints)
 * torchvision is not vendored by the reference: parity of this entry point is pinned only against oracle/augment.py's restatement. */
int vs_aug_warp(const float* src, float* dst, int planes, int H, int W, int oh, int ow, int kind, const float* coeffs,
                int bilinear, void* stream);
/* torchvision gaussian_blur: k x k (odd), reflect padding, separable (valuemetric.py:53-71)                              */
int vs_gaussian_blur(const float* src, float* tmp, float* dst, int planes, int H, int W, int k, float sigma, void* stream);
/* median of the k row-medians of the zero-padded k x k window, k in {3,5,7} (utils/image.py:60-84)                        */
int vs_median_filter(const float* src, float* dst, int planes, int H, int W, int k, void* stream);
/* clamp -> uint8 (truncation) -> libjpeg baseline 4:2:0 encode+decode at `quality` (bit-exact with libjpeg-turbo's
 * islow path used by Pillow) -> /255   (valuemetric.py:21-50, utils/image.py:13-34)                                      */
int64_t vs_jpeg_workspace_bytes(int F, int H, int W);
int vs_jpeg_roundtrip(const float* src, float* dst, int F, int H, int W, int quality, void* workspace, void* stream);

/* H.264-style transform-coding PROXY for VideoCompression / H264 / H264rgb / H265 (augmentation/video.py:20-205).  The reference
 * round-trips the clip through libx264 / libx265 via PyAV on the CPU; that codec has no in-repo arithmetic and is not available
 * offline, so this entry point is a DEFINED stand-in, not a bit-stream model: clamp -> uint8 -> [integer BT.601
 * YCbCr, 4:2:0] -> per 4x4 block H.264 core transform + quantisation / de-quantisation at qp (flat 128 prediction, no deblocking) ->
 * back to RGB / 255.  qp in [0, 51] (the wrappers pass clamp(crf)); rgb_mode = 1 codes the R, G, B planes directly (libx264rgb).
 * Bit-exact with oracle/h264_proxy.py.  workspace: vs_h264_proxy_workspace_bytes(F, H, W) bytes. */
int64_t vs_h264_proxy_workspace_bytes(int F, int H, int W);
int vs_h264_proxy_roundtrip(const float* src, float* dst, int F, int H, int W, int qp, int rgb_mode, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIDEOSEAL_HIP_H */
