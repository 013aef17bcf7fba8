// Backward building blocks of the ConvNeXt-V2 extractor (the detector fine-tuning step of train.py:517-523, 626-643 --
// embedder frozen, decoding loss only).  gfx950 only.
//
// Division of labour: every backward-DATA product (dX = dY W) is a launch of the forward GEMM / conv kernels on transposed weights
// (conv_gemm.hip; the host packs W^T once per step), with the exact 3 x bf16 operand split -- gradients span far more binades than
// activations, and the bf16 split keeps the full fp32 exponent range.  This file holds what has no forward counterpart:
//   gemm_wgrad        dW[n][k] = sum_rows dY[row][n] X[row][k]   (the reduction runs over the pixels: both operands are read row-major,
//                     no transposes; 64 x 64 output tiles, rows split over blockIdx.z, partials summed in fp64 in a fixed order)
//   dwconv7 / dwconv7_wgrad          depthwise 7x7: backward-data = the forward filter with flipped taps; weight gradient per (tap, channel)
//   ln_bwd + colreduce<LN>           LayerNorm over C: dx per row (one wave per row), d weight / d bias as column reductions
//   colreduce<GRN> + grn_*           GELU + GRN (common.py:158-169): the three per-(frame, channel) sums, the coupling through the
//                                    channel mean of the norms, and the element-wise pass
//   patchify / unpatch, col2im3x3_reflect, pool_gelu_bwd, colmean, matmul_small, bce_logits: the data-movement adjoints and the head
// All reductions are deterministic (fixed chunking, fixed summation order, fp64 across chunks).  First version: correctness and
// determinism first; the element-wise kernels are HBM-bound as they stand, gemm_wgrad is an fp32-FMA kernel (MFMA version: DESIGN 7).
#include "vs_common.h"
#include "conv_common.h"

namespace {

constexpr int CR_ROWS = 64;       // rows per chunk of the column reductions

// ---------------------------------------------------------------------------------------------------
// sum over chunks: out[i] = sum_k partial[k][i]  (fp64, fixed order);  n elements per chunk.  Workgroup = 64 elements x 4 chunk lanes: lane g
// adds the chunks k = g, g + 4, ... (eight loads in flight), the four lane sums are combined in the order g = 0..3 -- the order depends on
// nchunk only, never on the launch.
__global__ __launch_bounds__(256) void reduce_chunks_kernel(const float* __restrict__ partial, int nchunk, int64_t n, float* __restrict__ out) {
  __shared__ double sh[4][64];
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + e;
  double s = 0;
  if (i < n) {
    const float* p = partial + i;
    int k = g;
    for (; k + 28 < nchunk; k += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(k + 4 * u) * n];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; k < nchunk; k += 4) s += (double)p[(int64_t)k * n];
  }
  sh[g][e] = s;
  __syncthreads();
  if (g == 0 && i < n) out[i] = (float)(((sh[0][e] + sh[1][e]) + sh[2][e]) + sh[3][e]);
}

// ---------------------------------------------------------------------------------------------------
// dW[n][k] = sum_r dY[r][n] * X[r][k].  Workgroup = 64 (n) x 64 (k) outputs, thread = 4 x 4, 16 rows per LDS step.
__global__ __launch_bounds__(256) void gemm_wgrad_kernel(const float* __restrict__ dy, int64_t dy_ld, int N, const float* __restrict__ x,
                                                         int64_t x_ld, int K, int64_t rows, int64_t rows_per_split,
                                                         float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float sa[16][64], sb[16][64];
  const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
  const int64_t r_begin = (int64_t)blockIdx.z * rows_per_split;
  const int64_t r_end = r_begin + rows_per_split < rows ? r_begin + rows_per_split : rows;
  const int tn = threadIdx.x >> 4, tk = threadIdx.x & 15;
  const int lr = threadIdx.x >> 4, lc = (threadIdx.x & 15) * 4;        // this thread's piece of the 16 x 64 staging tiles
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int64_t r0 = r_begin; r0 < r_end; r0 += 16) {
    const int64_t r = r0 + lr;
    f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
    if (r < r_end) {
      if (n0 + lc + 4 <= dy_ld) va = *reinterpret_cast<const f32x4*>(dy + r * dy_ld + n0 + lc);
      if (k0 + lc + 4 <= x_ld) vb = *reinterpret_cast<const f32x4*>(x + r * x_ld + k0 + lc);
    }
    __syncthreads();                       // the previous step's reads are done
    *reinterpret_cast<f32x4*>(&sa[lr][lc]) = va;
    *reinterpret_cast<f32x4*>(&sb[lr][lc]) = vb;
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&sa[rr][4 * tn]);
      const f32x4 b = *reinterpret_cast<const f32x4*>(&sb[rr][4 * tk]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += a[i] * b;
    }
  }
  float* p = partial + (int64_t)blockIdx.z * N * K;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + 4 * tn + i;
    if (n >= N) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + 4 * tk + j;
      if (k < K) p[(int64_t)n * K + k] = acc[i][j];
    }
  }
}

// The same product on the matrix cores, fp32 in and out (v_mfma_f32_32x32x2_f32: A[m][k] / B[k][n] with lane = (m or n) + 32 k).  With the
// reduction index k = row, both fragments are plain reads of 32 consecutive channels of two rows: no transposes and no operand split
// (fp32 products).  Workgroup = 128 (n) x 128 (k) outputs, 4 waves of 64 x 64 (2 x 2 MFMA tiles), 16 rows per LDS step, the next step's
// rows are fetched into registers while the matrix cores work.  Selected by VS_WGRAD=mfma (see vs_gemm_wgrad).
__global__ __launch_bounds__(256) void gemm_wgrad_mfma_kernel(const float* __restrict__ dy, int64_t dy_ld, int N, const float* __restrict__ x,
                                                              int64_t x_ld, int K, int64_t rows, int64_t rows_per_split,
                                                              float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float sa[16][128], sb[16][128];
  const int n0 = blockIdx.y * 128, k0 = blockIdx.x * 128;
  const int64_t r_begin = (int64_t)blockIdx.z * rows_per_split;
  const int64_t r_end = r_begin + rows_per_split < rows ? r_begin + rows_per_split : rows;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wy = wave >> 1, wx = wave & 1;
  const int lr = threadIdx.x >> 5, lc = (threadIdx.x & 31) * 4;       // staging: rows lr and lr + 8, 4 floats at column lc
  const int kq = lane >> 5, c = lane & 31;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  f32x4 va[2], vb[2];
  auto fetch = [&](const int64_t r0) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t r = r0 + lr + 8 * h;
      va[h] = f32x4{0.f, 0.f, 0.f, 0.f};
      vb[h] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (r < r_end) {
        if (n0 + lc + 4 <= dy_ld) va[h] = *reinterpret_cast<const f32x4*>(dy + r * dy_ld + n0 + lc);
        if (k0 + lc + 4 <= x_ld) vb[h] = *reinterpret_cast<const f32x4*>(x + r * x_ld + k0 + lc);
      }
    }
  };
  fetch(r_begin);
  for (int64_t r0 = r_begin; r0 < r_end; r0 += 16) {
    __syncthreads();                       // the previous step's fragment reads are done
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<f32x4*>(&sa[lr + 8 * h][lc]) = va[h];
      *reinterpret_cast<f32x4*>(&sb[lr + 8 * h][lc]) = vb[h];
    }
    __syncthreads();
    if (r0 + 16 < r_end) fetch(r0 + 16);
#pragma unroll
    for (int rr = 0; rr < 16; rr += 2) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = sa[rr + kq][wy * 64 + i * 32 + c];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = sb[rr + kq][wx * 64 + j * 32 + c];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // D[m][n]: lane holds column n = lane & 31 and rows m = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  float* p = partial + (int64_t)blockIdx.z * N * K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wx * 64 + j * 32 + c;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + wy * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kq;
        if (n < N && k < K) p[(int64_t)n * K + k] = acc[i][j][e];
      }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same product on the 16-bit matrix cores with the exact 3 x bf16 operand split (six partial products, fp32 accumulate: the accuracy of
// the fp32 kernels above at 417 instead of 157 TFLOP/s of ceiling).  v_mfma_f32_32x32x16_bf16 wants 8 consecutive reduction indices per lane,
// and the reduction index here is the ROW: a thread therefore loads 4 consecutive columns of 8 consecutive rows (eight 16-byte loads; threads
// 0..127 stage dY, 128..255 stage X), splits the 8 values of each column and writes them as ONE 16-byte LDS store per plane into the transposed
// tile [plane][column][row] -- the fragment reads are then plain 16-byte reads.  Row pitch 80 bytes (32 rows + 8 pad): conflict-free b128
// reads; the 8-row group g of column col sits at group g ^ ((col >> 4) & 3), which makes the column-strided 16-byte stores conflict-free too.
// Workgroup = 128 (n) x 128 (k) outputs, 4 waves of 64 x 64 (2 x 2 MFMA tiles), 32 rows per step; the next step's rows are fetched into
// registers while the matrix cores work.  (First version: 4-byte loads, one column of 16 rows per thread -- 64 load instructions per thread
// and step against 48 MFMAs per wave; profiles/r03g_generator_step_torch_profile.log.)
namespace wg16 {
using vsconv::bf16x8;
using vsconv::u32x4;
constexpr int RS = 32;                 // rows per step
constexpr int PITCH = 40;              // bf16 elements per LDS row (32 + 8 pad)
constexpr int TILE = 128;

__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&pl)[3]) {
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned u = __float_as_uint(v[i]);
    h[i] = u & 0xffff0000u;
    const float r = v[i] - __uint_as_float(h[i]);
    m[i] = __float_as_uint(r) & 0xffff0000u;
    l[i] = __float_as_uint(r - __uint_as_float(m[i]));
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    pl[0][q] = vsconv::pack_hi(h[2 * q], h[2 * q + 1]);
    pl[1][q] = vsconv::pack_hi(m[2 * q], m[2 * q + 1]);
    pl[2][q] = vsconv::pack_hi(l[2 * q], l[2 * q + 1]);
  }
}
}  // namespace wg16

// CONV = true: the X operand is the implicit 3 x 3 patch matrix of an NHWC image (zero or reflection padding 1, stride cs): blockIdx.x = (tap, column
// tile of the image row), row r of dY is output pixel (b, oy, ox) and reads image pixel (oy * cs + ky - 1, ox * cs + kx - 1) -- the
// weight gradient of a 3 x 3 conv without materialising rows x 9 ld floats of patches.  K is then the image's ld, dw is [N][9 * ld].
struct wgrad_conv_t { int H, W, Ho, Wo, cs, ktiles, reflect; };
template <bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_wgrad_bf16x3_kernel(const float* __restrict__ dy, int64_t dy_ld, int N, const float* __restrict__ x,
                                                                int64_t x_ld, int K, int64_t rows, int64_t rows_per_split,
                                                                float* __restrict__ partial, const wgrad_conv_t cv) {
  using namespace wg16;
  __shared__ __attribute__((aligned(16))) unsigned short sm[2][3][TILE][PITCH];          // [operand: dY, X][plane][column][row]
  const int tap = CONV ? blockIdx.x / cv.ktiles : 0;
  const int n0 = blockIdx.y * TILE, k0 = (CONV ? blockIdx.x - tap * cv.ktiles : blockIdx.x) * TILE;
  const int64_t r_begin = (int64_t)blockIdx.z * rows_per_split;
  const int64_t r_end = r_begin + rows_per_split < rows ? r_begin + rows_per_split : rows;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wy = wave >> 1, wx = wave & 1;
  const int op = threadIdx.x >> 7;                                   // which operand this thread stages
  const int cg = threadIdx.x & 31, rgp = (threadIdx.x >> 5) & 3;     // columns 4 cg .. 4 cg + 3, rows 8 rgp .. 8 rgp + 7 of the step
  const int sw = (cg >> 2) & 3;                                      // 8-row group g of column col is stored at group g ^ ((col >> 4) & 3)
  const int kq = lane >> 5, c = lane & 31;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int64_t sld = op ? x_ld : dy_ld;
  const int col0 = (op ? k0 : n0) + 4 * cg;
  const bool c_ok = col0 + 4 <= sld;                                 // the 16 bytes lie inside the row (columns past N / K only feed outputs that
  const float* src = (op ? x : dy) + col0;                           // are never stored)
  const int ky = tap / 3, kx = tap - 3 * ky;
  auto fetch = [&](f32x4 (&v)[8], const int64_t r0) __attribute__((always_inline)) {
    if (CONV && op) {                      // image pixel of each of the 8 rows: one decomposition, then a walk along the output row
      const unsigned rf = (unsigned)(r0 + rgp * 8);
      unsigned q = rf / (unsigned)cv.Wo;
      int ox = (int)(rf - q * (unsigned)cv.Wo);
      unsigned b = q / (unsigned)cv.Ho;
      int oy = (int)(q - b * (unsigned)cv.Ho);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int iy = oy * cv.cs + ky - 1, ix = ox * cv.cs + kx - 1;
        if (cv.reflect) {                  // reflection padding 1: -1 -> 1, H -> H - 2
          iy = iy < 0 ? -iy : (iy >= cv.H ? 2 * cv.H - 2 - iy : iy);
          ix = ix < 0 ? -ix : (ix >= cv.W ? 2 * cv.W - 2 - ix : ix);
        }
        const bool in = c_ok && (int64_t)rf + j < r_end && iy >= 0 && iy < cv.H && ix >= 0 && ix < cv.W;
        v[j] = in ? *reinterpret_cast<const f32x4*>(src + (((int64_t)b * cv.H + iy) * cv.W + ix) * sld) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (++ox == cv.Wo) { ox = 0; if (++oy == cv.Ho) { oy = 0; ++b; } }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t r = r0 + rgp * 8 + j;
      v[j] = (c_ok && r < r_end) ? *reinterpret_cast<const f32x4*>(src + r * sld) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // one step: split + transposed LDS stores of the rows in `v`, refill `v` with the rows TWO steps ahead (the other register set holds the
  // rows of the next step, already in flight: a load has a whole step of matrix work to arrive), fragments, 48 MFMAs per wave
  auto step = [&](f32x4 (&v)[8], const int64_t r0) __attribute__((always_inline)) {
    __syncthreads();                       // the previous step's fragment reads are done
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t[8];
      u32x4 pl[3];
#pragma unroll
      for (int q = 0; q < 8; ++q) t[q] = v[q][j];
      split8(t, pl);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(&sm[op][p][4 * cg + j][(rgp ^ sw) * 8]) = pl[p];
    }
    __syncthreads();
    if (r0 + 2 * RS < r_end) fetch(v, r0 + 2 * RS);
#pragma unroll
    for (int ks = 0; ks < RS / 16; ++ks) {
      bf16x8 fa[2][3], fb[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const int ca = wy * 64 + i * 32 + c, cb = wx * 64 + i * 32 + c;
          fa[i][p] = *reinterpret_cast<const bf16x8*>(&sm[0][p][ca][((ks * 2 + kq) ^ ((ca >> 4) & 3)) * 8]);
          fb[i][p] = *reinterpret_cast<const bf16x8*>(&sm[1][p][cb][((ks * 2 + kq) ^ ((cb >> 4) & 3)) * 8]);
        }
#pragma unroll
      for (int q = 0; q < 6; ++q)           // smallest partial products first
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = vsconv::Arith<3>::mfma(fa[i][vsconv::Arith<3>::PA[q]], fb[j][vsconv::Arith<3>::PB[q]], acc[i][j]);
    }
  };
  f32x4 v0[8], v1[8];
  fetch(v0, r_begin);
  if (r_begin + RS < r_end) fetch(v1, r_begin + RS);
  for (int64_t r0 = r_begin; r0 < r_end; r0 += 2 * RS) {
    step(v0, r0);
    if (r0 + RS < r_end) step(v1, r0 + RS);
  }
  // D[m][n]: lane holds column n = lane & 31 and rows m = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  const int64_t Kt = CONV ? 9 * (int64_t)K : K;                      // row length of dw
  float* p = partial + (int64_t)blockIdx.z * N * Kt + (CONV ? (int64_t)tap * K : 0);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wx * 64 + j * 32 + c;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + wy * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kq;
        if (n < N && k < K) p[(int64_t)n * Kt + k] = acc[i][j][e];
      }
    }
}

// ---------------------------------------------------------------------------------------------------
// Weight gradient of a THIN 3 x 3 conv (co <= 32, ci <= 96: the U-Net's outer levels, unet.py:21-27 at 256^2 / 128^2 -- a million rows against a
// 16 x 144 result) straight from the image.  These layers are not GEMM-shaped: the patch-matrix route wrote and re-read rows x 9 ld floats to
// produce a few thousand numbers.  Here a workgroup walks over 16 x 16 (or 8 x 16) tiles of output pixels; per tile the dY tile and the x tile
// with its halo go to LDS once, thread (a, b, p) owns the 9 x 4 x 4 results of output channels 4a.., input channels 4b.. and adds the pixels
// p, p + L, .. of the tile: per pixel one 16-byte read of dY and nine of x feed 144 FMAs.  The workgroup keeps its sums in registers over
// all of its tiles; at the end the L pixel lanes are added through LDS in lane order and the workgroup writes ONE partial result
// [co][9 * ld]; reduce_chunks_kernel adds the workgroups' partials (fixed grid for a shape: deterministic).
struct wgrad_thin_t { int H, W, Ho, Wo, cs, TH, tw_shift, tiles_x, tiles_y, ntiles, co, L, reflect; };

__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_thin_kernel(const float* __restrict__ dy, int64_t dy_ld, const float* __restrict__ x, int64_t ld,
                                                                    const wgrad_thin_t g, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nb = (int)(ld >> 2), na = g.co >> 2, combos = na * nb;
  const int TW = 1 << g.tw_shift, TH = g.TH, cs = g.cs;
  const int IW = (TW - 1) * cs + 3, IH = (TH - 1) * cs + 3;
  float* sx = lds;                                   // [IH * IW][ld]
  float* sd = lds + (int64_t)IH * IW * ld;           // [TH * TW][co]
  const int t = threadIdx.x;
  const int p = t / combos, cb = t - p * combos;
  const int a = cb / nb, b = cb - a * nb;
  const bool active = p < g.L;
  f32x4 acc[9][4];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[k][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
    const int tx = tile % g.tiles_x, q = tile / g.tiles_x;
    const int ty = q % g.tiles_y, bi = q / g.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * cs - 1, ix0 = ox0 * cs - 1;
    __syncthreads();                                 // the previous tile's reads are done
    for (int i = t; i < IH * IW * nb; i += 256) {
      const int c4 = i % nb, pix = i / nb;
      const int yy = pix / IW, xx = pix - yy * IW;
      int gy = iy0 + yy, gx = ix0 + xx;
      if (g.reflect) {                               // the one-pixel ring of reflection padding; further out only dY = 0 pixels look
        gy = gy == -1 ? 1 : (gy == g.H ? g.H - 2 : gy);
        gx = gx == -1 ? 1 : (gx == g.W ? g.W - 2 : gx);
      }
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) v = *reinterpret_cast<const f32x4*>(x + (((int64_t)bi * g.H + gy) * g.W + gx) * ld + 4 * c4);
      *reinterpret_cast<f32x4*>(sx + (int64_t)pix * ld + 4 * c4) = v;
    }
    for (int i = t; i < TH * TW * na; i += 256) {
      const int c4 = i % na, pix = i / na;
      const int oy = oy0 + (pix >> g.tw_shift), ox = ox0 + (pix & (TW - 1));
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (oy < g.Ho && ox < g.Wo) v = *reinterpret_cast<const f32x4*>(dy + (((int64_t)bi * g.Ho + oy) * g.Wo + ox) * dy_ld + 4 * c4);
      *reinterpret_cast<f32x4*>(sd + (int64_t)pix * g.co + 4 * c4) = v;
    }
    __syncthreads();
    if (active)
      for (int i = p; i < TH * TW; i += g.L) {
        const int py = i >> g.tw_shift, px = i & (TW - 1);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(sd + i * g.co + 4 * a);
        const float* xb = sx + ((int64_t)(py * cs) * IW + px * cs) * ld + 4 * b;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const f32x4 x4 = *reinterpret_cast<const f32x4*>(xb + (int64_t)(ky * IW + kx) * ld);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[ky * 3 + kx][e] += d4[e] * x4;
          }
      }
  }
  // the L pixel lanes of every (a, b) are added in lane order, one tap at a time through 16 KB of LDS
  const int64_t K9 = 9 * ld;
  float* out = partial + (int64_t)blockIdx.x * g.co * K9;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) *reinterpret_cast<f32x4*>(lds + t * 16 + 4 * e) = acc[k][e];
    __syncthreads();
    for (int idx = t; idx < combos * 16; idx += 256) {
      const int c2 = idx >> 4, e = idx & 15;
      float sum = 0.f;
      for (int pp = 0; pp < g.L; ++pp) sum += lds[(pp * combos + c2) * 16 + e];
      const int a2 = c2 / nb, b2 = c2 - a2 * nb;
      out[(int64_t)(4 * a2 + (e >> 2)) * K9 + k * ld + 4 * b2 + (e & 3)] = sum;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Depthwise 7x7, zero padding 3, NHWC: out = (bias) + sum_t w[t'][c] x[.., y+ky-3, x+kx-3, c] (+ add),  t' = flip ? 48 - t : t.
// flip = 1 is the backward-data pass (convnext.py:43 transposed).  Work item = (4 consecutive x, 4 channels).
__global__ __launch_bounds__(256) void dwconv7_kernel(const float* __restrict__ x, int B, int H, int W, int64_t ld,
                                                      const float* __restrict__ w, const float* __restrict__ bias, int flip,
                                                      const float* __restrict__ add, int64_t add_ld, float* __restrict__ out,
                                                      int64_t out_ld, int C4, int spr, int64_t nitems) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nitems) return;
  const int cg = (int)(idx % C4);
  const int64_t sidx = idx / C4;
  const int xs = (int)(sidx % spr);
  const int64_t t = sidx / spr;
  const int y = (int)(t % H), b = (int)(t / H);
  const int c = cg * 4, x0 = xs * 4;
  f32x4 acc[4];
  const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 4; ++p) acc[p] = bv;
  const float* base = x + (int64_t)b * H * W * ld + c;
  for (int ky = 0; ky < 7; ++ky) {
    const int iy = y + ky - 3;
    if (iy < 0 || iy >= H) continue;
    f32x4 in[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const int ix = x0 - 3 + j;
      in[j] = (ix >= 0 && ix < W) ? *reinterpret_cast<const f32x4*>(base + ((int64_t)iy * W + ix) * ld) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      const int tt = flip ? 48 - (ky * 7 + kx) : ky * 7 + kx;
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (int64_t)tt * ld + c);
#pragma unroll
      for (int p = 0; p < 4; ++p) acc[p] += in[p + kx] * wv;
    }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if (x0 + p >= W) break;
    const int64_t pix = ((int64_t)b * H + y) * W + x0 + p;
    f32x4 v = acc[p];
    if (add) v += *reinterpret_cast<const f32x4*>(add + pix * add_ld + c);
    *reinterpret_cast<f32x4*>(out + pix * out_ld + c) = v;
  }
}

// d w[t][c] = sum_{b,y,x} dy[b,y,x,c] * x[b, y+ky-3, x+kx-3, c]: chunk = (frame, band of RB image rows), thread items = (tap ROW ky, 4 channels).
// An item walks along the image row with a sliding window of seven x pixels in registers, so every x / dy value is loaded once per tap row and
// feeds 7 x 4 FMAs (the first version had one item per tap: two loads per 4 FMAs, load-issue bound).  The walk is unrolled by seven: the
// window slot of pixel u is u % 7, a compile-time index.  partial[chunk][49][ld] in fp32 (RB * W terms each), summed over the chunks by
// reduce_chunks_kernel.
__global__ __launch_bounds__(256) void dwconv7_wgrad_kernel(const float* __restrict__ x, int64_t ld, const float* __restrict__ dy,
                                                            int64_t dy_ld, int H, int W, int RB, int nband, int C4,
                                                            float* __restrict__ partial) {
  const int b = blockIdx.x / nband, band = blockIdx.x % nband;
  const int ya = band * RB, yb = ya + RB < H ? ya + RB : H;
  const float* xb = x + (int64_t)b * H * W * ld;
  const float* db = dy + (int64_t)b * H * W * dy_ld;
  for (int it = threadIdx.x; it < 7 * C4; it += 256) {
    const int ky = it / C4, cg = it - ky * C4;
    f32x4 acc[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int y = ya; y < yb; ++y) {
      const int iy = y + ky - 3;
      if (iy < 0 || iy >= H) continue;
      const float* xr = xb + (int64_t)iy * W * ld + 4 * cg;
      const float* gr = db + (int64_t)y * W * dy_ld + 4 * cg;
      f32x4 win[7];                        // slot (u + 3) % 7 holds x[u] for u = xx - 3 .. xx + 3 (zeros outside the row)
#pragma unroll
      for (int k = 0; k < 7; ++k) win[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (u < W) win[u + 3] = *reinterpret_cast<const f32x4*>(xr + (int64_t)u * ld);
      for (int x0 = 0; x0 < W; x0 += 7) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          const int xx = x0 + j;
          if (xx < W) {
            // x[xx + 3] enters slot (xx + 6) % 7 = (j + 6) % 7 (x0 is a multiple of 7), replacing x[xx - 4]
            win[(j + 6) % 7] = xx + 3 < W ? *reinterpret_cast<const f32x4*>(xr + (int64_t)(xx + 3) * ld) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 g = *reinterpret_cast<const f32x4*>(gr + (int64_t)xx * dy_ld);
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) acc[kx] += g * win[(j + kx) % 7];       // x[xx + kx - 3] sits in slot (xx + kx) % 7
          }
        }
      }
    }
#pragma unroll
    for (int kx = 0; kx < 7; ++kx)
      *reinterpret_cast<f32x4*>(partial + ((int64_t)blockIdx.x * 49 + ky * 7 + kx) * ld + 4 * cg) = acc[kx];
  }
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm over C (biased variance, eps), backward per row; one wave per row:
//   xhat = (x - mean) * rstd,  g = dy * w,  dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat));   stats[row] = (mean, rstd)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, int64_t ld, const float* __restrict__ dy, int64_t dy_ld,
                                                     const float* __restrict__ w, int64_t rows, int C, float eps, float* __restrict__ dx,
                                                     int64_t dx_ld, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ld;
  const float* gr = dy + row * dy_ld;
  const float invC = 1.0f / (float)C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  const float mean = wave_sum(s) * invC;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; v += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(v) * invC + eps);
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float g = gr[c] * w[c];
    s1 += g;
    s2 += g * ((xr[c] - mean) * rstd);
  }
  s1 = wave_sum(s1) * invC;
  s2 = wave_sum(s2) * invC;
  float* o = dx + row * dx_ld;
  for (int c = lane; c < (int)dx_ld; c += 64) o[c] = c < C ? rstd * (gr[c] * w[c] - s1 - ((xr[c] - mean) * rstd) * s2) : 0.f;
  if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// Column reductions over chunks of CR_ROWS rows that never straddle a frame (HW rows per frame; LayerNorm passes B = 1, HW = rows):
//   MODE 0 (LayerNorm parameters):  p0 = sum b * xhat(a),  p1 = sum b          a = x (stats = mean / rstd per row), b = dy
//   MODE 1 (GELU + GRN):            p0 = sum gelu(a)^2,    p1 = sum b * gelu(a),  p2 = sum b        a = h1 (pre-GELU), b = d3
// partial[(frame * nchf + chunk)][NV][ldp] in fp32.  Thread = (4 channels, row lane); row lanes are combined through LDS in a fixed order.
template <int MODE>
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ a, int64_t a_ld, const float* __restrict__ b, int64_t b_ld,
                                                        const float* __restrict__ stats, int HW, int nchf, int C4, int64_t ldp,
                                                        float* __restrict__ partial) {
  constexpr int NV = MODE == 0 ? 2 : 3;
  __shared__ __attribute__((aligned(16))) float red[NV][256][4];
  const int frame = blockIdx.x / nchf, ch = blockIdx.x % nchf;
  const int64_t r0 = (int64_t)frame * HW + (int64_t)ch * CR_ROWS;
  const int64_t r1 = (int64_t)frame * HW + (((int64_t)ch + 1) * CR_ROWS < HW ? ((int64_t)ch + 1) * CR_ROWS : HW);
  const int G = C4 < 256 ? C4 : 256;                   // channel groups handled per workgroup (blockIdx.y = sweep)
  const int RL = 256 / G;                              // row lanes per channel group
  const int g = threadIdx.x % G, rl = threadIdx.x / G;
  const int gg = blockIdx.y * G + g;
  f32x4 acc[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto term = [&](const f32x4& va, const f32x4& vb, const int64_t r) __attribute__((always_inline)) {
    if constexpr (MODE == 0) {
      const float mean = stats[2 * r], rstd = stats[2 * r + 1];
      acc[0] += vb * ((va - mean) * rstd);
      acc[1] += vb;
    } else {
      f32x4 ge;
#pragma unroll
      for (int e = 0; e < 4; ++e) ge[e] = vs_gelu(va[e]);
      acc[0] += ge * ge;
      acc[1] += vb * ge;
      acc[2] += vb;
    }
  };
  if (rl < RL && gg < C4) {
    int64_t r = r0 + rl;
    for (; r + 3 * RL < r1; r += 4 * RL) {               // four rows in flight, added in row order
      f32x4 va[4], vb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        va[u] = *reinterpret_cast<const f32x4*>(a + (r + u * RL) * a_ld + 4 * gg);
        vb[u] = *reinterpret_cast<const f32x4*>(b + (r + u * RL) * b_ld + 4 * gg);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) term(va[u], vb[u], r + u * RL);
    }
    for (; r < r1; r += RL)
      term(*reinterpret_cast<const f32x4*>(a + r * a_ld + 4 * gg), *reinterpret_cast<const f32x4*>(b + r * b_ld + 4 * gg), r);
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) *reinterpret_cast<f32x4*>(&red[k][threadIdx.x][0]) = acc[k];
  __syncthreads();
  if (rl == 0 && gg < C4) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < RL; ++j) s += *reinterpret_cast<const f32x4*>(&red[k][j * G + g][0]);
      *reinterpret_cast<f32x4*>(partial + ((int64_t)blockIdx.x * NV + k) * ldp + 4 * gg) = s;
    }
  }
}

// GRN, per (frame, channel): sums over the chunks of the frame -> G = ||gelu(h1)||_2, s = sum d3 * gelu(h1), t = sum d3
__global__ __launch_bounds__(256) void grn_sums_kernel(const float* __restrict__ partial, int nchf, int64_t ldp, int C, float* __restrict__ G,
                                                       float* __restrict__ s, float* __restrict__ t) {
  const int frame = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double g2 = 0, ss = 0, tt = 0;
  const float* p0 = partial + (int64_t)frame * nchf * 3 * ldp + c;
  int k = 0;
  for (; k + 4 <= nchf; k += 4) {                       // twelve loads in flight, added in chunk order
    float v[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < 3; ++q) v[u][q] = p0[((int64_t)(k + u) * 3 + q) * ldp];
#pragma unroll
    for (int u = 0; u < 4; ++u) { g2 += (double)v[u][0]; ss += (double)v[u][1]; tt += (double)v[u][2]; }
  }
  for (; k < nchf; ++k) {
    const float* p = p0 + (int64_t)k * 3 * ldp;
    g2 += (double)p[0];
    ss += (double)p[ldp];
    tt += (double)p[2 * ldp];
  }
  G[(int64_t)frame * ldp + c] = (float)sqrt(g2);
  s[(int64_t)frame * ldp + c] = (float)ss;
  t[(int64_t)frame * ldp + c] = (float)tt;
}

// fp64 sum of one value per thread over the workgroup (256 threads), fixed tree order; every thread gets the total
__device__ __forceinline__ double block_sum64(double v, double* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

// GRN coupling, one workgroup per frame (common.py:163-168): Nx = G / (mean_c G + 1e-6), h3 = gamma * (h2 * Nx) + beta + h2
//   d h2 = d3 * ca + h2 * cb,   ca = gamma * Nx + 1,   cb = dG / G,   dG = dNx / (M + eps) - sum_c(dNx * G) / (C (M + eps)^2),  dNx = gamma * s
__global__ __launch_bounds__(256) void grn_coef_kernel(const float* __restrict__ G, const float* __restrict__ s, const float* __restrict__ gamma,
                                                       int C, int64_t ldp, float* __restrict__ ca, float* __restrict__ cb,
                                                       float* __restrict__ nx) {
  __shared__ double sh[256];
  const int64_t off = (int64_t)blockIdx.x * ldp;
  double sg = 0, sd = 0;
  for (int c = threadIdx.x; c < C; c += 256) {
    sg += (double)G[off + c];
    sd += (double)gamma[c] * (double)s[off + c] * (double)G[off + c];
  }
  const double M = block_sum64(sg, sh) / (double)C + 1e-6;
  const double SD = block_sum64(sd, sh);
  for (int c = threadIdx.x; c < (int)ldp; c += 256) {
    float a = 0.f, bq = 0.f, n = 0.f;
    if (c < C) {
      const double g = (double)G[off + c];
      const double nxc = g / M;
      const double dG = (double)gamma[c] * (double)s[off + c] / M - SD / ((double)C * M * M);
      n = (float)nxc;
      a = (float)((double)gamma[c] * nxc + 1.0);
      bq = g > 0 ? (float)(dG / g) : 0.f;
    }
    ca[off + c] = a;
    cb[off + c] = bq;
    nx[off + c] = n;
  }
}

// d gamma[c] = sum_b s[b][c] * Nx[b][c],  d beta[c] = sum_b t[b][c]
__global__ __launch_bounds__(256) void grn_param_grad_kernel(const float* __restrict__ s, const float* __restrict__ t, const float* __restrict__ nx,
                                                             int B, int C, int64_t ldp, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double a = 0, b = 0;
  for (int f = 0; f < B; ++f) {
    a += (double)s[(int64_t)f * ldp + c] * (double)nx[(int64_t)f * ldp + c];
    b += (double)t[(int64_t)f * ldp + c];
  }
  dgamma[c] = (float)a;
  dbeta[c] = (float)b;
}

// d h1 = (d3 * ca[b][c] + gelu(h1) * cb[b][c]) * gelu'(h1);  columns >= C of the output row are zeroed
__global__ __launch_bounds__(256) void gelu_grn_bwd_kernel(const float* __restrict__ h1, int64_t ld, const float* __restrict__ d3, int64_t d3_ld,
                                                           const float* __restrict__ ca, const float* __restrict__ cb, int64_t ldp, int HW, int C,
                                                           int C4, int64_t total, float* __restrict__ out, int64_t out_ld) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cg = (int)(idx % C4);
  const int64_t r = idx / C4;
  const int64_t f = r / HW;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  if (4 * cg < C) {
    const f32x4 h = *reinterpret_cast<const f32x4*>(h1 + r * ld + 4 * cg);
    const f32x4 d = *reinterpret_cast<const f32x4*>(d3 + r * d3_ld + 4 * cg);
    const f32x4 a = *reinterpret_cast<const f32x4*>(ca + f * ldp + 4 * cg);
    const f32x4 b = *reinterpret_cast<const f32x4*>(cb + f * ldp + 4 * cg);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * cg + e < C) o[e] = (d[e] * a[e] + vs_gelu(h[e]) * b[e]) * vs_gelu_grad(h[e]);
  }
  *reinterpret_cast<f32x4*>(out + r * out_ld + 4 * cg) = o;
}

// ---------------------------------------------------------------------------------------------------
// P x P, stride P patch convs (ConvNeXt stem and downsample layers): patch matrix in the k order of engine.pack_patch_conv,
// k = ky * CP + kx * pld + c (CP = P * pld rounded up to 16), and its adjoint
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, int H, int W, int64_t pld, int P, int S, int Ho, int Wo, int CP,
                                                       int64_t total, float* __restrict__ cols) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (row_out, ky, e4)
  if (idx >= total) return;
  const int E4 = CP / 4;
  const int e4 = (int)(idx % E4);
  const int64_t t = idx / E4;
  const int ky = (int)(t % P);
  const int64_t ro = t / P;
  const int ox = (int)(ro % Wo);
  const int64_t t2 = ro / Wo;
  const int oy = (int)(t2 % Ho);
  const int64_t b = t2 / Ho;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if ((int64_t)4 * e4 < (int64_t)P * pld) v = *reinterpret_cast<const f32x4*>(x + ((b * H + oy * S + ky) * W + (int64_t)ox * S) * pld + 4 * e4);
  *reinterpret_cast<f32x4*>(cols + (ro * P + ky) * CP + 4 * e4) = v;
}
__global__ __launch_bounds__(256) void unpatch_kernel(const float* __restrict__ dcols, int H, int W, int64_t pld, int P, int S, int Ho, int Wo, int CP,
                                                      int64_t total, float* __restrict__ dx) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (pixel, c4)
  if (idx >= total) return;
  const int C4 = (int)(pld / 4);
  const int c4 = (int)(idx % C4);
  const int64_t pix = idx / C4;
  const int xx = (int)(pix % W);
  const int64_t t = pix / W;
  const int yy = (int)(t % H);
  const int64_t b = t / H;
  // the patches (oy, ox) with oy * S <= yy < oy * S + P (and the same in x), in a fixed order: S == P is the non-overlapping case (one patch;
  // pixels outside the last whole patch receive no gradient), S < P the overlapping stem of ChunkySeal (4 x 4 stride 2: up to 2 x 2 patches)
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  const int oy1 = yy / S, ox1 = xx / S;
  for (int oy = oy1 - (P - 1) / S; oy <= oy1; ++oy) {
    if (oy < 0 || oy >= Ho || yy - oy * S >= P) continue;
    for (int ox = ox1 - (P - 1) / S; ox <= ox1; ++ox) {
      if (ox < 0 || ox >= Wo || xx - ox * S >= P) continue;
      v += *reinterpret_cast<const f32x4*>(dcols + (((b * Ho + oy) * Wo + ox) * P + (yy - oy * S)) * CP + (int64_t)(xx - ox * S) * pld + 4 * c4);
    }
  }
  *reinterpret_cast<f32x4*>(dx + pix * pld + 4 * c4) = v;
}

// adjoint of vs_im2col3x3 with reflection padding 1 (cols[row][tap * ld + c] = x[refl(y + ky - 1)][refl(x + kx - 1)][c]): gather form.
// Output pixel (y, x) collects, per tap, the rows whose (reflected) source it is: o = y - k + 1, plus o = 0 for (y = 1, k = 0) and
// o = H - 1 for (y = H - 2, k = 2).
__device__ __forceinline__ int refl_srcs(int y, int k, int H, int (&o)[2]) {
  int n = 0;
  const int o1 = y - k + 1;
  if (o1 >= 0 && o1 < H) o[n++] = o1;
  if (k == 0 && y == 1) o[n++] = 0;
  if (k == 2 && y == H - 2) o[n++] = H - 1;
  return n;
}
__global__ __launch_bounds__(256) void col2im3x3_reflect_kernel(const float* __restrict__ dcols, int H, int W, int64_t ld, int64_t total,
                                                                float* __restrict__ dx) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (pixel, c4)
  if (idx >= total) return;
  const int C4 = (int)(ld / 4);
  const int c4 = (int)(idx % C4);
  const int64_t pix = idx / C4;
  const int xx = (int)(pix % W);
  const int64_t t = pix / W;
  const int yy = (int)(t % H);
  const int64_t b = t / H;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int ky = 0; ky < 3; ++ky) {
    int oys[2];
    const int ny = refl_srcs(yy, ky, H, oys);
    for (int kx = 0; kx < 3; ++kx) {
      int oxs[2];
      const int nxs = refl_srcs(xx, kx, W, oxs);
      for (int i = 0; i < ny; ++i)
        for (int j = 0; j < nxs; ++j)
          acc += *reinterpret_cast<const f32x4*>(dcols + ((b * H + oys[i]) * W + oxs[j]) * 9 * ld + (int64_t)(ky * 3 + kx) * ld + 4 * c4);
    }
  }
  *reinterpret_cast<f32x4*>(dx + pix * ld + 4 * c4) = acc;
}

// ---------------------------------------------------------------------------------------------------
// head: pooled[b][c] = mean_p x[b][p][c];  dz[b][p][c] = dpooled[b][c] / HW * gelu'(z[b][p][c])
__global__ __launch_bounds__(256) void colmean_kernel(const float* __restrict__ x, int HW, int64_t ld, int C4, int64_t total, float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (frame, c4)
  if (idx >= total) return;
  const int c4 = (int)(idx % C4);
  const int64_t f = idx / C4;
  double s[4] = {0, 0, 0, 0};
  for (int p = 0; p < HW; ++p) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (f * HW + p) * ld + 4 * c4);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += (double)v[e];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) out[f * ld + 4 * c4 + e] = (float)(s[e] / (double)HW);
}
__global__ __launch_bounds__(256) void pool_gelu_bwd_kernel(const float* __restrict__ z, int64_t ld, const float* __restrict__ dpooled,
                                                            int64_t dp_ld, int HW, int C, int C4, int64_t total, float* __restrict__ dz,
                                                            int64_t dz_ld) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (row, c4)
  if (idx >= total) return;
  const int c4 = (int)(idx % C4);
  const int64_t r = idx / C4;
  const int64_t f = r / HW;
  const float inv = 1.0f / (float)HW;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (4 * c4 + e < C) o[e] = dpooled[f * dp_ld + 4 * c4 + e] * inv * vs_gelu_grad(z[r * ld + 4 * c4 + e]);
  *reinterpret_cast<f32x4*>(dz + r * dz_ld + 4 * c4) = o;
}
// C[m][n] = sum_k A[m][k] * Bm[k][n]: the head's few-row products (dpooled = dlogits * W_lin); one thread per output, fixed order
__global__ __launch_bounds__(256) void matmul_small_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm, int64_t ldb,
                                                           int M, int N, int K, float* __restrict__ Cc, int64_t ldc) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)M * N) return;
  const int n = (int)(idx % N);
  const int64_t m = idx / N;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s = __builtin_fmaf(A[m * lda + k], Bm[(int64_t)k * ldb + n], s);
  Cc[m * ldc + n] = s;
}

// Decoding loss (videosealloss.py:150-156: BCEWithLogits on preds[:, 1:] / T against the message bits, mean over B * k) and its gradient
// d preds[b][1 + j] = gscale / T * (sigmoid(z) - m) / (B k),  d preds[b][0] = 0.   One workgroup; fp64 sum in a fixed order.
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ preds, const int* __restrict__ msgs, int msg_rows, int B, int k,
                                                         float inv_t, float gscale, float* __restrict__ dpreds, float* __restrict__ loss) {
  __shared__ double sh[256];
  const int64_t n = (int64_t)B * k;
  const float gs = gscale * inv_t / (float)n;
  double acc = 0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const int b = (int)(i / k), j = (int)(i % k);
    const float z = preds[(int64_t)b * (k + 1) + 1 + j] * inv_t;
    const float m = (float)msgs[(int64_t)(msg_rows == 1 ? 0 : b) * k + j];
    acc += (double)(fmaxf(z, 0.f) - z * m + log1pf(__expf(-fabsf(z))));
    const float sg = 1.0f / (1.0f + __expf(-z));
    dpreds[(int64_t)b * (k + 1) + 1 + j] = gs * (sg - m);
  }
  for (int b = threadIdx.x; b < B; b += 256) dpreds[(int64_t)b * (k + 1)] = 0.f;
  const double tot = block_sum64(acc, sh);
  if (threadIdx.x == 0) loss[0] = (float)(tot / (double)n);
}

static inline unsigned blocks_for(int64_t n) { return (unsigned)cdiv64(n, 256); }

}  // namespace

// ===================================================================================================== C-ABI
// kernel choice and number of row slices: functions of the shape (and of VS_WGRAD, read once) only -- the summation order never depends on the launch
static int wgrad_mode() {
  static const int mode = [] { const char* e = getenv("VS_WGRAD"); return !e ? 0 : (!strcmp(e, "mfma") ? 1 : (!strcmp(e, "fma") ? 2 : 0)); }();
  return mode;
}
static int64_t wgrad_splits(int64_t rows, int N, int64_t K) {
  int64_t splits, maxs = cdiv64(rows, 256);
  if (wgrad_mode() == 0 && N >= 64 && K >= 64)           // 128 x 128 tiles, two workgroups per CU: aim at 512 workgroups
    splits = 512 / (cdiv64(N, 128) * cdiv64(K, 128));
  else
    splits = cdiv64(1024, cdiv64(N, 64) * cdiv64(K, 64));
  if (splits > maxs) splits = maxs;
  return splits < 1 ? 1 : splits;
}

extern "C" int64_t vs_gemm_wgrad_partial_floats(int64_t rows, int N, int K) {
  if (rows <= 0 || N <= 0 || K <= 0) return 0;
  return wgrad_splits(rows, N, K) * (int64_t)N * K;
}

extern "C" int vs_gemm_wgrad(const float* dy, int64_t dy_ld, int N, const float* x, int64_t x_ld, int K, int64_t rows, float* partial,
                             float* dw, void* stream) {
  VS_REQUIRE(dy && x && partial && dw && rows > 0 && N > 0 && K > 0 && dy_ld >= N && x_ld >= K && (dy_ld & 3) == 0 && (x_ld & 3) == 0);
  VS_REQUIRE((((uintptr_t)dy) & 15) == 0 && (((uintptr_t)x) & 15) == 0);
  const int64_t splits = vs_gemm_wgrad_partial_floats(rows, N, K) / ((int64_t)N * K);
  int64_t rps = cdiv64(rows, splits);
  rps = cdiv64(rps, 32) * 32;
  const int64_t used = cdiv64(rows, rps);                  // <= splits
  // VS_WGRAD=mfma selects the fp32 matrix-core kernel (validated on hardware: tests/test_gpu_bwd.py passes with it; v_mfma_f32_32x32x2_f32 has the
  // same 157 TFLOP/s peak as the vector FMAs, and the step time is the same within noise -- profiles/r03a_*).
  const int mode = wgrad_mode();
  // default: the 3 x bf16 matrix-core kernel where a 128 x 128 tile is at least half full, the fp32 vector kernel for the thin layers
  // (VS_WGRAD=fma / mfma force the two fp32 kernels everywhere)
  if (mode == 0 && N >= 64 && K >= 64)
    hipLaunchKernelGGL(gemm_wgrad_bf16x3_kernel<false>, dim3((unsigned)cdiv64(K, 128), (unsigned)cdiv64(N, 128), (unsigned)used), dim3(256), 0,
                       (hipStream_t)stream, dy, dy_ld, N, x, x_ld, K, rows, rps, partial, wgrad_conv_t{});
  else if (mode == 1)
    hipLaunchKernelGGL(gemm_wgrad_mfma_kernel, dim3((unsigned)cdiv64(K, 128), (unsigned)cdiv64(N, 128), (unsigned)used), dim3(256), 0,
                       (hipStream_t)stream, dy, dy_ld, N, x, x_ld, K, rows, rps, partial);
  else
    hipLaunchKernelGGL(gemm_wgrad_kernel, dim3((unsigned)cdiv64(K, 64), (unsigned)cdiv64(N, 64), (unsigned)used), dim3(256), 0, (hipStream_t)stream,
                       dy, dy_ld, N, x, x_ld, K, rows, rps, partial);
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)cdiv64((int64_t)N * K, 64)), dim3(256), 0, (hipStream_t)stream, partial, (int)used,
                     (int64_t)N * K, dw);
  return vs_launch_status();
}

// Weight gradient of a 3 x 3 conv (zero padding 1, stride 1 or 2) straight from the NHWC image: dw[n][tap * ld + c] = sum over output pixels of
// dy[b, oy, ox, n] * x[b, oy * stride + ky - 1, ox * stride + kx - 1, c] -- what vs_im2col3x3(_strided) + vs_gemm_wgrad compute, without the patch
// matrix; pad_mode VS_PAD_REFLECT (stride 1) is the Upsample conv of unet.py:170-197.  Two kernels: the matrix-core kernel on the implicit patch
// matrix (N >= 64 and ld >= 64) and the register-tile kernel of the thin outer levels (N <= 32, ld <= 96, N % 4 == 0); other shapes are
// not supported (the host keeps the patch-matrix route for them).
static bool c3w_mfma(int N, int64_t ld) { return N >= 64 && ld >= 64; }
static size_t c3w_thin_bytes(int N, int64_t ld, int stride, int TH) {
  return (size_t)(((TH - 1) * stride + 3) * (15 * stride + 3) * ld + TH * 16 * N) * sizeof(float);
}
static bool c3w_thin(int N, int64_t ld, int stride) {
  return N >= 4 && N <= 32 && (N & 3) == 0 && ld >= 4 && ld <= 96 && (ld & 3) == 0 && (N >> 2) * (ld >> 2) <= 256 &&
         c3w_thin_bytes(N, ld, stride, 2) <= ((size_t)40 << 10);
}
static wgrad_thin_t c3w_thin_geom(int N, int64_t ld, int B, int H, int W, int stride, int reflect, size_t* lds_bytes, int* nwg) {
  wgrad_thin_t g{};
  g.H = H; g.W = W; g.cs = stride; g.co = N; g.reflect = reflect;
  g.Ho = (H - 1) / stride + 1; g.Wo = (W - 1) / stride + 1;
  g.tw_shift = 4;
  g.TH = 16;
  while (g.TH > 2 && c3w_thin_bytes(N, ld, stride, g.TH) > ((size_t)40 << 10)) g.TH >>= 1;        // two workgroups per CU with room to spare
  g.tiles_x = (int)cdiv64(g.Wo, 16); g.tiles_y = (int)cdiv64(g.Ho, g.TH);
  g.ntiles = B * g.tiles_x * g.tiles_y;
  const int combos = (N >> 2) * (int)(ld >> 2);
  g.L = 256 / combos;
  const size_t tb = c3w_thin_bytes(N, ld, stride, g.TH);
  *lds_bytes = tb > ((size_t)16 << 10) ? tb : ((size_t)16 << 10);
  *nwg = g.ntiles < 512 ? g.ntiles : 512;
  return g;
}

extern "C" int vs_conv3x3_wgrad_supported(int N, int64_t ld, int stride) {
  return ((stride == 1 || stride == 2) && (c3w_mfma(N, ld) || c3w_thin(N, ld, stride))) ? 1 : 0;
}

extern "C" int64_t vs_conv3x3_wgrad_partial_floats(int N, int64_t ld, int B, int H, int W, int stride) {
  if (N <= 0 || ld <= 0 || B <= 0 || H <= 0 || W <= 0 || !vs_conv3x3_wgrad_supported(N, ld, stride)) return 0;
  if (c3w_mfma(N, ld)) return vs_gemm_wgrad_partial_floats((int64_t)B * ((H - 1) / stride + 1) * ((W - 1) / stride + 1), N, (int)(9 * ld));
  size_t lb; int nwg;
  c3w_thin_geom(N, ld, B, H, W, stride, 0, &lb, &nwg);
  return (int64_t)nwg * N * 9 * ld;
}

extern "C" int vs_conv3x3_wgrad(const float* dy, int64_t dy_ld, int N, const float* x, int64_t ld, int B, int H, int W, int stride, int pad_mode,
                                float* partial, float* dw, void* stream) {
  VS_REQUIRE(dy && x && partial && dw && vs_conv3x3_wgrad_supported(N, ld, stride) && B > 0 && H > 0 && W > 0 && dy_ld >= N &&
             (dy_ld & 3) == 0 && (ld & 3) == 0 && (pad_mode == VS_PAD_ZERO || (pad_mode == VS_PAD_REFLECT && stride == 1 && H >= 2 && W >= 2)));
  VS_REQUIRE((((uintptr_t)dy) & 15) == 0 && (((uintptr_t)x) & 15) == 0);
  const int reflect = pad_mode == VS_PAD_REFLECT ? 1 : 0;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int64_t rows = (int64_t)B * Ho * Wo;
  VS_REQUIRE(rows < ((int64_t)1 << 31));
  const int64_t K9 = 9 * ld;
  int64_t used;
  if (c3w_mfma(N, ld)) {
    const int64_t splits = vs_gemm_wgrad_partial_floats(rows, N, (int)K9) / ((int64_t)N * K9);
    int64_t rps = cdiv64(rows, splits);
    rps = cdiv64(rps, 32) * 32;
    used = cdiv64(rows, rps);
    const int ktiles = (int)cdiv64(ld, 128);
    hipLaunchKernelGGL(gemm_wgrad_bf16x3_kernel<true>, dim3((unsigned)(9 * ktiles), (unsigned)cdiv64(N, 128), (unsigned)used), dim3(256), 0,
                       (hipStream_t)stream, dy, dy_ld, N, x, ld, (int)ld, rows, rps, partial, wgrad_conv_t{H, W, Ho, Wo, stride, ktiles, reflect});
  } else {
    size_t lb; int nwg;
    const wgrad_thin_t g = c3w_thin_geom(N, ld, B, H, W, stride, reflect, &lb, &nwg);
    hipLaunchKernelGGL(conv3x3_wgrad_thin_kernel, dim3((unsigned)nwg), dim3(256), lb, (hipStream_t)stream, dy, dy_ld, x, ld, g, partial);
    used = nwg;
  }
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)cdiv64((int64_t)N * K9, 64)), dim3(256), 0, (hipStream_t)stream, partial, (int)used,
                     (int64_t)N * K9, dw);
  return vs_launch_status();
}

extern "C" int vs_dwconv7(const float* x, int B, int H, int W, int C, int64_t ld, const float* w, const float* bias, int flip,
                          const float* add, int64_t add_ld, float* out, int64_t out_ld, void* stream) {
  VS_REQUIRE(x && w && out && B > 0 && H > 0 && W > 0 && C > 0 && ld >= C && (ld & 3) == 0 && out_ld >= ld && (out_ld & 3) == 0);
  VS_REQUIRE(!add || (add_ld >= ld && (add_ld & 3) == 0));
  VS_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)out) & 15) == 0 && (((uintptr_t)w) & 15) == 0);
  const int C4 = (int)(ld >> 2), spr = (W + 3) / 4;
  const int64_t nitems = (int64_t)B * H * spr * C4;
  hipLaunchKernelGGL(dwconv7_kernel, dim3(blocks_for(nitems)), dim3(256), 0, (hipStream_t)stream, x, B, H, W, ld, w, bias, flip, add, add_ld, out,
                     out_ld, C4, spr, nitems);
  return vs_launch_status();
}

// rows per chunk: single rows up to 2048 chunks, then bands (shape-only rule)
static int vs_dwconv7_wgrad_rb(int B, int H) { return (int)cdiv64((int64_t)B * H, 2048); }
extern "C" int64_t vs_dwconv7_wgrad_partial_floats(int B, int H, int64_t ld) { return (int64_t)B * cdiv64(H, vs_dwconv7_wgrad_rb(B, H)) * 49 * ld; }

extern "C" int vs_dwconv7_wgrad(const float* x, int64_t ld, const float* dy, int64_t dy_ld, int B, int H, int W, int C, float* partial,
                                float* dw, void* stream) {
  VS_REQUIRE(x && dy && partial && dw && B > 0 && H > 0 && W > 0 && C > 0 && ld >= C && (ld & 3) == 0 && dy_ld >= ld && (dy_ld & 3) == 0);
  VS_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)dy) & 15) == 0);
  const int RB = vs_dwconv7_wgrad_rb(B, H), nband = (int)cdiv64(H, RB);
  hipLaunchKernelGGL(dwconv7_wgrad_kernel, dim3((unsigned)(B * nband)), dim3(256), 0, (hipStream_t)stream, x, ld, dy, dy_ld, H, W, RB, nband,
                     (int)(ld >> 2), partial);
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)cdiv64(49 * ld, 64)), dim3(256), 0, (hipStream_t)stream, partial, B * nband, 49 * ld, dw);
  return vs_launch_status();
}

extern "C" int64_t vs_colreduce_partial_floats(int B, int64_t HW, int64_t ld) { return ((int64_t)B * cdiv64(HW, CR_ROWS) + 1) * 3 * ld; }

extern "C" int vs_layernorm_bwd(const float* x, int64_t ld, const float* dy, int64_t dy_ld, const float* w, int64_t rows, int C, float eps,
                                float* dx, int64_t dx_ld, float* stats, float* partial, float* dw, float* db, void* stream) {
  VS_REQUIRE(x && dy && w && dx && stats && partial && dw && db && rows > 0 && C > 0 && ld >= C && dy_ld >= C && dx_ld >= C);
  VS_REQUIRE((ld & 3) == 0 && (dy_ld & 3) == 0 && (dx_ld & 3) == 0 && rows < (int64_t)1 << 31);
  VS_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)dy) & 15) == 0);
  hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, ld, dy, dy_ld, w, rows, C, eps, dx, dx_ld,
                     stats);
  const int C4 = (C + 3) >> 2;
  const int64_t ldp = 4 * (int64_t)C4;
  const int nch = (int)cdiv64(rows, CR_ROWS);
  VS_REQUIRE(ldp <= ld && ldp <= dy_ld);
  hipLaunchKernelGGL(colreduce_kernel<0>, dim3((unsigned)nch, (unsigned)cdiv64(C4, 256)), dim3(256), 0, (hipStream_t)stream, x, ld, dy, dy_ld, stats, (int)rows, nch, C4, ldp,
                     partial);
  // partial[chunk][2][ldp]: element (v, c) of chunk k sits at k * 2 * ldp + v * ldp + c -> one reduction over 2 * ldp values, split afterwards
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)cdiv64(2 * ldp, 64)), dim3(256), 0, (hipStream_t)stream, partial, nch, 2 * ldp,
                     partial + (int64_t)nch * 2 * ldp);
  const float* tot = partial + (int64_t)nch * 2 * ldp;
  if (hipMemcpyAsync(dw, tot, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return VS_ERR_LAUNCH;
  if (hipMemcpyAsync(db, tot + ldp, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return VS_ERR_LAUNCH;
  return vs_launch_status();
}

extern "C" int vs_gelu_grn_bwd(const float* h1, int64_t ld, const float* d3, int64_t d3_ld, const float* gamma, int B, int HW, int C, float* partial,
                               float* coef, float* dh1, int64_t dh1_ld, float* dgamma, float* dbeta, void* stream) {
  // coef: 6 * B * ldp floats (G, s, t, ca, cb, nx);  partial: vs_colreduce_partial_floats(B, HW, ldp)
  VS_REQUIRE(h1 && d3 && gamma && partial && coef && dh1 && dgamma && dbeta && B > 0 && HW > 0 && C > 0 && ld >= C && d3_ld >= C && dh1_ld >= C);
  VS_REQUIRE((ld & 3) == 0 && (d3_ld & 3) == 0 && (dh1_ld & 3) == 0);
  VS_REQUIRE((((uintptr_t)h1) & 15) == 0 && (((uintptr_t)d3) & 15) == 0 && (((uintptr_t)dh1) & 15) == 0 && (((uintptr_t)coef) & 15) == 0);
  const int C4 = (C + 3) >> 2;
  const int64_t ldp = 4 * (int64_t)C4;
  VS_REQUIRE(ldp <= ld && ldp <= d3_ld);
  const int nchf = (int)cdiv64(HW, CR_ROWS);
  float* G = coef;
  float* s = coef + (int64_t)B * ldp;
  float* t = coef + 2 * (int64_t)B * ldp;
  float* ca = coef + 3 * (int64_t)B * ldp;
  float* cb = coef + 4 * (int64_t)B * ldp;
  float* nx = coef + 5 * (int64_t)B * ldp;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colreduce_kernel<1>, dim3((unsigned)(B * nchf), (unsigned)cdiv64(C4, 256)), dim3(256), 0, st, h1, ld, d3, d3_ld, (const float*)nullptr, HW, nchf, C4, ldp,
                     partial);
  hipLaunchKernelGGL(grn_sums_kernel, dim3((unsigned)cdiv64(C, 256), (unsigned)B), dim3(256), 0, st, partial, nchf, ldp, C, G, s, t);
  hipLaunchKernelGGL(grn_coef_kernel, dim3((unsigned)B), dim3(256), 0, st, G, s, gamma, C, ldp, ca, cb, nx);
  hipLaunchKernelGGL(grn_param_grad_kernel, dim3((unsigned)cdiv64(C, 256)), dim3(256), 0, st, s, t, nx, B, C, ldp, dgamma, dbeta);
  const int O4 = (int)(dh1_ld >> 2);
  const int64_t total = (int64_t)B * HW * O4;
  // the element-wise pass walks the OUTPUT row (dh1_ld / 4 groups): groups at or beyond C are written as zeros
  hipLaunchKernelGGL(gelu_grn_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, st, h1, ld, d3, d3_ld, ca, cb, ldp, HW, C, O4, total, dh1, dh1_ld);
  return vs_launch_status();
}

extern "C" int vs_patchify_s(const float* x, int B, int H, int W, int64_t pld, int P, int S, float* cols, void* stream) {
  VS_REQUIRE(x && cols && B > 0 && P > 0 && S > 0 && S <= P && H >= P && W >= P && pld > 0 && (pld & 3) == 0 && (((uintptr_t)x) & 15) == 0 &&
             (((uintptr_t)cols) & 15) == 0);
  const int Ho = (H - P) / S + 1, Wo = (W - P) / S + 1;
  const int CP = (int)(cdiv64((int64_t)P * pld, 16) * 16);
  const int64_t total = (int64_t)B * Ho * Wo * P * (CP / 4);
  hipLaunchKernelGGL(patchify_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, H, W, pld, P, S, Ho, Wo, CP, total, cols);
  return vs_launch_status();
}
extern "C" int vs_patchify(const float* x, int B, int H, int W, int64_t pld, int P, float* cols, void* stream) {
  return vs_patchify_s(x, B, H, W, pld, P, P, cols, stream);
}

extern "C" int vs_unpatch_s(const float* dcols, int B, int H, int W, int64_t pld, int P, int S, float* dx, void* stream) {
  VS_REQUIRE(dcols && dx && B > 0 && P > 0 && S > 0 && S <= P && H >= P && W >= P && pld > 0 && (pld & 3) == 0 && (((uintptr_t)dx) & 15) == 0 &&
             (((uintptr_t)dcols) & 15) == 0);
  const int Ho = (H - P) / S + 1, Wo = (W - P) / S + 1;
  const int CP = (int)(cdiv64((int64_t)P * pld, 16) * 16);
  const int64_t total = (int64_t)B * H * W * (pld / 4);
  hipLaunchKernelGGL(unpatch_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dcols, H, W, pld, P, S, Ho, Wo, CP, total, dx);
  return vs_launch_status();
}
extern "C" int vs_unpatch(const float* dcols, int B, int H, int W, int64_t pld, int P, float* dx, void* stream) {
  return vs_unpatch_s(dcols, B, H, W, pld, P, P, dx, stream);
}

extern "C" int vs_col2im3x3_reflect(const float* dcols, int B, int H, int W, int64_t ld, float* dx, void* stream) {
  VS_REQUIRE(dcols && dx && B > 0 && H >= 2 && W >= 2 && ld > 0 && (ld & 3) == 0 && (((uintptr_t)dx) & 15) == 0 && (((uintptr_t)dcols) & 15) == 0);
  const int64_t total = (int64_t)B * H * W * (ld / 4);
  hipLaunchKernelGGL(col2im3x3_reflect_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dcols, H, W, ld, total, dx);
  return vs_launch_status();
}

extern "C" int vs_colmean(const float* x, int B, int HW, int64_t ld, float* out, void* stream) {
  VS_REQUIRE(x && out && B > 0 && HW > 0 && ld > 0 && (ld & 3) == 0 && (((uintptr_t)x) & 15) == 0);
  const int64_t total = (int64_t)B * (ld / 4);
  hipLaunchKernelGGL(colmean_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, HW, ld, (int)(ld / 4), total, out);
  return vs_launch_status();
}

extern "C" int vs_pool_gelu_bwd(const float* z, int64_t ld, const float* dpooled, int64_t dp_ld, int B, int HW, int C, float* dz, int64_t dz_ld,
                                void* stream) {
  VS_REQUIRE(z && dpooled && dz && B > 0 && HW > 0 && C > 0 && ld >= C && dp_ld >= C && dz_ld >= C && (dz_ld & 3) == 0 && (((uintptr_t)dz) & 15) == 0);
  const int C4 = (int)(dz_ld >> 2);
  const int64_t total = (int64_t)B * HW * C4;
  hipLaunchKernelGGL(pool_gelu_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, z, ld, dpooled, dp_ld, HW, C, C4, total, dz,
                     dz_ld);
  return vs_launch_status();
}

extern "C" int vs_matmul_small(const float* A, int64_t lda, const float* Bm, int64_t ldb, int M, int N, int K, float* C, int64_t ldc, void* stream) {
  VS_REQUIRE(A && Bm && C && M > 0 && N > 0 && K > 0 && lda >= K && ldb >= N && ldc >= N);
  hipLaunchKernelGGL(matmul_small_kernel, dim3(blocks_for((int64_t)M * N)), dim3(256), 0, (hipStream_t)stream, A, lda, Bm, ldb, M, N, K, C, ldc);
  return vs_launch_status();
}

extern "C" int vs_bce_logits(const float* preds, const int32_t* msgs, int msg_rows, int B, int k, float temperature, float gscale, float* dpreds,
                             float* loss, void* stream) {
  VS_REQUIRE(preds && msgs && dpreds && loss && B > 0 && k > 0 && temperature > 0.f && (msg_rows == 1 || msg_rows == B));
  hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, preds, msgs, msg_rows, B, k, 1.0f / temperature, gscale, dpreds,
                     loss);
  return vs_launch_status();
}
