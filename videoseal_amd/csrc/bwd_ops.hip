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

constexpr int CR_ROWS = 256;      // rows per chunk of the column reductions

// ---------------------------------------------------------------------------------------------------
// sum over chunks: out[i] = sum_k partial[k][i]  (fp64, fixed order);  n elements per chunk
__global__ __launch_bounds__(256) void reduce_chunks_kernel(const float* __restrict__ partial, int nchunk, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0;
  for (int k = 0; k < nchunk; ++k) s += (double)partial[(int64_t)k * n + i];
  out[i] = (float)s;
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
// and the reduction index here is the ROW: every thread therefore loads one column of 8 consecutive rows (4-byte loads, coalesced across the
// wave's 64 consecutive columns), splits the 8 values and writes them as ONE 16-byte LDS store per plane into the transposed tile
// [plane][column][row] -- the fragment reads are then plain 16-byte reads.  Row pitch 80 bytes (32 rows + 8 pad): conflict-free b128 reads.
// Workgroup = 128 (n) x 128 (k) outputs, 4 waves of 64 x 64 (2 x 2 MFMA tiles), 32 rows per step; the next step's rows are fetched into
// registers while the matrix cores work.
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

__global__ __launch_bounds__(256) void gemm_wgrad_bf16x3_kernel(const float* __restrict__ dy, int64_t dy_ld, int N, const float* __restrict__ x,
                                                                int64_t x_ld, int K, int64_t rows, int64_t rows_per_split,
                                                                float* __restrict__ partial) {
  using namespace wg16;
  __shared__ __attribute__((aligned(16))) unsigned short sa[3][TILE][PITCH], sb[3][TILE][PITCH];
  const int n0 = blockIdx.y * TILE, k0 = blockIdx.x * TILE;
  const int64_t r_begin = (int64_t)blockIdx.z * rows_per_split;
  const int64_t r_end = r_begin + rows_per_split < rows ? r_begin + rows_per_split : rows;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wy = wave >> 1, wx = wave & 1;
  const int col = threadIdx.x & 127, rg = threadIdx.x >> 7;          // staging: column `col`, rows rg * 16 .. + 16 of the step
  const int kq = lane >> 5, c = lane & 31;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float va[16], vb[16];
  const bool a_ok = n0 + col < N, b_ok = k0 + col < K;
  const float* pa = dy + n0 + col;
  const float* pb = x + k0 + col;
  auto fetch = [&](const int64_t r0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int64_t r = r0 + rg * 16 + j;
      const bool in = r < r_end;
      va[j] = (in && a_ok) ? pa[r * dy_ld] : 0.f;
      vb[j] = (in && b_ok) ? pb[r * x_ld] : 0.f;
    }
  };
  fetch(r_begin);
  for (int64_t r0 = r_begin; r0 < r_end; r0 += RS) {
    __syncthreads();                       // the previous step's fragment reads are done
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float t[8];
      u32x4 pl[3];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = va[half * 8 + j];
      split8(t, pl);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(&sa[p][col][rg * 16 + half * 8]) = pl[p];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = vb[half * 8 + j];
      split8(t, pl);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(&sb[p][col][rg * 16 + half * 8]) = pl[p];
    }
    __syncthreads();
    if (r0 + RS < r_end) fetch(r0 + RS);
#pragma unroll
    for (int ks = 0; ks < RS / 16; ++ks) {
      bf16x8 fa[2][3], fb[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          fa[i][p] = *reinterpret_cast<const bf16x8*>(&sa[p][wy * 64 + i * 32 + c][ks * 16 + kq * 8]);
          fb[i][p] = *reinterpret_cast<const bf16x8*>(&sb[p][wx * 64 + i * 32 + c][ks * 16 + kq * 8]);
        }
#pragma unroll
      for (int q = 0; q < 6; ++q)           // smallest partial products first
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = vsconv::Arith<3>::mfma(fa[i][vsconv::Arith<3>::PA[q]], fb[j][vsconv::Arith<3>::PB[q]], acc[i][j]);
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

// d w[t][c] = sum_{b,y,x} dy[b,y,x,c] * x[b, y+ky-3, x+kx-3, c]: chunk = (frame, band of RB image rows), thread items = (tap, 4 channels);
// partial[chunk][49][ld] in fp32 (RB * W terms each), summed over the chunks by reduce_chunks_kernel
__global__ __launch_bounds__(256) void dwconv7_wgrad_kernel(const float* __restrict__ x, int64_t ld, const float* __restrict__ dy,
                                                            int64_t dy_ld, int H, int W, int RB, int nband, int C4,
                                                            float* __restrict__ partial) {
  const int b = blockIdx.x / nband, band = blockIdx.x % nband;
  const int ya = band * RB, yb = ya + RB < H ? ya + RB : H;
  const float* xb = x + (int64_t)b * H * W * ld;
  const float* db = dy + (int64_t)b * H * W * dy_ld;
  for (int it = threadIdx.x; it < 49 * C4; it += 256) {
    const int t = it / C4, cg = it - t * C4;
    const int ky = t / 7, kx = t - ky * 7;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int y = ya; y < yb; ++y) {
      const int iy = y + ky - 3;
      if (iy < 0 || iy >= H) continue;
      const int xa = kx < 3 ? 3 - kx : 0, xe = W + 3 - kx < W ? W + 3 - kx : W;       // 0 <= xx + kx - 3 < W
      for (int xx = xa; xx < xe; ++xx) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(db + ((int64_t)y * W + xx) * dy_ld + 4 * cg);
        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + ((int64_t)iy * W + xx + kx - 3) * ld + 4 * cg);
        acc += g * v;
      }
    }
    *reinterpret_cast<f32x4*>(partial + ((int64_t)blockIdx.x * 49 + t) * ld + 4 * cg) = acc;
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
  const int G = C4 < 256 ? C4 : 256;                   // channel groups handled per sweep
  const int RL = 256 / G;                              // row lanes per channel group
  const int g = threadIdx.x % G, rl = threadIdx.x / G;
  for (int gb = 0; gb < C4; gb += G) {
    const int gg = gb + g;
    f32x4 acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (rl < RL && gg < C4)
      for (int64_t r = r0 + rl; r < r1; r += RL) {
        const f32x4 va = *reinterpret_cast<const f32x4*>(a + r * a_ld + 4 * gg);
        const f32x4 vb = *reinterpret_cast<const f32x4*>(b + r * b_ld + 4 * gg);
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
    __syncthreads();
  }
}

// GRN, per (frame, channel): sums over the chunks of the frame -> G = ||gelu(h1)||_2, s = sum d3 * gelu(h1), t = sum d3
__global__ __launch_bounds__(256) void grn_sums_kernel(const float* __restrict__ partial, int nchf, int64_t ldp, int C, float* __restrict__ G,
                                                       float* __restrict__ s, float* __restrict__ t) {
  const int frame = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double g2 = 0, ss = 0, tt = 0;
  for (int k = 0; k < nchf; ++k) {
    const float* p = partial + ((int64_t)frame * nchf + k) * 3 * ldp;
    g2 += (double)p[c];
    ss += (double)p[ldp + c];
    tt += (double)p[2 * ldp + c];
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
extern "C" int64_t vs_gemm_wgrad_partial_floats(int64_t rows, int N, int K) {
  if (rows <= 0 || N <= 0 || K <= 0) return 0;
  const int64_t tiles = cdiv64(N, 64) * cdiv64(K, 64);
  int64_t splits = cdiv64(1024, tiles);
  const int64_t maxs = cdiv64(rows, 256);
  if (splits > maxs) splits = maxs;
  if (splits < 1) splits = 1;
  return splits * (int64_t)N * K;
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
  static const int mode = [] { const char* e = getenv("VS_WGRAD"); return !e ? 0 : (!strcmp(e, "mfma") ? 1 : (!strcmp(e, "fma") ? 2 : 0)); }();
  // default: the 3 x bf16 matrix-core kernel where a 128 x 128 tile is at least half full, the fp32 vector kernel for the thin layers
  // (VS_WGRAD=fma / mfma force the two fp32 kernels everywhere)
  if (mode == 0 && N >= 64 && K >= 64)
    hipLaunchKernelGGL(gemm_wgrad_bf16x3_kernel, dim3((unsigned)cdiv64(K, 128), (unsigned)cdiv64(N, 128), (unsigned)used), dim3(256), 0,
                       (hipStream_t)stream, dy, dy_ld, N, x, x_ld, K, rows, rps, partial);
  else if (mode == 1)
    hipLaunchKernelGGL(gemm_wgrad_mfma_kernel, dim3((unsigned)cdiv64(K, 128), (unsigned)cdiv64(N, 128), (unsigned)used), dim3(256), 0,
                       (hipStream_t)stream, dy, dy_ld, N, x, x_ld, K, rows, rps, partial);
  else
    hipLaunchKernelGGL(gemm_wgrad_kernel, dim3((unsigned)cdiv64(K, 64), (unsigned)cdiv64(N, 64), (unsigned)used), dim3(256), 0, (hipStream_t)stream,
                       dy, dy_ld, N, x, x_ld, K, rows, rps, partial);
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3(blocks_for((int64_t)N * K)), dim3(256), 0, (hipStream_t)stream, partial, (int)used,
                     (int64_t)N * K, dw);
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

extern "C" int64_t vs_dwconv7_wgrad_partial_floats(int B, int H, int64_t ld) { return (int64_t)B * cdiv64(H, 4) * 49 * ld; }

extern "C" int vs_dwconv7_wgrad(const float* x, int64_t ld, const float* dy, int64_t dy_ld, int B, int H, int W, int C, float* partial,
                                float* dw, void* stream) {
  VS_REQUIRE(x && dy && partial && dw && B > 0 && H > 0 && W > 0 && C > 0 && ld >= C && (ld & 3) == 0 && dy_ld >= ld && (dy_ld & 3) == 0);
  VS_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)dy) & 15) == 0);
  const int RB = 4, nband = (int)cdiv64(H, RB);
  hipLaunchKernelGGL(dwconv7_wgrad_kernel, dim3((unsigned)(B * nband)), dim3(256), 0, (hipStream_t)stream, x, ld, dy, dy_ld, H, W, RB, nband,
                     (int)(ld >> 2), partial);
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3(blocks_for(49 * ld)), dim3(256), 0, (hipStream_t)stream, partial, B * nband, 49 * ld, dw);
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
  hipLaunchKernelGGL(colreduce_kernel<0>, dim3((unsigned)nch), dim3(256), 0, (hipStream_t)stream, x, ld, dy, dy_ld, stats, (int)rows, nch, C4, ldp,
                     partial);
  // partial[chunk][2][ldp]: element (v, c) of chunk k sits at k * 2 * ldp + v * ldp + c -> one reduction over 2 * ldp values, split afterwards
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3(blocks_for(2 * ldp)), dim3(256), 0, (hipStream_t)stream, partial, nch, 2 * ldp,
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
  hipLaunchKernelGGL(colreduce_kernel<1>, dim3((unsigned)(B * nchf)), dim3(256), 0, st, h1, ld, d3, d3_ld, (const float*)nullptr, HW, nchf, C4, ldp,
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
