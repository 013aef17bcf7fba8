// Bandwidth-side network kernels of the VideoSeal path (everything that is not a dense conv/GEMM):
// LayerNorm(+act), depthwise 7x7 fused with LayerNorm, GRN statistics, bilinear x2 of the skip concat,
// message latent + broadcast, the final 1x1+tanh, and the pooled linear head.
// All activations are NHWC fp32 with a channel stride `ld` (multiple of 4, pad lanes kept at zero).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "conv_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// LayerNorm over channels, one 64-lane wave per row.  mean / biased variance / (x-u)/sqrt(s+eps)*w+b
// (common.py:147-155; F.layer_norm for the channels_last flavour computes the same quantity).
__global__ __launch_bounds__(256) void layernorm_act_kernel(const float* __restrict__ x, int64_t rows, int C, int64_t ld,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            float eps, int act, float* __restrict__ out, int64_t out_ld) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ld;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  const float mean = wave_sum(s) / (float)C;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float dlt = xr[c] - mean;
    v += dlt * dlt;
  }
  const float var = wave_sum(v) / (float)C;
  const float den = sqrtf(var + eps);
  float* orow = out + row * out_ld;
  for (int c = lane; c < (int)out_ld; c += 64) {
    float y = 0.f;
    if (c < C) y = vs_apply_act(w[c] * ((xr[c] - mean) / den) + b[c], act);
    orow[c] = y;
  }
}

// Round 6: LPP lanes per row, the row in registers as 16-byte vectors (<= VMAX float4 per lane), 256 / LPP rows per workgroup.  The wave-per-row
// kernel above reads a 96-channel row with 1.5 scalar passes per phase, three phases and two 6-step butterflies per row -- one 384-byte row per wave
// at a time: 52 us for the 50 MB stem / first down-sampler tensors (1.9 TB/s in + out).  Here eight lanes share a 96-channel row (three float4 each,
// 128 contiguous bytes per row and load instruction), the two reductions are log2(LPP) xor steps, and a wave normalises eight rows at once.  Same
// expression per element (w * ((x - mean) / den) + b); the sums of the mean and the variance run over the row in a different order: fp32 rounding.
// pW > 0 (vs_layernorm_patch2x2, round 6): the rows are the pixels of B frames of pH x pW, and pixel (y, x) is stored as tap (y & 1) * 2 + (x & 1) of output
// row (y / 2, x / 2) -- [B][pH / 2][pW / 2][4 C] dense: the 2 x 2 / stride-2 down-sampling conv behind this LayerNorm (convnext.py:109-117) is then a
// plain GEMM over rows of K = 4 C (same (ky, kx, c) order as its packed weights), which the wave-specialised kernel runs instead of the generic one.
template <int LPP, int VMAX>
__global__ __launch_bounds__(256) void layernorm_act_lanes_kernel(const float* __restrict__ x, int64_t rows, int C, int64_t ld,
                                                                  const float* __restrict__ w, const float* __restrict__ b, float eps, int act,
                                                                  float* __restrict__ out, int64_t out_ld, int pH, int pW) {
  constexpr int RPB = 256 / LPP;
  const int q = threadIdx.x & (LPP - 1);
  const int64_t row = (int64_t)blockIdx.x * RPB + (threadIdx.x / LPP);
  bool live = row < rows;
  const int C4 = (C + 3) >> 2, O4 = (int)(out_ld >> 2);
  const float* xr = x + (live ? row : rows - 1) * ld;          // (rows past the end: a valid address, nothing stored -- the shuffles need every lane)
  f32x4 v[VMAX];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < VMAX; ++k) {
    const int c4 = q + k * LPP;
    v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c4 < C4) {
      v[k] = *reinterpret_cast<const f32x4*>(xr + 4 * c4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * c4 + e < C) s += v[k][e];
    }
  }
#pragma unroll
  for (int o = LPP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)C;
  float qv = 0.f;
#pragma unroll
  for (int k = 0; k < VMAX; ++k) {
    const int c4 = q + k * LPP;
    if (c4 < C4) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * c4 + e < C) { const float dlt = v[k][e] - mean; qv += dlt * dlt; }
    }
  }
#pragma unroll
  for (int o = LPP / 2; o > 0; o >>= 1) qv += __shfl_xor(qv, o, 64);
  const float den = sqrtf(qv / (float)C + eps);
  float* orow = out + row * out_ld;
  if (pW > 0 && live) {
    const int px = (int)(row % pW);
    const int64_t t = row / pW;
    const int py = (int)(t % pH);
    const int64_t fb = t / pH;
    const int oh = pH >> 1, ow = pW >> 1;
    live = (py >> 1) < oh && (px >> 1) < ow;             // odd maps: the last row / column has no output pixel (floor, as the stride-2 conv)
    orow = out + ((fb * oh + (py >> 1)) * ow + (px >> 1)) * (4 * (int64_t)C) + ((py & 1) * 2 + (px & 1)) * C;
  }
  if (!live) return;
#pragma unroll
  for (int k = 0; k < VMAX; ++k) {
    const int c4 = q + k * LPP;
    if (c4 < O4) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (c4 < C4) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + 4 * c4), bv = *reinterpret_cast<const f32x4*>(b + 4 * c4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * c4 + e < C) o[e] = vs_apply_act(wv[e] * ((v[k][e] - mean) / den) + bv[e], act);
      }
      *reinterpret_cast<f32x4*>(orow + 4 * c4) = o;
    }
  }
}

// Small-C flavour (C <= 64, the U-Net up-blocks at 128^2 / 256^2): one thread per row, the row lives in registers
// (float4 loads), so a 16-channel row does not waste 48 of the 64 lanes of a wave.
template <int C4MAX>
__global__ __launch_bounds__(256) void layernorm_act_small_kernel(const float* __restrict__ x, int64_t rows, int C, int64_t ld,
                                                                  const float* __restrict__ w, const float* __restrict__ b,
                                                                  float eps, int act, float* __restrict__ out, int64_t out_ld) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const int C4 = (C + 3) >> 2;
  f32x4 v[C4MAX];
  const float* xr = x + row * ld;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < C4MAX; ++i)
    if (i < C4) {
      v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * i + e < C) s += v[i][e];
    }
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < C4MAX; ++i)
    if (i < C4) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * i + e < C) { const float dlt = v[i][e] - mean; q += dlt * dlt; }
    }
  const float den = sqrtf(q / (float)C + eps);
  float* orow = out + row * out_ld;
#pragma unroll
  for (int i = 0; i < C4MAX; ++i)
    if (i < C4) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * i + e;
        o[e] = c < C ? vs_apply_act(w[c] * ((v[i][e] - mean) / den) + b[c], act) : 0.f;
      }
      *reinterpret_cast<f32x4*>(orow + 4 * i) = o;
    }
}

// LayerNorm over the C channels of pixels whose conv results sit in LDS rows of `stride` floats (stride/4 odd: conflict-free
// 16-byte reads).  Four adjacent lanes share a pixel (interleaved float4 columns) and combine with two quad shuffles: a wave
// normalises 16 pixels at a time instead of one pixel per wave with two 6-step butterflies (which left the one-row kernel
// latency-bound: 2 x 6 dependent cross-lane steps per pixel, one pixel in flight per wave).
// Planes output (pl.cstride != 0): the rows go out as the f16 operand planes of the all-DMA GEMM (gemm_pl.hip) instead of fp32 --
// [plane][c / 16][row][16] of value * a_mul; rowptr then returns planes + row * 32 bytes (out_ld = 8 floats) and the channel range is
// padded with zeros to a multiple of 16.
struct PlanesOut { int64_t cstride, pstride; float a_mul; int Cp; };      // bytes between 16-channel chunks / between the two planes; channels incl. padding
template <int NT, int LPP, bool WLDS, typename RowPtr>
__device__ __forceinline__ void ln_rows_from_lds_impl(const float* __restrict__ s, int stride, int npx, int C, const float* __restrict__ lnw,
                                                      const float* __restrict__ lnb, float eps, int out_ld, RowPtr rowptr, const PlanesOut pl,
                                                      float* s_wb, const bool prefilled = false) {
  // WLDS: the affine parameters are copied to LDS once per workgroup (one coalesced round trip) -- read per channel group from global memory they
  // are a chain of C / (4 LPP) dependent L2 round trips in every workgroup's tail (`LD LD s_waitcnt vmcnt(0)` per group in the ISA)
  if (WLDS && !prefilled) {            // (prefilled: the caller copied them at its start, in front of a barrier of its own)
    for (int i = threadIdx.x; i < C; i += NT) { s_wb[i] = lnw[i]; s_wb[C + i] = lnb[i]; }
    __syncthreads();
  }
  const int q = threadIdx.x & (LPP - 1);
  const int C4 = (C + 3) >> 2, O4 = pl.cstride ? pl.Cp >> 2 : out_ld >> 2;
  const float invC = 1.0f / (float)C;
  // Register-resident form (round 5): a lane's share of a pixel -- up to LN_MAXV float4 -- is read from LDS ONCE, all reads issued before the first
  // use, and stays in registers for the three passes (sum, squared deviations, output).  The loop form below reads it three times with a
  // dependent LDS round trip per channel group and the affine parameters as eight scalar LDS reads per group: this stage was 40-50 % of the
  // depthwise kernels (measured by leaving it out, LAB_NOTEBOOK round 5).  Same operations in the same order (which form runs depends on C and
  // the lanes per pixel only, i.e. on the layer, never on the batch).
  constexpr int LN_MAXV = 12;
  if (WLDS && (C & 3) == 0 && O4 <= LN_MAXV * LPP) {
    for (int p = threadIdx.x / LPP; p < npx; p += NT / LPP) {
      float* orow = rowptr(p);
      if (!orow) continue;                                  // the LPP lanes of a pixel skip together
      const float* r = s + (int64_t)p * stride;
      f32x4 v[LN_MAXV];
#pragma unroll
      for (int k = 0; k < LN_MAXV; ++k) {
        const int c4 = q + k * LPP;
        v[k] = *reinterpret_cast<const f32x4*>(r + 4 * (c4 < C4 ? c4 : 0));       // (clamped: unconditional reads; unused slots are never added)
      }
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < LN_MAXV; ++k)
        if (q + k * LPP < C4) sum += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
#pragma unroll
      for (int o = 1; o < LPP; o <<= 1) sum += __shfl_xor(sum, o, 64);
      const float mean = sum * invC;
      float var = 0.f;
#pragma unroll
      for (int k = 0; k < LN_MAXV; ++k)
        if (q + k * LPP < C4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dl = v[k][e] - mean;               // C % 4 == 0: every channel of the group is a real one
            var += dl * dl;
          }
        }
#pragma unroll
      for (int o = 1; o < LPP; o <<= 1) var += __shfl_xor(var, o, 64);
      const float rden = 1.0f / sqrtf(var * invC + eps);
#pragma unroll
      for (int k = 0; k < LN_MAXV; ++k) {
        const int c4 = q + k * LPP;
        if (c4 < O4) {
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
          if (c4 < C4) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(s_wb + 4 * c4), bv = *reinterpret_cast<const f32x4*>(s_wb + C + 4 * c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = wv[e] * ((v[k][e] - mean) * rden) + bv[e];
          }
          if (pl.cstride) {
            vsconv::u32x2 hi, lo;
            vsconv::split4h(o, pl.a_mul, hi, lo);
            char* dst = reinterpret_cast<char*>(orow) + (int64_t)(c4 >> 2) * pl.cstride + (c4 & 3) * 8;
            *reinterpret_cast<vsconv::u32x2*>(dst) = hi;
            *reinterpret_cast<vsconv::u32x2*>(dst + pl.pstride) = lo;
          } else {
            *reinterpret_cast<f32x4*>(orow + 4 * c4) = o;
          }
        }
      }
    }
    return;
  }
  for (int p = threadIdx.x / LPP; p < npx; p += NT / LPP) {
    float* orow = rowptr(p);
    if (!orow) continue;                                  // the LPP lanes of a pixel skip together
    const float* r = s + (int64_t)p * stride;
    float sum = 0.f;
    for (int c4 = q; c4 < C4; c4 += LPP) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(r + 4 * c4);
      sum += (v[0] + v[1]) + (v[2] + v[3]);               // channels >= C hold exact zeros (zero taps, zero bias)
    }
#pragma unroll
    for (int o = 1; o < LPP; o <<= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum * invC;
    float var = 0.f;
    for (int c4 = q; c4 < C4; c4 += LPP) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(r + 4 * c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dl = (4 * c4 + e < C) ? v[e] - mean : 0.f;
        var += dl * dl;
      }
    }
#pragma unroll
    for (int o = 1; o < LPP; o <<= 1) var += __shfl_xor(var, o, 64);
    const float rden = 1.0f / sqrtf(var * invC + eps);
    for (int c4 = q; c4 < O4; c4 += LPP) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (c4 < C4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(r + 4 * c4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * c4 + e < C) o[e] = (WLDS ? s_wb[4 * c4 + e] : lnw[4 * c4 + e]) * ((v[e] - mean) * rden) + (WLDS ? s_wb[C + 4 * c4 + e] : lnb[4 * c4 + e]);
      }
      if (pl.cstride) {
        vsconv::u32x2 hi, lo;
        vsconv::split4h(o, pl.a_mul, hi, lo);
        char* dst = reinterpret_cast<char*>(orow) + (int64_t)(c4 >> 2) * pl.cstride + (c4 & 3) * 8;
        *reinterpret_cast<vsconv::u32x2*>(dst) = hi;
        *reinterpret_cast<vsconv::u32x2*>(dst + pl.pstride) = lo;
      } else {
        *reinterpret_cast<f32x4*>(orow + 4 * c4) = o;
      }
    }
  }
}
// lanes per pixel: as few as keep the whole workgroup busy (many pixels of few channels: 4 lanes and two quad shuffles; a
// handful of pixels of many channels: up to a wave per pixel)
template <int NT, typename RowPtr>
__device__ __forceinline__ void ln_rows_from_lds(const float* __restrict__ s, int stride, int npx, int C, const float* __restrict__ lnw,
                                                 const float* __restrict__ lnb, float eps, int out_ld, RowPtr rowptr,
                                                 const PlanesOut pl = PlanesOut{0, 0, 1.f, 0}, float* s_wb = nullptr, const bool prefilled = false) {
  // s_wb: 2 C floats of LDS that no wave reads any more (the caller's barrier has passed) or that the caller filled itself (prefilled), or nullptr
  if (s_wb) {
    if (npx * 8 > NT) ln_rows_from_lds_impl<NT, 4, true>(s, stride, npx, C, lnw, lnb, eps, out_ld, rowptr, pl, s_wb, prefilled);
    else if (npx * 16 > NT) ln_rows_from_lds_impl<NT, 8, true>(s, stride, npx, C, lnw, lnb, eps, out_ld, rowptr, pl, s_wb, prefilled);
    else if (npx * 32 > NT) ln_rows_from_lds_impl<NT, 16, true>(s, stride, npx, C, lnw, lnb, eps, out_ld, rowptr, pl, s_wb, prefilled);
    else if (npx * 64 > NT) ln_rows_from_lds_impl<NT, 32, true>(s, stride, npx, C, lnw, lnb, eps, out_ld, rowptr, pl, s_wb, prefilled);
    else ln_rows_from_lds_impl<NT, 64, true>(s, stride, npx, C, lnw, lnb, eps, out_ld, rowptr, pl, s_wb, prefilled);
    return;
  }
  if (npx * 8 > NT) ln_rows_from_lds_impl<NT, 4, false>(s, stride, npx, C, lnw, lnb, eps, out_ld, rowptr, pl, nullptr);
  else if (npx * 16 > NT) ln_rows_from_lds_impl<NT, 8, false>(s, stride, npx, C, lnw, lnb, eps, out_ld, rowptr, pl, nullptr);
  else if (npx * 32 > NT) ln_rows_from_lds_impl<NT, 16, false>(s, stride, npx, C, lnw, lnb, eps, out_ld, rowptr, pl, nullptr);
  else if (npx * 64 > NT) ln_rows_from_lds_impl<NT, 32, false>(s, stride, npx, C, lnw, lnb, eps, out_ld, rowptr, pl, nullptr);
  else ln_rows_from_lds_impl<NT, 64, false>(s, stride, npx, C, lnw, lnb, eps, out_ld, rowptr, pl, nullptr);
}

// ---------------------------------------------------------------------------------------------------
// Depthwise 7x7 (pad 3) + bias, then LayerNorm over C, per pixel (convnext.py:43-46).
// A work item = (strip of 4 consecutive x, group of 4 channels): 70 float4 loads feed 16 outputs x 4 channels,
// results go to LDS [pixel][channel]; then one wave per pixel does the LayerNorm and the coalesced store.
// R = output rows per strip (round 6: R = 2 for the wide small maps of ChunkySeal -- 31 x 31 x 1448 -- where one strip of all channels already fills a
// workgroup: a 4 x 2 strip reads 8 input rows x 10 columns for 8 outputs instead of 7 x 10 for 4, i.e. 10 instead of 17.5 fetches per output through
// L1 / L2, which is what bounds this kernel; per output the taps are still accumulated in (ky, kx) order: bit-identical to R = 1 and to the tiled kernels)
template <int R>
__global__ __launch_bounds__(256) void dwconv7_ln_kernel(const float* __restrict__ x, int B, int H, int W, int C, int64_t ld,
                                                         const float* __restrict__ wdw, const float* __restrict__ bdw,
                                                         const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                         float eps, float* __restrict__ out, int64_t out_ld, int NS,
                                                         int spr, int64_t nstrips, PlanesOut pl) {
  extern __shared__ __attribute__((aligned(16))) float conv[];   // [NS*4*R][ld + 4] | the LayerNorm's affine parameters [2][C]
  float* const s_wb = conv + (int64_t)NS * 4 * R * (ld + 4);
  for (int i = threadIdx.x; i < C; i += 256) { s_wb[i] = lnw[i]; s_wb[C + i] = lnb[i]; }      // (visible behind the barrier below)
  const int C4 = (int)(ld >> 2);
  const int HR = (H + R - 1) / R;                  // strip rows per frame
  const int64_t s0 = (int64_t)blockIdx.x * NS;
  const int items = NS * C4;
  for (int it = threadIdx.x; it < items; it += blockDim.x) {
    const int sl = it / C4, cg = it - sl * C4;
    const int64_t sidx = s0 + sl;
    if (sidx >= nstrips) break;
    const int c = cg * 4;
    const int xs = (int)(sidx % spr);
    const int64_t t = sidx / spr;
    const int y = (int)(t % HR) * R;
    const int b = (int)(t / HR);
    const int x0 = xs * 4;
    f32x4 acc[R][4];
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bdw + c);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int p = 0; p < 4; ++p) acc[r][p] = bv;
    const float* base = x + (int64_t)b * H * W * ld + c;
    for (int kr = 0; kr < 6 + R; ++kr) {           // input row y - 3 + kr feeds output row r with ky = kr - r
      const int iy = y + kr - 3;
      if (iy < 0 || iy >= H) continue;
      f32x4 in[10];
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const int ix = x0 - 3 + j;
        in[j] = (ix >= 0 && ix < W) ? *reinterpret_cast<const f32x4*>(base + ((int64_t)iy * W + ix) * ld)
                                    : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int ky = kr - r;
        if (ky < 0 || ky > 6) continue;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wdw + (int64_t)(ky * 7 + kx) * ld + c);
#pragma unroll
          for (int p = 0; p < 4; ++p) acc[r][p] += in[p + kx] * wv;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(conv + (int64_t)((sl * R + r) * 4 + p) * (ld + 4) + c) = acc[r][p];
  }
  __syncthreads();
  ln_rows_from_lds<256>(conv, (int)ld + 4, NS * 4 * R, C, lnw, lnb, eps, (int)out_ld, [&](int p) -> float* {
    const int64_t sidx = s0 + p / (4 * R);
    if (sidx >= nstrips) return nullptr;
    const int r = (p / 4) % R;
    const int px = (int)(sidx % spr) * 4 + (p & 3);
    const int64_t t = sidx / spr;
    const int y = (int)(t % HR) * R + r;
    const int64_t b = t / HR;
    return (px < W && y < H) ? out + ((b * H + y) * W + px) * out_ld : nullptr;
  }, pl, s_wb, true);
}

// LDS-tiled flavour of the same op (same FMA order per output => bit-identical to dwconv7_ln_kernel): a workgroup owns a TH x TW
// pixel tile of one frame and ALL channels.  Channels are walked in chunks of CCH: the zero-padded (TH+6) x (TW+6) halo tile of the
// chunk and its 49 taps + bias are staged in LDS once (coalesced 16-byte loads, no per-tap bounds checks or address arithmetic in
// the inner loop), a work item = 4 consecutive pixels x 4 channels reads 7 x 10 float4 from LDS for 784 FMAs, and the conv results
// are parked in an LDS [pixel][channel] buffer until the per-pixel LayerNorm (one wave per pixel, as in the one-row kernel).
// The one-row kernel re-read every input 30x through L1 with 4.5 VALU instructions per FMA (rocprofv3).
template <int TH, int TW, int CCH, int NT>
__global__ __launch_bounds__(NT) void dwconv7_ln_tiled_kernel(const float* __restrict__ x, int H, int W, int C, int64_t ld,
                                                              const float* __restrict__ wdw, const float* __restrict__ bdw,
                                                              const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                              float eps, float* __restrict__ out, int64_t out_ld, int tiles_x,
                                                              int tiles_y, int nblk, PlanesOut pl, int wb_sep) {
  constexpr int IH = TH + 6, IW = TW + 6, CP = CCH + 4, SPR = TW / 4, N4 = CCH / 4;
  constexpr int NIN = IH * IW * N4, NWT = 50 * N4;            // float4 slots of a chunk: halo tile, taps + bias
  constexpr int LIN = (NIN + NT - 1) / NT, LWT = (NWT + NT - 1) / NT;
  constexpr int ITEMS = TH * SPR * N4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const s_in = smem;                       // [IH][IW][CP]
  float* const s_w = s_in + IH * IW * CP;         // [50][CCH]: 49 taps + bias
  float* const s_out = s_w + 50 * CCH;            // [TH*TW][ld + 4]
  float* const s_wbs = s_out + (int64_t)TH * TW * (ld + 4);      // wb_sep: the LayerNorm's affine parameters [2][C], copied right here
  const int tid = threadIdx.x;
  const int per = (nblk + 7) >> 3;                // XCD-aware order: neighbouring tiles (shared halos) on one XCD's L2
  const int vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (vb >= nblk) return;
  if (wb_sep)                                     // (visible behind the chunk loop's barriers; at the tail this was a global round trip + a barrier)
    for (int i = tid; i < C; i += NT) { s_wbs[i] = lnw[i]; s_wbs[C + i] = lnb[i]; }
  const int tx = vb % tiles_x;
  const int t1 = vb / tiles_x;
  const int ty = t1 % tiles_y;
  const int b = t1 / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const float* xb = x + (int64_t)b * H * W * ld;
  // per-thread slots of the halo tile are the same for every chunk: global offset (or -1 outside the image) + LDS offset
  int g_off[LIN], l_off[LIN];
#pragma unroll
  for (int k = 0; k < LIN; ++k) {
    const int i = tid + k * NT;
    const int cg = i % N4, pp = i / N4;
    const int px = pp % IW, py = pp / IW;
    const int gy = y0 + py - 3, gx = x0 + px - 3;
    const bool ok = i < NIN && gy >= 0 && gy < H && gx >= 0 && gx < W;
    g_off[k] = ok ? (int)(((int64_t)gy * W + gx) * ld) + cg * 4 : -1;
    l_off[k] = i < NIN ? (py * IW + px) * CP + cg * 4 : -1;
  }
  f32x4 rin[LIN], rwt[LWT];
  auto fetch = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < LIN; ++k) {
      rin[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (g_off[k] >= 0) rin[k] = *reinterpret_cast<const f32x4*>(xb + g_off[k] + c0);
    }
#pragma unroll
    for (int k = 0; k < LWT; ++k) {
      const int i = tid + k * NT;
      const int cg = i % N4, t = i / N4;
      if (i < NWT) rwt[k] = *reinterpret_cast<const f32x4*>((t < 49 ? wdw + (int64_t)t * ld : bdw) + c0 + cg * 4);
    }
  };
  const int nch = (int)ld / CCH;                   // the launcher guarantees ld % CCH == 0
  fetch(0);
  for (int ch = 0; ch < nch; ++ch) {
    const int c0 = ch * CCH;
#pragma unroll
    for (int k = 0; k < LIN; ++k)
      if (l_off[k] >= 0) *reinterpret_cast<f32x4*>(s_in + l_off[k]) = rin[k];
#pragma unroll
    for (int k = 0; k < LWT; ++k) {
      const int i = tid + k * NT;
      if (i < NWT) *reinterpret_cast<f32x4*>(s_w + (i / N4) * CCH + (i % N4) * 4) = rwt[k];
    }
    __syncthreads();
    if (ch + 1 < nch) fetch(c0 + CCH);             // next chunk's global loads fly during this chunk's FMAs
#ifdef VS_DWT_ABL
    if (!(VS_DWT_ABL & 2))          // timing ablation: no filter arithmetic
#endif
    for (int it = tid; it < ITEMS; it += NT) {
      const int cg = it % N4, rs = it / N4;
      const int sx = rs % SPR, r = rs / SPR;
      f32x4 acc[4];
      const f32x4 bv = *reinterpret_cast<const f32x4*>(s_w + 49 * CCH + cg * 4);
#pragma unroll
      for (int p = 0; p < 4; ++p) acc[p] = bv;
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const float* row = s_in + ((r + ky) * IW + sx * 4) * CP + cg * 4;
        f32x4 in[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) in[j] = *reinterpret_cast<const f32x4*>(row + j * CP);
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(s_w + (ky * 7 + kx) * CCH + cg * 4);
#pragma unroll
          for (int p = 0; p < 4; ++p) acc[p] += in[p + kx] * wv;
        }
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(s_out + (int64_t)(r * TW + sx * 4 + p) * (ld + 4) + c0 + cg * 4) = acc[p];
    }
    __syncthreads();
  }
#ifdef VS_DWT_ABL
  if (VS_DWT_ABL & 1) return;       // timing ablation: no LayerNorm stage
#endif
  // (the halo tile s_in is dead behind the last chunk's barrier: its first 2 C floats hold the LayerNorm parameters)
  ln_rows_from_lds<NT>(s_out, (int)ld + 4, TH * TW, C, lnw, lnb, eps, (int)out_ld, [&](int p) -> float* {
    const int gy = y0 + p / TW, gx = x0 + p % TW;
    return (gy < H && gx < W) ? out + (((int64_t)b * H + gy) * W + gx) * out_ld : nullptr;
  }, pl, wb_sep ? s_wbs : (2 * C <= (TH + 6) * (TW + 6) * CP) ? s_in : nullptr, wb_sep != 0);
}

template <int TH, int TW, int CCH, int NT>
static int launch_dwconv_tiled(const float* x, int B, int H, int W, int C, int64_t ld, const float* wdw, const float* bdw, const float* lnw,
                               const float* lnb, float eps, float* out, int64_t out_ld, hipStream_t st, PlanesOut pl = PlanesOut{0, 0, 1.f, 0}) {
  size_t smem = sizeof(float) * ((size_t)(TH + 6) * (TW + 6) * (CCH + 4) + 50 * CCH + (size_t)TH * TW * (ld + 4));
  if (smem > 160 * 1024 || ld % CCH != 0 || (int64_t)H * W * ld >= (1ll << 31)) return VS_ERR_UNSUPPORTED;
  const int wb_sep = smem + 2 * sizeof(float) * (size_t)C <= 160 * 1024;       // room for the LayerNorm parameters next to the tiles
  if (wb_sep) smem += 2 * sizeof(float) * (size_t)C;
  auto kern = dwconv7_ln_tiled_kernel<TH, TW, CCH, NT>;
  static size_t attr_set = 0;            // raise the dynamic-LDS limit once per size class (a cheap host-side call otherwise)
  if (smem > 64 * 1024 && smem > attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = 160 * 1024;
  }
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int64_t nblk = (int64_t)B * tiles_x * tiles_y;
  if (nblk >= (1 << 30)) return VS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3((unsigned)((nblk + 7) / 8 * 8)), dim3(NT), smem, st, x, H, W, C, ld, wdw, bdw, lnw, lnb, eps, out, out_ld,
                     tiles_x, tiles_y, (int)nblk, pl, wb_sep);
  return vs_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// GRN statistics (common.py:166-167), deterministic two-stage reduction.
constexpr int GRN_ROWS = 64;
__global__ __launch_bounds__(256) void grn_partial_kernel(const float* __restrict__ h, int HW, int C, int64_t ld,
                                                          float* __restrict__ partial, int B) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int chunk = blockIdx.y, b = blockIdx.z;
  if (c >= C) return;
  const int r0 = chunk * GRN_ROWS, r1 = min(HW, r0 + GRN_ROWS);
  const float* p = h + ((int64_t)b * HW + r0) * ld + c;
  float s = 0.f;
  for (int r = r0; r < r1; ++r, p += ld) s += (*p) * (*p);
  partial[((int64_t)chunk * B + b) * C + c] = s;
}
// frame_major: partial is [B][nchunk][C] (written by the GEMM epilogue) instead of [nchunk][B][C].
// One workgroup of 1024 threads per frame: 256 channels x 4 groups of chunks at a time; every group adds its chunks in
// order and the four group sums are combined in a fixed order -> deterministic, with 4x the loads in flight of a
// one-thread-per-channel loop (frame-major partials are up to HW/32 = 128 chunks deep).
// (round 6: NB x KB = 32 loads in flight per thread whatever C: with NB = 8 a 384-channel frame used two of the eight block slots and walked its
//  32 chunk rows per group four at a time -- eight dependent round trips; <2, 16> does it in two.  Same additions in the same order.)
template <int NB, int KB>
__global__ __launch_bounds__(1024) void grn_finish_kernel(const float* __restrict__ partial, int nchunk, int B, int C,
                                                          const float* __restrict__ gamma, float* __restrict__ scale,
                                                          int64_t sld, int frame_major) {
  // Channel blocks of 256 are walked NB at a time: the chunk sums of NB blocks are independent loads issued back to back, ONE barrier pair per
  // NB blocks (one pair per block made the kernel a chain of C / 256 dependent round trips: 11.6 us for 18 launches per extractor pass, whatever
  // the stage).  Same sums in the same order as before: per channel ((g0 + g1) + g2) + g3, per thread the blocks in ascending order.
  __shared__ float grp[NB][4][256];
  __shared__ float red[256];
  const int b = blockIdx.x;
  const int cl = threadIdx.x & 255, kg = threadIdx.x >> 8;
  const int kper = (nchunk + 3) / 4;
  const int k0 = kg * kper, k1 = min(nchunk, k0 + kper);
  float local = 0.f;
  for (int cb0 = 0; cb0 < C; cb0 += 256 * NB) {
    float s[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) s[j] = 0.f;
    // chunk rows four at a time: the 4 x NB loads are issued before the first addition (a loop that adds each value as it arrives is a chain of
    // k1 - k0 dependent L2 round trips: up to 32 of them for the 64 x 64 maps); additions stay in ascending chunk order
    const int64_t kstride = frame_major ? (int64_t)C : (int64_t)B * C;
    const float* rowp = frame_major ? partial + ((int64_t)b * nchunk + k0) * C : partial + ((int64_t)k0 * B + b) * C;
    int k = k0;
    for (; k + KB <= k1; k += KB, rowp += KB * kstride) {
      float v[KB][NB];
#pragma unroll
      for (int q = 0; q < KB; ++q)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int c = cb0 + j * 256 + cl;
          v[q][j] = c < C ? rowp[q * kstride + c] : 0.f;
        }
#pragma unroll
      for (int q = 0; q < KB; ++q)
#pragma unroll
        for (int j = 0; j < NB; ++j) s[j] += v[q][j];
    }
    if constexpr (KB > 4) {           // what is left of the group, four rows at a time as before
      for (; k + 4 <= k1; k += 4, rowp += 4 * kstride) {
        float v[4][NB];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const int c = cb0 + j * 256 + cl;
            v[q][j] = c < C ? rowp[q * kstride + c] : 0.f;
          }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < NB; ++j) s[j] += v[q][j];
      }
    }
    for (; k < k1; ++k, rowp += kstride) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int c = cb0 + j * 256 + cl;
        if (c < C) s[j] += rowp[c];
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) grp[j][kg][cl] = s[j];
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int c = cb0 + j * 256 + cl;
        if (c < C) {
          const float gx = sqrtf(((grp[j][0][cl] + grp[j][1][cl]) + grp[j][2][cl]) + grp[j][3][cl]);
          scale[(int64_t)b * sld + c] = gx;
          local += gx;
        }
      }
    }
    __syncthreads();
  }
  if (kg == 0) red[cl] = local;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float mean = red[0] / (float)C;
  for (int c = threadIdx.x; c < (int)sld; c += 1024) {
    float v = 0.f;
    if (c < C) v = 1.0f + gamma[c] * (scale[(int64_t)b * sld + c] / (mean + 1e-6f));
    scale[(int64_t)b * sld + c] = v;
  }
}

// ---------------------------------------------------------------------------------------------------
// Bilinear x2 (align_corners=False) of the channel concat [x | skip*scale]  (unet.py:186-187 + common.py:46).
__global__ __launch_bounds__(256) void upcat2x_kernel(const float* __restrict__ x, int C1, int64_t ld1,
                                                      const float* __restrict__ skip, int C2, int64_t ld2, float sscale,
                                                      int B, int H, int W, float* __restrict__ out, int64_t old, int64_t total) {
  const int G = (int)(old >> 2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cg = (int)(i % G);
    int64_t p = i / G;
    const int X = (int)(p % (2 * W)); p /= (2 * W);
    const int Y = (int)(p % (2 * H));
    const int b = (int)(p / (2 * H));
    const int c = cg * 4;
    float sy = fmaxf((Y + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((X + 0.5f) * 0.5f - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
    const float ly1 = sy - y0, lx1 = sx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float* src; int64_t ld; int cc; float mul;
    if (c < C1) { src = x; ld = ld1; cc = c; mul = 1.f; }
    else { src = skip; ld = ld2; cc = c - C1; mul = sscale; }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < C1 + C2) {
      const float* fb = src + (int64_t)b * H * W * ld + cc;
      const f32x4 p00 = *reinterpret_cast<const f32x4*>(fb + ((int64_t)y0 * W + x0) * ld);
      const f32x4 p01 = *reinterpret_cast<const f32x4*>(fb + ((int64_t)y0 * W + x1) * ld);
      const f32x4 p10 = *reinterpret_cast<const f32x4*>(fb + ((int64_t)y1 * W + x0) * ld);
      const f32x4 p11 = *reinterpret_cast<const f32x4*>(fb + ((int64_t)y1 * W + x1) * ld);
      v = (ly0 * (lx0 * (p00 * mul) + lx1 * (p01 * mul)) + ly1 * (lx0 * (p10 * mul) + lx1 * (p11 * mul)));
    }
    *reinterpret_cast<f32x4*>(out + (((int64_t)b * 2 * H + Y) * 2 * W + X) * old + c) = v;
  }
}

// ---------------------------------------------------------------------------------------------------
// Upsample group without the up-sampled tensor (common.py:45-52: bilinear x2 -> ReflectionPad2d(1) -> Conv3x3 (no bias) -> LayerNorm
// over C -> act).  Bilinear interpolation, the reflect padding and the tap shift are linear maps over space and commute with the
// channel mixing of the conv, so
//     conv3x3(pad(up(v)))[Y,X,c] = sum_t  up(z_t)[refl(Y+ky-1), refl(X+kx-1), c],     z_t[y,x,c] = sum_ci W[c,ci,t] v[y,x,ci]
// z (all nine taps: a plain [rows] x [9*Co] x [Cin] GEMM on the LOW-resolution map, a quarter of the conv's MACs) comes from
// vs_conv_gemm; this kernel does the 9-tap x 4-neighbour gather, the per-pixel LayerNorm and the activation.  The x2 up-sampled
// concat (384 MB per launch at 64^2 x 768 channels x 32 frames) is never materialised.
// Thread = (output pixel, group of CG channels); the TPP = Co / CG lanes of a pixel are adjacent and reduce with xor shuffles.
template <int CG>
__global__ __launch_bounds__(256) void upconv_gather_ln_kernel(const float* __restrict__ z, int64_t zld, int H, int W, int Co,
                                                               int tpp_log2, const float* __restrict__ lnw,
                                                               const float* __restrict__ lnb, float eps, int act,
                                                               float* __restrict__ out, int64_t old, int64_t npix, int nblk) {
  // XCD-aware order: workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md); give every XCD one contiguous range of rows so that
  // the low-resolution z rows shared by neighbouring output rows stay in that XCD's L2
  const int per = (nblk + 7) >> 3;
  const int64_t vb = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int64_t gid = vb * 256 + threadIdx.x;
  const int64_t pq = gid >> tpp_log2;
  const int g = (int)(gid & ((1 << tpp_log2) - 1));
  const bool live = pq < npix && vb < nblk;
  const int64_t p = live ? pq : 0;           // dead lanes still take part in the shuffles
  // LayerNorm parameters of this lane's channels: requested first (behind the gather they were a dependent L2 round trip at the tail of the kernel)
  constexpr int NVP = CG / 4;
  f32x4 lwv[NVP], lbv[NVP];
#pragma unroll
  for (int j = 0; j < NVP; ++j) {
    lwv[j] = *reinterpret_cast<const f32x4*>(lnw + g * CG + 4 * j);
    lbv[j] = *reinterpret_cast<const f32x4*>(lnb + g * CG + 4 * j);
  }
  const int W2 = 2 * W, H2 = 2 * H;
  const int X = (int)(p % W2);
  const int64_t t0 = p / W2;
  const int Y = (int)(t0 % H2);
  const int64_t b = t0 / H2;
  int ys[3][2], xs[3][2];
  float wy[3][2], wx[3][2];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int yr = Y + k - 1, xr = X + k - 1;
    yr = yr < 0 ? -yr : (yr >= H2 ? 2 * H2 - 2 - yr : yr);            // ReflectionPad2d(1) on the up-sampled grid
    xr = xr < 0 ? -xr : (xr >= W2 ? 2 * W2 - 2 - xr : xr);
    const float sy = fmaxf((yr + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((xr + 0.5f) * 0.5f - 0.5f, 0.f);   // align_corners=False
    const int y0 = (int)sy, x0 = (int)sx;
    ys[k][0] = y0; ys[k][1] = y0 + (y0 < H - 1);
    xs[k][0] = x0; xs[k][1] = x0 + (x0 < W - 1);
    wy[k][1] = sy - y0; wy[k][0] = 1.f - wy[k][1];
    wx[k][1] = sx - x0; wx[k][0] = 1.f - wx[k][1];
  }
  constexpr int NV = CG / 4;
  f32x4 acc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* zb = z + b * H * W * zld + g * CG;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const float* zr = zb + (int64_t)ys[ky][a] * W * zld + ky * 3 * Co;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const float w = wy[ky][a] * wx[kx][c];
          const float* q = zr + (int64_t)xs[kx][c] * zld + kx * Co;
#pragma unroll
          for (int j = 0; j < NV; ++j) acc[j] += w * *reinterpret_cast<const f32x4*>(q + 4 * j);
        }
    }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) s += (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]);
  for (int o = 1; o < (1 << tpp_log2); o <<= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)Co;
  float v = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float dl = acc[j][e] - mean; v += dl * dl; }
  for (int o = 1; o < (1 << tpp_log2); o <<= 1) v += __shfl_xor(v, o, 64);
  const float den = sqrtf(v / (float)Co + eps);
  if (!live) return;
  float* orow = out + p * old + g * CG;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const f32x4 wv = lwv[j], bv = lbv[j];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = vs_apply_act(wv[e] * ((acc[j][e] - mean) / den) + bv[e], act);
    *reinterpret_cast<f32x4*>(orow + 4 * j) = o;
  }
}

// 3x3 / stride 1 / pad 1 patch matrix of a SMALL map: out[m][t*ld + c] = in[b][p(y+ky-1)][p(x+kx-1)][c], p = reflect or zero padding.
// Turns the pixel decoder's 3x3 conv on the 8x8 map (pixel_decoder.py:44-48: 2048 rows at 32 frames, K = 6912) into a dense 1x1 GEMM
// that the wave-specialised split-K kernel can spread over all CUs (the implicit-GEMM conv had 96 workgroups for 256 CUs).
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ x, int H, int W, int64_t ld, int reflect,
                                                        float* __restrict__ out, int64_t total) {
  const int G = (int)(ld >> 2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cg = (int)(i % G);
    int64_t r = i / G;
    const int t = (int)(r % 9);
    r /= 9;                                   // output row m = (b, y, x)
    const int xx = (int)(r % W);
    const int yy = (int)((r / W) % H);
    const int64_t b = r / ((int64_t)W * H);
    int sy = yy + t / 3 - 1, sx = xx + t % 3 - 1;
    bool ok = true;
    if (reflect) {
      sy = sy < 0 ? -sy : (sy >= H ? 2 * H - 2 - sy : sy);
      sx = sx < 0 ? -sx : (sx >= W ? 2 * W - 2 - sx : sx);
    } else {
      ok = sy >= 0 && sy < H && sx >= 0 && sx < W;
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) v = *reinterpret_cast<const f32x4*>(x + ((b * H + sy) * W + sx) * ld + cg * 4);
    *reinterpret_cast<f32x4*>(out + (r * 9 + t) * ld + cg * 4) = v;
  }
}

// unet.py:183-185 first bottleneck conv over the spatially constant message channels: border-class table from the per-tap table P
// (see header).  class = cy*3 + cx, cy / cx = 0 first row / column (tap 0 falls outside), 1 interior, 2 last (tap 2 outside).
__global__ __launch_bounds__(256) void msg_pre_kernel(const float* __restrict__ P, int N, float* __restrict__ T, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i % N);
    const int cls = (int)((i / N) % 9);
    const int64_t bm = i / ((int64_t)9 * N);
    const int cy = cls / 3, cx = cls % 3;
    const float* pb = P + bm * 9 * N + n;
    float v = 0.f;
    for (int ky = (cy == 0 ? 1 : 0); ky <= (cy == 2 ? 1 : 2); ++ky)
      for (int kx = (cx == 0 ? 1 : 0); kx <= (cx == 2 ? 1 : 2); ++kx) v += pb[(ky * 3 + kx) * N];
    T[i] = v;
  }
}

// channel concat of two NHWC maps of the same pixels, the second one scaled: out[r] = [x[r] | skip[r] * s]  (unet.py:186-187 at the
// LOW resolution).  x may be NULL when its producer already wrote columns [0, C1) of `out`.
__global__ __launch_bounds__(256) void cat2_scale_kernel(const float* __restrict__ x, int C1, int64_t ld1, const float* __restrict__ skip,
                                                         int C2, int64_t ld2, float s, int64_t rows, float* __restrict__ out,
                                                         int64_t old) {
  const int g1 = x ? C1 >> 2 : 0, G = g1 + (C2 >> 2);
  const int64_t total = rows * G;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cg = (int)(i % G);
    const int64_t r = i / G;
    f32x4 v;
    int c;
    if (cg < g1) { c = cg * 4; v = *reinterpret_cast<const f32x4*>(x + r * ld1 + c); }
    else { const int c2 = (cg - g1) * 4; c = C1 + c2; v = *reinterpret_cast<const f32x4*>(skip + r * ld2 + c2) * s; }
    *reinterpret_cast<f32x4*>(out + r * old + c) = v;
  }
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void msg_latent_kernel(const float* __restrict__ table, const int32_t* __restrict__ msgs,
                                                         int nbits, int hidden, float* __restrict__ lat) {
  // the row index of every bit first (LDS), so that the table loads below do not depend on a load each: 16 of them are in
  // flight at a time instead of 2 x nbits serial round trips; the summation order (k ascending) is unchanged
  __shared__ int rowsel[1024];
  const int b = blockIdx.y;
  for (int k = threadIdx.x; k < nbits; k += 256) rowsel[k] = 2 * k + (msgs[(int64_t)b * nbits + k] != 0);
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= hidden) return;
  float s = 0.f;
  int k = 0;
  for (; k + 16 <= nbits; k += 16) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = table[(int64_t)rowsel[k + j] * hidden + c];
#pragma unroll
    for (int j = 0; j < 16; ++j) s += v[j];
  }
  for (; k < nbits; ++k) s += table[(int64_t)rowsel[k] * hidden + c];
  lat[(int64_t)b * hidden + c] = s;
}
// four channels per thread (hidden, ld, coff multiples of 4 and 16-byte aligned pointers: every shipped card): 16-byte stores, a quarter of the index
// arithmetic -- the one-float form ran at 1.4 TB/s of stores
__global__ __launch_bounds__(256) void broadcast_channels4_kernel(const float* __restrict__ lat, int Bm, int hidden4,
                                                                  float* __restrict__ dst, int HW, int64_t ld, int coff,
                                                                  int64_t total4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % hidden4);
    const int64_t p = i / hidden4;
    const int b = (int)(p / HW);
    *reinterpret_cast<f32x4*>(dst + p * ld + coff + 4 * c4) = *reinterpret_cast<const f32x4*>(lat + ((int64_t)(Bm == 1 ? 0 : b) * hidden4 + c4) * 4);
  }
}
__global__ __launch_bounds__(256) void broadcast_channels_kernel(const float* __restrict__ lat, int Bm, int hidden,
                                                                 float* __restrict__ dst, int HW, int64_t ld, int coff,
                                                                 int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % hidden);
    const int64_t p = i / hidden;
    const int b = (int)(p / HW);
    dst[p * ld + coff + c] = lat[(int64_t)(Bm == 1 ? 0 : b) * hidden + c];
  }
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void outc_tanh_kernel(const float* __restrict__ x, int64_t rpf, int64_t rows, int C,
                                                        int64_t ld, const float* __restrict__ w, const float* __restrict__ bias,
                                                        int Cout, int use_tanh, float* __restrict__ out) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* xr = x + row * ld;
  for (int c = 0; c < C; c += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
    for (int o = 0; o < Cout; ++o) {
      const float* wr = w + (int64_t)o * C + c;
      acc[o] += v[0] * wr[0];
      if (c + 1 < C) acc[o] += v[1] * wr[1];
      if (c + 2 < C) acc[o] += v[2] * wr[2];
      if (c + 3 < C) acc[o] += v[3] * wr[3];
    }
  }
  const int64_t b = row / rpf, p = row - b * rpf;
  for (int o = 0; o < Cout; ++o) {
    float v = acc[o] + bias[o];
    out[(b * Cout + o) * rpf + p] = use_tanh ? tanhf(v) : v;
  }
}

// ---------------------------------------------------------------------------------------------------
// grid = (frames, output slices): every block pools its frame (the activation is tiny and L2-resident) and then produces
// a slice of the N outputs, one 64-lane wave per output.
__global__ __launch_bounds__(256) void pool_linear_kernel(const float* __restrict__ x, int HW, int C, int64_t ld,
                                                          const float* __restrict__ w, const float* __restrict__ bias, int N,
                                                          float* __restrict__ out) {
  extern __shared__ float pooled[];   // [C]
  const int b = blockIdx.x;
  const float* xb = x + (int64_t)b * HW * ld;
  // round 6: a thread owns a 16-byte group of four channels and requests sixteen rows before the first is added -- 4 round trips for the 64 rows of
  // an 8 x 8 map instead of 3 channels x 8 (the kernel was 32 us for a 6 MB tensor, a chain of dependent L2 round trips).  Per channel the rows are
  // still added in row order: the same sums, bit for bit.  (ld is a multiple of 4 and the pad lanes are zero: vs_pool_linear checks alignment)
  const int C4 = (C + 3) >> 2;
  const bool vec = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  if (vec) {
    for (int g = threadIdx.x; g < C4; g += 256) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      const float* xg = xb + 4 * g;
      int r = 0;
      for (; r + 16 <= HW; r += 16) {
        f32x4 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = *reinterpret_cast<const f32x4*>(xg + (int64_t)(r + q) * ld);
#pragma unroll
        for (int q = 0; q < 16; ++q) s += v[q];
      }
      for (; r < HW; ++r) s += *reinterpret_cast<const f32x4*>(xg + (int64_t)r * ld);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * g + e < C) pooled[4 * g + e] = s[e] / (float)HW;
    }
  } else {
    for (int c = threadIdx.x; c < C; c += 256) {
      float s = 0.f;
      for (int r = 0; r < HW; ++r) s += xb[(int64_t)r * ld + c];
      pooled[c] = s / (float)HW;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // (round 6, second pass: the weight row of an output was read one value per loop iteration with a wait behind each -- C / 64 = 12 dependent L2 round
  //  trips per output, four or five outputs per wave: 28 us for a 32 x 257 result.  Now the row's values are requested first and multiplied in the
  //  same lane order: the same sums.)
  constexpr int CM = 16;                     // lanes x 16 = up to 1024 channels with every weight of a row in flight
  for (int n = blockIdx.y * 4 + wv; n < N; n += 4 * gridDim.y) {
    const float* wr = w + (int64_t)n * C;
    float s = 0.f;
    if (C <= 64 * CM) {
      float wv_[CM];
#pragma unroll
      for (int j = 0; j < CM; ++j) {
        const int c = lane + 64 * j;
        wv_[j] = c < C ? wr[c] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < CM; ++j) {
        const int c = lane + 64 * j;
        if (c < C) s += pooled[c] * wv_[j];
      }
    } else {
      for (int c = lane; c < C; c += 64) s += pooled[c] * wr[c];
    }
    s = wave_sum(s);
    if (lane == 0) out[(int64_t)b * N + n] = s + bias[n];
  }
}

// ---------------------------------------------------------------------------------------------------
// BatchNorm2d with BATCH statistics (the U-Net's ResnetBlocks under model.train(): unet.py:26,30 -> nn.BatchNorm2d).
// Two deterministic stages over an NHWC tensor [rows][ld]:
//   bn_partial_kernel : every block sums x and x^2 of one chunk of rows per channel (fp64 accumulators, fixed order)
//   bn_finish_kernel  : mean / biased variance -> scale = gamma/sqrt(var+eps), shift = beta - mean*scale, and the
//                       running-statistics update of nn.BatchNorm2d (momentum m, UNBIASED variance) in place.
// Chunk geometry: a block covers G = min(ld / 4, 64) float4 channel groups x RL = 256 / G row lanes; every row lane reads 16 rows (16
// independent 16-byte loads in flight per thread), so a chunk is 16 * RL rows and a launch has ceil(rows / chunk) x ceil(ld / 4 / G) blocks --
// hundreds of workgroups for every tensor of the U-Net (the previous 2048-row chunks gave 8 workgroups for a 32 x 32 x 384 map).
static inline int bn_groups(int64_t ld) { const int c4 = (int)(ld >> 2); return c4 < 64 ? c4 : 64; }
static inline int bn_chunk_rows(int64_t ld) { return 16 * (256 / bn_groups(ld)); }
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, int64_t rows, int64_t ld, int G, int RL,
                                                         double* __restrict__ partial) {
  __shared__ double red[2][256][4];
  const int C4 = (int)(ld >> 2);
  const int g = threadIdx.x % G, rl = threadIdx.x / G;
  const int gg = blockIdx.y * G + g;
  const int64_t r0 = (int64_t)blockIdx.x * (16 * RL);
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  const bool live = rl < RL && gg < C4;
  if (live) {
    f32x4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int64_t r = r0 + rl + (int64_t)j * RL;
      v[j] = r < rows ? *reinterpret_cast<const f32x4*>(x + r * ld + 4 * gg) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[e] += (double)v[j][e]; q[e] += (double)v[j][e] * (double)v[j][e]; }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[0][threadIdx.x][e] = s[e]; red[1][threadIdx.x][e] = q[e]; }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {         // fixed tree over the row lanes (RL <= 256)
    if (live && rl < o && rl + o < RL)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[0][threadIdx.x][e] += red[0][threadIdx.x + o * G][e];
        red[1][threadIdx.x][e] += red[1][threadIdx.x + o * G][e];
      }
    __syncthreads();
  }
  if (live && rl == 0)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      partial[((int64_t)blockIdx.x * 2 + 0) * ld + 4 * gg + e] = red[0][threadIdx.x][e];
      partial[((int64_t)blockIdx.x * 2 + 1) * ld + 4 * gg + e] = red[1][threadIdx.x][e];
    }
}
static inline void launch_bn_partial(const float* x, int64_t rows, int64_t ld, double* partial, hipStream_t st) {
  const int G = bn_groups(ld), RL = 256 / G;
  const int C4 = (int)(ld >> 2);
  hipLaunchKernelGGL(bn_partial_kernel, dim3((unsigned)cdiv64(rows, 16 * RL), (unsigned)((C4 + G - 1) / G)), dim3(256), 0, st, x, rows, ld, G, RL, partial);
}
// one wave per channel: the lanes stride the chunks, then a butterfly (every lane ends with the same totals; fixed order)
__device__ __forceinline__ void bn_chunk_totals(const double* __restrict__ partial, int nchunk, int64_t ld, int64_t c, double& s, double& q) {
  s = 0; q = 0;
  for (int k = threadIdx.x; k < nchunk; k += 64) { s += partial[((int64_t)k * 2 + 0) * ld + c]; q += partial[((int64_t)k * 2 + 1) * ld + c]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
}
__global__ __launch_bounds__(64) void bn_finish_kernel(const double* __restrict__ partial, int nchunk, int64_t rows, int C, int64_t ld,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                       float momentum, float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, float* __restrict__ scale,
                                                       float* __restrict__ shift, const double* __restrict__ count) {
  const int c = blockIdx.x;
  double s, q;
  bn_chunk_totals(partial, nchunk, ld, c, s, q);
  if (threadIdx.x != 0) return;
  const double n = count ? *count : (double)rows;      // SyncBatchNorm: the row count of ALL ranks, reduced with the sums
  const double mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0) var = 0;
  const float sc = gamma[c] * (float)(1.0 / sqrt(var + (double)eps));
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(n > 1 ? var * n / (n - 1) : var);
}
// SyncBatchNorm's local half: the chunk partials of bn_partial_kernel summed in the same fixed order as bn_finish_kernel does, into
// sums = [sum x | sum x^2 | rows] = 2*ld + 1 doubles -- the vector the ranks all-reduce (train.py:438-440).
__global__ __launch_bounds__(64) void bn_reduce_kernel(const double* __restrict__ partial, int nchunk, int64_t rows, int64_t ld,
                                                       double* __restrict__ sums) {
  const int64_t c = blockIdx.x;
  double s, q;
  bn_chunk_totals(partial, nchunk, ld, c, s, q);
  if (threadIdx.x != 0) return;
  if (c == 0) sums[2 * ld] = (double)rows;
  sums[c] = s;
  sums[ld + c] = q;
}
// out[r][c] = act(x[r][c] * scale[c] + shift[c]) (+ add[r][c]);  C % 4 == 0
__global__ __launch_bounds__(256) void scale_shift_act_kernel(const float* __restrict__ x, int64_t rows, int C, int64_t ld,
                                                              const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                              const float* __restrict__ add, int64_t add_ld, float* __restrict__ out,
                                                              int64_t out_ld) {
  const int C4 = C >> 2;
  const int64_t total = rows * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / C4;
    const int c = (int)(i - r * C4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ld + c);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = vs_apply_act(v[e] * sc[e] + sh[e], act);
    if (add) o += *reinterpret_cast<const f32x4*>(add + r * add_ld + c);
    *reinterpret_cast<f32x4*>(out + r * out_ld + c) = o;
  }
}

inline unsigned grid_for(int64_t total, int per_block = 256, int64_t cap = 256 * 32) {
  int64_t g = cdiv64(total, per_block);
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// flag |= 1 when x holds a non-finite value (the guard of the 2 x f16 arithmetic at the network boundary: an activation beyond the f16 range
// of the operand split surfaces as inf / NaN in the network's output -- NaN-propagating ReLU / clamp keep it alive up to here)
__global__ __launch_bounds__(256) void check_finite_kernel(const float* __restrict__ x, int64_t n, int* __restrict__ flag) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const unsigned u = __float_as_uint(x[i]);
    bad |= (u & 0x7f800000u) == 0x7f800000u;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// *bits = max(*bits, bit pattern of max |x|): non-negative floats order like their bit patterns (inf / NaN sort above every finite value).
// Calibration of the per-layer arithmetic choice (engine.py): which dense layers see operands beyond the f16 range of the 2 x f16 split.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ bits) {
  unsigned m = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const unsigned u = __float_as_uint(x[i]) & 0x7fffffffu;
    m = u > m ? u : m;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const unsigned v = (unsigned)__shfl_xor((int)m, o);
    m = v > m ? v : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(bits, m);
}

}  // namespace

extern "C" int vs_layernorm_act(const float* x, int64_t rows, int C, int64_t ld, const float* w, const float* b, float eps,
                                int act, float* out, int64_t out_ld, void* stream) {
  VS_REQUIRE(x && w && b && out && rows > 0 && C > 0 && ld >= C && out_ld >= C);
  if (C <= 64 && ld % 4 == 0 && out_ld % 4 == 0 && out_ld == ((C + 3) / 4) * 4 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
    const unsigned g = (unsigned)cdiv64(rows, 256);
    if (C <= 16) hipLaunchKernelGGL(layernorm_act_small_kernel<4>, dim3(g), dim3(256), 0, (hipStream_t)stream, x, rows, C, ld, w, b, eps, act, out, out_ld);
    else if (C <= 32) hipLaunchKernelGGL(layernorm_act_small_kernel<8>, dim3(g), dim3(256), 0, (hipStream_t)stream, x, rows, C, ld, w, b, eps, act, out, out_ld);
    else hipLaunchKernelGGL(layernorm_act_small_kernel<16>, dim3(g), dim3(256), 0, (hipStream_t)stream, x, rows, C, ld, w, b, eps, act, out, out_ld);
    return vs_launch_status();
  }
  // lanes-per-row form (round 6): 16-byte aligned rows, the output's pad lanes within reach of the lanes that own the row; the affine parameters
  // must be readable in whole float4 (the engine pads them to ld; vs_layernorm_act's contract says C entries: only C % 4 == 0 takes this form then)
  const int C4 = (C + 3) / 4, O4 = (int)(out_ld / 4);
  static const bool env_wave = [] { const char* e = getenv("VIDEOSEAL_LN"); return e && !strcmp(e, "wave"); }();      // A/B handle
  if (!env_wave && vs_debug_get(VS_DBG_LN_FORM) != 1 && ld % 4 == 0 && out_ld % 4 == 0 && C % 4 == 0 && O4 <= 64 * 12 &&
      (((uintptr_t)x | (uintptr_t)out | (uintptr_t)w | (uintptr_t)b) & 15) == 0 && rows < (1LL << 31)) {
    const int n4 = O4 > C4 ? O4 : C4;
    hipStream_t st = (hipStream_t)stream;
#define VS_LN_LAUNCH(LPP, VMAX) hipLaunchKernelGGL((layernorm_act_lanes_kernel<LPP, VMAX>), dim3((unsigned)cdiv64(rows, 256 / LPP)), dim3(256), 0, st, x, rows, C, ld, w, b, eps, act, out, out_ld, 0, 0)
    if (n4 <= 8 * 4) VS_LN_LAUNCH(8, 4);
    else if (n4 <= 16 * 4) VS_LN_LAUNCH(16, 4);
    else if (n4 <= 32 * 4) VS_LN_LAUNCH(32, 4);
    else if (n4 <= 64 * 4) VS_LN_LAUNCH(64, 4);
    else VS_LN_LAUNCH(64, 12);
#undef VS_LN_LAUNCH
    return vs_launch_status();
  }
  hipLaunchKernelGGL(layernorm_act_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, C, ld,
                     w, b, eps, act, out, out_ld);
  return vs_launch_status();
}

// LayerNorm of every pixel of [B][H][W] (C channels, stride ld) written as the patch matrix of a 2 x 2 / stride-2 conv: out [B][H/2][W/2][4 C]
// (tap-major (ky, kx), then channel; odd H / W: the last row / column is dropped).  C % 4 == 0, 16-byte aligned operands; otherwise unsupported
// (the caller keeps vs_layernorm_act + the strided conv).
extern "C" int vs_layernorm_patch2x2(const float* x, int B, int H, int W, int C, int64_t ld, const float* w, const float* b, float eps, float* out,
                                     void* stream) {
  VS_REQUIRE(x && w && b && out && B > 0 && H >= 2 && W >= 2 && C > 0 && ld >= C);
  const int C4 = C / 4;
  if (C % 4 != 0 || ld % 4 != 0 || C4 > 64 * 12 || (((uintptr_t)x | (uintptr_t)out | (uintptr_t)w | (uintptr_t)b) & 15) != 0) return VS_ERR_UNSUPPORTED;
  const int64_t rows = (int64_t)B * H * W;
  if (rows >= (1LL << 31)) return VS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int act = VS_ACT_NONE;
  const int64_t out_ld = C;
#define VS_LNP_LAUNCH(LPP, VMAX) hipLaunchKernelGGL((layernorm_act_lanes_kernel<LPP, VMAX>), dim3((unsigned)cdiv64(rows, 256 / LPP)), dim3(256), 0, st, x, rows, C, ld, w, b, eps, act, out, out_ld, H, W)
  if (C4 <= 8 * 4) VS_LNP_LAUNCH(8, 4);
  else if (C4 <= 16 * 4) VS_LNP_LAUNCH(16, 4);
  else if (C4 <= 32 * 4) VS_LNP_LAUNCH(32, 4);
  else if (C4 <= 64 * 4) VS_LNP_LAUNCH(64, 4);
  else VS_LNP_LAUNCH(64, 12);
#undef VS_LNP_LAUNCH
  return vs_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// Round 6: the extractor's stem -- 4 x 4 patchify conv (3 -> CO channels, stride 4 or 2; convnext.py:100-104) + its LayerNorm -- in ONE kernel on the
// vector ALUs.  The pair ran as a generic-GEMM launch (K = 48 of the 16-wide matrix instruction's step: 40 us, memory-bound at 2 TB/s) plus a LayerNorm
// launch over the 50 MB it had just written (25 us).  0.6 GMAC for 32 frames is nothing for fp32 FMAs (exact fp32 products, fp32 accumulation: at least the
// accuracy of the split-operand matrix path), and the LayerNorm is two 2-step shuffles when FOUR lanes share a pixel (CO / 4 channels each): the lanes of a
// pixel write 4 x (CO) bytes contiguously, a wave 16 pixels = 6 KiB at CO = 96.  Weights: the packed rows of the conv ([CO][4 ky][16 = 4 kx x (3 + pad)])
// transposed into LDS as [k][CO] once per workgroup.
template <int CO>
__global__ __launch_bounds__(256) void stem_conv_ln_kernel(const float* __restrict__ x, int B, int H, int W, int Ho, int Wo, int stride,
                                                           const float* __restrict__ wt, const float* __restrict__ bias, const float* __restrict__ lnw,
                                                           const float* __restrict__ lnb, float eps, float* __restrict__ out, int64_t out_ld) {
  constexpr int CQ = CO / 4, Q4 = CQ / 4;                   // channels / float4 per lane
  __shared__ __attribute__((aligned(16))) float s_w[48 * CO];          // [k = ky * 12 + kx * 3 + c][CO]
  for (int i = threadIdx.x; i < 48 * CO; i += 256) {
    const int k = i / CO, n = i - k * CO;
    const int ky = k / 12, r = k - ky * 12, kx = r / 3, c = r - kx * 3;
    s_w[i] = wt[(int64_t)n * 64 + ky * 16 + kx * 4 + c];
  }
  __syncthreads();
  const int qd = threadIdx.x & 3;
  const int64_t npix = (int64_t)B * Ho * Wo;
  const int64_t pix = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 2);
  const bool live = pix < npix;
  const int64_t pc = live ? pix : npix - 1;                 // (past the end: a valid pixel, nothing stored -- the shuffles need every lane)
  const int ox = (int)(pc % Wo);
  const int64_t t = pc / Wo;
  const int oy = (int)(t % Ho);
  const int64_t b = t / Ho;
  const float* xp = x + ((b * H + (int64_t)oy * stride) * W + (int64_t)ox * stride) * 4;
  f32x4 acc[Q4];
#pragma unroll
  for (int j = 0; j < Q4; ++j) acc[j] = *reinterpret_cast<const f32x4*>(bias + qd * CQ + 4 * j);
#pragma unroll
  for (int ky = 0; ky < 4; ++ky) {
    f32x4 in[4];
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) in[kx] = *reinterpret_cast<const f32x4*>(xp + ((int64_t)ky * W + kx) * 4);
#pragma unroll
    for (int kx = 0; kx < 4; ++kx)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v = in[kx][c];
        const float* wr = s_w + (ky * 12 + kx * 3 + c) * CO + qd * CQ;
#pragma unroll
        for (int j = 0; j < Q4; ++j) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + 4 * j);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[j][e] = __builtin_fmaf(v, wv[e], acc[j][e]);
        }
      }
  }
  // LayerNorm over the pixel's CO channels (four lanes): mean, then the squared deviations (common.py:147-155)
  float sm = 0.f;
#pragma unroll
  for (int j = 0; j < Q4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) sm += acc[j][e];
  sm += __shfl_xor(sm, 1, 64);
  sm += __shfl_xor(sm, 2, 64);
  const float mean = sm / (float)CO;
  float qv = 0.f;
#pragma unroll
  for (int j = 0; j < Q4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float dlt = acc[j][e] - mean; qv += dlt * dlt; }
  qv += __shfl_xor(qv, 1, 64);
  qv += __shfl_xor(qv, 2, 64);
  const float den = sqrtf(qv / (float)CO + eps);
  if (!live) return;
  float* orow = out + pix * out_ld + qd * CQ;
#pragma unroll
  for (int j = 0; j < Q4; ++j) {
    const f32x4 wv = *reinterpret_cast<const f32x4*>(lnw + qd * CQ + 4 * j), bv = *reinterpret_cast<const f32x4*>(lnb + qd * CQ + 4 * j);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = wv[e] * ((acc[j][e] - mean) / den) + bv[e];
    *reinterpret_cast<f32x4*>(orow + 4 * j) = o;
  }
}

// x: NHWC frames with 4 floats per pixel (rgb + a zero lane), wt: the stem's packed rows [CO][64] (engine.pack_patch_conv(w, 4)), out [B][Ho][Wo][out_ld].
// CO in {64, 96, 128} and out_ld == CO, 16-byte aligned operands; otherwise unsupported (the caller keeps the conv + vs_layernorm_act pair).
extern "C" int vs_stem_conv_ln(const float* x, int B, int H, int W, int stride, const float* wt, const float* bias, const float* lnw, const float* lnb,
                               float eps, int CO, float* out, int64_t out_ld, void* stream) {
  VS_REQUIRE(x && wt && bias && lnw && lnb && out && B > 0 && H >= 4 && W >= 4 && (stride == 4 || stride == 2));
  if ((CO != 64 && CO != 96 && CO != 128) || out_ld != CO ||
      (((uintptr_t)x | (uintptr_t)wt | (uintptr_t)bias | (uintptr_t)lnw | (uintptr_t)lnb | (uintptr_t)out) & 15) != 0)
    return VS_ERR_UNSUPPORTED;
  const int Ho = (H - 4) / stride + 1, Wo = (W - 4) / stride + 1;
  const int64_t npix = (int64_t)B * Ho * Wo;
  if (npix >= (1LL << 31) * 16) return VS_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)cdiv64(npix, 64));
  hipStream_t st = (hipStream_t)stream;
  if (CO == 96) hipLaunchKernelGGL(stem_conv_ln_kernel<96>, grid, dim3(256), 0, st, x, B, H, W, Ho, Wo, stride, wt, bias, lnw, lnb, eps, out, out_ld);
  else if (CO == 64) hipLaunchKernelGGL(stem_conv_ln_kernel<64>, grid, dim3(256), 0, st, x, B, H, W, Ho, Wo, stride, wt, bias, lnw, lnb, eps, out, out_ld);
  else hipLaunchKernelGGL(stem_conv_ln_kernel<128>, grid, dim3(256), 0, st, x, B, H, W, Ho, Wo, stride, wt, bias, lnw, lnb, eps, out, out_ld);
  return vs_launch_status();
}

static int dwconv7_ln_any(const float* x, int B, int H, int W, int C, int64_t ld, const float* wdw, const float* bdw,
                          const float* lnw, const float* lnb, float eps, float* out, int64_t out_ld, void* stream, PlanesOut pl) {
  VS_REQUIRE(x && wdw && bdw && lnw && lnb && out && B > 0 && H > 0 && W > 0 && C > 0);
  VS_REQUIRE(ld % 4 == 0 && ld >= C && (pl.cstride || (out_ld >= C && out_ld % 4 == 0)));
  // LDS-tiled kernels where the tile + the [pixel][channel] LayerNorm buffer fit the 160 KB of a CU and the map is large enough to
  // give every CU a tile; every variant produces bit-identical values (same FMA order), so the choice is purely a speed matter.
  // VS_DWCONV=0 forces the one-row kernel, 1 / 2 a tiled configuration (tools/bench_dwconv.py).
  static const int force = [] { const char* e = getenv("VS_DWCONV"); return e ? atoi(e) : -1; }();
  {
    const int64_t hw = (int64_t)H * W;
    int cfg = 0;
    if (force >= 0) cfg = force;
    else if (hw >= 4096) cfg = 5;              // measured on MI355X, B = 32 (tools/bench_dwconv.py): 64^2 x 96: 117 -> 60 us,
    else if (hw >= 1024) cfg = 3;              // 32^2 x 192: 65 -> 38 us; 16^2 x 384: 32 -> 23 us (128-channel chunks: 256 work items
    else if (hw >= 256) cfg = 6;               // per 4 x 8 tile); 8^2 x 768: the one-row kernel stays ahead (24 us)
    int rc = VS_ERR_UNSUPPORTED;
    if (cfg == 1) rc = launch_dwconv_tiled<8, 8, 48, 256>(x, B, H, W, C, ld, wdw, bdw, lnw, lnb, eps, out, out_ld, (hipStream_t)stream, pl);
    else if (cfg == 2) rc = launch_dwconv_tiled<4, 8, 64, 128>(x, B, H, W, C, ld, wdw, bdw, lnw, lnb, eps, out, out_ld, (hipStream_t)stream, pl);
    else if (cfg == 3) rc = launch_dwconv_tiled<8, 16, 32, 256>(x, B, H, W, C, ld, wdw, bdw, lnw, lnb, eps, out, out_ld, (hipStream_t)stream, pl);
    else if (cfg == 4) rc = launch_dwconv_tiled<4, 16, 64, 256>(x, B, H, W, C, ld, wdw, bdw, lnw, lnb, eps, out, out_ld, (hipStream_t)stream, pl);
    else if (cfg == 5) rc = launch_dwconv_tiled<4, 16, 48, 256>(x, B, H, W, C, ld, wdw, bdw, lnw, lnb, eps, out, out_ld, (hipStream_t)stream, pl);
    else if (cfg == 6) rc = launch_dwconv_tiled<4, 8, 128, 256>(x, B, H, W, C, ld, wdw, bdw, lnw, lnb, eps, out, out_ld, (hipStream_t)stream, pl);
    else if (cfg == 7) rc = launch_dwconv_tiled<4, 4, 128, 128>(x, B, H, W, C, ld, wdw, bdw, lnw, lnb, eps, out, out_ld, (hipStream_t)stream, pl);
    else if (cfg == 8) rc = launch_dwconv_tiled<4, 4, 64, 128>(x, B, H, W, C, ld, wdw, bdw, lnw, lnb, eps, out, out_ld, (hipStream_t)stream, pl);
    else if (cfg == 9) rc = launch_dwconv_tiled<4, 4, 32, 128>(x, B, H, W, C, ld, wdw, bdw, lnw, lnb, eps, out, out_ld, (hipStream_t)stream, pl);
    else if (cfg == 10) rc = launch_dwconv_tiled<2, 8, 64, 128>(x, B, H, W, C, ld, wdw, bdw, lnw, lnb, eps, out, out_ld, (hipStream_t)stream, pl);
    if (rc == VS_ERR_UNSUPPORTED && force < 0 && cfg == 3)      // tile does not fit / ld % 32: the other tiled shape
      rc = launch_dwconv_tiled<4, 16, 48, 256>(x, B, H, W, C, ld, wdw, bdw, lnw, lnb, eps, out, out_ld, (hipStream_t)stream, pl);
    if (rc != VS_ERR_UNSUPPORTED) return rc;
  }
  const int C4 = (int)(ld / 4);
  int NS = 256 / C4;
  if (NS < 1) NS = 1;
  const int spr = (W + 3) / 4;
  // two output rows per strip where ONE strip of all channels fills the workgroup (more than 512 channels: ChunkySeal's 1448- / 2896-channel maps,
  // VideoSeal's 8 x 8 x 768 stage) and the strips still cover the CUs; VS_DWCONV_ROWS=1 keeps single rows.  Measured (tools/bench_dwconv.py, one box):
  // 31 x 31 x 1472, 16 frames 161.6 -> 112.4 us; 15 x 15 x 2912 97.1 -> 86.5; 8 x 8 x 768, 32 frames 22.8 -> 15.1 us.  The conv values are bit-identical
  // to the single-row strips; the LayerNorm behind them sums a pixel's channels over more or fewer lanes depending on the pixels per workgroup
  // (ln_rows_from_lds), so the outputs agree to fp32 rounding, not bit for bit
  static const int force_rows = [] { const char* e = getenv("VS_DWCONV_ROWS"); return e ? atoi(e) : 0; }();
  const int dbg_rows = vs_debug_get(VS_DBG_DWCONV_ROWS);                    // tests: 1 / 2 rows per strip whatever the shape
  const bool two = dbg_rows ? dbg_rows == 2 : force_rows ? force_rows == 2 : (NS == 1 && H >= 2 && (int64_t)B * ((H + 1) / 2) * spr >= vs_num_cus());
  const int R = two ? 2 : 1;
  const int64_t nstrips = (int64_t)B * ((H + R - 1) / R) * spr;
  const size_t smem = ((size_t)NS * 4 * R * (ld + 4) + 2 * (size_t)C) * sizeof(float);
  if (smem > 160 * 1024) return VS_ERR_UNSUPPORTED;
  if (two) {
    if (smem > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)dwconv7_ln_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(dwconv7_ln_kernel<2>, dim3((unsigned)cdiv64(nstrips, NS)), dim3(256), smem, (hipStream_t)stream, x, B, H, W,
                       C, ld, wdw, bdw, lnw, lnb, eps, out, out_ld, NS, spr, nstrips, pl);
    return vs_launch_status();
  }
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)dwconv7_ln_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(dwconv7_ln_kernel<1>, dim3((unsigned)cdiv64(nstrips, NS)), dim3(256), smem, (hipStream_t)stream, x, B, H, W,
                     C, ld, wdw, bdw, lnw, lnb, eps, out, out_ld, NS, spr, nstrips, pl);
  return vs_launch_status();
}

extern "C" int vs_dwconv7_ln(const float* x, int B, int H, int W, int C, int64_t ld, const float* wdw, const float* bdw,
                             const float* lnw, const float* lnb, float eps, float* out, int64_t out_ld, void* stream) {
  return dwconv7_ln_any(x, B, H, W, C, ld, wdw, bdw, lnw, lnb, eps, out, out_ld, stream, PlanesOut{0, 0, 1.f, 0});
}
// the same with the LayerNorm output written as the operand planes of vs_conv_gemm tile codes 24 / 25: [2][ceil(C/16)][B*H*W][16] f16
// (hi / lo of value * a_mul, channels >= C zero)
extern "C" int vs_dwconv7_ln_planes(const float* x, int B, int H, int W, int C, int64_t ld, const float* wdw, const float* bdw,
                                    const float* lnw, const float* lnb, float eps, float a_mul, int planes_C, void* planes, void* stream) {
  VS_REQUIRE(planes && a_mul > 0.f && ((uintptr_t)planes & 15) == 0 && planes_C >= C && planes_C % 16 == 0);
  const int64_t rows = (int64_t)B * H * W, cstride = rows * 32;
  return dwconv7_ln_any(x, B, H, W, C, ld, wdw, bdw, lnw, lnb, eps, static_cast<float*>(planes), 8, stream,
                        PlanesOut{cstride, (int64_t)(planes_C / 16) * cstride, a_mul, planes_C});
}

extern "C" int vs_grn_scale(const float* h, int B, int HW, int C, int64_t ld, const float* gamma, float* partial,
                            float* scale, void* stream) {
  VS_REQUIRE(h && gamma && partial && scale && B > 0 && HW > 0 && C > 0 && ld >= C);
  const int nchunk = (HW + GRN_ROWS - 1) / GRN_ROWS;
  hipLaunchKernelGGL(grn_partial_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)nchunk, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, h, HW, C, ld, partial, B);
  if (C <= 512) hipLaunchKernelGGL((grn_finish_kernel<2, 16>), dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, partial, nchunk, B, C, gamma,
                     scale, ld, 0);
  else if (C <= 1024) hipLaunchKernelGGL((grn_finish_kernel<4, 8>), dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, partial, nchunk, B, C, gamma,
                     scale, ld, 0);
  else hipLaunchKernelGGL((grn_finish_kernel<8, 4>), dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, partial, nchunk, B, C, gamma,
                     scale, ld, 0);
  return vs_launch_status();
}

// h[b, r, c] = h[b, r, c] * scale[b][c] + beta[c] in place (pad lanes untouched): the GRN apply as its own pass, for layers whose
// frames do not align with the 64-row halves the fused GEMM transform assumes
__global__ __launch_bounds__(256) void grn_apply_kernel(float* __restrict__ h, int64_t rows, int HW, int C, int64_t ld,
                                                        const float* __restrict__ scale, int64_t sld, const float* __restrict__ beta) {
  const int C4 = (C + 3) >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * C4) return;
  const int64_t m = idx / C4;
  const int c = (int)(idx - m * C4) * 4;
  const int b = (int)(m / HW);
  f32x4 v = *reinterpret_cast<f32x4*>(h + m * ld + c);
  const f32x4 s = *reinterpret_cast<const f32x4*>(scale + (int64_t)b * sld + c);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(beta + c);
  const f32x4 o = v * s + sh;                     // same expression as the GEMM's fused transform
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = c + e < C ? o[e] : v[e];
  *reinterpret_cast<f32x4*>(h + m * ld + c) = v;
}

extern "C" int vs_grn_apply(float* h, int B, int HW, int C, int64_t ld, const float* scale, int64_t scale_ld, const float* beta,
                            void* stream) {
  VS_REQUIRE(h && scale && beta && B > 0 && HW > 0 && C > 0 && ld >= C && (ld & 3) == 0 && (scale_ld & 3) == 0 && scale_ld >= ((C + 3) & ~3));
  const int64_t rows = (int64_t)B * HW, items = rows * ((C + 3) >> 2);
  hipLaunchKernelGGL(grn_apply_kernel, dim3((unsigned)cdiv64(items, 256)), dim3(256), 0, (hipStream_t)stream, h, rows, HW, C, ld, scale,
                     scale_ld, beta);
  return vs_launch_status();
}

extern "C" int vs_grn_scale_from_partials(const float* partial, int B, int HW, int C, const float* gamma, float* scale,
                                          int64_t scale_ld, void* stream) {
  VS_REQUIRE(partial && gamma && scale && B > 0 && HW > 0 && HW % 32 == 0 && C > 0 && scale_ld >= C);
  if (C <= 512) hipLaunchKernelGGL((grn_finish_kernel<2, 16>), dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, partial, HW / 32, B, C, gamma,
                     scale, scale_ld, 1);
  else if (C <= 1024) hipLaunchKernelGGL((grn_finish_kernel<4, 8>), dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, partial, HW / 32, B, C, gamma,
                     scale, scale_ld, 1);
  else hipLaunchKernelGGL((grn_finish_kernel<8, 4>), dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, partial, HW / 32, B, C, gamma,
                     scale, scale_ld, 1);
  return vs_launch_status();
}

// Gx of frame b from the straddling 32-row-group partials of conv_common.h::write_sumsq_straddle ([groups][2][C]): the groups g_lo .. g_hi that
// overlap the frame's rows, ascending; a group whose first row lies in the frame contributes slot 0, the one that starts in frame b - 1 slot 1.
// One workgroup per frame, a thread walks channels c = tid, tid + 1024, ...; eight groups' loads are issued before the first addition.
__global__ __launch_bounds__(1024) void grn_finish_straddle_kernel(const float* __restrict__ partial, int HW, int C, const float* __restrict__ gamma,
                                                                   float* __restrict__ scale, int64_t sld) {
  __shared__ float red[1024];
  const int b = blockIdx.x;
  const int64_t r_lo = (int64_t)b * HW, r_hi = r_lo + HW - 1;
  const int g_lo = (int)(r_lo >> 5), g_hi = (int)(r_hi >> 5);
  // a thread owns the channels c = tid + 1024 j (up to CPT = 6 of them per pass: ChunkySeal's 5 792 in one pass); the loads of eight groups x CPT channels are
  // issued before the first addition (the first version walked one channel at a time: ~24 dependent L2 round trips per thread on ChunkySeal's
  // 5 792 channels x 31 groups, 16 workgroups on the whole chip)
  constexpr int CPT = 6, GB = 8;
  float local = 0.f;
  for (int cb = 0; cb < C; cb += 1024 * CPT) {
    float s[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) s[j] = 0.f;
    for (int g0 = g_lo; g0 <= g_hi; g0 += GB) {
      float v[GB][CPT];
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        const int g = g0 + q <= g_hi ? g0 + q : g_hi;              // (past the last group: a valid address, value unused)
        const int slot = ((int64_t)g * 32) / HW == b ? 0 : 1;
        const float* row = partial + ((int64_t)g * 2 + slot) * C;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
          const int c = cb + threadIdx.x + 1024 * j;
          v[q][j] = row[c < C ? c : C - 1];
        }
      }
#pragma unroll
      for (int q = 0; q < GB; ++q)
        if (g0 + q <= g_hi) {
#pragma unroll
          for (int j = 0; j < CPT; ++j) s[j] += v[q][j];
        }
    }
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int c = cb + threadIdx.x + 1024 * j;
      if (c < C) {
        const float gx = sqrtf(s[j]);
        scale[(int64_t)b * sld + c] = gx;
        local += gx;
      }
    }
  }
  red[threadIdx.x] = local;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float mean = red[0] / (float)C;
  for (int c = threadIdx.x; c < (int)sld; c += 1024) {
    float v = 0.f;
    if (c < C) v = 1.0f + gamma[c] * (scale[(int64_t)b * sld + c] / (mean + 1e-6f));
    scale[(int64_t)b * sld + c] = v;
  }
}

extern "C" int vs_grn_scale_from_straddle_partials(const float* partial, int B, int HW, int C, const float* gamma, float* scale,
                                                   int64_t scale_ld, void* stream) {
  VS_REQUIRE(partial && gamma && scale && B > 0 && HW >= 32 && C > 0 && scale_ld >= C);
  hipLaunchKernelGGL(grn_finish_straddle_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, partial, HW, C, gamma, scale, scale_ld);
  return vs_launch_status();
}

extern "C" int vs_upcat2x(const float* x, int C1, int64_t ld1, const float* skip, int C2, int64_t ld2, float skip_scale,
                          int B, int H, int W, float* out, int64_t out_ld, void* stream) {
  VS_REQUIRE(x && skip && out && C1 > 0 && C2 > 0 && C1 % 4 == 0 && C2 % 4 == 0 && ld1 % 4 == 0 && ld2 % 4 == 0);
  VS_REQUIRE(out_ld % 4 == 0 && out_ld >= C1 + C2 && B > 0 && H > 0 && W > 0);
  const int64_t total = (int64_t)B * 2 * H * 2 * W * (out_ld / 4);
  hipLaunchKernelGGL(upcat2x_kernel, dim3(grid_for(total, 256, 1 << 20)), dim3(256), 0, (hipStream_t)stream, x, C1, ld1, skip,
                     C2, ld2, skip_scale, B, H, W, out, out_ld, total);
  return vs_launch_status();
}

extern "C" int vs_cat2_scale(const float* x, int C1, int64_t ld1, const float* skip, int C2, int64_t ld2, float skip_scale,
                             int64_t rows, float* out, int64_t out_ld, void* stream) {
  VS_REQUIRE(skip && out && rows > 0 && C1 >= 0 && C2 > 0 && C1 % 4 == 0 && C2 % 4 == 0 && ld2 % 4 == 0 && out_ld % 4 == 0 &&
             out_ld >= C1 + C2 && (!x || (ld1 % 4 == 0 && ld1 >= C1)));
  const int64_t total = rows * ((x ? C1 : 0) + C2) / 4;
  hipLaunchKernelGGL(cat2_scale_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 1 << 16)), dim3(256), 0, (hipStream_t)stream,
                     x, C1, ld1, skip, C2, ld2, skip_scale, rows, out, out_ld);
  return vs_launch_status();
}

extern "C" int vs_im2col3x3(const float* x, int B, int H, int W, int64_t ld, int pad_mode, float* out, void* stream) {
  VS_REQUIRE(x && out && B > 0 && H > 1 && W > 1 && ld > 0 && ld % 4 == 0);
  const int64_t total = (int64_t)B * H * W * 9 * (ld / 4);
  hipLaunchKernelGGL(im2col3x3_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 1 << 16)), dim3(256), 0, (hipStream_t)stream, x, H,
                     W, ld, pad_mode == VS_PAD_REFLECT ? 1 : 0, out, total);
  return vs_launch_status();
}

extern "C" int vs_msg_pre(const float* P, int Bm, int N, float* table, void* stream) {
  VS_REQUIRE(P && table && Bm > 0 && N > 0);
  const int64_t total = (int64_t)Bm * 9 * N;
  hipLaunchKernelGGL(msg_pre_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 1 << 16)), dim3(256), 0, (hipStream_t)stream, P, N, table,
                     total);
  return vs_launch_status();
}

extern "C" int vs_upconv_supported(int Co) {
  if (Co < 16 || Co % 16) return 0;
  const int cg = Co >= 64 ? 16 : Co / 4;
  const int tpp = Co / cg;
  return (cg == 4 || cg == 8 || cg == 16) && tpp <= 64 && (tpp & (tpp - 1)) == 0;
}

extern "C" int vs_upconv_gather_ln(const float* z, int64_t z_ld, int B, int H, int W, int Co, const float* lnw, const float* lnb,
                                   float eps, int act, float* out, int64_t out_ld, void* stream) {
  VS_REQUIRE(z && lnw && lnb && out && B > 0 && H > 0 && W > 0 && vs_upconv_supported(Co) && z_ld >= 9 * (int64_t)Co && z_ld % 4 == 0 &&
             out_ld >= Co && out_ld % 4 == 0);
  const int cg = Co >= 64 ? 16 : Co / 4;
  int tl = 0;
  while ((cg << tl) < Co) ++tl;
  const int64_t npix = (int64_t)B * 4 * H * W;
  const int64_t nblk64 = cdiv64(npix << tl, 256);
  VS_REQUIRE(nblk64 < (1 << 30));
  const int nblk = (int)nblk64;
  const int grid = ((nblk + 7) / 8) * 8;
  hipStream_t st = (hipStream_t)stream;
#define VS_UPG(CG_) hipLaunchKernelGGL(upconv_gather_ln_kernel<CG_>, dim3(grid), dim3(256), 0, st, z, z_ld, H, W, Co, tl, lnw, lnb, eps, act, out, out_ld, npix, nblk)
  if (cg == 4) VS_UPG(4);
  else if (cg == 8) VS_UPG(8);
  else VS_UPG(16);
#undef VS_UPG
  return vs_launch_status();
}

extern "C" int vs_msg_latent(const float* table, const int32_t* msgs, int Bm, int nbits, int hidden, float* lat, void* stream) {
  VS_REQUIRE(table && msgs && lat && Bm > 0 && nbits > 0 && nbits <= 1024 && hidden > 0);
  hipLaunchKernelGGL(msg_latent_kernel, dim3((unsigned)((hidden + 255) / 256), (unsigned)Bm), dim3(256), 0, (hipStream_t)stream,
                     table, msgs, nbits, hidden, lat);
  return vs_launch_status();
}

extern "C" int vs_broadcast_channels(const float* lat, int Bm, int hidden, float* dst, int B, int HW, int64_t ld, int coff,
                                     void* stream) {
  VS_REQUIRE(lat && dst && (Bm == 1 || Bm == B) && hidden > 0 && B > 0 && HW > 0 && coff >= 0 && coff + hidden <= ld);
  const int64_t total = (int64_t)B * HW * hidden;
  if ((hidden & 3) == 0 && (ld & 3) == 0 && (coff & 3) == 0 && (((uintptr_t)lat | (uintptr_t)dst) & 15) == 0) {
    hipLaunchKernelGGL(broadcast_channels4_kernel, dim3(grid_for(total / 4, 256, 1 << 20)), dim3(256), 0, (hipStream_t)stream, lat, Bm,
                       hidden / 4, dst, HW, ld, coff, total / 4);
    return vs_launch_status();
  }
  hipLaunchKernelGGL(broadcast_channels_kernel, dim3(grid_for(total, 256, 1 << 20)), dim3(256), 0, (hipStream_t)stream, lat, Bm,
                     hidden, dst, HW, ld, coff, total);
  return vs_launch_status();
}

extern "C" int vs_outc_tanh(const float* x, int64_t rows_per_frame, int B, int C, int64_t ld, const float* w, const float* bias,
                            int Cout, int use_tanh, float* out, void* stream) {
  VS_REQUIRE(x && w && bias && out && rows_per_frame > 0 && B > 0 && C > 0 && ld % 4 == 0 && ld >= C && Cout > 0 && Cout <= 4);
  const int64_t rows = rows_per_frame * B;
  hipLaunchKernelGGL(outc_tanh_kernel, dim3((unsigned)cdiv64(rows, 256)), dim3(256), 0, (hipStream_t)stream, x, rows_per_frame,
                     rows, C, ld, w, bias, Cout, use_tanh, out);
  return vs_launch_status();
}

extern "C" int vs_pool_linear(const float* x, int B, int HW, int C, int64_t ld, const float* w, const float* bias, int N,
                              float* out, void* stream) {
  VS_REQUIRE(x && w && bias && out && B > 0 && HW > 0 && C > 0 && ld >= C && N > 0);
  const size_t smem = (size_t)C * sizeof(float);
  if (smem > 64 * 1024) return VS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(pool_linear_kernel, dim3((unsigned)B, 16), dim3(256), smem, (hipStream_t)stream, x, HW, C, ld, w, bias, N, out);
  return vs_launch_status();
}

extern "C" int64_t vs_bn_partial_doubles(int64_t rows, int64_t ld) { return ld >= 4 ? cdiv64(rows, bn_chunk_rows(ld)) * 2 * ld : 0; }

extern "C" int vs_bn_batch_stats(const float* x, int64_t rows, int C, int64_t ld, const float* gamma, const float* beta, float eps,
                                 float momentum, float* running_mean, float* running_var, double* partial, float* scale, float* shift,
                                 void* stream) {
  VS_REQUIRE(x && gamma && beta && partial && scale && shift && rows > 0 && C > 0 && ld >= C && (ld & 3) == 0);
  VS_REQUIRE((((uintptr_t)x) & 15) == 0);
  const int nchunk = (int)cdiv64(rows, bn_chunk_rows(ld));
  launch_bn_partial(x, rows, ld, partial, (hipStream_t)stream);
  hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)C), dim3(64), 0, (hipStream_t)stream, partial, nchunk, rows, C, ld,
                     gamma, beta, eps, momentum, running_mean, running_var, scale, shift, (const double*)nullptr);
  return vs_launch_status();
}

extern "C" int vs_bn_partial_sums(const float* x, int64_t rows, int C, int64_t ld, double* partial, double* sums, void* stream) {
  VS_REQUIRE(x && partial && sums && rows > 0 && C > 0 && ld >= C && (ld & 3) == 0);
  VS_REQUIRE((((uintptr_t)x) & 15) == 0);
  const int nchunk = (int)cdiv64(rows, bn_chunk_rows(ld));
  launch_bn_partial(x, rows, ld, partial, (hipStream_t)stream);
  hipLaunchKernelGGL(bn_reduce_kernel, dim3((unsigned)ld), dim3(64), 0, (hipStream_t)stream, partial, nchunk, rows, ld, sums);
  return vs_launch_status();
}

extern "C" int vs_bn_finish_sums(const double* sums, int C, int64_t ld, const float* gamma, const float* beta, float eps, float momentum,
                                 float* running_mean, float* running_var, float* scale, float* shift, void* stream) {
  VS_REQUIRE(sums && gamma && beta && scale && shift && C > 0 && ld >= C);
  hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)C), dim3(64), 0, (hipStream_t)stream, sums, 1, (int64_t)0, C, ld,
                     gamma, beta, eps, momentum, running_mean, running_var, scale, shift, sums + 2 * ld);
  return vs_launch_status();
}

extern "C" int vs_scale_shift_act(const float* x, int64_t rows, int C, int64_t ld, const float* scale, const float* shift, int act,
                                  const float* add, int64_t add_ld, float* out, int64_t out_ld, void* stream) {
  VS_REQUIRE(x && scale && shift && out && rows > 0 && C > 0 && (C & 3) == 0 && ld >= C && out_ld >= C && (ld & 3) == 0 && (out_ld & 3) == 0);
  VS_REQUIRE(!add || (add_ld >= C && (add_ld & 3) == 0));
  hipLaunchKernelGGL(scale_shift_act_kernel, dim3(grid_for(rows * (C >> 2), 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, x, rows, C,
                     ld, scale, shift, act, add, add_ld, out, out_ld);
  return vs_launch_status();
}

extern "C" int vs_check_finite(const float* x, int64_t n, int* flag, void* stream) {
  VS_REQUIRE(x && flag && n > 0);
  int64_t g = cdiv64(n, 256 * 8);
  g = g > 1024 ? 1024 : (g < 1 ? 1 : g);
  hipLaunchKernelGGL(check_finite_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, n, flag);
  return vs_launch_status();
}

extern "C" int vs_absmax(const float* x, int64_t n, unsigned* bits, void* stream) {
  VS_REQUIRE(x && bits && n > 0);
  int64_t g = cdiv64(n, 256 * 8);
  g = g > 2048 ? 2048 : (g < 1 ? 1 : g);
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, n, bits);
  return vs_launch_status();
}
