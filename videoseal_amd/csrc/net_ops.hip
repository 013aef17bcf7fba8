// Bandwidth-side network kernels of the VideoSeal path (everything that is not a dense conv/GEMM):
// LayerNorm(+act), depthwise 7x7 fused with LayerNorm, GRN statistics, bilinear x2 of the skip concat,
// message latent + broadcast, the final 1x1+tanh, and the pooled linear head.
// All activations are NHWC fp32 with a channel stride `ld` (multiple of 4, pad lanes kept at zero).
#include "vs_common.h"

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

// ---------------------------------------------------------------------------------------------------
// Depthwise 7x7 (pad 3) + bias, then LayerNorm over C, per pixel (convnext.py:43-46).
// A work item = (strip of 4 consecutive x, group of 4 channels): 70 float4 loads feed 16 outputs x 4 channels,
// results go to LDS [pixel][channel]; then one wave per pixel does the LayerNorm and the coalesced store.
__global__ __launch_bounds__(256) void dwconv7_ln_kernel(const float* __restrict__ x, int B, int H, int W, int C, int64_t ld,
                                                         const float* __restrict__ wdw, const float* __restrict__ bdw,
                                                         const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                         float eps, float* __restrict__ out, int64_t out_ld, int NS,
                                                         int spr, int64_t nstrips) {
  extern __shared__ __attribute__((aligned(16))) float conv[];   // [NS*4][ld]
  const int C4 = (int)(ld >> 2);
  const int64_t s0 = (int64_t)blockIdx.x * NS;
  const int items = NS * C4;
  for (int it = threadIdx.x; it < items; it += blockDim.x) {
    const int sl = it / C4, cg = it - sl * C4;
    const int64_t sidx = s0 + sl;
    if (sidx >= nstrips) break;
    const int c = cg * 4;
    const int xs = (int)(sidx % spr);
    const int64_t t = sidx / spr;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    const int x0 = xs * 4;
    f32x4 acc[4];
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bdw + c);
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
        in[j] = (ix >= 0 && ix < W) ? *reinterpret_cast<const f32x4*>(base + ((int64_t)iy * W + ix) * ld)
                                    : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wdw + (int64_t)(ky * 7 + kx) * ld + c);
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p] += in[p + kx] * wv;
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(conv + (int64_t)(sl * 4 + p) * ld + c) = acc[p];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int p = wv; p < NS * 4; p += nw) {
    const int64_t sidx = s0 + (p >> 2);
    if (sidx >= nstrips) break;
    const int xs = (int)(sidx % spr);
    const int64_t t = sidx / spr;
    const int px = xs * 4 + (p & 3);
    if (px >= W) continue;
    const float* cr = conv + (int64_t)p * ld;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += cr[c];
    const float mean = wave_sum(s) / (float)C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float dlt = cr[c] - mean;
      v += dlt * dlt;
    }
    const float den = sqrtf(wave_sum(v) / (float)C + eps);
    float* orow = out + (t * W + px) * out_ld;
    for (int c = lane; c < (int)out_ld; c += 64) orow[c] = c < C ? lnw[c] * ((cr[c] - mean) / den) + lnb[c] : 0.f;
  }
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
__global__ __launch_bounds__(1024) void grn_finish_kernel(const float* __restrict__ partial, int nchunk, int B, int C,
                                                          const float* __restrict__ gamma, float* __restrict__ scale,
                                                          int64_t sld, int frame_major) {
  __shared__ float grp[4][256];
  __shared__ float red[256];
  const int b = blockIdx.x;
  const int cl = threadIdx.x & 255, kg = threadIdx.x >> 8;
  const int kper = (nchunk + 3) / 4;
  const int k0 = kg * kper, k1 = min(nchunk, k0 + kper);
  float local = 0.f;
  for (int cb = 0; cb < C; cb += 256) {
    const int c = cb + cl;
    float s = 0.f;
    if (c < C) {
      if (frame_major) for (int k = k0; k < k1; ++k) s += partial[((int64_t)b * nchunk + k) * C + c];
      else for (int k = k0; k < k1; ++k) s += partial[((int64_t)k * B + b) * C + c];
    }
    grp[kg][cl] = s;
    __syncthreads();
    if (kg == 0 && c < C) {
      const float gx = sqrtf(((grp[0][cl] + grp[1][cl]) + grp[2][cl]) + grp[3][cl]);
      scale[(int64_t)b * sld + c] = gx;
      local += gx;
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
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int r = 0; r < HW; ++r) s += xb[(int64_t)r * ld + c];
    pooled[c] = s / (float)HW;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int n = blockIdx.y * 4 + wv; n < N; n += 4 * gridDim.y) {
    const float* wr = w + (int64_t)n * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += pooled[c] * wr[c];
    s = wave_sum(s);
    if (lane == 0) out[(int64_t)b * N + n] = s + bias[n];
  }
}

// ---------------------------------------------------------------------------------------------------
// BatchNorm2d with BATCH statistics (the U-Net's ResnetBlocks under model.train(): unet.py:26,30 -> nn.BatchNorm2d).
// Two deterministic stages over an NHWC tensor [rows][ld]:
//   bn_partial_kernel : every block sums x and x^2 of BN_ROWS rows per channel (fp64 accumulators, fixed order)
//   bn_finish_kernel  : mean / biased variance -> scale = gamma/sqrt(var+eps), shift = beta - mean*scale, and the
//                       running-statistics update of nn.BatchNorm2d (momentum m, UNBIASED variance) in place.
constexpr int BN_ROWS = 2048;
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, int64_t rows, int64_t ld,
                                                         double* __restrict__ partial) {
  __shared__ double red[2][256][4];
  const int C4 = (int)(ld >> 2);
  const int RL = 256 / C4 > 0 ? 256 / C4 : 1;          // row lanes per channel group
  const int g = threadIdx.x % C4, rl = threadIdx.x / C4;
  const int64_t r0 = (int64_t)blockIdx.x * BN_ROWS, r1 = r0 + BN_ROWS < rows ? r0 + BN_ROWS : rows;
  for (int gb = 0; gb < C4; gb += 256) {                // C4 > 256 only for very wide tensors
    const int gg = gb + g;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (rl < RL && gg < C4)
      for (int64_t r = r0 + rl; r < r1; r += RL) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ld + 4 * gg);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += (double)v[e]; q[e] += (double)v[e] * (double)v[e]; }
      }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][threadIdx.x][e] = s[e]; red[1][threadIdx.x][e] = q[e]; }
    __syncthreads();
    if (rl == 0 && gg < C4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double ss = 0, qq = 0;
        for (int j = 0; j < RL && j * C4 + g < 256; ++j) { ss += red[0][j * C4 + g][e]; qq += red[1][j * C4 + g][e]; }
        partial[((int64_t)blockIdx.x * 2 + 0) * ld + 4 * gg + e] = ss;
        partial[((int64_t)blockIdx.x * 2 + 1) * ld + 4 * gg + e] = qq;
      }
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void bn_finish_kernel(const double* __restrict__ partial, int nchunk, int64_t rows, int C, int64_t ld,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                        float momentum, float* __restrict__ running_mean,
                                                        float* __restrict__ running_var, float* __restrict__ scale,
                                                        float* __restrict__ shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s = 0, q = 0;
  for (int k = 0; k < nchunk; ++k) { s += partial[((int64_t)k * 2 + 0) * ld + c]; q += partial[((int64_t)k * 2 + 1) * ld + c]; }
  const double n = (double)rows;
  const double mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0) var = 0;
  const float sc = gamma[c] * (float)(1.0 / sqrt(var + (double)eps));
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(n > 1 ? var * n / (n - 1) : var);
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
  hipLaunchKernelGGL(layernorm_act_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, C, ld,
                     w, b, eps, act, out, out_ld);
  return vs_launch_status();
}

extern "C" int vs_dwconv7_ln(const float* x, int B, int H, int W, int C, int64_t ld, const float* wdw, const float* bdw,
                             const float* lnw, const float* lnb, float eps, float* out, int64_t out_ld, void* stream) {
  VS_REQUIRE(x && wdw && bdw && lnw && lnb && out && B > 0 && H > 0 && W > 0 && C > 0);
  VS_REQUIRE(ld % 4 == 0 && ld >= C && out_ld >= C);
  const int C4 = (int)(ld / 4);
  int NS = 256 / C4;
  if (NS < 1) NS = 1;
  const int spr = (W + 3) / 4;
  const int64_t nstrips = (int64_t)B * H * spr;
  const size_t smem = (size_t)NS * 4 * ld * sizeof(float);
  if (smem > 160 * 1024) return VS_ERR_UNSUPPORTED;
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)dwconv7_ln_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(dwconv7_ln_kernel, dim3((unsigned)cdiv64(nstrips, NS)), dim3(256), smem, (hipStream_t)stream, x, B, H, W,
                     C, ld, wdw, bdw, lnw, lnb, eps, out, out_ld, NS, spr, nstrips);
  return vs_launch_status();
}

extern "C" int vs_grn_scale(const float* h, int B, int HW, int C, int64_t ld, const float* gamma, float* partial,
                            float* scale, void* stream) {
  VS_REQUIRE(h && gamma && partial && scale && B > 0 && HW > 0 && C > 0 && ld >= C);
  const int nchunk = (HW + GRN_ROWS - 1) / GRN_ROWS;
  hipLaunchKernelGGL(grn_partial_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)nchunk, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, h, HW, C, ld, partial, B);
  hipLaunchKernelGGL(grn_finish_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, partial, nchunk, B, C, gamma,
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
  hipLaunchKernelGGL(grn_finish_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, partial, HW / 32, B, C, gamma,
                     scale, scale_ld, 1);
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

extern "C" int64_t vs_bn_partial_doubles(int64_t rows, int64_t ld) { return cdiv64(rows, BN_ROWS) * 2 * ld; }

extern "C" int vs_bn_batch_stats(const float* x, int64_t rows, int C, int64_t ld, const float* gamma, const float* beta, float eps,
                                 float momentum, float* running_mean, float* running_var, double* partial, float* scale, float* shift,
                                 void* stream) {
  VS_REQUIRE(x && gamma && beta && partial && scale && shift && rows > 0 && C > 0 && ld >= C && (ld & 3) == 0);
  VS_REQUIRE((((uintptr_t)x) & 15) == 0);
  const int nchunk = (int)cdiv64(rows, BN_ROWS);
  hipLaunchKernelGGL(bn_partial_kernel, dim3((unsigned)nchunk), dim3(256), 0, (hipStream_t)stream, x, rows, ld, partial);
  hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial, nchunk, rows, C, ld,
                     gamma, beta, eps, momentum, running_mean, running_var, scale, shift);
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
