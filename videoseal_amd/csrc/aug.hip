// Augmentation kernels of the embed -> augment -> extract path (reference: videoseal/augmentation/{valuemetric,geometric}.py
// + utils/image.py).  All operate on NCHW fp32 frames resident in HBM; they are bandwidth-bound by construction.
//   colour      : brightness / contrast / saturation / hue / grayscale  (torchvision _functional_tensor semantics)
//   geometric   : hflip, crop, (anti-aliased) bilinear resize
//   filters     : separable Gaussian blur (reflect pad), median-of-row-medians k x k (zero pad, utils/image.py:80-83)
//   JPEG        : libjpeg(-turbo) baseline 4:2:0 encode + decode emulated bit-exactly in integer arithmetic
//                 (fixed-point colour transform, h2v2 box down-sampling with alternating bias, jpeg_fdct_islow,
//                 IJG quantisation, jpeg_idct_islow, h2v2 fancy up-sampling, fixed-point YCC->RGB); entropy coding is
//                 lossless and skipped.  Pinned against Pillow in tests (oracle/jpeg_ref.py is the numpy restatement).
#include "vs_common.h"
#include "resize_taps.h"
#include "resize_stream.h"

namespace {

// ------------------------------------------------------------------------------------------------ colour ops
enum { OP_BRIGHTNESS = 0, OP_CONTRAST = 1, OP_SATURATION = 2, OP_HUE = 3, OP_GRAYSCALE = 4 };

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
// The pixel arithmetic of the colour ops is compiled WITHOUT fma contraction (round 5): torchvision's tensor ops round every product before the
// add, and left to the compiler `f * r + (1 - f) * gray` is fused around EITHER product depending on the surrounding code -- the single-op kernel
// and the fused run of ops (color_chain_kernel, crop_resize_color_kernel) disagreed in the last bit.  One definition, one rounding sequence.
#pragma clang fp contract(off)
__device__ __forceinline__ float tv_gray(float r, float g, float b) { return 0.2989f * r + 0.587f * g + 0.114f * b; }

__device__ __forceinline__ void hue_shift(float& r, float& g, float& b, float factor) {
  // torchvision _rgb2hsv -> h = (h + factor) % 1 -> _hsv2rgb
  const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
  const bool eqc = maxc == minc;
  const float cr = maxc - minc;
  const float s = cr / (eqc ? 1.f : maxc);
  const float div = eqc ? 1.f : cr;
  const float rc = (maxc - r) / div, gc = (maxc - g) / div, bc = (maxc - b) / div;
  const float hr = (maxc == r) ? (bc - gc) : 0.f;
  const float hg = ((maxc == g) && (maxc != r)) ? (2.0f + rc - bc) : 0.f;
  const float hb = ((maxc != g) && (maxc != r)) ? (4.0f + gc - rc) : 0.f;
  float h = fmodf((hr + hg + hb) / 6.0f + 1.0f, 1.0f);
  const float v = maxc;
  h = h + factor;
  h = h - floorf(h);                       // python-style % 1.0
  const float h6 = h * 6.0f;
  const float fi = floorf(h6);
  const float f = h6 - fi;
  int i = (int)fi;
  i = ((i % 6) + 6) % 6;
  const float p = clamp01(v * (1.0f - s)), q = clamp01(v * (1.0f - s * f)), t = clamp01(v * (1.0f - s * (1.0f - f)));
  switch (i) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

// one colour op on one pixel (torchvision _functional_tensor semantics); `mean` = per-frame mean of the gray image for OP_CONTRAST
__device__ __forceinline__ void apply_color(const int op, const float factor, const float mean, float& r, float& g, float& b) {
  switch (op) {
    case OP_BRIGHTNESS: r = clamp01(factor * r); g = clamp01(factor * g); b = clamp01(factor * b); break;   // _blend(x, 0, f)
    case OP_CONTRAST: {
      const float m = (1.0f - factor) * mean;
      r = clamp01(factor * r + m); g = clamp01(factor * g + m); b = clamp01(factor * b + m);
    } break;
    case OP_SATURATION: {
      const float m = (1.0f - factor) * tv_gray(r, g, b);
      r = clamp01(factor * r + m); g = clamp01(factor * g + m); b = clamp01(factor * b + m);
    } break;
    case OP_HUE: hue_shift(r, g, b, factor); break;
    default: { const float y = 0.299f * r + 0.587f * g + 0.114f * b; r = g = b = y; } break;   // valuemetric.py:205-206
  }
}

#pragma clang fp contract(fast)

__global__ __launch_bounds__(256) void color_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t plane, int op,
                                                    float factor, const float* __restrict__ means) {
  const int f = blockIdx.y;
  const float* s = src + (int64_t)f * 3 * plane;
  float* d = dst + (int64_t)f * 3 * plane;
  const float mean = op == OP_CONTRAST ? means[f] : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (int64_t)gridDim.x * 256) {
    float r = s[i], g = s[plane + i], b = s[2 * plane + i];
    apply_color(op, factor, mean, r, g, b);
    d[i] = r; d[plane + i] = g; d[2 * plane + i] = b;
  }
}

// A run of colour ops in ONE pass over the frames (round 5: the validation chains of augmentation/__init__.py:107-123 apply Brightness ->
// Contrast -> Saturation -> Hue back to back, each a full read + write of the clip).  Every op is the expression of color_kernel, applied
// in order on the pixel's registers -> bit-identical to the separate launches.  Contrast needs the mean of ITS input: it may only be the
// first op of a run (`means` = gray mean of src); the host cuts longer sequences there.
struct ColorChain { int n; int op[6]; float factor[6]; };

__global__ __launch_bounds__(256) void color_chain_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t plane, ColorChain cc,
                                                          const float* __restrict__ means) {
  const int f = blockIdx.y;
  const float* s = src + (int64_t)f * 3 * plane;
  float* d = dst + (int64_t)f * 3 * plane;
  const float mean = (cc.n > 0 && cc.op[0] == OP_CONTRAST) ? means[f] : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (int64_t)gridDim.x * 256) {
    float r = s[i], g = s[plane + i], b = s[2 * plane + i];
    for (int k = 0; k < cc.n; ++k) apply_color(cc.op[k], cc.factor[k], mean, r, g, b);
    d[i] = r; d[plane + i] = g; d[2 * plane + i] = b;
  }
}

// per-frame mean of the torchvision gray image, deterministic two-stage reduction
__global__ __launch_bounds__(256) void gray_partial_kernel(const float* __restrict__ src, int64_t plane, float* __restrict__ partial) {
  __shared__ float red[256];
  const int f = blockIdx.y;
  const float* s = src + (int64_t)f * 3 * plane;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (int64_t)gridDim.x * 256)
    acc += tv_gray(s[i], s[plane + i], s[2 * plane + i]);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(int64_t)f * gridDim.x + blockIdx.x] = red[0];
}
// one wave per frame: lane l adds the partial sums l, l + 64, ... in ascending order, then a fixed butterfly -> deterministic.  (Until round 5
// ONE thread per frame walked the up to 256 partial sums as a chain of dependent loads: 17 us of the 38 us contrast op on 16 frames.)
__global__ __launch_bounds__(64) void gray_finish_kernel(const float* __restrict__ partial, int nblk, int64_t plane, float* __restrict__ means) {
  const int f = blockIdx.x;
  float s = 0.f;
  for (int k = threadIdx.x; k < nblk; k += 64) s += partial[(int64_t)f * nblk + k];
  s = wave_sum(s);
  if (threadIdx.x == 0) means[f] = s / (float)plane;
}

// ------------------------------------------------------------------------------------------------ mask blend / additive noise
// augmenter.py:175: imgs_aug = imgs_w * m + imgs * (1 - m), m = [F][1][H][W] broadcast over the C planes of a frame
__global__ __launch_bounds__(256) void mask_blend_kernel(const float* __restrict__ iw, const float* __restrict__ im,
                                                         const float* __restrict__ m, float* __restrict__ dst, int Cc, int64_t plane) {
  const int64_t pl = blockIdx.y;                 // frame * Cc + channel
  const float* mk = m + (pl / Cc) * plane;
  const int64_t o = pl * plane;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (int64_t)gridDim.x * 256) {
    const float w = mk[i];
    dst[o + i] = __fadd_rn(__fmul_rn(iw[o + i], w), __fmul_rn(im[o + i], __fsub_rn(1.f, w)));   // no fma contraction: ATen rounds each product
  }
}
// valuemetric.py:188-191: image + noise * std (the noise itself is the caller's torch.randn_like draw)
__global__ __launch_bounds__(256) void add_scaled_kernel(const float* __restrict__ x, const float* __restrict__ nz, float std,
                                                         float* __restrict__ dst, int64_t n) {
#pragma clang fp contract(off)      // torch rounds the product before the add; HIP's __fmul_rn is a plain '*' that hipcc would contract into an fma
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float t = nz[i] * std;
    dst[i] = x[i] + t;
  }
}

// video.py:411-486 (WindowAveraging): out[i] = (1 - alpha) * f[i] + alpha * mean(f[max(0, i - hw) .. min(F, i + hw + 1))), hw = window / 2.
// ATen's mean over the (few) frames of the window is the sum in frame order divided by the count; the blend rounds both products.
__global__ __launch_bounds__(256) void window_average_kernel(const float* __restrict__ src, float* __restrict__ dst, int F, int64_t fsz, int hw,
                                                             float alpha) {
#pragma clang fp contract(off)
  const int64_t i = blockIdx.y;
  const int a = (int)(i - hw < 0 ? 0 : i - hw), b = (int)(i + hw + 1 > F ? F : i + hw + 1);
  const float inv = (float)(b - a);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < fsz; e += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int k = a; k < b; ++k) s += src[(int64_t)k * fsz + e];
    const float avg = s / inv;
    const float t0 = (1.f - alpha) * src[i * fsz + e];
    const float t1 = alpha * avg;
    dst[i * fsz + e] = t0 + t1;
  }
}
// video.py:507-526 (DropFrame) / 283-313 (SpeedChange): dst[f] = src[idx[f]], whole frames of `fsz` floats (fsz % 4 == 0 fast path)
__global__ __launch_bounds__(256) void gather_frames_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                            float* __restrict__ dst, int64_t fsz) {
  const int64_t f = blockIdx.y;
  const float* s = src + (int64_t)idx[f] * fsz;
  float* d = dst + f * fsz;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < fsz; i += (int64_t)gridDim.x * 256) d[i] = s[i];
}

// ------------------------------------------------------------------------------------------------ geometric
__global__ __launch_bounds__(256) void crop_flip_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int i0,
                                                        int j0, int h, int w, int flip) {
  const int64_t pl = blockIdx.y;
  const float* s = src + pl * (int64_t)H * W;
  float* d = dst + pl * (int64_t)h * w;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)h * w; i += (int64_t)gridDim.x * 256) {
    const int y = (int)(i / w), x = (int)(i - (int64_t)y * w);
    const int sy = y + i0, sx = (flip ? (w - 1 - x) : x) + j0;
    float v = 0.f;                                           // torchvision crop pads with zeros outside the image
    if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = s[(int64_t)sy * W + sx];
    d[i] = v;
  }
}

using namespace vs_taps;       // ATen-compatible taps (resize_taps.h)
__global__ __launch_bounds__(256) void resize_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int oh,
                                                          int ow, int antialias) {
  const int ox = blockIdx.x * 32 + (threadIdx.x & 31), oy = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (ox >= ow || oy >= oh) return;
  const float* s = src + (int64_t)blockIdx.z * H * W;
  const Taps ty = make_taps(oy, H, oh, antialias), tx = make_taps(ox, W, ow, antialias);
  float acc = 0.f;
  constexpr int MAXW = 12;
  if (tx.n <= MAXW) {        // the column weights once per output (they were re-evaluated -- a triangle and a division each -- for every row of the window)
    float wx[MAXW];
#pragma unroll
    for (int j = 0; j < MAXW; ++j) wx[j] = j < tx.n ? tap_w(tx, j) : 0.f;
    for (int jy = 0; jy < ty.n; ++jy) {
      const float* row = s + (int64_t)(ty.lo + jy) * W + tx.lo;
      float r = 0.f;
#pragma unroll
      for (int jx = 0; jx < MAXW; ++jx)
        if (jx < tx.n) r = __builtin_fmaf(wx[jx], row[jx], r);
      acc = __builtin_fmaf(tap_w(ty, jy), r, acc);
    }
  } else {
    for (int jy = 0; jy < ty.n; ++jy) {
      const float* row = s + (int64_t)(ty.lo + jy) * W + tx.lo;
      float r = 0.f;
      for (int jx = 0; jx < tx.n; ++jx) r = __builtin_fmaf(tap_w(tx, jx), row[jx], r);
      acc = __builtin_fmaf(tap_w(ty, jy), r, acc);
    }
  }
  dst[((int64_t)blockIdx.z * oh + oy) * ow + ox] = acc;
}

// Crop -> (anti-aliased) bilinear Resize -> a run of pointwise colour ops in ONE kernel (round 5; Crop / Resize / Brightness of the validation
// chains).  The crop is an index offset of the resize's source window, so the cropped clip is never written; the 32 x 8 output tile's source
// window of all three planes is staged once in LDS by coalesced loads (resize_nchw_kernel reads every tap from global memory: L1-bound at 0.14
// of the HBM rate), and each output pixel evaluates the SAME taps in the SAME order as resize_nchw_kernel on the cropped tensor -- horizontal
// sums first, then the vertical one -- followed by apply_color on its three channel values: bit-identical to the separate launches.
constexpr int CRC_MAXW = 12;      // taps per axis whose weights are tabulated per tile (anti-aliased down-scaling up to 5.5 x)

__global__ __launch_bounds__(256) void crop_resize_color_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int i0, int j0,
                                                                int ch, int cw, int oh, int ow, int antialias, int win_h, int win_w, ColorChain cc) {
  extern __shared__ __attribute__((aligned(16))) float crc_smem[];
  // [40][CRC_MAXW] tap weights of the tile's 32 columns and 8 rows | [40] first source index | [40] tap count | [3][win_h][win_w] source window
  float* const wtab = crc_smem;
  int* const tlo = reinterpret_cast<int*>(crc_smem + 40 * CRC_MAXW);
  int* const tn = tlo + 40;
  float* const crc_win = crc_smem + 40 * CRC_MAXW + 80;
  const int f = blockIdx.z;
  const int ox0 = blockIdx.x * 32, oy0 = blockIdx.y * 8;
  // The tap weights once per TILE: a weight is a triangle and an IEEE division (tap_w), and evaluated per output pixel -- 24 of them, after the
  // per-pixel tabulation -- they made the kernel instruction-bound (81 us for the configs[2] clip, ~1000 instructions per wave).  Threads 0-31
  // tabulate the tile's columns, 32-39 its rows; every pixel then reads its two weight vectors from LDS.  Same values, same order of sums.
  if (threadIdx.x < 40) {
    const bool col = threadIdx.x < 32;
    const int o = col ? min(ox0 + (int)threadIdx.x, ow - 1) : min(oy0 + (int)threadIdx.x - 32, oh - 1);
    const Taps t = col ? make_taps(o, cw, ow, antialias) : make_taps(o, ch, oh, antialias);
    tlo[threadIdx.x] = t.lo;
    tn[threadIdx.x] = t.n;
    for (int q = 0; q < CRC_MAXW; ++q) wtab[threadIdx.x * CRC_MAXW + q] = q < t.n ? tap_w(t, q) : 0.f;
  }
  // source window of the tile (in crop coordinates): first tap of the first row / column .. last tap of the last row / column of the tile
  int ylo, yn, xlo, xn, t0, t1;
  tap_range(oy0, ch, oh, antialias, ylo, yn);
  tap_range(min(oy0 + 7, oh - 1), ch, oh, antialias, t0, t1);
  const int yhi = t0 + t1;                                            // one past the last source row
  tap_range(ox0, cw, ow, antialias, xlo, xn);
  tap_range(min(ox0 + 31, ow - 1), cw, ow, antialias, t0, t1);
  const int xhi = t0 + t1;
  const int wh = yhi - ylo, ww = xhi - xlo;                           // <= win_h, win_w (checked by the launcher's bound)
  const float* sf = src + (int64_t)f * 3 * H * W;
  // rows of the window: 64 lanes along x (coalesced), the four waves take rows (c, y) round-robin, EIGHT rows per batch: the loads of a batch
  // are unconditional (clamped addresses) and issued back to back, then stored -- a load -> store loop per row is a chain of ~15 dependent
  // round trips per workgroup
  {
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nrow = 3 * wh;
    for (int xc = 0; xc < ww; xc += 64) {
      const int x = xc + l;
      const int xl = x < ww ? x : ww - 1;
      for (int r0 = wv; r0 < nrow; r0 += 32) {
        float v[8];
        int off[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int rowi = min(r0 + 4 * u, nrow - 1);
          const int c = rowi / wh, y = rowi - c * wh;
          off[u] = (c * win_h + y) * win_w;
          v[u] = sf[((int64_t)c * H + (i0 + ylo + y)) * W + (j0 + xlo + xl)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (x < ww && r0 + 4 * u < nrow) crc_win[off[u] + x] = v[u];
      }
    }
  }
  __syncthreads();
  const int cx = threadIdx.x & 31, cy = threadIdx.x >> 5;
  const int ox = ox0 + cx, oy = oy0 + cy;
  if (ox >= ow || oy >= oh) return;
  const int nx = tn[cx], ny = tn[32 + cy], lox = tlo[cx] - xlo, loy = tlo[32 + cy] - ylo;
  float v[3];
  if (nx <= CRC_MAXW && ny <= CRC_MAXW) {
    // weights straight from the tile's table (one LDS read per tap and axis, shared by the three planes): register arrays indexed by the
    // per-pixel tap count needed 12 predicated iterations per row and a 11-deep select chain per weight (~500 instructions per pixel)
    const float* const wxp = wtab + cx * CRC_MAXW;
    const float* const wyp = wtab + (32 + cy) * CRC_MAXW;
    const float* const base = crc_win + loy * win_w + lox;
    const int cs = win_h * win_w;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int jy = 0; jy < ny; ++jy) {
      const float* row = base + jy * win_w;
      float r0 = 0.f, r1 = 0.f, r2 = 0.f;
      for (int jx = 0; jx < nx; ++jx) {
        const float w = wxp[jx];
        r0 = __builtin_fmaf(w, row[jx], r0);
        r1 = __builtin_fmaf(w, row[cs + jx], r1);
        r2 = __builtin_fmaf(w, row[2 * cs + jx], r2);
      }
      const float wyj = wyp[jy];
      a0 = __builtin_fmaf(wyj, r0, a0);
      a1 = __builtin_fmaf(wyj, r1, a1);
      a2 = __builtin_fmaf(wyj, r2, a2);
    }
    v[0] = a0; v[1] = a1; v[2] = a2;
  } else {          // longer filters: the weights as resize_nchw_kernel evaluates them
    const Taps ty = make_taps(oy, ch, oh, antialias), tx = make_taps(ox, cw, ow, antialias);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float acc = 0.f;
      for (int jy = 0; jy < ty.n; ++jy) {
        const float* row = crc_win + (c * win_h + (loy + jy)) * win_w + lox;
        float r = 0.f;
        for (int jx = 0; jx < tx.n; ++jx) r = __builtin_fmaf(tap_w(tx, jx), row[jx], r);
        acc = __builtin_fmaf(tap_w(ty, jy), r, acc);
      }
      v[c] = acc;
    }
  }
  for (int k = 0; k < cc.n; ++k) apply_color(cc.op[k], cc.factor[k], 0.f, v[0], v[1], v[2]);
  float* df = dst + (int64_t)f * 3 * oh * ow + (int64_t)oy * ow + ox;
  df[0] = v[0]; df[(int64_t)oh * ow] = v[1]; df[2 * (int64_t)oh * ow] = v[2];
}

// Round 6: the same fused pass on the row-streaming resize of resize_stream.h (the kernel `vs_resize_pre` runs on): a workgroup walks the source
// rows of a strip once, filters every (source row, output column) pair once and applies the colour ops where the vertical sums complete.  The tile
// kernel above spends its time on per-tile set-up and ~64 scalar LDS reads per pixel (0.18 of the HBM rate on the configs[2] clip); the streaming
// form does 2.1 x fewer multiply-adds and reads the window with 16-byte loads.  Same taps in the same order -> the same bits
// (tests/test_gpu_aug.py::test_sequential_fuses_the_validation_chain..., test_crop_resize_color_forms_are_bit_identical).
struct CropResizeEpi {
  float* dst; int oh, ow; ColorChain cc;
  __device__ __forceinline__ void operator()(const int b, const int oy, const int ox, const float (&acc)[3]) const {
    float v0 = acc[0], v1 = acc[1], v2 = acc[2];
    for (int k = 0; k < cc.n; ++k) apply_color(cc.op[k], cc.factor[k], 0.f, v0, v1, v2);
    float* df = dst + (int64_t)b * 3 * oh * ow + (int64_t)oy * ow + ox;
    df[0] = v0; df[(int64_t)oh * ow] = v1; df[2 * (int64_t)oh * ow] = v2;
  }
};
template <int OW>
__global__ __launch_bounds__(256) void crop_resize_color_stream_kernel(const float* __restrict__ src, int H, int W, int i0, int j0, int ch, int cw,
                                                                       int oh, int ow, int antialias, CropResizeEpi epi, int strip) {
  extern __shared__ __attribute__((aligned(16))) float crs_smem[];
  vs_rs::resize_stream_body<OW>(crs_smem, src, H, W, i0, j0, ch, cw, oh, ow, antialias, strip, epi);
}

// ------------------------------------------------------------------------------------------------ filters
struct GaussK { float w[33]; int k; };
__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__global__ __launch_bounds__(256) void blur_pass_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, GaussK g,
                                                        int vertical) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= W || y >= H) return;
  const float* s = src + (int64_t)blockIdx.z * H * W;
  const int half = g.k >> 1;
  float acc = 0.f;
  for (int j = 0; j < g.k; ++j) {
    const int yy = vertical ? reflect_idx(y + j - half, H) : y;
    const int xx = vertical ? x : reflect_idx(x + j - half, W);
    acc += g.w[j] * s[(int64_t)yy * W + xx];
  }
  dst[((int64_t)blockIdx.z * H + y) * W + x] = acc;
}

// median of the k row-medians of the zero-padded k x k window (torch .median on an odd count = the middle element)
template <int K>
__global__ __launch_bounds__(256) void median_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= W || y >= H) return;
  const float* s = src + (int64_t)blockIdx.z * H * W;
  constexpr int half = K / 2;
  float rowmed[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    float v[K];
    const int yy = y + i - half;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int xx = x + j - half;
      v[j] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? s[(int64_t)yy * W + xx] : 0.f;
    }
#pragma unroll
    for (int a = 1; a < K; ++a)          // insertion sort (K <= 7)
#pragma unroll
      for (int b = a; b > 0; --b) {
        const float lo = fminf(v[b - 1], v[b]), hi = fmaxf(v[b - 1], v[b]);
        v[b - 1] = lo; v[b] = hi;
      }
    rowmed[i] = v[half];
  }
#pragma unroll
  for (int a = 1; a < K; ++a)
#pragma unroll
    for (int b = a; b > 0; --b) {
      const float lo = fminf(rowmed[b - 1], rowmed[b]), hi = fmaxf(rowmed[b - 1], rowmed[b]);
      rowmed[b - 1] = lo; rowmed[b] = hi;
    }
  dst[((int64_t)blockIdx.z * H + y) * W + x] = rowmed[half];
}

// ------------------------------------------------------------------------------------------------ JPEG (libjpeg islow, 4:2:0)
constexpr int CONST_BITS = 13, PASS1_BITS = 2;
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

__device__ __forceinline__ int px_u8(float v) {   // torchvision ToPILImage on float: clamp (JPEG.forward) then mul(255).byte() = truncation
  v = fminf(fmaxf(v, 0.f), 1.f);
  return (int)(v * 255.0f);
}

// stage 1: RGB -> Y (full res, padded to 16) and Cb, Cr (h2v2 box down-sampled, padded).  One thread per chroma sample.
__global__ __launch_bounds__(256) void jpeg_ycc_kernel(const float* __restrict__ src, int H, int W, int Hp, int Wp,
                                                       unsigned char* __restrict__ Y, unsigned char* __restrict__ Cb,
                                                       unsigned char* __restrict__ Cr) {
  const int cx = blockIdx.x * 32 + (threadIdx.x & 31), cy = blockIdx.y * 8 + (threadIdx.x >> 5);
  const int cw = Wp / 2, ch = Hp / 2;
  if (cx >= cw || cy >= ch) return;
  const int f = blockIdx.z;
  const int64_t plane = (int64_t)H * W;
  const float* s = src + (int64_t)f * 3 * plane;
  const int He = (H + 1) / 2;                // chroma rows backed by real pixels; below: replicate the last DOWNSAMPLED row
  const int cys = cy < He ? cy : He - 1;
  int sb = 0, sr = 0;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      // luma: pixel-domain edge replication (rows and columns)
      const int py = 2 * cy + dy, px = 2 * cx + dx;
      const int yy = py < H ? py : H - 1, xx = px < W ? px : W - 1;
      const int64_t o = (int64_t)yy * W + xx;
      const int R = px_u8(s[o]), G = px_u8(s[plane + o]), B = px_u8(s[2 * plane + o]);
      Y[((int64_t)f * Hp + py) * Wp + px] = (unsigned char)((19595 * R + 38470 * G + 7471 * B + 32768) >> 16);
      // chroma: rows from the (1-row replicated) real image, columns replicated
      const int py2 = 2 * cys + dy;
      const int yy2 = py2 < H ? py2 : H - 1;
      const int64_t o2 = (int64_t)yy2 * W + xx;
      const int R2 = px_u8(s[o2]), G2 = px_u8(s[plane + o2]), B2 = px_u8(s[2 * plane + o2]);
      sb += (-11059 * R2 - 21709 * G2 + 32768 * B2 + (128 << 16) + 32767) >> 16;
      sr += (32768 * R2 - 27439 * G2 - 5329 * B2 + (128 << 16) + 32767) >> 16;
    }
  const int bias = 1 + (cx & 1);             // alternating 1,2,1,2...
  Cb[((int64_t)f * ch + cy) * cw + cx] = (unsigned char)((sb + bias) >> 2);
  Cr[((int64_t)f * ch + cy) * cw + cx] = (unsigned char)((sr + bias) >> 2);
}

__device__ __forceinline__ void fdct8(int* d, int stride, bool first) {
  const int d0 = d[0], d1 = d[stride], d2 = d[2 * stride], d3 = d[3 * stride], d4 = d[4 * stride], d5 = d[5 * stride],
            d6 = d[6 * stride], d7 = d[7 * stride];
  int t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6, t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;
  const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  const int n = first ? CONST_BITS - PASS1_BITS : CONST_BITS + PASS1_BITS;
  d[0] = first ? ((t10 + t11) << PASS1_BITS) : descale(t10 + t11, PASS1_BITS);
  d[4 * stride] = first ? ((t10 - t11) << PASS1_BITS) : descale(t10 - t11, PASS1_BITS);
  int z1 = (t12 + t13) * FIX_0_541196100;
  d[2 * stride] = descale(z1 + t13 * FIX_0_765366865, n);
  d[6 * stride] = descale(z1 - t12 * FIX_1_847759065, n);
  z1 = t4 + t7; int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
  const int z5 = (z3 + z4) * FIX_1_175875602;
  t4 *= FIX_0_298631336; t5 *= FIX_2_053119869; t6 *= FIX_3_072711026; t7 *= FIX_1_501321110;
  z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
  z3 += z5; z4 += z5;
  d[7 * stride] = descale(t4 + z1 + z3, n); d[5 * stride] = descale(t5 + z2 + z4, n);
  d[3 * stride] = descale(t6 + z2 + z3, n); d[stride] = descale(t7 + z1 + z4, n);
}
__device__ __forceinline__ void idct8(int* c, int stride, bool first) {
  int z2 = c[2 * stride], z3 = c[6 * stride];
  int z1 = (z2 + z3) * FIX_0_541196100;
  int t2 = z1 - z3 * FIX_1_847759065, t3 = z1 + z2 * FIX_0_765366865;
  z2 = c[0]; z3 = c[4 * stride];
  int t0 = (z2 + z3) << CONST_BITS, t1 = (z2 - z3) << CONST_BITS;
  const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  t0 = c[7 * stride]; t1 = c[5 * stride]; t2 = c[3 * stride]; t3 = c[stride];
  z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; int z4 = t1 + t3;
  const int z5 = (z3 + z4) * FIX_1_175875602;
  t0 *= FIX_0_298631336; t1 *= FIX_2_053119869; t2 *= FIX_3_072711026; t3 *= FIX_1_501321110;
  z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
  z3 += z5; z4 += z5;
  t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
  const int n = first ? CONST_BITS - PASS1_BITS : CONST_BITS + PASS1_BITS + 3;
  c[0] = descale(t10 + t3, n); c[7 * stride] = descale(t10 - t3, n);
  c[stride] = descale(t11 + t2, n); c[6 * stride] = descale(t11 - t2, n);
  c[2 * stride] = descale(t12 + t1, n); c[5 * stride] = descale(t12 - t1, n);
  c[3 * stride] = descale(t13 + t0, n); c[4 * stride] = descale(t13 - t0, n);
}

struct QTab { unsigned char q[64]; };
// stage 2: per 8x8 block: level shift, forward DCT, quantise, de-quantise, inverse DCT, range limit -- in place.
// One thread per block (the whole block lives in registers / LDS-free private arrays with static indexing).
// One launch covers the three components (round 6: three launches before): blocks [0, nby) are luma blocks of the plane at `Y`, [nby, nby + 2 nbc)
// the blocks of the two chroma planes that follow it in the workspace (Cb, then Cr: each nbc blocks, row length Wp / 2).
struct QTab2 { QTab l, c; };
__global__ __launch_bounds__(64) void jpeg_block_kernel(unsigned char* __restrict__ Yp, unsigned char* __restrict__ Cbp, int Wp, int64_t nby,
                                                        int64_t nbc, QTab2 q2) {
  int64_t blk = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (blk >= nby + 2 * nbc) return;
  const bool luma = blk < nby;
  unsigned char* plane = Yp;
  if (!luma) { blk -= nby; plane = Cbp; Wp >>= 1; }           // (Cr follows Cb contiguously: block rows run on through both planes)
  const QTab& qt = luma ? q2.l : q2.c;
  const int bw = Wp / 8;
  const int64_t by = blk / bw;
  const int bx = (int)(blk - by * bw);
  unsigned char* p = plane + by * 8 * Wp + bx * 8;
  int d[64];
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    const uint2 v = *reinterpret_cast<const uint2*>(p + (int64_t)y * Wp);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      d[y * 8 + x] = (int)((v.x >> (8 * x)) & 255) - 128;
      d[y * 8 + 4 + x] = (int)((v.y >> (8 * x)) & 255) - 128;
    }
  }
#pragma unroll
  for (int y = 0; y < 8; ++y) fdct8(d + y * 8, 1, true);
#pragma unroll
  for (int x = 0; x < 8; ++x) fdct8(d + x, 8, false);
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const int q = qt.q[i], qv = q << 3;
    int t = d[i];
    const int a = t < 0 ? -t : t;
    const int c = (a + (qv >> 1)) / qv;
    d[i] = (t < 0 ? -c : c) * q;
  }
#pragma unroll
  for (int x = 0; x < 8; ++x) idct8(d + x, 8, true);
#pragma unroll
  for (int y = 0; y < 8; ++y) idct8(d + y * 8, 1, false);
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    uint2 v = {0u, 0u};
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int a = min(max(d[y * 8 + x] + 128, 0), 255), b = min(max(d[y * 8 + 4 + x] + 128, 0), 255);
      v.x |= (unsigned)a << (8 * x);
      v.y |= (unsigned)b << (8 * x);
    }
    *reinterpret_cast<uint2*>(p + (int64_t)y * Wp) = v;
  }
}

// stage 3: h2v2 fancy (triangle) up-sampling of the chroma planes + fixed-point YCbCr -> RGB, output float /255.
__device__ __forceinline__ int chroma_up(const unsigned char* __restrict__ C, int cw_alloc, int ch, int cw, int y, int x) {
  const int cy = y >> 1, cx = x >> 1;
  if (cw <= 2) return C[(int64_t)cy * cw_alloc + cx];            // libjpeg: fancy up-sampling only if downsampled_width > 2
  const int fy = (y & 1) ? min(cy + 1, ch - 1) : max(cy - 1, 0); // farther row (edge rows replicate)
  auto colsum = [&](int xx) { return 3 * (int)C[(int64_t)cy * cw_alloc + xx] + (int)C[(int64_t)fy * cw_alloc + xx]; };
  const int cs = colsum(cx);
  if ((x & 1) == 0) return cx == 0 ? ((cs * 4 + 8) >> 4) : ((3 * cs + colsum(cx - 1) + 8) >> 4);
  return cx == cw - 1 ? ((cs * 4 + 7) >> 4) : ((3 * cs + colsum(cx + 1) + 7) >> 4);
}
__global__ __launch_bounds__(256) void jpeg_rgb_kernel(const unsigned char* __restrict__ Y, const unsigned char* __restrict__ Cb,
                                                       const unsigned char* __restrict__ Cr, int H, int W, int Hp, int Wp,
                                                       float* __restrict__ dst) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= W || y >= H) return;
  const int f = blockIdx.z;
  const int cwa = Wp / 2, cha = Hp / 2, cw = (W + 1) / 2, ch = (H + 1) / 2;
  const int yv = Y[((int64_t)f * Hp + y) * Wp + x];
  const int cb = chroma_up(Cb + (int64_t)f * cha * cwa, cwa, ch, cw, y, x) - 128;
  const int cr = chroma_up(Cr + (int64_t)f * cha * cwa, cwa, ch, cw, y, x) - 128;
  const int R = yv + ((91881 * cr + 32768) >> 16);
  const int B = yv + ((116130 * cb + 32768) >> 16);
  const int G = yv + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
  const int64_t plane = (int64_t)H * W, o = (int64_t)y * W + x;
  float* d = dst + (int64_t)f * 3 * plane;
  d[o] = (float)min(max(R, 0), 255) / 255.0f;           // ToTensor: uint8 / 255
  d[plane + o] = (float)min(max(G, 0), 255) / 255.0f;
  d[2 * plane + o] = (float)min(max(B, 0), 255) / 255.0f;
}

// ---- round 6: the same three stages for frames that are whole MCUs (H % 16 == 0, W % 16 == 0: no edge replication anywhere) with 16-byte global
// accesses.  The scalar kernels above moved one pixel per lane: byte stores of Y at a two-byte stride, 4-byte loads / stores of the fp32 planes -- the
// round trip ran at 0.23 of 8 TB/s on the configs[2] clip.  Same integer expressions per pixel -> the same bytes (tests/test_gpu_aug.py: torch.equal
// with Pillow / libjpeg and with the scalar kernels).
// stage 1: thread = 4 x 2 pixels = 2 chroma samples: six float4 loads, two uint32 Y stores, one uint16 store per chroma plane
__global__ __launch_bounds__(256) void jpeg_ycc4_kernel(const float* __restrict__ src, int H, int W, unsigned char* __restrict__ Y,
                                                        unsigned char* __restrict__ Cb, unsigned char* __restrict__ Cr) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), j = blockIdx.y * 4 + (threadIdx.x >> 6);      // cell (4 px, 2 rows)
  if (4 * i >= W || 2 * j >= H) return;
  const int f = blockIdx.z;
  const int64_t plane = (int64_t)H * W;
  const float* s = src + (int64_t)f * 3 * plane + (int64_t)(2 * j) * W + 4 * i;
  f32x4 px[2][3];
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int c = 0; c < 3; ++c) px[dy][c] = *reinterpret_cast<const f32x4*>(s + c * plane + (int64_t)dy * W);
  unsigned yw[2] = {0u, 0u};
  int sb[2] = {0, 0}, sr[2] = {0, 0};
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int R = px_u8(px[dy][0][e]), G = px_u8(px[dy][1][e]), B = px_u8(px[dy][2][e]);
      yw[dy] |= (unsigned)((19595 * R + 38470 * G + 7471 * B + 32768) >> 16) << (8 * e);
      sb[e >> 1] += (-11059 * R - 21709 * G + 32768 * B + (128 << 16) + 32767) >> 16;
      sr[e >> 1] += (32768 * R - 27439 * G - 5329 * B + (128 << 16) + 32767) >> 16;
    }
  unsigned char* yp = Y + ((int64_t)f * H + 2 * j) * W + 4 * i;
  *reinterpret_cast<unsigned*>(yp) = yw[0];
  *reinterpret_cast<unsigned*>(yp + W) = yw[1];
  // chroma samples cx = 2 i (bias 1) and 2 i + 1 (bias 2): the alternating rounding of h2v2_downsample
  const int cw = W / 2;
  const int64_t co = ((int64_t)f * (H / 2) + j) * cw + 2 * i;
  *reinterpret_cast<unsigned short*>(Cb + co) = (unsigned short)(((sb[0] + 1) >> 2) | (((sb[1] + 2) >> 2) << 8));
  *reinterpret_cast<unsigned short*>(Cr + co) = (unsigned short)(((sr[0] + 1) >> 2) | (((sr[1] + 2) >> 2) << 8));
}

// stage 3: thread = 4 pixels of a row: one uint32 of Y, the chroma bytes through chroma_up (the same function as the scalar kernel), three float4 stores
__global__ __launch_bounds__(256) void jpeg_rgb4_kernel(const unsigned char* __restrict__ Y, const unsigned char* __restrict__ Cb,
                                                        const unsigned char* __restrict__ Cr, int H, int W, float* __restrict__ dst) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (4 * i >= W || y >= H) return;
  const int f = blockIdx.z;
  const int cw = W / 2, ch = H / 2;
  const unsigned yw = *reinterpret_cast<const unsigned*>(Y + ((int64_t)f * H + y) * W + 4 * i);
  const unsigned char* cbp = Cb + (int64_t)f * ch * cw;
  const unsigned char* crp = Cr + (int64_t)f * ch * cw;
  // the chroma samples of the four pixels: columns 2i - 1 .. 2i + 2 of the near and the farther chroma row, CLAMPED to the plane -- at the plane's
  // edges chroma_up's special cases (4 cs + 8, 4 cs + 7) are exactly its general expressions with the missing neighbour replaced by the sample itself.
  // Sixteen unconditional byte loads in one batch instead of eight calls with a branch around every neighbour (tools/isa_scan.py order: ten dependent
  // groups); W % 16 == 0 here, so the `cw <= 2` form of chroma_up never applies.  Same integers per pixel.
  const int cy = y >> 1, fy = (y & 1) ? min(cy + 1, ch - 1) : max(cy - 1, 0);
  int cs[2][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int col = min(max(2 * i - 1 + k, 0), cw - 1);
    cs[0][k] = 3 * (int)cbp[(int64_t)cy * cw + col] + (int)cbp[(int64_t)fy * cw + col];
    cs[1][k] = 3 * (int)crp[(int64_t)cy * cw + col] + (int)crp[(int64_t)fy * cw + col];
  }
  f32x4 o[3];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int yv = (int)((yw >> (8 * e)) & 255u);
    const int kc = 1 + (e >> 1), kn = (e & 1) ? kc + 1 : kc - 1, rnd = (e & 1) ? 7 : 8;      // own column, neighbour (left for even x, right for odd x)
    const int cb = ((3 * cs[0][kc] + cs[0][kn] + rnd) >> 4) - 128;
    const int cr = ((3 * cs[1][kc] + cs[1][kn] + rnd) >> 4) - 128;
    const int R = yv + ((91881 * cr + 32768) >> 16);
    const int B = yv + ((116130 * cb + 32768) >> 16);
    const int G = yv + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    o[0][e] = (float)min(max(R, 0), 255) / 255.0f;
    o[1][e] = (float)min(max(G, 0), 255) / 255.0f;
    o[2][e] = (float)min(max(B, 0), 255) / 255.0f;
  }
  const int64_t plane = (int64_t)H * W;
  float* d = dst + (int64_t)f * 3 * plane + (int64_t)y * W + 4 * i;
#pragma unroll
  for (int c = 0; c < 3; ++c) *reinterpret_cast<f32x4*>(d + c * plane) = o[c];
}

const int BASE_L[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                        18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const int BASE_C[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                        99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
QTab make_qtab(const int* base, int quality) {   // IJG quality scaling (jcparam.c jpeg_quality_scaling + jpeg_add_quant_table, baseline)
  quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
  const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
  QTab t;
  for (int i = 0; i < 64; ++i) {
    int v = (base[i] * scale + 50) / 100;
    v = v < 1 ? 1 : (v > 255 ? 255 : v);
    t.q[i] = (unsigned char)v;
  }
  return t;
}

inline unsigned gridx(int64_t n, int per = 256, int64_t cap = 4096) {
  int64_t g = cdiv64(n, per);
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---------------------------------------------------------------------------------------------------
// Rotate / Perspective (geometric.py:28-59, 127-183): torchvision builds a sampling grid and calls grid_sample(padding zeros,
// align_corners=False).  The grid of output pixel (ox, oy), restated from torchvision/_functional_tensor.py
//   affine (_gen_affine_grid):       base = (ox + 0.5 - ow/2, oy + 0.5 - oh/2, 1);  g = base . (theta^T / (0.5 w, 0.5 h))
//   perspective (_perspective_grid): base = (ox + 0.5, oy + 0.5, 1);  g = base . (theta1^T / (0.5 ow, 0.5 oh)) / (base . theta2^T) - 1
// then grid_sample (ATen GridSamplerKernel): ix = ((gx + 1) * W - 1) / 2; nearest = nearbyint (half to even); bilinear =
// nw, ne, sw, se taps weighted by the opposite areas, taps outside the image contribute zero.
struct WarpArgs { float t[8]; int kind, bilinear; };   // kind 0: t[0..5] = rescaled theta^T columns (see host), 1: perspective

__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int oh, int ow,
                                                   WarpArgs a) {
  const int ox = blockIdx.x * 32 + (threadIdx.x & 31), oy = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (ox >= ow || oy >= oh) return;
  const float* sp = src + (int64_t)blockIdx.z * H * W;
  float gx, gy;
  if (a.kind == 0) {
    const float bx = (float)ox + (0.5f - ow * 0.5f), by = (float)oy + (0.5f - oh * 0.5f);
    gx = (bx * a.t[0] + by * a.t[1]) + a.t[2];
    gy = (bx * a.t[3] + by * a.t[4]) + a.t[5];
  } else {
    const float bx = (float)ox + 0.5f, by = (float)oy + 0.5f;
    const float n1 = (bx * (a.t[0] / (0.5f * ow)) + by * (a.t[1] / (0.5f * ow))) + a.t[2] / (0.5f * ow);
    const float n2 = (bx * (a.t[3] / (0.5f * oh)) + by * (a.t[4] / (0.5f * oh))) + a.t[5] / (0.5f * oh);
    const float den = (bx * a.t[6] + by * a.t[7]) + 1.0f;
    gx = n1 / den - 1.0f;
    gy = n2 / den - 1.0f;
  }
  const float ix = ((gx + 1.f) * W - 1.f) / 2.f, iy = ((gy + 1.f) * H - 1.f) / 2.f;
  float v = 0.f;
  if (!a.bilinear) {
    const float rx = nearbyintf(ix), ry = nearbyintf(iy);
    if (rx >= 0.f && rx <= (float)(W - 1) && ry >= 0.f && ry <= (float)(H - 1)) v = sp[(int64_t)(int)ry * W + (int)rx];
  } else {
    const float x0 = floorf(ix), y0 = floorf(iy);
    const float x1 = x0 + 1.f, y1 = y0 + 1.f;
    const float nw = (x1 - ix) * (y1 - iy), ne = (ix - x0) * (y1 - iy), sw = (x1 - ix) * (iy - y0), se = (ix - x0) * (iy - y0);
    auto at = [&](float xf, float yf) -> float {
      return (xf >= 0.f && xf <= (float)(W - 1) && yf >= 0.f && yf <= (float)(H - 1)) ? sp[(int64_t)(int)yf * W + (int)xf] : 0.f;
    };
    v = at(x0, y0) * nw;
    v += at(x1, y0) * ne;
    v += at(x0, y1) * sw;
    v += at(x1, y1) * se;
  }
  dst[((int64_t)blockIdx.z * oh + oy) * ow + ox] = v;
}

}  // namespace

extern "C" int vs_aug_warp(const float* src, float* dst, int planes, int H, int W, int oh, int ow, int kind, const float* coeffs,
                           int bilinear, void* stream) {
  VS_REQUIRE(src && dst && coeffs && planes > 0 && H > 0 && W > 0 && oh > 0 && ow > 0 && (kind == 0 || kind == 1));
  WarpArgs a;
  for (int i = 0; i < 8; ++i) a.t[i] = i < (kind == 0 ? 6 : 8) ? coeffs[i] : 0.f;
  a.kind = kind; a.bilinear = bilinear;
  hipLaunchKernelGGL(warp_kernel, dim3((ow + 31) / 32, (oh + 7) / 8, planes), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, oh, ow, a);
  return vs_launch_status();
}



extern "C" int vs_aug_color(const float* src, float* dst, int F, int H, int W, int op, float factor, float* scratch, void* stream) {
  VS_REQUIRE(src && dst && F > 0 && H > 0 && W > 0 && op >= 0 && op <= 4);
  hipStream_t st = (hipStream_t)stream;
  const int64_t plane = (int64_t)H * W;
  const float* means = nullptr;
  if (op == OP_CONTRAST) {           // scratch: F * (nblk + 1) floats
    VS_REQUIRE(scratch);
    const int nblk = (int)gridx(plane, 256, 256);
    hipLaunchKernelGGL(gray_partial_kernel, dim3(nblk, F), dim3(256), 0, st, src, plane, scratch);
    hipLaunchKernelGGL(gray_finish_kernel, dim3(F), dim3(64), 0, st, scratch, nblk, plane, scratch + (int64_t)F * nblk);
    means = scratch + (int64_t)F * nblk;
  }
  hipLaunchKernelGGL(color_kernel, dim3(gridx(plane), F), dim3(256), 0, st, src, dst, plane, op, factor, means);
  return vs_launch_status();
}
// n colour ops (ops[k] in 0..4, OP_CONTRAST only at k = 0) in one pass; scratch as for vs_aug_color when ops[0] is OP_CONTRAST
extern "C" int vs_aug_color_chain(const float* src, float* dst, int F, int H, int W, int n, const int* ops, const float* factors, float* scratch,
                                  void* stream) {
  VS_REQUIRE(src && dst && F > 0 && H > 0 && W > 0 && n >= 1 && n <= 6 && ops && factors);
  ColorChain cc{};
  cc.n = n;
  for (int k = 0; k < n; ++k) {
    VS_REQUIRE(ops[k] >= 0 && ops[k] <= 4 && (k == 0 || ops[k] != OP_CONTRAST));
    cc.op[k] = ops[k];
    cc.factor[k] = factors[k];
  }
  hipStream_t st = (hipStream_t)stream;
  const int64_t plane = (int64_t)H * W;
  const float* means = nullptr;
  if (ops[0] == OP_CONTRAST) {       // the same two-stage gray mean as vs_aug_color (same partial sums, same order)
    VS_REQUIRE(scratch);
    const int nblk = (int)gridx(plane, 256, 256);
    hipLaunchKernelGGL(gray_partial_kernel, dim3(nblk, F), dim3(256), 0, st, src, plane, scratch);
    hipLaunchKernelGGL(gray_finish_kernel, dim3(F), dim3(64), 0, st, scratch, nblk, plane, scratch + (int64_t)F * nblk);
    means = scratch + (int64_t)F * nblk;
  }
  hipLaunchKernelGGL(color_chain_kernel, dim3(gridx(plane), F), dim3(256), 0, st, src, dst, plane, cc, means);
  return vs_launch_status();
}

// crop window (i0, j0, ch, cw) INSIDE the H x W frames -> resize to oh x ow -> n (0..6) colour ops without OP_CONTRAST; 3-channel frames.
// VS_ERR_UNSUPPORTED when the source window of a 32 x 8 output tile does not fit the LDS budget (very strong down-scaling): callers fall
// back to the separate launches.
extern "C" int vs_aug_crop_resize_color(const float* src, float* dst, int F, int H, int W, int i0, int j0, int ch, int cw, int oh, int ow,
                                        int antialias, int n, const int* ops, const float* factors, void* stream) {
  VS_REQUIRE(src && dst && F > 0 && H > 0 && W > 0 && ch > 0 && cw > 0 && oh > 0 && ow > 0 && n >= 0 && n <= 6 && (n == 0 || (ops && factors)));
  VS_REQUIRE(i0 >= 0 && j0 >= 0 && i0 + ch <= H && j0 + cw <= W);
  ColorChain cc{};
  cc.n = n;
  for (int k = 0; k < n; ++k) {
    VS_REQUIRE(ops[k] >= 0 && ops[k] <= 4 && ops[k] != OP_CONTRAST);
    cc.op[k] = ops[k];
    cc.factor[k] = factors[k];
  }
  // the row-streaming form where its windows fit (resize_stream.h); VIDEOSEAL_CROP_RESIZE=tile / vs_debug_set(4, 1) keep the 32 x 8 tile kernel
  static const bool env_tile = [] { const char* e = getenv("VIDEOSEAL_CROP_RESIZE"); return e && !strcmp(e, "tile"); }();
  const int OWsel = (env_tile || vs_debug_get(VS_DBG_CROP_RESIZE_FORM) == 1) ? 0 : vs_rs::rs_pick(ch, cw, oh, ow, antialias);
  if (OWsel) {
    const int cols = (ow + OWsel - 1) / OWsel;
    int strip = 32;
    for (int cand : {64, 48, 32, 24, 16})
      if ((int64_t)cols * ((oh + cand - 1) / cand) * F >= 512) { strip = cand; break; }
    const size_t lds_s = OWsel == 128 ? vs_rs::rs_lds_bytes<128>() : vs_rs::rs_lds_bytes<64>();
    if ((int64_t)lds_s <= vs_max_lds_bytes()) {
      dim3 gs(cols, (oh + strip - 1) / strip, F);
      const CropResizeEpi epi{dst, oh, ow, cc};
      if (OWsel == 128)
        hipLaunchKernelGGL(crop_resize_color_stream_kernel<128>, gs, dim3(256), lds_s, (hipStream_t)stream, src, H, W, i0, j0, ch, cw, oh, ow, antialias, epi, strip);
      else
        hipLaunchKernelGGL(crop_resize_color_stream_kernel<64>, gs, dim3(256), lds_s, (hipStream_t)stream, src, H, W, i0, j0, ch, cw, oh, ow, antialias, epi, strip);
      return vs_launch_status();
    }
  }
  // bound of the tile's source window: (rows or columns of the tile - 1) * scale + 2 * support + 2 (integer rounding of lo / hi), + 1 of slack
  auto span = [&](int in, int out, int t) {
    const float scale = (float)in / (float)out;
    const float support = antialias ? (scale >= 1.f ? scale : 1.f) : 1.f;
    return (int)((t - 1) * scale + 2.f * support + 4.f);
  };
  const int win_h = span(ch, oh, 8), win_w = span(cw, ow, 32) | 1;      // odd pitch: the 32 lanes of a row walk distinct banks
  const size_t lds = ((size_t)3 * win_h * win_w + 40 * CRC_MAXW + 80) * sizeof(float);
  if (lds > 60 * 1024) return VS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(crop_resize_color_kernel, dim3((ow + 31) / 32, (oh + 7) / 8, F), dim3(256), lds, (hipStream_t)stream, src, dst, H, W, i0, j0,
                     ch, cw, oh, ow, antialias, win_h, win_w, cc);
  return vs_launch_status();
}

extern "C" int64_t vs_aug_color_scratch_floats(int F, int H, int W) {
  const int64_t plane = (int64_t)H * W;
  int64_t g = (plane + 255) / 256;
  g = g > 256 ? 256 : (g < 1 ? 1 : g);
  return (int64_t)F * (g + 1);
}

extern "C" int vs_aug_crop_flip(const float* src, float* dst, int planes, int H, int W, int i0, int j0, int h, int w, int flip, void* stream) {
  VS_REQUIRE(src && dst && planes > 0 && H > 0 && W > 0 && h > 0 && w > 0);
  hipLaunchKernelGGL(crop_flip_kernel, dim3(gridx((int64_t)h * w), planes), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, i0, j0, h, w, flip);
  return vs_launch_status();
}

extern "C" int vs_resize_nchw(const float* src, float* dst, int planes, int H, int W, int oh, int ow, int antialias, void* stream) {
  VS_REQUIRE(src && dst && planes > 0 && H > 0 && W > 0 && oh > 0 && ow > 0);
  hipLaunchKernelGGL(resize_nchw_kernel, dim3((ow + 31) / 32, (oh + 7) / 8, planes), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, oh, ow, antialias);
  return vs_launch_status();
}

extern "C" int vs_gaussian_blur(const float* src, float* tmp, float* dst, int planes, int H, int W, int k, float sigma, void* stream) {
  VS_REQUIRE(src && tmp && dst && planes > 0 && H > 0 && W > 0 && k >= 1 && (k & 1) && k <= 33 && sigma > 0.f);
  VS_REQUIRE(k / 2 < H && k / 2 < W);
  GaussK g; g.k = k;
  // torchvision _get_gaussian_kernel1d: x = linspace(-(k-1)/2, (k-1)/2, k); pdf = exp(-0.5 (x/sigma)^2); pdf / sum
  float sum = 0.f;
  const float half = (k - 1) * 0.5f;
  for (int i = 0; i < k; ++i) {
    const float x = (k == 1) ? 0.f : (-half + (2.f * half) * (float)i / (float)(k - 1));
    g.w[i] = expf(-0.5f * (x / sigma) * (x / sigma));
    sum += g.w[i];
  }
  for (int i = 0; i < k; ++i) g.w[i] /= sum;
  dim3 grid((W + 31) / 32, (H + 7) / 8, planes);
  hipLaunchKernelGGL(blur_pass_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, tmp, H, W, g, 0);
  hipLaunchKernelGGL(blur_pass_kernel, grid, dim3(256), 0, (hipStream_t)stream, tmp, dst, H, W, g, 1);
  return vs_launch_status();
}

extern "C" int vs_median_filter(const float* src, float* dst, int planes, int H, int W, int k, void* stream) {
  VS_REQUIRE(src && dst && planes > 0 && H > 0 && W > 0);
  dim3 grid((W + 31) / 32, (H + 7) / 8, planes);
  hipStream_t st = (hipStream_t)stream;
  switch (k) {
    case 3: hipLaunchKernelGGL(median_kernel<3>, grid, dim3(256), 0, st, src, dst, H, W); break;
    case 5: hipLaunchKernelGGL(median_kernel<5>, grid, dim3(256), 0, st, src, dst, H, W); break;
    case 7: hipLaunchKernelGGL(median_kernel<7>, grid, dim3(256), 0, st, src, dst, H, W); break;
    default: return VS_ERR_UNSUPPORTED;
  }
  return vs_launch_status();
}

extern "C" int64_t vs_jpeg_workspace_bytes(int F, int H, int W) {
  const int64_t Hp = (H + 15) / 16 * 16, Wp = (W + 15) / 16 * 16;
  return (int64_t)F * (Hp * Wp + 2 * (Hp / 2) * (Wp / 2));
}
extern "C" int vs_jpeg_roundtrip(const float* src, float* dst, int F, int H, int W, int quality, void* workspace, void* stream) {
  VS_REQUIRE(src && dst && workspace && F > 0 && H > 0 && W > 0 && quality >= 1 && quality <= 100);
  hipStream_t st = (hipStream_t)stream;
  const int Hp = (H + 15) / 16 * 16, Wp = (W + 15) / 16 * 16;
  unsigned char* Y = (unsigned char*)workspace;
  unsigned char* Cb = Y + (int64_t)F * Hp * Wp;
  unsigned char* Cr = Cb + (int64_t)F * (Hp / 2) * (Wp / 2);
  // whole-MCU frames with 16-byte-aligned tensors take the vector kernels (same bytes); VIDEOSEAL_JPEG=scalar / vs_debug_set(3, 1) keep the scalar ones
  static const bool env_scalar = [] { const char* e = getenv("VIDEOSEAL_JPEG"); return e && !strcmp(e, "scalar"); }();
  const bool vec = !env_scalar && vs_debug_get(VS_DBG_JPEG_FORM) != 1 && H % 16 == 0 && W % 16 == 0 &&
                   ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0 &&
                   ((int64_t)F * Hp * Wp) % 16 == 0;
  if (vec) hipLaunchKernelGGL(jpeg_ycc4_kernel, dim3((W / 4 + 63) / 64, (H / 2 + 3) / 4, F), dim3(256), 0, st, src, H, W, Y, Cb, Cr);
  else hipLaunchKernelGGL(jpeg_ycc_kernel, dim3((Wp / 2 + 31) / 32, (Hp / 2 + 7) / 8, F), dim3(256), 0, st, src, H, W, Hp, Wp, Y, Cb, Cr);
  QTab2 q2;
  q2.l = make_qtab(BASE_L, quality);
  q2.c = make_qtab(BASE_C, quality);
  // the frames are stacked vertically and Cr follows Cb, so ONE launch covers every block of every frame and component (block rows never straddle
  // frames or planes)
  const int64_t nby = ((int64_t)F * Hp / 8) * (Wp / 8), nbc = ((int64_t)F * (Hp / 2) / 8) * (Wp / 16);
  hipLaunchKernelGGL(jpeg_block_kernel, dim3((unsigned)cdiv64(nby + 2 * nbc, 64)), dim3(64), 0, st, Y, Cb, Wp, nby, nbc, q2);
  if (vec) hipLaunchKernelGGL(jpeg_rgb4_kernel, dim3((W / 4 + 63) / 64, (H + 3) / 4, F), dim3(256), 0, st, Y, Cb, Cr, H, W, dst);
  else hipLaunchKernelGGL(jpeg_rgb_kernel, dim3((W + 31) / 32, (H + 7) / 8, F), dim3(256), 0, st, Y, Cb, Cr, H, W, Hp, Wp, dst);
  return vs_launch_status();
}

extern "C" int vs_aug_mask_blend(const float* imgs_w, const float* imgs, const float* mask, float* dst, int F, int C, int H, int W,
                                 void* stream) {
  VS_REQUIRE(imgs_w && imgs && mask && dst && F > 0 && C > 0 && H > 0 && W > 0);
  const int64_t plane = (int64_t)H * W;
  hipLaunchKernelGGL(mask_blend_kernel, dim3(gridx(plane), F * C), dim3(256), 0, (hipStream_t)stream, imgs_w, imgs, mask, dst, C, plane);
  return vs_launch_status();
}

extern "C" int vs_aug_add_scaled(const float* x, const float* noise, float std, float* dst, int64_t n, void* stream) {
  VS_REQUIRE(x && noise && dst && n > 0);
  hipLaunchKernelGGL(add_scaled_kernel, dim3(gridx(n, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, x, noise, std, dst, n);
  return vs_launch_status();
}

extern "C" int vs_aug_gather_frames(const float* src, const int32_t* idx, float* dst, int n_out, int64_t frame_floats, void* stream) {
  VS_REQUIRE(src && idx && dst && n_out > 0 && frame_floats > 0);
  hipLaunchKernelGGL(gather_frames_kernel, dim3(gridx(frame_floats), n_out), dim3(256), 0, (hipStream_t)stream, src, idx, dst, frame_floats);
  return vs_launch_status();
}

extern "C" int vs_aug_window_average(const float* src, float* dst, int F, int64_t frame_floats, int half_window, float alpha, void* stream) {
  VS_REQUIRE(src && dst && F > 0 && frame_floats > 0 && half_window >= 0);
  hipLaunchKernelGGL(window_average_kernel, dim3(gridx(frame_floats), F), dim3(256), 0, (hipStream_t)stream, src, dst, F, frame_floats, half_window, alpha);
  return vs_launch_status();
}
