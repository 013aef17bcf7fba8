// Full-resolution "shell" of the VideoSeal path: the kernels that touch every pixel of every frame.
//   vs_resize_pre    : (anti-aliased) bilinear resize NCHW -> NHWC fused with RGB->Y and the x*2-1 preprocess
//   vs_jnd_heatmap   : JND heat-map at the processing resolution (low-res attenuation)
//   vs_embed_tail    : delta up-resize x JND(img) -> blend -> clamp, one pass over the frame
// These are the HBM-bound kernels (7.08 MB read + 7.08 MB written per 768x768 frame in the tail).
#include "vs_common.h"
#include "resize_taps.h"
#include "resize_stream.h"

namespace {

using namespace vs_taps;

// ---- frame element access: fp32 NCHW planes, or uint8 RGB24 (HWC, what inference_streaming.py:26 reads from the ffmpeg
// pipe).  uint8 -> float is exactly `torch.tensor(clip, dtype=float32) / 255.0`; float -> uint8 is `(x * 255.0).byte()`
// (truncation) of inference_streaming.py:31.
// float(u) / 255.0f, correctly rounded, for u in 0..255: one Newton correction of u * (1/255) (checked exhaustively against IEEE
// division in tests/test_gpu_kernels.py::test_u8_unit_conversion_is_exact); 3 VALU ops instead of the 13 of a true division
__device__ __forceinline__ float u8_unit(float u) {
  const float r = 1.0f / 255.0f;
  const float q = u * r;
  const float rem = __builtin_fmaf(-q, 255.0f, u);
  return __builtin_fmaf(rem, r, q);
}
template <typename T> struct Px;
template <> struct Px<float> {
  static __device__ __forceinline__ float ld(const float* img, int64_t plane, int c, int64_t pix) { return img[c * plane + pix]; }
  static __device__ __forceinline__ void st(float* img, int64_t plane, int c, int64_t pix, float v) { img[c * plane + pix] = v; }
};
template <> struct Px<unsigned char> {
  static __device__ __forceinline__ float ld(const unsigned char* img, int64_t, int c, int64_t pix) { return u8_unit((float)img[pix * 3 + c]); }
  static __device__ __forceinline__ void st(unsigned char* img, int64_t, int c, int64_t pix, float v) { img[pix * 3 + c] = (unsigned char)(v * 255.0f); }
};

// ---------------------------------------------------------------------------------------------------
constexpr int TTW = 256, TTH = 16, TLW = TTW + 4;   // embed tail tile: 256 columns (one per thread) x 16 rows
constexpr int DW_W = 136, DW_H = 12;      // its delta source window in LDS (covers up-resizes by >= 2x; a 3x one needs ~90 x 9)
constexpr int RS_WW = 113, RS_WH = 32;   // LDS window of resize_pre (odd row pitch); covers down-scales up to ~3.3x
constexpr int MAXT = 8;   // taps held in registers by the resize kernels (triangle filter support 2*scale + 1 <= 8 for scale <= 3.5)

template <typename T>
__global__ __launch_bounds__(256) void resize_pre_kernel(const T* __restrict__ src, int B, int C, int H, int W, int oh, int ow,
                                                         int antialias, float* __restrict__ dst_rgb, float mul, float add,
                                                         float* __restrict__ dst_key, int key_step, int key_mode, float y0,
                                                         float y1, float y2) {
  // The block's 32 x 8 outputs read an input window of about (32*scale + support) x (8*scale + support) pixels.  Reading it
  // tap by tap from global memory is L1-bound (4-byte loads 12 bytes apart, every line touched ~7 times), so the window is
  // staged once through LDS with coalesced loads whenever it fits; the arithmetic (and its order) is the same either way.
  __shared__ float Lw[3 * RS_WH * RS_WW];
  const int ox0 = blockIdx.x * 32, oy0 = blockIdx.y * 8;
  const int b = blockIdx.z;
  const int64_t plane = (int64_t)H * W;
  const T* base = src + (int64_t)b * C * plane;
  int x_lo, y_lo, x_hi, y_hi, n_;
  tap_range(ox0, W, ow, antialias, x_lo, n_);
  tap_range(min(ox0 + 31, ow - 1), W, ow, antialias, x_hi, n_);
  const int ww = x_hi + n_ - x_lo;
  tap_range(oy0, H, oh, antialias, y_lo, n_);
  tap_range(min(oy0 + 7, oh - 1), H, oh, antialias, y_hi, n_);
  const int wh = y_hi + n_ - y_lo;
  const bool staged = C == 3 && ww <= RS_WW && wh <= RS_WH;          // block-uniform
  if (staged) {
    // division-free thread -> element maps, every thread's loads issued back to back (the staging is latency-bound otherwise)
    if (sizeof(T) == 1) {            // RGB24: the 3*ww bytes of a window row are contiguous, (x, c) interleaved
      // aligned 4-byte loads (thread = (dword column, row parity)), unpacked into the three LDS planes
      const int jd = threadIdx.x & 127, y0r = threadIdx.x >> 7;
      const unsigned char* fb = reinterpret_cast<const unsigned char*>(base);
      unsigned wrd[RS_WH / 2];
      int mis[RS_WH / 2];
#pragma unroll
      for (int k = 0; k < RS_WH / 2; ++k) {
        wrd[k] = 0u; mis[k] = 0;
        if (y0r + 2 * k < wh) {
          const unsigned char* rowp = fb + ((int64_t)(y_lo + y0r + 2 * k) * W + x_lo) * 3;
          const int m = (int)(reinterpret_cast<uintptr_t>(rowp) & 3);
          mis[k] = m;
          if (4 * jd < 3 * ww + m) wrd[k] = *reinterpret_cast<const unsigned*>(rowp - m + 4 * jd);
        }
      }
#pragma unroll
      for (int k = 0; k < RS_WH / 2; ++k)
        if (y0r + 2 * k < wh) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int bo = 4 * jd + q - mis[k];
            if (bo >= 0 && bo < 3 * ww) {
              const int xx = bo / 3, c = bo - xx * 3;
              Lw[(c * RS_WH + y0r + 2 * k) * RS_WW + xx] = u8_unit((float)((wrd[k] >> (8 * q)) & 0xffu));
            }
          }
        }
    } else {                         // planar: thread = (column, row parity); 3 channels x wh/2 rows each
      const int xx = threadIdx.x & 127, y0r = threadIdx.x >> 7;
      if (xx < ww) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float v[RS_WH / 2];
#pragma unroll
          for (int k = 0; k < RS_WH / 2; ++k)
            if (y0r + 2 * k < wh) v[k] = Px<T>::ld(base, plane, c, (int64_t)(y_lo + y0r + 2 * k) * W + x_lo + xx);
#pragma unroll
          for (int k = 0; k < RS_WH / 2; ++k)
            if (y0r + 2 * k < wh) Lw[(c * RS_WH + y0r + 2 * k) * RS_WW + xx] = v[k];
        }
      }
    }
    __syncthreads();
  }
  const int ox = ox0 + (threadIdx.x & 31);
  const int oy = oy0 + (threadIdx.x >> 5);
  if (ox >= ow || oy >= oh) return;
  const Taps ty = make_taps(oy, H, oh, antialias), tx = make_taps(ox, W, ow, antialias);
  float acc[3] = {0.f, 0.f, 0.f};
  if (staged && tx.n <= MAXT) {
    float wxs[MAXT];
#pragma unroll
    for (int jx = 0; jx < MAXT; ++jx) wxs[jx] = jx < tx.n ? tap_w(tx, jx) : 0.f;
    const float* Lp = Lw + (ty.lo - y_lo) * RS_WW + (tx.lo - x_lo);
    for (int jy = 0; jy < ty.n; ++jy) {
      const float wy = tap_w(ty, jy);
      float r[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int jx = 0; jx < MAXT; ++jx)
        if (jx < tx.n) {
#pragma unroll
          for (int c = 0; c < 3; ++c) r[c] = __builtin_fmaf(wxs[jx], Lp[(c * RS_WH + jy) * RS_WW + jx], r[c]);     // (explicit fma chains: see tail_taps)
        }
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = __builtin_fmaf(wy, r[c], acc[c]);
    }
  } else {
    for (int jy = 0; jy < ty.n; ++jy) {
      const float wy = tap_w(ty, jy);
      const int64_t row = (int64_t)(ty.lo + jy) * W + tx.lo;
      float r[3] = {0.f, 0.f, 0.f};
      for (int jx = 0; jx < tx.n; ++jx) {
        const float wx = tap_w(tx, jx);
        for (int c = 0; c < 3; ++c)
          if (c < C) r[c] = __builtin_fmaf(wx, Px<T>::ld(base, plane, c, row + jx), r[c]);
      }
      for (int c = 0; c < 3; ++c) acc[c] = __builtin_fmaf(wy, r[c], acc[c]);
    }
  }
  const int64_t opix = ((int64_t)oy * ow + ox);
  if (dst_rgb) {
    f32x4 v = {acc[0] * mul + add, C > 1 ? acc[1] * mul + add : 0.f, C > 2 ? acc[2] * mul + add : 0.f, 0.f};
    *reinterpret_cast<f32x4*>(dst_rgb + ((int64_t)b * oh * ow + opix) * 4) = v;
  }
  if (dst_key && (b % key_step) == 0) {
    f32x4 v;
    if (key_mode == 0) {
      const float y = y0 * acc[0] + y1 * acc[1] + y2 * acc[2];
      v = f32x4{y * 2.f - 1.f, 0.f, 0.f, 0.f};
    } else {
      v = f32x4{acc[0] * 2.f - 1.f, acc[1] * 2.f - 1.f, acc[2] * 2.f - 1.f, 0.f};
    }
    *reinterpret_cast<f32x4*>(dst_key + ((int64_t)(b / key_step) * oh * ow + opix) * 4) = v;
  }
}

// ---- row-streaming separable form of resize_pre (round 4) -------------------------------------------------------------------------------
// The tile kernel above stages a (32 scale + support) x (8 scale + support) window per 32 x 8 outputs: at 768 -> 256 that is 102 x 30 pixels for
// 96 x 24 useful ones, and a 408-byte row piece at an arbitrary offset touches 4-5 cache lines -- 1.51 x the frame through HBM (calibrated
// counters, profiles/r03r_hbm_counter_calibration.md) and the horizontal filter of an input row is recomputed by every output row that uses it.
// Here a workgroup owns 128 output columns x a strip of output rows and walks the INPUT rows of the strip top to bottom, eight at a time:
//   1. the eight rows x three planes x (128 scale + support) columns go to LDS with coalesced loads (the next eight are requested first);
//   2. horizontal pass: every (input row, output column) pair is filtered ONCE into a 16-row ring of horizontally resized rows;
//   3. vertical pass: every output row whose tap window is complete is combined from the ring and stored (NHWC rgb / Y key frames).
// The expressions and their order are those of the tile kernel (r += wx * pixel over the taps, then acc += wy * r): bit-identical outputs.
// An input row is fetched once per 128-column tile (+ the strip's first window): ~1.15 x the frame.
// The row-streaming form lives in resize_stream.h (shared with aug.hip's Crop -> Resize -> colour kernel); this is its `resize_pre` epilogue:
// NHWC(4) rgb * mul + add and / or the key frames' Y (or rgb) mapped to [-1, 1].
struct ResizePreEpi {
  float* dst_rgb; float* dst_key; float mul, add; int key_step, key_mode; float y0c, y1c, y2c; int oh, ow;
  __device__ __forceinline__ void operator()(const int b, const int oy, const int ox, const float (&acc)[3]) const {
    const int64_t opix = ((int64_t)oy * ow + ox);
    if (dst_rgb) {
      f32x4 v = {acc[0] * mul + add, acc[1] * mul + add, acc[2] * mul + add, 0.f};
      *reinterpret_cast<f32x4*>(dst_rgb + ((int64_t)b * oh * ow + opix) * 4) = v;
    }
    if (dst_key && (b % key_step) == 0) {
      f32x4 v;
      if (key_mode == 0) {
        const float y = y0c * acc[0] + y1c * acc[1] + y2c * acc[2];
        v = f32x4{y * 2.f - 1.f, 0.f, 0.f, 0.f};
      } else {
        v = f32x4{acc[0] * 2.f - 1.f, acc[1] * 2.f - 1.f, acc[2] * 2.f - 1.f, 0.f};
      }
      *reinterpret_cast<f32x4*>(dst_key + ((int64_t)(b / key_step) * oh * ow + opix) * 4) = v;
    }
  }
};
template <int OW>
__global__ __launch_bounds__(256) void resize_pre_stream_kernel(const float* __restrict__ src, int B, int H, int W, int oh, int ow, int antialias,
                                                                ResizePreEpi epi, int strip) {
  extern __shared__ __attribute__((aligned(16))) float rs_smem[];
  vs_rs::resize_stream_body<OW>(rs_smem, src, H, W, 0, 0, H, W, oh, ow, antialias, strip, epi);
}

// ---- JND (jnd.py:63-108) on a luminance tile held in LDS --------------------------------------------------
struct JndTaps { float lum[25]; float sx[9]; float sy[9]; };

// luminance of 255 * rgb (jnd.py:86-89) and the blend (blender.py:61-68, wam.py:103-113 / jnd.py:110-114) evaluated the way ATen does: separate
// multiplications and additions, no fused multiply-add.  Every kernel of the tail calls these two functions, so its forms agree bit for bit
// (hipcc's contraction of `a * b + c * d` picks either product for the fma depending on the surrounding code).
__device__ __forceinline__ float lum255(const float r, const float g, const float b) {
#pragma clang fp contract(off)
  return 0.299f * (255.f * r) + 0.587f * (255.f * g) + 0.114f * (255.f * b);
}
__device__ __forceinline__ float blend_px(const float si, const float sw, const float p, const float dw, const float hm, const bool fwd_order) {
#pragma clang fp contract(off)
  float v = si * p + sw * dw;
  if (fwd_order) v = p + hm * (v - p);
  return v;
}

__device__ __forceinline__ float jnd_at(const float* L, int stride, int x, int y, const JndTaps& k) {
  // L points at tile origin, (x,y) is the centre inside the tile with a 2-pixel halo available
  float la = 0.f;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) la += k.lum[i * 5 + j] * L[(y + i - 2) * stride + (x + j - 2)];
  la = la / 32.f;
  la = la <= 127.f ? 17.f * (1.f - sqrtf(la / 127.f + 1e-5f)) : 3.f / 128.f * (la - 127.f) + 3.f;
  float gx = 0.f, gy = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float v = L[(y + i - 1) * stride + (x + j - 1)];
      gx += k.sx[i * 3 + j] * v;
      gy += k.sy[i * 3 + j] * v;
    }
  float cm = sqrtf(__builtin_fmaf(gx, gx, gy * gy));
  // cm^2.4 through the hardware log2/exp2 (1 ulp each): relative error <= ~1e-6 on a term that enters the frame scaled by
  // 1/255 * delta * scaling_w, i.e. < 1e-8 absolute -- ocml's powf costs ~10x the instructions
  cm = 16.f * (cm > 0.f ? __builtin_amdgcn_exp2f(2.4f * __builtin_amdgcn_logf(cm)) : 0.f) / (cm * cm + 676.f);
  cm = 0.117f * cm;
  const float h = la + cm - 0.3f * fminf(la, cm);
  return fmaxf(h, 0.f) / 255.f;
}

constexpr int TW = 32, TH = 8, HALO = 2, LW = TW + 2 * HALO, LH = TH + 2 * HALO;

// luminance of 255*img (jnd.py:86-89), zero outside the image (conv zero padding)
__device__ __forceinline__ void load_lum_tile(float* L, const float* img, int64_t sc, int64_t sy, int64_t sx, int H, int W,
                                              int x0, int y0) {
  for (int i = threadIdx.x; i < LW * LH; i += 256) {
    const int ly = i / LW, lx = i - ly * LW;
    const int gx = x0 + lx - HALO, gy = y0 + ly - HALO;
    float v = 0.f;
    if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
      const float* p = img + (int64_t)gy * sy + (int64_t)gx * sx;
      v = lum255(p[0], p[sc], p[2 * sc]);
    }
    L[i] = v;
  }
}

template <typename T, int ROWS>
__device__ __forceinline__ void load_lum_tile_px(float* L, const T* img, int64_t plane, int H, int W, int x0, int y0) {
  for (int i = threadIdx.x; i < LW * (ROWS + 2 * HALO); i += 256) {
    const int ly = i / LW, lx = i - ly * LW;
    const int gx = x0 + lx - HALO, gy = y0 + ly - HALO;
    float v = 0.f;
    if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
      const int64_t pix = (int64_t)gy * W + gx;
      v = lum255(Px<T>::ld(img, plane, 0, pix), Px<T>::ld(img, plane, 1, pix), Px<T>::ld(img, plane, 2, pix));
    }
    L[i] = v;
  }
}

__global__ __launch_bounds__(256) void jnd_heatmap_kernel(const float* __restrict__ img, int H, int W, int64_t sb, int64_t sc,
                                                          int64_t sy, int64_t sx, JndTaps k, float* __restrict__ hmap) {
  __shared__ float L[LW * LH];
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, b = blockIdx.z;
  load_lum_tile(L, img + (int64_t)b * sb, sc, sy, sx, H, W, x0, y0);
  __syncthreads();
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
  const int x = x0 + lx, y = y0 + ly;
  if (x < W && y < H) hmap[((int64_t)b * H + y) * W + x] = jnd_at(L, LW, lx + HALO, ly + HALO, k);
}

// delta taps of one output pixel from the staged source window: d[c] = sum_jy wy * sum_jx wx * Dw[c][..] (Dw = hm*(wa*key_a+wb*key_b))
template <int NT, int DWH = DW_H>
__device__ __forceinline__ void tail_taps(float (&d)[3], const float* Dw, const int Cd, const int base, const int xn, const int yn,
                                          const float* wxp, const float* wyp) {
  int ox[NT], oy[NT];
  float wx[NT], wy[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {       // clamped offsets + zero weights past the real tap count: no branches, all reads in flight
    ox[j] = j < xn ? j : xn - 1;
    oy[j] = (j < yn ? j : yn - 1) * DW_W;
    wx[j] = wxp[j];
    wy[j] = wyp[j];
  }
  for (int c = 0; c < Cd; ++c) {
    const float* p = Dw + c * (DW_W * DWH) + base;
    float v[NT][NT];
#pragma unroll
    for (int jy = 0; jy < NT; ++jy)
#pragma unroll
      for (int jx = 0; jx < NT; ++jx) v[jy][jx] = p[oy[jy] + ox[jx]];
    // explicit fma chains: `r += wx * v` with r starting at 0 leaves hipcc an `a * b + c * d` to contract, and it fuses EITHER product depending on
    // the surrounding code -- two kernels calling this function disagreed in the last bit (round 4)
    float dc = 0.f;
#pragma unroll
    for (int jy = 0; jy < NT; ++jy) {
      float r = 0.f;
#pragma unroll
      for (int jx = 0; jx < NT; ++jx) r = __builtin_fmaf(wx[jx], v[jy][jx], r);
      dc = __builtin_fmaf(wy[jy], r, dc);
    }
    d[c] = dc;
  }
}

// STANDARD JND kernels (jnd.py:24-41: Sobel pair + the 5 x 5 luminance mask = box5 + box3 - 2 centre) evaluated separably inside
// embed_tail_kernel<T, true>: for a group of RG output rows the RG + 4 luminance rows are read once (5 LDS reads per row: x-2 .. x+2), giving per
// row H5 = box5 row sum, H3, S = L[x+1] - L[x-1] and T = L[x-1] + 2 L[x] + L[x+1]; the vertical combinations run over registers -- 10 LDS reads
// per pixel instead of 43.  The luminance mask jumps at la = 127 (jnd.py:66-68): a pixel whose separable sum lands within 0.02 of the jump is
// re-evaluated with the 25-tap order of jnd_at, so the branch taken is the one the generic path / the low-resolution heat-map kernel take.
// Measured (32 x 768^2, fp32, full JND, no preds_w): 43-tap form 227 us, this form 215 us -- the LDS taps were not the limit; what the JND costs
// over the plain tail (137 us) is the separate luminance phase (1.27 x frame read + LDS fill + barrier before the first output row).  A
// column-register form that reads the frame once and keeps the 16 centre pixels in registers needed 195 VGPRs (two waves per SIMD) and
// ran at 391 us: profiles/r03h_shell_*.log, r03i_shell_separable.log.
__device__ __forceinline__ float jnd_finish(float la_sum, float gx, float gy) {
  float la = la_sum / 32.f;
  la = la <= 127.f ? 17.f * (1.f - sqrtf(la / 127.f + 1e-5f)) : 3.f / 128.f * (la - 127.f) + 3.f;
  float cm = sqrtf(__builtin_fmaf(gx, gx, gy * gy));
  cm = 16.f * (cm > 0.f ? __builtin_amdgcn_exp2f(2.4f * __builtin_amdgcn_logf(cm)) : 0.f) / (cm * cm + 676.f);
  cm = 0.117f * cm;
  const float h = la + cm - 0.3f * fminf(la, cm);
  return fmaxf(h, 0.f) / 255.f;
}

// ---------------------------------------------------------------------------------------------------
struct TailArgs {
  const void* imgs; void* out; float* preds_w; const float* delta; const float* hmap_lowres;
  int F, H, W, Sh, Sw, Cd, step, video_mode, total_key, attenuate, clamp, antialias;
  float scaling_i, scaling_w;
};

// Workgroup = 256 consecutive columns x TTH rows of one frame: every wave touches 256 contiguous bytes (fp32) per row and
// plane -- 32-pixel-wide tiles (128-byte pieces at a 3 KB stride) ran at ~1.3 TB/s, a quarter of what a plain copy reaches.
// Each thread owns one column: its horizontal taps are computed once and reused for the TTH rows; the vertical taps
// of the TTH rows are computed by TTH threads into LDS; the source window of delta (x low-res heat-map, x key-frame
// weights) is staged in LDS so the 4..9 taps per pixel are LDS reads.
// TH = rows per tile; KEEP: the tile's own pixels are parked in LDS by the luminance phase and the blend phase reads them from there -- the frame is
// fetched from HBM ONCE.  (Calibrated counters, profiles/r03r_hbm_counter_calibration.md: with the second read from global memory the kernel fetches
// 543 MiB per launch for a 216 MiB batch -- the re-read does not hit in L2 -- and runs at the 4 TB/s the HBM delivers for that traffic; KEEP with
// 8-row tiles (24 KiB of pixels + 12 luminance rows: still three workgroups per CU) fetches 1.5 x the frame instead of 2.27 x.)
template <typename T, bool SEP, int TH, bool KEEP>
__global__ __launch_bounds__(256) void embed_tail_kernel(TailArgs a, JndTaps k) {
  constexpr int TTH = TH;
  constexpr int DWH = TH == 8 ? 8 : DW_H;              // rows of the watermark's source window (8-row tiles at >= 2x up-scale need <= 8)
  __shared__ float Pk[KEEP ? 3 * TH * TTW : 4];
  __shared__ float L[TLW * (TTH + 2 * HALO)];
  __shared__ float Dw[3 * DW_W * DWH];
  __shared__ int ty_lo[TTH], ty_n[TTH];
  __shared__ float ty_w[TTH][4];
  __shared__ int s_xhi, s_nmax, s_xlo;
  const int x0 = blockIdx.x * TTW, y0 = blockIdx.y * TTH, f = blockIdx.z;
  const int64_t plane = (int64_t)a.H * a.W;
  const T* img = static_cast<const T*>(a.imgs) + (int64_t)f * 3 * plane;
  T* outf = static_cast<T*>(a.out) + (int64_t)f * 3 * plane;
  const bool full_jnd = a.attenuate && !a.hmap_lowres;
  if (threadIdx.x == 0) { s_xhi = 0; s_nmax = 0; }
  __syncthreads();
  if (full_jnd) {       // luminance of the tile + 2-pixel halo: thread = column (4 extra halo columns on threads 0-3), all rows
    constexpr int NR = TTH + 2 * HALO;
    auto lum_column = [&](const int lxx) __attribute__((always_inline)) {      // lxx: column inside the LDS tile
      const int gx = x0 + lxx - HALO;
      float v[NR][3];
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) {
        const int gy = y0 + rr - HALO;
        const bool ok = gx >= 0 && gx < a.W && gy >= 0 && gy < a.H;
        const int cx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx), cy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
        const int64_t pix = (int64_t)cy * a.W + cx;          // always a valid address: unconditional loads, zeroed afterwards
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float t = Px<T>::ld(img, plane, c, pix);
          v[rr][c] = ok ? t : 0.f;
        }
      }
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) {
        L[rr * TLW + lxx] = lum255(v[rr][0], v[rr][1], v[rr][2]);
        if (KEEP && rr >= HALO && rr < HALO + TTH && lxx >= HALO && lxx < HALO + TTW)
#pragma unroll
          for (int c = 0; c < 3; ++c) Pk[(c * TTH + rr - HALO) * TTW + lxx - HALO] = v[rr][c];
      }
    };
    lum_column(threadIdx.x + HALO);
    if (threadIdx.x < 2 * HALO) lum_column(threadIdx.x < HALO ? threadIdx.x : TTW + threadIdx.x);
  }
  // taps: own column (registers), rows of the tile (LDS)
  const int lx = threadIdx.x;
  const int x = x0 + lx;
  const Taps tx = make_taps(x < a.W ? x : a.W - 1, a.Sw, a.W, a.antialias);
  float wxs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wxs[j] = j < tx.n ? tap_w(tx, j) : 0.f;
  if (x < a.W) { atomicMax(&s_xhi, tx.lo + tx.n); atomicMax(&s_nmax, tx.n); }
  if (threadIdx.x == 0) s_xlo = tx.lo;        // taps start monotonically: column 0 / row 0 of the tile come first
  if (threadIdx.x < TTH) {
    const int yy = y0 + threadIdx.x;
    const Taps tp = make_taps(yy < a.H ? yy : a.H - 1, a.Sh, a.H, a.antialias);
    ty_lo[threadIdx.x] = tp.lo;
    ty_n[threadIdx.x] = tp.n;
#pragma unroll
    for (int j = 0; j < 4; ++j) ty_w[threadIdx.x][j] = j < tp.n ? tap_w(tp, j) : 0.f;
    if (yy < a.H) atomicMax(&s_nmax, tp.n);
  }
  __syncthreads();

  // key-frame expansion (videoseal.py:80-118): value = wa*key[ka] + wb*key[kb]
  int ka = 0, kb = 0;
  float wa = 1.f, wb = 0.f;
  if (a.video_mode == VS_VIDEO_REPEAT) {
    ka = f / a.step;
  } else if (a.video_mode == VS_VIDEO_ALTERNATE) {
    ka = f / a.step;
    wa = (f % a.step) == 0 ? 1.f : 0.f;
  } else {
    const int ninter = ((a.F - 1) / a.step) * a.step;
    if (f < ninter) {
      ka = f / a.step; kb = ka + 1;
      const int j = f % a.step;
      const float lin = a.step > 1 ? (float)j / (float)(a.step - 1) : 0.f;
      wa = 1.f - lin; wb = 1.f - wa;
    } else {
      ka = a.total_key - 1;
    }
  }
  if (ka >= a.total_key) ka = a.total_key - 1;
  if (kb >= a.total_key) kb = a.total_key - 1;
  const int splane = a.Sh * a.Sw;
  const float* dka = a.delta + (int64_t)ka * a.Cd * splane;
  const float* dkb = a.delta + (int64_t)kb * a.Cd * splane;
  const float* hml = a.hmap_lowres ? a.hmap_lowres + (int64_t)f * splane : nullptr;

  // source window -> LDS, already combined: Dw[c] = hm * (wa*key_a + wb*key_b)  (the per-tap operand of wam.py:184-190)
  const int lasty = min(TTH, a.H - y0) - 1;
  const int wx0 = s_xlo, wy0 = ty_lo[0];
  const int dww = s_xhi - wx0, dwh = ty_lo[lasty] + ty_n[lasty] - wy0;
  const int blk_n = s_nmax;
  const bool staged = blk_n <= 4 && dww <= DW_W && dwh <= DWH;
  const int wx0b = wx0;
  if (staged) {
    for (int i = threadIdx.x; i < dwh * dww; i += 256) {
      const int yy = i / dww, xx = i - yy * dww;
      const int sp = (wy0 + yy) * a.Sw + wx0b + xx;
      const float hm = hml ? hml[sp] : 1.f;
      for (int c = 0; c < a.Cd; ++c) {
        float v = wa * dka[c * splane + sp];
        if (wb != 0.f) v = __builtin_fmaf(wb, dkb[c * splane + sp], v);
        Dw[c * (DW_W * DWH) + yy * DW_W + xx] = hm * v;
      }
    }
  }
  __syncthreads();      // L, Dw
  if (x >= a.W) return;

  auto tap = [&](const int sp, const float wx, float (&r)[3]) __attribute__((always_inline)) {
    const float hm = hml ? hml[sp] : 1.f;
    for (int c = 0; c < a.Cd; ++c) {
      float v = wa * dka[c * splane + sp];
      if (wb != 0.f) v += wb * dkb[c * splane + sp];
      r[c] += wx * (hm * v);
    }
  };
  // rows in groups of RG: the frame loads of a whole group are issued before any of it is consumed -- with one row (3 loads) in
  // flight per wave and ~16 waves per CU the kernel cannot cover the HBM latency (measured 2 TB/s)
  constexpr int RG = 4;
  for (int lyg = 0; lyg <= lasty; lyg += RG) {      // no workgroup barrier below this line
    float px[RG][3];
#pragma unroll
    for (int q = 0; q < RG; ++q)
      if (lyg + q <= lasty) {
        const int64_t pix = (int64_t)(y0 + lyg + q) * a.W + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) px[q][c] = (KEEP && full_jnd) ? Pk[(c * TTH + lyg + q) * TTW + lx] : Px<T>::ld(img, plane, c, pix);
      }
    float hmg[RG] = {1.f, 1.f, 1.f, 1.f};
    if (SEP && full_jnd) {                 // separable stencils for the RG rows of the group: RG + 4 luminance rows, 5 LDS reads each
      float la_s[RG], gxs[RG], gys[RG];
#pragma unroll
      for (int q = 0; q < RG; ++q) { la_s[q] = 0.f; gxs[q] = 0.f; gys[q] = 0.f; }
#pragma unroll
      for (int r8 = 0; r8 < RG + 2 * HALO; ++r8) {
        const float* row = L + (lyg + r8) * TLW + lx + HALO;
        const float m2 = row[-2], m1 = row[-1], c0 = row[0], p1 = row[1], p2 = row[2];
        const float h3 = m1 + c0 + p1;
        const float h5 = h3 + m2 + p2;
        const float sd = p1 - m1;
        const float tt = m1 + 2.f * c0 + p1;
#pragma unroll
        for (int q = 0; q < RG; ++q) {
          const int dy = r8 - (q + HALO);       // row offset of this luminance row relative to output row lyg + q
          if (dy >= -2 && dy <= 2) la_s[q] += h5;
          if (dy >= -1 && dy <= 1) la_s[q] += h3;
          if (dy == 0) la_s[q] -= 2.f * c0;
          if (dy == -1 || dy == 1) gxs[q] += sd;
          if (dy == 0) gxs[q] += 2.f * sd;
          if (dy == -1) gys[q] += tt;
          if (dy == 1) gys[q] -= tt;
        }
      }
#pragma unroll
      for (int q = 0; q < RG; ++q) {
        if (fabsf(la_s[q] * (1.f / 32.f) - 127.f) < 0.02f) hmg[q] = jnd_at(L, TLW, lx + HALO, min(lyg + q, TTH - 1) + HALO, k);   // at the jump: the 25-tap order decides
        else hmg[q] = jnd_finish(la_s[q], gxs[q], gys[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < RG; ++q) {
      const int ly = lyg + q;
      if (ly > lasty) break;
      const int y = y0 + ly;
      float d[3] = {0.f, 0.f, 0.f};
      const int ylo = ty_lo[ly], yn = ty_n[ly];
      if (staged) {
        const int basep = (ylo - wy0) * DW_W + (tx.lo - wx0b);
        if (blk_n <= 3) tail_taps<3, DWH>(d, Dw, a.Cd, basep, tx.n, yn, wxs, ty_w[ly]);
        else tail_taps<4, DWH>(d, Dw, a.Cd, basep, tx.n, yn, wxs, ty_w[ly]);
      } else {
        const Taps ty = make_taps(y, a.Sh, a.H, a.antialias);
        for (int jy = 0; jy < ty.n; ++jy) {
          const float wy = tap_w(ty, jy);
          float r[3] = {0.f, 0.f, 0.f};
          for (int jx = 0; jx < tx.n; ++jx) tap((ty.lo + jy) * a.Sw + tx.lo + jx, tap_w(tx, jx), r);
          for (int c = 0; c < a.Cd; ++c) d[c] += wy * r[c];
        }
      }
      // attenuate == 2: the training forward's order of operations (wam.py:103-113): blend first, then
      // imgs + hmaps * (imgs_w - imgs) (jnd.py:110-114); preds_w stays the un-attenuated resized delta
      const bool fwd_order = full_jnd && a.attenuate == 2;
      float hm = 1.f;
      if (full_jnd) {
        hm = SEP ? hmg[q] : jnd_at(L, TLW, lx + HALO, ly + HALO, k);
        if (!fwd_order)
          for (int c = 0; c < a.Cd; ++c) d[c] = hm * d[c];
      }
      const int64_t pix = (int64_t)y * a.W + x;
      if (a.preds_w)
        for (int c = 0; c < a.Cd; ++c) a.preds_w[((int64_t)f * a.Cd + c) * plane + pix] = d[c];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = blend_px(a.scaling_i, a.scaling_w, px[q][c], d[a.Cd == 1 ? 0 : c], hm, fwd_order);
        if (a.clamp) v = v <= 0.f ? 0.f : (v >= 1.f ? 1.f : v);      // (NaN passes: fminf / fmaxf would turn it into 0)
        Px<T>::st(outf, plane, c, pix, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Row-streaming form of the tail (round 4): workgroup = 256 columns x a STRIP of `strip` rows of one frame, walked top to bottom in groups
// of SG = 4 rows.  What the 16-row-tile kernel above pays per tile is paid once per strip: the column taps, the key-frame expansion, the
// pipeline fill; the frame is fetched ONCE (the 16-row form reads every tile for the luminance phase and again for the blend: 2.27 x the
// frame through HBM, profiles/r03r_hbm_counter_calibration.md) and the halo is 4 rows per strip instead of 4 per 16.
//   * every thread owns one column; the pixels of the next group (SG rows x 3 planes) are requested before the current group is consumed;
//   * JND: the luminance of an incoming row goes into a 12-row LDS ring (the only cross-thread traffic: 5 reads per pixel); its horizontal sums
//     (box5, box3, centre, Sobel pair) are added straight into the accumulators of the five output rows it touches -- eight accumulator sets
//     per thread, rotated by four per group -- in the SAME order (top row first) and with the same expressions as the separable 16-row form,
//     so the heat-map is bit-identical to it; an output row is finished two rows behind the incoming one, its pixels wait in registers;
//   * a pixel whose luminance mask lands within 0.02 of the la = 127 jump is re-evaluated in the 25-tap order from the ring (jnd.py:66-68);
//   * the watermark's source window (x low-res heat-map, x key-frame weights) is staged per 16 output rows into a double-buffered LDS window
//     one group ahead of its first use; ONE barrier per group covers ring, window and row taps (three-slot ring / two-slot window: a writer
//     is always at least one barrier behind the last reader of the slot it overwrites).
// Needs an up-scale by >= 2 in both directions (<= 3 taps per direction, window <= 136 x 12); everything else takes the tile kernel.
constexpr int SG = 4, RING = 3 * SG, SUBR = 16;

__device__ __forceinline__ float jnd_at_ring(const float* Lr, const int cslot, const int lxh, const JndTaps& k) {
  // jnd_at() on the ring: rows cslot - 2 .. cslot + 2 (mod RING), columns lxh - 2 .. lxh + 2; same summation order
  float la = 0.f;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const float* row = Lr + ((cslot + i + RING - 2) % RING) * TLW + lxh;
#pragma unroll
    for (int j = 0; j < 5; ++j) la += k.lum[i * 5 + j] * row[j - 2];
  }
  la = la / 32.f;
  la = la <= 127.f ? 17.f * (1.f - sqrtf(la / 127.f + 1e-5f)) : 3.f / 128.f * (la - 127.f) + 3.f;
  float gx = 0.f, gy = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float* row = Lr + ((cslot + i + RING - 1) % RING) * TLW + lxh;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float v = row[j - 1];
      gx += k.sx[i * 3 + j] * v;
      gy += k.sy[i * 3 + j] * v;
    }
  }
  float cm = sqrtf(__builtin_fmaf(gx, gx, gy * gy));
  cm = 16.f * (cm > 0.f ? __builtin_amdgcn_exp2f(2.4f * __builtin_amdgcn_logf(cm)) : 0.f) / (cm * cm + 676.f);
  cm = 0.117f * cm;
  const float h = la + cm - 0.3f * fminf(la, cm);
  return fmaxf(h, 0.f) / 255.f;
}

// (No waves-per-SIMD hint and no wrapper around the body: `__launch_bounds__(256, 3 | 4)` as well as a __forceinline__ body called from two
// __global__ wrappers made hipcc spill 16-40 bytes per lane; a scratch reload waits for every older global load -- the row prefetch -- and the
// JND form ran 40 % slower: 175-180 -> 245-265 us at 32 x 768^2, profiles/r04_tail_forms.md.)
template <bool JND, int CD>
__global__ __launch_bounds__(256) void embed_tail_stream_kernel(TailArgs a, JndTaps k, const int strip) {
  extern __shared__ float dyn_smem[];
  float* Lr = dyn_smem;                                        // [RING][TLW] luminance ring (JND only)
  float* DwB = dyn_smem + (JND ? RING * TLW : 0);              // [2][Cd][DW_H][DW_W] watermark source windows
  __shared__ int ty_lo[2][SUBR], ty_n[2][SUBR], s_wy0[2];
  __shared__ float ty_w[2][SUBR][4];
  __shared__ int s_xhi, s_xlo;
  constexpr int dwsz = CD * DW_W * DW_H;
  const int x0 = blockIdx.x * TTW, y0 = blockIdx.y * strip, f = blockIdx.z;
  const int ys_end = min(a.H, y0 + strip);
  const int64_t plane = (int64_t)a.H * a.W;
  const float* img = static_cast<const float*>(a.imgs) + (int64_t)f * 3 * plane;
  float* outf = static_cast<float*>(a.out) + (int64_t)f * 3 * plane;
  const int lx = threadIdx.x;
  const int x = x0 + lx;
  const bool xin = x < a.W;
  const int cx = xin ? x : a.W - 1;
  if (threadIdx.x == 0) s_xhi = 0;
  __syncthreads();
  const Taps tx = make_taps(cx, a.Sw, a.W, a.antialias);
  float wxs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wxs[j] = j < tx.n ? tap_w(tx, j) : 0.f;
  if (xin) atomicMax(&s_xhi, tx.lo + tx.n);
  if (threadIdx.x == 0) s_xlo = tx.lo;
  // halo columns of the luminance ring: threads 0 .. 3 carry one extra column each (x0 - 2, x0 - 1, x0 + 256, x0 + 257)
  const bool hal = JND && lx < 2 * HALO;
  const int hgx = lx < HALO ? x0 - HALO + lx : x0 + TTW + (lx - HALO);
  const int hl = lx < HALO ? lx : TTW + lx;
  const bool hin = hal && hgx >= 0 && hgx < a.W;
  const int hcx = hgx < 0 ? 0 : (hgx >= a.W ? a.W - 1 : hgx);

  // key-frame expansion (videoseal.py:80-118): value = wa*key[ka] + wb*key[kb]   (same as the tile kernel)
  int ka = 0, kb = 0;
  float wa = 1.f, wb = 0.f;
  if (a.video_mode == VS_VIDEO_REPEAT) {
    ka = f / a.step;
  } else if (a.video_mode == VS_VIDEO_ALTERNATE) {
    ka = f / a.step;
    wa = (f % a.step) == 0 ? 1.f : 0.f;
  } else {
    const int ninter = ((a.F - 1) / a.step) * a.step;
    if (f < ninter) {
      ka = f / a.step; kb = ka + 1;
      const int j = f % a.step;
      const float lin = a.step > 1 ? (float)j / (float)(a.step - 1) : 0.f;
      wa = 1.f - lin; wb = 1.f - wa;
    } else {
      ka = a.total_key - 1;
    }
  }
  if (ka >= a.total_key) ka = a.total_key - 1;
  if (kb >= a.total_key) kb = a.total_key - 1;
  const int splane = a.Sh * a.Sw;
  const float* dka = a.delta + (int64_t)ka * CD * splane;
  const float* dkb = a.delta + (int64_t)kb * CD * splane;
  const float* hml = a.hmap_lowres ? a.hmap_lowres + (int64_t)f * splane : nullptr;
  __syncthreads();                                             // s_xlo / s_xhi
  const int wx0 = s_xlo, dww = s_xhi - s_xlo;

  auto stage = [&](const int sub, const int b) __attribute__((always_inline)) {      // source window + row taps of output rows [16 sub, 16 sub + 16)
    const int sy0 = y0 + SUBR * sub;
    const int lasty = min(SUBR, ys_end - sy0) - 1;
    int wy0, n0, wyl, nl;
    tap_range(sy0, a.Sh, a.H, a.antialias, wy0, n0);
    tap_range(sy0 + lasty, a.Sh, a.H, a.antialias, wyl, nl);
    const int dwh = wyl + nl - wy0;
    if (threadIdx.x < SUBR) {
      const int yy = sy0 + threadIdx.x;
      const Taps tp = make_taps(yy < a.H ? yy : a.H - 1, a.Sh, a.H, a.antialias);
      ty_lo[b][threadIdx.x] = tp.lo;
      ty_n[b][threadIdx.x] = tp.n;
#pragma unroll
      for (int j = 0; j < 4; ++j) ty_w[b][threadIdx.x][j] = j < tp.n ? tap_w(tp, j) : 0.f;
      if (threadIdx.x == 0) s_wy0[b] = wy0;
    }
    float* Dw = DwB + b * dwsz;
    // the window (<= 136 x 12 elements) in batches of four per thread: every load of a batch is issued before the first value is used -- a loop
    // that loads, combines and stores one element at a time is a chain of dependent L2 round trips, and hipcc guards the registers such a loop
    // re-uses with `s_waitcnt vmcnt(0)`, which also waits for the row prefetch issued just before
    const int nel = dwh * dww;
    for (int i0 = 0; i0 < nel; i0 += 4 * 256) {
      int sp[4], lo[4];
      float hmv[4], da[4][CD], db[4][CD];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = min(i0 + q * 256 + (int)threadIdx.x, nel - 1);
        const int yy = i / dww, xx = i - yy * dww;
        sp[q] = (wy0 + yy) * a.Sw + wx0 + xx;
        lo[q] = yy * DW_W + xx;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        hmv[q] = hml ? hml[sp[q]] : 1.f;
#pragma unroll
        for (int c = 0; c < CD; ++c) {
          da[q][c] = dka[c * splane + sp[q]];
          db[q][c] = wb != 0.f ? dkb[c * splane + sp[q]] : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (i0 + q * 256 + (int)threadIdx.x < nel) {
#pragma unroll
          for (int c = 0; c < CD; ++c) {
            float v = wa * da[q][c];
            if (wb != 0.f) v = __builtin_fmaf(wb, db[q][c], v);
            Dw[c * (DW_W * DW_H) + lo[q]] = hmv[q] * v;
          }
        }
    }
  };
  const int ymax_need = min(a.H - 1, ys_end + (JND ? HALO - 1 : -1));     // last row any thread of this strip reads
  float pf[SG][3], pfh[SG][3];
  auto load_rows = [&](const int r0) __attribute__((always_inline)) {     // rows r0 .. r0 + SG - 1 of the own (and halo) column -> pf / pfh
#pragma unroll
    for (int q = 0; q < SG; ++q) {
      const int gy = r0 + q;
      const int cy = gy < 0 ? 0 : (gy > ymax_need ? ymax_need : gy);       // always a valid address; rows outside are zeroed when used
      const int64_t pix = (int64_t)cy * a.W;
#pragma unroll
      for (int c = 0; c < 3; ++c) pf[q][c] = img[c * plane + pix + cx];
      if (hal) {
#pragma unroll
        for (int c = 0; c < 3; ++c) pfh[q][c] = img[c * plane + pix + hcx];
      }
    }
  };

  const int nrows = ys_end - y0;
  const int niter = (nrows + SG - 1) / SG;
  float acc_la[2 * SG], acc_gx[2 * SG], acc_gy[2 * SG];
#pragma unroll
  for (int i = 0; i < 2 * SG; ++i) { acc_la[i] = 0.f; acc_gx[i] = 0.f; acc_gy[i] = 0.f; }
  float hist[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  load_rows(y0 - HALO);
  stage(0, 0);
  for (int it = -1; it < niter; ++it) {
    const int r0 = y0 + SG * it + HALO;                       // first incoming row of this group; its output rows are r0 - 2 .. r0 + 1
    const int slot0 = ((it + 1) % 3) * SG;
    float cu[SG][3];
#pragma unroll
    for (int q = 0; q < SG; ++q)
#pragma unroll
      for (int c = 0; c < 3; ++c) cu[q][c] = pf[q][c];
    if (JND) {
#pragma unroll
      for (int q = 0; q < SG; ++q) {
        const int gy = r0 + q;
        const bool rok = gy >= 0 && gy < a.H;
        Lr[(slot0 + q) * TLW + lx + HALO] = (rok && xin) ? lum255(cu[q][0], cu[q][1], cu[q][2]) : 0.f;
        if (hal) Lr[(slot0 + q) * TLW + hl] = (rok && hin) ? lum255(pfh[q][0], pfh[q][1], pfh[q][2]) : 0.f;
      }
    }
    if (it + 1 < niter) {
      load_rows(r0 + SG);                                      // in flight while this group is consumed
      if (((it + 1) & 3) == 0 && it + 1 > 0) stage((it + 1) >> 2, ((it + 1) >> 2) & 1);
    }
    __syncthreads();
    const int sb = (it >> 2) & 1;
    const float* Dw = DwB + sb * dwsz;
#pragma unroll
    for (int j = 0; j < SG; ++j) {
      if (JND) {
        const float* row = Lr + (slot0 + j) * TLW + lx + HALO;
        const float m2 = row[-2], m1 = row[-1], c0 = row[0], p1 = row[1], p2 = row[2];
        const float h3 = m1 + c0 + p1;
        const float h5 = h3 + m2 + p2;
        const float sd = p1 - m1;
        const float tt = m1 + 2.f * c0 + p1;
#pragma unroll
        for (int i = j; i <= j + 4; ++i) {
          const int dy = 2 + j - i;                            // incoming row - output row
          if (dy >= -2 && dy <= 2) acc_la[i] += h5;
          if (dy >= -1 && dy <= 1) acc_la[i] += h3;
          if (dy == 0) acc_la[i] -= 2.f * c0;
          if (dy == -1 || dy == 1) acc_gx[i] += sd;
          if (dy == 0) acc_gx[i] += 2.f * sd;
          if (dy == -1) acc_gy[i] += tt;
          if (dy == 1) acc_gy[i] -= tt;
        }
      }
      const int y = r0 - HALO + j;                             // the output row this incoming row completes
      if (it >= 0 && y < ys_end && xin) {
        float hm = 1.f;
        if (JND) {
          if (fabsf(acc_la[j] * (1.f / 32.f) - 127.f) < 0.02f) hm = jnd_at_ring(Lr, (slot0 + j + RING - HALO) % RING, lx + HALO, k);
          else hm = jnd_finish(acc_la[j], acc_gx[j], acc_gy[j]);
        }
        const int ly = (SG * it + j) & (SUBR - 1);
        float d[3] = {0.f, 0.f, 0.f};
        const int basep = (ty_lo[sb][ly] - s_wy0[sb]) * DW_W + (tx.lo - wx0);
        tail_taps<3, DW_H>(d, Dw, CD, basep, tx.n, ty_n[sb][ly], wxs, ty_w[sb][ly]);
        const bool fwd_order = JND && a.attenuate == 2;
        if (JND && !fwd_order)
#pragma unroll
          for (int c = 0; c < CD; ++c) d[c] = hm * d[c];
        const int64_t pix = (int64_t)y * a.W + x;
        if (a.preds_w)
#pragma unroll
          for (int c = 0; c < CD; ++c) a.preds_w[((int64_t)f * CD + c) * plane + pix] = d[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float p = j < 2 ? hist[j][c] : cu[j - 2][c];
          float v = blend_px(a.scaling_i, a.scaling_w, p, d[CD == 1 ? 0 : c], hm, fwd_order);
          if (a.clamp) v = v <= 0.f ? 0.f : (v >= 1.f ? 1.f : v);      // (NaN passes)
          outf[c * plane + pix] = v;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < SG; ++i) {
      acc_la[i] = acc_la[i + SG]; acc_gx[i] = acc_gx[i + SG]; acc_gy[i] = acc_gy[i + SG];
      acc_la[i + SG] = 0.f; acc_gx[i + SG] = 0.f; acc_gy[i + SG] = 0.f;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { hist[0][c] = cu[2][c]; hist[1][c] = cu[3][c]; }
  }
}

// (no waves-per-SIMD hint on the JND form: with `__launch_bounds__(256, 3)` or `(256, 4)` hipcc schedules the same body 40 % slower -- 132
// VGPRs and three waves per SIMD either way, measured 175 -> 245-260 us at 32 x 768^2, profiles/r04_tail_forms.md)


// ---------------------------------------------------------------------------------------------------
// the taps of jnd.py:24-41 (what every released card carries in its state dict)
bool standard_jnd_taps(const float* t) {
  static const float lum[25] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 2, 0, 2, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1};
  static const float sx[9] = {-1, 0, 1, -2, 0, 2, -1, 0, 1}, sy[9] = {1, 2, 1, 0, 0, 0, -1, -2, -1};
  for (int i = 0; i < 25; ++i) if (t[i] != lum[i]) return false;
  for (int i = 0; i < 9; ++i) if (t[25 + i] != sx[i] || t[34 + i] != sy[i]) return false;
  return true;
}

JndTaps taps_from(const float* t43) {
  JndTaps k;
  for (int i = 0; i < 25; ++i) k.lum[i] = t43[i];
  for (int i = 0; i < 9; ++i) { k.sx[i] = t43[25 + i]; k.sy[i] = t43[34 + i]; }
  return k;
}

}  // namespace

extern "C" int vs_resize_pre(const float* src, int B, int C, int H, int W, int oh, int ow, int antialias, float* dst_rgb,
                             float mul, float add, float* dst_key, int key_step, const float* ymat3, void* stream) {
  VS_REQUIRE(src && B > 0 && C >= 1 && C <= 3 && H > 0 && W > 0 && oh > 0 && ow > 0 && (dst_rgb || dst_key));
  VS_REQUIRE(!dst_key || key_step >= 1);
  const int key_mode = ymat3 ? 0 : 1;
  const float y0 = ymat3 ? ymat3[0] : 0.f, y1 = ymat3 ? ymat3[1] : 0.f, y2 = ymat3 ? ymat3[2] : 0.f;
  // row-streaming separable kernel (default) where its windows fit: 3 planes, <= 8 taps per direction, <= 448 input columns per 128 outputs
  // (scales up to 3.4); VIDEOSEAL_RESIZE=tile keeps the 32 x 8 tile kernel, VS_RESIZE_STRIP=<output rows> overrides the strip height
  static const bool env_tile = [] { const char* e = getenv("VIDEOSEAL_RESIZE"); return e && !strcmp(e, "tile"); }();       // process-wide default, read once
  const bool tile_only = env_tile || vs_debug_get(VS_DBG_RESIZE_FORM) == 1;
  static const int env_strip = [] { const char* e = getenv("VS_RESIZE_STRIP"); return e ? atoi(e) : 0; }();
  // two tile shapes (resize_stream.h): 128 output columns for scales up to 3.4 (<= 8 taps), 64 columns for scales up to 4 (<= 10 taps, 1024 -> 256)
  const int OWsel = vs_rs::rs_pick(H, W, oh, ow, antialias);
  if (!tile_only && C == 3 && OWsel) {
    const int cols = (ow + OWsel - 1) / OWsel;
    int strip = 32;
    for (int cand : {64, 48, 32, 24, 16})            // tallest strip that still gives every CU two workgroups (one round at two resident per CU)
      if ((int64_t)cols * ((oh + cand - 1) / cand) * B >= 512) { strip = cand; break; }
    if (env_strip >= 2) strip = env_strip;
    if (const int v = vs_debug_get(VS_DBG_RESIZE_STRIP); v >= 1) strip = v;       // tests: any strip height, per call
    dim3 gs(cols, (oh + strip - 1) / strip, B);
    const size_t lds = OWsel == 128 ? vs_rs::rs_lds_bytes<128>() : vs_rs::rs_lds_bytes<64>();
    const ResizePreEpi epi{dst_rgb, dst_key, mul, add, key_step < 1 ? 1 : key_step, key_mode, y0, y1, y2, oh, ow};
    if ((int64_t)lds <= vs_max_lds_bytes()) {           // 68 KiB: beyond the 64 KiB of pre-gfx950 parts -> the tile kernel below
      if (OWsel == 128) hipLaunchKernelGGL(resize_pre_stream_kernel<128>, gs, dim3(256), lds, (hipStream_t)stream, src, B, H, W, oh, ow, antialias, epi, strip);
      else hipLaunchKernelGGL(resize_pre_stream_kernel<64>, gs, dim3(256), lds, (hipStream_t)stream, src, B, H, W, oh, ow, antialias, epi, strip);
      return vs_launch_status();
    }
  }
  dim3 grid((ow + 31) / 32, (oh + 7) / 8, B);
  hipLaunchKernelGGL(resize_pre_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src, B, C, H, W, oh, ow, antialias, dst_rgb, mul,
                     add, dst_key, key_step < 1 ? 1 : key_step, key_mode, y0, y1, y2);
  return vs_launch_status();
}

extern "C" int vs_resize_pre_u8(const unsigned char* src, int B, int H, int W, int oh, int ow, int antialias, float* dst_rgb,
                                float mul, float add, float* dst_key, int key_step, const float* ymat3, void* stream) {
  VS_REQUIRE(src && B > 0 && H > 0 && W > 0 && oh > 0 && ow > 0 && (dst_rgb || dst_key));
  VS_REQUIRE(!dst_key || key_step >= 1);
  const int key_mode = ymat3 ? 0 : 1;
  const float y0 = ymat3 ? ymat3[0] : 0.f, y1 = ymat3 ? ymat3[1] : 0.f, y2 = ymat3 ? ymat3[2] : 0.f;
  dim3 grid((ow + 31) / 32, (oh + 7) / 8, B);
  hipLaunchKernelGGL(resize_pre_kernel<unsigned char>, grid, dim3(256), 0, (hipStream_t)stream, src, B, 3, H, W, oh, ow, antialias,
                     dst_rgb, mul, add, dst_key, key_step < 1 ? 1 : key_step, key_mode, y0, y1, y2);
  return vs_launch_status();
}

extern "C" int vs_jnd_heatmap(const float* img, int B, int H, int W, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                              const float* taps43, float* hmap, void* stream) {
  VS_REQUIRE(img && taps43 && hmap && B > 0 && H > 0 && W > 0);
  dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, B);
  hipLaunchKernelGGL(jnd_heatmap_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, H, W, sb, sc, sy, sx, taps_from(taps43),
                     hmap);
  return vs_launch_status();
}

extern "C" int vs_embed_tail(const vs_tail_desc_t* d, void* stream) {
  VS_REQUIRE(d && d->imgs && d->out && d->delta && d->F > 0 && d->H > 0 && d->W > 0 && d->S_h > 0 && d->S_w > 0);
  VS_REQUIRE((d->Cd == 1 || d->Cd == 3) && d->step >= 1 && d->total_key >= 1);
  VS_REQUIRE(d->video_mode >= 0 && d->video_mode <= 2);
  VS_REQUIRE(!(d->attenuate && !d->hmap_lowres) || d->taps43);
  TailArgs a{d->imgs, d->out, d->preds_w, d->delta, d->attenuate ? d->hmap_lowres : nullptr, d->F, d->H, d->W, d->S_h, d->S_w,
             d->Cd, d->step, d->video_mode, d->total_key, d->attenuate, d->clamp, d->antialias, d->scaling_i, d->scaling_w};
  JndTaps k{};
  if (d->taps43) k = taps_from(d->taps43);
  dim3 grid((d->W + TTW - 1) / TTW, (d->H + TTH - 1) / TTH, d->F);
  dim3 grid8((d->W + TTW - 1) / TTW, (d->H + 7) / 8, d->F);
  // full-resolution heat-map with the standard kernels: separable stencil evaluation on 16-row tiles (default).  VIDEOSEAL_TAIL=v1 keeps the 43-tap
  // form, =keep selects 8-row tiles whose pixels stay in LDS between the luminance and the blend phase -- it fetches 1.5 x the frame instead of
  // 2.27 x but runs at 281 us against 215 us (32 x 768^2): twice the workgroups, and the per-workgroup set-up (taps, watermark window, two
  // barriers before the first output row), not HBM traffic, is what the tail pays for (profiles/r03s_shell_keep_form.log)
  // VIDEOSEAL_TAIL: stream (default) = the row-streaming kernel where it applies, sep = separable stencils on 16-row tiles (round 3's default),
  // v1 = 43-tap form, keep = 8-row tiles with the pixels parked in LDS.  d->variant (1 = v1, 2 = sep, 3 = keep, 4 = stream) overrides it per call.
  static const int env_mode = [] { const char* e = getenv("VIDEOSEAL_TAIL"); return !e ? 3 : (!strcmp(e, "v1") ? 0 : (!strcmp(e, "keep") ? 2 : (!strcmp(e, "sep") ? 1 : 3))); }();
  const int mode = d->variant >= 1 && d->variant <= 4 ? d->variant - 1 : env_mode;
  const bool full_jnd = d->attenuate && !d->hmap_lowres;
  const bool sep = mode > 0 && full_jnd && d->taps43 && standard_jnd_taps(d->taps43);
  if (mode == 3 && !d->io_u8 && (sep || !full_jnd) && d->W >= 2 * d->S_w && d->H >= 2 * d->S_h) {
    // VS_TAIL_STRIP overrides the strip height
    static const int env_strip = [] { const char* e = getenv("VS_TAIL_STRIP"); return e ? atoi(e) : 0; }();
    const int64_t cols = (d->W + TTW - 1) / TTW;
    // strip height, from tools/bench_tail_sweep.py (profiles/r04zz_tail_sweep.log, 768^2; us for 16 / 32 / 128 frames): without the luminance phase
    // short strips win (32 rows 43 / 104 / 364, 48 rows 44 / 94 / 360, 96 rows 61 / 114 / 354); with it the strip should be tall (4 halo rows + the
    // pipeline fill per strip) as long as the launch still has three workgroups per CU: 96 rows 115 / 155 / 503, 48 rows 76 / 151 / 510, 32 rows 87 / 152 / 521
    int strip = 32;
    if (full_jnd) {
      for (int cand : {96, 48})
        if (cols * ((d->H + cand - 1) / cand) * d->F >= 768) { strip = cand; break; }
    } else if (cols * ((d->H + 47) / 48) * d->F >= 1024) {
      strip = 48;
    }
    if (env_strip >= 4) strip = (env_strip + 3) / 4 * 4;
    if (const int v = vs_debug_get(VS_DBG_TAIL_STRIP); v >= 4) strip = (v + 3) / 4 * 4;     // tests: every strip height, per call
    dim3 gs((unsigned)cols, (d->H + strip - 1) / strip, d->F);
    const size_t lds_w = (size_t)2 * d->Cd * DW_W * DW_H * sizeof(float);
    const size_t lds_j = lds_w + RING * TLW * sizeof(float);
    if (full_jnd && d->Cd == 1) hipLaunchKernelGGL((embed_tail_stream_kernel<true, 1>), gs, dim3(256), lds_j, (hipStream_t)stream, a, k, strip);
    else if (full_jnd) hipLaunchKernelGGL((embed_tail_stream_kernel<true, 3>), gs, dim3(256), lds_j, (hipStream_t)stream, a, k, strip);
    else if (d->Cd == 1) hipLaunchKernelGGL((embed_tail_stream_kernel<false, 1>), gs, dim3(256), lds_w, (hipStream_t)stream, a, k, strip);
    else hipLaunchKernelGGL((embed_tail_stream_kernel<false, 3>), gs, dim3(256), lds_w, (hipStream_t)stream, a, k, strip);
    return vs_launch_status();
  }
  if (d->io_u8) {
    VS_REQUIRE(d->clamp);        // (x * 255).byte() is only defined for x in [0, 1]
    // (uint8 frames keep the 43-tap form: measured 215 us against 248 us for the separable one, profiles/r03i_shell_separable.log)
    hipLaunchKernelGGL((embed_tail_kernel<unsigned char, false, 16, false>), grid, dim3(256), 0, (hipStream_t)stream, a, k);
  } else {
    if (sep && mode == 2) hipLaunchKernelGGL((embed_tail_kernel<float, true, 8, true>), grid8, dim3(256), 0, (hipStream_t)stream, a, k);
    else if (sep) hipLaunchKernelGGL((embed_tail_kernel<float, true, 16, false>), grid, dim3(256), 0, (hipStream_t)stream, a, k);
    else hipLaunchKernelGGL((embed_tail_kernel<float, false, 16, false>), grid, dim3(256), 0, (hipStream_t)stream, a, k);
  }
  return vs_launch_status();
}
