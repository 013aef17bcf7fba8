// Full-resolution "shell" of the VideoSeal path: the kernels that touch every pixel of every frame.
//   vs_resize_pre    : (anti-aliased) bilinear resize NCHW -> NHWC fused with RGB->Y and the x*2-1 preprocess
//   vs_jnd_heatmap   : JND heat-map at the processing resolution (low-res attenuation)
//   vs_embed_tail    : delta up-resize x JND(img) -> blend -> clamp, one pass over the frame
// These are the HBM-bound kernels (7.08 MB read + 7.08 MB written per 768x768 frame in the tail).
#include "vs_common.h"

namespace {

// ---- ATen-compatible 1-D interpolation taps (aten/native/cpu/UpSampleKernel.cpp semantics) -------------------
// antialias: triangle filter, support = max(scale,1), weights normalised over the taps that fall inside the input.
// plain    : 2-tap bilinear, align_corners=False, source index clamped at 0.
struct Taps {
  int lo, n;          // first input index, tap count
  float center, inv, total;   // antialias parameters
  float l1;           // plain: weight of the second tap
  bool aa;
};

__device__ __forceinline__ float tri(float x) { x = fabsf(x); return x < 1.f ? 1.f - x : 0.f; }

__device__ __forceinline__ Taps make_taps(int i, int in, int out, bool antialias) {
  Taps t;
  t.aa = antialias;
  const float scale = (float)in / (float)out;
  if (antialias) {
    const float support = scale >= 1.f ? scale : 1.f;
    t.center = scale * (i + 0.5f);
    t.inv = scale >= 1.f ? 1.f / scale : 1.f;
    int lo = (int)(t.center - support + 0.5f);
    lo = lo < 0 ? 0 : lo;
    int hi = (int)(t.center + support + 0.5f);
    hi = hi > in ? in : hi;
    t.lo = lo;
    t.n = hi - lo;
    float tot = 0.f;
    for (int j = 0; j < t.n; ++j) tot += tri((j + lo - t.center + 0.5f) * t.inv);
    t.total = tot;
    t.l1 = 0.f;
  } else {
    float src = scale * (i + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    int i0 = (int)src;
    i0 = i0 > in - 1 ? in - 1 : i0;
    t.lo = i0;
    t.n = 1 + (i0 < in - 1);
    t.l1 = src - i0;
    t.center = t.inv = t.total = 0.f;
  }
  return t;
}
__device__ __forceinline__ float tap_w(const Taps& t, int j) {
  if (t.aa) {
    const float w = tri((j + t.lo - t.center + 0.5f) * t.inv);
    return t.total != 0.f ? w / t.total : w;
  }
  if (t.n == 1) return 1.f;   // last row/column: ATen blends the border pixel with itself
  return j == 0 ? 1.f - t.l1 : t.l1;
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_pre_kernel(const float* __restrict__ src, int B, int C, int H, int W, int oh, int ow,
                                                         int antialias, float* __restrict__ dst_rgb, float mul, float add,
                                                         float* __restrict__ dst_key, int key_step, int key_mode, float y0,
                                                         float y1, float y2) {
  const int ox = blockIdx.x * 32 + (threadIdx.x & 31);
  const int oy = blockIdx.y * 8 + (threadIdx.x >> 5);
  const int b = blockIdx.z;
  if (ox >= ow || oy >= oh) return;
  const Taps ty = make_taps(oy, H, oh, antialias), tx = make_taps(ox, W, ow, antialias);
  float acc[3] = {0.f, 0.f, 0.f};
  const int64_t plane = (int64_t)H * W;
  const float* base = src + (int64_t)b * C * plane;
  for (int jy = 0; jy < ty.n; ++jy) {
    const float wy = tap_w(ty, jy);
    const float* row = base + (int64_t)(ty.lo + jy) * W + tx.lo;
    float r[3] = {0.f, 0.f, 0.f};
    for (int jx = 0; jx < tx.n; ++jx) {
      const float wx = tap_w(tx, jx);
      for (int c = 0; c < 3; ++c)
        if (c < C) r[c] += wx * row[c * plane + jx];
    }
    for (int c = 0; c < 3; ++c) acc[c] += wy * r[c];
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

// ---- JND (jnd.py:63-108) on a luminance tile held in LDS --------------------------------------------------
struct JndTaps { float lum[25]; float sx[9]; float sy[9]; };

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
  float cm = sqrtf(gx * gx + gy * gy);
  cm = 16.f * powf(cm, 2.4f) / (cm * cm + 676.f);
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
      v = 0.299f * (255.f * p[0]) + 0.587f * (255.f * p[sc]) + 0.114f * (255.f * p[2 * sc]);
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

// ---------------------------------------------------------------------------------------------------
struct TailArgs {
  const float* imgs; float* out; float* preds_w; const float* delta; const float* hmap_lowres;
  int F, H, W, Sh, Sw, Cd, step, video_mode, total_key, attenuate, clamp, antialias;
  float scaling_i, scaling_w;
};

__global__ __launch_bounds__(256) void embed_tail_kernel(TailArgs a, JndTaps k) {
  __shared__ float L[LW * LH];
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, f = blockIdx.z;
  const int64_t plane = (int64_t)a.H * a.W;
  const float* img = a.imgs + (int64_t)f * 3 * plane;
  const bool full_jnd = a.attenuate && !a.hmap_lowres;
  if (full_jnd) {
    load_lum_tile(L, img, plane, a.W, 1, a.H, a.W, x0, y0);
    __syncthreads();
  }
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
  const int x = x0 + lx, y = y0 + ly;
  if (x >= a.W || y >= a.H) return;

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

  const Taps ty = make_taps(y, a.Sh, a.H, a.antialias), tx = make_taps(x, a.Sw, a.W, a.antialias);
  const int64_t splane = (int64_t)a.Sh * a.Sw;
  float d[3] = {0.f, 0.f, 0.f};
  for (int jy = 0; jy < ty.n; ++jy) {
    const float wy = tap_w(ty, jy);
    float r[3] = {0.f, 0.f, 0.f};
    for (int jx = 0; jx < tx.n; ++jx) {
      const float wx = tap_w(tx, jx);
      const int64_t sp = (int64_t)(ty.lo + jy) * a.Sw + (tx.lo + jx);
      const float hm = a.hmap_lowres ? a.hmap_lowres[(int64_t)f * splane + sp] : 1.f;
      for (int c = 0; c < a.Cd; ++c) {
        float v = wa * a.delta[((int64_t)ka * a.Cd + c) * splane + sp];
        if (wb != 0.f) v += wb * a.delta[((int64_t)kb * a.Cd + c) * splane + sp];
        r[c] += wx * (hm * v);
      }
    }
    for (int c = 0; c < a.Cd; ++c) d[c] += wy * r[c];
  }
  if (full_jnd) {
    const float hm = jnd_at(L, LW, lx + HALO, ly + HALO, k);
    for (int c = 0; c < a.Cd; ++c) d[c] = hm * d[c];
  }
  const int64_t pix = (int64_t)y * a.W + x;
  if (a.preds_w)
    for (int c = 0; c < a.Cd; ++c) a.preds_w[((int64_t)f * a.Cd + c) * plane + pix] = d[c];
  for (int c = 0; c < 3; ++c) {
    float v = a.scaling_i * img[c * plane + pix] + a.scaling_w * d[a.Cd == 1 ? 0 : c];
    if (a.clamp) v = fminf(fmaxf(v, 0.f), 1.f);
    a.out[(int64_t)f * 3 * plane + c * plane + pix] = v;
  }
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
  dim3 grid((ow + 31) / 32, (oh + 7) / 8, B);
  hipLaunchKernelGGL(resize_pre_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, B, C, H, W, oh, ow, antialias, dst_rgb, mul,
                     add, dst_key, key_step < 1 ? 1 : key_step, key_mode, y0, y1, y2);
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
  dim3 grid((d->W + TW - 1) / TW, (d->H + TH - 1) / TH, d->F);
  hipLaunchKernelGGL(embed_tail_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, k);
  return vs_launch_status();
}
