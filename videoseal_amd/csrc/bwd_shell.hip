// Adjoints of the full-resolution shell and of the augmentations between embed and detect: what turns d(loss)/d(imgs_w) and
// d(loss)/d(imgs_aug) into d(delta), the gradient the U-Net backward (bwd_unet.hip) starts from -- the generator side of
// train.py:626-643.  Reference operators whose transposes live here:
//   models/wam.py:93-113 / videoseal.py:196-228   resize-up of the watermark, blend, attenuation(imgs, imgs_w), clamp  -> vs_embed_tail_bwd,
//                                                 vs_resize_nchw_bwd, vs_tail_key_reduce
//   augmentation/augmenter.py:175                 imgs_w * m + imgs * (1 - m)                                        -> vs_mask_mul
//   augmentation/geometric.py:94-124, 186-196     Crop / HorizontalFlip                                             -> vs_aug_crop_flip_bwd
//   F.interpolate / torchvision resize (antialias) wam.py:117, geometric.py:62-91                                    -> vs_resize_nchw_bwd
//   augmentation/valuemetric.py:53-175            Brightness / Contrast / Saturation / Grayscale                     -> vs_aug_color_bwd
//   models/extractor.py:163  x * 2 - 1 (+ the NCHW -> NHWC re-layout in front of the stem)                           -> vs_nhwc_to_nchw_scaled
//   losses/perceptual.py:20-28, yuvloss.py:11-27  mse / yuv perceptual terms and their gradient                      -> vs_percep_mse
// Every adjoint is written in GATHER form (one thread per input-gradient element, fixed summation order): deterministic, no atomics.
// All tensors fp32, frames NCHW planes; HBM-bound elementwise / small-stencil work.
#include "vs_common.h"
#include "resize_taps.h"

namespace {
using namespace vs_taps;

inline unsigned gridx(int64_t n, int per = 256, int64_t cap = 4096) {
  int64_t g = (n + per - 1) / per;
  return (unsigned)(g > cap ? cap : (g < 1 ? 1 : g));
}

// ---- separable resize adjoint: forward y[oy][ox] = sum_jy wy(oy, jy) sum_jx wx(ox, jx) x[lo_y + jy][lo_x + jx]  (resize_nchw_kernel, aug.hip;
// the delta taps of embed_tail_kernel, shell.hip).  Pass 1 transposes the x taps (tmp[oy][ix] = sum_ox wx(ox; ix) dy[oy][ox]), pass 2 the y taps.
// `dir` = 0: along x (row length n_in -> n_out columns of dy), 1: along y.
__global__ __launch_bounds__(256) void resize_bwd_pass_kernel(const float* __restrict__ dy, float* __restrict__ dx, int n_in, int n_out,
                                                              int other, int dir, int antialias) {
  // dir 0: dy [planes][other = rows of the image][n_out], dx [planes][other][n_in]
  // dir 1: dy [planes][n_out][other = columns],            dx [planes][n_in][other]
  const int64_t plane = blockIdx.z;
  const int a = blockIdx.x * 64 + (threadIdx.x & 63);        // fastest index of the image (a column)
  const int b = blockIdx.y * 4 + (threadIdx.x >> 6);          // a row
  const int ncol = dir == 0 ? n_in : other, nrow = dir == 0 ? other : n_in;
  if (a >= ncol || b >= nrow) return;
  const int i = dir == 0 ? a : b;                             // the resized coordinate of this thread's input-gradient element
  int o_lo, o_hi;
  adjoint_range(i, n_in, n_out, antialias, o_lo, o_hi);
  const float* src = dy + plane * (int64_t)(dir == 0 ? other * n_out : n_out * other);
  float acc = 0.f;
  for (int o = o_lo; o <= o_hi; ++o) {
    const Taps t = make_taps(o, n_in, n_out, antialias);
    const int j = i - t.lo;
    if (j < 0 || j >= t.n) continue;
    const float w = tap_w(t, j);
    acc += w * (dir == 0 ? src[(int64_t)b * n_out + o] : src[(int64_t)o * other + a]);
  }
  dx[plane * (int64_t)nrow * ncol + (int64_t)b * ncol + a] = acc;
}

// ---- crop / flip adjoint: zero outside the window, the (mirrored) gradient inside
__global__ __launch_bounds__(256) void crop_flip_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int i0, int j0, int h,
                                                            int w, int flip) {
  const int64_t pl = blockIdx.y;
  const float* s = dy + pl * (int64_t)h * w;
  float* d = dx + pl * (int64_t)H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)H * W; i += (int64_t)gridDim.x * 256) {
    const int sy = (int)(i / W), sx = (int)(i - (int64_t)sy * W);
    const int y = sy - i0, xx = sx - j0;
    float v = 0.f;
    if (y >= 0 && y < h && xx >= 0 && xx < w) v = s[(int64_t)y * w + (flip ? (w - 1 - xx) : xx)];
    d[i] = v;
  }
}

// ---- dst[f][c][pix] = mul * src[(f, pix)][c]: the extractor's input gradient (NHWC, ld floats per pixel) back to frame planes
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, int64_t plane, int C, int64_t ld, float mul,
                                                           float* __restrict__ dst) {
  const int64_t f = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (int64_t)gridDim.x * 256) {
    const float* p = src + (f * plane + i) * ld;
    for (int c = 0; c < C; ++c) dst[(f * C + c) * plane + i] = mul * p[c];
  }
}

// ---- dx[f][c] = dy[f][c] * m[f]  (m broadcast over the C planes; complement = 1: (1 - m))
__global__ __launch_bounds__(256) void mask_mul_kernel(const float* __restrict__ dy, const float* __restrict__ m, float* __restrict__ dx, int Cc,
                                                       int64_t plane, int complement) {
  const int64_t pl = blockIdx.y;
  const float* mk = m + (pl / Cc) * plane;
  const int64_t o = pl * plane;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (int64_t)gridDim.x * 256)
    dx[o + i] = dy[o + i] * (complement ? 1.f - mk[i] : mk[i]);
}

// ---- blend / attenuation / clamp adjoint at full resolution.  Forward (embed_tail_kernel): d = resized watermark (x low-res heat-map),
//   v = si * px + sw * d;  with the full-resolution heat-map (the training order of wam.py:103-113, attenuate == 2): v = px + hm * (v - px);  clamp.
//   torch.clamp passes the gradient where 0 <= v <= 1.  g[f][cd] = sum over the frame channels that read d[cd]; `preds` is the forward's preds_w
//   output (the resized, un-attenuated watermark d), whose own upstream gradient d_preds is added.
struct TailBwdArgs {
  const float* imgs; const float* preds; const float* hmap; const float* d_out; const float* d_preds; float* g;
  int F, Cd, clamp;
  int64_t plane;
  float si, sw;
};
__global__ __launch_bounds__(256) void embed_tail_bwd_kernel(TailBwdArgs a) {
  const int64_t f = blockIdx.y;
  const float* img = a.imgs + f * 3 * a.plane;
  const float* dout = a.d_out ? a.d_out + f * 3 * a.plane : nullptr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.plane; i += (int64_t)gridDim.x * 256) {
    const float hm = a.hmap ? a.hmap[f * a.plane + i] : 1.f;
    float gd[3] = {0.f, 0.f, 0.f};
    if (dout) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int cd = a.Cd == 1 ? 0 : c;
        const float d = a.preds[(f * a.Cd + cd) * a.plane + i];
        const float px = img[c * a.plane + i];
        float v = a.si * px + a.sw * d;
        if (a.hmap) v = px + hm * (v - px);
        const bool pass = !a.clamp || (v >= 0.f && v <= 1.f);
        if (pass) gd[cd] += dout[c * a.plane + i] * (a.sw * hm);
      }
    }
    for (int cd = 0; cd < a.Cd; ++cd) {
      float g = gd[cd];
      if (a.d_preds) g += a.d_preds[(f * a.Cd + cd) * a.plane + i];
      a.g[(f * a.Cd + cd) * a.plane + i] = g;
    }
  }
}

// ---- key-frame expansion adjoint (videoseal.py:80-118) + the low-resolution heat-map factor: d_delta[k] = sum_f w(f, k) * hm_low[f] * g_low[f]
__global__ __launch_bounds__(256) void tail_key_reduce_kernel(const float* __restrict__ g_low, const float* __restrict__ hm_low, int F, int Cd,
                                                              int64_t splane, int step, int video_mode, int total_key,
                                                              float* __restrict__ d_delta) {
  const int k = blockIdx.y / Cd, cd = blockIdx.y % Cd;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < splane; i += (int64_t)gridDim.x * 256) {
    float acc = 0.f;
    for (int f = 0; f < F; ++f) {
      int ka = 0, kb = 0;
      float wa = 1.f, wb = 0.f;
      if (video_mode == VS_VIDEO_REPEAT) {
        ka = f / step;
      } else if (video_mode == VS_VIDEO_ALTERNATE) {
        ka = f / step;
        wa = (f % step) == 0 ? 1.f : 0.f;
      } else {
        const int ninter = ((F - 1) / step) * step;
        if (f < ninter) {
          ka = f / step; kb = ka + 1;
          const int j = f % step;
          const float lin = step > 1 ? (float)j / (float)(step - 1) : 0.f;
          wa = 1.f - lin; wb = 1.f - wa;
        } else {
          ka = total_key - 1;
        }
      }
      if (ka >= total_key) ka = total_key - 1;
      if (kb >= total_key) kb = total_key - 1;
      float w = 0.f;
      if (ka == k) w += wa;
      if (kb == k && wb != 0.f) w += wb;
      if (w == 0.f) continue;
      const float hm = hm_low ? hm_low[(int64_t)f * splane + i] : 1.f;
      acc += w * (hm * g_low[((int64_t)f * Cd + cd) * splane + i]);
    }
    d_delta[((int64_t)k * Cd + cd) * splane + i] = acc;
  }
}

// ---- colour augmentations (aug.hip::color_kernel): y = clamp01(f * x + (1 - f) * ref), ref = 0 / frame mean of gray / gray
enum { OP_BRIGHTNESS = 0, OP_CONTRAST = 1, OP_SATURATION = 2, OP_HUE = 3, OP_GRAYSCALE = 4 };
__device__ __forceinline__ float tv_gray(float r, float g, float b) { return 0.2989f * r + 0.587f * g + 0.114f * b; }
__device__ __forceinline__ bool pass01(float v) { return v >= 0.f && v <= 1.f; }

// masked gradient sum per frame for the contrast op (the frame mean couples every pixel): partial[f][block]
__global__ __launch_bounds__(256) void color_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t plane, float factor,
                                                                const float* __restrict__ means, double* __restrict__ partial) {
  __shared__ double red[256];
  const int f = blockIdx.y;
  const float* s = x + (int64_t)f * 3 * plane;
  const float* g = dy + (int64_t)f * 3 * plane;
  const float m = (1.0f - factor) * means[f];
  double acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (int64_t)gridDim.x * 256)
    for (int c = 0; c < 3; ++c)
      if (pass01(factor * s[c * plane + i] + m)) acc += (double)g[c * plane + i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(int64_t)f * gridDim.x + blockIdx.x] = red[0];
}
__global__ void color_bwd_finish_kernel(const double* __restrict__ partial, int nblk, int F, float* __restrict__ sums) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  double s = 0;
  for (int k = 0; k < nblk; ++k) s += partial[(int64_t)f * nblk + k];
  sums[f] = (float)s;
}
__global__ __launch_bounds__(256) void color_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t plane,
                                                        int op, float factor, const float* __restrict__ means, const float* __restrict__ sums) {
  const int f = blockIdx.y;
  const float* s = x + (int64_t)f * 3 * plane;
  const float* g = dy + (int64_t)f * 3 * plane;
  float* d = dx + (int64_t)f * 3 * plane;
  const float gw[3] = {0.2989f, 0.587f, 0.114f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (int64_t)gridDim.x * 256) {
    const float v[3] = {s[i], s[plane + i], s[2 * plane + i]};
    const float gy[3] = {g[i], g[plane + i], g[2 * plane + i]};
    float o[3];
    switch (op) {
      case OP_BRIGHTNESS:
        for (int c = 0; c < 3; ++c) o[c] = pass01(factor * v[c]) ? factor * gy[c] : 0.f;
        break;
      case OP_CONTRAST: {
        const float m = (1.0f - factor) * means[f];
        const float coup = (1.0f - factor) * sums[f] / (float)plane;            // d mean(gray) / d x_c = gw[c] / plane
        for (int c = 0; c < 3; ++c) o[c] = (pass01(factor * v[c] + m) ? factor * gy[c] : 0.f) + gw[c] * coup;
      } break;
      case OP_SATURATION: {
        const float m = (1.0f - factor) * tv_gray(v[0], v[1], v[2]);
        float pg[3], tot = 0.f;
        for (int c = 0; c < 3; ++c) { pg[c] = pass01(factor * v[c] + m) ? gy[c] : 0.f; tot += pg[c]; }
        for (int c = 0; c < 3; ++c) o[c] = factor * pg[c] + (1.0f - factor) * gw[c] * tot;
      } break;
      default: {                                                                 // grayscale: y_c = 0.299 r + 0.587 g + 0.114 b for every c
        const float tot = gy[0] + gy[1] + gy[2];
        o[0] = 0.299f * tot; o[1] = 0.587f * tot; o[2] = 0.114f * tot;
      } break;
    }
    d[i] = o[0]; d[plane + i] = o[1]; d[2 * plane + i] = o[2];
  }
}

// ---- JPEG.forward's clamp in front of the straight-through estimator (valuemetric.py:41): dx = dy where 0 <= x <= 1
__global__ __launch_bounds__(256) void clamp01_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dx[i] = pass01(x[i]) ? dy[i] : 0.f;
}

// ---- perceptual mse / yuv term: loss = mean((T (imgs_w - imgs))^2) over F * 3 * plane elements, T = identity or the BT.601 matrix of
// data/transforms.py:46-48;  d loss / d imgs_w = 2 T^T T e / n.  partial [grid] doubles -> loss[0].
struct Yuv { float m[9]; };
__device__ __forceinline__ void percep_err(const float* a, const float* b, int64_t plane, int64_t i, int yuv, const Yuv& M, float (&t)[3], float (&e)[3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) e[c] = b[c * plane + i] - a[c * plane + i];
  if (yuv) {
#pragma unroll
    for (int r = 0; r < 3; ++r) t[r] = M.m[3 * r] * e[0] + M.m[3 * r + 1] * e[1] + M.m[3 * r + 2] * e[2];
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) t[c] = e[c];
  }
}
__global__ __launch_bounds__(256) void percep_partial_kernel(const float* __restrict__ imgs, const float* __restrict__ imgs_w, int F, int64_t plane,
                                                             int yuv, Yuv M, double* __restrict__ partial) {
  __shared__ double red[256];
  double acc = 0;
  const int64_t n = (int64_t)F * plane;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (int64_t)gridDim.x * 256) {
    const int64_t f = q / plane, i = q - f * plane;
    float t[3], e[3];
    percep_err(imgs + f * 3 * plane, imgs_w + f * 3 * plane, plane, i, yuv, M, t, e);
    acc += (double)t[0] * t[0] + (double)t[1] * t[1] + (double)t[2] * t[2];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void percep_finish_kernel(const double* __restrict__ partial, int nblk, double n_elem, float* __restrict__ loss) {
  __shared__ double red[256];
  double acc = 0;
  for (int k = threadIdx.x; k < nblk; k += 256) acc += partial[k];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(red[0] / n_elem);
}
__global__ __launch_bounds__(256) void percep_grad_kernel(const float* __restrict__ imgs, const float* __restrict__ imgs_w, int F, int64_t plane, int yuv,
                                                          Yuv M, float gscale, float* __restrict__ d_imgs_w) {
  const int64_t n = (int64_t)F * plane;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (int64_t)gridDim.x * 256) {
    const int64_t f = q / plane, i = q - f * plane;
    float t[3], e[3];
    percep_err(imgs + f * 3 * plane, imgs_w + f * 3 * plane, plane, i, yuv, M, t, e);
    float* d = d_imgs_w + f * 3 * plane;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = yuv ? (M.m[c] * t[0] + M.m[3 + c] * t[1] + M.m[6 + c] * t[2]) : t[c];       // T^T t
      d[c * plane + i] = gscale * v;
    }
  }
}

}  // namespace

// ===================================================================================================== C-ABI
extern "C" int vs_resize_nchw_bwd(const float* dy, float* dx, int planes, int H, int W, int oh, int ow, int antialias, float* tmp, void* stream) {
  // forward: vs_resize_nchw(src [planes][H][W] -> dst [planes][oh][ow]); here dy is [planes][oh][ow], dx [planes][H][W], tmp planes * oh * W floats
  VS_REQUIRE(dy && dx && tmp && planes > 0 && H > 0 && W > 0 && oh > 0 && ow > 0);
  hipLaunchKernelGGL(resize_bwd_pass_kernel, dim3((W + 63) / 64, (oh + 3) / 4, planes), dim3(256), 0, (hipStream_t)stream, dy, tmp, W, ow, oh, 0, antialias);
  hipLaunchKernelGGL(resize_bwd_pass_kernel, dim3((W + 63) / 64, (H + 3) / 4, planes), dim3(256), 0, (hipStream_t)stream, tmp, dx, H, oh, W, 1, antialias);
  return vs_launch_status();
}

extern "C" int vs_aug_crop_flip_bwd(const float* dy, float* dx, int planes, int H, int W, int i0, int j0, int h, int w, int flip, void* stream) {
  VS_REQUIRE(dy && dx && planes > 0 && H > 0 && W > 0 && h > 0 && w > 0);
  hipLaunchKernelGGL(crop_flip_bwd_kernel, dim3(gridx((int64_t)H * W), planes), dim3(256), 0, (hipStream_t)stream, dy, dx, H, W, i0, j0, h, w, flip);
  return vs_launch_status();
}

extern "C" int vs_nhwc_to_nchw_scaled(const float* src, int F, int H, int W, int C, int64_t ld, float mul, float* dst, void* stream) {
  VS_REQUIRE(src && dst && F > 0 && H > 0 && W > 0 && C > 0 && ld >= C);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(gridx((int64_t)H * W), F), dim3(256), 0, (hipStream_t)stream, src, (int64_t)H * W, C, ld, mul, dst);
  return vs_launch_status();
}

extern "C" int vs_mask_mul(const float* dy, const float* mask, float* dx, int F, int C, int H, int W, int complement, void* stream) {
  VS_REQUIRE(dy && mask && dx && F > 0 && C > 0 && H > 0 && W > 0);
  hipLaunchKernelGGL(mask_mul_kernel, dim3(gridx((int64_t)H * W), F * C), dim3(256), 0, (hipStream_t)stream, dy, mask, dx, C, (int64_t)H * W, complement);
  return vs_launch_status();
}

extern "C" int vs_embed_tail_bwd(const float* imgs, const float* preds, const float* hmap_full, const float* d_imgs_w, const float* d_preds_w, int F,
                                 int H, int W, int Cd, int clamp, float scaling_i, float scaling_w, float* g_full, void* stream) {
  VS_REQUIRE(imgs && preds && g_full && (d_imgs_w || d_preds_w) && F > 0 && H > 0 && W > 0 && (Cd == 1 || Cd == 3));
  TailBwdArgs a{imgs, preds, hmap_full, d_imgs_w, d_preds_w, g_full, F, Cd, clamp, (int64_t)H * W, scaling_i, scaling_w};
  hipLaunchKernelGGL(embed_tail_bwd_kernel, dim3(gridx((int64_t)H * W), F), dim3(256), 0, (hipStream_t)stream, a);
  return vs_launch_status();
}

extern "C" int vs_tail_key_reduce(const float* g_low, const float* hmap_low, int F, int Cd, int S_h, int S_w, int step, int video_mode, int total_key,
                                  float* d_delta, void* stream) {
  VS_REQUIRE(g_low && d_delta && F > 0 && (Cd == 1 || Cd == 3) && S_h > 0 && S_w > 0 && step >= 1 && total_key >= 1 && video_mode >= 0 && video_mode <= 2);
  hipLaunchKernelGGL(tail_key_reduce_kernel, dim3(gridx((int64_t)S_h * S_w), total_key * Cd), dim3(256), 0, (hipStream_t)stream, g_low, hmap_low, F, Cd,
                     (int64_t)S_h * S_w, step, video_mode, total_key, d_delta);
  return vs_launch_status();
}

extern "C" int64_t vs_aug_color_bwd_scratch_floats(int F, int H, int W) {
  const int64_t nblk = gridx((int64_t)H * W, 256, 256);
  return 2 * (int64_t)F * nblk + 2 * (int64_t)F + 8;        // doubles as 2 floats each + the per-frame sums and means
}

extern "C" int vs_aug_color_bwd(const float* x, const float* dy, float* dx, int F, int H, int W, int op, float factor, const float* means, float* scratch,
                                void* stream) {
  // `means`: the per-frame gray means the forward pass used (vs_aug_color's scratch, first F floats); only the contrast op reads it
  VS_REQUIRE(x && dy && dx && F > 0 && H > 0 && W > 0 && op >= 0 && op <= 4 && op != OP_HUE);
  const int64_t plane = (int64_t)H * W;
  float* sums = nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (op == OP_CONTRAST) {
    VS_REQUIRE(means && scratch && (((uintptr_t)scratch) & 7) == 0);
    const int nblk = (int)gridx(plane, 256, 256);
    double* partial = reinterpret_cast<double*>(scratch);
    sums = scratch + 2 * (int64_t)F * nblk;
    hipLaunchKernelGGL(color_bwd_partial_kernel, dim3(nblk, F), dim3(256), 0, st, x, dy, plane, factor, means, partial);
    hipLaunchKernelGGL(color_bwd_finish_kernel, dim3((F + 63) / 64), dim3(64), 0, st, partial, nblk, F, sums);
  }
  hipLaunchKernelGGL(color_bwd_kernel, dim3(gridx(plane), F), dim3(256), 0, st, x, dy, dx, plane, op, factor, means, sums);
  return vs_launch_status();
}

extern "C" int vs_clamp01_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
  VS_REQUIRE(x && dy && dx && n > 0);
  hipLaunchKernelGGL(clamp01_bwd_kernel, dim3(gridx(n)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n);
  return vs_launch_status();
}

static Yuv yuv_matrix() {
  return Yuv{{0.299f, 0.587f, 0.114f, -0.14713f, -0.28886f, 0.436f, 0.615f, -0.51499f, -0.10001f}};
}

extern "C" int64_t vs_percep_partial_doubles(int F, int H, int W) { return gridx((int64_t)F * H * W, 256, 1024); }

extern "C" int vs_percep_mse(const float* imgs, const float* imgs_w, int F, int H, int W, int yuv, double* partial, float* loss, void* stream) {
  VS_REQUIRE(imgs && imgs_w && partial && loss && F > 0 && H > 0 && W > 0);
  const int64_t plane = (int64_t)H * W;
  const int nblk = (int)gridx((int64_t)F * plane, 256, 1024);
  hipLaunchKernelGGL(percep_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, imgs, imgs_w, F, plane, yuv, yuv_matrix(), partial);
  hipLaunchKernelGGL(percep_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblk, 3.0 * (double)F * (double)plane, loss);
  return vs_launch_status();
}

extern "C" int vs_percep_mse_grad(const float* imgs, const float* imgs_w, int F, int H, int W, int yuv, float upstream, float* d_imgs_w, void* stream) {
  VS_REQUIRE(imgs && imgs_w && d_imgs_w && F > 0 && H > 0 && W > 0);
  const int64_t plane = (int64_t)H * W;
  const float gs = upstream * 2.0f / (3.0f * (float)F * (float)plane);
  hipLaunchKernelGGL(percep_grad_kernel, dim3(gridx((int64_t)F * plane)), dim3(256), 0, (hipStream_t)stream, imgs, imgs_w, F, plane, yuv, yuv_matrix(), gs,
                     d_imgs_w);
  return vs_launch_status();
}
