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
// ---- Hue (torchvision adjust_hue = _rgb2hsv -> (h + factor) % 1 -> _hsv2rgb, valuemetric.py:168-171): the chain rule through exactly the
// operations of aug.hip::hue_shift, with autograd's conventions: max / min send their gradient to the arg-max / arg-min channel (first index on
// ties), comparisons and floor carry none, fmod / % pass it through, clamp passes it inside [0, 1] (bounds included).
__device__ __forceinline__ void hue_bwd(const float (&x)[3], const float (&gy)[3], float factor, float (&dx)[3]) {
  const float r = x[0], g = x[1], b = x[2];
  const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
  const int imax = (r == maxc) ? 0 : ((g == maxc) ? 1 : 2);
  const int imin = (r == minc) ? 0 : ((g == minc) ? 1 : 2);
  const bool eqc = maxc == minc;
  const float cr = maxc - minc;
  const float sdiv = eqc ? 1.f : maxc, div = eqc ? 1.f : cr;
  const float s = cr / sdiv;
  const float rc = (maxc - r) / div, gc = (maxc - g) / div, bc = (maxc - b) / div;
  const int branch = (maxc == r) ? 0 : ((maxc == g) ? 1 : 2);
  const float hsum = branch == 0 ? (bc - gc) : (branch == 1 ? (2.0f + rc - bc) : (4.0f + gc - rc));
  float h = fmodf(hsum / 6.0f + 1.0f, 1.0f);
  const float v = maxc;
  h = h + factor;
  h = h - floorf(h);
  const float h6 = h * 6.0f;
  const float fi = floorf(h6);
  const float f = h6 - fi;
  int i = (int)fi;
  i = ((i % 6) + 6) % 6;
  const float pu = v * (1.0f - s), qu = v * (1.0f - s * f), tu = v * (1.0f - s * (1.0f - f));
  // gradients of the selected outputs
  float dv = 0.f, dp = 0.f, dq = 0.f, dt = 0.f;
  switch (i) {
    case 0: dv = gy[0]; dt = gy[1]; dp = gy[2]; break;
    case 1: dq = gy[0]; dv = gy[1]; dp = gy[2]; break;
    case 2: dp = gy[0]; dv = gy[1]; dt = gy[2]; break;
    case 3: dp = gy[0]; dq = gy[1]; dv = gy[2]; break;
    case 4: dt = gy[0]; dp = gy[1]; dv = gy[2]; break;
    default: dv = gy[0]; dp = gy[1]; dq = gy[2]; break;
  }
  if (!pass01(pu)) dp = 0.f;
  if (!pass01(qu)) dq = 0.f;
  if (!pass01(tu)) dt = 0.f;
  float ds = 0.f, df = 0.f;
  dv += dp * (1.0f - s) + dq * (1.0f - s * f) + dt * (1.0f - s * (1.0f - f));
  ds += -dp * v - dq * v * f - dt * v * (1.0f - f);
  df += -dq * v * s + dt * v * s;
  const float dhsum = df;                              // f = 6 h - i, h = (hsum / 6 + 1) mod 1 (+ factor, mod 1)
  float drc = 0.f, dgc = 0.f, dbc = 0.f;
  if (branch == 0) { dbc += dhsum; dgc -= dhsum; }
  else if (branch == 1) { drc += dhsum; dbc -= dhsum; }
  else { dgc += dhsum; drc -= dhsum; }
  float dmax = dv, dmin = 0.f, dcr = 0.f;
  dx[0] = -drc / div; dx[1] = -dgc / div; dx[2] = -dbc / div;
  dmax += (drc + dgc + dbc) / div;
  if (!eqc) {
    dcr -= (drc * (maxc - r) + dgc * (maxc - g) + dbc * (maxc - b)) / (div * div);
    dmax -= ds * cr / (maxc * maxc);
  }
  dcr += ds / sdiv;
  dmax += dcr; dmin -= dcr;
  dx[imax] += dmax;
  dx[imin] += dmin;
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
      case OP_HUE:
        hue_bwd(v, gy, factor, o);
        break;
      default: {                                                                 // grayscale: y_c = 0.299 r + 0.587 g + 0.114 b for every c
        const float tot = gy[0] + gy[1] + gy[2];
        o[0] = 0.299f * tot; o[1] = 0.587f * tot; o[2] = 0.114f * tot;
      } break;
    }
    d[i] = o[0]; d[plane + i] = o[1]; d[2 * plane + i] = o[2];
  }
}

// ---- GaussianBlur (torchvision gaussian_blur: reflection padding k / 2, separable kernel; valuemetric.py:108-128): adjoint of one 1-d pass,
// gather form.  Forward y[o] = sum_t w[t] x[refl(o + t - r)]; the padded position i = o + t - r collects dxp[i] = sum_t w[t] dy[i - t + r]
// (outputs outside the row contribute nothing), and position u of the row receives dxp[u] + dxp[-u] (1 <= u <= r) + dxp[2 (n - 1) - u]
// (n - 1 - r <= u <= n - 2): the adjoint of the reflection.
struct GaussW { float w[33]; int k; };
__global__ __launch_bounds__(256) void blur_bwd_pass_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, GaussW g, int vertical) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= W || y >= H) return;
  const float* sp = src + (int64_t)blockIdx.z * H * W;
  const int r = g.k / 2, n = vertical ? H : W, u = vertical ? y : x;
  const int64_t stride = vertical ? W : 1;
  const float* line = sp + (vertical ? x : (int64_t)y * W);
  auto dxp = [&](int i) -> float {
    float a = 0.f;
    for (int t = 0; t < g.k; ++t) {
      const int o = i - t + r;
      if (o >= 0 && o < n) a += g.w[t] * line[(int64_t)o * stride];
    }
    return a;
  };
  float acc = dxp(u);
  if (u >= 1 && u <= r) acc += dxp(-u);
  if (u >= n - 1 - r && u <= n - 2) acc += dxp(2 * (n - 1) - u);
  dst[((int64_t)blockIdx.z * H + y) * W + x] = acc;
}

// ---- Rotate (nearest) / Perspective (bilinear): adjoint of aug.hip::warp_kernel (torchvision's affine / perspective grid + grid_sample with zero
// padding, align_corners = False; geometric.py:28-59, 127-183), gather form: thread = input pixel u.  `inv` maps the pixel-centre coordinates of
// the INPUT to those of the output (the 3 x 3 inverse of the sampling map, computed by the host in fp64): it only bounds the search -- every output
// pixel of the bounding box of u's 2 x 2 footprint (+- 1.5) is re-sampled with the forward kernel's own arithmetic and contributes with the
// weight the forward gave u (nearest: 1 if it rounded to u).
struct WarpB { float t[8]; float inv[9]; int kind, bilinear; };
__device__ __forceinline__ void warp_sample(const WarpB& a, int ox, int oy, int ow, int oh, int W, int H, float& ix, float& iy) {
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
  ix = ((gx + 1.f) * W - 1.f) / 2.f;
  iy = ((gy + 1.f) * H - 1.f) / 2.f;
}
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int oh, int ow, WarpB a) {
  const int ux = blockIdx.x * 32 + (threadIdx.x & 31), uy = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (ux >= W || uy >= H) return;
  const float* gp = dy + (int64_t)blockIdx.z * oh * ow;
  // bounding box, in output pixels, of the input square [ux - 1, ux + 1] x [uy - 1, uy + 1]
  float lox = 1e30f, hix = -1e30f, loy = 1e30f, hiy = -1e30f;
  bool whole = false;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float px = (float)ux + 0.5f + ((c & 1) ? 1.f : -1.f), py = (float)uy + 0.5f + ((c & 2) ? 1.f : -1.f);
    const float cw = a.inv[6] * px + a.inv[7] * py + a.inv[8];
    if (!(cw > 1e-6f)) { whole = true; continue; }
    const float cx = (a.inv[0] * px + a.inv[1] * py + a.inv[2]) / cw - 0.5f, cy = (a.inv[3] * px + a.inv[4] * py + a.inv[5]) / cw - 0.5f;
    lox = fminf(lox, cx); hix = fmaxf(hix, cx); loy = fminf(loy, cy); hiy = fmaxf(hiy, cy);
  }
  int x0 = 0, x1 = ow - 1, y0 = 0, y1 = oh - 1;
  if (!whole) {
    x0 = max(0, (int)floorf(lox - 1.5f)); x1 = min(ow - 1, (int)ceilf(hix + 1.5f));
    y0 = max(0, (int)floorf(loy - 1.5f)); y1 = min(oh - 1, (int)ceilf(hiy + 1.5f));
  }
  float acc = 0.f;
  for (int oy = y0; oy <= y1; ++oy)
    for (int ox = x0; ox <= x1; ++ox) {
      float ix, iy;
      warp_sample(a, ox, oy, ow, oh, W, H, ix, iy);
      float wgt = 0.f;
      if (!a.bilinear) {
        if (nearbyintf(ix) == (float)ux && nearbyintf(iy) == (float)uy) wgt = 1.f;
      } else {
        const float fx = floorf(ix), fy = floorf(iy);
        const float wx = fx == (float)ux ? (fx + 1.f - ix) : (fx + 1.f == (float)ux ? (ix - fx) : 0.f);
        const float wy = fy == (float)uy ? (fy + 1.f - iy) : (fy + 1.f == (float)uy ? (iy - fy) : 0.f);
        wgt = wx * wy;
      }
      if (wgt != 0.f) acc += wgt * gp[(int64_t)oy * ow + ox];
    }
  dx[((int64_t)blockIdx.z * H + uy) * W + ux] = acc;
}

// ---- whole-frame gathers (DropFrame / SpeedChange / TemporalReorder, video.py:283-313, 319-405, 491-529: dst[o] = src[idx[o]]): adjoint in gather
// form.  The host lists, per SOURCE frame f, the outputs that copied it (CSR: start[f] .. start[f + 1] into `outs`, ascending): dx[f] = their sum.
__global__ __launch_bounds__(256) void gather_frames_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ start, const int32_t* __restrict__ outs,
                                                                float* __restrict__ dx, int64_t fsz) {
  const int64_t f = blockIdx.y;
  const int a = start[f], b = start[f + 1];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < fsz; i += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int k = a; k < b; ++k) s += dy[(int64_t)outs[k] * fsz + i];
    dx[f * fsz + i] = s;
  }
}
// ---- WindowAveraging (video.py:411-486): y[i] = (1 - alpha) x[i] + alpha mean(x[a_i .. b_i)) with the clipped window of aug.hip::window_average_kernel;
// adjoint: dx[k] = (1 - alpha) dy[k] + alpha sum over the frames i whose window holds k of dy[i] / (b_i - a_i), ascending i
__global__ __launch_bounds__(256) void window_average_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int F, int64_t fsz, int hw, float alpha) {
  const int64_t k = blockIdx.y;
  const int lo = (int)(k - hw < 0 ? 0 : k - hw), hi = (int)(k + hw + 1 > F ? F : k + hw + 1);       // frames i with |i - k| <= hw
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < fsz; e += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int i = lo; i < hi; ++i) {
      const int a = i - hw < 0 ? 0 : i - hw, b = i + hw + 1 > F ? F : i + hw + 1;
      s += dy[(int64_t)i * fsz + e] / (float)(b - a);
    }
    dx[k * fsz + e] = (1.f - alpha) * dy[k * fsz + e] + alpha * s;
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
  VS_REQUIRE(x && dy && dx && F > 0 && H > 0 && W > 0 && op >= 0 && op <= 4);
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

extern "C" int vs_gaussian_blur_bwd(const float* dy, float* tmp, float* dx, int planes, int H, int W, int k, float sigma, void* stream) {
  VS_REQUIRE(dy && tmp && dx && planes > 0 && H > 0 && W > 0 && k >= 1 && (k & 1) && k <= 33 && sigma > 0.f && k / 2 < H && k / 2 < W);
  GaussW g; g.k = k;
  float sum = 0.f;                       // the weights of aug.hip::vs_gaussian_blur (torchvision _get_gaussian_kernel1d), same arithmetic
  const float half = (k - 1) * 0.5f;
  for (int i = 0; i < k; ++i) {
    const float x = (k == 1) ? 0.f : (-half + (2.f * half) * (float)i / (float)(k - 1));
    g.w[i] = expf(-0.5f * (x / sigma) * (x / sigma));
    sum += g.w[i];
  }
  for (int i = 0; i < k; ++i) g.w[i] /= sum;
  dim3 grid((W + 31) / 32, (H + 7) / 8, planes);
  // the forward runs the horizontal pass, then the vertical one: adjoint in the reverse order
  hipLaunchKernelGGL(blur_bwd_pass_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, tmp, H, W, g, 1);
  hipLaunchKernelGGL(blur_bwd_pass_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float*)tmp, dx, H, W, g, 0);
  return vs_launch_status();
}

extern "C" int vs_aug_warp_bwd(const float* dy, float* dx, int planes, int H, int W, int oh, int ow, int kind, const float* coeffs, int bilinear,
                               const float* inv, void* stream) {
  VS_REQUIRE(dy && dx && coeffs && inv && planes > 0 && H > 0 && W > 0 && oh > 0 && ow > 0 && (kind == 0 || kind == 1));
  WarpB a;
  for (int i = 0; i < 8; ++i) a.t[i] = i < (kind == 0 ? 6 : 8) ? coeffs[i] : 0.f;
  for (int i = 0; i < 9; ++i) a.inv[i] = inv[i];
  a.kind = kind; a.bilinear = bilinear;
  hipLaunchKernelGGL(warp_bwd_kernel, dim3((W + 31) / 32, (H + 7) / 8, planes), dim3(256), 0, (hipStream_t)stream, dy, dx, H, W, oh, ow, a);
  return vs_launch_status();
}

extern "C" int vs_aug_gather_frames_bwd(const float* dy, const int32_t* start, const int32_t* outs, float* dx, int n_src, int64_t frame_floats, void* stream) {
  VS_REQUIRE(dy && start && outs && dx && n_src > 0 && frame_floats > 0);
  hipLaunchKernelGGL(gather_frames_bwd_kernel, dim3(gridx(frame_floats), n_src), dim3(256), 0, (hipStream_t)stream, dy, start, outs, dx, frame_floats);
  return vs_launch_status();
}

extern "C" int vs_aug_window_average_bwd(const float* dy, float* dx, int F, int64_t frame_floats, int half_window, float alpha, void* stream) {
  VS_REQUIRE(dy && dx && F > 0 && frame_floats > 0 && half_window >= 0);
  hipLaunchKernelGGL(window_average_bwd_kernel, dim3(gridx(frame_floats), F), dim3(256), 0, (hipStream_t)stream, dy, dx, F, frame_floats, half_window, alpha);
  return vs_launch_status();
}
