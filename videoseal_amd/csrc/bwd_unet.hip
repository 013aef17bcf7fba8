// Backward building blocks of the U-Net embedder under model.train() (the generator side of train.py:626-643):
// BatchNorm-on-batch-statistics + ReLU, the stride-2 down convs, the Upsample groups, the message table, the output conv.  gfx950 only.
// As in bwd_ops.hip, backward-DATA products are launches of the forward conv / GEMM kernels on flipped / transposed weights and weight
// gradients are vs_gemm_wgrad over patch matrices; this file holds the adjoints that have no forward counterpart.  Reductions are
// deterministic (fixed chunking and order, fp64 across chunks).
// Index formulas checked in float64 against autograd on the CPU (tools/check_bwd_formulas.py); kernels against autograd on the GPU
// (tests/test_gpu_bwd_unet.py).
#include "vs_common.h"

namespace {

constexpr int CR_ROWS = 256;

struct BnBwd { const float* mean; const float* rstd; const float* scale; const float* shift; };

// column sums of the BatchNorm backward over chunks of CR_ROWS rows: p0 = sum g * xhat, p1 = sum g, with xhat = (raw - mean) * rstd and
// g = dy masked by the ReLU that followed the normalisation in the forward (y = raw * scale + shift > 0)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ raw, int64_t ld, const float* __restrict__ dy, int64_t dy_ld,
                                                             BnBwd bp, int relu, int64_t rows, int C4, int64_t ldp, float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float red[2][256][4];
  const int64_t r0 = (int64_t)blockIdx.x * CR_ROWS;
  const int64_t r1 = r0 + CR_ROWS < rows ? r0 + CR_ROWS : rows;
  const int G = C4 < 256 ? C4 : 256;
  const int RL = 256 / G;
  const int g = threadIdx.x % G, rl = threadIdx.x / G;
  for (int gb = 0; gb < C4; gb += G) {
    const int gg = gb + g;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    if (rl < RL && gg < C4) {
      const f32x4 mu = *reinterpret_cast<const f32x4*>(bp.mean + 4 * gg), rs = *reinterpret_cast<const f32x4*>(bp.rstd + 4 * gg);
      const f32x4 sc = *reinterpret_cast<const f32x4*>(bp.scale + 4 * gg), sh = *reinterpret_cast<const f32x4*>(bp.shift + 4 * gg);
      for (int64_t r = r0 + rl; r < r1; r += RL) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(raw + r * ld + 4 * gg);
        f32x4 d = *reinterpret_cast<const f32x4*>(dy + r * dy_ld + 4 * gg);
        if (relu)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (!(__builtin_fmaf(v[e], sc[e], sh[e]) > 0.f)) d[e] = 0.f;
        a0 += d * ((v - mu) * rs);
        a1 += d;
      }
    }
    *reinterpret_cast<f32x4*>(&red[0][threadIdx.x][0]) = a0;
    *reinterpret_cast<f32x4*>(&red[1][threadIdx.x][0]) = a1;
    __syncthreads();
    if (rl == 0 && gg < C4) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < RL; ++j) s += *reinterpret_cast<const f32x4*>(&red[k][j * G + g][0]);
        *reinterpret_cast<f32x4*>(partial + ((int64_t)blockIdx.x * 2 + k) * ldp + 4 * gg) = s;
      }
    }
    __syncthreads();
  }
}
// sums = [sum g * xhat (ldp) | sum g (ldp) | rows] in fp64 (the vector SyncBatchNorm all-reduces in its backward); dgamma / dbeta = the LOCAL sums.
// One wave per channel: the lanes stride the chunks, then a butterfly (fixed order).
__global__ __launch_bounds__(64) void bn_bwd_reduce_kernel(const float* __restrict__ partial, int nchunk, int64_t rows, int64_t ldp, int C,
                                                           double* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int64_t c = blockIdx.x;
  double s = 0, q = 0;
  for (int k = threadIdx.x; k < nchunk; k += 64) { s += (double)partial[((int64_t)k * 2 + 0) * ldp + c]; q += (double)partial[((int64_t)k * 2 + 1) * ldp + c]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  if (threadIdx.x != 0) return;
  if (c == 0) sums[2 * ldp] = (double)rows;
  sums[c] = s;
  sums[ldp + c] = q;
  if (c < C) { dgamma[c] = (float)s; dbeta[c] = (float)q; }
}
// d raw = scale * (g - mean_rows(g) - xhat * mean_rows(g * xhat));  the means come from `sums` (global over the ranks after the all-reduce)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ raw, int64_t ld, const float* __restrict__ dy, int64_t dy_ld,
                                                           BnBwd bp, int relu, const double* __restrict__ sums, int64_t ldp, int C, int O4,
                                                           int64_t total, float* __restrict__ dx, int64_t dx_ld) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cg = (int)(idx % O4);
  const int64_t r = idx / O4;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  if (4 * cg < C) {
    const float inv_n = (float)(1.0 / sums[2 * ldp]);
    const f32x4 v = *reinterpret_cast<const f32x4*>(raw + r * ld + 4 * cg);
    const f32x4 d = *reinterpret_cast<const f32x4*>(dy + r * dy_ld + 4 * cg);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * cg + e;
      if (c >= C) break;
      const float sc = bp.scale[c];
      float g = d[e];
      if (relu && !(__builtin_fmaf(v[e], sc, bp.shift[c]) > 0.f)) g = 0.f;
      const float xh = (v[e] - bp.mean[c]) * bp.rstd[c];
      o[e] = sc * (g - (float)sums[ldp + c] * inv_n - xh * ((float)sums[c] * inv_n));
    }
  }
  *reinterpret_cast<f32x4*>(dx + r * dx_ld + 4 * cg) = o;
}
// mean / rstd of the forward's batch statistics from its (all-reduced) moment vector [sum x | sum x^2 | rows] (vs_bn_partial_sums)
__global__ __launch_bounds__(256) void bn_mean_rstd_kernel(const double* __restrict__ sums, int C, int64_t ld, float eps, float* __restrict__ mean,
                                                           float* __restrict__ rstd) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= (int)ld) return;
  float m = 0.f, r = 0.f;
  if (c < C) {
    const double n = sums[2 * ld];
    const double mu = sums[c] / n;
    double var = sums[ld + c] / n - mu * mu;
    if (var < 0) var = 0;
    m = (float)mu;
    r = (float)(1.0 / sqrt(var + (double)eps));
  }
  mean[c] = m;
  rstd[c] = r;
}

// ---------------------------------------------------------------------------------------------------
// stride-2 conv (pad 1): backward-data = the flipped conv over the zero-dilated gradient, out[b][2 oy][2 ox] = dy[b][oy][ox], zeros elsewhere
__global__ __launch_bounds__(256) void dilate2_kernel(const float* __restrict__ dy, int Ho, int Wo, int64_t ld, int H, int W, int64_t total,
                                                      float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (pixel of the H x W map, c4)
  if (idx >= total) return;
  const int C4 = (int)(ld / 4);
  const int c4 = (int)(idx % C4);
  const int64_t pix = idx / C4;
  const int x = (int)(pix % W);
  const int64_t t = pix / W;
  const int y = (int)(t % H);
  const int64_t b = t / H;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (!(y & 1) && !(x & 1) && (y >> 1) < Ho && (x >> 1) < Wo) v = *reinterpret_cast<const f32x4*>(dy + ((b * Ho + (y >> 1)) * Wo + (x >> 1)) * ld + 4 * c4);
  *reinterpret_cast<f32x4*>(out + pix * ld + 4 * c4) = v;
}
// Adjoint of reflection padding 1 (the Upsample conv of unet.py:170-197 pads its input by reflection): dxp is the gradient on the padded
// (H + 2) x (W + 2) map, out[y][x] = dxp[y + 1][x + 1] + the ring pixels that mirror onto (y, x): padded row 0 is row 1 again, padded row
// H + 1 is row H - 2 (same for the columns).  Gather form, fixed order.
__global__ __launch_bounds__(256) void reflect_fold_kernel(const float* __restrict__ dxp, int H, int W, int64_t ld, int64_t total, float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (pixel of the H x W map, c4)
  if (idx >= total) return;
  const int C4 = (int)(ld / 4);
  const int c4 = (int)(idx % C4);
  const int64_t pix = idx / C4;
  const int x = (int)(pix % W);
  const int64_t t = pix / W;
  const int y = (int)(t % H);
  const int64_t b = t / H;
  int ys[3], xs[3], ny = 0, nx = 0;
  ys[ny++] = y + 1;
  if (y == 1) ys[ny++] = 0;
  if (y == H - 2) ys[ny++] = H + 1;
  xs[nx++] = x + 1;
  if (x == 1) xs[nx++] = 0;
  if (x == W - 2) xs[nx++] = W + 1;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < ny; ++i)
    for (int j = 0; j < nx; ++j) acc += *reinterpret_cast<const f32x4*>(dxp + ((b * (H + 2) + ys[i]) * (W + 2) + xs[j]) * ld + 4 * c4);
  *reinterpret_cast<f32x4*>(out + pix * ld + 4 * c4) = acc;
}
// dy [B, H, W] rows -> the interior of a zero canvas [B, H + 2, W + 2] (the backward-data conv of a padded input runs over the padded map)
__global__ __launch_bounds__(256) void pad_embed_kernel(const float* __restrict__ dy, int H, int W, int64_t ld, int64_t total, float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (pixel of the (H + 2) x (W + 2) canvas, c4)
  if (idx >= total) return;
  const int C4 = (int)(ld / 4);
  const int c4 = (int)(idx % C4);
  const int64_t pix = idx / C4;
  const int x = (int)(pix % (W + 2));
  const int64_t t = pix / (W + 2);
  const int y = (int)(t % (H + 2));
  const int64_t b = t / (H + 2);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (y >= 1 && y <= H && x >= 1 && x <= W) v = *reinterpret_cast<const f32x4*>(dy + ((b * H + y - 1) * W + x - 1) * ld + 4 * c4);
  *reinterpret_cast<f32x4*>(out + pix * ld + 4 * c4) = v;
}
// patch matrix of a 3x3 conv with zero padding 1 and stride s: cols[(b, oy, ox)][tap * ld + c] = x[b][oy s + ky - 1][ox s + kx - 1][c] or 0
__global__ __launch_bounds__(256) void im2col3x3_zs_kernel(const float* __restrict__ x, int H, int W, int64_t ld, int s, int Ho, int Wo, int64_t total,
                                                           float* __restrict__ cols) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (row_out, tap, c4)
  if (idx >= total) return;
  const int C4 = (int)(ld / 4);
  const int c4 = (int)(idx % C4);
  const int64_t t = idx / C4;
  const int tap = (int)(t % 9);
  const int64_t ro = t / 9;
  const int ox = (int)(ro % Wo);
  const int64_t t2 = ro / Wo;
  const int oy = (int)(t2 % Ho);
  const int64_t b = t2 / Ho;
  const int iy = oy * s + tap / 3 - 1, ix = ox * s + tap % 3 - 1;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const f32x4*>(x + ((b * H + iy) * W + ix) * ld + 4 * c4);
  *reinterpret_cast<f32x4*>(cols + ro * 9 * ld + (int64_t)tap * ld + 4 * c4) = v;
}

// ---------------------------------------------------------------------------------------------------
// adjoint of the bilinear x2 up-sampling (align_corners = False, source index clamped at 0 like ATen) of cat(x, skip * s): low-res pixel y
// collects from the high-res rows Y in [2y - 1, 2y + 2] whose two taps (i0 = floor(src), i1 = min(i0 + 1, H - 1)) include y.
__device__ __forceinline__ int up2_contrib(int y, int H, int (&Ys)[4], float (&ws)[4]) {
  int n = 0;
  for (int Y = 2 * y - 1; Y <= 2 * y + 2; ++Y) {
    if (Y < 0 || Y >= 2 * H) continue;
    float src = ((float)Y + 0.5f) * 0.5f - 0.5f;
    if (src < 0.f) src = 0.f;
    const int i0 = (int)src;
    const int i1 = i0 + 1 < H ? i0 + 1 : H - 1;
    const float f = src - (float)i0;
    const float w = (i0 == y ? 1.f - f : 0.f) + (i1 == y ? f : 0.f);
    if (w != 0.f) { Ys[n] = Y; ws[n] = w; ++n; }
  }
  return n;
}
__global__ __launch_bounds__(256) void upcat2x_bwd_kernel(const float* __restrict__ dhi, int64_t hi_ld, int H, int W, int C1, int C2, float skip_scale,
                                                          int64_t total, float* __restrict__ dx, int64_t ld1, float* __restrict__ dskip, int64_t ld2) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (low-res pixel, c4 over C1 + C2)
  if (idx >= total) return;
  const int C4 = (C1 + C2) / 4;
  const int c4 = (int)(idx % C4);
  const int64_t pix = idx / C4;
  const int x = (int)(pix % W);
  const int64_t t = pix / W;
  const int y = (int)(t % H);
  const int64_t b = t / H;
  int Ys[4], Xs[4];
  float wy[4], wx[4];
  const int ny = up2_contrib(y, H, Ys, wy), nx = up2_contrib(x, W, Xs, wx);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < ny; ++i)
    for (int j = 0; j < nx; ++j)
      acc += (wy[i] * wx[j]) * *reinterpret_cast<const f32x4*>(dhi + ((b * 2 * H + Ys[i]) * 2 * W + Xs[j]) * hi_ld + 4 * c4);
  if (4 * c4 < C1) *reinterpret_cast<f32x4*>(dx + pix * ld1 + 4 * c4) = acc;
  else *reinterpret_cast<f32x4*>(dskip + pix * ld2 + (4 * c4 - C1)) = acc * skip_scale;
}

// ---------------------------------------------------------------------------------------------------
// message table (msg_processor.py:94-127: lat[b] = sum_k table[2 k + bit_k(b)]): d table[r][h] = sum over the messages that select row r
__global__ __launch_bounds__(256) void msg_table_grad_kernel(const float* __restrict__ dlat, const int32_t* __restrict__ msgs, int Bm, int nbits,
                                                             int hidden, float* __restrict__ dtable) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)2 * nbits * hidden) return;
  const int h = (int)(idx % hidden);
  const int r = (int)(idx / hidden);
  const int k = r >> 1, bit = r & 1;
  float s = 0.f;
  for (int b = 0; b < Bm; ++b)
    if ((msgs[(int64_t)b * nbits + k] != 0) == (bit != 0)) s += dlat[(int64_t)b * hidden + h];
  dtable[idx] = s;
}

// output conv (1x1, <= 4 outputs, optional tanh; unet.py:193-197) backward: dv = d delta * (1 - delta^2), dx = W^T dv; dv is also written as a
// dense [rows][4] matrix for the weight / bias gradients (vs_gemm_wgrad, column sums)
__global__ __launch_bounds__(256) void outc_tanh_bwd_kernel(const float* __restrict__ delta, const float* __restrict__ ddelta, int64_t rpf, int64_t rows,
                                                            int C, const float* __restrict__ w, int Cout, int use_tanh, float* __restrict__ dx,
                                                            int64_t dx_ld, float* __restrict__ dv) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const int64_t b = row / rpf, p = row - b * rpf;
  float g[4] = {0.f, 0.f, 0.f, 0.f};
  for (int o = 0; o < Cout; ++o) {
    const float d = delta[(b * Cout + o) * rpf + p];
    g[o] = ddelta[(b * Cout + o) * rpf + p] * (use_tanh ? 1.f - d * d : 1.f);
  }
  *reinterpret_cast<f32x4*>(dv + row * 4) = f32x4{g[0], g[1], g[2], g[3]};
  float* o_ = dx + row * dx_ld;
  for (int c = 0; c < (int)dx_ld; ++c) {
    float s = 0.f;
    if (c < C)
      for (int o = 0; o < Cout; ++o) s += g[o] * w[(int64_t)o * C + c];
    o_[c] = s;
  }
}

// ReLU backward on rows: dz = z > 0 ? dy : 0 (columns >= C of the output row are zeroed)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ z, int64_t ld, const float* __restrict__ dy, int64_t dy_ld, int C, int O4,
                                                       int64_t total, float* __restrict__ dz, int64_t dz_ld) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cg = (int)(idx % O4);
  const int64_t r = idx / O4;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (4 * cg + e < C && z[r * ld + 4 * cg + e] > 0.f) o[e] = dy[r * dy_ld + 4 * cg + e];
  *reinterpret_cast<f32x4*>(dz + r * dz_ld + 4 * cg) = o;
}

// derivative of vs_apply_act at the PRE-activation value
__device__ __forceinline__ float act_grad(float v, int act) {
  switch (act) {
    case VS_ACT_RELU: return v > 0.f ? 1.f : 0.f;
    case VS_ACT_GELU: return vs_gelu_grad(v);
    case VS_ACT_TANH: { const float t = tanhf(v); return 1.f - t * t; }
    case VS_ACT_SILU: { const float sg = 1.0f / (1.0f + __expf(-v)); return sg * (1.f + v * (1.f - sg)); }
    default: return 1.f;
  }
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ z, int64_t ld, const float* __restrict__ dy, int64_t dy_ld, int C, int O4,
                                                      int act, int64_t total, float* __restrict__ dz, int64_t dz_ld) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cg = (int)(idx % O4);
  const int64_t r = idx / O4;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (4 * cg + e < C) o[e] = dy[r * dy_ld + 4 * cg + e] * act_grad(z[r * ld + 4 * cg + e], act);
  *reinterpret_cast<f32x4*>(dz + r * dz_ld + 4 * cg) = o;
}

// ChanRMSNorm + activation (common.py:172-179: y = act(F.normalize(x, dim = channels) * sqrt(C) * gamma)) backward, LPP lanes per pixel row as in
// the forward (vit_ops.hip::rmsnorm_act_kernel).  With n = x / den (den = max(|x|, 1e-12)), u = n * sqrt(C) * gamma:
//   du = dy * act'(u),  g = du * sqrt(C) * gamma,  dx = (g - n (n . g)) / den;   term = du * sqrt(C) * n  (its column sums are d gamma)
template <int LPP>
__global__ __launch_bounds__(256) void rmsnorm_act_bwd_kernel(const float* __restrict__ x, int64_t rows, int C, int64_t ld, const float* __restrict__ gamma,
                                                              float scale, int act, const float* __restrict__ dy, int64_t dy_ld, float* __restrict__ dx,
                                                              int64_t dx_ld, float* __restrict__ term, int64_t t_ld) {
  const int q = threadIdx.x & (LPP - 1);
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPP;
  const bool live = row < rows;
  const float* xr = x + (live ? row : 0) * ld;
  const float* gr = dy + (live ? row : 0) * dy_ld;
  const int C4 = C >> 2, O4 = (int)(dx_ld >> 2), T4 = (int)(t_ld >> 2);
  float ss = 0.f;
  for (int c4 = q; c4 < C4; c4 += LPP) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * c4);
    ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
#pragma unroll
  for (int o = 1; o < LPP; o <<= 1) ss += __shfl_xor(ss, o, 64);
  const float nrm = sqrtf(ss);
  const float den = fmaxf(nrm, 1e-12f);
  float dot = 0.f;                                   // n . g
  for (int c4 = q; c4 < C4; c4 += LPP) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * c4), gm = *reinterpret_cast<const f32x4*>(gamma + 4 * c4);
    const f32x4 d = *reinterpret_cast<const f32x4*>(gr + 4 * c4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float n = v[e] / den;
      const float u = (n * scale) * gm[e];
      dot += n * (d[e] * act_grad(u, act) * scale * gm[e]);
    }
  }
#pragma unroll
  for (int o = 1; o < LPP; o <<= 1) dot += __shfl_xor(dot, o, 64);
  if (!live) return;
  const bool clamped = nrm < 1e-12f;                 // F.normalize's clamp_min: den is then a constant, no projection term
  float* orow = dx + row * dx_ld;
  float* trow = term + row * t_ld;
  const int M4 = O4 > T4 ? O4 : T4;
  for (int c4 = q; c4 < M4; c4 += LPP) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f}, tt = {0.f, 0.f, 0.f, 0.f};
    if (c4 < C4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * c4), gm = *reinterpret_cast<const f32x4*>(gamma + 4 * c4);
      const f32x4 d = *reinterpret_cast<const f32x4*>(gr + 4 * c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float n = v[e] / den;
        const float du = d[e] * act_grad((n * scale) * gm[e], act);
        const float g = du * scale * gm[e];
        o[e] = (g - (clamped ? 0.f : n * dot)) / den;
        tt[e] = du * scale * n;
      }
    }
    if (c4 < O4) *reinterpret_cast<f32x4*>(orow + 4 * c4) = o;
    if (c4 < T4) *reinterpret_cast<f32x4*>(trow + 4 * c4) = tt;
  }
}

static inline unsigned blocks_for(int64_t n) { return (unsigned)cdiv64(n, 256); }

}  // namespace

// ===================================================================================================== C-ABI
extern "C" int vs_bn_mean_rstd(const double* sums, int C, int64_t ld, float eps, float* mean, float* rstd, void* stream) {
  VS_REQUIRE(sums && mean && rstd && C > 0 && ld >= C);
  hipLaunchKernelGGL(bn_mean_rstd_kernel, dim3(blocks_for(ld)), dim3(256), 0, (hipStream_t)stream, sums, C, ld, eps, mean, rstd);
  return vs_launch_status();
}

extern "C" int64_t vs_bn_bwd_partial_floats(int64_t rows, int64_t ld) { return cdiv64(rows, CR_ROWS) * 2 * ld; }

// sums: 2 * ldp + 1 doubles with ldp = 4 * ceil(C / 4): [sum g * xhat | sum g | rows]
extern "C" int vs_bn_relu_bwd_sums(const float* raw, int64_t ld, const float* dy, int64_t dy_ld, const float* mean, const float* rstd,
                                   const float* scale, const float* shift, int relu, int64_t rows, int C, float* partial, double* sums, float* dgamma,
                                   float* dbeta, void* stream) {
  VS_REQUIRE(raw && dy && mean && rstd && scale && shift && partial && sums && dgamma && dbeta && rows > 0 && C > 0);
  const int C4 = (C + 3) >> 2;
  const int64_t ldp = 4 * (int64_t)C4;
  VS_REQUIRE(ld >= ldp && dy_ld >= ldp && (ld & 3) == 0 && (dy_ld & 3) == 0 && (((uintptr_t)raw) & 15) == 0 && (((uintptr_t)dy) & 15) == 0);
  VS_REQUIRE((((uintptr_t)mean) & 15) == 0 && (((uintptr_t)rstd) & 15) == 0 && (((uintptr_t)scale) & 15) == 0 && (((uintptr_t)shift) & 15) == 0);
  const int nch = (int)cdiv64(rows, CR_ROWS);
  const BnBwd bp{mean, rstd, scale, shift};
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3((unsigned)nch), dim3(256), 0, (hipStream_t)stream, raw, ld, dy, dy_ld, bp, relu, rows, C4, ldp, partial);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)ldp), dim3(64), 0, (hipStream_t)stream, partial, nch, rows, ldp, C, sums, dgamma, dbeta);
  return vs_launch_status();
}

extern "C" int vs_bn_relu_bwd_apply(const float* raw, int64_t ld, const float* dy, int64_t dy_ld, const float* mean, const float* rstd,
                                    const float* scale, const float* shift, int relu, const double* sums, int64_t rows, int C, float* dx,
                                    int64_t dx_ld, void* stream) {
  VS_REQUIRE(raw && dy && mean && rstd && scale && shift && sums && dx && rows > 0 && C > 0 && ld >= C && dy_ld >= C && dx_ld >= C);
  VS_REQUIRE((ld & 3) == 0 && (dy_ld & 3) == 0 && (dx_ld & 3) == 0 && (((uintptr_t)raw) & 15) == 0 && (((uintptr_t)dy) & 15) == 0 && (((uintptr_t)dx) & 15) == 0);
  const int64_t ldp = 4 * (int64_t)((C + 3) >> 2);
  const int O4 = (int)(dx_ld >> 2);
  const int64_t total = rows * O4;
  const BnBwd bp{mean, rstd, scale, shift};
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, raw, ld, dy, dy_ld, bp, relu, sums, ldp, C, O4, total,
                     dx, dx_ld);
  return vs_launch_status();
}

extern "C" int vs_dilate2(const float* dy, int B, int Ho, int Wo, int64_t ld, int H, int W, float* out, void* stream) {
  VS_REQUIRE(dy && out && B > 0 && Ho > 0 && Wo > 0 && H >= 2 * Ho - 1 && W >= 2 * Wo - 1 && ld > 0 && (ld & 3) == 0);
  VS_REQUIRE((((uintptr_t)dy) & 15) == 0 && (((uintptr_t)out) & 15) == 0);
  const int64_t total = (int64_t)B * H * W * (ld / 4);
  hipLaunchKernelGGL(dilate2_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dy, Ho, Wo, ld, H, W, total, out);
  return vs_launch_status();
}

extern "C" int vs_pad_embed1(const float* dy, int B, int H, int W, int64_t ld, float* out, void* stream) {
  VS_REQUIRE(dy && out && B > 0 && H > 0 && W > 0 && ld > 0 && (ld & 3) == 0 && (((uintptr_t)dy) & 15) == 0 && (((uintptr_t)out) & 15) == 0);
  const int64_t total = (int64_t)B * (H + 2) * (W + 2) * (ld / 4);
  hipLaunchKernelGGL(pad_embed_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dy, H, W, ld, total, out);
  return vs_launch_status();
}

extern "C" int vs_reflect_fold1(const float* dxp, int B, int H, int W, int64_t ld, float* out, void* stream) {
  VS_REQUIRE(dxp && out && B > 0 && H >= 2 && W >= 2 && ld > 0 && (ld & 3) == 0 && (((uintptr_t)dxp) & 15) == 0 && (((uintptr_t)out) & 15) == 0);
  const int64_t total = (int64_t)B * H * W * (ld / 4);
  hipLaunchKernelGGL(reflect_fold_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dxp, H, W, ld, total, out);
  return vs_launch_status();
}

extern "C" int vs_im2col3x3_strided(const float* x, int B, int H, int W, int64_t ld, int stride, float* cols, void* stream) {
  VS_REQUIRE(x && cols && B > 0 && H > 0 && W > 0 && ld > 0 && (ld & 3) == 0 && stride >= 1 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)cols) & 15) == 0);
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int64_t total = (int64_t)B * Ho * Wo * 9 * (ld / 4);
  hipLaunchKernelGGL(im2col3x3_zs_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, H, W, ld, stride, Ho, Wo, total, cols);
  return vs_launch_status();
}

extern "C" int vs_upcat2x_bwd(const float* dhi, int64_t hi_ld, int B, int H, int W, int C1, int C2, float skip_scale, float* dx, int64_t ld1,
                              float* dskip, int64_t ld2, void* stream) {
  VS_REQUIRE(dhi && dx && dskip && B > 0 && H > 0 && W > 0 && C1 > 0 && C2 > 0 && (C1 & 3) == 0 && (C2 & 3) == 0 && hi_ld >= C1 + C2 && (hi_ld & 3) == 0);
  VS_REQUIRE(ld1 >= C1 && ld2 >= C2 && (ld1 & 3) == 0 && (ld2 & 3) == 0);
  VS_REQUIRE((((uintptr_t)dhi) & 15) == 0 && (((uintptr_t)dx) & 15) == 0 && (((uintptr_t)dskip) & 15) == 0);
  const int64_t total = (int64_t)B * H * W * ((C1 + C2) / 4);
  hipLaunchKernelGGL(upcat2x_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dhi, hi_ld, H, W, C1, C2, skip_scale, total, dx, ld1,
                     dskip, ld2);
  return vs_launch_status();
}

extern "C" int vs_msg_table_grad(const float* dlat, const int32_t* msgs, int Bm, int nbits, int hidden, float* dtable, void* stream) {
  VS_REQUIRE(dlat && msgs && dtable && Bm > 0 && nbits > 0 && hidden > 0);
  hipLaunchKernelGGL(msg_table_grad_kernel, dim3(blocks_for((int64_t)2 * nbits * hidden)), dim3(256), 0, (hipStream_t)stream, dlat, msgs, Bm, nbits,
                     hidden, dtable);
  return vs_launch_status();
}

extern "C" int vs_outc_tanh_bwd(const float* delta, const float* ddelta, int64_t rows_per_frame, int B, int C, const float* w, int Cout, int use_tanh,
                                float* dx, int64_t dx_ld, float* dv, void* stream) {
  VS_REQUIRE(delta && ddelta && w && dx && dv && rows_per_frame > 0 && B > 0 && C > 0 && Cout > 0 && Cout <= 4 && dx_ld >= C && (((uintptr_t)dv) & 15) == 0);
  const int64_t rows = rows_per_frame * B;
  hipLaunchKernelGGL(outc_tanh_bwd_kernel, dim3(blocks_for(rows)), dim3(256), 0, (hipStream_t)stream, delta, ddelta, rows_per_frame, rows, C, w, Cout,
                     use_tanh, dx, dx_ld, dv);
  return vs_launch_status();
}

extern "C" int vs_act_bwd(const float* z, int64_t ld, const float* dy, int64_t dy_ld, int64_t rows, int C, int act, float* dz, int64_t dz_ld,
                          void* stream) {
  VS_REQUIRE(z && dy && dz && rows > 0 && C > 0 && ld >= C && dy_ld >= C && dz_ld >= C && (dz_ld & 3) == 0 && (((uintptr_t)dz) & 15) == 0);
  const int O4 = (int)(dz_ld >> 2);
  const int64_t total = rows * O4;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, z, ld, dy, dy_ld, C, O4, act, total, dz, dz_ld);
  return vs_launch_status();
}

extern "C" int vs_rmsnorm_act_bwd(const float* x, int64_t rows, int C, int64_t ld, const float* gamma, int act, const float* dy, int64_t dy_ld,
                                  float* dx, int64_t dx_ld, float* term, int64_t term_ld, void* stream) {
  VS_REQUIRE(x && gamma && dy && dx && term && rows > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0 && ld >= C && dy_ld % 4 == 0 && dy_ld >= C &&
             dx_ld % 4 == 0 && dx_ld >= C && term_ld % 4 == 0 && term_ld >= C);
  VS_REQUIRE(((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)term) | ((uintptr_t)gamma)) & 15) == 0);
  const float scale = sqrtf((float)C);
  hipStream_t st = (hipStream_t)stream;
#define VS_RMSB(L_) hipLaunchKernelGGL(rmsnorm_act_bwd_kernel<L_>, dim3((unsigned)cdiv64(rows * L_, 256)), dim3(256), 0, st, x, rows, C, ld, gamma, scale, act, dy, dy_ld, dx, dx_ld, term, term_ld)
  if (C <= 64) VS_RMSB(4);
  else if (C <= 256) VS_RMSB(16);
  else VS_RMSB(64);
#undef VS_RMSB
  return vs_launch_status();
}

extern "C" int vs_relu_bwd(const float* z, int64_t ld, const float* dy, int64_t dy_ld, int64_t rows, int C, float* dz, int64_t dz_ld, void* stream) {
  VS_REQUIRE(z && dy && dz && rows > 0 && C > 0 && ld >= C && dy_ld >= C && dz_ld >= C && (dz_ld & 3) == 0 && (((uintptr_t)dz) & 15) == 0);
  const int O4 = (int)(dz_ld >> 2);
  const int64_t total = rows * O4;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, z, ld, dy, dy_ld, C, O4, total, dz, dz_ld);
  return vs_launch_status();
}
