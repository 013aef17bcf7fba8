// Kernels of the legacy videoseal_0.0 card: ChanRMSNorm + activation for its U-Net and the multi-head
// self-attention (windowed / global, decomposed relative positions) of its SAM-style ViT extractor.  Everything fp32.
#include "vs_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// common.py:172-179  F.normalize(x, dim=1) * sqrt(C) * gamma, then act (+ add): 4 lanes per row (C <= 64) ... a wave per row.
template <int LPP>
__global__ __launch_bounds__(256) void rmsnorm_act_kernel(const float* __restrict__ x, int64_t rows, int C, int64_t ld,
                                                          const float* __restrict__ gamma, float scale, int act,
                                                          const float* __restrict__ add, int64_t add_ld, float* __restrict__ out,
                                                          int64_t out_ld) {
  const int q = threadIdx.x & (LPP - 1);
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPP;
  const bool live = row < rows;
  const float* xr = x + (live ? row : 0) * ld;
  const int C4 = C >> 2, O4 = (int)(out_ld >> 2);
  float ss = 0.f;
  for (int c4 = q; c4 < C4; c4 += LPP) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * c4);
    ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
#pragma unroll
  for (int o = 1; o < LPP; o <<= 1) ss += __shfl_xor(ss, o, 64);
  if (!live) return;
  const float den = fmaxf(sqrtf(ss), 1e-12f);
  float* orow = out + row * out_ld;
  const float* ar = add ? add + row * add_ld : nullptr;
  for (int c4 = q; c4 < O4; c4 += LPP) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (c4 < C4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * c4);
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = vs_apply_act(((v[e] / den) * scale) * g[e], act);
      if (ar) o += *reinterpret_cast<const f32x4*>(ar + 4 * c4);
    }
    *reinterpret_cast<f32x4*>(orow + 4 * c4) = o;
  }
}

// ---------------------------------------------------------------------------------------------------
// Attention group = (frame, window, head); one thread per query token, K and V of the group in LDS (every lane reads the same
// k_j / v_j: broadcast reads), online softmax.  T <= 256 tokens x HD <= 64: 2 x 64 KB of LDS at most.
template <int HD>
__global__ __launch_bounds__(256) void vit_attention_kernel(const float* __restrict__ qkv, int H, int W, int heads, int window,
                                                            const float* __restrict__ rel_h, const float* __restrict__ rel_w,
                                                            float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float kv[];     // K [T][HD] then V [T][HD]
  const int Th = window ? window : H, Tw = window ? window : W, T = Th * Tw;
  const int head = blockIdx.x % heads;
  int grp = blockIdx.x / heads;
  const int nwx = W / Tw, nwy = H / Th;
  const int wx = grp % nwx; grp /= nwx;
  const int wy = grp % nwy;
  const int frame = grp / nwy;
  const int D = heads * HD;
  const int64_t tok0 = (int64_t)frame * H * W;
  auto token = [&](int t) -> int64_t { return tok0 + (int64_t)(wy * Th + t / Tw) * W + wx * Tw + t % Tw; };
  float* Ks = kv;
  float* Vs = kv + (size_t)T * HD;
  for (int i = threadIdx.x; i < T * (HD / 4); i += blockDim.x) {
    const int t = i / (HD / 4), c4 = i % (HD / 4);
    const float* r = qkv + token(t) * (3 * D) + head * HD + 4 * c4;
    *reinterpret_cast<f32x4*>(Ks + t * HD + 4 * c4) = *reinterpret_cast<const f32x4*>(r + D);
    *reinterpret_cast<f32x4*>(Vs + t * HD + 4 * c4) = *reinterpret_cast<const f32x4*>(r + 2 * D);
  }
  __syncthreads();
  const int i = threadIdx.x;
  if (i >= T) return;
  const int yi = i / Tw, xi = i % Tw;
  f32x4 q[HD / 4];
  const float* qr = qkv + token(i) * (3 * D) + head * HD;
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) q[c] = *reinterpret_cast<const f32x4*>(qr + 4 * c);
  // decomposed relative positions (vit.py:436-470): rel_h[i][yj] = q_i . Rh[yi - yj + Th - 1], likewise along x, with the UNSCALED q
  constexpr int TMAX = 16;
  float rh[TMAX], rw[TMAX];
#pragma unroll
  for (int k = 0; k < TMAX; ++k) {
    rh[k] = 0.f; rw[k] = 0.f;
    if (rel_h && k < Th) {
      const float* t = rel_h + (int64_t)(yi - k + Th - 1) * HD;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) { const f32x4 v = *reinterpret_cast<const f32x4*>(t + 4 * c); s += q[c][0] * v[0] + q[c][1] * v[1] + q[c][2] * v[2] + q[c][3] * v[3]; }
      rh[k] = s;
    }
    if (rel_w && k < Tw) {
      const float* t = rel_w + (int64_t)(xi - k + Tw - 1) * HD;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) { const f32x4 v = *reinterpret_cast<const f32x4*>(t + 4 * c); s += q[c][0] * v[0] + q[c][1] * v[1] + q[c][2] * v[2] + q[c][3] * v[3]; }
      rw[k] = s;
    }
  }
  const float scale = 1.0f / sqrtf((float)HD);
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) q[c] *= scale;          // vit.py:349 (q * scale) @ k^T
  f32x4 acc[HD / 4];
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  for (int yj = 0; yj < Th; ++yj) {
    float rhy = 0.f;
#pragma unroll
    for (int k = 0; k < TMAX; ++k) rhy = (k == yj) ? rh[k] : rhy;       // register select (no dynamic indexing -> no scratch)
    for (int xj = 0; xj < Tw; ++xj) {
      float rwx = 0.f;
#pragma unroll
      for (int k = 0; k < TMAX; ++k) rwx = (k == xj) ? rw[k] : rwx;
      const int j = yj * Tw + xj;
      const float* kj = Ks + j * HD;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) { const f32x4 v = *reinterpret_cast<const f32x4*>(kj + 4 * c); s += q[c][0] * v[0] + q[c][1] * v[1] + q[c][2] * v[2] + q[c][3] * v[3]; }
      s += rhy + rwx;
      const float mn = fmaxf(m, s);
      const float corr = __expf(m - mn), p = __expf(s - mn);
      l = l * corr + p;
      const float* vj = Vs + j * HD;
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) acc[c] = acc[c] * corr + p * *reinterpret_cast<const f32x4*>(vj + 4 * c);
      m = mn;
    }
  }
  const float inv = 1.0f / l;
  float* orow = out + token(i) * D + head * HD;
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) *reinterpret_cast<f32x4*>(orow + 4 * c) = acc[c] * inv;
}

template <int HD>
int launch_attn(const float* qkv, int frames, int H, int W, int heads, int window, const float* rel_h, const float* rel_w, float* out,
                hipStream_t st) {
  const int Th = window ? window : H, Tw = window ? window : W, T = Th * Tw;
  const size_t smem = 2 * (size_t)T * HD * sizeof(float);
  auto kern = vit_attention_kernel<HD>;
  static bool attr = false;
  if (smem > 64 * 1024 && !attr) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int64_t groups = (int64_t)frames * (H / Th) * (W / Tw) * heads;
  if (groups >= (1ll << 31)) return VS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3((T + 63) / 64 * 64), smem, st, qkv, H, W, heads, window, rel_h, rel_w, out);
  return vs_launch_status();
}

}  // namespace

extern "C" int vs_rmsnorm_act(const float* x, int64_t rows, int C, int64_t ld, const float* gamma, int act, const float* add,
                              int64_t add_ld, float* out, int64_t out_ld, void* stream) {
  VS_REQUIRE(x && gamma && out && rows > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0 && ld >= C && out_ld % 4 == 0 && out_ld >= C);
  VS_REQUIRE(!add || (add_ld % 4 == 0 && add_ld >= C));
  const float scale = sqrtf((float)C);
  hipStream_t st = (hipStream_t)stream;
#define VS_RMS(L_) hipLaunchKernelGGL(rmsnorm_act_kernel<L_>, dim3((unsigned)cdiv64(rows * L_, 256)), dim3(256), 0, st, x, rows, C, ld, gamma, scale, act, add, add_ld, out, out_ld)
  if (C <= 64) VS_RMS(4);
  else if (C <= 256) VS_RMS(16);
  else VS_RMS(64);
#undef VS_RMS
  return vs_launch_status();
}

extern "C" int vs_vit_attention(const float* qkv, int frames, int H, int W, int heads, int hd, int window, const float* rel_h,
                                const float* rel_w, float* out, void* stream) {
  VS_REQUIRE(qkv && out && frames > 0 && H > 0 && W > 0 && heads > 0 && window >= 0);
  const int Th = window ? window : H, Tw = window ? window : W;
  if (window && (H % window || W % window)) return VS_ERR_UNSUPPORTED;      // padded windows (vit.py:374-379) are not implemented
  if (Th > 16 || Tw > 16 || Th * Tw > 256) return VS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  switch (hd) {
    case 16: return launch_attn<16>(qkv, frames, H, W, heads, window, rel_h, rel_w, out, st);
    case 32: return launch_attn<32>(qkv, frames, H, W, heads, window, rel_h, rel_w, out, st);
    case 64: return launch_attn<64>(qkv, frames, H, W, heads, window, rel_h, rel_w, out, st);
    default: return VS_ERR_UNSUPPORTED;
  }
}
