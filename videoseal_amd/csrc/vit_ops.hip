// Kernels of the legacy videoseal_0.0 card: ChanRMSNorm + activation for its U-Net and the multi-head
// self-attention (windowed / global, decomposed relative positions) of its SAM-style ViT extractor.  The attention products run on the
// matrix cores with the exact 3 x bf16 operand split (vit_attention_mfma_kernel); the vector kernel stays for token counts that are
// not a multiple of 32.
#include "vs_common.h"
#include "conv_common.h"

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

// ---------------------------------------------------------------------------------------------------
// The same attention on the matrix cores (v_mfma_f32_32x32x16_bf16, 3 x bf16 operand split: six partial products, fp32 accumulate -- fp32-level
// accuracy).  Workgroup = one (frame, window, head) group of T = 32 NT tokens, one wave per 32 queries.
//   S^T = K (q * scale)^T is computed TRANSPOSED: in the 32 x 32 accumulator layout a lane then owns ONE query (column lane % 32) and holds
//   its scores against the keys 8 (r / 4) + 4 (lane / 32) + r % 4 of every 32-key tile in registers -- max / exp / sum over the keys are in-lane
//   loops plus one exchange with lane ^ 32, no LDS round trip for the T x T matrix.
//   O^T = V^T P^T: the registers 8 s .. 8 s + 7 of a score tile ARE the B operand of the second product (lane = query column, k = 8 (lane / 32)
//   + v) if its reduction index is enumerated as key(s, half, v) = 16 s + 8 (v / 4) + 4 half + v % 4 -- the A operand (V^T fragments) is read
//   from LDS in that key order.  The accumulator of O^T again has the lane's own query as its column: 1 / sum is a per-lane scalar.
// LDS: K (fp32, staged once per group), then V in the same buffer; the decomposed relative-position terms rel[q][yk | Th + xk]
// (vit.py:436-470, with the UNSCALED q) are computed once per query on the vector ALUs and added to the score registers.
template <int HD, int NT>
__global__ __launch_bounds__(NT * 64) void vit_attention_mfma_kernel(const float* __restrict__ qkv, int H, int W, int heads, int window,
                                                                     const float* __restrict__ rel_h, const float* __restrict__ rel_w,
                                                                     float* __restrict__ out) {
  using vsconv::bf16x8;
  using vsconv::u32x2;
  using vsconv::u32x4;
  using A3 = vsconv::Arith<3>;
  constexpr int T = NT * 32, PITCH = HD + 4, RP = 33, NKB = HD / 16, NDT = (HD + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* kvs = sm;                       // [T][PITCH]: K, later V
  float* relb = sm + T * PITCH;          // [T][RP]: rel_h terms at [0, Th), rel_w terms at [Th, Th + Tw)
  const int Th = window ? window : H, Tw = window ? window : W;
  const int head = blockIdx.x % heads;
  int grp = blockIdx.x / heads;
  const int nwx = W / Tw, nwy = H / Th;
  const int wx = grp % nwx; grp /= nwx;
  const int wy = grp % nwy;
  const int frame = grp / nwy;
  const int D = heads * HD;
  const int64_t tok0 = (int64_t)frame * H * W;
  auto token = [&](int t) -> int64_t { return tok0 + (int64_t)(wy * Th + t / Tw) * W + wx * Tw + t % Tw; };
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  auto stage = [&](int which) {          // rows of K (1) or V (2) of the group -> kvs
    for (int i = tid; i < T * (HD / 4); i += NT * 64) {
      const int t = i / (HD / 4), c4 = i % (HD / 4);
      *reinterpret_cast<f32x4*>(kvs + t * PITCH + 4 * c4) = *reinterpret_cast<const f32x4*>(qkv + token(t) * (3 * D) + which * D + head * HD + 4 * c4);
    }
  };
  stage(1);
  // relative-position terms: thread t < T the rel_h terms of query t, thread T + t its rel_w terms (NT * 64 = 2 T threads)
  {
    const int qi = tid < T ? tid : tid - T;
    const bool is_w = tid >= T;
    const float* table = is_w ? rel_w : rel_h;
    const int n = is_w ? Tw : Th, pos = is_w ? qi % Tw : qi / Tw;
    float* dst = relb + qi * RP + (is_w ? Th : 0);
    if (table) {
      f32x4 q[HD / 4];
      const float* qr = qkv + token(qi) * (3 * D) + head * HD;
#pragma unroll
      for (int i = 0; i < HD / 4; ++i) q[i] = *reinterpret_cast<const f32x4*>(qr + 4 * i);
      for (int k = 0; k < n; ++k) {
        const float* t = table + (int64_t)(pos - k + n - 1) * HD;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) { const f32x4 v = *reinterpret_cast<const f32x4*>(t + 4 * i); s += q[i][0] * v[0] + q[i][1] * v[1] + q[i][2] * v[2] + q[i][3] * v[3]; }
        dst[k] = s;
      }
    } else {
      for (int k = 0; k < n; ++k) dst[k] = 0.f;
    }
  }
  // this wave's queries as B fragments: lane (c, hf) holds (q * scale)[query 32 wave + c][16 kb + 8 hf .. + 7], split into three bf16 planes
  const int qi = wave * 32 + c;
  const float scale = 1.0f / sqrtf((float)HD);
  bf16x8 qf[NKB][3];
  {
    const float* qr = qkv + token(qi) * (3 * D) + head * HD + 8 * hf;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qr + 16 * kb) * scale, b = *reinterpret_cast<const f32x4*>(qr + 16 * kb + 4) * scale;
      u32x2 pa[3], pb[3];
      vsconv::split4(a, pa[0], pa[1], pa[2]);
      vsconv::split4(b, pb[0], pb[1], pb[2]);
#pragma unroll
      for (int p = 0; p < 3; ++p) qf[kb][p] = __builtin_bit_cast(bf16x8, u32x4{pa[p][0], pa[p][1], pb[p][0], pb[p][1]});
    }
  }
  __syncthreads();
  // ---- S^T tiles: sc[kt][r] = score of query c against key 32 kt + 8 (r / 4) + 4 hf + r % 4
  f32x16 sc[NT];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
    for (int e = 0; e < 16; ++e) sc[kt][e] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const float* kr = kvs + (kt * 32 + c) * PITCH + 16 * kb + 8 * hf;
      const f32x4 a = *reinterpret_cast<const f32x4*>(kr), b = *reinterpret_cast<const f32x4*>(kr + 4);
      u32x2 pa[3], pb[3];
      vsconv::split4(a, pa[0], pa[1], pa[2]);
      vsconv::split4(b, pb[0], pb[1], pb[2]);
      bf16x8 kf[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) kf[p] = __builtin_bit_cast(bf16x8, u32x4{pa[p][0], pa[p][1], pb[p][0], pb[p][1]});
#pragma unroll
      for (int q = 0; q < 6; ++q) sc[kt] = A3::mfma(kf[A3::PA[q]], qf[kb][A3::PB[q]], sc[kt]);
    }
  }
  // ---- + relative positions, softmax over the keys (unnormalised p stays in the score registers)
  const float* rb = relb + qi * RP;
  const unsigned inv_tw = 65536u / (unsigned)Tw + 1u;             // j / Tw for j < 256, Tw <= 16
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned j = (unsigned)(kt * 32 + 8 * (r >> 2) + 4 * hf + (r & 3));
      const unsigned yk = (j * inv_tw) >> 16, xk = j - yk * (unsigned)Tw;
      const float s = sc[kt][r] + rb[yk] + rb[Th + xk];
      sc[kt][r] = s;
      m = fmaxf(m, s);
    }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float l = 0.f;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __expf(sc[kt][r] - m);
      sc[kt][r] = pv;
      l += pv;
    }
  l += __shfl_xor(l, 32, 64);
  __syncthreads();                       // every wave is done with K
  stage(2);
  __syncthreads();
  // ---- O^T = V^T P^T: A = V^T fragments (row d = 32 dt + c, k = 8 hf + v <-> key 32 kt + 16 s + 8 (v / 4) + 4 hf + v % 4), B = the score registers
  f32x16 oacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      u32x2 pa[3], pb[3];
      vsconv::split4(f32x4{sc[kt][8 * s2], sc[kt][8 * s2 + 1], sc[kt][8 * s2 + 2], sc[kt][8 * s2 + 3]}, pa[0], pa[1], pa[2]);
      vsconv::split4(f32x4{sc[kt][8 * s2 + 4], sc[kt][8 * s2 + 5], sc[kt][8 * s2 + 6], sc[kt][8 * s2 + 7]}, pb[0], pb[1], pb[2]);
      bf16x8 pf[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) pf[p] = __builtin_bit_cast(bf16x8, u32x4{pa[p][0], pa[p][1], pb[p][0], pb[p][1]});
      const int key0 = kt * 32 + 16 * s2 + 4 * hf;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const int d = dt * 32 + c;
        f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
        if (HD >= 32 || d < HD) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            va[e] = kvs[(key0 + e) * PITCH + d];
            vb[e] = kvs[(key0 + 8 + e) * PITCH + d];
          }
        }
        u32x2 qa[3], qb[3];
        vsconv::split4(va, qa[0], qa[1], qa[2]);
        vsconv::split4(vb, qb[0], qb[1], qb[2]);
        bf16x8 vf[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) vf[p] = __builtin_bit_cast(bf16x8, u32x4{qa[p][0], qa[p][1], qb[p][0], qb[p][1]});
#pragma unroll
        for (int q = 0; q < 6; ++q) oacc[dt] = A3::mfma(vf[A3::PA[q]], pf[A3::PB[q]], oacc[dt]);
      }
    }
  // oacc[dt][r] = O[query c][d = 32 dt + 8 (r / 4) + 4 hf + r % 4]
  const float inv = 1.0f / l;
  float* orow = out + token(qi) * D + head * HD;
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = dt * 32 + 8 * g + 4 * hf;
      if (d < HD) *reinterpret_cast<f32x4*>(orow + d) = f32x4{oacc[dt][4 * g], oacc[dt][4 * g + 1], oacc[dt][4 * g + 2], oacc[dt][4 * g + 3]} * inv;
    }
}

template <int HD, int NT>
int launch_attn_mfma(const float* qkv, int frames, int H, int W, int heads, int window, const float* rel_h, const float* rel_w, float* out,
                     hipStream_t st) {
  const int Th = window ? window : H, Tw = window ? window : W, T = Th * Tw;
  const size_t smem = ((size_t)T * (HD + 4) + (size_t)T * 33) * sizeof(float);
  auto kern = vit_attention_mfma_kernel<HD, NT>;
  static bool attr = false;
  if (smem > 64 * 1024 && !attr) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int64_t groups = (int64_t)frames * (H / Th) * (W / Tw) * heads;
  if (groups >= (1ll << 31)) return VS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(NT * 64), smem, st, qkv, H, W, heads, window, rel_h, rel_w, out);
  return vs_launch_status();
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
  // matrix-core kernel for 64 / 128 / 256 tokens per group (VS_VIT_ATTN=valu forces the vector kernel: A/B and the odd shapes' path)
  static const bool valu = [] { const char* e = getenv("VS_VIT_ATTN"); return e && !strcmp(e, "valu"); }();
  const int T = Th * Tw;
  if (!valu && (T == 64 || T == 128 || T == 256) && (hd == 16 || hd == 32 || hd == 64)) {
#define VS_ATT(HD_, NT_) return launch_attn_mfma<HD_, NT_>(qkv, frames, H, W, heads, window, rel_h, rel_w, out, st)
    if (hd == 64) { if (T == 64) VS_ATT(64, 2); if (T == 128) VS_ATT(64, 4); VS_ATT(64, 8); }
    if (hd == 32) { if (T == 64) VS_ATT(32, 2); if (T == 128) VS_ATT(32, 4); VS_ATT(32, 8); }
    if (T == 64) VS_ATT(16, 2); if (T == 128) VS_ATT(16, 4); VS_ATT(16, 8);
#undef VS_ATT
  }
  switch (hd) {
    case 16: return launch_attn<16>(qkv, frames, H, W, heads, window, rel_h, rel_w, out, st);
    case 32: return launch_attn<32>(qkv, frames, H, W, heads, window, rel_h, rel_w, out, st);
    case 64: return launch_attn<64>(qkv, frames, H, W, heads, window, rel_h, rel_w, out, st);
    default: return VS_ERR_UNSUPPORTED;
  }
}
