// Backward building blocks of the SAM-style ViT extractor of the legacy videoseal_0.0 card (SURVEY.md 8(f)1 x 8(f)4: the detector fine-tuning
// step of train.py:517-523 for `extractor: sam`): what csrc/bwd_ops.hip does not already hold.  gfx950 only, everything fp32, deterministic
// (no atomics: every sum has a fixed order).
//   gelu_bwd                     MLP activation (vit.py:165-169): dz = dy * gelu'(z)
//   vit_attention_bwd            multi-head self-attention with decomposed relative positions (vit.py:302-360, 436-470), windowed or global.
//     With P = softmax(S), S[i][j] = scale q_i.k_j + q_i.Rh[yi - yj + Th - 1] + q_i.Rw[xi - xj + Tw - 1] and O = P V:
//       D_i = dO_i . O_i,  dS[i][j] = P[i][j] (dO_i . v_j - D_i)
//       dq_i = scale sum_j dS[i][j] k_j + sum_yj gh[i][yj] Rh[yi - yj + Th - 1] + sum_xj gw[i][xj] Rw[xi - xj + Tw - 1]
//       dk_j = scale sum_i dS[i][j] q_i,   dv_j = sum_i P[i][j] dO_i
//       dRh[r] = sum over groups, heads, i, yj with yi - yj + Th - 1 = r of gh[i][yj] q_i   (gh[i][yj] = sum_xj dS[i][(yj, xj)]; gw likewise)
//     Nothing but qkv and O is kept from the forward: the scores are recomputed.  Kernel Q (one thread per query, K / V of the group in LDS):
//     row maximum, normaliser, dq, gh / gw and the per-query terms kernel K needs.  Kernel K (one thread per key, q / dO of the group in LDS):
//     dk and dv.  vit_relpos_grad: the two tables.
#include "vs_common.h"

namespace {

static inline unsigned blocks_for(int64_t n) { return (unsigned)cdiv64(n, 256); }

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ z, int64_t ld, const float* __restrict__ dy, int64_t dy_ld, int C, int O4,
                                                       int64_t total, float* __restrict__ dz, int64_t dz_ld) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cg = (int)(idx % O4);
  const int64_t r = idx / O4;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (4 * cg + e < C) o[e] = dy[r * dy_ld + 4 * cg + e] * vs_gelu_grad(z[r * ld + 4 * cg + e]);
  *reinterpret_cast<f32x4*>(dz + r * dz_ld + 4 * cg) = o;
}

constexpr int TMAX = 16;        // tokens per window side (vs_vit_attention's limit)
constexpr int RP = 2 * TMAX + 4;   // per-query record: rh[TMAX] | rw[TMAX] | m, l, D, pad

struct AttnGeom {
  int H, W, heads, window, Th, Tw, T, nwx, nwy, D;
  __device__ void group(int g, int& head, int& frame, int& wy, int& wx) const {
    head = g % heads; g /= heads;
    wx = g % nwx; g /= nwx;
    wy = g % nwy;
    frame = g / nwy;
  }
  __device__ int64_t token(int frame, int wy, int wx, int t) const {
    return (int64_t)frame * H * W + (int64_t)(wy * Th + t / Tw) * W + wx * Tw + t % Tw;
  }
};

// ---- kernel Q: thread = query i of the group.  rec[group][i][RP] receives rh / rw (the relative-position terms of the scores), the row maximum,
// the normaliser and D; gsum[group][i][2 * TMAX] the row sums gh | gw of dS; dqkv the gradient of q.
template <int HD>
__global__ __launch_bounds__(256) void vit_attn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ o, const float* __restrict__ d_o,
                                                             const AttnGeom G, const float* __restrict__ rel_h, const float* __restrict__ rel_w,
                                                             float* __restrict__ dqkv, float* __restrict__ rec, float* __restrict__ gsum) {
  extern __shared__ __attribute__((aligned(16))) float kv[];     // K [T][HD] then V [T][HD]
  int head, frame, wy, wx;
  G.group(blockIdx.x, head, frame, wy, wx);
  const int T = G.T, Th = G.Th, Tw = G.Tw, D = G.D;
  float* Ks = kv;
  float* Vs = kv + (size_t)T * HD;
  for (int i = threadIdx.x; i < T * (HD / 4); i += blockDim.x) {
    const int t = i / (HD / 4), c4 = i % (HD / 4);
    const float* r = qkv + G.token(frame, wy, wx, t) * (3 * D) + head * HD + 4 * c4;
    *reinterpret_cast<f32x4*>(Ks + t * HD + 4 * c4) = *reinterpret_cast<const f32x4*>(r + D);
    *reinterpret_cast<f32x4*>(Vs + t * HD + 4 * c4) = *reinterpret_cast<const f32x4*>(r + 2 * D);
  }
  __syncthreads();
  const int i = threadIdx.x;
  if (i >= T) return;
  const int yi = i / Tw, xi = i % Tw;
  const int64_t tok = G.token(frame, wy, wx, i);
  f32x4 q[HD / 4], go[HD / 4];
  float Dq = 0.f;
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    q[c] = *reinterpret_cast<const f32x4*>(qkv + tok * (3 * D) + head * HD + 4 * c);
    go[c] = *reinterpret_cast<const f32x4*>(d_o + tok * D + head * HD + 4 * c);
    const f32x4 ov = *reinterpret_cast<const f32x4*>(o + tok * D + head * HD + 4 * c);
    Dq += go[c][0] * ov[0] + go[c][1] * ov[1] + go[c][2] * ov[2] + go[c][3] * ov[3];
  }
  auto dot = [&](const f32x4 (&a)[HD / 4], const float* p) __attribute__((always_inline)) -> float {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) { const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * c); s += a[c][0] * v[0] + a[c][1] * v[1] + a[c][2] * v[2] + a[c][3] * v[3]; }
    return s;
  };
  float* myrec = rec + ((int64_t)blockIdx.x * T + i) * RP;
  float* mysum = gsum + ((int64_t)blockIdx.x * T + i) * (2 * TMAX);
  for (int k = 0; k < TMAX; ++k) {               // the same terms, in the same arithmetic, as the forward kernels (unscaled q)
    myrec[k] = (rel_h && k < Th) ? dot(q, rel_h + (int64_t)(yi - k + Th - 1) * HD) : 0.f;
    myrec[TMAX + k] = (rel_w && k < Tw) ? dot(q, rel_w + (int64_t)(xi - k + Tw - 1) * HD) : 0.f;
    mysum[k] = 0.f;
    mysum[TMAX + k] = 0.f;
  }
  const float scale = 1.0f / sqrtf((float)HD);
  f32x4 qs[HD / 4];
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) qs[c] = q[c] * scale;
  // pass 1: row maximum and normaliser
  float m = -INFINITY;
  for (int j = 0; j < T; ++j) m = fmaxf(m, dot(qs, Ks + j * HD) + myrec[j / Tw] + myrec[TMAX + j % Tw]);
  float l = 0.f;
  for (int j = 0; j < T; ++j) l += __expf(dot(qs, Ks + j * HD) + myrec[j / Tw] + myrec[TMAX + j % Tw] - m);
  myrec[2 * TMAX] = m; myrec[2 * TMAX + 1] = l; myrec[2 * TMAX + 2] = Dq;
  // pass 2: dS, its row sums, dq
  f32x4 dq[HD / 4];
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) dq[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float inv = 1.0f / l;
  for (int yj = 0; yj < Th; ++yj) {
    float gh = 0.f;
    for (int xj = 0; xj < Tw; ++xj) {
      const int j = yj * Tw + xj;
      const float p = __expf(dot(qs, Ks + j * HD) + myrec[yj] + myrec[TMAX + xj] - m) * inv;
      const float ds = p * (dot(go, Vs + j * HD) - Dq);
      gh += ds;
      mysum[TMAX + xj] += ds;                      // (own record: plain read-modify-write)
      const float dss = ds * scale;
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) dq[c] += dss * *reinterpret_cast<const f32x4*>(Ks + j * HD + 4 * c);
    }
    mysum[yj] = gh;
  }
  for (int k = 0; k < TMAX; ++k) {
    if (rel_h && k < Th) {
      const float g = mysum[k];
      const float* t = rel_h + (int64_t)(yi - k + Th - 1) * HD;
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) dq[c] += g * *reinterpret_cast<const f32x4*>(t + 4 * c);
    }
    if (rel_w && k < Tw) {
      const float g = mysum[TMAX + k];
      const float* t = rel_w + (int64_t)(xi - k + Tw - 1) * HD;
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) dq[c] += g * *reinterpret_cast<const f32x4*>(t + 4 * c);
    }
  }
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) *reinterpret_cast<f32x4*>(dqkv + tok * (3 * D) + head * HD + 4 * c) = dq[c];
}

// ---- kernel K: thread = key j of the group; q (scaled) and dO of the group in LDS, the per-query records read from global memory
template <int HD>
__global__ __launch_bounds__(256) void vit_attn_bwd_k_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o, const AttnGeom G,
                                                             const float* __restrict__ rec, float* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) float sm[];     // Q * scale [T][HD], dO [T][HD]
  int head, frame, wy, wx;
  G.group(blockIdx.x, head, frame, wy, wx);
  const int T = G.T, Tw = G.Tw, D = G.D;
  const float scale = 1.0f / sqrtf((float)HD);
  float* Qs = sm;
  float* Gs = sm + (size_t)T * HD;
  for (int i = threadIdx.x; i < T * (HD / 4); i += blockDim.x) {
    const int t = i / (HD / 4), c4 = i % (HD / 4);
    const int64_t tok = G.token(frame, wy, wx, t);
    *reinterpret_cast<f32x4*>(Qs + t * HD + 4 * c4) = *reinterpret_cast<const f32x4*>(qkv + tok * (3 * D) + head * HD + 4 * c4) * scale;
    *reinterpret_cast<f32x4*>(Gs + t * HD + 4 * c4) = *reinterpret_cast<const f32x4*>(d_o + tok * D + head * HD + 4 * c4);
  }
  __syncthreads();
  const int j = threadIdx.x;
  if (j >= T) return;
  const int yj = j / Tw, xj = j % Tw;
  const int64_t tok = G.token(frame, wy, wx, j);
  f32x4 k[HD / 4], v[HD / 4], dk[HD / 4], dv[HD / 4];
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    k[c] = *reinterpret_cast<const f32x4*>(qkv + tok * (3 * D) + D + head * HD + 4 * c);
    v[c] = *reinterpret_cast<const f32x4*>(qkv + tok * (3 * D) + 2 * D + head * HD + 4 * c);
    dk[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    dv[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float* grec = rec + (int64_t)blockIdx.x * T * RP;
  for (int i = 0; i < T; ++i) {
    const float* r = grec + (int64_t)i * RP;
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      const f32x4 qv = *reinterpret_cast<const f32x4*>(Qs + i * HD + 4 * c), gv = *reinterpret_cast<const f32x4*>(Gs + i * HD + 4 * c);
      s += qv[0] * k[c][0] + qv[1] * k[c][1] + qv[2] * k[c][2] + qv[3] * k[c][3];
      dp += gv[0] * v[c][0] + gv[1] * v[c][1] + gv[2] * v[c][2] + gv[3] * v[c][3];
    }
    const float p = __expf(s + r[yj] + r[TMAX + xj] - r[2 * TMAX]) / r[2 * TMAX + 1];
    const float ds = p * (dp - r[2 * TMAX + 2]);
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      dk[c] += ds * *reinterpret_cast<const f32x4*>(Qs + i * HD + 4 * c);          // (Qs holds q * scale)
      dv[c] += p * *reinterpret_cast<const f32x4*>(Gs + i * HD + 4 * c);
    }
  }
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    *reinterpret_cast<f32x4*>(dqkv + tok * (3 * D) + D + head * HD + 4 * c) = dk[c];
    *reinterpret_cast<f32x4*>(dqkv + tok * (3 * D) + 2 * D + head * HD + 4 * c) = dv[c];
  }
}

// ---- the relative-position tables: workgroup = (table row r: blockIdx.x < 2 Th - 1 is rel_h, then rel_w; chunk of groups blockIdx.y),
// thread = (lane g of 4, channel d).  Lane g adds the chunk's groups g, g + 4, ... in order, the four lane sums are combined in the order 0..3
// into partial[chunk][row][HD]; vit_relpos_reduce_kernel adds the chunks in order.
constexpr int RP_CHUNKS = 64;
__global__ __launch_bounds__(256) void vit_relpos_grad_kernel(const float* __restrict__ qkv, const float* __restrict__ gsum, const AttnGeom G, int HD,
                                                              int ngroups, float* __restrict__ partial) {
  __shared__ float sh[4][64];
  const int nh = 2 * G.Th - 1, nrows = nh + 2 * G.Tw - 1;
  const bool is_w = (int)blockIdx.x >= nh;
  const int r = is_w ? blockIdx.x - nh : blockIdx.x;
  const int n = is_w ? G.Tw : G.Th;
  const int d = threadIdx.x & 63, lane = threadIdx.x >> 6;
  const int per = (ngroups + RP_CHUNKS - 1) / RP_CHUNKS;
  const int g0 = blockIdx.y * per, g1 = min(ngroups, g0 + per);
  float acc = 0.f;
  if (d < HD)
    for (int grp = g0 + lane; grp < g1; grp += 4) {
      int head, frame, wy, wx;
      G.group(grp, head, frame, wy, wx);
      for (int i = 0; i < G.T; ++i) {
        const int pos = is_w ? i % G.Tw : i / G.Tw;
        const int k = pos - r + n - 1;               // the key coordinate whose offset from this query is table row r
        if (k < 0 || k >= n) continue;
        const float g = gsum[((int64_t)grp * G.T + i) * (2 * TMAX) + (is_w ? TMAX : 0) + k];
        acc += g * qkv[G.token(frame, wy, wx, i) * (3 * G.D) + head * HD + d];
      }
    }
  sh[lane][d] = acc;
  __syncthreads();
  if (lane == 0 && d < HD) partial[((int64_t)blockIdx.y * nrows + blockIdx.x) * HD + d] = ((sh[0][d] + sh[1][d]) + sh[2][d]) + sh[3][d];
}
__global__ __launch_bounds__(256) void vit_relpos_reduce_kernel(const float* __restrict__ partial, int nh, int nw, int HD, float* __restrict__ drel_h,
                                                                float* __restrict__ drel_w) {
  const int idx = blockIdx.x * 256 + threadIdx.x, nrows = nh + nw;
  if (idx >= nrows * HD) return;
  float s = 0.f;
  for (int c = 0; c < RP_CHUNKS; ++c) s += partial[(int64_t)c * nrows * HD + idx];
  const int r = idx / HD, d = idx - r * HD;
  if (r < nh) drel_h[(int64_t)r * HD + d] = s;
  else drel_w[(int64_t)(r - nh) * HD + d] = s;
}

template <int HD>
int launch_attn_bwd(const float* qkv, const float* o, const float* d_o, const AttnGeom& G, int64_t groups, const float* rel_h, const float* rel_w,
                    float* dqkv, float* rec, float* gsum, float* part, float* drel_h, float* drel_w, hipStream_t st) {
  const size_t smem = 2 * (size_t)G.T * HD * sizeof(float);
  auto kq = vit_attn_bwd_q_kernel<HD>;
  auto kk = vit_attn_bwd_k_kernel<HD>;
  static bool attr = false;
  if (smem > 64 * 1024 && !attr) {
    (void)hipFuncSetAttribute((const void*)kq, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const unsigned nt = (unsigned)((G.T + 63) / 64 * 64);
  hipLaunchKernelGGL(kq, dim3((unsigned)groups), dim3(nt), smem, st, qkv, o, d_o, G, rel_h, rel_w, dqkv, rec, gsum);
  hipLaunchKernelGGL(kk, dim3((unsigned)groups), dim3(nt), smem, st, qkv, d_o, G, (const float*)rec, dqkv);
  if (rel_h && rel_w && drel_h && drel_w) {
    const int nh = 2 * G.Th - 1, nw = 2 * G.Tw - 1;
    hipLaunchKernelGGL(vit_relpos_grad_kernel, dim3((unsigned)(nh + nw), RP_CHUNKS), dim3(256), 0, st, qkv, (const float*)gsum, G, HD, (int)groups, part);
    hipLaunchKernelGGL(vit_relpos_reduce_kernel, dim3(blocks_for((int64_t)(nh + nw) * HD)), dim3(256), 0, st, (const float*)part, nh, nw, HD, drel_h, drel_w);
  }
  return vs_launch_status();
}

}  // namespace

extern "C" int vs_gelu_bwd(const float* z, int64_t ld, const float* dy, int64_t dy_ld, int64_t rows, int C, float* dz, int64_t dz_ld, void* stream) {
  VS_REQUIRE(z && dy && dz && rows > 0 && C > 0 && ld >= C && dy_ld >= C && dz_ld >= C && (dz_ld & 3) == 0 && (((uintptr_t)dz) & 15) == 0);
  const int O4 = (int)(dz_ld >> 2);
  const int64_t total = rows * O4;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, z, ld, dy, dy_ld, C, O4, total, dz, dz_ld);
  return vs_launch_status();
}

extern "C" int64_t vs_vit_attention_bwd_scratch_floats(int frames, int H, int W, int heads, int window) {
  if (frames <= 0 || H <= 0 || W <= 0 || heads <= 0 || window < 0) return 0;
  return (int64_t)frames * H * W * heads * (RP + 2 * TMAX)        // one record + one row-sum vector per (token, head)
         + (int64_t)RP_CHUNKS * 4 * TMAX * 64;                     // + the partial sums of the two relative-position tables
}

extern "C" int vs_vit_attention_bwd(const float* qkv, const float* out, const float* dout, int frames, int H, int W, int heads, int hd, int window,
                                    const float* rel_h, const float* rel_w, float* dqkv, float* scratch, float* drel_h, float* drel_w, void* stream) {
  VS_REQUIRE(qkv && out && dout && dqkv && scratch && frames > 0 && H > 0 && W > 0 && heads > 0 && window >= 0);
  VS_REQUIRE((!rel_h) == (!rel_w) && (!drel_h) == (!drel_w) && (rel_h || !drel_h));
  VS_REQUIRE(((((uintptr_t)qkv) | ((uintptr_t)out) | ((uintptr_t)dout) | ((uintptr_t)dqkv) | ((uintptr_t)scratch)) & 15) == 0);
  AttnGeom G{};
  G.H = H; G.W = W; G.heads = heads; G.window = window;
  G.Th = window ? window : H; G.Tw = window ? window : W;
  if (window && (H % window || W % window)) return VS_ERR_UNSUPPORTED;      // padded windows (vit.py:374-379) are not implemented
  if (G.Th > TMAX || G.Tw > TMAX || G.Th * G.Tw > 256) return VS_ERR_UNSUPPORTED;
  G.T = G.Th * G.Tw; G.nwx = W / G.Tw; G.nwy = H / G.Th; G.D = heads * hd;
  const int64_t groups = (int64_t)frames * G.nwy * G.nwx * heads;
  if (groups >= (1ll << 31)) return VS_ERR_UNSUPPORTED;
  float* rec = scratch;
  float* gsum = scratch + (int64_t)frames * H * W * heads * RP;
  float* part = gsum + (int64_t)frames * H * W * heads * (2 * TMAX);
  hipStream_t st = (hipStream_t)stream;
  switch (hd) {
    case 16: return launch_attn_bwd<16>(qkv, out, dout, G, groups, rel_h, rel_w, dqkv, rec, gsum, part, drel_h, drel_w, st);
    case 32: return launch_attn_bwd<32>(qkv, out, dout, G, groups, rel_h, rel_w, dqkv, rec, gsum, part, drel_h, drel_w, st);
    case 64: return launch_attn_bwd<64>(qkv, out, dout, G, groups, rel_h, rel_w, dqkv, rec, gsum, part, drel_h, drel_w, st);
    default: return VS_ERR_UNSUPPORTED;
  }
}
