// Upsample group of the U-Net in ONE kernel for the thin full-resolution levels (common.py:45-52 after unet.py:186-187):
//   cat(x, skip * 2^-1/2) -> bilinear x2 -> ReflectionPad2d(1) -> Conv3x3 (no bias) -> LayerNorm(C) -> ReLU
// computed as in vs_upconv_gather_ln (net_ops.hip): the nine per-tap products z_t = W_t [x | skip] are taken on the LOW-resolution
// map (a quarter of the conv's MACs) and the interpolation / padding / tap shift become a 36-term gather.  Here the GEMM runs
// inside the workgroup on the matrix cores (3 x bf16 split, six v_mfma_f32_32x32x16_bf16 per product like vs_conv_gemm) and z
// lives only in LDS: at 128^2 x 64 -> 16 channels x 32 frames the two-kernel form writes and re-reads 302 MB of z for 134 MB of
// input and 134 MB of output; this kernel touches the 268 MB only.
//
// Workgroup = 8 x 8 low-resolution cells (16 x 16 output pixels) of one frame, 4 waves:
//   1. K loop over 16-channel chunks of [x | skip]: the 10 x 10 halo pixels (rows padded to 128) are loaded, scaled, split into
//      three bf16 planes and stored to LDS next to the chunk of the pre-split weights ([3][9*Co][K] bf16); wave w multiplies
//      row block w by all NB column blocks;
//   2. accumulators -> z tile in LDS [100][9*Co (+4)] (the staging buffers are dead by then and are reused);
//   3. gather + LayerNorm + activation: 4 lanes per output pixel, Co/4 channels each, two quad shuffles, 16-byte stores.
#include <algorithm>
#include <cstdlib>

#include "conv_common.h"

namespace {

using namespace vsconv;

constexpr int UT = 8;                 // low-resolution cells per tile edge
constexpr int UH = UT + 2;            // with halo
constexpr int UROWS = 128;            // UH*UH = 100 halo pixels padded to 4 MFMA row blocks

// NB = 5 (Co = 16, the 128^2 -> 256^2 level): two workgroups per CU (round 5).  The kernel's phases -- K loop, z tile to LDS, gather + LayerNorm -- run
// one after the other with barriers in between; at 300 registers (220 + 80 accumulator registers) a CU held ONE workgroup of four waves and
// nothing covered a phase's latencies.  With the activation prefetch sized to the layer (four 16-channel chunks: K = 64) and a 256-register
// budget the compiler needs 192 registers and no scratch; 2 x 59 KB of LDS fit.
template <int NB, int CG, int NP>
__global__ __launch_bounds__(256, (NB <= 5 ? 2 : 1)) void upconv_fused_kernel(const float* __restrict__ x, int C1, int64_t ld1, const float* __restrict__ skip,
                                                           int C2, int64_t ld2, float sscale, const unsigned short* __restrict__ wsplit,
                                                           int H, int W, int Co, const float* __restrict__ lnw,
                                                           const float* __restrict__ lnb, float eps, int act, float* __restrict__ out,
                                                           int64_t old, int tiles_x, int tiles_y, int nblk, int abl, float a_mul, float acc_mul) {
  using AR = Arith<NP>;
  constexpr int BN = NB * 32;
  constexpr int A_BYTES = NP * UROWS * ROWB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_u[];
  unsigned char* const As = smem_u;
  unsigned char* const Bs = smem_u + A_BYTES;
  float* const Z = reinterpret_cast<float*>(smem_u);      // reused after the K loop
  const int N = 9 * Co, K = C1 + C2, ZS = N + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, g = lane >> 5;
  const int per = (nblk + 7) >> 3;                        // XCD-aware order (block b runs on XCD b % 8): neighbouring tiles share halos in L2
  const int vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (vb >= nblk) return;
  const int tx = vb % tiles_x;
  const int t1 = vb / tiles_x;
  const int ty = t1 % tiles_y;
  const int b = t1 / tiles_y;
  const int y0 = ty * UT - 1, x0 = tx * UT - 1;           // low-resolution origin of the halo tile

  // ---- A rows of this thread: 2 float4 slots per chunk (128 rows x 4)
  int64_t a_pix[2];
  int a_lds[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int s = tid + i * 256;
    const int row = s >> 2;
    const int hy = row < UH * UH ? row / UH : 0, hx = row < UH * UH ? row % UH : 0;
    const int py = min(max(y0 + hy, 0), H - 1), px = min(max(x0 + hx, 0), W - 1);   // clamped: rows outside the image are never gathered
    a_pix[i] = ((int64_t)b * H + py) * W + px;
    a_lds[i] = row * ROWB + (s & 3) * 8;
  }
  const int k4 = (tid & 3) * 4;
  constexpr int BSLOT = BN * 2;                           // 16-byte slots per plane per chunk
  constexpr int NBL = (BSLOT + 255) / 256;

  f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  // The activation rows of up to KG chunks (128 channels) are requested in ONE burst per group -- with one chunk (8 KB per workgroup)
  // in flight the kernel paid the full HBM latency once per chunk and ran at a quarter of the bandwidth; the weight chunk of step
  // c+1 (L2-resident) is fetched while step c multiplies.
  constexpr int KG = NB <= 5 ? 4 : 8;
  u32x4 rb[NBL][NP];
  auto fetch_b = [&](const int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      const int s = tid + i * 256;
      const int row = s >> 1, sub = s & 1;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        rb[i][p] = u32x4{0u, 0u, 0u, 0u};
        if (s < BSLOT && row < N) rb[i][p] = *reinterpret_cast<const u32x4*>(wsplit + ((int64_t)p * N + row) * K + kc + sub * 8);
      }
    }
  };
  fetch_b(0);
  for (int kg = 0; kg < ((abl & 1) ? 0 : K); kg += KG * BK) {      // abl: tools/bench_upconv.py ablations (1: no GEMM, 2: no gather)
    f32x4 ra[KG][2];
#pragma unroll
    for (int c = 0; c < KG; ++c) {
      // (branch-free since round 6: with the loads inside `if (kc < K)` / `if (from_x) .. else ..` regions hipcc waited for every chunk's pair at the
      //  join -- four dependent round trips per group instead of the ONE burst this loop is for (tools/isa_scan.py order: `LDx2 vmcnt(0)` x 8).  A chunk
      //  past K re-reads the last one (its registers are never stored); x * 1.0f is exact)
      const int kc = kg + c * BK < K ? kg + c * BK : K - BK;
      const bool from_x = kc < C1;                        // C1 % 16 == 0: a chunk never straddles the concat boundary
      const float mul = from_x ? 1.0f : sscale;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float* src = from_x ? x + a_pix[i] * ld1 + kc + k4 : skip + a_pix[i] * ld2 + (kc - C1) + k4;
        ra[c][i] = *reinterpret_cast<const f32x4*>(src) * mul;
      }
    }
#pragma unroll
    for (int c = 0; c < KG; ++c) {
      const int kc = kg + c * BK;
      if (kc < K) {
        if (kc) __syncthreads();                          // the previous chunk's fragments have been read
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          u32x2 pl[NP];
          split4n<NP>(ra[c][i], a_mul, pl);
#pragma unroll
          for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(As + p * UROWS * ROWB + a_lds[i]) = pl[p];
        }
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
          const int s = tid + i * 256;
          if (s < BSLOT) {
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(Bs + p * BN * ROWB + (s >> 1) * ROWB + (s & 1) * 16) = rb[i][p];
          }
        }
        __syncthreads();
        if (kc + BK < K) fetch_b(kc + BK);
        bf16x8 af[NP], bf[NB][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) af[p] = *reinterpret_cast<const bf16x8*>(As + p * UROWS * ROWB + (wave * 32 + r) * ROWB + g * 16);
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int p = 0; p < NP; ++p) bf[j][p] = *reinterpret_cast<const bf16x8*>(Bs + p * BN * ROWB + (j * 32 + r) * ROWB + g * 16);
#pragma unroll
        for (int q = 0; q < AR::NPROD; ++q) {             // smallest partial products first, as in conv_gemm.hip
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[j] = AR::mfma(af[AR::PA[q]], bf[j][AR::PB[q]], acc[j]);
        }
      }
    }
  }
  __syncthreads();                                        // all fragment reads done: the staging area becomes the z tile
  // C/D layout of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = j * 32 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
      if (m < UH * UH && n < N) Z[m * ZS + n] = NP == 2 ? acc[j][e] * acc_mul : acc[j][e];     // back to real units (exact: a power of two)
    }
  }
  __syncthreads();

  // ---- gather + LayerNorm + activation: 16 x 16 output pixels, 4 lanes per pixel
  constexpr int NV = CG / 4;
  const int q4 = tid & 3;
  f32x4 lwv[NV], lbv[NV];                   // LayerNorm parameters of this lane's channels, once (inside the loop: a dependent L2 round trip per pixel)
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    lwv[j] = *reinterpret_cast<const f32x4*>(lnw + q4 * CG + 4 * j);
    lbv[j] = *reinterpret_cast<const f32x4*>(lnb + q4 * CG + 4 * j);
  }
  const int H2 = 2 * H, W2 = 2 * W;
  const float invC = 1.0f / (float)Co;
  for (int it = tid >> 2; it < ((abl & 2) ? 0 : 4 * UT * UT); it += 64) {
    const int oy = it / (2 * UT), ox = it % (2 * UT);
    const int Y = (y0 + 1) * 2 + oy, X = (x0 + 1) * 2 + ox;
    const bool live = Y < H2 && X < W2;
    const int Yc = min(Y, H2 - 1), Xc = min(X, W2 - 1);
    int ys[3][2], xs[3][2];
    float wy[3][2], wx[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int yr = Yc + k - 1, xr = Xc + k - 1;
      yr = yr < 0 ? -yr : (yr >= H2 ? 2 * H2 - 2 - yr : yr);
      xr = xr < 0 ? -xr : (xr >= W2 ? 2 * W2 - 2 - xr : xr);
      const float sy = fmaxf((yr + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((xr + 0.5f) * 0.5f - 0.5f, 0.f);
      const int yl = (int)sy, xl = (int)sx;
      ys[k][0] = yl - y0; ys[k][1] = yl + (yl < H - 1) - y0;          // tile-relative low-resolution coordinates
      xs[k][0] = xl - x0; xs[k][1] = xl + (xl < W - 1) - x0;
      wy[k][1] = sy - yl; wy[k][0] = 1.f - wy[k][1];
      wx[k][1] = sx - xl; wx[k][0] = 1.f - wx[k][1];
    }
    f32x4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* zb = Z + q4 * CG;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const float* zr = zb + ys[ky][a] * UH * ZS + ky * 3 * Co;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const float w = wy[ky][a] * wx[kx][c];
            const float* qz = zr + xs[kx][c] * ZS + kx * Co;
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] += w * *reinterpret_cast<const f32x4*>(qz + 4 * j);
          }
      }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    const float mean = s * invC;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float dl = v[j][e] - mean; var += dl * dl; }
    var += __shfl_xor(var, 1, 64);
    var += __shfl_xor(var, 2, 64);
    const float den = sqrtf(var * invC + eps);
    if (!live) continue;
    float* orow = out + (((int64_t)b * H2 + Y) * W2 + X) * old + q4 * CG;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const f32x4 wv = lwv[j], bv = lbv[j];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = vs_apply_act(wv[e] * ((v[j][e] - mean) / den) + bv[e], act);
      *reinterpret_cast<f32x4*>(orow + 4 * j) = o;
    }
  }
}

template <int NB, int CG>
int launch_fused(const float* x, int C1, int64_t ld1, const float* skip, int C2, int64_t ld2, float s, const void* wsplit, int B, int H, int W,
                 int Co, const float* lnw, const float* lnb, float eps, int act, float* out, int64_t old, int arith, float a_mul, float acc_mul,
                 hipStream_t st) {
  const size_t stage = 3 * (size_t)(UROWS + NB * 32) * ROWB, ztile = (size_t)UH * UH * (9 * Co + 4) * sizeof(float);
  const size_t smem = std::max(stage, ztile);
  if (smem > 160 * 1024) return VS_ERR_UNSUPPORTED;
  auto kern = arith == 2 ? upconv_fused_kernel<NB, CG, 2> : upconv_fused_kernel<NB, CG, 3>;
  static bool attr[2] = {false, false};
  if (smem > 64 * 1024 && !attr[arith == 2]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr[arith == 2] = true;
  }
  const int tiles_x = (W + UT - 1) / UT, tiles_y = (H + UT - 1) / UT;
  const int64_t nblk = (int64_t)B * tiles_x * tiles_y;
  if (nblk >= (1 << 30)) return VS_ERR_UNSUPPORTED;
  static const int abl = [] { const char* e = getenv("VS_UPCONV_ABL"); return e ? atoi(e) : 0; }();
  hipLaunchKernelGGL(kern, dim3((unsigned)((nblk + 7) / 8 * 8)), dim3(256), smem, st, x, C1, ld1, skip, C2, ld2, s,
                     static_cast<const unsigned short*>(wsplit), H, W, Co, lnw, lnb, eps, act, out, old, tiles_x, tiles_y, (int)nblk, abl, a_mul, acc_mul);
  return vs_launch_status();
}

}  // namespace

extern "C" int vs_upconv_fused_supported(int C1, int C2, int Co) {
  return (Co == 16 || Co == 32) && C1 > 0 && C2 > 0 && C1 % 16 == 0 && C2 % 16 == 0 && C1 + C2 <= 256;
}
// Measured on MI355X at 32 frames (tools/bench_upconv.py): 128^2 x (32+32) -> 16: 253 us fused vs 441 us as cat + GEMM + gather (z is
// 302 MB there); 64^2 x (64+64) -> 32: 218 us fused vs 204 us -- the in-kernel GEMM pays for the halo rows (100 of 64) and the row /
// column padding (128 x 288) and its phases do not overlap (ablations: GEMM 110 us, gather 75-107 us, rest 30-39 us).
extern "C" int vs_upconv_fused_preferred(int C1, int C2, int Co) { return vs_upconv_fused_supported(C1, C2, Co) && Co == 16; }

extern "C" int vs_upconv_fused(const float* x, int C1, int64_t ld1, const float* skip, int C2, int64_t ld2, float skip_scale,
                               const void* wt_split, int B, int H, int W, int Co, const float* lnw, const float* lnb, float eps, int act,
                               float* out, int64_t out_ld, int arith, float a_mul, float acc_mul, void* stream) {
  VS_REQUIRE((arith == 0 || arith == 3) || (arith == 2 && a_mul > 0.f && acc_mul > 0.f));
  VS_REQUIRE(x && skip && wt_split && lnw && lnb && out && B > 0 && H > 0 && W > 0 && vs_upconv_fused_supported(C1, C2, Co));
  VS_REQUIRE(ld1 % 4 == 0 && ld2 % 4 == 0 && ld1 >= C1 && ld2 >= C2 && out_ld % 4 == 0 && out_ld >= Co && ((uintptr_t)wt_split & 15) == 0);
  hipStream_t st = (hipStream_t)stream;
  if (Co == 16) return launch_fused<5, 4>(x, C1, ld1, skip, C2, ld2, skip_scale, wt_split, B, H, W, Co, lnw, lnb, eps, act, out, out_ld, arith, a_mul, acc_mul, st);
  return launch_fused<9, 8>(x, C1, ld1, skip, C2, ld2, skip_scale, wt_split, B, H, W, Co, lnw, lnb, eps, act, out, out_ld, arith, a_mul, acc_mul, st);
}
