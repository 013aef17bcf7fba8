// Implicit-GEMM convolution / linear layer on the gfx950 matrix cores, fp32 in / fp32 out.
//
//   out[m][n] = epilogue( sum_k A[m][k] * wt[n][k] ),  m = (frame, oy, ox) row-major, k = (tap, channel)
//
// Three arithmetic back-ends (template parameter NPL = number of 16-bit operand planes), all with fp32-rounding-level accuracy:
//   NPL = 0 : v_mfma_f32_32x32x2_f32 -- exact fmaf chain at the f32 vector rate (157 TF peak).
//   NPL = 2 : "2 x f16" (the default): round-to-nearest split into two f16 terms with power-of-two range scaling, three partial
//             products per K chunk on v_mfma_f32_32x32x16_f16 -- see conv_common.h (Arith<2>) for the error analysis and the probe.
//   NPL = 3 : "3 x bf16".  Every fp32 operand is split EXACTLY into three bf16 terms by truncation,
//                   x = x1 + x2 + x3 (8+8+8 mantissa bits), and a product is accumulated as its six partial products
//                   of weight >= 2^-16:  a1b1 + (a1b2 + a2b1) + (a2b2 + a1b3 + a3b1)   (dropped: <= 2^-24 |ab|),
//                   each one v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  bf16 MFMA has 16x the rate of the
//                   f32-input MFMA, so six of them are 2.67x faster at the same accuracy (417 TF-equivalent peak).
//                   Weights are pre-split at pack time ([3][N][Ktot] bf16); activations are split while they are
//                   staged global -> registers -> LDS (2 v_and + 2 v_sub + 1.5 v_perm per element).
//
// Common structure (MI355X: 64-lane waves, 4 SIMDs/CU, 160 KB LDS):
//   * a 4-wave workgroup owns a BM x BN output tile, each wave TM x TN MFMA tiles of 32x32;
//   * K is walked in chunks of 16 channels of one filter tap.  A is gathered straight from the NHWC activation,
//     B from the packed weight; both go global -> registers -> LDS with two LDS buffers (loads of chunk s+1 are in
//     flight while chunk s is multiplied; one barrier per chunk);
//   * addresses are a uniform (SGPR) base + a per-lane 32-bit byte offset, so a chunk costs no per-lane address
//     arithmetic; the per-row tap validity (zero padding) / reflected pixel offset is refreshed only when the tap
//     changes; loads are unconditional (out-of-range rows are clamped to valid memory and masked afterwards);
//   * LDS rows are padded (f32: 16+4 floats, bf16: 16+8 halves) so that the 16 lanes of every ds_read_b128 service
//     group hit 16 distinct 4-bank slots;
//   * epilogue in registers: bias, ReLU/GELU/tanh, optional second K phase (the ResnetBlock's 1x1 res_conv on the
//     block input accumulated on top of relu(bn(conv))), optional residual, coalesced 128-byte row stores.
#include <algorithm>

#include "conv_common.h"

namespace {

using namespace vsconv;

// NPL: 0 = f32 MFMA, 3 = "3 x bf16", 2 = "2 x f16" (conv_common.h, Arith<NP>)
template <int WM, int WN, int TM, int TN, int NPL>
__global__ __launch_bounds__(NT, (TM * TN > 4 ? 1 : 2)) void conv_gemm_kernel(const vs_conv_desc_t d, const int M, const int mtiles) {
  constexpr bool SPLIT = NPL != 0;
  using AR = Arith<SPLIT ? NPL : 3>;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int NA = BM * 4 / NT;                       // float4 A loads per thread per chunk
  constexpr int BSLOTS = SPLIT ? BN * 2 : BN * 4;       // 16-byte B slots per plane per chunk
  constexpr int NB = (BSLOTS + NT - 1) / NT;
  constexpr int NP = SPLIT ? NPL : 1;                   // weight planes
  static_assert(WM * WN == 4 && (BM * 4) % NT == 0, "4 waves, whole A tile per pass");

  constexpr int A_BYTES = SPLIT ? NP * BM * ROWB : BM * LDKF * 4;
  constexpr int B_BYTES = SPLIT ? NP * BN * ROWB : BN * LDKF * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];
  unsigned char* const As0 = smem;
  unsigned char* const Bs0 = smem + 2 * A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, g = lane >> 5;

  const int bm = blockIdx.x % mtiles;
  const int bn = blockIdx.x / mtiles;
  const int64_t m0 = (int64_t)bm * BM;
  const int n0 = bn * BN;

  const int spt = d.CinP / BK;                  // chunks per tap
  const int n1 = d.KH * d.KW * spt;
  const int n2 = d.in2 ? d.Cin2P / BK : 0;
  const int n1e = n2 > 0 ? ((n1 + 1) & ~1) : n1;   // the 2-deep pipeline switches phase on an even step: pad with a null chunk
  const int total = n1e + n2;
  const int64_t Ktot = (int64_t)d.KH * d.KW * d.CinP;
  const int HoWo = d.Ho * d.Wo;
  const bool reflect = d.pad_mode == VS_PAD_REFLECT;
  const int abl = VS_KERNEL_ABL(d);   // debug ablation mask (tools/bench_conv.py): 1 no global loads, 2 no MFMA, 4 no LDS stores, 8 no barrier
  const bool chk_c = (d.Cin % BK) != 0 || (d.in2 && (d.Cin2 % BK) != 0);   // partial channel chunks exist

  // ---- A rows of this thread (fixed for the whole K loop)
  bool a_ok[NA];
  int a_oy[NA], a_ox[NA];
  unsigned a_pix[NA];       // byte offset of (frame, oy*SH, ox*SW, k4)
  unsigned a_cur[NA];       // byte offset used for the current tap (tap offset folded in, 0 when the tap is padding)
  bool a_tap[NA];           // current tap is inside the image for this row
  unsigned a_off2[NA];      // byte offset of row m in in2
  unsigned a_soff[NA];      // byte offset of (frame, k4) in a_scale
  const int k4 = (tid & 3) * 4;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (tid + i * NT) >> 2;
    const int64_t m = m0 + row;
    a_ok[i] = m < M;
    const int64_t mm = a_ok[i] ? m : 0;
    const int b = (int)(mm / HoWo);
    const int rem = (int)(mm - (int64_t)b * HoWo);
    const int oy = rem / d.Wo, ox = rem - oy * d.Wo;
    a_oy[i] = oy * d.SH;
    a_ox[i] = ox * d.SW;
    a_pix[i] = (unsigned)(((int64_t)b * d.in_sb + (int64_t)a_oy[i] * d.in_sy + (int64_t)a_ox[i] * d.in_sx + k4) * 4);
    a_off2[i] = (unsigned)((mm * d.in2_ld + k4) * 4);
    a_soff[i] = (unsigned)(((int64_t)b * d.a_scale_ld + k4) * 4);
    a_cur[i] = 0;
    a_tap[i] = false;
  }
  // ---- B slots of this thread: rows beyond N are clamped to row N-1 (masked in the epilogue)
  unsigned b_off[NB], b_off2[NB];
  int b_lds[NB];
  bool b_have[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int idx = tid + i * NT;
    b_have[i] = idx < BSLOTS;
    const int row = SPLIT ? (idx >> 1) : (idx >> 2);
    const int sub = SPLIT ? (idx & 1) : (idx & 3);
    int nrow = n0 + row;
    nrow = nrow < d.N ? nrow : d.N - 1;
    if (SPLIT) {
      b_off[i] = (unsigned)(((int64_t)nrow * Ktot + sub * 8) * 2);
      b_off2[i] = (unsigned)(((int64_t)nrow * d.Cin2P + sub * 8) * 2);
      b_lds[i] = row * ROWB + sub * 16;
    } else {
      b_off[i] = (unsigned)(((int64_t)nrow * Ktot + sub * 4) * 4);
      b_off2[i] = (unsigned)(((int64_t)nrow * d.Cin2P + sub * 4) * 4);
      b_lds[i] = (row * LDKF + sub * 4) * 4;
    }
  }
  const int64_t plane1 = (int64_t)d.N * Ktot * 2;        // bytes per bf16 weight plane
  const int64_t plane2 = (int64_t)d.N * d.Cin2P * 2;
  const char* const wbase = SPLIT ? reinterpret_cast<const char*>(d.wt_split) : reinterpret_cast<const char*>(d.wt);
  const char* const wbase2 = SPLIT ? reinterpret_cast<const char*>(d.wt2_split) : reinterpret_cast<const char*>(d.wt2);
  constexpr int WELT = SPLIT ? 2 : 4;                    // bytes per weight element

  const float amul = NPL == 2 ? d.a_mul : 1.f;
  int ld_ky = 0, ld_kx = 0, ld_cc = 0, ld_step = 0;
  // two register sets: global loads run two chunks ahead of the MFMAs (chunk c lives in set c&1, LDS buffer c&1)
  struct Regs { f32x4 a[NA]; u32x4 b[NB][NP]; };
  Regs R0, R1;

  auto refresh_tap = [&]() __attribute__((always_inline)) {
    const int64_t tap_delta = ((int64_t)(ld_ky - d.PH) * d.in_sy + (int64_t)(ld_kx - d.PW) * d.in_sx) * 4;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int iy = a_oy[i] - d.PH + ld_ky, ix = a_ox[i] - d.PW + ld_kx;
      if (reflect) {
        const int ry = iy < 0 ? -iy : (iy >= d.H ? 2 * d.H - 2 - iy : iy);
        const int rx = ix < 0 ? -ix : (ix >= d.W ? 2 * d.W - 2 - ix : ix);
        const int64_t refl = ((int64_t)(ry - iy) * d.in_sy + (int64_t)(rx - ix) * d.in_sx) * 4;
        a_tap[i] = a_ok[i];
        a_cur[i] = a_ok[i] ? (unsigned)((int64_t)a_pix[i] + tap_delta + refl) : 0u;
      } else {
        a_tap[i] = a_ok[i] && (iy >= 0) && (iy < d.H) && (ix >= 0) && (ix < d.W);
        a_cur[i] = a_tap[i] ? (unsigned)((int64_t)a_pix[i] + tap_delta) : 0u;
      }
    }
  };

  auto load_chunk = [&](Regs& R) __attribute__((always_inline)) {
    f32x4 (&ra)[NA] = R.a;
    u32x4 (&rb)[NB][NP] = R.b;
    const int s = ld_step;
    if (s < n1) {
      if (ld_cc == 0) refresh_tap();
      const char* abase = reinterpret_cast<const char*>(d.in) + (int64_t)ld_cc * 4;   // uniform
      if (!chk_c) {
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(abase + a_cur[i]);
      } else {
        const bool cok = (ld_cc + k4) < d.Cin;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (cok) v = *reinterpret_cast<const f32x4*>(abase + a_cur[i]);
          ra[i] = v;
        }
      }
      if (d.a_scale) {   // GRN apply (1x1 convs only)
        const char* sbase = reinterpret_cast<const char*>(d.a_scale) + (int64_t)ld_cc * 4;
        const f32x4 sh = *reinterpret_cast<const f32x4*>(d.a_shift + ld_cc + k4);
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = ra[i] * *reinterpret_cast<const f32x4*>(sbase + a_soff[i]) + sh;
      }
#pragma unroll
      for (int i = 0; i < NA; ++i)
        if (!a_tap[i]) ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      const char* bbase = wbase + (int64_t)s * (BK * WELT);
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if (b_have[i]) {
#pragma unroll
          for (int p = 0; p < NP; ++p) rb[i][p] = *reinterpret_cast<const u32x4*>(bbase + p * plane1 + b_off[i]);
        }
      ld_cc += BK;
      if (ld_cc >= d.CinP) {
        ld_cc = 0;
        if (++ld_kx == d.KW) { ld_kx = 0; ++ld_ky; }
      }
    } else if (s < n1e) {   // null chunk
#pragma unroll
      for (int i = 0; i < NA; ++i) ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int p = 0; p < NP; ++p) rb[i][p] = u32x4{0u, 0u, 0u, 0u};
    } else {
      const int cb = (s - n1e) * BK;
      const char* abase = reinterpret_cast<const char*>(d.in2) + (int64_t)cb * 4;
      const bool cok = (cb + k4) < d.Cin2;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (a_ok[i] && cok) v = *reinterpret_cast<const f32x4*>(abase + a_off2[i]);
        ra[i] = v;
      }
      const char* bbase = wbase2 + (int64_t)cb * WELT;
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if (b_have[i]) {
#pragma unroll
          for (int p = 0; p < NP; ++p) rb[i][p] = *reinterpret_cast<const u32x4*>(bbase + p * plane2 + b_off2[i]);
        }
    }
    ++ld_step;
  };

  auto store_chunk = [&](const Regs& R, int buf) __attribute__((always_inline)) {
    const f32x4 (&ra)[NA] = R.a;
    const u32x4 (&rb)[NB][NP] = R.b;
    unsigned char* Ab = As0 + buf * A_BYTES;
    unsigned char* Bb = Bs0 + buf * B_BYTES;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = (tid + i * NT) >> 2;
      if constexpr (SPLIT) {
        u32x2 pl[NP];
        split4n<NP>(ra[i], amul, pl);
        const int off = row * ROWB + k4 * 2;
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(Ab + p * BM * ROWB + off) = pl[p];
      } else {
        *reinterpret_cast<f32x4*>(Ab + (row * LDKF + k4) * 4) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (b_have[i]) {
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(Bb + p * BN * ROWB + b_lds[i]) = rb[i][p];
      }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int col[TN];
  float bias1[TN], bias2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    col[j] = n0 + (wn * TN + j) * 32 + r;
    const bool ok = col[j] < d.N;
    bias1[j] = (ok && d.bias) ? d.bias[col[j]] : 0.f;
    bias2[j] = (ok && d.bias2) ? d.bias2[col[j]] : 0.f;
  }

  auto compute = [&](const int buf) __attribute__((always_inline)) {
    const unsigned char* Ab = As0 + buf * A_BYTES;
    const unsigned char* Bb = Bs0 + buf * B_BYTES;
    if (abl & 2) {
    } else if constexpr (SPLIT) {
      bf16x8 af[TM][NP], bf[TN][NP];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int p = 0; p < NP; ++p)
          af[i][p] = *reinterpret_cast<const bf16x8*>(Ab + p * BM * ROWB + ((wm * TM + i) * 32 + r) * ROWB + g * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int p = 0; p < NP; ++p)
          bf[j][p] = *reinterpret_cast<const bf16x8*>(Bb + p * BN * ROWB + ((wn * TN + j) * 32 + r) * ROWB + g * 16);
      // product term outermost, tiles innermost: consecutive MFMAs hit different accumulators (no dependent-issue stall);
      // smallest terms first
#pragma unroll
      for (int q = 0; q < AR::NPROD; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = AR::mfma(af[i][AR::PA[q]], bf[j][AR::PB[q]], acc[i][j]);
      }
    } else {
      // lanes 0-31 hold k 0..7, lanes 32-63 k 8..15 of the chunk; MFMA q multiplies the k pair (q, 8+q)
      f32x4 alo[TM], ahi[TM], blo[TN], bhi[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float* p = reinterpret_cast<const float*>(Ab) + ((wm * TM + i) * 32 + r) * LDKF + g * 8;
        alo[i] = *reinterpret_cast<const f32x4*>(p);
        ahi[i] = *reinterpret_cast<const f32x4*>(p + 4);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float* p = reinterpret_cast<const float*>(Bb) + ((wn * TN + j) * 32 + r) * LDKF + g * 8;
        blo[j] = *reinterpret_cast<const f32x4*>(p);
        bhi[j] = *reinterpret_cast<const f32x4*>(p + 4);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(alo[i][q], blo[j][q], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ahi[i][q], bhi[j][q], acc[i][j], 0, 0, 0);
    }
  };

  // ---- software pipeline, prefetch distance 2:   step s:  MFMA(chunk s) | store(chunk s+1 -> LDS) | barrier | load(chunk s+3)
  load_chunk(R0);
  store_chunk(R0, 0);
  if (total > 1) load_chunk(R1);
  if (total > 2) load_chunk(R0);
  __syncthreads();
  auto k_loop = [&](const int s_begin, const int s_end) __attribute__((always_inline)) {   // s_begin is even
    for (int s = s_begin; s < s_end; s += 2) {
      compute(0);
      if (s + 1 < total && !(abl & 4)) store_chunk(R1, 1);
      if (!(abl & 8)) __syncthreads();
      if (s + 3 < total && !(abl & 1)) load_chunk(R1);
      if (s + 1 >= s_end) break;
      compute(1);
      if (s + 2 < total && !(abl & 4)) store_chunk(R0, 0);
      if (!(abl & 8)) __syncthreads();
      if (s + 4 < total && !(abl & 1)) load_chunk(R0);
    }
  };
  k_loop(0, n1e);
  if constexpr (NPL == 2) scale_all<TM, TN>(acc, d.acc_mul);     // back to real units (exact: a power of two)
  if (n2 > 0) {   // phase 2: finish phase 1 in registers (bias, activation) and keep accumulating the 1x1 conv on top
    apply_act_all<TM, TN>(acc, bias1, bias2, d.act);
    if constexpr (NPL == 2) scale_all<TM, TN>(acc, 1.f / d.acc_mul2);   // into the units of the phase-2 products
    k_loop(n1e, total);
    if constexpr (NPL == 2) scale_all<TM, TN>(acc, d.acc_mul2);
  }

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  if (n2 == 0) {
    float zero[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) zero[j] = 0.f;
    apply_act_all<TM, TN>(acc, bias1, zero, d.act);
  }
  if (d.sumsq_part) write_sumsq<TM, TN>(acc, d.sumsq_part, d.N, m0 + (int64_t)wm * TM * 32, M, col, g);
  // stores through the helpers of conv_common.h: wave-uniform base + one 32-bit lane offset (whole tiles: residual values in one batch)
  {
    const int64_t row0 = m0 + (int64_t)wm * TM * 32;
    const int c0 = n0 + wn * TN * 32;
    char* const ob = reinterpret_cast<char*>(d.out + row0 * d.out_ld + d.out_coff + c0);
    const char* const rb = d.res ? reinterpret_cast<const char*>(d.res + row0 * d.res_ld + c0) : nullptr;
    const bool small = (int64_t)BM * max(d.out_ld, d.res_ld) * 4 < (1LL << 31);
    if (small && m0 + BM <= M && n0 + BN <= d.N) store_tile_full<TM, TN>(acc, ob, (int)d.out_ld, rb, (int)d.res_ld, r, g);
    else if (small) store_tile_guarded<TM, TN>(acc, ob, (int)d.out_ld, rb, (int)d.res_ld, r, g, (int)min((int64_t)TM * 32, (int64_t)M - row0), d.N - c0, d.n_store - c0);
    else {
  #pragma unroll
    for (int i = 0; i < TM; ++i) {
  #pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
        if (m >= M) continue;
  #pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = col[j];
          if (n >= d.n_store) continue;
          float v = 0.f;                      // columns in [N, n_store) are padding lanes: always zero
          if (n < d.N) {
            v = acc[i][j][e];
            if (d.res) v += d.res[m * d.res_ld + n];
          }
          d.out[m * d.out_ld + d.out_coff + n] = v;
        }
      }
    }
    }
  }
}

template <int WM, int WN, int TM, int TN, int SPLIT>
int launch(const vs_conv_desc_t& d, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int64_t M = (int64_t)d.B * d.Ho * d.Wo;
  const int64_t mt = cdiv64(M, BM), nt = cdiv64(d.n_store, BN);
  if (mt * nt > 0x7fffffffLL || M > 0x7fffffffLL) return VS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((conv_gemm_kernel<WM, WN, TM, TN, SPLIT>), dim3((unsigned)(mt * nt)), dim3(NT), 0, st, d, (int)M, (int)mt);
  return vs_launch_status();
}

template <int SPLIT>
int dispatch(const vs_conv_desc_t& d, int tile, hipStream_t st) {
  switch (tile) {
    case 1: return launch<2, 2, 2, 2, SPLIT>(d, st);   // 128 x 128
    case 2: return launch<2, 2, 2, 1, SPLIT>(d, st);   // 128 x 64
    case 3: return launch<4, 1, 2, 1, SPLIT>(d, st);   // 256 x 32
    case 4: return launch<2, 2, 2, 3, SPLIT>(d, st);   // 128 x 192
    case 5: return launch<4, 1, 1, 3, SPLIT>(d, st);   // 128 x 96
    case 13: return launch<2, 2, 1, 1, SPLIT>(d, st);  // 64 x 64   (small-M layers: more workgroups than CUs)
    case 14: return launch<2, 2, 1, 2, SPLIT>(d, st);  // 64 x 128
    default: return VS_ERR_UNSUPPORTED;
  }
}

}  // namespace
int vs_conv3x3_patch_dispatch(const vs_conv_desc_t& d, int tile, hipStream_t st);  // conv3x3_patch.hip
int vs_conv3x3_patch_pc_dispatch(const vs_conv_desc_t& d, int tile, hipStream_t st);  // conv3x3_patch_pc.hip
int vs_gemm1x1_pc_dispatch(const vs_conv_desc_t& d, int tile, hipStream_t st);         // gemm1x1_pc.hip
int vs_conv3x3_small_dispatch(const vs_conv_desc_t& d, hipStream_t st);                // conv3x3_small.hip
int vs_conv3x3_pl_dispatch(const vs_conv_desc_t& d, int tile, hipStream_t st);          // conv3x3_pl.hip
int vs_gemm_pl_dispatch(const vs_conv_desc_t& d, int tile, hipStream_t st);             // gemm_pl.hip
namespace {

inline bool fits_u32(int64_t bytes) { return bytes >= 0 && bytes < 0xffffffffLL; }

// the wave-specialised 3x3 kernel reads / writes the epilogue operands in float4 channel groups
inline bool pc_vec_ok(const vs_conv_desc_t& d) {
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  return d.n_store % 4 == 0 && d.n_store >= ((d.N + 3) & ~3) && d.out_ld % 4 == 0 && d.out_coff % 4 == 0 && al16(d.out) && al16(d.bias) && al16(d.bias2) &&
         (!d.res || (d.res_ld % 4 == 0 && al16(d.res))) && (d.split_k <= 1 || (d.splitk_ld % 4 == 0 && al16(d.splitk_ws))) &&
         (!(d.tile_hint & VS_CONV_PRE) || (d.a_scale_ld % 4 == 0 && al16(d.a_scale)));
}

}  // namespace

// tile codes 22 / 23: every operand as pre-split planes (conv3x3_pl.hip)
static int conv_planes(const vs_conv_desc_t& d, int tile, hipStream_t st) {
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  VS_REQUIRE(d.arith == 2 && d.a_mul > 0.f && d.acc_mul > 0.f && d.in_pl && d.wt_blk && (d.out || d.out_pl));
  VS_REQUIRE(d.B > 0 && d.H > 0 && d.W > 0 && d.N > 0 && d.KH == 3 && d.KW == 3 && d.SH == 1 && d.SW == 1 && d.PH == 1 && d.PW == 1);
  VS_REQUIRE(d.Ho == d.H && d.Wo == d.W && d.pad_mode == VS_PAD_ZERO && d.CinP >= BK && d.CinP % BK == 0);
  VS_REQUIRE(al16(d.in_pl) && al16(d.wt_blk) && al16(d.bias) && al16(d.bias2) && !d.sumsq_part);
  if (d.split_k > 1) VS_REQUIRE(d.splitk_ws && al16(d.splitk_ws) && d.splitk_ld >= d.N && d.splitk_ld % 4 == 0 && !(d.tile_hint & VS_CONV_PRE) && !d.res);
  if (d.H % 16 || d.W % 16) return VS_ERR_UNSUPPORTED;
  if (2 * (int64_t)d.B * d.H * d.W * std::max(d.CinP, d.in2_pl ? d.Cin2P : 0) * 2 >= 0xffffffffLL) return VS_ERR_UNSUPPORTED;   // 32-bit DMA offsets
  if (d.in2_pl) VS_REQUIRE(d.wt2_blk && al16(d.in2_pl) && al16(d.wt2_blk) && d.Cin2P >= BK && d.Cin2P % BK == 0 && d.acc_mul2 > 0.f);
  if (d.out) VS_REQUIRE(al16(d.out) && d.out_ld % 4 == 0 && d.out_coff % 4 == 0 && d.n_store % 4 == 0 && d.n_store >= ((d.N + 3) & ~3) &&
                        d.out_coff >= 0 && d.out_coff + d.n_store <= d.out_ld);
  else VS_REQUIRE(d.n_store >= d.N);
  if (d.out_pl) VS_REQUIRE(al16(d.out_pl) && d.N % 16 == 0);
  if (d.res) VS_REQUIRE(d.res_ld >= d.N && d.res_ld % 4 == 0 && al16(d.res));
  if (d.tile_hint & VS_CONV_PRE) VS_REQUIRE(d.a_scale && (d.a_scale_ld == 0 || d.a_scale_ld >= 9 * (int64_t)d.N) && d.a_scale_ld % 4 == 0 && al16(d.a_scale));
  else VS_REQUIRE(!d.a_scale);
  return vs_conv3x3_pl_dispatch(d, tile, st);
}

// tile codes 24 / 25: 1x1 GEMM with the activations as pre-split planes (gemm_pl.hip)
static int gemm_planes(const vs_conv_desc_t& d, int tile, hipStream_t st) {
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  VS_REQUIRE(d.arith == 2 && d.a_mul > 0.f && d.acc_mul > 0.f && d.in_pl && d.wt_blk && d.out && !d.in2 && !d.in2_pl && !d.out_pl && !d.a_scale);
  VS_REQUIRE(d.B > 0 && d.H > 0 && d.W > 0 && d.N > 0 && d.KH == 1 && d.KW == 1 && d.SH == 1 && d.SW == 1 && d.PH == 0 && d.PW == 0);
  VS_REQUIRE(d.Ho == d.H && d.Wo == d.W && d.CinP >= BK && d.CinP % BK == 0 && al16(d.in_pl) && al16(d.wt_blk));
  VS_REQUIRE(d.n_store >= d.N && d.out_coff >= 0 && d.out_coff + d.n_store <= d.out_ld);
  if (d.res) VS_REQUIRE(d.res_ld >= d.N);
  if (d.sumsq_part) VS_REQUIRE(d.split_k <= 1 && !d.res);
  if (d.sumsq_hw) VS_REQUIRE(d.sumsq_part && d.sumsq_hw >= 32 && d.sumsq_hw == d.H * d.W);      // straddling 32-row groups: [M/32][2][N] partials
  if (d.split_k > 1) VS_REQUIRE(d.splitk_ws && d.splitk_ld >= d.N && d.split_k <= d.CinP / BK);
  const int64_t M = (int64_t)d.B * d.H * d.W;
  if (2 * M * d.CinP * 2 >= 0xffffffffLL) return VS_ERR_UNSUPPORTED;          // 32-bit DMA offsets
  return vs_gemm_pl_dispatch(d, tile, st);
}

extern "C" int vs_conv_gemm(const vs_conv_desc_t* dp, void* stream) {
  VS_REQUIRE(dp);
  {
    const int t = (dp->tile_hint & 0xf) + ((dp->tile_hint & VS_CONV_TILE_HI) ? 16 : 0);
    if (t == 22 || t == 23) return conv_planes(*dp, t, (hipStream_t)stream);
    if (t == 24 || t == 25 || t == 27) return gemm_planes(*dp, t, (hipStream_t)stream);
  }
  VS_REQUIRE(dp->in && dp->wt && dp->out);
  const vs_conv_desc_t& d = *dp;
  VS_REQUIRE(d.B > 0 && d.H > 0 && d.W > 0 && d.Ho > 0 && d.Wo > 0 && d.N > 0 && d.Cin > 0);
  VS_REQUIRE(d.KH > 0 && d.KW > 0 && d.SH > 0 && d.SW > 0 && d.PH >= 0 && d.PW >= 0);
  VS_REQUIRE(d.Cin % 4 == 0 && d.CinP % BK == 0 && d.CinP >= d.Cin);
  VS_REQUIRE(d.in_sx % 4 == 0 && d.in_sy % 4 == 0 && d.in_sb % 4 == 0 && d.in_sx > 0 && d.in_sy > 0 && d.in_sb >= 0);
  VS_REQUIRE(d.n_store >= d.N && d.out_coff >= 0 && d.out_coff + d.n_store <= d.out_ld);
  VS_REQUIRE(((uintptr_t)d.in & 15) == 0 && ((uintptr_t)d.wt & 15) == 0);
  if (d.pad_mode == VS_PAD_REFLECT) VS_REQUIRE(d.PH < d.H && d.PW < d.W);
  const bool pre_add = (d.tile_hint & VS_CONV_PRE) != 0;      // a_scale = pre-activation addend of the wave-specialised 3x3 kernel
  if (pre_add) VS_REQUIRE(d.a_scale && (d.a_scale_ld == 0 || d.a_scale_ld >= 9 * (int64_t)d.N) && d.split_k <= 1 && d.H > 1 && d.W > 1);
  else if (d.a_scale) VS_REQUIRE(d.a_shift && d.KH == 1 && d.KW == 1 && d.a_scale_ld % 4 == 0);
  if (d.in2) VS_REQUIRE(d.wt2 && d.Cin2 > 0 && d.Cin2 % 4 == 0 && d.Cin2P % BK == 0 && d.Cin2P >= d.Cin2 && d.in2_ld % 4 == 0);
  if (d.res) VS_REQUIRE(d.res_ld >= d.N);
  // the kernels address every operand as base + 32-bit byte offset
  const int64_t M = (int64_t)d.B * d.Ho * d.Wo;
  const int64_t Ktot = (int64_t)d.KH * d.KW * d.CinP;
  if (!fits_u32(((int64_t)d.B * d.in_sb + (int64_t)(d.H + d.PH) * d.in_sy + (int64_t)d.W * d.in_sx) * 4) ||
      !fits_u32((int64_t)d.N * Ktot * 4) || (d.in2 && !fits_u32(M * d.in2_ld * 4)) ||
      (d.a_scale && !pre_add && !fits_u32((int64_t)d.B * d.a_scale_ld * 4)))
    return VS_ERR_UNSUPPORTED;   // > 4 GiB operand: the caller chunks the batch
  hipStream_t st = (hipStream_t)stream;
  int tile = (d.tile_hint & 0xf) + ((d.tile_hint & VS_CONV_TILE_HI) ? 16 : 0);
  // ABI v3: the GRN finish folded into the GEMM exists in the wave-specialised 1x1 kernel only; a_scale is not written by anybody in that
  // mode, so every other route must refuse instead of reading it
  if (d.grn_part && !(tile == 17 || tile == 18 || tile == 26)) return VS_ERR_UNSUPPORTED;
  if (d.sumsq_hw && !(tile == 24 || tile == 25 || tile == 27)) return VS_ERR_UNSUPPORTED;                      // (the [M/32][2][N] partials exist in gemm_pl.hip only)
  const bool can_split0 = d.wt_split && (!d.in2 || d.wt2_split) && ((uintptr_t)d.wt_split & 15) == 0;
  // 3x3 / stride 1 / "same" convs on tile-aligned frames go to the patch kernel (input patch staged once per channel chunk)
  const bool patch_ok = can_split0 && d.KH == 3 && d.KW == 3 && d.SH == 1 && d.SW == 1 && d.PH == 1 && d.PW == 1 && d.Ho == d.H &&
                        d.Wo == d.W && (!d.a_scale || pre_add) && !(d.tile_hint & VS_CONV_FORCE_F32);
  if (pre_add) VS_REQUIRE(patch_ok && (tile == 15 || tile == 16));
  if (tile >= 10 && tile <= 12) {
    VS_REQUIRE(patch_ok);
    return vs_conv3x3_patch_dispatch(d, tile, st);
  }
  VS_REQUIRE(d.arith == 0 || d.arith == 2 || d.arith == 3);
  if (d.arith == 2) VS_REQUIRE(d.a_mul > 0.f && d.acc_mul > 0.f && (!d.in2 || d.acc_mul2 > 0.f));
  if (tile == 15 || tile == 16 || tile == 19 || tile == 21) {   // wave-specialised patch kernel
    VS_REQUIRE(patch_ok && d.wt_blk && (!d.in2 || d.wt2_blk) && ((uintptr_t)d.wt_blk & 15) == 0);
    if (d.split_k > 1) VS_REQUIRE(d.splitk_ws && d.splitk_ld >= d.N && d.split_k <= d.CinP / 16 && !d.sumsq_part);
    if (!pc_vec_ok(d)) return VS_ERR_UNSUPPORTED;
    return vs_conv3x3_patch_pc_dispatch(d, tile, st);
  }
  // thin full-resolution layers (16 input channels, <= 32 outputs): persistent kernel with the weights in registers
  const bool small_ok = patch_ok && d.CinP == 16 && d.N <= 32 && d.n_store <= 32 && (!d.in2 || d.Cin2P == 16) && d.split_k <= 1 &&
                        !d.sumsq_part;
  if (tile == 20) {
    VS_REQUIRE(small_ok);
    return vs_conv3x3_small_dispatch(d, st);
  }
  if (tile == 17 || tile == 18 || tile == 26) {   // wave-specialised 1x1 GEMM: dense rows, whole 32-wide K pairs, no second phase
    if (tile == 26) VS_REQUIRE(d.arith == 2);     // 128 x 96 tiles: 2 x f16 arithmetic only
    VS_REQUIRE(can_split0 && d.wt_blk && ((uintptr_t)d.wt_blk & 15) == 0 && !(d.tile_hint & VS_CONV_FORCE_F32));
    VS_REQUIRE(d.KH == 1 && d.KW == 1 && d.SH == 1 && d.SW == 1 && d.PH == 0 && d.PW == 0 && d.Ho == d.H && d.Wo == d.W && !d.in2);
    VS_REQUIRE(d.Cin % 32 == 0 && d.CinP == d.Cin && d.in_sy == (int64_t)d.W * d.in_sx && d.in_sb == (int64_t)d.H * d.in_sy);
    if (d.a_scale) VS_REQUIRE(d.H * d.W >= 128 || d.H * d.W == 64);   // a 128-row tile touches at most two frames
    if (d.sumsq_part) VS_REQUIRE(d.split_k <= 1 && !d.res);
    if (d.split_k > 1) VS_REQUIRE(d.splitk_ws && d.splitk_ld >= d.N && d.split_k <= d.CinP / 32);
    return vs_gemm1x1_pc_dispatch(d, tile, st);
  }
  VS_REQUIRE(d.split_k <= 1);
  if (d.sumsq_part) VS_REQUIRE(d.KH == 1 && d.KW == 1 && !d.in2 && !d.res && !patch_ok);
  if (tile == 0 && small_ok && !d.in2) return vs_conv3x3_small_dispatch(d, st);   // (with the fused 1x1 the patch kernel is as fast)
  if (tile == 0 && patch_ok && d.W % 16 == 0 && d.H % 8 == 0) {
    const bool blk_ok = d.wt_blk && (!d.in2 || d.wt2_blk) && ((uintptr_t)d.wt_blk & 15) == 0;
    if (blk_ok && pc_vec_ok(d) && d.N >= 128) return vs_conv3x3_patch_pc_dispatch(d, d.N % 192 == 0 ? 16 : 15, st);   // wave-specialised for wide layers
    if (blk_ok && pc_vec_ok(d) && d.N > 32 && d.N <= 64 && d.H % 16 == 0 && d.CinP >= 64) return vs_conv3x3_patch_pc_dispatch(d, 21, st);   // 256 px x 64 ch
    return vs_conv3x3_patch_dispatch(d, d.N <= 32 ? 10 : (d.N <= 64 ? 11 : 12), st);
  }
  if (tile == 0) {
    // static choice (no tuner: model-level C API, hipGraph capture, VIDEOSEAL_AUTOTUNE=0), following what the tuner measures on
    // the benchmark shapes: deep-K 1x1 GEMMs on the wave-specialised kernel, shallow-K ones on 128x128 generic tiles
    const bool gemm_ok = can_split0 && d.wt_blk && ((uintptr_t)d.wt_blk & 15) == 0 && !(d.tile_hint & VS_CONV_FORCE_F32) && d.KH == 1 &&
                         d.KW == 1 && d.SH == 1 && d.SW == 1 && d.PH == 0 && d.PW == 0 && d.Ho == d.H && d.Wo == d.W && !d.in2 &&
                         d.Cin % 32 == 0 && d.CinP == d.Cin && d.in_sy == (int64_t)d.W * d.in_sx && d.in_sb == (int64_t)d.H * d.in_sy &&
                         (!d.a_scale || d.H * d.W >= 128 || d.H * d.W == 64) && !(d.sumsq_part && d.res) &&
                         (!d.a_scale || (int64_t)((d.CinP / 32 + std::max(1, d.split_k) - 1) / std::max(1, d.split_k)) * 32 <= 3072);   // GRN rows of a K slice in LDS
    if (gemm_ok && d.CinP >= 384 && d.N > 128 && !(d.act == VS_ACT_TANH && d.split_k <= 1)) return vs_gemm1x1_pc_dispatch(d, 18, st);      // (tile 18 has no tanh epilogue)
    if (d.KH == 1 && d.KW == 1 && d.CinP < 384 && d.N >= 256) tile = 1;
    else tile = d.N <= 32 ? 3 : (d.N <= 64 ? 2 : (d.N <= 96 ? 5 : ((d.N % 192 == 0 || (d.N > 128 && d.N <= 192)) ? 4 : 1)));
    // few output tiles (down-sampling convs, the 8x8 head conv): smaller tiles until the 256 CUs have work
    auto blocks_of = [&](int t) -> int64_t {
      const int bm = (t == 13 || t == 14) ? 64 : (t == 3 ? 256 : 128);
      const int bn = t == 1 ? 128 : t == 2 ? 64 : t == 3 ? 32 : t == 4 ? 192 : t == 5 ? 96 : t == 13 ? 64 : 128;
      return cdiv64(M, bm) * cdiv64(d.n_store, bn);
    };
    if (d.N > 32 && blocks_of(tile) < 384) {
      if (tile == 4 || tile == 1) tile = blocks_of(2) >= 384 ? 2 : 13;
      else if (tile == 5 || tile == 2) tile = 13;
    }
  }
  if (tile >= 6 && tile <= 9) return VS_ERR_UNSUPPORTED;   // (codes of a retired generic producer/consumer kernel)
  const bool can_split = d.wt_split && (!d.in2 || d.wt2_split) && ((uintptr_t)d.wt_split & 15) == 0;
  if (d.tile_hint & VS_CONV_FORCE_SPLIT) VS_REQUIRE(can_split);
  if (can_split && !(d.tile_hint & VS_CONV_FORCE_F32)) return d.arith == 2 ? dispatch<2>(d, tile, st) : dispatch<3>(d, tile, st);
  return dispatch<0>(d, tile, st);
}
