// Implicit-GEMM convolution / linear layer on the gfx950 fp32 matrix cores.
//
//   out[m][n] = epilogue( sum_k A[m][k] * wt[n][k] ),  m = (frame, oy, ox) row-major, k = (tap, channel)
//
// Design (MI355X_MICROARCH.md: f32-input MFMA = 64 FLOP/clk/SIMD = 157 TF, exact fmaf chain):
//   * v_mfma_f32_32x32x2_f32; a 4-wave workgroup owns a BM x BN output tile, each wave TM x TN tiles of 32x32.
//   * K is walked in chunks of 16 channels of one filter tap.  The A chunk is gathered straight from the NHWC
//     activation (zero or reflect padding resolved per row), the B chunk from the pre-packed [N][Ktot] weight.
//     Both are staged global -> registers -> LDS with two LDS buffers: the global loads of chunk s+1 are in
//     flight while chunk s is multiplied; one barrier per chunk.
//   * LDS rows are [row][16 + 4 pad] floats: a lane reads its 8 k-values with two ds_read_b128 and the 20-dword
//     row stride makes every 16-lane service group of ds_read_b128 hit 16 distinct 4-bank slots (conflict-free).
//     Lanes 0-31 take k 0..7 and lanes 32-63 k 8..15 of the chunk; MFMA j multiplies k-pair (j, 8+j).
//   * Epilogue in registers: bias, ReLU/GELU/tanh, optional second K phase (the ResnetBlock's 1x1 res_conv on
//     the block input, accumulated on top of relu(bn(conv))), optional residual, coalesced 128-B row stores.
#include "vs_common.h"

namespace {

constexpr int BK = 16;
constexpr int LDK = BK + 4;   // padded LDS row (floats)
constexpr int NT = 256;

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(NT) void conv_gemm_kernel(const vs_conv_desc_t d, const int M, const int mtiles) {
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int NA = (BM * 4 + NT - 1) / NT;   // float4 A loads per thread per chunk
  constexpr int NB = (BN * 4 + NT - 1) / NT;
  static_assert(WM * WN == 4, "4 waves per workgroup");

  __shared__ __attribute__((aligned(16))) float As[2][BM][LDK];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN][LDK];

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
  const int total = n1 + n2;
  const int64_t Ktot = (int64_t)d.KH * d.KW * d.CinP;
  const int HoWo = d.Ho * d.Wo;

  // ---- per-thread A rows (fixed for the whole K loop)
  int a_row[NA];
  bool a_ok[NA];
  int a_iy0[NA], a_ix0[NA], a_b[NA];
  int64_t a_m[NA];
  const int k4 = (tid & 3) * 4;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int idx = tid + i * NT;
    const int row = idx >> 2;
    a_row[i] = row;
    const int64_t m = m0 + row;
    a_m[i] = m;
    a_ok[i] = (row < BM) && (m < M);
    const int64_t mm = a_ok[i] ? m : 0;
    const int b = (int)(mm / HoWo);
    const int rem = (int)(mm - (int64_t)b * HoWo);
    const int oy = rem / d.Wo, ox = rem - oy * d.Wo;
    a_b[i] = b;
    a_iy0[i] = oy * d.SH - d.PH;
    a_ix0[i] = ox * d.SW - d.PW;
  }
  int b_row[NB];
  bool b_ok[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int idx = tid + i * NT;
    b_row[i] = idx >> 2;
    b_ok[i] = (b_row[i] < BN) && (n0 + b_row[i] < d.N);
  }

  // running position of the loader inside phase 1
  int ld_ky = 0, ld_kx = 0, ld_cc = 0, ld_step = 0;
  f32x4 ra[NA], rb[NB];

  auto load_chunk = [&]() {
    const int s = ld_step;
    if (s < n1) {
      const int c = ld_cc + k4;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        int iy = a_iy0[i] + ld_ky, ix = a_ix0[i] + ld_kx;
        bool ok = a_ok[i] && (c < d.Cin);
        if (d.pad_mode == VS_PAD_REFLECT) {
          iy = iy < 0 ? -iy : (iy >= d.H ? 2 * d.H - 2 - iy : iy);
          ix = ix < 0 ? -ix : (ix >= d.W ? 2 * d.W - 2 - ix : ix);
        } else {
          ok = ok && (iy >= 0) && (iy < d.H) && (ix >= 0) && (ix < d.W);
        }
        if (ok) {
          v = *reinterpret_cast<const f32x4*>(d.in + (int64_t)a_b[i] * d.in_sb + (int64_t)iy * d.in_sy +
                                              (int64_t)ix * d.in_sx + c);
          if (d.a_scale) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(d.a_scale + (int64_t)a_b[i] * d.a_scale_ld + c);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(d.a_shift + c);
            v = v * sc + sh;
          }
        }
        ra[i] = v;
      }
      const int64_t koff = (int64_t)s * BK + k4;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b_ok[i]) v = *reinterpret_cast<const f32x4*>(d.wt + (int64_t)(n0 + b_row[i]) * Ktot + koff);
        rb[i] = v;
      }
      ld_cc += BK;
      if (ld_cc >= d.CinP) {
        ld_cc = 0;
        if (++ld_kx == d.KW) { ld_kx = 0; ++ld_ky; }
      }
    } else {
      const int c = (s - n1) * BK + k4;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (a_ok[i] && c < d.Cin2) v = *reinterpret_cast<const f32x4*>(d.in2 + a_m[i] * d.in2_ld + c);
        ra[i] = v;
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b_ok[i]) v = *reinterpret_cast<const f32x4*>(d.wt2 + (int64_t)(n0 + b_row[i]) * d.Cin2P + c);
        rb[i] = v;
      }
    }
    ++ld_step;
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (a_row[i] < BM) *reinterpret_cast<f32x4*>(&As[buf][a_row[i]][k4]) = ra[i];
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (b_row[i] < BN) *reinterpret_cast<f32x4*>(&Bs[buf][b_row[i]][k4]) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // per-lane epilogue constants: one output column per (tn)
  int col[TN];
  float bias1[TN], bias2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    col[j] = n0 + (wn * TN + j) * 32 + r;
    const bool ok = col[j] < d.N;
    bias1[j] = (ok && d.bias) ? d.bias[col[j]] : 0.f;
    bias2[j] = (ok && d.bias2) ? d.bias2[col[j]] : 0.f;
  }

  load_chunk();
  store_chunk(0);
  __syncthreads();

  for (int s = 0; s < total; ++s) {
    const int buf = s & 1;
    if (s + 1 < total) load_chunk();
    if (s == n1) {   // entering phase 2: finish phase 1 in registers, keep accumulating on top of it
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = vs_apply_act(acc[i][j][e] + bias1[j], d.act) + bias2[j];
    }
    f32x4 alo[TM], ahi[TM], blo[TN], bhi[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float* p = &As[buf][(wm * TM + i) * 32 + r][g * 8];
      alo[i] = *reinterpret_cast<const f32x4*>(p);
      ahi[i] = *reinterpret_cast<const f32x4*>(p + 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float* p = &Bs[buf][(wn * TN + j) * 32 + r][g * 8];
      blo[j] = *reinterpret_cast<const f32x4*>(p);
      bhi[j] = *reinterpret_cast<const f32x4*>(p + 4);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(alo[i][k], blo[j][k], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ahi[i][k], bhi[j][k], acc[i][j], 0, 0, 0);
    if (s + 1 < total) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const bool two_phase = n2 > 0;
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
        float v = acc[i][j][e];
        if (!two_phase) v = vs_apply_act(v + bias1[j], d.act);
        if (d.res && n < d.N) v += d.res[m * d.res_ld + n];
        d.out[m * d.out_ld + d.out_coff + n] = v;
      }
    }
  }
}

template <int WM, int WN, int TM, int TN>
int launch(const vs_conv_desc_t& d, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int64_t M = (int64_t)d.B * d.Ho * d.Wo;
  const int64_t mt = cdiv64(M, BM), nt = cdiv64(d.n_store, BN);
  if (mt * nt > 0x7fffffffLL || M > 0x7fffffffLL) return VS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((conv_gemm_kernel<WM, WN, TM, TN>), dim3((unsigned)(mt * nt)), dim3(NT), 0, st, d, (int)M, (int)mt);
  return vs_launch_status();
}

}  // namespace

int vs_conv_gemm_split_dispatch(const vs_conv_desc_t& d, int tile, hipStream_t st);   // conv_gemm_split.hip

extern "C" int vs_conv_gemm(const vs_conv_desc_t* dp, void* stream) {
  VS_REQUIRE(dp && dp->in && dp->wt && dp->out);
  const vs_conv_desc_t& d = *dp;
  VS_REQUIRE(d.B > 0 && d.H > 0 && d.W > 0 && d.Ho > 0 && d.Wo > 0 && d.N > 0 && d.Cin > 0);
  VS_REQUIRE(d.KH > 0 && d.KW > 0 && d.SH > 0 && d.SW > 0);
  VS_REQUIRE(d.Cin % 4 == 0 && d.CinP % BK == 0 && d.CinP >= d.Cin);
  VS_REQUIRE(d.in_sx % 4 == 0 && d.in_sy % 4 == 0 && d.in_sb % 4 == 0);
  VS_REQUIRE(d.n_store >= d.N && d.out_coff >= 0 && d.out_coff + d.n_store <= d.out_ld);
  VS_REQUIRE(((uintptr_t)d.in & 15) == 0 && ((uintptr_t)d.wt & 15) == 0);
  if (d.pad_mode == VS_PAD_REFLECT) VS_REQUIRE(d.PH < d.H && d.PW < d.W);
  if (d.a_scale) VS_REQUIRE(d.a_shift && d.KH == 1 && d.KW == 1 && d.a_scale_ld % 4 == 0);
  if (d.in2) VS_REQUIRE(d.wt2 && d.Cin2 > 0 && d.Cin2 % 4 == 0 && d.Cin2P % BK == 0 && d.Cin2P >= d.Cin2 && d.in2_ld % 4 == 0);
  if (d.res) VS_REQUIRE(d.res_ld >= d.N);
  hipStream_t st = (hipStream_t)stream;
  int tile = d.tile_hint & 0xf;
  if (tile == 0) tile = d.N <= 32 ? 3 : (d.N <= 64 ? 2 : 1);
  const bool can_split = d.wt_split && (!d.in2 || d.wt2_split) && ((uintptr_t)d.wt_split & 15) == 0;
  if (d.tile_hint & VS_CONV_FORCE_SPLIT) VS_REQUIRE(can_split);
  if (can_split && !(d.tile_hint & VS_CONV_FORCE_F32)) return vs_conv_gemm_split_dispatch(d, tile, st);
  switch (tile) {
    case 1: return launch<2, 2, 2, 2>(d, st);   // 128 x 128
    case 2: return launch<2, 2, 2, 1>(d, st);   // 128 x 64
    case 3: return launch<4, 1, 2, 1>(d, st);   // 256 x 32
    default: return VS_ERR_UNSUPPORTED;
  }
}
