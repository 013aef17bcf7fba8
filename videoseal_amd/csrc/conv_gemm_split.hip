// Implicit-GEMM convolution with fp32-accurate results on the bf16 matrix cores ("3 x bf16" operand splitting).
//
// Every fp32 operand x is split EXACTLY into three bf16 terms by truncation, x = x1 + x2 + x3 (8 + 8 + 8 mantissa
// bits), and a product is accumulated as the six partial products of weight <= 2^-16:
//     a*b ~= a1*b1 + (a1*b2 + a2*b1) + (a2*b2 + a1*b3 + a3*b1)            (dropped terms <= 2^-24 |a*b|)
// each of them one v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  bf16 MFMA runs at 16x the rate of the
// f32-input MFMA, so six of them are 2.67x faster than v_mfma_f32_32x32x2_f32 at the same (fp32-rounding-level)
// accuracy -- this is what keeps the bit decisions identical to the fp32 reference while leaving the 157 TF
// fp32 roof.  Weights are pre-split once at pack time ([3][N][Ktot] bf16); activations are split on the fly
// while they are staged global -> registers -> LDS (2 v_and + 2 v_sub + 1.5 v_perm per element).
//
// Tiling is the same as conv_gemm.hip: 4 waves, BM x BN output tile, K walked in 16-channel chunks of one tap,
// double-buffered LDS.  LDS rows are 16 bf16 + 8 pad (48 B): the 12-dword stride puts the 16 lanes of every
// ds_read_b128 service group on 16 distinct 4-bank slots.
#include "vs_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 16;
constexpr int ROWB = 48;      // bytes per LDS row (16 bf16 + pad)
constexpr int NT = 256;

__device__ __forceinline__ unsigned pack_hi(unsigned lo_elem, unsigned hi_elem) {
  // {hi16(lo_elem), hi16(hi_elem)} as two bf16 in one dword
  return __builtin_amdgcn_perm(hi_elem, lo_elem, 0x07060302);
}

// split 4 floats into 3 planes of 4 bf16 (8 bytes each)
__device__ __forceinline__ void split4(const f32x4& v, u32x2& p1, u32x2& p2, u32x2& p3) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned u = __float_as_uint(v[i]);
    h[i] = u & 0xffff0000u;
    const float r = v[i] - __uint_as_float(h[i]);
    const unsigned ur = __float_as_uint(r);
    m[i] = ur & 0xffff0000u;
    const float s = r - __uint_as_float(m[i]);
    l[i] = __float_as_uint(s);
  }
  p1 = u32x2{pack_hi(h[0], h[1]), pack_hi(h[2], h[3])};
  p2 = u32x2{pack_hi(m[0], m[1]), pack_hi(m[2], m[3])};
  p3 = u32x2{pack_hi(l[0], l[1]), pack_hi(l[2], l[3])};
}

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(NT) void conv_gemm_split_kernel(const vs_conv_desc_t d, const int M, const int mtiles) {
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int NA = (BM * 4 + NT - 1) / NT;   // float4 A loads per thread per chunk
  constexpr int NBL = (BN * 2 + NT - 1) / NT;  // (row, half) B slots per thread per chunk; each slot = 3 planes x 16 B
  static_assert(WM * WN == 4, "4 waves per workgroup");

  __shared__ __attribute__((aligned(16))) unsigned char As[2][3][BM * ROWB];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[2][3][BN * ROWB];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, g = lane >> 5;

  const int bm = blockIdx.x % mtiles;
  const int bn = blockIdx.x / mtiles;
  const int64_t m0 = (int64_t)bm * BM;
  const int n0 = bn * BN;

  const int spt = d.CinP / BK;
  const int n1 = d.KH * d.KW * spt;
  const int n2 = d.in2 ? d.Cin2P / BK : 0;
  const int total = n1 + n2;
  const int64_t Ktot = (int64_t)d.KH * d.KW * d.CinP;
  const int HoWo = d.Ho * d.Wo;
  const __bf16* wsp = reinterpret_cast<const __bf16*>(d.wt_split);
  const __bf16* wsp2 = reinterpret_cast<const __bf16*>(d.wt2_split);
  const int64_t plane1 = (int64_t)d.N * Ktot;          // elements per weight plane
  const int64_t plane2 = (int64_t)d.N * d.Cin2P;

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
  int b_row[NBL], b_half[NBL];
  bool b_ok[NBL];
#pragma unroll
  for (int i = 0; i < NBL; ++i) {
    const int idx = tid + i * NT;
    b_row[i] = idx >> 1;
    b_half[i] = idx & 1;
    b_ok[i] = (b_row[i] < BN) && (n0 + b_row[i] < d.N);
  }

  int ld_ky = 0, ld_kx = 0, ld_cc = 0, ld_step = 0;
  f32x4 ra[NA];
  u32x4 rb[NBL][3];

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
      const int64_t kbase = (int64_t)s * BK;
#pragma unroll
      for (int i = 0; i < NBL; ++i) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (b_ok[i])
            v = *reinterpret_cast<const u32x4*>(wsp + p * plane1 + (int64_t)(n0 + b_row[i]) * Ktot + kbase + b_half[i] * 8);
          rb[i][p] = v;
        }
      }
      ld_cc += BK;
      if (ld_cc >= d.CinP) {
        ld_cc = 0;
        if (++ld_kx == d.KW) { ld_kx = 0; ++ld_ky; }
      }
    } else {
      const int cb = (s - n1) * BK;
      const int c = cb + k4;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (a_ok[i] && c < d.Cin2) v = *reinterpret_cast<const f32x4*>(d.in2 + a_m[i] * d.in2_ld + c);
        ra[i] = v;
      }
#pragma unroll
      for (int i = 0; i < NBL; ++i) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (b_ok[i])
            v = *reinterpret_cast<const u32x4*>(wsp2 + p * plane2 + (int64_t)(n0 + b_row[i]) * d.Cin2P + cb + b_half[i] * 8);
          rb[i][p] = v;
        }
      }
    }
    ++ld_step;
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (a_row[i] < BM) {
        u32x2 p1, p2, p3;
        split4(ra[i], p1, p2, p3);
        const int off = a_row[i] * ROWB + k4 * 2;
        *reinterpret_cast<u32x2*>(&As[buf][0][off]) = p1;
        *reinterpret_cast<u32x2*>(&As[buf][1][off]) = p2;
        *reinterpret_cast<u32x2*>(&As[buf][2][off]) = p3;
      }
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      if (b_row[i] < BN) {
        const int off = b_row[i] * ROWB + b_half[i] * 16;
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(&Bs[buf][p][off]) = rb[i][p];
      }
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

  load_chunk();
  store_chunk(0);
  __syncthreads();

  for (int s = 0; s < total; ++s) {
    const int buf = s & 1;
    if (s + 1 < total) load_chunk();
    if (s == n1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = vs_apply_act(acc[i][j][e] + bias1[j], d.act) + bias2[j];
    }
    bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        af[i][p] = *reinterpret_cast<const bf16x8*>(&As[buf][p][((wm * TM + i) * 32 + r) * ROWB + g * 16]);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        bf[j][p] = *reinterpret_cast<const bf16x8*>(&Bs[buf][p][((wn * TN + j) * 32 + r) * ROWB + g * 16]);
    // smallest terms first
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        f32x16 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], c, 0, 0, 0);
        acc[i][j] = c;
      }
    if (s + 1 < total) store_chunk(buf ^ 1);
    __syncthreads();
  }

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
int launch_split(const vs_conv_desc_t& d, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int64_t M = (int64_t)d.B * d.Ho * d.Wo;
  const int64_t mt = cdiv64(M, BM), nt = cdiv64(d.n_store, BN);
  if (mt * nt > 0x7fffffffLL || M > 0x7fffffffLL) return VS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((conv_gemm_split_kernel<WM, WN, TM, TN>), dim3((unsigned)(mt * nt)), dim3(NT), 0, st, d, (int)M, (int)mt);
  return vs_launch_status();
}

}  // namespace

// called by vs_conv_gemm (conv_gemm.hip) after argument validation
int vs_conv_gemm_split_dispatch(const vs_conv_desc_t& d, int tile, hipStream_t st) {
  switch (tile) {
    case 1: return launch_split<2, 2, 2, 2>(d, st);   // 128 x 128
    case 2: return launch_split<2, 2, 2, 1>(d, st);   // 128 x 64
    case 3: return launch_split<4, 1, 2, 1>(d, st);   // 256 x 32
    default: return VS_ERR_UNSUPPORTED;
  }
}
