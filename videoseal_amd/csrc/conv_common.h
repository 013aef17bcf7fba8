// Shared pieces of the conv_gemm kernels: operand splitting, activation epilogue.
#pragma once
#include "vs_common.h"

namespace vsconv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 16;
constexpr int NT = 256;
constexpr int LDKF = BK + 4;   // f32 LDS row (floats)
constexpr int ROWB = 48;       // bf16 LDS row (bytes): 16 halves + 8 pad

__device__ __forceinline__ unsigned pack_hi(unsigned lo_elem, unsigned hi_elem) {
  return __builtin_amdgcn_perm(hi_elem, lo_elem, 0x07060302);   // {hi16(lo_elem), hi16(hi_elem)}
}

// 4 floats -> 3 planes of 4 bf16 (truncation split, exact)
__device__ __forceinline__ void split4(const f32x4& v, u32x2& p1, u32x2& p2, u32x2& p3) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned u = __float_as_uint(v[i]);
    h[i] = u & 0xffff0000u;
    const float r = v[i] - __uint_as_float(h[i]);
    m[i] = __float_as_uint(r) & 0xffff0000u;
    l[i] = __float_as_uint(r - __uint_as_float(m[i]));
  }
  p1 = u32x2{pack_hi(h[0], h[1]), pack_hi(h[2], h[3])};
  p2 = u32x2{pack_hi(m[0], m[1]), pack_hi(m[2], m[3])};
  p3 = u32x2{pack_hi(l[0], l[1]), pack_hi(l[2], l[3])};
}

template <int TM, int TN>
__device__ __forceinline__ void apply_act_all(f32x16 (&acc)[TM][TN], const float (&b1)[TN], const float (&b2)[TN], int act) {
  // act is wave-uniform: one branch around the whole register tile instead of a switch per element
  if (act == VS_ACT_RELU) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaxf(acc[i][j][e] + b1[j], 0.f) + b2[j];
  } else if (act == VS_ACT_GELU) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          acc[i][j][e] = vs_gelu(acc[i][j][e] + b1[j]) + b2[j];
          if ((e & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // keep the expansions from being interleaved across the whole tile (VGPRs)
        }
  } else if (act == VS_ACT_TANH) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          acc[i][j][e] = tanhf(acc[i][j][e] + b1[j]) + b2[j];
          if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = acc[i][j][e] + b1[j] + b2[j];
  }
}


// GRN statistics fused into the producing GEMM (common.py:166): per 32-row group and output column the sum of squares of the
// values this wave is about to store.  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5);
// fixed order (16 rows in a lane, then the other half-wave) -> deterministic.  part is [M/32][N].
template <int TM, int TN>
__device__ __forceinline__ void write_sumsq(const f32x16 (&acc)[TM][TN], float* part, const int N, const int64_t mrow0, const int M,
                                            const int (&col)[TN], const int g) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = mrow0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
        const float v = acc[i][j][e];
        s += m < M ? v * v : 0.f;
      }
      s += __shfl_xor(s, 32);
      if (g == 0 && col[j] < N && mrow0 + i * 32 < M) part[((mrow0 + i * 32) >> 5) * N + col[j]] = s;   // groups past M do not exist
    }
}

}  // namespace vsconv
