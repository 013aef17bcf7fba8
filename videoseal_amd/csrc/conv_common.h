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

// ---------------------------------------------------------------------------------------------------------------------------------
// Arithmetic of the split back-ends (vs_conv_desc_t::arith).  NP = number of 16-bit planes an fp32 operand is split into.
//   NP = 3  "3 x bf16": truncation split, exact (8+8+8 mantissa bits), six partial products of weight >= 2^-16.
//   NP = 2  "2 x f16" : a*a_mul = h + l with h = RN_f16(a*a_mul), l = RN_f16(a*a_mul - h)  (|a*a_mul - h - l| <= 2^-24 |a*a_mul|, the size
//           of the fp32 rounding of a itself, as long as l is a normal f16, i.e. |a*a_mul| >= 2^-2; below that the absolute error is
//           <= 2^-25: f16 denormals are honoured by the matrix cores -- measured, tools/micro/f16x2_probe.hip), three partial products
//           h_a h_b + h_a l_b + l_a h_b (dropped: l_a l_b <= 2^-24 |ab|).  Half the MFMA work of NP = 3 at the accuracy of an fp32 fmaf
//           chain (probe: rms error 1.9e-7 of |y|max at K = 3456, against 3.0e-7 for NP = 3 and 3.2e-7 for fmaf).  The f16 exponent
//           range is handled by power-of-two scales that cancel exactly: the weights are pre-scaled per layer so that max|w| lands in
//           [2^13, 2^14) (engine.split_f16x2 / model_api), the activations by the descriptor's a_mul (default 2^4: |a| < 4094), and the
//           epilogue multiplies the accumulator by acc_mul = 1 / (a_mul * w_mul) before the bias.  |a| * a_mul >= 65520 overflows to
//           inf -> NaN in the output (loud, not silent); VIDEOSEAL_CONV=bf16x3 selects the range-free NP = 3 path.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int NP> struct Arith;
template <> struct Arith<3> {
  static constexpr int NPROD = 6;
  // smallest partial products first; (plane of the first operand, plane of the second operand)
  static constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
  static __device__ __forceinline__ f32x16 mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Arith<2> {
  static constexpr int NPROD = 3;
  static constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
  // fragments travel as 16-byte bf16x8 containers in every kernel; here they hold eight f16
  static __device__ __forceinline__ f32x16 mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

// 4 floats -> 2 planes of 4 f16 (round-to-nearest split of v * amul; the residual is exact in fp32: one fma)
__device__ __forceinline__ void split4h(const f32x4& v, const float amul, u32x2& p1, u32x2& p2) {
  f16x4 hi, lo;
#pragma unroll
  for (int i = 0; i < 4; ++i) hi[i] = (_Float16)(v[i] * amul);
#pragma unroll
  for (int i = 0; i < 4; ++i) lo[i] = (_Float16)__builtin_fmaf(v[i], amul, -(float)hi[i]);
  p1 = __builtin_bit_cast(u32x2, hi);
  p2 = __builtin_bit_cast(u32x2, lo);
}

// 4 floats -> NP planes (p[0] = most significant)
template <int NP>
__device__ __forceinline__ void split4n(const f32x4& v, const float amul, u32x2 (&p)[NP]) {
  if constexpr (NP == 3) split4(v, p[0], p[1], p[2]);
  else split4h(v, amul, p[0], p[1]);
}

// Debug ablation bits of the conv / GEMM kernels (tile_hint >> 8; tools/bench_conv.py, bench_ppc.py) exist only in a build with
// EXTRA=-DVS_KERNEL_ABLATION: as run-time branches in a K loop they cut it into basic blocks at whose joins hipcc's wait-count bookkeeping
// falls back to s_waitcnt vmcnt(0) (measured on the fused ConvNeXt block: 10 %).
#ifdef VS_KERNEL_ABLATION
#define VS_KERNEL_ABL(d) ((d).tile_hint >> 8)
#else
#define VS_KERNEL_ABL(d) 0
#endif

template <int TM, int TN>
__device__ __forceinline__ void scale_all(f32x16 (&acc)[TM][TN], const float mul) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] *= mul;
}

// Store of a WHOLE register tile in the C[row][column] layout (column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) of a 32 x 32
// block; blocks i down, j across) without per-lane guards: `ob` / `rb` are wave-uniform addresses of the wave's first element in out / res
// (rb may be null), every element is that base + one 32-bit lane offset + a wave-uniform row offset + j * 128 (an instruction immediate).
// The guarded general forms keep a 64-bit address per element, and hipcc waits vmcnt(0) between a residual fetch and its store -- which
// also waits for the previous store's acknowledge: 16 * TN dependent round trips per row group; here a row group's residual values are
// fetched in one batch.
// The same store for ragged tiles: rows >= rows_left / columns >= cols_left (counted from the wave's first element) are skipped, columns in
// [cols_left, store_left) are written as zeros.  Same addressing (one 32-bit offset per lane, everything else wave-uniform): the guards are
// the only per-lane state.
template <int TM, int TN>
__device__ __forceinline__ void store_tile_guarded(const f32x16 (&acc)[TM][TN], char* const ob, const int out_ld, const char* const rb, const int res_ld,
                                                   const int r_e, const int g_e, const int rows_left, const int cols_left, const int store_left) {
  const unsigned olane = (unsigned)(4 * g_e * out_ld + r_e) * 4u;
  const unsigned rlane = (unsigned)(4 * g_e * res_ld + r_e) * 4u;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = i * 32 + (e & 3) + 8 * (e >> 2);                                      // + 4 g_e: this lane's row
      if (row + 4 * g_e >= rows_left) continue;
      const unsigned oo = (unsigned)(row * out_ld) * 4u, ro = (unsigned)(row * res_ld) * 4u;   // wave-uniform
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int c = j * 32 + r_e;
        if (c >= store_left) continue;
        float v = 0.f;
        if (c < cols_left) {
          v = acc[i][j][e];
          if (rb) v += *reinterpret_cast<const float*>(rb + (size_t)(ro + rlane) + j * 128);
        }
        *reinterpret_cast<float*>(ob + (size_t)(oo + olane) + j * 128) = v;
      }
    }
  }
}

template <int TM, int TN>
__device__ __forceinline__ void store_tile_full(const f32x16 (&acc)[TM][TN], char* const ob, const int out_ld, const char* const rb, const int res_ld,
                                                const int r_e, const int g_e) {
  const unsigned olane = (unsigned)(4 * g_e * out_ld + r_e) * 4u;
  const unsigned rlane = (unsigned)(4 * g_e * res_ld + r_e) * 4u;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    float rv[16][TN];
    if (rb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const unsigned ro = (unsigned)((i * 32 + (e & 3) + 8 * (e >> 2)) * res_ld) * 4u;       // wave-uniform
#pragma unroll
        for (int j = 0; j < TN; ++j) rv[e][j] = *reinterpret_cast<const float*>(rb + (size_t)(ro + rlane) + j * 128);
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const unsigned oo = (unsigned)((i * 32 + (e & 3) + 8 * (e >> 2)) * out_ld) * 4u;         // wave-uniform
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float v = acc[i][j][e];
        if (rb) v += rv[e][j];
        *reinterpret_cast<float*>(ob + (size_t)(oo + olane) + j * 128) = v;
      }
      if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (scheduled as one region, the values of the whole tile are formed first: spills)
    }
  }
}

// TANH = false (round 6: the TN = 3 register tiles of gemm_pl / gemm1x1_pc): the tanh variant is not compiled in -- no shipped layer takes it on those
// tiles (the network's only tanh is the 1-channel output conv), and it alone accounted for 45 of 67 (gemm_pl) / 56 of 97 (gemm1x1_pc) spilled registers
// of their epilogues; their dispatchers refuse VS_ACT_TANH instead (tile codes 18 / 24 / 26 without K slices).
template <int TM, int TN, bool TANH = true>
__device__ __forceinline__ void apply_act_all(f32x16 (&acc)[TM][TN], const float (&b1)[TN], const float (&b2)[TN], int act) {
  // act is wave-uniform: one branch around the whole register tile instead of a switch per element
  if (act == VS_ACT_RELU) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = vs_relu(acc[i][j][e] + b1[j]) + b2[j];
  } else if (act == VS_ACT_GELU) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {          // two values per VALU slot (vs_common.h::vs_gelu2: the same values as vs_gelu)
          const vs_f32x2 y = vs_gelu2(vs_f32x2{acc[i][j][e] + b1[j], acc[i][j][e + 1] + b1[j]});
          acc[i][j][e] = y[0] + b2[j];
          acc[i][j][e + 1] = y[1] + b2[j];
          if ((e & 7) == 6) __builtin_amdgcn_sched_barrier(0);   // keep the expansions from being interleaved across the whole tile (VGPRs)
        }
  } else if (TANH && act == VS_ACT_TANH) {
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

// The same statistics for frames whose row count HW is NOT a multiple of 32 (ChunkySeal: 31 x 31 = 961): a 32-row group may then straddle two
// frames (never three: HW >= 32), so every group writes TWO sums -- slot 0: its rows inside the frame f0 its first row lies in, slot 1: its rows in
// frame f0 + 1.  part is [ceil(M / 32)][2][N]; vs_grn_scale_from_straddle_partials adds, per frame, the slots that belong to it in ascending
// group order.  Round 6: replaces a separate pass over h (356 MB per ConvNeXt block at ChunkySeal's size, 6 % of its extractor pass).
template <int TM, int TN>
__device__ __forceinline__ void write_sumsq_straddle(const f32x16 (&acc)[TM][TN], float* part, const int N, const int64_t mrow0, const int M,
                                                     const int (&col)[TN], const int g, const int HW) {
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int64_t mg = mrow0 + i * 32;
    const int64_t bnd = (mg / HW + 1) * HW;          // first row of the next frame
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = mg + (e & 3) + 8 * (e >> 2) + 4 * g;
        const float v = acc[i][j][e];
        const float q = m < M ? v * v : 0.f;
        s0 += m < bnd ? q : 0.f;
        s1 += m < bnd ? 0.f : q;
      }
      s0 += __shfl_xor(s0, 32);
      s1 += __shfl_xor(s1, 32);
      if (g == 0 && col[j] < N && mg < M) {
        part[((mg >> 5) * 2) * N + col[j]] = s0;
        part[((mg >> 5) * 2 + 1) * N + col[j]] = s1;
      }
    }
  }
}

}  // namespace vsconv
