// ConvNeXt-V2 block body  pwconv1 -> GELU -> GRN -> pwconv2 (+ residual)  (modules/convnext.py:47-56, common.py:158-169) with the 4C-wide
// tensor h kept ON CHIP, for the stages where h is the traffic: stage 0 / 1 of the extractor (h = 201 / 100 MB per block at 32 frames, written by
// pwconv1 and re-read by pwconv2 in the unfused path).  2 x f16 arithmetic (three products per fp32-accurate product), same operand scaling as
// the rest of the library.
//
// GRN needs ||h||^2 over the whole frame before h can be used, so the block runs as TWO launches of one kernel template:
//   STATS: pwconv1 + GELU, only the per-(32-pixel group, channel) sums of squares leave the chip (vs_grn_scale_from_partials finishes them);
//   APPLY: pwconv1 + GELU again (bit-identical values), h * scale[frame] + beta, split, pwconv2, + bias + residual -> out.
// pwconv1 is computed twice (1.5 x the block's MFMA work) and h never touches HBM: per block 2 x (tn planes) + residual + out instead of
// planes + 2 x h + residual + out.
//
// The register-level fusion: pwconv1 is issued TRANSPOSED, D1[h-channel][pixel] = W1 . tn^T.  In the 32x32 MFMA accumulator layout a lane then
// holds, for ITS pixel (lane & 31), the h-channels 8q + 4 (lane >> 5) + e (q = reg / 4, e = reg % 4) -- and the A operand of the second MFMA wants,
// for row = pixel (lane & 31), eight k values per half-wave.  Registers 8s .. 8s + 7 of D1 ARE such an operand for k-step s if the reduction index
// of pwconv2 is enumerated as k(s, half, v) = 8 (2s + v / 4) + 4 half + v % 4: a permutation of the 32 h-channels of the block that is folded into
// the packed image of W2 on the host.  No LDS transpose of h, no shuffles: GELU, GRN apply and the f16 split run on the accumulator registers.
//
// Workgroup = 4 waves (one per SIMD: the kernel wants the whole register file -- tn fragments for all of K, two accumulator sets), each wave owns
// 32 * PB pixels for the whole kernel; the h-channels are walked in blocks of 32.  Per block the packed weights (W1 rows, permuted W2 columns,
// bias / beta) are one contiguous image that the four waves DMA into a double-buffered LDS stage (global_load_lds, 1 KiB per instruction).
#include <type_traits>

#include "conv_common.h"
#include "lds_dma.h"

namespace {

using namespace vsconv;

// LDS-DMA forms (lds_dma.h): the builtin for the serial kernel, the untracked M0-neutral asm form for the pipelined one (hipcc would put
// s_waitcnt vmcnt(0) in front of the next LDS read of the same array -- in a loop that requests block hb + 3 and then reads block hb + 1 that
// is a wait for the blocks just requested: 26 of 94 us at C = 192; the waits of that kernel are counted by hand)
__device__ __forceinline__ void dma16(const char* gp, unsigned char* lds_base) { vs_lds_dma16(gp, lds_base); }
__device__ __forceinline__ void dma16_asm(const char* gp, unsigned char* lds_base) { vs_lds_dma16_untracked(gp, lds_base); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
// vs_gelu on two values at once: the polynomial and the blends as 2-wide fp32 operations (v_pk_fma_f32 / v_pk_mul_f32: two results per VALU slot),
// only the exp2 stays scalar.  Same coefficients and operation order as vs_gelu / vs_erfc_sqrt2 -> the same values.
// (every multiply-add below is an explicit fma and contraction is off: left to the compiler, the serial and the pipelined kernel contracted
// `m - t * e` and `h * h + s` differently and disagreed in the last bit)
__device__ __forceinline__ f32x2 fma2(const f32x2 a, const f32x2 b, const f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#pragma clang fp contract(off)
__device__ __forceinline__ f32x2 gelu2(const f32x2 v) {
  const f32x2 a = {fabsf(v[0]), fabsf(v[1])};
  const f32x2 u = {fminf(a[0], 5.65685424949238f), fminf(a[1], 5.65685424949238f)};
  f32x2 q = {5.128553084e-07f, 5.128553084e-07f};
  q = fma2(q, u, f32x2{-9.560153558e-06f, -9.560153558e-06f});
  q = fma2(q, u, f32x2{7.497344632e-05f, 7.497344632e-05f});
  q = fma2(q, u, f32x2{-2.843466646e-04f, -2.843466646e-04f});
  q = fma2(q, u, f32x2{1.498938855e-05f, 1.498938855e-05f});
  q = fma2(q, u, f32x2{6.931120995e-03f, 6.931120995e-03f});
  q = fma2(q, u, f32x2{-5.243476480e-02f, -5.243476480e-02f});
  q = fma2(q, u, f32x2{-4.592214525e-01f, -4.592214525e-01f});
  q = fma2(q, u, f32x2{-1.151104212e+00f, -1.151104212e+00f});
  const f32x2 t = q * u;
  const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  const f32x2 m = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)};
  return fma2(a * -0.5f, e, m);
}
// one pair of pwconv1 outputs -> GELU -> GRN apply -> f16 split (APPLY) / GELU -> square sum (STATS): shared by both kernels
struct CnxSplit2 { _Float16 h0, h1, l0, l1; };
__device__ __forceinline__ CnxSplit2 cnx_pair_apply(const f32x2 acc, const float mul, const f32x2 b1, const f32x2 sc, const f32x2 be, const bool no_gelu) {
  const f32x2 x1 = fma2(acc, f32x2{mul, mul}, b1);
  const f32x2 h = no_gelu ? x1 : gelu2(x1);
  const f32x2 h3 = fma2(h, sc, be);                  // the unfused A transform (a_mul = 1)
  CnxSplit2 r;
  r.h0 = (_Float16)h3[0];
  r.h1 = (_Float16)h3[1];
  r.l0 = (_Float16)(h3[0] - (float)r.h0);
  r.l1 = (_Float16)(h3[1] - (float)r.h1);
  return r;
}
__device__ __forceinline__ f32x2 cnx_pair_stats(const f32x2 acc, const float mul, const float b1c, const f32x2 sq2) {
  const f32x2 h = gelu2(fma2(acc, f32x2{mul, mul}, f32x2{b1c, b1c}));
  return fma2(h, h, sq2);
}
#pragma clang fp contract(fast)

struct CnxArgs {
  const char* tn_pl;          // [2][C/16][rows][16] f16: LayerNorm output * a_mul1 (vs_dwconv7_ln_planes)
  const char* wimg;           // per 32-channel block of h: [W1: KS1 x 2 planes x 1 KiB][W2: NB x 2 steps x 2 planes x 1 KiB][aux 1 KiB: b1[32], beta[32]]
  const float* scale;         // [B][scale_ld]: 1 + gamma * Nx (APPLY)
  const float* bias2;         // [C]
  const float* res;           // [rows][res_ld] (APPLY)
  float* out;                 // [rows][out_ld] (APPLY)
  float* part32;              // [rows / 32][4C] (STATS)
  int64_t rows, scale_ld, res_ld, out_ld;
  int HW;
  float acc_mul1, acc_mul2;   // 1 / (a_mul1 * w_mul1), 1 / (1 * w_mul2)
  int abl;                    // ablation bits (tools/bench_cnx.py, VS_CNX_ABL): 1 no GELU, 2 no pwconv2 MFMAs, 4 no pwconv1 MFMAs, 8 no output stores, 16 no weight DMA in the loop (pipelined kernel)
};

// The ablation bits are a compile-time switch (make EXTRA=-DVS_CNX_ABLATION, for tools/bench_cnx.py): as run-time branches around the MFMA
// groups they cut the K loop into ~100 basic blocks, and at every join hipcc's wait-count bookkeeping fell back to s_waitcnt vmcnt(0) in
// front of the first LDS read -- i.e. each iteration waited for the weight block it had just requested two iterations ahead.
#ifdef VS_CNX_ABLATION
#define CNX_ABL(a) ((a).abl)
#else
#define CNX_ABL(a) 0
#endif

template <int KS1, int PB, bool STATS>
__global__ __launch_bounds__(256) void cnx_block_kernel(const CnxArgs a) {
  using AR = Arith<2>;
  constexpr int C = KS1 * 16, NB = C / 32, NHB = 4 * C / 32;
  constexpr int W1_KB = KS1 * 2, W2_KB = NB * 4;
  constexpr int BLK_KB = W1_KB + W2_KB + 1;                 // KiB per h-block image
  constexpr int NST = 3;                                    // weight stages: block hb + 2 is in flight while block hb is multiplied (an L2 round trip is
                                                            // about as long as one block's arithmetic: with two stages every iteration waited for its DMA)
  __shared__ __attribute__((aligned(16))) unsigned char smem[NST * BLK_KB * 1024];
  __shared__ __attribute__((aligned(16))) float s_scale[STATS ? 4 : 4 * C];      // GRN scale of the workgroup's frame (its 128 * PB pixels lie in one frame)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31, hf = lane >> 5;
  const int64_t p0 = ((int64_t)blockIdx.x * 4 + wave) * (32 * PB);       // first pixel of this wave
  const int frame = (int)(p0 / a.HW);

  // ---- weight image DMA: chunk k (1 KiB) of block hb -> stage; wave w moves chunks w, w + 4, ...
  auto dma_block = [&](const int hb, const int stage) __attribute__((always_inline)) {
    const char* src = a.wimg + (int64_t)hb * (BLK_KB * 1024) + lane * 16;
    unsigned char* dst = smem + stage * (BLK_KB * 1024);
#pragma unroll
    for (int k = 0; k < BLK_KB; ++k) {
      if ((k & 3) != (wave & 3)) continue;
      if (STATS && k >= W1_KB && k < W1_KB + W2_KB) continue;          // STATS never reads W2
      dma16(src + k * 1024, dst + k * 1024);
    }
  };

  // ---- this wave's pixels of tn: fragments for all of K stay in registers
  bf16x8 tnf[PB][KS1][2];
  {
    const int64_t pstride = (int64_t)KS1 * a.rows * 32;
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          tnf[pb][ks][pl] = *reinterpret_cast<const bf16x8*>(a.tn_pl + pl * pstride + ((int64_t)ks * a.rows + p0 + pb * 32 + r32) * 32 + hf * 16);
  }
  f32x16 acc2[PB][NB];
  if constexpr (!STATS) {
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[pb][nb][e] = 0.f;
  }
  if constexpr (!STATS) {
    for (int i = threadIdx.x; i < 4 * C; i += 256) s_scale[i] = a.scale[(int64_t)frame * a.scale_ld + i];
  }
  dma_block(0, 0);
  dma_block(1, 1);
  // DMA instructions this wave issues per block (chunks wave, wave + 4, ...): the counted wait below leaves exactly one block in flight
  constexpr int NCH = STATS ? W1_KB + 1 : BLK_KB;           // chunks a launch moves per block (W1 is a multiple of 4 chunks, then W2, then aux)
  const int my_dmas = STATS ? (W1_KB / 4 + ((W1_KB + W2_KB) % 4 == wave ? 1 : 0)) : (BLK_KB - wave + 3) / 4;
  auto wait_older = [&]() __attribute__((always_inline)) {  // everything but the youngest block's DMAs has landed
    if (my_dmas == NCH / 4 + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCH / 4 + 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCH / 4) : "memory");
  };
  wait_older();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int stage = 0;
  for (int hb = 0; hb < NHB; ++hb) {
    const int st2 = stage + 2 >= NST ? stage + 2 - NST : stage + 2;
    if (hb + 2 < NHB) dma_block(hb + 2, st2);                // stage st2 held block hb - 1: last read before the barrier that ended iteration hb - 1
    const unsigned char* Wb = smem + stage * (BLK_KB * 1024);
    stage = stage + 1 == NST ? 0 : stage + 1;
    // GRN scale of this block's 32 channels for the wave's frame: channels 8q + 4 hf + (0..3)
    f32x4 sc[4];
    if constexpr (!STATS) {
#pragma unroll
      for (int q = 0; q < 4; ++q) sc[q] = *reinterpret_cast<const f32x4*>(s_scale + hb * 32 + 8 * q + 4 * hf);
    }
    // ---- pwconv1, transposed: D1[h-channel][pixel].  Two accumulators per pixel block (even / odd k-steps, summed afterwards): with one, the
    // 3 * KS1 MFMAs of a block form a single dependent chain and the matrix pipe idles between them (measured: 39 of 183 us at C = 96)
    f32x16 acc1[PB], accb[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc1[pb][e] = 0.f; accb[pb][e] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      const bf16x8 whi = *reinterpret_cast<const bf16x8*>(Wb + (ks * 2 + 0) * 1024 + lane * 16);
      const bf16x8 wlo = *reinterpret_cast<const bf16x8*>(Wb + (ks * 2 + 1) * 1024 + lane * 16);
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        if (CNX_ABL(a) & 4) continue;
        f32x16& acc = (ks & 1) ? accb[pb] : acc1[pb];
        if constexpr (STATS) {      // D1[pixel][h-channel]: the same fragments with the operands swapped -- a lane then holds ONE channel for 16 pixels
          acc = AR::mfma(tnf[pb][ks][0], wlo, acc);      // (identical products and order: the values equal the APPLY pass's bit for bit)
          acc = AR::mfma(tnf[pb][ks][1], whi, acc);
          acc = AR::mfma(tnf[pb][ks][0], whi, acc);
        } else {
          acc = AR::mfma(wlo, tnf[pb][ks][0], acc);      // smallest partial products first (Arith<2>)
          acc = AR::mfma(whi, tnf[pb][ks][1], acc);
          acc = AR::mfma(whi, tnf[pb][ks][0], acc);
        }
      }
    }
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) acc1[pb] += accb[pb];
    const float* aux = reinterpret_cast<const float*>(Wb + (W1_KB + W2_KB) * 1024);
    f32x4 b1[4], be[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      b1[q] = *reinterpret_cast<const f32x4*>(aux + 8 * q + 4 * hf);
      be[q] = *reinterpret_cast<const f32x4*>(aux + 32 + 8 * q + 4 * hf);
    }
    if constexpr (STATS) {
      const float b1c = aux[r32];                               // this lane's channel hb * 32 + r32
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        f32x2 sq2 = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          sq2 = cnx_pair_stats(f32x2{acc1[pb][e], acc1[pb][e + 1]}, a.acc_mul1, b1c, sq2);
        }
        float sq = sq2[0] + sq2[1];
        sq += __shfl_xor(sq, 32);                               // the other half-wave holds the other 16 pixels of the group
        if (hf == 0) a.part32[((p0 >> 5) + pb) * (int64_t)(4 * C) + hb * 32 + r32] = sq;
      }
    } else {
      // ---- GELU, GRN apply, f16 split on the accumulator registers: registers 8s .. 8s + 7 are the A operand of k-step s of pwconv2
      bf16x8 ahi[PB][2], alo[PB][2];
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          f16x8 hi, lo;
#pragma unroll
          for (int v = 0; v < 8; v += 2) {
            const int e = 8 * s + v;
            const CnxSplit2 sp = cnx_pair_apply(f32x2{acc1[pb][e], acc1[pb][e + 1]}, a.acc_mul1, f32x2{b1[e >> 2][e & 3], b1[e >> 2][(e & 3) + 1]},
                                                f32x2{sc[e >> 2][e & 3], sc[e >> 2][(e & 3) + 1]}, f32x2{be[e >> 2][e & 3], be[e >> 2][(e & 3) + 1]},
                                                (CNX_ABL(a) & 1) != 0);
            hi[v] = sp.h0; hi[v + 1] = sp.h1; lo[v] = sp.l0; lo[v + 1] = sp.l1;
          }
          ahi[pb][s] = __builtin_bit_cast(bf16x8, hi);
          alo[pb][s] = __builtin_bit_cast(bf16x8, lo);
        }
      // ---- pwconv2 over this block's 32 (permuted) h-channels
      const unsigned char* W2b = Wb + W1_KB * 1024;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 whi = *reinterpret_cast<const bf16x8*>(W2b + ((nb * 2 + s) * 2 + 0) * 1024 + lane * 16);
          const bf16x8 wlo = *reinterpret_cast<const bf16x8*>(W2b + ((nb * 2 + s) * 2 + 1) * 1024 + lane * 16);
#pragma unroll
          for (int pb = 0; pb < PB; ++pb) {
            if (CNX_ABL(a) & 2) continue;
            acc2[pb][nb] = AR::mfma(alo[pb][s], whi, acc2[pb][nb]);
            acc2[pb][nb] = AR::mfma(ahi[pb][s], wlo, acc2[pb][nb]);
            acc2[pb][nb] = AR::mfma(ahi[pb][s], whi, acc2[pb][nb]);
          }
        }
    }
    if (hb + 2 < NHB) wait_older();                           // block hb + 1 has landed (block hb + 2 may still be in flight)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);                       // this wave's LDS reads have returned
    __builtin_amdgcn_s_barrier();
  }
  if constexpr (!STATS) {
    if (CNX_ABL(a) & 8) return;
    // D2[pixel][channel]: lane holds channel nb * 32 + r32 of pixels 8 (e >> 2) + 4 hf + (e & 3) of its block.  Written through LDS (the weight
    // stages are free now) so that the block's 32 pixels x C channels leave as whole rows: with out_ld == C that is 32 * C * 4 contiguous bytes,
    // 16 bytes per lane -- the direct form (one 128-byte piece per store instruction, 16 * NB * PB instructions) cost 53 of 183 us.
    float* T = reinterpret_cast<float*>(smem) + wave * (32 * (C + 4));          // [32 pixels][C + 4] floats per wave (pad: bank spread)
    static_assert(4 * 32 * (C + 4) * 4 <= NST * BLK_KB * 1024, "transpose tile must fit the weight stages");
    constexpr int C4 = C / 4, ITS = (32 * C4) / 64;
    float b2v[NB];                               // (before the residual rows: vmcnt retires in order, the transpose below needs only these)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) b2v[nb] = a.bias2[nb * 32 + r32];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int64_t pbase = p0 + pb * 32;
      // the residual rows first, all in flight at once (tn's registers are dead): fetched inside the store loop each one was a dependent
      // round trip -- load, s_waitcnt vmcnt(0) (which also waits for the previous store's acknowledge), store -- 12 to 24 of them in a row
      f32x4 rr[ITS];
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        const int idx = it * 64 + lane;
        const int px = idx / C4, c4 = idx - px * C4;
        rr[it] = *reinterpret_cast<const f32x4*>(a.res + (pbase + px) * a.res_ld + 4 * c4);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float b2 = b2v[nb];
#pragma unroll
        for (int e = 0; e < 16; ++e) T[(8 * (e >> 2) + 4 * hf + (e & 3)) * (C + 4) + nb * 32 + r32] = acc2[pb][nb][e] * a.acc_mul2 + b2;
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        const int idx = it * 64 + lane;
        const int px = idx / C4, c4 = idx - px * C4;
        const f32x4 t = *reinterpret_cast<const f32x4*>(T + px * (C + 4) + 4 * c4);
        *reinterpret_cast<f32x4*>(a.out + (pbase + px) * a.out_ld + 4 * c4) = t + rr[it];
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---- the same block as a SOFTWARE PIPELINE over the h-blocks: pwconv1 of block hb + 1 is issued while the GELU / GRN / split arithmetic of
// block hb runs.  In cnx_block_kernel an iteration is three serial phases of one wave -- 3 KS1 MFMAs (each group behind its own LDS read),
// ~300 VALU instructions with the matrix pipe idle, 6 NB MFMAs -- and at C = 192 there is one wave per SIMD, so nothing covers them: 36 %
// MFMA utilisation inside the loop.  Here the VALU work of block hb is cut into pairs of values and dealt out between the MFMAs of
// pwconv1(hb + 1) (independent registers: the next block's accumulators), the weight fragments are read one k-step ahead, and the rings
// are per operand, three stages each (in use / landed / in flight): W1 holds blocks hb + 1 .. hb + 3, W2 + bias / beta blocks hb .. hb + 2
// (with two stages -- 98 KiB at C = 192 -- a block had to land within the iteration that requested it: 26 of 94 us were that wait).
// Same products in the same order: the results equal cnx_block_kernel's bit for bit.
template <int KS1, int PB, bool STATS>
__global__ __launch_bounds__(256) void cnx_pipe_kernel(const CnxArgs a) {
  using AR = Arith<2>;
  constexpr int C = KS1 * 16, NB = C / 32, NHB = 4 * C / 32;
  constexpr int W1_KB = KS1 * 2, W2_KB = NB * 4;
  constexpr int BLK_KB = W1_KB + W2_KB + 1;                 // KiB per h-block image (the packed image of cnx_block_kernel)
  constexpr int S2_KB = STATS ? 1 : W2_KB + 1;              // second ring: (W2 +) aux
  constexpr int NST = 3;                                     // stages per ring: in use, landed, in flight
  __shared__ __attribute__((aligned(16))) unsigned char smem[NST * (W1_KB + S2_KB) * 1024];
  __shared__ __attribute__((aligned(16))) float s_scale[STATS ? 4 : 4 * C];
  unsigned char* const R1 = smem;
  unsigned char* const R2 = smem + NST * W1_KB * 1024;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31, hf = lane >> 5;
  const int64_t p0 = ((int64_t)blockIdx.x * 4 + wave) * (32 * PB);
  const int frame = (int)(p0 / a.HW);

  // chunk k (1 KiB) of a block image goes through wave k & 3: wave w moves chunks w, w + 4, ... -- addressed directly (the `if ((k & 3) == wave)`
  // ladder of cnx_block_kernel is ~50 scalar compare-and-branch pairs per block and wave, and with one wave per SIMD nothing runs beside them)
  auto dma_w1 = [&](const int hb, const int stage) __attribute__((always_inline)) {
    const char* src = a.wimg + ((int64_t)hb * BLK_KB + wave) * 1024 + lane * 16;
    unsigned char* dst = R1 + (stage * W1_KB + wave) * 1024;
#pragma unroll
    for (int j = 0; j < W1_KB / 4; ++j) dma16_asm(src + j * 4096, dst + j * 4096);
  };
  auto dma_w2 = [&](const int hb, const int stage) __attribute__((always_inline)) {
    const char* src = a.wimg + ((int64_t)hb * BLK_KB + W1_KB + (STATS ? W2_KB : 0) + wave) * 1024 + lane * 16;
    unsigned char* dst = R2 + (stage * S2_KB + wave) * 1024;
#pragma unroll
    for (int j = 0; j < S2_KB / 4; ++j) dma16_asm(src + j * 4096, dst + j * 4096);
    if ((S2_KB & 3) > wave) dma16_asm(src + (S2_KB / 4) * 4096, dst + (S2_KB / 4) * 4096);
  };
  auto wait_all = [&]() __attribute__((always_inline)) {     // this wave's DMAs have landed, its LDS reads have returned; then everybody's
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
  };
  // the same, but the DMAs of the CURRENT iteration (one W1 block, one W2 / aux block) may stay in flight: a wave issues W1_KB / 4 of the
  // former and S2_KB / 4 (+ 1 for the waves that take the remainder chunks) of the latter
  constexpr int N_IT = W1_KB / 4 + S2_KB / 4 + (STATS ? PB : 0);      // (+ the statistics stores of the iteration: stores count in vmcnt too)
  const bool one_more = (S2_KB & 3) > (wave & 3);
  auto wait_older = [&]() __attribute__((always_inline)) {
    if (one_more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_IT + 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_IT) : "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
  };

  bf16x8 tnf[PB][KS1][2];
  {
    const int64_t pstride = (int64_t)KS1 * a.rows * 32;
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          tnf[pb][ks][pl] = *reinterpret_cast<const bf16x8*>(a.tn_pl + pl * pstride + ((int64_t)ks * a.rows + p0 + pb * 32 + r32) * 32 + hf * 16);
  }
  f32x16 acc2[PB][NB];
  if constexpr (!STATS) {
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[pb][nb][e] = 0.f;
    for (int i = threadIdx.x; i < 4 * C; i += 256) s_scale[i] = a.scale[(int64_t)frame * a.scale_ld + i];
  }
  dma_w1(0, 0);
  dma_w2(0, 0);
  dma_w1(1, 1);
  dma_w2(1, 1);
  dma_w1(2, 2);
  wait_all();

  // one k-step of pwconv1 (products and order of cnx_block_kernel: two accumulators, even / odd k-steps)
  auto pw1_step = [&](const int ks, const bf16x8 whi, const bf16x8 wlo, f32x16 (&A0)[PB], f32x16 (&A1)[PB]) __attribute__((always_inline)) {
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      f32x16& acc = (ks & 1) ? A1[pb] : A0[pb];
      if constexpr (STATS) {
        acc = AR::mfma(tnf[pb][ks][0], wlo, acc);
        acc = AR::mfma(tnf[pb][ks][1], whi, acc);
        acc = AR::mfma(tnf[pb][ks][0], whi, acc);
      } else {
        acc = AR::mfma(wlo, tnf[pb][ks][0], acc);
        acc = AR::mfma(whi, tnf[pb][ks][1], acc);
        acc = AR::mfma(whi, tnf[pb][ks][0], acc);
      }
    }
  };
  // pwconv1 of one block with `between(ks)` issued behind the MFMAs of every k-step; the weight fragments are read one k-step ahead
  auto pw1_block = [&](const unsigned char* W1s, f32x16 (&A0)[PB], f32x16 (&A1)[PB], auto between) __attribute__((always_inline)) {
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int e = 0; e < 16; ++e) { A0[pb][e] = 0.f; A1[pb][e] = 0.f; }
    const unsigned char* W = W1s + lane * 16;
    bf16x8 whi = *reinterpret_cast<const bf16x8*>(W), wlo = *reinterpret_cast<const bf16x8*>(W + 1024);
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      bf16x8 nhi = whi, nlo = wlo;
      if (ks + 1 < KS1) {
        nhi = *reinterpret_cast<const bf16x8*>(W + (2 * ks + 2) * 1024);
        nlo = *reinterpret_cast<const bf16x8*>(W + (2 * ks + 3) * 1024);
      }
      if (!(CNX_ABL(a) & 4)) pw1_step(ks, whi, wlo, A0, A1);
      between(std::integral_constant<int, 0>{}, ks);
      whi = nhi;
      wlo = nlo;
      __builtin_amdgcn_sched_barrier(0);        // keep the deal: MFMAs of k-step ks, then its share of the VALU work
    }
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) A0[pb] += A1[pb];
  };

  f32x16 acc1[PB], accb[PB];
  pw1_block(R1, acc1, accb, [](auto, int) {});
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();                  // stage 0 of the W1 ring is refilled by iteration 0

  // ---- the arithmetic on block hb's accumulators, one pair of values at a time
  f32x4 sc[4], b1[4], be[4];
  float b1c = 0.f;
  f32x2 sq2[PB];
  f16x8 hi[PB][2], lo[PB][2];
  auto block_consts = [&](const int hb, const int st2) __attribute__((always_inline)) {
    const float* aux = reinterpret_cast<const float*>(R2 + st2 * (S2_KB * 1024) + (STATS ? 0 : W2_KB * 1024));
    if constexpr (STATS) {
      b1c = aux[r32];
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) sq2[pb] = f32x2{0.f, 0.f};
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        sc[q] = *reinterpret_cast<const f32x4*>(s_scale + hb * 32 + 8 * q + 4 * hf);
        b1[q] = *reinterpret_cast<const f32x4*>(aux + 8 * q + 4 * hf);
        be[q] = *reinterpret_cast<const f32x4*>(aux + 32 + 8 * q + 4 * hf);
      }
    }
  };
  auto do_pair = [&](const int t) __attribute__((always_inline)) {          // t = pb * 8 + pair
    const int pb = t >> 3, e = 2 * (t & 7);
    if constexpr (STATS) {
      sq2[pb] = cnx_pair_stats(f32x2{acc1[pb][e], acc1[pb][e + 1]}, a.acc_mul1, b1c, sq2[pb]);
    } else {
      const int s = e >> 3, v = e & 7;
      const CnxSplit2 sp = cnx_pair_apply(f32x2{acc1[pb][e], acc1[pb][e + 1]}, a.acc_mul1, f32x2{b1[e >> 2][e & 3], b1[e >> 2][(e & 3) + 1]},
                                          f32x2{sc[e >> 2][e & 3], sc[e >> 2][(e & 3) + 1]}, f32x2{be[e >> 2][e & 3], be[e >> 2][(e & 3) + 1]},
                                          (CNX_ABL(a) & 1) != 0);
      hi[pb][s][v] = sp.h0; hi[pb][s][v + 1] = sp.h1; lo[pb][s][v] = sp.l0; lo[pb][s][v + 1] = sp.l1;
    }
  };
  auto pairs_of = [&](auto, const int ks) __attribute__((always_inline)) {   // the pairs dealt to k-step ks of the concurrent pwconv1
#pragma unroll
    for (int t = 0; t < 8 * PB; ++t)
      if (t * KS1 / (8 * PB) == ks) do_pair(t);
  };
  auto finish_block = [&](const int hb, const int st2) __attribute__((always_inline)) {     // what follows the pair arithmetic of block hb
    if constexpr (STATS) {
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        float sq = sq2[pb][0] + sq2[pb][1];
        sq += __shfl_xor(sq, 32);
        if (hf == 0) a.part32[((p0 >> 5) + pb) * (int64_t)(4 * C) + hb * 32 + r32] = sq;
      }
    } else {
      const unsigned char* W2b = R2 + st2 * (S2_KB * 1024) + lane * 16;
      bf16x8 whi = *reinterpret_cast<const bf16x8*>(W2b), wlo = *reinterpret_cast<const bf16x8*>(W2b + 1024);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 nhi = whi, nlo = wlo;
          if (nb * 2 + s + 1 < NB * 2) {
            nhi = *reinterpret_cast<const bf16x8*>(W2b + ((nb * 2 + s + 1) * 2 + 0) * 1024);
            nlo = *reinterpret_cast<const bf16x8*>(W2b + ((nb * 2 + s + 1) * 2 + 1) * 1024);
          }
#pragma unroll
          for (int pb = 0; pb < PB; ++pb) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, hi[pb][s]), al = __builtin_bit_cast(bf16x8, lo[pb][s]);
            if (CNX_ABL(a) & 2) continue;
            acc2[pb][nb] = AR::mfma(al, whi, acc2[pb][nb]);
            acc2[pb][nb] = AR::mfma(ah, wlo, acc2[pb][nb]);
            acc2[pb][nb] = AR::mfma(ah, whi, acc2[pb][nb]);
          }
          whi = nhi;
          wlo = nlo;
        }
    }
  };

  // iteration hb: W2 / aux of block hb in stage s2 (hb % 3), W1 of block hb + 1 in stage s1n ((hb + 1) % 3); issued here: W1 of block hb + 3
  // into the stage block hb's W1 left (read during iteration hb - 1), W2 of block hb + 2 into the stage block hb - 1's W2 left
  int s2 = 0, s1n = 1;
  for (int hb = 0; hb + 1 < NHB; ++hb) {
    const int s1i = s1n == 0 ? NST - 1 : s1n - 1;            // hb % 3
    const int s2i = s2 == 0 ? NST - 1 : s2 - 1;              // (hb + 2) % 3
    const bool both = hb + 3 < NHB;
    if (!(CNX_ABL(a) & 16)) {
      if (both) dma_w1(hb + 3, s1i);
      if (hb + 2 < NHB) dma_w2(hb + 2, s2i);
    }
    block_consts(hb, s2);
    f32x16 acc1n[PB], accbn[PB];
    pw1_block(R1 + s1n * (W1_KB * 1024), acc1n, accbn, pairs_of);
    finish_block(hb, s2);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) acc1[pb] = acc1n[pb];
    if (both) wait_older();
    else wait_all();
    s2 = s2 + 1 == NST ? 0 : s2 + 1;
    s1n = s1n + 1 == NST ? 0 : s1n + 1;
  }
  block_consts(NHB - 1, s2);
#pragma unroll
  for (int t = 0; t < 8 * PB; ++t) do_pair(t);
  finish_block(NHB - 1, s2);
  if constexpr (!STATS) {
    if (CNX_ABL(a) & 8) return;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                // every wave is done with the rings: they become the transpose tiles
    // D2[pixel][channel] -> whole rows through LDS, residual rows fetched in one batch (as cnx_block_kernel)
    float* T = reinterpret_cast<float*>(smem) + wave * (32 * (C + 4));
    static_assert(4 * 32 * (C + 4) * 4 <= NST * (W1_KB + S2_KB) * 1024, "transpose tile must fit the rings");
    constexpr int C4 = C / 4, ITS = (32 * C4) / 64;
    float b2v[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) b2v[nb] = a.bias2[nb * 32 + r32];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int64_t pbase = p0 + pb * 32;
      f32x4 rr[ITS];
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        const int idx = it * 64 + lane;
        const int px = idx / C4, c4 = idx - px * C4;
        rr[it] = *reinterpret_cast<const f32x4*>(a.res + (pbase + px) * a.res_ld + 4 * c4);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float b2 = b2v[nb];
#pragma unroll
        for (int e = 0; e < 16; ++e) T[(8 * (e >> 2) + 4 * hf + (e & 3)) * (C + 4) + nb * 32 + r32] = acc2[pb][nb][e] * a.acc_mul2 + b2;
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        const int idx = it * 64 + lane;
        const int px = idx / C4, c4 = idx - px * C4;
        const f32x4 t = *reinterpret_cast<const f32x4*>(T + px * (C + 4) + 4 * c4);
        *reinterpret_cast<f32x4*>(a.out + (pbase + px) * a.out_ld + 4 * c4) = t + rr[it];
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <int KS1, int PB>
int launch_cnx(const CnxArgs& a, bool stats, bool serial, hipStream_t st) {
  const unsigned grid = (unsigned)(a.rows / (128 * PB));
  static const bool pipe = [] { const char* e = getenv("VS_CNX_PIPE"); return e ? atoi(e) != 0 : true; }();   // (A/B handle: 0 = cnx_block_kernel)
  if (pipe && !serial) {
    if (stats) hipLaunchKernelGGL((cnx_pipe_kernel<KS1, PB, true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((cnx_pipe_kernel<KS1, PB, false>), dim3(grid), dim3(256), 0, st, a);
    return vs_launch_status();
  }
  if (stats) hipLaunchKernelGGL((cnx_block_kernel<KS1, PB, true>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((cnx_block_kernel<KS1, PB, false>), dim3(grid), dim3(256), 0, st, a);
  return vs_launch_status();
}

}  // namespace

extern "C" int vs_cnx_block_supported(int C, int64_t rows, int HW) {
  if (C != 96 && C != 192) return 0;
  const int pb = C == 96 ? 2 : 1;
  return (rows % (128 * pb) == 0 && HW % (32 * pb) == 0) ? 1 : 0;
}

extern "C" int64_t vs_cnx_block_image_bytes(int C) { return (int64_t)(4 * C / 32) * ((C / 16) * 2 + (C / 32) * 4 + 1) * 1024; }

// stats & 1: part32 [rows / 32][4C] <- per-group sums of gelu(pwconv1)^2;  else: out = res + pwconv2(grn(gelu(pwconv1))) (out may alias res).
// stats & 2: the serial kernel (cnx_block_kernel) instead of the software pipeline -- same bits, kept for the comparison in tests / tools.
extern "C" int vs_cnx_block(const void* tn_planes, const void* wimg, int C, int64_t rows, int HW, int stats, float acc_mul1, float acc_mul2,
                            const float* scale, int64_t scale_ld, const float* bias2, const float* res, int64_t res_ld, float* out, int64_t out_ld,
                            float* part32, void* stream) {
  VS_REQUIRE(tn_planes && wimg && rows > 0 && HW > 0 && vs_cnx_block_supported(C, rows, HW));
  VS_REQUIRE((stats & 1) ? part32 != nullptr : (scale && bias2 && res && out && scale_ld >= 4 * C && (scale_ld & 3) == 0 && res_ld >= C && out_ld >= C && (res_ld & 3) == 0 && (out_ld & 3) == 0 && (((uintptr_t)res | (uintptr_t)out) & 15) == 0));
  VS_REQUIRE((((uintptr_t)tn_planes) & 15) == 0 && (((uintptr_t)wimg) & 15) == 0 && ((stats & 1) || (((uintptr_t)scale) & 15) == 0));
  CnxArgs a{static_cast<const char*>(tn_planes), static_cast<const char*>(wimg), scale, bias2, res, out, part32, rows, scale_ld, res_ld, out_ld, HW,
            acc_mul1, acc_mul2, 0};
#ifdef VS_CNX_ABLATION
  { const char* e = getenv("VS_CNX_ABL"); a.abl = e ? atoi(e) : 0; }        // (read per call: tools/bench_cnx.py walks the bits in one process)
#endif
  // C = 96: the statistics launch takes 64 pixels per wave (weight fragments shared by two pixel blocks, one wave per SIMD: 52 vs 56 us at 32
  // frames), the apply launch 32 (139 + 80 registers: TWO workgroups per CU, whose MFMAs cover each other's GELU / split arithmetic: 103 vs 130 us)
  const bool st1 = (stats & 1) != 0, serial = (stats & 2) != 0;
  if (C == 96) return st1 ? launch_cnx<6, 2>(a, true, serial, (hipStream_t)stream) : launch_cnx<6, 1>(a, false, serial, (hipStream_t)stream);
  return launch_cnx<12, 1>(a, st1, serial, (hipStream_t)stream);
}
