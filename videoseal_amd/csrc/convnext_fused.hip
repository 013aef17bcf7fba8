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
#include "conv_common.h"

namespace {

using namespace vsconv;

__device__ __forceinline__ void dma16(const char* gp, unsigned char* lds_base) {      // lane l: 16 bytes at gp -> lds_base + 16 * l
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, (__attribute__((address_space(3))) void*)lds_base,
                                   16, 0, 0);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// vs_gelu on two values at once: the polynomial and the blends as 2-wide fp32 operations (v_pk_fma_f32 / v_pk_mul_f32: two results per VALU slot),
// only the exp2 stays scalar.  Same coefficients and operation order as vs_gelu / vs_erfc_sqrt2 -> the same values.
__device__ __forceinline__ f32x2 gelu2(const f32x2 v) {
  const f32x2 a = {fabsf(v[0]), fabsf(v[1])};
  const f32x2 u = {fminf(a[0], 5.65685424949238f), fminf(a[1], 5.65685424949238f)};
  f32x2 q = {5.128553084e-07f, 5.128553084e-07f};
  q = q * u + f32x2{-9.560153558e-06f, -9.560153558e-06f};
  q = q * u + f32x2{7.497344632e-05f, 7.497344632e-05f};
  q = q * u + f32x2{-2.843466646e-04f, -2.843466646e-04f};
  q = q * u + f32x2{1.498938855e-05f, 1.498938855e-05f};
  q = q * u + f32x2{6.931120995e-03f, 6.931120995e-03f};
  q = q * u + f32x2{-5.243476480e-02f, -5.243476480e-02f};
  q = q * u + f32x2{-4.592214525e-01f, -4.592214525e-01f};
  q = q * u + f32x2{-1.151104212e+00f, -1.151104212e+00f};
  const f32x2 t = q * u;
  const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  const f32x2 m = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)};
  return m - (a * 0.5f) * e;
}

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
  int abl;                    // ablation bits (tools/bench_cnx.py, VS_CNX_ABL): 1 no GELU, 2 no pwconv2 MFMAs, 4 no pwconv1 MFMAs, 8 no output stores
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
          const f32x2 h = gelu2(f32x2{acc1[pb][e], acc1[pb][e + 1]} * a.acc_mul1 + f32x2{b1c, b1c});
          sq2 = h * h + sq2;
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
            const f32x2 x1 = f32x2{acc1[pb][e], acc1[pb][e + 1]} * a.acc_mul1 + f32x2{b1[e >> 2][e & 3], b1[e >> 2][(e & 3) + 1]};
            const f32x2 h = (CNX_ABL(a) & 1) ? x1 : gelu2(x1);
            const f32x2 h3 = h * f32x2{sc[e >> 2][e & 3], sc[e >> 2][(e & 3) + 1]} + f32x2{be[e >> 2][e & 3], be[e >> 2][(e & 3) + 1]};   // the unfused A transform (a_mul = 1)
            hi[v] = (_Float16)h3[0];
            hi[v + 1] = (_Float16)h3[1];
            lo[v] = (_Float16)(h3[0] - (float)hi[v]);
            lo[v + 1] = (_Float16)(h3[1] - (float)hi[v + 1]);
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

template <int KS1, int PB>
int launch_cnx(const CnxArgs& a, bool stats, hipStream_t st) {
  const unsigned grid = (unsigned)(a.rows / (128 * PB));
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

// stats != 0: part32 [rows / 32][4C] <- per-group sums of gelu(pwconv1)^2;  stats == 0: out = res + pwconv2(grn(gelu(pwconv1))) (out may alias res)
extern "C" int vs_cnx_block(const void* tn_planes, const void* wimg, int C, int64_t rows, int HW, int stats, float acc_mul1, float acc_mul2,
                            const float* scale, int64_t scale_ld, const float* bias2, const float* res, int64_t res_ld, float* out, int64_t out_ld,
                            float* part32, void* stream) {
  VS_REQUIRE(tn_planes && wimg && rows > 0 && HW > 0 && vs_cnx_block_supported(C, rows, HW));
  VS_REQUIRE(stats ? part32 != nullptr : (scale && bias2 && res && out && scale_ld >= 4 * C && (scale_ld & 3) == 0 && res_ld >= C && out_ld >= C && (res_ld & 3) == 0 && (out_ld & 3) == 0 && (((uintptr_t)res | (uintptr_t)out) & 15) == 0));
  VS_REQUIRE((((uintptr_t)tn_planes) & 15) == 0 && (((uintptr_t)wimg) & 15) == 0 && (stats || (((uintptr_t)scale) & 15) == 0));
  CnxArgs a{static_cast<const char*>(tn_planes), static_cast<const char*>(wimg), scale, bias2, res, out, part32, rows, scale_ld, res_ld, out_ld, HW,
            acc_mul1, acc_mul2, 0};
  static const int abl = [] { const char* e = getenv("VS_CNX_ABL"); return e ? atoi(e) : 0; }();
  a.abl = abl;
  // C = 96: the statistics launch takes 64 pixels per wave (weight fragments shared by two pixel blocks, one wave per SIMD: 52 vs 56 us at 32
  // frames), the apply launch 32 (139 + 80 registers: TWO workgroups per CU, whose MFMAs cover each other's GELU / split arithmetic: 103 vs 130 us)
  if (C == 96) return stats ? launch_cnx<6, 2>(a, true, (hipStream_t)stream) : launch_cnx<6, 1>(a, false, (hipStream_t)stream);
  return launch_cnx<12, 1>(a, stats != 0, (hipStream_t)stream);
}
