// Producer/consumer ("wave-specialised") implicit-GEMM convolution on the bf16 matrix cores with the exact
// 3 x bf16 operand split (see conv_gemm.hip for the arithmetic).
//
// Why: in the 4-wave kernel every wave does load -> split -> LDS store -> LDS read -> MFMA in turn, and the measured
// time is the SUM of those phases (ablation in tools/bench_conv.py).  Here a 512-thread workgroup puts two waves on
// every SIMD with different jobs, so the phases overlap by construction (MFMA and VALU are separate pipes):
//   waves 0-3  consumers : ds_read_b128 fragments + v_mfma_f32_32x32x16_bf16 only (2 x 2 waves, TM x TN tiles each)
//   waves 4-5  A producers: global_load (fp32 NHWC gather, two chunks ahead, 4 rotating register sets) -> truncation
//                          split into 3 bf16 planes -> ds_write
//   waves 6-7  B producers: the pre-split weights are packed host-side in the exact LDS image order
//                          ([n/32][chunk][plane][1 KiB], bank-swizzled), so one global_load_lds_dwordx4 per wave moves
//                          a whole 32-row block L2 -> LDS with no VGPRs, no ds_write, full 1 KiB lines; counted
//                          s_waitcnt vmcnt(N) keeps two chunks of DMA in flight across the barriers.
// A and B producers are DIFFERENT waves on purpose: hipcc drains vmcnt(0) at every use of an ordinary load result while
// an LDS-DMA is outstanding in the same wave, which would serialise the A prefetch (cdna_hip_programming.md, trap (b)).
// LDS ring of D stages (4 for 128-row tiles, 3 for 256-row tiles), chunk c in stage c % D, one s_barrier per chunk:
// while the consumers multiply chunk s the producers fill chunk s+D-1.
#include "conv_common.h"

namespace {

using namespace vsconv;

constexpr int PT = 512;          // threads per workgroup

// one wave moves a 1 KiB block global -> LDS (lane l: 16 bytes at gp, landing at lds_base + 16*l)
__device__ __forceinline__ void dma_1k(const char* gp, unsigned char* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                   (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

__device__ __forceinline__ int bswz(int r, int h) { return 2 * r + (h ^ ((r >> 3) & 1)); }   // 16-B slot of (row, k-half)

template <int TM, int TN, bool PREF, bool RAWB>
__global__ __launch_bounds__(PT, 2) void conv_gemm_pc_kernel(const vs_conv_desc_t d, const int M, const int mtiles) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int NG = BN / 32;                              // 32-row weight groups per tile
  constexpr int A_STAGE = 3 * BM * ROWB;
  constexpr int B_STAGE = 3 * NG * 1024;
  constexpr int STAGE = A_STAGE + B_STAGE;
  constexpr int NSTAGE = 3;                                // LDS ring depth (a 4-deep ring measured no faster)
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave & 3;                                 // index inside the role
  const int wm = cw >> 1, wn = cw & 1;
  const int r = lane & 31, g = lane >> 5;

  const int bm = blockIdx.x % mtiles;
  const int bn = blockIdx.x / mtiles;
  const int64_t m0 = (int64_t)bm * BM;
  const int n0 = bn * BN;

  const int spt = d.CinP / BK;
  const int n1 = d.KH * d.KW * spt;
  const int n2 = d.in2 ? d.Cin2P / BK : 0;
  const int n1e = n2 > 0 ? ((n1 + 1) & ~1) : n1;
  const int total = n1e + n2;
  const int HoWo = d.Ho * d.Wo;
  const bool reflect = d.pad_mode == VS_PAD_REFLECT;
  const int abl = (d.tile_hint >> 10) & 3;   // debug ablation: 1 = producers idle, 2 = consumers skip the MFMAs
  const bool chk_c = (d.Cin % BK) != 0 || (d.in2 && (d.Cin2 % BK) != 0);

  constexpr int D = NSTAGE;
  if (wave >= 6) {
    // ================================================================ B producers (waves 6-7): weight blocks by LDS-DMA
    const int bw = wave - 6;
    const char* const wblk = reinterpret_cast<const char*>(d.wt_blk);
    const int64_t w2off = d.wt2_blk ? reinterpret_cast<const char*>(d.wt2_blk) - wblk : 0;
    const int g0 = n0 / 32;                                  // first weight group of this tile
    const int ngroups = (d.N + 31) / 32;
    // DMA the weight blocks of chunk c into stage c % NSTAGE
    auto dma_stage = [&, n1, n2, n1e, g0, ngroups, bw, lane, wblk, w2off](const int c) __attribute__((always_inline)) {
      unsigned char* st = smem + (c % NSTAGE) * STAGE;
      // NB: never select between two captured variables here (c ? n2 : n1): LLVM turns select-of-loads into a load of
      // a selected closure address, which pins the whole closure (and every uniform in it) in scratch memory.
      const int nsel = n1 + (c >= n1e ? 1 : 0) * (__builtin_amdgcn_readfirstlane(n2) - n1);
      const int64_t gstride = (int64_t)nsel * 3072;    // bytes between consecutive 32-row groups
      int64_t woff;                                      // byte offset of (group g0, chunk c) from wblk
      if (c < n1) woff = ((int64_t)g0 * n1 + c) * 3072;
      else if (c < n1e) woff = (int64_t)g0 * n1 * 3072;                         // null chunk: A is zero, any finite block
      else woff = w2off + ((int64_t)g0 * __builtin_amdgcn_readfirstlane(n2) + (c - n1e)) * 3072;
      const char* src = wblk + woff;
      // weight groups of the tile are split between the two B-producer waves, three planes each
      unsigned char* const lb = st + A_STAGE;
#pragma unroll
      for (int q = 0; q < NG / 2; ++q) {
        const int gi = bw * (NG / 2) + q;
        const int gsafe = (g0 + gi) < ngroups ? gi : 0;      // tile wider than N: re-read a valid group
        const char* gp = src + gsafe * gstride + lane * 16;
        dma_1k(gp, lb + (0 * NG + gi) * 1024);
        dma_1k(gp + 1024, lb + (1 * NG + gi) * 1024);
        dma_1k(gp + 2048, lb + (2 * NG + gi) * 1024);
      }
    };
    constexpr int ND = 3 * NG / 2;                           // DMA instructions per wave per chunk
    // chunk c is issued D-1 steps before the consumers need it and has to be complete one barrier earlier: at the end
    // of step s everything except the two youngest chunks (D = 4) / the youngest chunk (D = 3) must have landed
#pragma unroll
    for (int c = 0; c < D - 1; ++c)
      if (c < total) dma_stage(c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < total; ++s) {
      if (s + D - 1 < total && !(abl & 1)) dma_stage(s + D - 1);
      if (s + D - 1 < total) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * ND) : "memory"); }
      else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __builtin_amdgcn_s_barrier();
    }
    return;
  }
  if (wave >= 4) {
    // ================================================================ A producers (waves 4-5): gather, split, ds_write
    constexpr int NA2 = BM * 4 / 128;                        // float4 loads per thread per chunk
    const int pt = tid & 127;
    const int k4 = (pt & 3) * 4;
    bool a_ok[NA2];
    int a_oy[NA2], a_ox[NA2];
    unsigned a_pix[NA2], a_cur[NA2], a_off2[NA2], a_soff[NA2];
    bool a_tap[NA2];
  #pragma unroll
    for (int i = 0; i < NA2; ++i) {
      const int row = (pt + i * 128) >> 2;
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

    int ld_ky = 0, ld_kx = 0, ld_cc = 0, ld_step = 0;

    auto refresh_tap = [&]() __attribute__((always_inline)) {
      const int64_t tap_delta = ((int64_t)(ld_ky - d.PH) * d.in_sy + (int64_t)(ld_kx - d.PW) * d.in_sx) * 4;
  #pragma unroll
      for (int i = 0; i < NA2; ++i) {
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

    // A chunk `ld_step` -> registers (chunks are requested strictly in order)
    auto load_a = [&, n1, n1e, k4, chk_c](f32x4 (&ra)[NA2]) __attribute__((always_inline)) {
      const int s = ld_step++;
      if (s < n1) {
        refresh_tap();   // K order is (channel chunk, tap): the 9 taps of a chunk re-read the same rows back to back (L1/L2 hits)
        const char* abase = reinterpret_cast<const char*>(d.in) + (int64_t)ld_cc * 4;
        if (!chk_c) {
  #pragma unroll
          for (int i = 0; i < NA2; ++i) ra[i] = *reinterpret_cast<const f32x4*>(abase + a_cur[i]);
        } else {
          const bool cok = (ld_cc + k4) < d.Cin;
  #pragma unroll
          for (int i = 0; i < NA2; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (cok) v = *reinterpret_cast<const f32x4*>(abase + a_cur[i]);
            ra[i] = v;
          }
        }
        if (d.a_scale) {
          const char* sbase = reinterpret_cast<const char*>(d.a_scale) + (int64_t)ld_cc * 4;
          const f32x4 sh = *reinterpret_cast<const f32x4*>(d.a_shift + ld_cc + k4);
  #pragma unroll
          for (int i = 0; i < NA2; ++i) ra[i] = ra[i] * *reinterpret_cast<const f32x4*>(sbase + a_soff[i]) + sh;
        }
  #pragma unroll
        for (int i = 0; i < NA2; ++i)
          if (!a_tap[i]) ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (++ld_kx == d.KW) {
          ld_kx = 0;
          if (++ld_ky == d.KH) { ld_ky = 0; ld_cc += BK; }
        }
      } else if (s < n1e) {
  #pragma unroll
        for (int i = 0; i < NA2; ++i) ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        const int cb = (s - n1e) * BK;
        const char* abase = reinterpret_cast<const char*>(d.in2) + (int64_t)cb * 4;
        const bool cok = (cb + k4) < d.Cin2;
  #pragma unroll
        for (int i = 0; i < NA2; ++i) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (a_ok[i] && cok) v = *reinterpret_cast<const f32x4*>(abase + a_off2[i]);
          ra[i] = v;
        }
      }
    };

    // split the A registers of chunk c into three bf16 planes and store them into stage c % NSTAGE
    auto store_stage = [&, pt, k4](const f32x4 (&ra)[NA2], const int c) __attribute__((always_inline)) {
      unsigned char* st = smem + (c % NSTAGE) * STAGE;
  #pragma unroll
      for (int i = 0; i < NA2; ++i) {
        const int row = (pt + i * 128) >> 2;
        u32x2 p1, p2, p3;
        split4(ra[i], p1, p2, p3);
        const int off = row * ROWB + k4 * 2;
        *reinterpret_cast<u32x2*>(st + off) = p1;
        *reinterpret_cast<u32x2*>(st + BM * ROWB + off) = p2;
        *reinterpret_cast<u32x2*>(st + 2 * BM * ROWB + off) = p3;
      }
    };

    // chunk c waits in register set c & 3; it is loaded two steps before it is split into stage c % D
    f32x4 RA[4][NA2];
#pragma unroll
    for (int c = 0; c < D - 1; ++c)
      if (c < total) { load_a(RA[c & 3]); store_stage(RA[c & 3], c); }
#pragma unroll
    for (int c = D - 1; c < D + 1; ++c)
      if (c < total) load_a(RA[c & 3]);
    __syncthreads();
    for (int s = 0; s < total; s += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ss = s + j;
        if (ss >= total) break;
        const int cfill = ss + D - 1, cload = cfill + 2;
        if (!(abl & 1)) {
          if (cload < total) load_a(RA[(j + D + 1) & 3]);
          if (cfill < total) store_stage(RA[(j + D - 1) & 3], cfill);
        }
        __syncthreads();
      }
    }
    return;
  }

  // ================================================================== consumers (waves 0-3)
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
  const int a_frag = (wm * TM * 32 + r) * ROWB + g * 16;                     // + p*BM*ROWB + i*32*ROWB
  const int b_frag = A_STAGE + (wn * TN) * 1024 + bswz(r, g) * 16;           // + (p*NG + j)*1024

  // Fragments are double-buffered in registers: while the MFMAs of chunk s run, the ds_reads of chunk s+1 (published by
  // the barrier that ended step s-1) are already in flight, so the matrix pipe does not drain at the barrier.
  struct Frags { bf16x8 a[TM][3]; bf16x8 b[TN][3]; };
  Frags F0, F1;
  auto load_frags = [&, a_frag, b_frag](Frags& F, const int c) __attribute__((always_inline)) {
    const unsigned char* st = smem + (c % NSTAGE) * STAGE;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) F.b[j][p] = *reinterpret_cast<const bf16x8*>(st + b_frag + (p * NG + j) * 1024);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) F.a[i][p] = *reinterpret_cast<const bf16x8*>(st + a_frag + p * BM * ROWB + i * 32 * ROWB);
  };
  auto mfma_all = [&](const Frags& F) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {   // smallest terms first
        f32x16 v = acc[i][j];
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[i][2], F.b[j][0], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[i][0], F.b[j][2], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[i][1], F.b[j][1], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[i][1], F.b[j][0], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[i][0], F.b[j][1], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[i][0], F.b[j][0], v, 0, 0, 0);
        acc[i][j] = v;
      }
  };
  auto compute = [&, a_frag, b_frag](const int c) __attribute__((always_inline)) {
    const unsigned char* st = smem + (c % NSTAGE) * STAGE;
    bf16x8 bf[TN][3];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) bf[j][p] = *reinterpret_cast<const bf16x8*>(st + b_frag + (p * NG + j) * 1024);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      bf16x8 af[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) af[p] = *reinterpret_cast<const bf16x8*>(st + a_frag + p * BM * ROWB + i * 32 * ROWB);
#pragma unroll
      for (int j = 0; j < TN; ++j) {   // smallest terms first
        f32x16 v = acc[i][j];
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[j][0], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[j][2], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[j][1], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[j][0], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[j][1], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[j][0], v, 0, 0, 0);
        acc[i][j] = v;
      }
    }
  };
  auto cbar = [&]() __attribute__((always_inline)) { if (RAWB) __builtin_amdgcn_s_barrier(); else __syncthreads(); };
  auto k_loop = [&, total](const int s_begin, const int s_end) __attribute__((always_inline)) {   // s_begin even; F0 holds chunk s_begin
    if (PREF) {
      for (int s = s_begin; s < s_end; s += 2) {
        if (s + 1 < total) load_frags(F1, s + 1);
        mfma_all(F0);
        cbar();
        if (s + 1 >= s_end) break;
        if (s + 2 < total) load_frags(F0, s + 2);
        mfma_all(F1);
        cbar();
      }
    } else {
      constexpr int CPB = NSTAGE == 4 ? 2 : 1;
      for (int s = s_begin; s < s_end; s += CPB) {
        if (!(abl & 2)) {
          compute(s);
          if (CPB == 2 && s + 1 < s_end) compute(s + 1);
        }
        cbar();
      }
    }
  };
  cbar();                                                    // stages 0 and 1 are filled
  if (PREF) load_frags(F0, 0);
  k_loop(0, n1e);
  if (n2 > 0) {   // phase 2: finish phase 1 in registers (bias, activation) and keep accumulating the 1x1 conv on top
    apply_act_all<TM, TN>(acc, bias1, bias2, d.act);
    k_loop(n1e, total);
  }

  // ---- epilogue
  if (n2 == 0) {
    float zero[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) zero[j] = 0.f;
    apply_act_all<TM, TN>(acc, bias1, zero, d.act);
  }
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
        float v = 0.f;
        if (n < d.N) {
          v = acc[i][j][e];
          if (d.res) v += d.res[m * d.res_ld + n];
        }
        d.out[m * d.out_ld + d.out_coff + n] = v;
      }
    }
  }
}

template <int TM, int TN, bool PREF, bool RAWB>
int launch_pc1(const vs_conv_desc_t& d, hipStream_t st) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  const int64_t M = (int64_t)d.B * d.Ho * d.Wo;
  const int64_t mt = cdiv64(M, BM), nt = cdiv64(d.n_store, BN);
  if (mt * nt > 0x7fffffffLL || M > 0x7fffffffLL) return VS_ERR_UNSUPPORTED;
  auto kern = conv_gemm_pc_kernel<TM, TN, PREF, RAWB>;
  static bool attr_set = false;
  if (!attr_set) {   // > 64 KiB of static LDS needs the opt-in on some ROCm versions; harmless otherwise
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 0);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(mt * nt)), dim3(PT), 0, st, d, (int)M, (int)mt);
  return vs_launch_status();
}

template <int TM, int TN>
int launch_pc(const vs_conv_desc_t& d, hipStream_t st) {
  const int v = (d.tile_hint >> 8) & 3;     // experiment bits: 1 = prefetch fragments, 2 = raw barriers
  (void)v;
  return launch_pc1<TM, TN, false, false>(d, st);
}

}  // namespace

// called from vs_conv_gemm (conv_gemm.hip) after validation.  tile: 6 = 256x128, 7 = 128x128, 8 = 128x64, 9 = 256x64
int vs_conv_gemm_pc_dispatch(const vs_conv_desc_t& d, int tile, hipStream_t st) {
  switch (tile) {
    case 6: return launch_pc<4, 2>(d, st);
    case 7: return launch_pc<2, 2>(d, st);
    case 8: return launch_pc<2, 1>(d, st);
    case 9: return launch_pc<4, 1>(d, st);
    default: return VS_ERR_UNSUPPORTED;
  }
}
