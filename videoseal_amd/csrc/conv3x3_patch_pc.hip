// Wave-specialised version of the 3x3 patch kernel (conv3x3_patch.hip): 512 threads, one workgroup per CU.
//   waves 0-3  consumers : nothing but ds_read_b128 fragment loads and MFMAs (template NP: 3 x bf16 split, 6 v_mfma_f32_32x32x16_bf16
//                          per 32x32x16 block; 2 x f16 split, 3 v_mfma_f32_32x32x16_f16 -- conv_common.h).  Fragments are double-buffered in registers: while the MFMAs of tap-step s
//                          run, the fragment reads of step s+1 are already in flight (its weight tile was published one
//                          barrier earlier, its A rows come from the resident input patch), so the matrix pipe does not
//                          drain at the barriers.
//   waves 4-5  activation producers : once per 16-channel chunk load + split the 10 x 18 input patch of the NEXT chunk
//                          into the second patch buffer (loads at tap 0, split + ds_write at tap 4).
//   waves 6-7  weight producers : LDS-DMA (global_load_lds_dwordx4) of the pre-split weight tile of step s+5 into a 6-deep
//                          LDS ring, counted s_waitcnt keeps three tiles (~1 us) in flight across the barriers.
// With the patch staging the producer work is ~1/9 of the im2col kernel's, so the consumers are never starved and the
// workgroup is bound by the MFMA issue of its four consumer waves.  One s_barrier per tap-step.
// Second K phase (ResnetBlock's 1x1 res_conv on the block input): the producers stage the in2 rows as plain 128-row
// chunks in the same two buffers; consumers do not prefetch across those (short) steps.
#include <type_traits>
#include "conv_common.h"
#include "lds_dma.h"

int vs_splitk_epilogue(const vs_conv_desc_t& d, int M, hipStream_t st);   // gemm1x1_pc.hip

namespace {

using namespace vsconv;

constexpr int TW = 16, PW = TW + 2;
// tile geometry: TH x 16 output pixels.  TH = 8: 128 pixels, consumers 2 x 2 (64 px x 32*TN ch each), BN = 64*TN.
//                                        TH = 16: 256 pixels, consumers 4 x 1 (64 px x 32*TN ch each), BN = 32*TN -- for
// 64-channel layers (768->64 @64^2): with 128-pixel tiles a wave reads 9 fragments per 12 MFMAs and the kernel is LDS-bound.
template <int TH_>
struct Geo {
  static constexpr int TH = TH_, PH = TH_ + 2, PROWS = PW * PH;              // 180 / 324 patch pixels
  static constexpr int PITEMS = PROWS * 4;
  static constexpr int BMP = TH_ * TW;
  static constexpr int NPI = (PITEMS + 127) / 128;                            // patch float4 items per producer thread (6 / 11)
  static constexpr int IPT = (NPI + 5) / 6;                                   // of which per tap-step (taps 2..7)
  template <int NP> static constexpr int p_bytes() { return NP * PROWS * ROWB; }   // one patch buffer, NP 16-bit planes
  static constexpr int WN = TH_ == 8 ? 2 : 1, WM = 4 / WN;
};

// one wave moves a 1 KiB block global -> LDS (lane l: 16 bytes at gp, landing at lds_base + 16*l)
__device__ __forceinline__ void dma_1k(const char* gp, unsigned char* lds_base) { vs_lds_dma16(gp, lds_base); }      // lds_dma.h

template <int TN, int TH_, int NP>
__global__ __launch_bounds__(512, 2) void conv3x3_patch_pc_kernel(const vs_conv_desc_t d, const int tiles_x, const int tiles_y,
                                                                  const int mtiles, const int ntiles, const int cps) {
  using G = Geo<TH_>;
  using AR = Arith<NP>;
  constexpr int TH = G::TH, PROWS = G::PROWS, PITEMS = G::PITEMS, BMP = G::BMP, NPI = G::NPI, P_BYTES = G::template p_bytes<NP>();
  constexpr int WBLK = NP * 1024;                        // bytes of one (32 rows x 16 k) weight block, all planes
  constexpr int TM = 2;
  constexpr int BN = G::WN * TN * 32;
  constexpr int NG = BN / 32;                            // 32-row weight groups per tile
  constexpr int NRING = TN >= 3 ? 5 : 6;                 // weight ring depth (LDS: 2 patches + NRING tiles <= 160 KiB)
  constexpr int B_STAGE = NP * BN * 32;                  // NP planes of [BN rows][16 halves], rows unpadded (DMA-written)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * P_BYTES + NRING * B_STAGE];
  unsigned char* const Pbuf = smem;
  unsigned char* const Bring = smem + 2 * P_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, g = lane >> 5;

  const int bm = blockIdx.x % mtiles;
  const int bn = (blockIdx.x / mtiles) % ntiles;
  const int ks = blockIdx.x / (mtiles * ntiles);        // K slice (split_k > 1), see below
  const int n0 = bn * BN;
  const int tx = bm % tiles_x;
  const int ty = (bm / tiles_x) % tiles_y;
  const int fb = bm / (tiles_x * tiles_y);
  const int y0 = ty * TH, x0 = tx * TW;

  // split_k > 1 (few output tiles, e.g. 4 key frames of a streaming chunk: 64 tiles for 256 CUs): slice ks < split_k
  // accumulates the 16-channel chunks [ks*cps, (ks+1)*cps) of phase 1; one extra slice (ks == split_k) does the 1x1
  // second phase.  All of them store raw partial sums; splitk_epilogue_kernel applies bias / activation / residual.
  const int spt_all = d.CinP / BK;
  const int n1_all = 9 * spt_all;
  const int sk = d.split_k > 1 ? d.split_k : 1;
  const bool p2_slice = sk > 1 && ks == sk;
  const int c_off = (sk > 1 && !p2_slice) ? ks * cps : 0;
  const int spt = p2_slice ? 0 : (sk > 1 ? min(cps, spt_all - c_off) : spt_all);
  const int n1 = 9 * spt;
  const int n2 = (d.in2 && (sk == 1 || p2_slice)) ? d.Cin2P / BK : 0;
  const int total = n1 + n2;
  const int abl = VS_KERNEL_ABL(d);   // debug ablation (tools/bench_ppc.py): 1 no weight DMA, 4 no activation staging

  if (wave >= 6) {
    // ================================================================== weight producers (waves 6-7): LDS-DMA only
    // Tile t (= the [BN x 16] slice of the split weights that step t multiplies) goes to ring stage t % NRING.  It is
    // issued NRING-1 steps before its step and must have landed one barrier before the consumers prefetch it, i.e. at
    // the end of step s everything up to tile s+2 is complete: at most the NRING-3 youngest tiles stay in flight.
    // These waves issue no ordinary loads, so the counted s_waitcnt below is exact (hipcc would drain to vmcnt(0)).
    const int bw = wave - 6;
    // weights in LDS-image order (engine.pack_blocked): [N/32 groups][step t = chunk*9 + tap][3 planes][1 KiB], so every DMA
    // instruction reads 1 KiB of consecutive memory (whole cache lines) and a tile's blocks follow each other in step order
    const char* const wblk = reinterpret_cast<const char*>(d.wt_blk);
    const int64_t w2delta = d.wt2_blk ? reinterpret_cast<const char*>(d.wt2_blk) - wblk : 0;
    const int g0 = n0 / 32;
    const int ngroups = (d.N + 31) / 32;
    constexpr int NGW = NG / 2;                          // weight groups per producer wave
    int gsel[NGW];
#pragma unroll
    for (int q = 0; q < NGW; ++q) {
      const int gi = g0 + bw * NGW + q;
      gsel[q] = gi < ngroups ? gi : ngroups - 1;         // tile wider than N: re-read a valid group (columns are discarded)
    }
    auto dma_tile = [&, n1, n2, n1_all, c_off, bw, lane](const int t) __attribute__((always_inline)) {
      unsigned char* st = Bring + (t % NRING) * B_STAGE;
      // NB: never select between two captured variables here (ph2 ? n2 : n1_all): LLVM turns select-of-loads into a load of a
      // selected closure address, which pins the whole closure (and every uniform in it) in scratch memory
      const int ph2 = t >= n1 ? 1 : 0;
      const int nsel = n1_all + ph2 * (__builtin_amdgcn_readfirstlane(n2) - n1_all);     // blocks per group in this phase
      const int64_t off = (int64_t)ph2 * w2delta + (int64_t)(t - ph2 * n1 + (1 - ph2) * c_off * 9) * WBLK + lane * 16;
#pragma unroll
      for (int q = 0; q < NGW; ++q) {
        const char* gp = wblk + off + (int64_t)gsel[q] * nsel * WBLK;
#pragma unroll
        for (int p = 0; p < NP; ++p) dma_1k(gp + p * 1024, st + p * (BN * 32) + (bw * NGW + q) * 1024);
      }
    };
    constexpr int ND = NP * NGW;                         // DMA instructions per wave per tile
#pragma unroll
    for (int t = 0; t < NRING - 1; ++t)
      if (t < total) dma_tile(t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < total; ++s) {
      if (s + NRING - 1 < total) {
        if (!(abl & 1)) dma_tile(s + NRING - 1);                         // stage of tile s-1: its fragments were consumed before the last barrier
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NRING - 3) * ND) : "memory");   // tiles <= s+2 have landed
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  if (wave >= 4) {
    // ================================================================== activation producers (waves 4-5)
    const int pt = tid & 127;
    const bool reflect = d.pad_mode == VS_PAD_REFLECT;
    unsigned p_off[NPI];
    int p_lds[NPI];
    bool p_ok[NPI], p_have[NPI];
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      const int item = pt + i * 128;
      p_have[i] = item < PITEMS;
      const int prow = p_have[i] ? item >> 2 : 0;
      const int k4 = (item & 3) * 4;
      int iy = y0 - 1 + prow / PW, ix = x0 - 1 + prow % PW;
      if (reflect) {
        iy = iy < 0 ? -iy : (iy >= d.H ? 2 * d.H - 2 - iy : iy);
        ix = ix < 0 ? -ix : (ix >= d.W ? 2 * d.W - 2 - ix : ix);
      }
      const bool ok = iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
      p_ok[i] = ok && p_have[i];
      p_off[i] = ok ? (unsigned)(((int64_t)fb * d.in_sb + (int64_t)iy * d.in_sy + (int64_t)ix * d.in_sx) * 4) : 0u;     // (without the lane's channel offset)
      p_lds[i] = prow * ROWB + k4 * 2;
    }
    constexpr int NQ = BMP * 4 / 128;                    // in2 float4 items per thread per chunk (4)
    unsigned q_off[NQ];
    bool q_ok[NQ];
    const int qk4 = (pt & 3) * 4;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int p = (pt + i * 128) >> 2;
      const int y = y0 + (p >> 4), x = x0 + (p & 15);
      q_ok[i] = y < d.H && x < d.W;
      const int64_t m = ((int64_t)fb * d.H + (q_ok[i] ? y : 0)) * d.W + (q_ok[i] ? x : 0);
      q_off[i] = (unsigned)((m * d.in2_ld) * 4);                                          // (without the lane's channel offset)
    }
    const float amul = NP == 2 ? d.a_mul : 1.f;
    f32x4 rp[NPI];          // patch registers
    f32x4 rq[NQ];           // in2 rows

    auto load_patch = [&, pt, c_off](const int ccl) __attribute__((always_inline)) {
      const int cc = c_off + ccl;                                         // global 16-channel chunk
      const char* base = reinterpret_cast<const char*>(d.in) + (int64_t)cc * (BK * 4);
      const bool cok = (cc * BK + (pt & 3) * 4) < d.Cin;
      // every item is loaded UNCONDITIONALLY from a valid address (items outside the image / the patch have offset 0, lanes past the last
      // channel re-read channels 0-3 of their pixel) and zeroed by a select: with `if (have && cok) v = load` each of the NPI loads sat in its
      // own divergent region and hipcc waited vmcnt(0) behind every one of them -- NPI dependent round trips per chunk instead of one batch
      // (round 5).  The lane's channel offset is part of the base, NOT of the pixel offsets: a lane past the last channel that kept it read
      // up to 48 bytes behind the last pixel of a 1 - 12-channel tensor (a sporadic memory fault in test_conv_gemm_matches_conv2d, tile 15).
      const char* const basel = cok ? base + (pt & 3) * 16 : reinterpret_cast<const char*>(d.in);
      f32x4 v[NPI];
#pragma unroll
      for (int i = 0; i < NPI; ++i) v[i] = *reinterpret_cast<const f32x4*>(basel + p_off[i]);
#pragma unroll
      for (int i = 0; i < NPI; ++i) rp[i] = (p_ok[i] && cok) ? v[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // item i of the patch is split + stored at tap `first + i` (first = -1: all at once): spreading the VALU burst
    // over six steps keeps these waves from arriving late at one barrier per chunk
    auto store_patch = [&](const int pb, const int tap, const int first) __attribute__((always_inline)) {
      unsigned char* Ps = Pbuf + pb * P_BYTES;
#pragma unroll
      for (int i = 0; i < NPI; ++i)
        if (p_have[i] && (first < 0 || tap == first + i / G::IPT)) {
          u32x2 pl[NP];
          split4n<NP>(rp[i], amul, pl);
#pragma unroll
          for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(Ps + p * PROWS * ROWB + p_lds[i]) = pl[p];
        }
    };
    auto load_rows2 = [&, qk4](const int c2) __attribute__((always_inline)) {      // in2 chunk c2 -> registers
      const char* base = reinterpret_cast<const char*>(d.in2) + (int64_t)c2 * (BK * 4);
      const bool cok = (c2 * BK + qk4) < d.Cin2;
      const char* const basel = cok ? base + qk4 * 4 : reinterpret_cast<const char*>(d.in2);        // (unconditional loads, as load_patch)
      f32x4 v[NQ];
#pragma unroll
      for (int i = 0; i < NQ; ++i) v[i] = *reinterpret_cast<const f32x4*>(basel + q_off[i]);
#pragma unroll
      for (int i = 0; i < NQ; ++i) rq[i] = (q_ok[i] && cok) ? v[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_rows2 = [&, pt, qk4](const int pb) __attribute__((always_inline)) {  // as plain pixel rows 0..127
      unsigned char* Ps = Pbuf + pb * P_BYTES;
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        u32x2 pl[NP];
        split4n<NP>(rq[i], amul, pl);
        const int off = ((pt + i * 128) >> 2) * ROWB + qk4 * 2;
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(Ps + p * PROWS * ROWB + off) = pl[p];
      }
    };

    if (spt > 0) {
      load_patch(0);
      store_patch(0, 0, -1);
      if (spt > 1 && !(abl & 4)) load_patch(1);          // patch cc+1 is requested at tap 8 of chunk cc-1: three steps before its first use
    } else {                                             // phase-2-only slice: first in2 chunk goes straight to buffer 0
      load_rows2(0);
      store_rows2(0);
      if (n2 > 1) load_rows2(1);
    }
    __syncthreads();
    int cc = 0, tap = 0;
    for (int s = 0; s < total; ++s) {
      if (abl & 4) {
      } else if (s < n1) {
        if (tap >= 2 && tap < 8 && cc + 1 < spt) store_patch((cc + 1) & 1, tap, 2);   // split + stored at taps 2..7 (other buffer)
        if (tap == 8 && cc + 2 < spt) load_patch(cc + 2);                 // registers are free again: request the patch after next
        if (n2 > 0 && cc == spt - 1) {
          if (tap == 1) load_rows2(0);
          if (tap == 8) { store_rows2(spt & 1); if (n2 > 1) load_rows2(1); }   // chunk "spt" = first in2 chunk
        }
        if (++tap == 9) { tap = 0; ++cc; }
      } else {
        const int c2 = s - n1;                                             // consumers are on in2 chunk c2
        if (c2 + 1 < n2) { store_rows2((spt + c2 + 1) & 1); if (c2 + 2 < n2) load_rows2(c2 + 2); }
      }
      __syncthreads();
    }
    return;
  }

  // ==================================================================== consumers (2 x 2 waves, 64 px x 32*TN... each)
  const int wm = G::WN == 2 ? wave >> 1 : wave, wn = G::WN == 2 ? wave & 1 : 0;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // Operands are SWAPPED in the MFMA (weights first), so the accumulator holds C^T: element e of acc[i][j] is
  //   channel  cbase[j] + (e & 3) + 8 * (e >> 2)      (cbase already contains the half-wave's 4 * g)
  //   pixel    (wm * TM + i) * 32 + r
  // i.e. a lane owns 4 x 4 CONSECUTIVE channels of one pixel per 32 x 32 block and the epilogue works on float4 groups: 24 16-byte
  // stores per lane instead of 96 4-byte ones, one validity test per pixel.  (The scalar epilogue was 20 k instructions with 700
  // bytes of scratch per lane; stores alone were 16 us of a 386 us launch.)  Same products, same K order: bit-identical values.
  int cbase[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) cbase[j] = n0 + (wn * TN + j) * 32 + 4 * g;
  auto ld4 = [](const float* p, int n, int N) __attribute__((always_inline)) -> f32x4 {      // p[n .. n+3], zero past N
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (p) {
      if (n + 4 <= N) v = *reinterpret_cast<const f32x4*>(p + n);
      else
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < N) v[e] = p[n + e];
    }
    return v;
  };
  auto grp = [&](int i, int j, int q) __attribute__((always_inline)) -> f32x4 {
    return f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
  };
  auto put = [&](int i, int j, int q, const f32x4& v) __attribute__((always_inline)) {
    acc[i][j][4 * q] = v[0]; acc[i][j][4 * q + 1] = v[1]; acc[i][j][4 * q + 2] = v[2]; acc[i][j][4 * q + 3] = v[3];
  };
  // pixel of this lane in block i, its row in the output and whether it lies inside the image
  int64_t mrow[TM];
  bool pin[TM];
  int pcls[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int p = (wm * TM + i) * 32 + r;
    const int y = y0 + (p >> 4), x = x0 + (p & 15);
    pin[i] = y < d.H && x < d.W;
    mrow[i] = ((int64_t)fb * d.H + (pin[i] ? y : 0)) * d.W + (pin[i] ? x : 0);
    pcls[i] = (y == 0 ? 0 : (y >= d.H - 1 ? 2 : 1)) * 3 + (x == 0 ? 0 : (x >= d.W - 1 ? 2 : 1));
  }
  // v = act(acc (+ table) + bias1) + bias2 on every float4 group (the activation sits between the two K phases)
  auto finish_phase1 = [&](const bool with_b2) __attribute__((always_inline)) {
    const float* tb = (d.tile_hint & VS_CONV_PRE) ? d.a_scale + (int64_t)fb * d.a_scale_ld : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = cbase[j] + 8 * q;
        const f32x4 b1 = ld4(d.bias, n, d.N);
        const f32x4 b2 = with_b2 ? ld4(d.bias2, n, d.N) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          f32x4 v = grp(i, j, q) + b1;
          if (tb) v += ld4(tb + pcls[i] * d.N, n, d.N);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = vs_apply_act(v[e], d.act);
          put(i, j, q, v + b2);
        }
      }
  };
  int a_patch[TM], a_plain[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int p = (wm * TM + i) * 32 + r;
    a_patch[i] = ((p >> 4) * PW + (p & 15)) * ROWB + g * 16;     // tap (0,0) row of output pixel p inside the 10 x 18 patch
    a_plain[i] = p * ROWB + g * 16;                              // phase 2: plain pixel rows
  }
  const int b_frag = (wn * TN) * 1024 + (2 * r + (g ^ ((r >> 3) & 1))) * 16;   // pack_blocked's bank swizzle

  struct Frags { bf16x8 a[TM][NP]; bf16x8 b[TN][NP]; };
  Frags F0, F1;
  auto load_frags = [&, n1, spt, b_frag](Frags& F, const int s) __attribute__((always_inline)) {
    const unsigned char* Bb = Bring + (s % NRING) * B_STAGE;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < NP; ++p) F.b[j][p] = *reinterpret_cast<const bf16x8*>(Bb + p * (BN * 32) + b_frag + j * 1024);
    int pb, toff;
    if (s < n1) {
      const int cc = s / 9, tap = s - cc * 9;
      const int ky = tap / 3, kx = tap - ky * 3;
      pb = cc & 1;
      toff = (ky * PW + kx) * ROWB;
    } else {
      pb = (spt + (s - n1)) & 1;
      toff = 0;
    }
    const bool ph2 = s >= n1;
    const unsigned char* Ps = Pbuf + pb * P_BYTES + toff;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int ao = ph2 ? a_plain[i] : a_patch[i];
#pragma unroll
      for (int p = 0; p < NP; ++p) F.a[i][p] = *reinterpret_cast<const bf16x8*>(Ps + p * PROWS * ROWB + ao);
    }
  };
  auto mfma_all = [&](const Frags& F) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < AR::NPROD; ++q) {   // smallest partial products first; consecutive MFMAs hit different accumulators
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = AR::mfma(F.b[j][AR::PB[q]], F.a[i][AR::PA[q]], acc[i][j]);   // C^T: see the epilogue
    }
  };

  __syncthreads();                 // prologue of the producers
  if (abl & 16) __builtin_amdgcn_s_setprio(3);      // experiment (tools/bench_ppc.py): consumers win every issue arbitration against the producer waves
  int s = 0;
  if (n1 > 0) {
    load_frags(F0, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): otherwise hipcc drains the PREFETCH below before the first MFMA of every iteration
    // phase 1: the fragment reads of step s+1 are in flight while the MFMAs of step s issue (straight-line body, no
    // conditionals: the compiler's waitcnt placement is exact only then)
    // (measured, round 2: raw s_barrier + sched_barrier-pinned prefetch instead of __syncthreads() -- hipcc then issues all 15 fragment
    // reads of step s+1 ahead of the MFMAs of step s instead of draining 13 of them with lgkmcnt(0) in front of the barrier -- left
    // the per-step time at 0.80 us = 36 MFMAs + ~1/3 bubbles: the LDS round trip in front of the barrier is not what the loop waits for)
    for (; s + 2 < n1; s += 2) {
      load_frags(F1, s + 1);
      mfma_all(F0);
      __syncthreads();
      load_frags(F0, s + 2);
      mfma_all(F1);
      __syncthreads();
    }
    if (s + 1 < n1) {
      load_frags(F1, s + 1);
      mfma_all(F0);
      __syncthreads();
      mfma_all(F1);
      __syncthreads();
    } else {
      mfma_all(F0);
      __syncthreads();
    }
  }
  if (sk > 1) {                    // K slice: raw partial sums -> workspace [slice][M][splitk_ld]
    for (s = n1; s < total; ++s) {
      load_frags(F0, s);
      mfma_all(F0);
      __syncthreads();
    }
    if constexpr (NP == 2) scale_all<TM, TN>(acc, p2_slice ? d.acc_mul2 : d.acc_mul);     // exact (power of two): the slices add up in real units
    const int64_t Mrows = (int64_t)d.B * d.H * d.W;
    float* ws = d.splitk_ws + (int64_t)ks * Mrows * d.splitk_ld;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (!pin[i]) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = cbase[j] + 8 * q;
          if (n < d.N) *reinterpret_cast<f32x4*>(ws + mrow[i] * d.splitk_ld + n) = grp(i, j, q);
        }
    }
    return;
  }
  if constexpr (NP == 2) scale_all<TM, TN>(acc, d.acc_mul);
  if (n2 > 0) {
    finish_phase1(true);
    if constexpr (NP == 2) scale_all<TM, TN>(acc, 1.f / d.acc_mul2);     // into the units of the phase-2 products (exact: powers of two)
    for (s = n1; s < total; ++s) {       // short: no prefetch across these steps (the rows are published one barrier before)
      load_frags(F0, s);
      mfma_all(F0);
      __syncthreads();
    }
    if constexpr (NP == 2) scale_all<TM, TN>(acc, d.acc_mul2);
  } else {
    finish_phase1(false);
  }
  if (abl & 32) return;             // ablation (tools/bench_ppc.py): no output stores
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (!pin[i]) continue;
    float* orow = d.out + mrow[i] * d.out_ld + d.out_coff;
    const float* rrow = d.res ? d.res + mrow[i] * d.res_ld : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = cbase[j] + 8 * q;
        if (n >= d.n_store) continue;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};          // columns in [N, n_store) are written as zeros
        if (n < d.N) {
          v = grp(i, j, q);
          if (rrow) v += *reinterpret_cast<const f32x4*>(rrow + n);
          if (n + 4 > d.N)           // N % 4 != 0: the tail of the last group is padding
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e >= d.N) v[e] = 0.f;
        }
        *reinterpret_cast<f32x4*>(orow + n) = v;
      }
  }
}

template <int TN, int TH_, int NP>
int launch_ppc_np(const vs_conv_desc_t& d, hipStream_t st) {
  constexpr int BN = Geo<TH_>::WN * TN * 32;
  constexpr int TH = TH_;
  const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
  const int sk = d.split_k > 1 ? d.split_k : 1;
  const int64_t mt = (int64_t)d.B * tiles_x * tiles_y, nt = cdiv64(sk > 1 ? d.N : d.n_store, BN);
  const int spt = d.CinP / BK;
  const int cps = (spt + sk - 1) / sk;
  if (sk > 1 && (int64_t)(sk - 1) * cps >= spt) return VS_ERR_BAD_ARG;       // an empty K slice
  const int slices = sk > 1 ? sk + (d.in2 ? 1 : 0) : 1;
  if (mt * nt * slices > 0x7fffffffLL) return VS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((conv3x3_patch_pc_kernel<TN, TH_, NP>), dim3((unsigned)(mt * nt * slices)), dim3(512), 0, st, d, tiles_x, tiles_y, (int)mt,
                     (int)nt, cps);
  int rc = vs_launch_status();
  if (rc != VS_OK || sk == 1) return rc;
  return vs_splitk_epilogue(d, (int)((int64_t)d.B * d.H * d.W), st);
}

template <int TN, int TH_ = 8>
int launch_ppc(const vs_conv_desc_t& d, hipStream_t st) {
  return d.arith == 2 ? launch_ppc_np<TN, TH_, 2>(d, st) : launch_ppc_np<TN, TH_, 3>(d, st);
}

}  // namespace

// tile 15 = 128 pixels x 128 channels, tile 9 (alias kept free) -- called from vs_conv_gemm when patch_ok
int vs_conv3x3_patch_pc_dispatch(const vs_conv_desc_t& d, int tile, hipStream_t st) {
  switch (tile) {
    case 15: return launch_ppc<2>(d, st);
    case 16: return launch_ppc<3>(d, st);
    case 19: return launch_ppc<1>(d, st);
    case 21: return launch_ppc<2, 16>(d, st);      // 256 pixels x 64 channels
    default: return VS_ERR_UNSUPPORTED;
  }
}
