// 3x3 stride-1 "same" convolution on PRE-SPLIT activation planes (arith 2, "2 x f16"): the bottleneck chain of the U-Net
// (unet.py:24-39, eight ResnetBlocks of 3x3 384->384 @32x32 = 39 % of a step) where producer and consumer are the same kernel.
//
// Why a second kernel next to conv3x3_patch_pc.hip: with the 2 x f16 arithmetic a K16 step is 18 MFMAs per wave instead of 36, and the
// wave-specialised kernel (ONE consumer wave per SIMD, a barrier per step) cannot hide its barrier / fragment-read latencies behind
// that little matrix work any more (0.30 ms = 35 % of the MFMA rate at the bottleneck shape).  Here
//   * the activations arrive already split: two f16 planes [plane][C/16][pixel][16] of v * a_mul (written by the previous launch's
//     epilogue, or by vs_to_planes at the entry of the chain), i.e. EVERY operand is moved global -> LDS by LDS-DMA
//     (global_load_lds_dwordx4) and the loop contains no VGPR-destination load at all -- the one structure for which hipcc keeps
//     counted s_waitcnt vmcnt(N) across raw barriers (cdna_hip_programming.md, "three .s-level traps" (b));
//   * all EIGHT waves multiply (two per SIMD: one wave's barrier / LDS waits are covered by its sibling's MFMAs), 4 x 2 waves of
//     64 pixels x 32*TN channels on a 16 x 16-pixel x 64*TN-channel workgroup tile; waves 0-3 also issue the weight DMAs (TN per
//     step), waves 4-7 the activation DMAs (the 18 x 18-pixel patch of the next 16-channel chunk: 21 KiB per nine steps);
//   * patch layout in LDS: plane-major, pixel q of the patch, half h (8 channels) at 16-byte unit 2q + (h ^ ((q >> 3) & 1)) -- the
//     DMA writes lane-linear, so the permutation is applied on the SOURCE side (which pixel/half a lane fetches) and on the read
//     side; 16 consecutive pixels then cover all sixteen 4-bank slots (conflict-free ds_read_b128 without row padding);
//   * the channel-chunk-major plane layout makes a patch row 18 x 32 contiguous bytes in global memory (whole cache lines).
// Same products, same K order (chunk, tap), same scaling as conv3x3_patch_pc_kernel<.., NP = 2>: results are bit-identical to it.
// Second K phase (ResnetBlock's 1x1 res_conv on the block input, also as planes): plain 256-pixel rows, four 16 KiB LDS buffers,
// three chunks in flight.
#include <type_traits>

#include "conv_common.h"
#include "lds_dma.h"

int vs_splitk_epilogue(const vs_conv_desc_t& d, int M, hipStream_t st);   // gemm1x1_pc.hip

namespace {

using namespace vsconv;

constexpr int PT = 16, PPW = PT + 2, PPIX = PPW * PPW;      // 16 x 16 output pixels, 18 x 18 patch
constexpr int PL_UNITS = PPIX * 2;                          // 16-byte units per plane (648)
constexpr int P_DMAS = (2 * PL_UNITS + 63) / 64;            // 1 KiB DMA instructions per patch (21)
constexpr int P_BYTES = P_DMAS * 1024;
constexpr int Q_BYTES = 256 * 2 * 2 * 16;                   // phase 2: 256 pixels x 2 halves x 2 planes
constexpr int QX_OFF = 2 * P_BYTES;
constexpr int RING_OFF = QX_OFF + 2 * Q_BYTES;
constexpr int NRING = 6;

__device__ __forceinline__ void dma16(const char* gp, unsigned char* lds_base) { vs_lds_dma16(gp, lds_base); }      // lds_dma.h: lane l: 16 bytes at gp -> lds_base + 16 * l
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int TN>
__global__ __launch_bounds__(512, 2) void conv3x3_pl_kernel(const vs_conv_desc_t d, const int tiles_x, const int tiles_y, const int mtiles,
                                                            const int ntiles, const int cps) {
  using AR = Arith<2>;
  constexpr int TM = 2;
  constexpr int BN = 2 * TN * 32;
  constexpr int NG = BN / 32;                 // 32-row weight groups per tile
  constexpr int WBLK = 2048;                  // one (32 rows x 16 k) weight block, both planes
  constexpr int B_STAGE = NG * WBLK;
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING_OFF + NRING * B_STAGE];
  unsigned char* const Bring = smem + RING_OFF;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, g = lane >> 5;

  const int bm = blockIdx.x % mtiles;
  const int bn = (blockIdx.x / mtiles) % ntiles;
  const int ks = blockIdx.x / (mtiles * ntiles);          // K slice (split_k > 1), see below
  const int n0 = bn * BN;
  const int tx = bm % tiles_x;
  const int ty = (bm / tiles_x) % tiles_y;
  const int fb = bm / (tiles_x * tiles_y);
  const int y0 = ty * PT, x0 = tx * PT;

  // split_k > 1 (few output tiles: the 4 - 8 key frames of a video / streaming call give 32 - 64 tiles for 256 CUs): slice ks < split_k
  // accumulates the 16-channel chunks [ks * cps, (ks + 1) * cps) of phase 1, one extra slice (ks == split_k) the 1x1 second phase; all of them
  // store raw partial sums, splitk_epilogue_kernel adds them in slice order and applies bias / activation / the plane split (deterministic).
  const int spt_all = d.CinP / BK;
  const int sk = d.split_k > 1 ? d.split_k : 1;
  const bool p2_slice = sk > 1 && ks == sk;
  const int c_off = (sk > 1 && !p2_slice) ? ks * cps : 0;
  const int spt = p2_slice ? 0 : (sk > 1 ? min(cps, spt_all - c_off) : spt_all);
  const int n1 = 9 * spt;
  const int n2 = (d.in2_pl && (sk == 1 || p2_slice)) ? d.Cin2P / BK : 0;
  const int64_t Mpix = (int64_t)d.B * d.H * d.W;
  const int abl = VS_KERNEL_ABL(d);           // ablation (tools/bench_ppc.py): 32 no output stores

  // ------------------------------------------------------------------ DMA issue (waves 0-3: weights, waves 4-7: activations)
  // Every per-lane source OFFSET (32 bits, relative to a wave-uniform base that is advanced with scalar arithmetic) is computed once
  // (prologue / phase switch): address arithmetic on descriptor fields inside the K loop made hipcc re-load kernel arguments (s_load +
  // lgkmcnt(0), ~200 idle cycles) in front of every DMA, and 64-bit per-lane pointers did not fit the register budget (spilled pointers
  // are re-loaded from scratch: a VGPR-destination load -> vmcnt(0) -> the DMA pipeline drains).
  //   waves 0-3: doff[jj] = this lane's 16 bytes of weight block (wave, jj) inside a K step's WBLK * groups image; wbase walks the steps.
  //   waves 4-7: doff[j]  = this lane's 16 bytes of patch DMA slot j (instruction k = pw + 4j) inside chunk 0 of the planes; pvalid bit j
  //              clear = halo unit outside the image: never fetched -- it is zero-filled once in both patch buffers and the DMAs run
  //              with those lanes masked off.
  // The second phase sets both up again after the first one has drained (one ~1 us pipeline refill per launch buys a K loop without
  // any phase logic in it).
  const int64_t cstride = Mpix * 32;                            // bytes between 16-channel chunks of a plane
  const int pw = wave & 3;                    // index inside the DMA group
  const int g0 = n0 / 32;
  const int ngroups = (d.N + 31) / 32;
  unsigned doff[6];
  unsigned pvalid = 0;
  int wdst[TN];                               // LDS offset of weight block (wave, jj) inside a ring stage
#pragma unroll
  for (int jj = 0; jj < TN; ++jj) {
    const int idx = pw * TN + jj;
    const int p = idx / NG, gi = idx - p * NG;
    wdst[jj] = __builtin_amdgcn_readfirstlane(p * (BN * 32) + gi * 1024);
  }
  auto weight_off = [&](const int nst) __attribute__((always_inline)) {      // offsets inside tile 0 of a phase
#pragma unroll
    for (int jj = 0; jj < TN; ++jj) {
      const int idx = pw * TN + jj;
      const int p = idx / NG, gi = idx - p * NG;
      const int gs = (g0 + gi) < ngroups ? g0 + gi : ngroups - 1;       // tile wider than N: re-read a valid group (columns discarded)
      doff[jj] = (unsigned)(gs * nst) * WBLK + p * 1024 + lane * 16;
    }
  };
  const char* wbase = reinterpret_cast<const char*>(d.wt_blk) + (int64_t)c_off * 9 * WBLK;  // + WBLK per issued tile
  if (wave < 4) {
    weight_off(9 * spt_all);
  } else {
    const unsigned pstride = (unsigned)(spt_all * cstride);     // bytes between the two planes (< 4 GiB: checked by the launcher)
#pragma unroll
    for (int j = 0; j < 6; ++j) {             // unit U = 64k + lane of [plane][648 units]: pixel q of the patch, half hs
      const int U = (pw + 4 * j) * 64 + lane;
      const int plane = U >= PL_UNITS ? 1 : 0;
      const int u = U - plane * PL_UNITS;
      const int q = u >> 1;
      const int py = q / PPW, px = q - py * PPW;
      const int hs = (u & 1) ^ (py & 1);
      const int iy = y0 - 1 + py, ix = x0 - 1 + px;
      const bool ok = U < 2 * PL_UNITS && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
      doff[j] = ok ? plane * pstride + (unsigned)(((fb * d.H + iy) * d.W + ix) * 32 + hs * 16) : 0u;
      pvalid |= ok ? 1u << j : 0u;
      if (!ok && pw + 4 * j < P_DMAS) {       // zero the unit in both patch buffers, once
        *reinterpret_cast<u32x4*>(smem + (pw + 4 * j) * 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(smem + P_BYTES + (pw + 4 * j) * 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
      }
    }
  }
  int wstage = 0;                             // ring stage of the next weight tile (phase 2; phase 1 knows it at compile time)
  auto dma_w2 = [&]() __attribute__((always_inline)) {          // waves 0-3: the next weight tile of phase 2
    unsigned char* st = Bring + wstage * B_STAGE;
#pragma unroll
    for (int jj = 0; jj < TN; ++jj) dma16(wbase + doff[jj], st + wdst[jj]);
    wbase += WBLK;
    wstage = wstage + 1 == NRING ? 0 : wstage + 1;
  };
  auto dma_w = [&](auto stagec) __attribute__((always_inline)) {      // waves 0-3: the next weight tile of phase 1 into a known stage
    unsigned char* st = Bring + decltype(stagec)::value * B_STAGE;
#pragma unroll
    for (int jj = 0; jj < TN; ++jj) dma16(wbase + doff[jj], st + wdst[jj]);
    wbase += WBLK;
  };
  // patch of chunk cc, slots j0 and j0 + 1
  const char* const inpl = reinterpret_cast<const char*>(d.in_pl);
  auto dma_patch2 = [&](const int cc, const int j0) __attribute__((always_inline)) {
    unsigned char* pb = smem + (cc & 1) * P_BYTES + pw * 1024;
    const char* cbase = inpl + (c_off + cc) * cstride;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = j0 + jj;                  // compile-time after inlining
      if (pw + 4 * j < P_DMAS) {
        if ((pvalid >> j) & 1) dma16(cbase + doff[j], pb + j * 4096);      // halo lanes outside the image stay masked off (zeros in LDS)
      }
    }
  };
  // phase 2: in2 chunks as plain rows of the tile's 256 pixels; slots j = 0..3 (instruction k = pw + 4j), buffer c2 & 3
  auto qbuf = [&](const int c2) __attribute__((always_inline)) -> unsigned char* {
    const int b = c2 & 3;
    return smem + (b < 2 ? QX_OFF + b * Q_BYTES : (b - 2) * Q_BYTES);
  };
  const char* qbase = reinterpret_cast<const char*>(d.in2_pl);  // + cstride per issued chunk
  auto q_off = [&]() __attribute__((always_inline)) {
    const unsigned pstride2 = (unsigned)((d.Cin2P / BK) * cstride);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int U = (pw + 4 * j) * 64 + lane;
      const int plane = U >> 9;
      const int u = U & 511;
      const int p = u >> 1;
      const int hs = (u & 1) ^ ((p >> 3) & 1);
      const int iy = y0 + (p >> 4), ix = x0 + (p & 15);
      doff[j] = plane * pstride2 + (unsigned)(((fb * d.H + iy) * d.W + ix) * 32 + hs * 16);
    }
  };
  auto dma_q = [&](const int c2) __attribute__((always_inline)) {
    unsigned char* qb = qbuf(c2) + pw * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16(qbase + doff[j], qb + j * 4096);
    qbase += cstride;
  };

  // ------------------------------------------------------------------ consumers: all eight waves, 4 (pixels) x 2 (channels)
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  int q0[TM], p0[TM], prow[TM];               // patch pixel of tap (0,0) / tile pixel of this lane in block i / parity of its patch row
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    p0[i] = (wm * TM + i) * 32 + r;
    q0[i] = (p0[i] >> 4) * PPW + (p0[i] & 15);
    prow[i] = (p0[i] >> 4) & 1;
  }
  const int b_frag = (wn * TN) * 1024 + (2 * r + (g ^ ((r >> 3) & 1))) * 16;   // pack_blocked's bank swizzle

  // Fragment registers.  A step multiplies  lo_a x hi_b,  hi_a x lo_b,  hi_a x hi_b  (smallest partial products first: Arith<2>::PA / PB).
  // Only the hi fragments are double-buffered (the set of step s + 1 is read at the start of step s); the lo fragments of step s + 1
  // are read into the SAME registers as soon as the group of MFMAs that consumes them has been issued (lo_a after the first group,
  // lo_b after the second): 60 fragment registers instead of 80 -- with 96 accumulator registers the full double buffer did not
  // fit 256 VGPRs (hipcc then spilled the DMA offsets, and a scratch re-load is a VGPR-destination load: vmcnt(0) in the K loop).
  struct Hi { bf16x8 a[TM]; bf16x8 b[TN]; };
  Hi H0, H1;
  bf16x8 lo_a[TM], lo_b[TN];
  auto b_ptr = [&](const int s) __attribute__((always_inline)) -> const unsigned char* { return Bring + (s % NRING) * B_STAGE + b_frag; };
  auto load_all2 = [&](Hi& H, const int c2) __attribute__((always_inline)) {      // phase 2: plain rows of in2
    const unsigned char* Bb = b_ptr(c2);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      H.b[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 1024);
      lo_b[j] = *reinterpret_cast<const bf16x8*>(Bb + BN * 32 + j * 1024);
    }
    const unsigned char* Qs = qbuf(c2);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int ao = (2 * p0[i] + (g ^ ((p0[i] >> 3) & 1))) * 16;
      H.a[i] = *reinterpret_cast<const bf16x8*>(Qs + ao);
      lo_a[i] = *reinterpret_cast<const bf16x8*>(Qs + 512 * 16 + ao);
    }
  };
  // operands swapped (weights first): the accumulator holds C^T, see the epilogue
  auto mfma_q0 = [&](const Hi& H) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = AR::mfma(H.b[j], lo_a[i], acc[i][j]);
  };
  auto mfma_q1 = [&](const Hi& H) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = AR::mfma(lo_b[j], H.a[i], acc[i][j]);
  };
  auto mfma_q2 = [&](const Hi& H) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = AR::mfma(H.b[j], H.a[i], acc[i][j]);
  };
  // ---- phase 1.  The K loop is written per CHUNK with its nine taps unrolled and the chunk parity as a template argument: ring stage,
  // patch buffer, tap offset and swizzle parity of every fragment read are then compile-time constants (per-lane base + immediate), and
  // the DMA / wait schedule of a step is decided by the compiler -- the first version derived all of that from the step index at run
  // time (s / 9, tap / 3, s % 6 ...): 41 vector and 50 scalar instructions per step and wave next to 18 MFMAs (tools/pmc_pl.sh).
  // Weights: tile t is issued at step t - (NRING-1) into the stage tile t - NRING left at the previous barrier; at the end of step
  // s everything up to tile s + 2 has landed (tile s + 1 is being read during step s, tile s + 2 will be read during step s + 1
  // after one more barrier): NRING - 3 tiles = (NRING-3) * TN instructions of this wave stay in flight.
  // Patch of chunk cc + 1: issued at taps 0-2 of chunk cc (its buffer was last read before the barrier of (cc-1, tap 7)), waited
  // for at tap 7 (first read: the prefetch during tap 8).
  int ab[TM];                                 // hi-plane unit of tap (0,0) of this lane's pixel, swizzle bit for an even patch row offset
#pragma unroll
  for (int i = 0; i < TM; ++i) ab[i] = (2 * q0[i] + (g ^ prow[i])) * 16;
  auto ld_hi = [&](Hi& H, auto parc, auto tapc) __attribute__((always_inline)) {
    constexpr int PAR = decltype(parc)::value, TAP = decltype(tapc)::value;
    constexpr int ky = TAP / 3, kx = TAP % 3, STAGE = (PAR * 3 + TAP) % NRING;
    const unsigned char* Bb = Bring + STAGE * B_STAGE + b_frag;
#pragma unroll
    for (int j = 0; j < TN; ++j) H.b[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 1024);
    const unsigned char* Ps = smem + PAR * P_BYTES + (ky * PPW + kx) * 32;
#pragma unroll
    for (int i = 0; i < TM; ++i) H.a[i] = *reinterpret_cast<const bf16x8*>(Ps + ((ky & 1) ? (ab[i] ^ 16) : ab[i]));
  };
  auto ld_alo = [&](auto parc, auto tapc) __attribute__((always_inline)) {
    constexpr int PAR = decltype(parc)::value, TAP = decltype(tapc)::value;
    constexpr int ky = TAP / 3, kx = TAP % 3;
    const unsigned char* Ps = smem + PAR * P_BYTES + PL_UNITS * 16 + (ky * PPW + kx) * 32;
#pragma unroll
    for (int i = 0; i < TM; ++i) lo_a[i] = *reinterpret_cast<const bf16x8*>(Ps + ((ky & 1) ? (ab[i] ^ 16) : ab[i]));
  };
  auto ld_blo = [&](auto parc, auto tapc) __attribute__((always_inline)) {
    constexpr int PAR = decltype(parc)::value, TAP = decltype(tapc)::value;
    constexpr int STAGE = (PAR * 3 + TAP) % NRING;
    const unsigned char* Bb = Bring + STAGE * B_STAGE + BN * 32 + b_frag;
#pragma unroll
    for (int j = 0; j < TN; ++j) lo_b[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 1024);
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  // one step: chunk cc (parity PAR), tap TAP; `last`: cc is the last chunk of the phase
  auto tapstep = [&](auto parc, auto tapc, const int cc, const bool last) __attribute__((always_inline)) {
    constexpr int PAR = decltype(parc)::value, TAP = decltype(tapc)::value;
    constexpr bool ODD = ((PAR + TAP) & 1) != 0;              // step index cc * 9 + TAP is odd
    const Hi& Hc = ODD ? H1 : H0;
    Hi& Hn = ODD ? H0 : H1;
    const int s = cc * 9 + TAP;
    // the first group of MFMAs goes out right behind the barrier (its operands are in registers): DMA issue and fragment reads run in its shadow
    mfma_q0(Hc);
    __builtin_amdgcn_sched_barrier(0);
    if (wave < 4) {
      if (s + NRING - 1 < n1) dma_w(std::integral_constant<int, (PAR * 3 + TAP + NRING - 1) % NRING>{});
    } else if constexpr (TAP < 3) {
      if (!last) dma_patch2(cc + 1, 2 * TAP);
    }
    if constexpr (TAP < 8) {
      ld_hi(Hn, parc, std::integral_constant<int, TAP + 1>{});
      ld_alo(parc, std::integral_constant<int, TAP + 1>{});       // lo_a is dead: its six MFMAs have been issued
    } else if (!last) {
      ld_hi(Hn, std::integral_constant<int, 1 - PAR>{}, P0{});
      ld_alo(std::integral_constant<int, 1 - PAR>{}, P0{});
    }
    mfma_q1(Hc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TAP < 8) ld_blo(parc, std::integral_constant<int, TAP + 1>{});
    else if (!last) ld_blo(std::integral_constant<int, 1 - PAR>{}, P0{});
    mfma_q2(Hc);
    if (wave < 4) {
      if (s + NRING - 1 < n1) wait_vm<(NRING - 3) * TN>();
      else wait_vm<0>();
    } else if constexpr (TAP == 7) {
      if (!last) wait_vm<0>();                // the next patch (nothing else of this wave is in flight)
    }
    // (the builtin, not an asm statement: hipcc's own waitcnt bookkeeping must see it, otherwise it waits lgkmcnt(0) in front of the next
    // step's first MFMA -- after that step's prefetch reads have been issued)
    __builtin_amdgcn_s_waitcnt(0xC07F);       // this wave's fragment reads have returned: the stages they came from may be refilled
    __builtin_amdgcn_s_barrier();
    // (a barrier per TWO steps -- legal with this ring: the stage a step refills was last read two steps earlier, waits for two tiles in
    // flight instead of three -- measured identical, 0.1828 vs 0.1832 ms: the barrier is not what the K loop waits for)
  };
  auto chunk = [&](auto parc, const int cc) __attribute__((always_inline)) {
    const bool last = cc + 1 >= spt;
    tapstep(parc, std::integral_constant<int, 0>{}, cc, last);
    tapstep(parc, std::integral_constant<int, 1>{}, cc, last);
    tapstep(parc, std::integral_constant<int, 2>{}, cc, last);
    tapstep(parc, std::integral_constant<int, 3>{}, cc, last);
    tapstep(parc, std::integral_constant<int, 4>{}, cc, last);
    tapstep(parc, std::integral_constant<int, 5>{}, cc, last);
    tapstep(parc, std::integral_constant<int, 6>{}, cc, last);
    tapstep(parc, std::integral_constant<int, 7>{}, cc, last);
    tapstep(parc, std::integral_constant<int, 8>{}, cc, last);
  };

  if (spt > 0) {
  // ---- prologue: weight tiles 0 .. NRING-2, patch of chunk 0
  if (wave < 4) {
    dma_w(std::integral_constant<int, 0>{});                 // (n1 >= 9 > NRING - 1)
    dma_w(std::integral_constant<int, 1>{});
    dma_w(std::integral_constant<int, 2>{});
    dma_w(std::integral_constant<int, 3>{});
    dma_w(std::integral_constant<int, 4>{});
  } else {
    dma_patch2(0, 0);
    dma_patch2(0, 2);
    dma_patch2(0, 4);
  }
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();

  ld_hi(H0, P0{}, P0{});
  ld_alo(P0{}, P0{});
  ld_blo(P0{}, P0{});
  __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0): otherwise hipcc drains the PREFETCH of every iteration in front of its first MFMA
  for (int cc = 0; cc < spt; cc += 2) {
    chunk(P0{}, cc);
    if (cc + 1 < spt) chunk(P1{}, cc + 1);
  }
  }          // (spt > 0: the phase-2 slice of a split launch has no first phase)
  // ---- epilogue state.  Operands are swapped in the MFMA (weights first): element e of acc[i][j] is
  //   channel cbase[j] + (e & 3) + 8 * (e >> 2)   (cbase contains the half-wave's 4 * g),  pixel p0[i]
  // (everything below is re-derived from an opaque copy of the lane id: values shared with the K loop's address set-up would otherwise be
  // kept alive across the loop, and the loop has no registers to spare)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int r_e = lane_e & 31, g_e = lane_e >> 5;
  int cbase[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) cbase[j] = n0 + (wn * TN + j) * 32 + 4 * g_e;
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
  int64_t mrow[TM];
  int pcls[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int pe = (wm * TM + i) * 32 + r_e;
    const int y = y0 + (pe >> 4), x = x0 + (pe & 15);               // tiles are whole: every pixel lies inside the image
    mrow[i] = ((int64_t)fb * d.H + y) * d.W + x;
    pcls[i] = (y == 0 ? 0 : (y >= d.H - 1 ? 2 : 1)) * 3 + (x == 0 ? 0 : (x >= d.W - 1 ? 2 : 1));
  }
  scale_all<TM, TN>(acc, d.acc_mul);          // back to real units (exact: a power of two)
  // Whole-width tiles (every layer of the shipped cards) take straight-line forms of the two epilogue sections: the general forms guard
  // every 4-channel group against N / n_store per lane, and across those divergent regions hipcc waits vmcnt(0) in front of every use --
  // each bias / table / residual fetch became its own dependent round trip (12 + 24 of them per wave, with one workgroup per CU nothing
  // else covers them), and since stores count in vmcnt too, each one also waited for the previous store's acknowledge.
  const bool full = n0 + BN <= d.N;
  if (sk == 1) {   // v = act(acc (+ border-class table) + bias1) (+ bias2: the activation sits between the two K phases)
    const float* tb = (d.tile_hint & VS_CONV_PRE) ? d.a_scale + (int64_t)fb * d.a_scale_ld : nullptr;
    if (full && d.bias && (n2 == 0 || d.bias2)) {
      // three passes over the register tile -- + bias1 (+ table), activation, + bias2 -- with the switch on d.act around the whole second
      // pass: a switch per element made the section ~100 tiny basic blocks (hipcc's wait-count bookkeeping gives up across them)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        f32x4 b1[4], tv[TM][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b1[q] = *reinterpret_cast<const f32x4*>(d.bias + cbase[j] + 8 * q);
        if (tb) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) tv[i][q] = *reinterpret_cast<const f32x4*>(tb + pcls[i] * d.N + cbase[j] + 8 * q);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            f32x4 v = grp(i, j, q) + b1[q];
            if (tb) v += tv[i][q];
            put(i, j, q, v);
          }
        __builtin_amdgcn_sched_barrier(0);      // (one channel group at a time: scheduled as one region the loads of all TN groups are hoisted and spill)
      }
      auto act_tile = [&](auto actc) __attribute__((always_inline)) {
        constexpr int ACT = decltype(actc)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = vs_apply_act(acc[i][j][e], ACT < 0 ? d.act : ACT);
      };
      if (d.act == VS_ACT_RELU) act_tile(std::integral_constant<int, VS_ACT_RELU>{});
      else if (d.act != VS_ACT_NONE) act_tile(std::integral_constant<int, -1>{});
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        f32x4 b2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b2[q] = n2 > 0 ? *reinterpret_cast<const f32x4*>(d.bias2 + cbase[j] + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < TM; ++i) put(i, j, q, grp(i, j, q) + b2[q]);
      }
    } else {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = cbase[j] + 8 * q;
        const f32x4 b1 = ld4(d.bias, n, d.N);
        const f32x4 b2 = n2 > 0 ? ld4(d.bias2, n, d.N) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          f32x4 v = grp(i, j, q) + b1;
          if (tb) v += ld4(tb + pcls[i] * d.N, n, d.N);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = vs_apply_act(v[e], d.act);
          put(i, j, q, v + b2);
        }
      }
    }
  }
  if (n2 > 0) {                               // phase 2 on top: its own pipeline fill, no fragment prefetch across the (few) steps
    scale_all<TM, TN>(acc, 1.f / d.acc_mul2);
    // chunk c2 is issued three steps ahead (four 16 KiB buffers), weight tile c2 NRING - 1 steps ahead, as in phase 1
    wstage = 0;
    wbase = reinterpret_cast<const char*>(d.wt2_blk);
    if (wave < 4) {
      weight_off(n2);
      for (int t = 0; t < NRING - 1 && t < n2; ++t) dma_w2();
    } else {
      q_off();
      for (int c = 0; c < 3 && c < n2; ++c) dma_q(c);
    }
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    for (int c2 = 0; c2 < n2; ++c2) {
      if (wave < 4) {
        if (c2 + NRING - 1 < n2) dma_w2();
      } else if (c2 + 3 < n2) {
        dma_q(c2 + 3);
      }
      load_all2(H0, c2);
      mfma_q0(H0);
      mfma_q1(H0);
      mfma_q2(H0);
      if (wave < 4) {
        if (c2 + NRING - 1 < n2) wait_vm<(NRING - 3) * TN>();
        else wait_vm<0>();
      } else {
        const int allow = min(n2 - 1, c2 + 3) - (c2 + 1);     // chunks younger than the one the next step reads
        if (allow >= 2) wait_vm<8>();
        else if (allow == 1) wait_vm<4>();
        else wait_vm<0>();
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
    }
    scale_all<TM, TN>(acc, d.acc_mul2);
  }
  if (abl & 32) return;
  if (sk > 1) {      // raw partial sums of this slice -> workspace [slice][pixel][splitk_ld]
    float* ws = d.splitk_ws + (int64_t)ks * Mpix * d.splitk_ld;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = cbase[j] + 8 * q;
          if (n < d.N) *reinterpret_cast<f32x4*>(ws + mrow[i] * d.splitk_ld + n) = grp(i, j, q);       // (N % 16 == 0 with planes operands)
        }
    return;
  }
  auto store_all = [&](auto fullc) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(fullc)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float* orow = d.out ? d.out + mrow[i] * d.out_ld + d.out_coff : nullptr;
      const float* rrow = d.res ? d.res + mrow[i] * d.res_ld : nullptr;
      f32x4 rv[TN][4];                            // FULL: the residual groups of this pixel block in flight at once
      if constexpr (FULL) {
        if (rrow) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) rv[j][q] = *reinterpret_cast<const f32x4*>(rrow + cbase[j] + 8 * q);
        }
      }
      char* prow = d.out_pl ? reinterpret_cast<char*>(d.out_pl) + mrow[i] * 32 : nullptr;
      const int64_t opstride = (int64_t)(d.N / BK) * cstride;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {           // the two channel groups q = 2 q2, 2 q2 + 1 of one 16-channel chunk
          f32x4 v[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int n = cbase[j] + 8 * (2 * q2 + h);
            if constexpr (FULL) {
              v[h] = grp(i, j, 2 * q2 + h);
              if (rrow) v[h] += rv[j][2 * q2 + h];
            } else {
              v[h] = f32x4{0.f, 0.f, 0.f, 0.f};      // columns in [N, n_store) are written as zeros
              if (n >= d.n_store) continue;
              if (n < d.N) {
                v[h] = grp(i, j, 2 * q2 + h);
                if (rrow) v[h] += *reinterpret_cast<const f32x4*>(rrow + n);
                if (n + 4 > d.N)
#pragma unroll
                  for (int e = 0; e < 4; ++e)
                    if (n + e >= d.N) v[h][e] = 0.f;
              }
            }
            if (orow) *reinterpret_cast<f32x4*>(orow + n) = v[h];
          }
          // the next conv's operand planes: hi / lo f16 of v * a_mul, [plane][n / 16][pixel][16].  A lane holds channels 4g .. 4g+3 and
          // 8 + 4g .. 8 + 4g+3 of its pixel; v_permlane32_swap trades the middle pieces between the half-waves (same pixel, g = 0 / 1) so
          // that g = 0 writes channels 0..7 and g = 1 channels 8..15 as ONE 16-byte store each: a tile row becomes 512 contiguous bytes
          const int n16 = n0 + (wn * TN + j) * 32 + 16 * q2;          // wave-uniform (N % 16 == 0 with planes output)
          if (prow && (FULL || n16 < d.N)) {
            u32x2 h0, l0, h1, l1;
            split4h(v[0], d.a_mul, h0, l0);
            split4h(v[1], d.a_mul, h1, l1);
            const auto sh0 = __builtin_amdgcn_permlane32_swap(h0[0], h1[0], false, false);
            const auto sh1 = __builtin_amdgcn_permlane32_swap(h0[1], h1[1], false, false);
            const auto sl0 = __builtin_amdgcn_permlane32_swap(l0[0], l1[0], false, false);
            const auto sl1 = __builtin_amdgcn_permlane32_swap(l0[1], l1[1], false, false);
            char* dst = prow + (int64_t)(n16 >> 4) * cstride + g_e * 16;
            *reinterpret_cast<u32x4*>(dst) = u32x4{sh0[0], sh1[0], sh0[1], sh1[1]};
            *reinterpret_cast<u32x4*>(dst + opstride) = u32x4{sl0[0], sl1[0], sl0[1], sl1[1]};
          }
        }
    }
  };
  if (full) store_all(std::true_type{});
  else store_all(std::false_type{});
}

// fp32 NHWC rows [rows][ld] -> operand planes [2][C/16][rows][16] f16 of x * a_mul (hi, lo): the entry of a planes chain
__global__ __launch_bounds__(256) void to_planes_kernel(const float* __restrict__ x, const int64_t rows, const int C, const int64_t ld,
                                                        const float a_mul, char* __restrict__ pl) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;     // (chunk, row, half): half fastest -> 32 contiguous bytes per row
  const int64_t per_chunk = rows * 2;
  const int chunks = C / 16;
  if (idx >= per_chunk * chunks) return;
  const int c = (int)(idx / per_chunk);
  const int64_t rem = idx - (int64_t)c * per_chunk;
  const int64_t row = rem >> 1;
  const int h = (int)(rem & 1);
  const float* src = x + row * ld + c * 16 + h * 8;
  const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
  u32x2 ah, al, bh, bl;
  split4h(a, a_mul, ah, al);
  split4h(b, a_mul, bh, bl);
  char* dst = pl + ((int64_t)c * rows + row) * 32 + h * 16;
  *reinterpret_cast<u32x4*>(dst) = u32x4{ah[0], ah[1], bh[0], bh[1]};
  *reinterpret_cast<u32x4*>(dst + (int64_t)chunks * rows * 32) = u32x4{al[0], al[1], bl[0], bl[1]};
}

template <int TN>
int launch_pl(const vs_conv_desc_t& d, hipStream_t st) {
  constexpr int BN = 2 * TN * 32;
  const int tiles_x = d.W / PT, tiles_y = d.H / PT;
  const int sk = d.split_k > 1 ? d.split_k : 1;
  const int64_t mt = (int64_t)d.B * tiles_x * tiles_y, nt = cdiv64(sk > 1 ? d.N : d.n_store, BN);
  const int spt = d.CinP / BK;
  const int cps = (spt + sk - 1) / sk;
  if (sk > 1 && (int64_t)(sk - 1) * cps >= spt) return VS_ERR_BAD_ARG;       // an empty K slice
  const int slices = sk > 1 ? sk + (d.in2_pl ? 1 : 0) : 1;
  if (mt * nt * slices > 0x7fffffffLL) return VS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((conv3x3_pl_kernel<TN>), dim3((unsigned)(mt * nt * slices)), dim3(512), 0, st, d, tiles_x, tiles_y, (int)mt, (int)nt, cps);
  int rc = vs_launch_status();
  if (rc != VS_OK || sk == 1) return rc;
  return vs_splitk_epilogue(d, (int)((int64_t)d.B * d.H * d.W), st);
}

}  // namespace

// tile 22 = 256 pixels x 192 channels, tile 23 = 256 pixels x 128 channels.  Preconditions are checked by vs_conv_gemm.
int vs_conv3x3_pl_dispatch(const vs_conv_desc_t& d, int tile, hipStream_t st) {
  switch (tile) {
    case 22: return launch_pl<3>(d, st);
    case 23: return launch_pl<2>(d, st);
    default: return VS_ERR_UNSUPPORTED;
  }
}

extern "C" int vs_to_planes(const float* x, int64_t rows, int C, int64_t ld, float a_mul, void* planes, void* stream) {
  VS_REQUIRE(x && planes && rows > 0 && C > 0 && C % 16 == 0 && ld >= C && ld % 4 == 0 && a_mul > 0.f);
  VS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)planes & 15) == 0);
  const int64_t items = rows * 2 * (C / 16);
  hipLaunchKernelGGL(to_planes_kernel, dim3((unsigned)cdiv64(items, 256)), dim3(256), 0, (hipStream_t)stream, x, rows, C, ld, a_mul,
                     static_cast<char*>(planes));
  return vs_launch_status();
}
