// 1x1 convolution / linear layer on PRE-SPLIT operand planes (arith 2, "2 x f16"): the ConvNeXt pwconv1 / pwconv2 GEMMs of the
// extractor (convnext.py:47-51) and ChunkySeal's wide layers, in the structure of conv3x3_pl.hip:
//   out[m][n] = epilogue( sum_k A[m][k] * W[n][k] ),   A as two f16 planes [plane][K/16][M][16] of value * a_mul (vs_to_planes), W as the
//   blocked f16 image of engine.pack_blocked -- every operand moves global -> LDS by LDS-DMA, all EIGHT waves multiply (4 x 2 waves of
//   64 rows x 32*TN columns on a 256-row x 64*TN-column tile), waves 0-3 issue the weight DMAs (TN per K16 step), waves 4-7 the
//   activation DMAs (four 1 KiB pieces per step: a 256-row x 16-k chunk is 8 KiB of contiguous memory per plane).
// Rings: five 16 KiB A stages (two steps ahead of the reads stay in flight), six weight stages (three tiles in flight).  Fragment
// registers as in conv3x3_pl.hip (hi fragments double-buffered, lo fragments re-read in place).  Same products and K order as
// conv_gemm_kernel / gemm1x1_pc_kernel in the 2 x f16 arithmetic: bit-identical to them when K is not split.
// Epilogue as gemm1x1_pc_kernel: bias, activation, residual, GRN sum-of-squares partials (common.py:166), or raw K-slice partial sums
// for the shared split-K epilogue (small-M layers).
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"
#include "lds_dma.h"

int vs_splitk_epilogue(const vs_conv_desc_t& d, int M, hipStream_t st);   // gemm1x1_pc.hip

namespace {

using namespace vsconv;

constexpr int GBM = 256;                       // rows per tile
constexpr int A_STAGE = GBM * 2 * 2 * 16;      // 256 rows x 2 halves x 2 planes x 16 bytes
constexpr int NRA = 5, NRB = 6;

__device__ __forceinline__ void gdma16(const char* gp, unsigned char* lds_base) { vs_lds_dma16(gp, lds_base); }      // lds_dma.h: lane l: 16 bytes at gp -> lds_base + 16 * l
// a wave-uniform pointer the compiler has lost track of (loop-carried through uniform branches): back into SGPRs, so that the DMA uses the
// scalar-base + 32-bit-VGPR-offset form instead of 64-bit per-lane addresses
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}
template <int N>
__device__ __forceinline__ void gwait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int TN>
__global__ __launch_bounds__(512, 2) void gemm_pl_kernel(const vs_conv_desc_t d, const int M, const int mtiles, const int ntiles, const int sps,
                                                         const int per_xcd) {
  using AR = Arith<2>;
  constexpr int TM = 2;
  constexpr int BN = 2 * TN * 32;
  constexpr int NG = BN / 32;
  constexpr int WBLK = 2048;                  // one (32 rows x 16 k) weight block, both planes
  constexpr int B_STAGE = NG * WBLK;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NRA * A_STAGE + NRB * B_STAGE];
  unsigned char* const Bring = smem + NRA * A_STAGE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, g = lane >> 5;

  // Tile order.  per_xcd == 0: row tiles fastest (the order until round 5; right for the one-round launches of VideoSeal's stage 2 / 3).
  // per_xcd > 0 (many rounds per launch: ChunkySeal's 60 x 31 tiles): workgroup b runs on XCD b % 8, so the b-th workgroup OF an XCD takes
  // the b-th tile of that XCD's contiguous range of a grouped order -- four row tiles x all column tiles, column-major inside the group --
  // and the 32 workgroups an XCD runs at a time cover 4 row tiles x 8 column tiles: 4 x 1.5 MB of activations + 8 x 1.1 MB of weights
  // through its L2 instead of 32 row tiles + 1 column tile (49 MB) per round.  Same tiles, same arithmetic.
  int bm, bn, ks;
  if (per_xcd > 0) {
    const int total = mtiles * ntiles;
    const int slot = blockIdx.x % (8 * per_xcd);
    ks = blockIdx.x / (8 * per_xcd);
    const int v = (slot & 7) * per_xcd + (slot >> 3);
    if (v >= total) return;                               // (the grid is padded to 8 x per_xcd slots per K slice)
    constexpr int GROUP_M = 4;
    const int width = GROUP_M * ntiles;
    const int group = v / width, rem = v - group * width;
    const int first_m = group * GROUP_M;
    const int gsz = min(mtiles - first_m, GROUP_M);
    bm = first_m + rem % gsz;
    bn = rem / gsz;
  } else {
    bm = blockIdx.x % mtiles;
    bn = (blockIdx.x / mtiles) % ntiles;
    ks = blockIdx.x / (mtiles * ntiles);                  // K slice (split_k > 1)
  }
  const int m0 = bm * GBM;
  const int n0 = bn * BN;
  const int nsteps_all = d.CinP / BK;
  const int step0 = ks * sps;
  const int nsteps = min(sps, nsteps_all - step0);      // K16 steps of this workgroup

  // ------------------------------------------------------------------ DMA issue: per-lane 32-bit offsets, wave-uniform bases
  const int64_t cstride = (int64_t)M * 32;              // bytes between 16-channel chunks of a plane
  const int pw = wave & 3;
  unsigned doff[4];
  int wdst[TN];
  // (bases and ring stages are recomputed from the step index: loop-carried pointers / counters updated inside the wave-role branches
  // ended up in VGPRs -- and from there in scratch, whose re-loads are VGPR-destination loads: vmcnt(0) in the K loop)
  const char* const wbase = reinterpret_cast<const char*>(d.wt_blk) + (int64_t)step0 * WBLK;
  const char* const abase = reinterpret_cast<const char*>(d.in_pl) + (int64_t)step0 * cstride;
  if (wave < 4) {
    const int g0 = n0 / 32, ngroups = (d.N + 31) / 32;
#pragma unroll
    for (int jj = 0; jj < TN; ++jj) {
      const int idx = pw * TN + jj;
      const int p = idx / NG, gi = idx - p * NG;
      const int gs = (g0 + gi) < ngroups ? g0 + gi : ngroups - 1;       // tile wider than N: re-read a valid group (columns discarded)
      doff[jj] = (unsigned)(gs * nsteps_all) * WBLK + p * 1024 + lane * 16;
      wdst[jj] = __builtin_amdgcn_readfirstlane(p * (BN * 32) + gi * 1024);
    }
  } else {
    const unsigned pstride = (unsigned)(nsteps_all * cstride);          // bytes between the two planes (< 4 GiB: checked by the launcher)
#pragma unroll
    for (int j = 0; j < 4; ++j) {             // unit U = 64 (pw + 4j) + lane of [plane][256 rows][2 halves]
      const int U = (pw + 4 * j) * 64 + lane;
      const int plane = U >> 9;
      const int u = U & 511;
      const int p = u >> 1;
      const int hs = (u & 1) ^ ((p >> 3) & 1);
      const int m = min(m0 + p, M - 1);       // ragged last tile: re-read a valid row (masked in the epilogue)
      doff[j] = plane * pstride + (unsigned)m * 32u + hs * 16;
    }
#pragma unroll
    for (int jj = 0; jj < TN; ++jj) wdst[jj] = 0;
  }
  auto dma_w = [&](const int t) __attribute__((always_inline)) {          // weight tile of step t
    unsigned char* st = Bring + (t % NRB) * B_STAGE;
    const char* wb = wbase + (int64_t)t * WBLK;
#pragma unroll
    for (int jj = 0; jj < TN; ++jj) gdma16(wb + doff[jj], st + wdst[jj]);
  };
  auto dma_a = [&](const int t) __attribute__((always_inline)) {          // activation chunk of step t
    unsigned char* st = smem + (t % NRA) * A_STAGE + pw * 1024;
    const char* ab = abase + t * cstride;
#pragma unroll
    for (int j = 0; j < 4; ++j) gdma16(ab + doff[j], st + j * 4096);
  };

  // ------------------------------------------------------------------ consumers: 4 (rows) x 2 (columns) waves
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  int a_frag[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int p = (wm * TM + i) * 32 + r;
    a_frag[i] = (2 * p + (g ^ ((p >> 3) & 1))) * 16;
  }
  const int b_frag = (wn * TN) * 1024 + (2 * r + (g ^ ((r >> 3) & 1))) * 16;   // pack_blocked's bank swizzle

  struct Hi { bf16x8 a[TM]; bf16x8 b[TN]; };
  Hi H0, H1;
  bf16x8 lo_a[TM], lo_b[TN];
  auto b_ptr = [&](const int s) __attribute__((always_inline)) -> const unsigned char* { return Bring + (s % NRB) * B_STAGE + b_frag; };
  auto a_ptr = [&](const int s) __attribute__((always_inline)) -> const unsigned char* { return smem + (s % NRA) * A_STAGE; };
  auto load_hi = [&](Hi& H, const int s) __attribute__((always_inline)) {
    const unsigned char* Bb = b_ptr(s);
#pragma unroll
    for (int j = 0; j < TN; ++j) H.b[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 1024);
    const unsigned char* Ab = a_ptr(s);
#pragma unroll
    for (int i = 0; i < TM; ++i) H.a[i] = *reinterpret_cast<const bf16x8*>(Ab + a_frag[i]);
  };
  auto load_alo = [&](const int s) __attribute__((always_inline)) {
    const unsigned char* Ab = a_ptr(s) + 512 * 16;
#pragma unroll
    for (int i = 0; i < TM; ++i) lo_a[i] = *reinterpret_cast<const bf16x8*>(Ab + a_frag[i]);
  };
  auto load_blo = [&](const int s) __attribute__((always_inline)) {
    const unsigned char* Bb = b_ptr(s) + BN * 32;
#pragma unroll
    for (int j = 0; j < TN; ++j) lo_b[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 1024);
  };
  // accumulator = C[row][column] (activations first): column = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
  auto mfma_q0 = [&](const Hi& H) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = AR::mfma(lo_a[i], H.b[j], acc[i][j]);
  };
  auto mfma_q1 = [&](const Hi& H) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = AR::mfma(H.a[i], lo_b[j], acc[i][j]);
  };
  auto mfma_q2 = [&](const Hi& H) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = AR::mfma(H.a[i], H.b[j], acc[i][j]);
  };
  // Weights: tile t issued at step t - (NRB-1); at the end of step s everything up to tile s + 2 has landed: NRB - 3 tiles stay in flight.
  // Activations: chunk t issued at step t - (NRA-1) into the stage chunk t - NRA left two barriers ago; at the end of step s chunk
  // s + 2 has landed: NRA - 3 chunks (4 instructions each) stay in flight.
  auto issue = [&](const int s) __attribute__((always_inline)) {
    if (wave < 4) {
      if (s + NRB - 1 < nsteps) dma_w(s + NRB - 1);
    } else if (s + NRA - 1 < nsteps) {
      dma_a(s + NRA - 1);
    }
  };
  auto finish = [&](const int s) __attribute__((always_inline)) {
    if (wave < 4) {
      if (s + NRB - 1 < nsteps) gwait_vm<(NRB - 3) * TN>();
      else gwait_vm<0>();
    } else {
      if (s + NRA - 1 < nsteps) gwait_vm<(NRA - 3) * 4>();
      else gwait_vm<0>();
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0) (the builtin: hipcc's waitcnt bookkeeping must see it, see conv3x3_pl.hip)
    __builtin_amdgcn_s_barrier();
  };

  if (wave < 4) {
    for (int t = 0; t < NRB - 1 && t < nsteps; ++t) dma_w(t);
  } else {
    for (int t = 0; t < NRA - 1 && t < nsteps; ++t) dma_a(t);
  }
  gwait_vm<0>();
  __builtin_amdgcn_s_barrier();

  int s = 0;
  load_hi(H0, 0);
  load_alo(0);
  load_blo(0);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  auto step = [&](const Hi& Hc, Hi& Hn, const int s, auto more) __attribute__((always_inline)) {
    constexpr bool MORE = decltype(more)::value;
    mfma_q0(Hc);                              // (first: the DMA issue and the next fragments' addresses run in its shadow)
    __builtin_amdgcn_sched_barrier(0);
    issue(s);
    if constexpr (MORE) {
      load_hi(Hn, s + 1);
      load_alo(s + 1);
    }
    mfma_q1(Hc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MORE) load_blo(s + 1);
    mfma_q2(Hc);
    finish(s);
  };
  for (; s + 2 < nsteps; s += 2) {
    step(H0, H1, s, std::true_type{});
    step(H1, H0, s + 1, std::true_type{});
  }
  if (s + 1 < nsteps) {
    step(H0, H1, s, std::true_type{});
    step(H1, H0, s + 1, std::false_type{});
  } else {
    step(H0, H1, s, std::false_type{});
  }

  // ---- epilogue (as gemm1x1_pc_kernel)
  scale_all<TM, TN>(acc, d.acc_mul);          // back to real units (exact: a power of two), also for the K-slice partial sums
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));            // (nothing of the epilogue's addressing stays alive across the K loop)
  const int r_e = lane_e & 31, g_e = lane_e >> 5;
  int col[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) col[j] = n0 + (wn * TN + j) * 32 + r_e;
  // ONE store path for the K-slice partial sums and the final output (round 6): with the two inlined copies of store_tile_full next to each other
  // hipcc spilled 22 registers of the TN = 3 tile (either copy alone: 0; conv_common.h::apply_act_all for the other 45 of the 67)
  const bool sk = d.split_k > 1;
  float* const wsb = sk ? d.splitk_ws + (int64_t)ks * M * d.splitk_ld : nullptr;
  const int o_ld = sk ? (int)d.splitk_ld : (int)d.out_ld;
  const bool whole = m0 + GBM <= M && n0 + BN <= d.N && (int64_t)GBM * max((int64_t)o_ld, sk ? (int64_t)0 : d.res_ld) * 4 < (1LL << 31);
  if (!sk) {
  float bias1[TN], zero[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    bias1[j] = (col[j] < d.N && d.bias) ? d.bias[col[j]] : 0.f;
    zero[j] = 0.f;
  }
  const int abl = VS_KERNEL_ABL(d);           // ablation build only (tools/bench_gemm.py ksweep2): 64 no activation, 32 no output stores
  if (!(abl & 64)) apply_act_all<TM, TN, (TN < 3)>(acc, bias1, zero, d.act);
  if (d.sumsq_part) {
    if (d.sumsq_hw > 0) write_sumsq_straddle<TM, TN>(acc, d.sumsq_part, d.N, (int64_t)m0 + (int64_t)wm * TM * 32, M, col, g_e, d.sumsq_hw);
    else write_sumsq<TM, TN>(acc, d.sumsq_part, d.N, (int64_t)m0 + (int64_t)wm * TM * 32, M, col, g_e);
  }
    if (abl & 32) return;
  }
  if (whole) {       // every layer of the shipped cards: store_tile_full (conv_common.h)
    const int64_t row0 = (int64_t)m0 + wm * TM * 32;                                   // wave-uniform
    char* const ob = sk ? reinterpret_cast<char*>(wsb + row0 * d.splitk_ld + n0 + wn * TN * 32)
                        : reinterpret_cast<char*>(d.out + row0 * d.out_ld + d.out_coff + n0 + wn * TN * 32);
    const char* const rb = (!sk && d.res) ? reinterpret_cast<const char*>(d.res + row0 * d.res_ld + n0 + wn * TN * 32) : nullptr;
    store_tile_full<TM, TN>(acc, ob, o_ld, rb, (int)d.res_ld, r_e, g_e);
    return;
  }
  if (sk) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * g_e;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j)
          if (col[j] < d.N) wsb[(int64_t)m * d.splitk_ld + col[j]] = acc[i][j][e];
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int64_t m = (int64_t)m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * g_e;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = col[j];
        if (n >= d.n_store) continue;
        float v = 0.f;                        // columns in [N, n_store) are padding lanes: always zero
        if (n < d.N) {
          v = acc[i][j][e];
          if (d.res) v += d.res[m * d.res_ld + n];
        }
        d.out[m * d.out_ld + d.out_coff + n] = v;
      }
    }
  }
}

// ---- round 6: 256-row x 384-column tiles, ONE wave per SIMD (tile code 27) ---------------------------------------------------------------
// gemm_pl_kernel moves 28 KiB global -> LDS per K16 step and workgroup for 4.7 MFLOP; measured 0.92 us per step against 0.55 us of matrix time:
// the chip-wide LDS-DMA stream (6.4-6.8 TB/s, MI355X_MICROARCH.md) is the bound, not the matrix cores (LAB_NOTEBOOK round 5).  A 256 x 384 tile
// moves 40 KiB for 9.4 MFLOP -- 30 % fewer operand bytes per FLOP.  Its 96 accumulator registers per 32 x 32 block row do not fit two waves per
// SIMD: four waves (2 x 2, each 128 rows x 192 columns = 24 blocks = 384 accumulator registers, upper half of the register file) run one per
// SIMD, so nothing but the wave's own schedule hides its LDS latencies: the four fragment banks (lo_a 16 | hi_b 24 | hi_a 16 | lo_b 24 registers)
// rotate -- the fragments of step s + 1 are read into the bank whose last use in step s has issued, one 24-MFMA group (~770 cycles) ahead of
// their first use.  Four 40 KiB stages (the whole 160 KiB of LDS): tile s + 3 is requested during step s, tile s + 2 has landed at its end.
// Same products in the same order per accumulator as gemm_pl_kernel (q0 = lo_a x hi_b, q1 = hi_a x lo_b, q2 = hi_a x hi_b per K16 step):
// bit-identical outputs (tests/test_gpu_kernels.py::test_gemm_planes_big_tile).
constexpr int BIG_NS = 4;
template <int TN>
__global__ __launch_bounds__(256, 1) void gemm_pl_big_kernel(const vs_conv_desc_t d, const int M, const int mtiles, const int ntiles, const int sps,
                                                            const int per_xcd) {
  using AR = Arith<2>;
  constexpr int TM = 4;
  constexpr int BN = 2 * TN * 32;                // 384 (TN = 6), 256 (TN = 4)
  constexpr int NG = BN / 32;
  constexpr int WBLK = 2048;
  constexpr int B_STAGE = NG * WBLK;             // 24 KiB
  constexpr int STAGE = A_STAGE + B_STAGE;       // 40 KiB
  __shared__ __attribute__((aligned(16))) unsigned char smem[BIG_NS * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, g = lane >> 5;
  int bm, bn, ks;
  if (per_xcd > 0) {                              // XCD-aware grouped order, as gemm_pl_kernel
    const int total = mtiles * ntiles;
    const int slot = blockIdx.x % (8 * per_xcd);
    ks = blockIdx.x / (8 * per_xcd);
    const int v = (slot & 7) * per_xcd + (slot >> 3);
    if (v >= total) return;
    constexpr int GROUP_M = 4;
    const int width = GROUP_M * ntiles;
    const int group = v / width, rem = v - group * width;
    const int first_m = group * GROUP_M;
    const int gsz = min(mtiles - first_m, GROUP_M);
    bm = first_m + rem % gsz;
    bn = rem / gsz;
  } else {
    bm = blockIdx.x % mtiles;
    bn = (blockIdx.x / mtiles) % ntiles;
    ks = blockIdx.x / (mtiles * ntiles);
  }
  const int m0 = bm * GBM;
  const int n0 = bn * BN;
  const int nsteps_all = d.CinP / BK;
  const int step0 = ks * sps;
  const int nsteps = min(sps, nsteps_all - step0);

  // ---- DMA: every wave requests 4 activation pieces + 6 weight pieces of 1 KiB per step
  const int64_t cstride = (int64_t)M * 32;
  const char* const wbase = reinterpret_cast<const char*>(d.wt_blk) + (int64_t)step0 * WBLK;
  const char* const abase = reinterpret_cast<const char*>(d.in_pl) + (int64_t)step0 * cstride;
  constexpr int NWP = NG * 2 / 4;                // weight pieces per wave and step
  unsigned aoff[4], woff[NWP];
  int wdst[NWP];
  {
    const unsigned pstride = (unsigned)(nsteps_all * cstride);
#pragma unroll
    for (int j = 0; j < 4; ++j) {                 // unit U = 64 (wave + 4 j) + lane of [plane][256 rows][2 halves]
      const int U = (wave + 4 * j) * 64 + lane;
      const int plane = U >> 9;
      const int u = U & 511;
      const int p = u >> 1;
      const int hs = (u & 1) ^ ((p >> 3) & 1);
      const int m = min(m0 + p, M - 1);
      aoff[j] = plane * pstride + (unsigned)m * 32u + hs * 16;
    }
    const int g0 = n0 / 32, ngroups = (d.N + 31) / 32;
#pragma unroll
    for (int jj = 0; jj < NWP; ++jj) {
      const int idx = wave * NWP + jj;            // (plane, column group)
      const int p = idx / NG, gi = idx - p * NG;
      const int gs = (g0 + gi) < ngroups ? g0 + gi : ngroups - 1;
      woff[jj] = (unsigned)(gs * nsteps_all) * WBLK + p * 1024 + lane * 16;
      wdst[jj] = __builtin_amdgcn_readfirstlane(A_STAGE + p * (BN * 32) + gi * 1024);
    }
  }
  auto dma = [&](const int t) __attribute__((always_inline)) {
    unsigned char* st = smem + (t % BIG_NS) * STAGE;
    const char* ab = abase + t * cstride;
#pragma unroll
    for (int j = 0; j < 4; ++j) gdma16(ab + aoff[j], st + wave * 1024 + j * 4096);
    const char* wb = wbase + (int64_t)t * WBLK;
#pragma unroll
    for (int jj = 0; jj < NWP; ++jj) gdma16(wb + woff[jj], st + wdst[jj]);
  };

  // ---- consumers: 2 (rows) x 2 (columns) waves of 128 x 32 TN
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  int a_frag[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int p = (wm * TM + i) * 32 + r;
    a_frag[i] = (2 * p + (g ^ ((p >> 3) & 1))) * 16;
  }
  const int b_frag = A_STAGE + (wn * TN) * 1024 + (2 * r + (g ^ ((r >> 3) & 1))) * 16;
  bf16x8 X[TM], Y[TN], Z[TM], W[TN];             // fragment banks (roles rotate, see below)
  auto ld_a = [&](bf16x8 (&F)[TM], const int s, const int plane) __attribute__((always_inline)) {
    const unsigned char* Ab = smem + (s % BIG_NS) * STAGE + plane * (512 * 16);
#pragma unroll
    for (int i = 0; i < TM; ++i) F[i] = *reinterpret_cast<const bf16x8*>(Ab + a_frag[i]);
  };
  auto ld_b = [&](bf16x8 (&F)[TN], const int s, const int plane) __attribute__((always_inline)) {
    const unsigned char* Bb = smem + (s % BIG_NS) * STAGE + b_frag + plane * (BN * 32);
#pragma unroll
    for (int j = 0; j < TN; ++j) F[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 1024);
  };
  auto mm = [&](const bf16x8 (&A)[TM], const bf16x8 (&B)[TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = AR::mfma(A[i], B[j], acc[i][j]);
  };
  auto finish = [&](const int s) __attribute__((always_inline)) {      // tile s + 2 landed, everybody done with stage s
    if (s + BIG_NS - 1 < nsteps) gwait_vm<4 + NWP>();
    else gwait_vm<0>();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
  };

  for (int t = 0; t < BIG_NS - 1 && t < nsteps; ++t) dma(t);
  gwait_vm<0>();
  __builtin_amdgcn_s_barrier();
  // One K16 step with the banks in the roles (LOA, HIB, HIA, LOB).  Its own hi_a / lo_b are read at its start (behind the previous step's
  // barrier; first needed by q1, one MFMA group later); lo_a of step s + 1 goes to LOA as soon as q0 has issued, hi_b of step s + 1 to the LOB
  // bank as soon as q1 has issued -- so the next step runs with HIB and LOB swapped.
  auto step = [&](const int s, bf16x8 (&LOA)[TM], bf16x8 (&HIB)[TN], bf16x8 (&HIA)[TM], bf16x8 (&LOB)[TN], auto first) __attribute__((always_inline)) {
    if constexpr (!decltype(first)::value) {
      ld_a(HIA, s, 0);
      ld_b(LOB, s, 1);
    }
    mm(LOA, HIB);                                 // q0: lo_a x hi_b
    __builtin_amdgcn_sched_barrier(0);
    if (s + BIG_NS - 1 < nsteps) dma(s + BIG_NS - 1);
    const bool more = s + 1 < nsteps;
    if (more) ld_a(LOA, s + 1, 1);
    mm(HIA, LOB);                                 // q1: hi_a x lo_b
    __builtin_amdgcn_sched_barrier(0);
    if (more) ld_b(LOB, s + 1, 0);
    mm(HIA, HIB);                                 // q2: hi_a x hi_b
    finish(s);
  };
  ld_a(X, 0, 1);
  ld_b(Y, 0, 0);
  ld_a(Z, 0, 0);
  ld_b(W, 0, 1);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  step(0, X, Y, Z, W, std::true_type{});
  int s = 1;
  for (; s + 1 < nsteps; s += 2) {
    step(s, X, W, Z, Y, std::false_type{});
    step(s + 1, X, Y, Z, W, std::false_type{});
  }
  if (s < nsteps) step(s, X, W, Z, Y, std::false_type{});

  // ---- epilogue (as gemm_pl_kernel)
  scale_all<TM, TN>(acc, d.acc_mul);
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int r_e = lane_e & 31, g_e = lane_e >> 5;
  int col[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) col[j] = n0 + (wn * TN + j) * 32 + r_e;
  const int64_t row0 = (int64_t)m0 + wm * TM * 32;
  const bool whole = m0 + GBM <= M && n0 + BN <= d.N && (int64_t)GBM * max(max(d.out_ld, d.res_ld), d.split_k > 1 ? d.splitk_ld : 0) * 4 < (1LL << 31);
  if (d.split_k > 1) {
    float* ws = d.splitk_ws + (int64_t)ks * M * d.splitk_ld;
    if (whole) store_tile_full<TM, TN>(acc, reinterpret_cast<char*>(ws + row0 * d.splitk_ld + n0 + wn * TN * 32), (int)d.splitk_ld, nullptr, 0, r_e, g_e);
    else store_tile_guarded<TM, TN>(acc, reinterpret_cast<char*>(ws + row0 * d.splitk_ld + n0 + wn * TN * 32), (int)d.splitk_ld, nullptr, 0, r_e, g_e,
                                    (int)min<int64_t>(M - row0, TM * 32), d.N - (n0 + wn * TN * 32), d.N - (n0 + wn * TN * 32));
    return;
  }
  float bias1[TN], zero[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    bias1[j] = (col[j] < d.N && d.bias) ? d.bias[col[j]] : 0.f;
    zero[j] = 0.f;
  }
  apply_act_all<TM, TN>(acc, bias1, zero, d.act);
  if (d.sumsq_part) {
    if (d.sumsq_hw > 0) write_sumsq_straddle<TM, TN>(acc, d.sumsq_part, d.N, row0, M, col, g_e, d.sumsq_hw);
    else write_sumsq<TM, TN>(acc, d.sumsq_part, d.N, row0, M, col, g_e);
  }
  char* const ob = reinterpret_cast<char*>(d.out + row0 * d.out_ld + d.out_coff + n0 + wn * TN * 32);
  const char* const rb = d.res ? reinterpret_cast<const char*>(d.res + row0 * d.res_ld + n0 + wn * TN * 32) : nullptr;
  if (whole) store_tile_full<TM, TN>(acc, ob, (int)d.out_ld, rb, (int)d.res_ld, r_e, g_e);
  else store_tile_guarded<TM, TN>(acc, ob, (int)d.out_ld, rb, (int)d.res_ld, r_e, g_e, (int)min<int64_t>(M - row0, TM * 32), d.N - (n0 + wn * TN * 32),
                                  d.n_store - (n0 + wn * TN * 32));
}

template <int TN>
int launch_gpl_big(const vs_conv_desc_t& d, hipStream_t st) {
  constexpr int BN = 64 * TN;
  const int64_t M = (int64_t)d.B * d.H * d.W;
  const int sk = d.split_k > 1 ? d.split_k : 1;
  const int64_t mt = cdiv64(M, GBM), nt = cdiv64(sk > 1 ? d.N : d.n_store, BN);
  const int nsteps = d.CinP / BK;
  const int sps = (nsteps + sk - 1) / sk;
  if (sk > 1 && (int64_t)(sk - 1) * sps >= nsteps) return VS_ERR_BAD_ARG;
  if (mt * nt * sk > 0x7fffffffLL || M > 0x7fffffffLL) return VS_ERR_UNSUPPORTED;
  if ((int64_t)GBM * std::max<int64_t>(std::max<int64_t>(d.out_ld, d.res ? d.res_ld : 0), sk > 1 ? d.splitk_ld : 0) * 4 >= (1LL << 31)) return VS_ERR_UNSUPPORTED;
  if (vs_max_lds_bytes() < BIG_NS * (A_STAGE + 2 * TN * 2048)) return VS_ERR_UNSUPPORTED;
  static const bool raster = [] { const char* e = getenv("VS_GEMM_PL_RASTER"); return !(e && e[0] == '0'); }();
  const int per_xcd = (raster && mt * nt >= 2 * vs_num_cus()) ? (int)((mt * nt + 7) / 8) : 0;
  const int64_t grid = per_xcd > 0 ? (int64_t)8 * per_xcd * sk : mt * nt * sk;
  hipLaunchKernelGGL((gemm_pl_big_kernel<TN>), dim3((unsigned)grid), dim3(256), 0, st, d, (int)M, (int)mt, (int)nt, sps, per_xcd);
  int rc = vs_launch_status();
  if (rc != VS_OK || sk == 1) return rc;
  return vs_splitk_epilogue(d, (int)M, st);
}

// fp32 rows [rows][ld] (optionally x * scale[frame][c] + shift[c]: GRN apply, common.py:166-169) -> operand planes [2][C/16][rows][16]
__global__ __launch_bounds__(256) void to_planes_affine_kernel(const float* __restrict__ x, const int64_t rows, const int C, const int64_t ld,
                                                               const float a_mul, const float* __restrict__ scale, const int64_t scale_ld,
                                                               const float* __restrict__ shift, const int rows_per_frame,
                                                               char* __restrict__ pl) {
  // thread = (row, 16-channel chunk pair member): idx -> (row, c8) with c8 fastest so that a wave reads 2 KiB of contiguous row memory
  const int c8n = C / 8;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * c8n) return;
  const int64_t row = idx / c8n;
  const int c8 = (int)(idx - row * c8n);
  const float* src = x + row * ld + c8 * 8;
  f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
  if (scale) {
    const float* sp = scale + (row / rows_per_frame) * scale_ld + c8 * 8;
    const float* hp = shift + c8 * 8;
    a = a * *reinterpret_cast<const f32x4*>(sp) + *reinterpret_cast<const f32x4*>(hp);
    b = b * *reinterpret_cast<const f32x4*>(sp + 4) + *reinterpret_cast<const f32x4*>(hp + 4);
  }
  u32x2 ah, al, bh, bl;
  split4h(a, a_mul, ah, al);
  split4h(b, a_mul, bh, bl);
  char* dst = pl + ((int64_t)(c8 >> 1) * rows + row) * 32 + (c8 & 1) * 16;
  *reinterpret_cast<u32x4*>(dst) = u32x4{ah[0], ah[1], bh[0], bh[1]};
  *reinterpret_cast<u32x4*>(dst + (int64_t)(C / 16) * rows * 32) = u32x4{al[0], al[1], bl[0], bl[1]};
}

// Round 6: the same conversion through an LDS transpose.  The kernel above maps a wave onto 64 consecutive 8-channel pieces of ONE row: its reads are
// 2 KiB of contiguous row memory, but its writes are 32-byte pieces rows * 32 bytes apart (one piece per 16-channel chunk) -- a quarter of a cache
// line each; on ChunkySeal's h tensors (15 376 x 5 792: 356 MB in, 356 MB out, 33 launches per extractor pass) it ran at 2.5 TB/s, 10.5 % of the
// step.  Here a workgroup owns 32 rows x 128 channels: 512-byte row pieces in, the split planes staged in LDS as [plane][chunk][row][32 B], and
// every (plane, chunk) goes out as ONE contiguous 1 KiB run (32 rows x 32 bytes).  Same expression per element -> the same bits.
__global__ __launch_bounds__(256) void to_planes_affine_tiled_kernel(const float* __restrict__ x, const int64_t rows, const int C, const int64_t ld,
                                                                     const float a_mul, const float* __restrict__ scale, const int64_t scale_ld,
                                                                     const float* __restrict__ shift, const int rows_per_frame,
                                                                     char* __restrict__ pl) {
  __shared__ __attribute__((aligned(16))) unsigned char stage[2 * 8 * 32 * 32];      // 16 KiB
  const int64_t row0 = (int64_t)blockIdx.x * 32;
  const int c0 = blockIdx.y * 128;
  const int t = threadIdx.x;
  const int c4 = t & 31, rl = t >> 5;                      // float4 column of the tile, row lane (rows rl + 8 j)
  const int c = c0 + 4 * c4;
  const bool c_ok = c < C;                                 // (C % 16 == 0: a float4 is inside or outside as a whole)
  f32x4 v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t row = row0 + rl + 8 * j;
    const int64_t rr = row < rows ? row : rows - 1;        // ragged last tile: a valid row, never written
    v[j] = *reinterpret_cast<const f32x4*>(x + rr * ld + (c_ok ? c : 0));
  }
  if (scale) {
    const f32x4 h4 = *reinterpret_cast<const f32x4*>(shift + (c_ok ? c : 0));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t row = row0 + rl + 8 * j;
      const int64_t rr = row < rows ? row : rows - 1;
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(scale + (rr / rows_per_frame) * scale_ld + (c_ok ? c : 0));
      v[j] = v[j] * s4 + h4;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u32x2 hi, lo;
    split4h(v[j], a_mul, hi, lo);
    const int off = (((c4 >> 2) * 32) + rl + 8 * j) * 32 + (c4 & 3) * 8;
    *reinterpret_cast<u32x2*>(stage + off) = hi;
    *reinterpret_cast<u32x2*>(stage + 8 * 32 * 32 + off) = lo;
  }
  __syncthreads();
  const int nrow = (int)(rows - row0 < 32 ? rows - row0 : 32);
  const int64_t pstride = (int64_t)(C / 16) * rows * 32;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int seg = (t >> 6) + 4 * j;                      // (plane, chunk of the tile): 1 KiB = 64 lanes x 16 bytes
    const int plane = seg >> 3, ch = seg & 7;
    const int chunk = (c0 >> 4) + ch;
    const int o = (t & 63) * 16;
    if (chunk * 16 < C && o < nrow * 32)
      *reinterpret_cast<u32x4*>(pl + plane * pstride + ((int64_t)chunk * rows + row0) * 32 + o) = *reinterpret_cast<const u32x4*>(stage + seg * 1024 + o);
  }
}

template <int TN>
int launch_gpl(const vs_conv_desc_t& d, hipStream_t st) {
  constexpr int BN = 2 * TN * 32;
  const int64_t M = (int64_t)d.B * d.H * d.W;
  const int sk = d.split_k > 1 ? d.split_k : 1;
  const int64_t mt = cdiv64(M, GBM), nt = cdiv64(sk > 1 ? d.N : d.n_store, BN);
  const int nsteps = d.CinP / BK;
  const int sps = (nsteps + sk - 1) / sk;
  if (sk > 1 && (int64_t)(sk - 1) * sps >= nsteps) return VS_ERR_BAD_ARG;       // an empty K slice
  if (mt * nt * sk > 0x7fffffffLL || M > 0x7fffffffLL) return VS_ERR_UNSUPPORTED;
  if (TN == 3 && sk == 1 && d.act == VS_ACT_TANH) return VS_ERR_UNSUPPORTED;       // (the TN = 3 register tile is compiled without the tanh epilogue; tile 25 / 27 have it)
  // XCD-aware grouped tile order for launches of several rounds (VS_GEMM_PL_RASTER=0: the row-major order)
  static const bool raster = [] { const char* e = getenv("VS_GEMM_PL_RASTER"); return !(e && e[0] == '0'); }();
  const int per_xcd = (raster && mt * nt >= 4 * vs_num_cus()) ? (int)((mt * nt + 7) / 8) : 0;
  const int64_t grid = per_xcd > 0 ? (int64_t)8 * per_xcd * sk : mt * nt * sk;
  hipLaunchKernelGGL((gemm_pl_kernel<TN>), dim3((unsigned)grid), dim3(512), 0, st, d, (int)M, (int)mt, (int)nt, sps, per_xcd);
  int rc = vs_launch_status();
  if (rc != VS_OK || sk == 1) return rc;
  return vs_splitk_epilogue(d, (int)M, st);
}

}  // namespace

// tile 24 = 256 rows x 192 columns, tile 25 = 256 rows x 128 columns, tile 27 = 256 rows x 384 columns (one wave per SIMD).  Preconditions are
// checked by vs_conv_gemm.
int vs_gemm_pl_dispatch(const vs_conv_desc_t& d, int tile, hipStream_t st) {
  switch (tile) {
    case 24: return launch_gpl<3>(d, st);
    case 25: return launch_gpl<2>(d, st);
    case 27: return launch_gpl_big<4>(d, st);
    default: return VS_ERR_UNSUPPORTED;
  }
}

extern "C" int vs_to_planes_affine(const float* x, int64_t rows, int C, int64_t ld, float a_mul, const float* scale, int64_t scale_ld,
                                   const float* shift, int rows_per_frame, void* planes, void* stream) {
  VS_REQUIRE(x && planes && rows > 0 && C > 0 && C % 16 == 0 && ld >= C && ld % 4 == 0 && a_mul > 0.f);
  VS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)planes & 15) == 0);
  if (scale) VS_REQUIRE(shift && rows_per_frame > 0 && scale_ld % 4 == 0 && ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0);
  // the LDS-transposed form for tensors of at least a few tiles (development switch 5 = 1 keeps the one-row-per-wave kernel)
  if (rows >= 32 && vs_debug_get(VS_DBG_TO_PLANES_FORM) != 1 && cdiv64(rows, 32) <= 0x7fffffffLL && (C + 127) / 128 <= 65535) {
    hipLaunchKernelGGL(to_planes_affine_tiled_kernel, dim3((unsigned)cdiv64(rows, 32), (unsigned)((C + 127) / 128)), dim3(256), 0, (hipStream_t)stream,
                       x, rows, C, ld, a_mul, scale, scale_ld, shift, rows_per_frame, static_cast<char*>(planes));
    return vs_launch_status();
  }
  const int64_t items = rows * (C / 8);
  hipLaunchKernelGGL(to_planes_affine_kernel, dim3((unsigned)cdiv64(items, 256)), dim3(256), 0, (hipStream_t)stream, x, rows, C, ld, a_mul,
                     scale, scale_ld, shift, rows_per_frame, static_cast<char*>(planes));
  return vs_launch_status();
}
