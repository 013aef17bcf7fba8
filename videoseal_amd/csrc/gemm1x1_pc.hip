// Wave-specialised GEMM for the 1x1 convolutions (ConvNeXt pwconv1 / pwconv2 of the extractor, convnext.py:96-105 of the
// reference): out[M][N] = act(A'[M][K] * W[N][K]^T + bias) (+ res),  A' = A or A * grn_scale[frame] + grn_shift.
// fp32 operands, split arithmetic of conv_common.h (template NP: 2 x f16 = 3 MFMAs, 3 x bf16 = 6 MFMAs per 32x32x16 block), fp32
// accumulate -- the same arithmetic and the same K order as conv_gemm_kernel of the same NP, so results are bit-identical to it (split_k = 1).
//
// 512 threads, one workgroup per CU, one s_barrier per 16-wide K step:
//   waves 0-3  consumers   : 2 x 2 waves, 64 rows x 32*TN columns each; fragment reads (ds_read_b128) of step s+1 are in
//                            flight while the 24 / 36 MFMAs of step s issue.
//   waves 4-5  A producers : read 128-byte row segments (32 fp32 = two K steps, whole cache lines), apply the GRN
//                            transform, split into three bf16 planes, ds_write them two steps ahead of use.  Register
//                            sets rotate three deep (loads ~4 steps ahead); the loop body is straight-line so that
//                            hipcc's s_waitcnt vmcnt(N) placement is exact.
//   waves 6-7  B producers : LDS-DMA (global_load_lds_dwordx4) of the pre-split, pre-swizzled weight blocks
//                            (engine.pack_blocked) into a ring of 6 (TN=2) / 5 (TN=3) tiles, counted s_waitcnt.
// split_k > 1 (small-M layers: 2048 / 8192 rows cannot fill 256 CUs with 128-row tiles): every K slice writes its raw
// partial sums to the workspace, vs_conv_gemm then runs splitk_epilogue_kernel (fixed summation order -> deterministic).
#include <algorithm>
#include <cstdlib>

#include "conv_common.h"
#include "lds_dma.h"

int vs_splitk_epilogue(const vs_conv_desc_t& d, int M, hipStream_t st);

#ifndef VS_PC_ABL
#define VS_PC_ABL 0       // timing ablations of the K loop (EXTRA=-DVS_PC_ABL=<bits>, tools/bench_gemm.py ksweep2; results are garbage): 1 no GRN apply / operand
#endif                    // split, 2 operands loaded once (no global loads in the loop), 4 no MFMAs, 8 no A stores to LDS, 16 no B stores to LDS (VS_PC_BDMA=0), 32 no wait for the weight DMA
#ifndef VS_PC_BDMA
#define VS_PC_BDMA 1      // weights by LDS-DMA from the consumer waves (2 x f16 arithmetic); 0: the producers copy them through registers (until round 5)
#endif
#ifndef VS_PC_PRIO
#define VS_PC_PRIO 0      // s_setprio of the producer waves (EXTRA=-DVS_PC_PRIO=1|3): they are the second-dispatched half, i.e. the VALU arbitration
                          // losers -- measured: no effect on any K (profiles/r05t_producer_priority.txt), so VALU issue is not what the producers wait for
#endif
namespace {

using namespace vsconv;

constexpr int BM = 128;
constexpr int GRN_KCAP = 3072;      // K elements of a K slice whose GRN scale / shift rows live in LDS (36 KB)
template <int NP> constexpr int a_stage() { return NP * BM * ROWB; }     // NP 16-bit planes of [128 rows][16 k], 48-byte rows (conflict-free b128 reads)

// One output tile (tile id -> (row tile, column tile, K slice)).  The kernel below is PERSISTENT: a workgroup walks tiles
// blockIdx.x, blockIdx.x + gridDim.x, ...  Both roles execute the same number of barriers per tile, all LDS fragment reads of a
// tile are retired before its last barrier, and the consumers' output stores are not waited for -- so the producers fetch the next
// tile's first operands while the consumers are in the epilogue, and the 50 MB of output of a 128 x 192-tile round drain under the
// next tile's MFMAs instead of in a stores-only phase at the end of every round (measured: stores 16.6 us + activation 6 us of a
// 75.7 us K = 384 launch, tools/bench_gemm.py ksweep).
// lane l: 16 bytes at gp -> lds_base + 16 * l, written out so that hipcc does not track it as a pending LDS write (convnext_fused.hip: it would put
// s_waitcnt vmcnt(0) in front of the next fragment read); the waits for these transfers are counted by hand
__device__ __forceinline__ void pc_dma16(const char* gp, unsigned char* lds_base) { vs_lds_dma16_untracked(gp, lds_base); }      // lds_dma.h
// ((g0 + g1) + g2) + g3 over four groups of KPER consecutive chunk sums each, chunks added in ascending order inside a group: the association of
// net_ops.hip::grn_finish_kernel (its four thread groups) for nch <= KMAX <= 16 chunks, kper = ceil(nch / 4) = KPER
template <int KPER, int KMAX>
__device__ __forceinline__ float grn_group_sums(const float (&v)[KMAX], const int nch) {
  float gs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kg = 0; kg < 4; ++kg)
#pragma unroll
    for (int q = 0; q < KPER; ++q) {
      constexpr int KL = KMAX - 1;
      const int k = kg * KPER + q;
      if (k < KMAX && k < nch) gs[kg] += v[k < KMAX ? k : KL];
    }
  return ((gs[0] + gs[1]) + gs[2]) + gs[3];
}
// Gx = sqrt(sum of the chunk sums) of the channels c = pt, pt + 256, ... of one frame: 48 / KMAX channels per batch, every load of a batch issued
// before the first addition (one L2 / MALL round trip per batch: the partials were written by the previous launch on other XCDs).  Returns the
// thread's sum of Gx (ascending channels); Gx of the K slice's channels go to `row`.
template <int KMAX>
__device__ __forceinline__ float grn_reduce_frame(const float* __restrict__ pf, const int nch, const int Kall, const int pt, const int c_lo,
                                                  const int c_hi, float* __restrict__ row) {
  constexpr int CB = 48 / KMAX;
  const int kper = (nch + 3) >> 2;
  float local = 0.f;
#pragma unroll 1
  for (int c0 = pt; c0 < Kall; c0 += 256 * CB) {
    float v[CB][KMAX];
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int c = c0 + 256 * j, cc = c < Kall ? c : Kall - 1;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) v[j][k] = pf[(int64_t)(k < nch ? k : nch - 1) * Kall + cc];      // (chunks past nch: a valid address, value unused)
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int c = c0 + 256 * j;
      const float gsum = kper == 1 ? grn_group_sums<1, KMAX>(v[j], nch) : kper == 2 ? grn_group_sums<2, KMAX>(v[j], nch)
                       : kper == 3 ? grn_group_sums<3, KMAX>(v[j], nch) : grn_group_sums<4, KMAX>(v[j], nch);
      const float gx = sqrtf(gsum);
      if (c < Kall) {
        local += gx;
        if (c >= c_lo && c < c_hi) row[c - c_lo] = gx;
      }
    }
  }
  return local;
}
template <int NP> constexpr bool pc_bdma() { return VS_PC_BDMA && NP == 2; }
template <int NP> constexpr int pc_bstages() { return pc_bdma<NP>() ? 4 : 2; }

template <int TN, bool GRN, int NP, int WN, int KC>
__device__ __forceinline__ void gemm1x1_pc_tile(const vs_conv_desc_t& d, const int M, const int mtiles, const int ntiles,
                                                const int pairs_per_split, const int tile, unsigned char* const smem) {
  // WN = 2: 2 x 2 consumer waves of 64 rows x 32*TN columns (tiles 128 x 128 / 128 x 192); WN = 1 (round 5): 4 x 1 waves of 32 rows x 32*TN
  // columns -- a 128 x 96 tile, twice as many workgroups per layer, for the GEMMs whose 128 x 192 tiles needed K slices (and a second
  // launch to add them up) to occupy 256 CUs: ConvNeXt stage-2 pwconv2, 8192 x 384 x 1536 = 64 x 4 tiles with the whole K in one workgroup
  constexpr int TM = WN;
  constexpr int BN = 32 * TN * WN;
  constexpr int NG = BN / 32;
  static_assert((NP * NG) % 2 == 0, "weight chunks per producer thread");
  using AR = Arith<NP>;
  constexpr int A_STAGE = a_stage<NP>();
  constexpr int WBLK = NP * 1024;                        // one (32 rows x 16 k) weight block, all planes
  constexpr int B_STAGE = NP * BN * 32;                  // NP planes of NG pre-swizzled 1 KiB blocks
  unsigned char* const Aring = smem;
  unsigned char* const Bring = smem + 2 * A_STAGE;
  // BDMA (round 5): the weight blocks are copied verbatim, so they do not have to pass through the producers' registers and the LDS store path
  // (79-85 B/clk per CU for the wide stores, shared by SIMD pairs: "no B stores" was worth 17 % of the K16 step, profiles/r05r_*): the four
  // CONSUMER waves -- no other vector memory traffic in their K loop -- fetch them by LDS-DMA, a pair at a time, into a ring of two pair slots.
  constexpr bool BDMA = pc_bdma<NP>();
  constexpr int NBS = pc_bstages<NP>();
  float* const Sc = reinterpret_cast<float*>(smem + 2 * A_STAGE + NBS * B_STAGE);    // GRN: [3][K slice] = scale of the two frames | shift

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, g = lane >> 5;

  const int bm = tile % mtiles;
  const int bn = (tile / mtiles) % ntiles;
  const int ks = tile / (mtiles * ntiles);
  const int m0 = bm * BM;
  const int n0 = bn * BN;
  const int pairs_total = d.CinP / 32;
  const int pair0 = ks * pairs_per_split;
  const int npairs = min(pairs_per_split, pairs_total - pair0);
  const int total = 2 * npairs;                 // 16-wide K steps of this workgroup, step s <-> K chunk 2*pair0 + s

  if (wave >= 4) {
    // ================================================================== producers (waves 4-7, one per SIMD)
    if (VS_PC_PRIO) __builtin_amdgcn_s_setprio(VS_PC_PRIO);
    // All data movement, with ordinary loads only, balanced over the four SIMDs (an earlier version had two waves splitting
    // A and two waves DMA-ing B: the splitting waves were the bottleneck, 152 vs 230 TF-eq with them idle).
    //   A: item (row, seg) = 16 bytes = 4 fp32 of row `row` at K offset seg*4 inside the 32-wide pair; 8 consecutive lanes read
    //      one 128-byte line.  seg 0-3 belong to the even step of the pair, seg 4-7 to the odd step.
    //   B: the pair's 2 * 3 * NG pre-split, pre-swizzled 1 KiB weight blocks, copied 16 bytes per thread and chunk.
    // Operands of step t are written during step t-2 and read (prefetched by the consumers) during step t-1: two LDS
    // stages each, stage = t & 1.  Three register sets rotate (a pair is loaded 3 pairs = 6 steps before it is stored); the
    // loop body is straight-line so that hipcc's s_waitcnt vmcnt(N) placement is exact.
    const int pt = tid & 255;
    constexpr int NI = BM * 8 / 256;       // 4 A items per thread per pair: 2 rows x the two K16 steps of the pair
    constexpr int NBP = BDMA ? 0 : NP * NG / 2;       // 16-byte weight chunks per thread per pair (2 steps * NP planes * NG * 64 / 256)
    constexpr int NBR = NBP ? NBP : 1;
    constexpr int CPS = NP * NG * 64;      // chunks per step
    // Item (j, half): 4 fp32 of row (pt >> 2) + 64 j at K offset (seg4 + 4 half) * 4 of the pair -- EVERY lane of a wave has work in both half
    // steps.  (Until round 5 a thread's items all sat in one half -- seg = pt & 7 -- so the GRN apply + split of a half step ran with half the
    // lanes masked off, twice per pair: the producers, not the consumers' MFMAs, set the K loop's 0.45 us per step, tools/bench_gemm.py ksweep2.)
    // Four consecutive lanes read 64 contiguous bytes; the other half of the line is the same lanes' `half = 1` load.
    const int seg4 = pt & 3;
    const int HW = d.H * d.W;
    unsigned a_off[2];
    int l_off[2];
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const int row = (pt >> 2) + j2 * 64;
      int m = m0 + row;
      m = m < M ? m : M - 1;                                            // ragged last tile: re-read a valid row (discarded)
      a_off[j2] = (unsigned)(((int64_t)m * d.in_sx + seg4 * 4) * 4);
      l_off[j2] = row * ROWB + seg4 * 8;                                // position inside a stage (the half selects the stage)
    }
    // GRN scale is per (frame, channel).  A 128-row tile touches at most two frames (H*W >= 128, or H*W == 64 with aligned tiles:
    // checked by the dispatcher): frame f_lo up to the row `rb` where the next frame starts, f_hi from there on.  The slice's scale rows of
    // both frames and the shift are copied to LDS once per tile (round 5: they were re-fetched from global memory with every pair, three
    // loads and twelve registers per register set).
    const int f_lo = min(m0 / HW, d.B - 1), f_hi = min(f_lo + 1, d.B - 1);
    const int rb = (f_lo + 1) * HW - m0;              // first tile row of frame f_lo + 1 (>= 128: the tile lies in one frame)
    bool hi_sel[2];
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) hi_sel[j2] = ((pt >> 2) + j2 * 64) >= rb;
    const char* const abase = reinterpret_cast<const char*>(d.in) + (int64_t)pair0 * 128;
    constexpr bool grn = GRN;
    const int kslice = npairs * 32;
    // (the thread id of the GRN prologue through an opaque copy, as the consumers' lane id in the epilogue below: hipcc otherwise hoists the prologue's
    // per-thread addresses out of the persistent tile loop and keeps them alive across every K loop -- 14 spilled registers of the TN = 3 x 2 tile)
    int pt_o = pt;
    asm volatile("" : "+v"(pt_o));
    if (grn && d.grn_part) {
      // Round 6: the GRN finish (net_ops.hip::grn_finish_kernel, one launch of ~8 us per block between pwconv1 and pwconv2) folded into this
      // prologue.  d.grn_part = pwconv1's ||h||^2 partials [B][nchunk][K] (nchunk <= 16 chunks of 32 rows per frame): every workgroup reduces
      // the chunks of ITS frame(s) for all K channels -- the channel mean of Gx needs them all --, keeps Gx of its K slice in the scale rows, and
      // turns them into scale = 1 + gamma * Gx / (mean + 1e-6) (common.py:166-169).  Same sums in the same order as the kernel it replaces
      // (per channel ((g0 + g1) + g2) + g3 over four groups of chunks; channels c = pt + 256 j ascending per thread; the 256 thread sums by the
      // halving tree, here evaluated redundantly by every producer wave) -> the same scale values bit for bit.  48 KB of L2 reads per workgroup.
      const int Kall = d.CinP, nch = d.grn_nchunk;
      float* const Red = Sc + 3 * KC;                 // [2 frames][256 thread sums]
      const int c_lo = pair0 * 32, c_hi = c_lo + kslice;
#pragma unroll 1
      for (int fi = 0; fi < 2; ++fi) {
        const int f = fi ? f_hi : f_lo;
        if (fi && f_hi == f_lo) break;
        const float* pf = d.grn_part + (int64_t)f * nch * Kall;
        float* row = Sc + fi * kslice;
        Red[fi * 256 + pt_o] = nch <= 2 ? grn_reduce_frame<2>(pf, nch, Kall, pt_o, c_lo, c_hi, row)
                           : nch <= 4 ? grn_reduce_frame<4>(pf, nch, Kall, pt_o, c_lo, c_hi, row)
                           : nch <= 8 ? grn_reduce_frame<8>(pf, nch, Kall, pt_o, c_lo, c_hi, row)
                                      : grn_reduce_frame<16>(pf, nch, Kall, pt_o, c_lo, c_hi, row);
      }
      __syncthreads();                                // (matched by the consumers' extra cbar)
      const float* ghp = d.a_shift + c_lo;
      const float* gmp = d.grn_gamma + c_lo;
      float mean[2];
#pragma unroll
      for (int fi = 0; fi < 2; ++fi) {
        const float* rr = Red + (f_hi == f_lo ? 0 : fi * 256);
        float a = (rr[lane] + rr[lane + 128]) + (rr[lane + 64] + rr[lane + 192]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
        mean[fi] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a))) / (float)Kall;
      }
      for (int k = pt_o; k < kslice; k += 256) {
        const float gm = gmp[k];
        const float g0v = Sc[k], g1v = f_hi == f_lo ? g0v : Sc[kslice + k];
        Sc[k] = 1.0f + gm * (g0v / (mean[0] + 1e-6f));
        Sc[kslice + k] = 1.0f + gm * (g1v / (mean[1] + 1e-6f));
        Sc[2 * kslice + k] = ghp[k];
      }
      __syncthreads();                                // (matched by the consumers' first cbar)
    } else if (grn) {
      const float* g0p = d.a_scale + (int64_t)f_lo * d.a_scale_ld + pair0 * 32;
      const float* g1p = d.a_scale + (int64_t)f_hi * d.a_scale_ld + pair0 * 32;
      const float* ghp = d.a_shift + pair0 * 32;
      for (int k4 = pt_o * 4; k4 < kslice; k4 += 1024) {
        *reinterpret_cast<f32x4*>(Sc + k4) = *reinterpret_cast<const f32x4*>(g0p + k4);
        *reinterpret_cast<f32x4*>(Sc + kslice + k4) = *reinterpret_cast<const f32x4*>(g1p + k4);
        *reinterpret_cast<f32x4*>(Sc + 2 * kslice + k4) = *reinterpret_cast<const f32x4*>(ghp + k4);
      }
      __syncthreads();                                // (matched by the consumers' first cbar)
    }
    // weight chunks of this thread
    const int g0 = n0 / 32;
    const int ngroups = (d.N + 31) / 32;
    const int nch = d.CinP / 16;
    unsigned b_goff[NBR];
    int b_loff[NBR];
    bool b_half[NBR];
#pragma unroll
    for (int j = 0; j < NBP; ++j) {
      const int cid = pt + j * 256;
      const int half = cid >= CPS ? 1 : 0;
      const int c = cid - half * CPS;
      const int p = c / (NG * 64), rem = c - p * (NG * 64);
      const int gi = rem >> 6, l = rem & 63;
      const int gsel = (g0 + gi) < ngroups ? g0 + gi : ngroups - 1;     // tile wider than N: re-read a valid group (columns discarded)
      b_goff[j] = (unsigned)(((int64_t)gsel * nch + half) * WBLK + p * 1024 + l * 16);
      b_loff[j] = half * B_STAGE + p * (BN * 32) + gi * 1024 + l * 16;
      b_half[j] = half != 0;
    }
    const char* const wbase = reinterpret_cast<const char*>(d.wt_blk) + (int64_t)(2 * pair0) * WBLK;
    const float amul = NP == 2 ? d.a_mul : 1.f;
    const int lastp = npairs - 1;

    struct PSet { f32x4 r[NI]; u32x4 b[NBR]; int pair; };
    PSet rs0, rs1, rs2;      // three rotating register sets
    auto load_pair = [&](PSet& R, int j) __attribute__((always_inline)) {
      j = j < lastp ? j : lastp;                                        // past the end: harmless re-read, keeps the body branch-free
      if ((VS_PC_ABL & 2) && j > 2) { R.pair = j; return; }
      R.pair = j;
      const char* base = abase + (int64_t)j * 128;
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
        for (int h = 0; h < 2; ++h) R.r[j2 * 2 + h] = *reinterpret_cast<const f32x4*>(base + a_off[j2] + h * 64);
      const char* wb = wbase + (int64_t)j * (2 * WBLK);
#pragma unroll
      for (int q = 0; q < NBP; ++q) R.b[q] = *reinterpret_cast<const u32x4*>(wb + b_goff[q]);
    };
    // half h of the pair in R -> stage h: every thread splits + stores its two rows' items of that K16 step
    auto store_half = [&](const PSet& R, const int h) __attribute__((always_inline)) {
      f32x4 sc0, sc1, sh;
      if (grn) {
        const int ko = R.pair * 32 + (seg4 + 4 * h) * 4;
        sc0 = *reinterpret_cast<const f32x4*>(Sc + ko);
        sc1 = *reinterpret_cast<const f32x4*>(Sc + kslice + ko);
        sh = *reinterpret_cast<const f32x4*>(Sc + 2 * kslice + ko);
      }
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        f32x4 v = R.r[j2 * 2 + h];
        u32x2 pl[NP];
        if (VS_PC_ABL & 1) {
#pragma unroll
          for (int p = 0; p < NP; ++p) pl[p] = u32x2{__float_as_uint(v[p]), __float_as_uint(v[p + 1])};
        } else {
          if (grn) v = v * (hi_sel[j2] ? sc1 : sc0) + sh;               // GRN apply (same expression as conv_gemm_kernel)
          split4n<NP>(v, amul, pl);
        }
        unsigned char* dst = Aring + h * A_STAGE + l_off[j2];
        if (!(VS_PC_ABL & 8)) {
#pragma unroll
          for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(dst + p * BM * ROWB) = pl[p];
        } else {
          asm volatile("" ::"v"(pl[0]), "v"(pl[NP - 1]));
        }
      }
      if (!(VS_PC_ABL & 16)) {
#pragma unroll
        for (int q = 0; q < NBP; ++q)
          if (b_half[q] == (h != 0)) *reinterpret_cast<u32x4*>(Bring + b_loff[q]) = R.b[q];
      }
    };
    // A(t) (step t) is written during step t-2 and read (prefetched) during step t-1: two stages, stage = t & 1.
    // prologue: pair 0 -> both stages; pairs 1, 2, 3 -> registers.
    load_pair(rs0, 0);
    load_pair(rs1, 1);
    load_pair(rs2, 2);
    store_half(rs0, 0);
    store_half(rs0, 1);
    load_pair(rs0, 3);
    __syncthreads();
    __syncthreads();        // the consumers read the fragments of step 0 between these two barriers: stage 0 is rewritten in step 0
    // steps 2j, 2j+1 (consumers on pair j): store pair j+1 (set (j+1)%3), then reload that set with pair j+4.
    // The main loop takes whole triples and has NO exit inside its body: with `if (j + 1 >= npairs) break;` between the sub-iterations (the
    // form until round 5) hipcc's wait-count bookkeeping gave up at the loop header and the first sub-iteration of every trip waited
    // vmcnt(0) -- for the two pairs just requested as well, i.e. the three-deep prefetch drained every third pair (ISA: vmcnt(7..0) / no
    // waits / vmcnt(29..23) in the three sub-iterations; now vmcnt(33..26) in each).  The 0 - 2 pairs that are left run behind the loop.
    auto sub = [&](PSet& R, const int nxt) __attribute__((always_inline)) {
      store_half(R, 0);
      __syncthreads();
      store_half(R, 1);
      load_pair(R, nxt);
      __syncthreads();
    };
    int j = 0;
    for (; j + 3 <= npairs; j += 3) {
      sub(rs1, j + 4);
      sub(rs2, j + 5);
      sub(rs0, j + 6);
    }
    if (j < npairs) {
      sub(rs1, j + 4);
      if (j + 1 < npairs) sub(rs2, j + 5);
    }
    return;
  }

  // ==================================================================== consumers
  const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
  // consumer-side barrier: LDS reads retired, but NOT vmcnt(0) -- __syncthreads() would wait for the previous tile's output stores
  // (persistent kernel) before the first barrier of the next tile; the consumers exchange nothing through global memory
  auto cbar = []() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int a_frag = (wm * TM * 32 + r) * ROWB + g * 16;
  const int b_frag = (wn * TN) * 1024 + (2 * r + (g ^ ((r >> 3) & 1))) * 16;   // pack_blocked's bank swizzle

  struct Frags { bf16x8 a[TM][NP]; bf16x8 b[TN][NP]; };
  Frags F0, F1;
  auto load_frags = [&, a_frag, b_frag](Frags& F, const int s) __attribute__((always_inline)) {
    const unsigned char* Bb = Bring + (s & (NBS - 1)) * B_STAGE;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < NP; ++p) F.b[j][p] = *reinterpret_cast<const bf16x8*>(Bb + p * (BN * 32) + b_frag + j * 1024);
    const unsigned char* Ab = Aring + (s & 1) * A_STAGE;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int p = 0; p < NP; ++p) F.a[i][p] = *reinterpret_cast<const bf16x8*>(Ab + p * BM * ROWB + a_frag + i * 32 * ROWB);
  };
  auto mfma_all = [&](const Frags& F) __attribute__((always_inline)) {
    if (VS_PC_ABL & 4) {
      asm volatile("" ::"v"(F.a[0][0]), "v"(F.b[0][0]), "v"(F.a[TM - 1][NP - 1]), "v"(F.b[TN - 1][NP - 1]));
      return;
    }
#pragma unroll
    for (int q = 0; q < AR::NPROD; ++q) {   // smallest partial products first; consecutive MFMAs hit different accumulators
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = AR::mfma(F.a[i][AR::PA[q]], F.b[j][AR::PB[q]], acc[i][j]);
    }
  };

  // BDMA: pair P = the 2 NP NG weight blocks of steps 2P, 2P+1 -> stages 2 (P & 1), 2 (P & 1) + 1 (stage of step s = s & 3); wave w moves blocks
  // w NBW .. w NBW + NBW - 1 of every pair, one 1 KiB block per instruction.  A slot is read (fragment prefetch) during steps 2P-1 and 2P, so pair
  // P + 2 is requested at the start of step 2P+1 and has to be complete at the barrier that ends step 2P+2: each wave waits for its OWN transfers
  // (vmcnt(0): nothing else of the K loop is in flight; after a persistent workgroup's previous tile that includes the tail of its output stores).
  constexpr int NBW = NP * NG / 2;
  unsigned w_goff[NBW];
  int w_loff[NBW];
  const char* wbase_c = nullptr;
  if constexpr (BDMA) {
    const int g0 = n0 / 32, ngroups = (d.N + 31) / 32, nch = d.CinP / 16;
#pragma unroll
    for (int q = 0; q < NBW; ++q) {
      const int bi = wave * NBW + q;                                    // wave-uniform
      const int half = bi / (NP * NG), c = bi - half * (NP * NG);
      const int p = c / NG, gi = c - p * NG;
      const int gsel = (g0 + gi) < ngroups ? g0 + gi : ngroups - 1;     // tile wider than N: re-read a valid group (columns discarded)
      w_goff[q] = (unsigned)(((int64_t)gsel * nch + half) * WBLK + p * 1024) + lane * 16;
      w_loff[q] = half * B_STAGE + p * (BN * 32) + gi * 1024;
    }
    wbase_c = reinterpret_cast<const char*>(d.wt_blk) + (int64_t)(2 * pair0) * WBLK;
  }
  auto dma_pair = [&](const int P) __attribute__((always_inline)) {
    if constexpr (BDMA) {
      const char* wb = wbase_c + (int64_t)P * (2 * WBLK);
      unsigned char* slot = Bring + (P & 1) * (2 * B_STAGE);
#pragma unroll
      for (int q = 0; q < NBW; ++q) pc_dma16(wb + w_goff[q], slot + w_loff[q]);
    }
  };
  auto dma_wait0 = []() __attribute__((always_inline)) {
    if constexpr (BDMA && !(VS_PC_ABL & 32)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  const int npairs_c = total >> 1;
  if constexpr (BDMA) {                  // (every fragment read of the previous tile retired before its last barrier: the ring is free)
    dma_pair(0);
    if (npairs_c > 1) dma_pair(1);
    if (npairs_c > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBW) : "memory");      // pair 0 (transfers complete in order: at most pair 1's are left)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if constexpr (GRN) {
    if (d.grn_part) cbar();             // the producers' Gx reduction (round 6: the GRN finish folded into this kernel)
    cbar();                             // the producers' copy of the GRN scale rows to LDS
  }
  cbar();
  load_frags(F0, 0);
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0), see conv3x3_patch_pc.hip
  cbar();                       // step 0's operands are in registers: the producers may now overwrite stage 0 with step 2's
  int s = 0;
  for (; s + 2 < total; s += 2) {
    load_frags(F1, s + 1);
    mfma_all(F0);
    dma_wait0();                          // pair s/2 + 1 (requested during step s - 1, or in the prologue)
    cbar();
    if (BDMA && (s >> 1) + 2 < npairs_c) dma_pair((s >> 1) + 2);
    load_frags(F0, s + 2);
    mfma_all(F1);
    cbar();
  }
  load_frags(F1, s + 1);              // total is even: two steps left
  mfma_all(F0);
  cbar();
  mfma_all(F1);
  cbar();

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  int col[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) col[j] = n0 + (wn * TN + j) * 32 + r;
  if constexpr (NP == 2) scale_all<TM, TN>(acc, d.acc_mul);       // back to real units (exact: a power of two), also for the K-slice partial sums
  // (the lane id through an opaque copy: this function is the body of the persistent tile loop, and hipcc otherwise hoists the 64 lane offsets of
  // the store helpers out of that loop -- they do not depend on the tile -- and keeps them alive, i.e. in scratch, across every K loop)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int r_e = lane_e & 31, g_e = lane_e >> 5;
  // wave-uniform position of the wave's first element, and what is left of the matrix from there (rows / columns / stored columns)
  const int64_t row0 = (int64_t)m0 + wm * TM * 32;
  const int c0 = n0 + wn * TN * 32;
  const bool whole = m0 + BM <= M && n0 + BN <= d.N;      // every layer of the shipped cards
  const int rows_left = (int)min((int64_t)TM * 32, (int64_t)M - row0);
  // ONE store path for the K-slice partial sums and the final output (round 6, as gemm_pl.hip: two inlined copies of the store helpers next to each other
  // and the tanh variant of the activation were what the TN = 3 x 2 tile spilled 92 - 97 registers for)
  const bool sk = d.split_k > 1;
  if (!sk) {
    float bias1[TN], zero[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bias1[j] = (col[j] < d.N && d.bias) ? d.bias[col[j]] : 0.f;
      zero[j] = 0.f;
    }
    const int abl = VS_KERNEL_ABL(d);       // ablations (tools/bench_gemm.py ksweep): 64 no activation, 32 no output stores
    if (!(abl & 64)) apply_act_all<TM, TN, (TN * WN < 6)>(acc, bias1, zero, d.act);
    if (d.sumsq_part) write_sumsq<TM, TN>(acc, d.sumsq_part, d.N, m0 + (int64_t)wm * TM * 32, M, col, g);
    if (abl & 32) return;
  }
  // raw partial sums -> workspace [ks][M][ws_ld]; or the output (+ residual)
  char* const ob = sk ? reinterpret_cast<char*>(d.splitk_ws + ((int64_t)ks * M + row0) * d.splitk_ld + c0)
                      : reinterpret_cast<char*>(d.out + row0 * d.out_ld + d.out_coff + c0);
  const char* const rb = (!sk && d.res) ? reinterpret_cast<const char*>(d.res + row0 * d.res_ld + c0) : nullptr;
  const int o_ld = sk ? (int)d.splitk_ld : (int)d.out_ld;
  if (whole) store_tile_full<TM, TN>(acc, ob, o_ld, rb, (int)d.res_ld, r_e, g_e);
  else store_tile_guarded<TM, TN>(acc, ob, o_ld, rb, (int)d.res_ld, r_e, g_e, rows_left, d.N - c0, sk ? d.N - c0 : d.n_store - c0);
}

// KC: K elements of a K slice whose GRN rows fit the LDS area (GRN_KCAP, or half of it: with the four weight stages of BDMA the 128 x 96 tile stays
// at two workgroups per CU only with the smaller area -- ConvNeXt stage-2 pwconv2, K = 1536)
template <int TN, bool GRN, int NP, int WN = 2, int KC = GRN_KCAP>
__global__ __launch_bounds__(512, 2) void gemm1x1_pc_kernel(const vs_conv_desc_t d, const int M, const int mtiles, const int ntiles,
                                                            const int pairs_per_split, const int ntot) {
  constexpr int B_STAGE = NP * (32 * TN * WN) * 32;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * a_stage<NP>() + pc_bstages<NP>() * B_STAGE + (GRN ? 3 * KC * 4 + 2 * 256 * 4 : 0)];
  for (int tile = blockIdx.x; tile < ntot; tile += gridDim.x) gemm1x1_pc_tile<TN, GRN, NP, WN, KC>(d, M, mtiles, ntiles, pairs_per_split, tile, smem);
}

// out = act(sum_ks ws[ks] + bias) [+ ws[split_k] + bias2 : the 1x1 second phase of the patch kernel] (+ res); columns in
// [N, n_store) are written as zero.  One thread per (row, 4 columns), 16-byte accesses where the operands allow it.
// Same epilogue order as the kernels themselves; the slices are added in slice order (deterministic).
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const vs_conv_desc_t d, const int M, const int vec) {
  const int ncol4 = (d.n_store + 3) / 4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)M * ncol4) return;
  const int64_t m = idx / ncol4;
  const int n4 = (int)(idx % ncol4) * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  const bool full = n4 + 4 <= d.N;
  if (full) {            // splitk_ld % 4 == 0 and the workspace is 16-byte aligned (checked by the launcher)
    for (int k = 0; k < d.split_k; ++k) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(d.splitk_ws + ((int64_t)k * M + m) * d.splitk_ld + n4);
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] += t[c];
    }
  } else {
    for (int c = 0; c < 4; ++c)
      if (n4 + c < d.N)
        for (int k = 0; k < d.split_k; ++k) v[c] += d.splitk_ws[((int64_t)k * M + m) * d.splitk_ld + n4 + c];
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int n = n4 + c;
    if (n >= d.N) { v[c] = 0.f; continue; }
    float t = v[c] + (d.bias ? d.bias[n] : 0.f);
    if (d.act == VS_ACT_RELU) t = vs_relu(t);
    else if (d.act == VS_ACT_GELU) t = vs_gelu(t);
    else if (d.act == VS_ACT_TANH) t = tanhf(t);
    if (d.in2 || d.in2_pl) t += d.splitk_ws[((int64_t)d.split_k * M + m) * d.splitk_ld + n] + (d.bias2 ? d.bias2[n] : 0.f);
    if (d.res) t += d.res[m * d.res_ld + n];
    v[c] = t;
  }
  if (d.out_pl && n4 + 4 <= d.N) {      // the next conv's operand planes (conv3x3_pl.hip): hi / lo f16 of v * a_mul, [plane][N / 16][pixel][16]
    vsconv::u32x2 hi, lo;
    vsconv::split4h(f32x4{v[0], v[1], v[2], v[3]}, d.a_mul, hi, lo);
    char* dst = reinterpret_cast<char*>(d.out_pl) + ((int64_t)(n4 >> 4) * M + m) * 32 + (n4 & 15) * 2;
    *reinterpret_cast<vsconv::u32x2*>(dst) = hi;
    *reinterpret_cast<vsconv::u32x2*>(dst + (int64_t)(d.N / 16) * M * 32) = lo;
  }
  if (!d.out) return;
  float* o = d.out + m * d.out_ld + d.out_coff + n4;
  if (vec && n4 + 4 <= d.n_store) {
    *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
  } else {
    for (int c = 0; c < 4; ++c)
      if (n4 + c < d.n_store) o[c] = v[c];
  }
}

// The same for the shapes every shipped layer has (N % 4 == 0, 16-byte aligned out / res / bias rows): round 6.  The kernel above is a chain of
// dependent round trips -- hipcc waits vmcnt(0) behind every slice's load in its run-time loop, and bias / second-phase / residual values arrive as
// twelve scalar loads with a wait each (tools/isa_scan.py order: `LOOP LD vmcnt(0)` x split_k, then `LD vmcnt(0)` x 12) -- 9 - 13 us for a few MB.
// Here EVERY load of an item is issued before the first use: SK 16-byte slice loads (compile-time count; SK = 0: run-time count, four in flight),
// bias, second-phase slice + bias2, residual.  Same additions in the same order per element -> the same values bit for bit.
template <int SK>
__global__ __launch_bounds__(256) void splitk_epilogue_vec_kernel(const vs_conv_desc_t d, const int M) {
  const int ncol4 = d.n_store / 4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)M * ncol4) return;
  const int64_t m = idx / ncol4;
  const int n4 = (int)(idx % ncol4) * 4;
  float* const o = d.out ? d.out + m * d.out_ld + d.out_coff + n4 : nullptr;
  if (n4 >= d.N) {                          // columns in [N, n_store): zeros
    if (o) *reinterpret_cast<f32x4*>(o) = f32x4{0.f, 0.f, 0.f, 0.f};
    return;
  }
  const float* const w0 = d.splitk_ws + m * d.splitk_ld + n4;
  const int64_t sstride = (int64_t)M * d.splitk_ld;
  const bool two = d.in2 || d.in2_pl;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 bv = z4, b2v = z4, p2 = z4, rv = z4;
  f32x4 v = z4;
  if constexpr (SK > 0) {
    f32x4 t[SK];
#pragma unroll
    for (int k = 0; k < SK; ++k) t[k] = *reinterpret_cast<const f32x4*>(w0 + k * sstride);
    if (d.bias) bv = *reinterpret_cast<const f32x4*>(d.bias + n4);
    if (two) {
      p2 = *reinterpret_cast<const f32x4*>(w0 + SK * sstride);
      if (d.bias2) b2v = *reinterpret_cast<const f32x4*>(d.bias2 + n4);
    }
    if (d.res) rv = *reinterpret_cast<const f32x4*>(d.res + m * d.res_ld + n4);
#pragma unroll
    for (int k = 0; k < SK; ++k) v += t[k];
  } else {
    const int sk = d.split_k;
    if (d.bias) bv = *reinterpret_cast<const f32x4*>(d.bias + n4);
    if (two) {
      p2 = *reinterpret_cast<const f32x4*>(w0 + sk * sstride);
      if (d.bias2) b2v = *reinterpret_cast<const f32x4*>(d.bias2 + n4);
    }
    if (d.res) rv = *reinterpret_cast<const f32x4*>(d.res + m * d.res_ld + n4);
    int k = 0;
    for (; k + 4 <= sk; k += 4) {
      f32x4 t[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) t[q] = *reinterpret_cast<const f32x4*>(w0 + (k + q) * sstride);
#pragma unroll
      for (int q = 0; q < 4; ++q) v += t[q];
    }
    for (; k < sk; ++k) v += *reinterpret_cast<const f32x4*>(w0 + k * sstride);
  }
  float r[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float t = v[c] + bv[c];                  // (bias absent: + 0.f, as `v + (d.bias ? .. : 0.f)` above)
    if (d.act == VS_ACT_RELU) t = vs_relu(t);
    else if (d.act == VS_ACT_GELU) t = vs_gelu(t);
    else if (d.act == VS_ACT_TANH) t = tanhf(t);
    if (two) t += p2[c] + b2v[c];
    if (d.res) t += rv[c];
    r[c] = t;
  }
  if (d.out_pl) {      // the next conv's operand planes (conv3x3_pl.hip): hi / lo f16 of v * a_mul, [plane][N / 16][pixel][16]
    vsconv::u32x2 hi, lo;
    vsconv::split4h(f32x4{r[0], r[1], r[2], r[3]}, d.a_mul, hi, lo);
    char* dst = reinterpret_cast<char*>(d.out_pl) + ((int64_t)(n4 >> 4) * M + m) * 32 + (n4 & 15) * 2;
    *reinterpret_cast<vsconv::u32x2*>(dst) = hi;
    *reinterpret_cast<vsconv::u32x2*>(dst + (int64_t)(d.N / 16) * M * 32) = lo;
  }
  if (o) *reinterpret_cast<f32x4*>(o) = f32x4{r[0], r[1], r[2], r[3]};
}

template <int TN, int WN = 2>
int launch_g(const vs_conv_desc_t& d, hipStream_t st) {
  constexpr int BN = 32 * TN * WN;
  const int64_t M = (int64_t)d.B * d.H * d.W;
  const int64_t mt = cdiv64(M, BM), nt = cdiv64(d.split_k > 1 ? d.N : d.n_store, BN);
  const int pairs = d.CinP / 32;
  const int sk = d.split_k > 1 ? d.split_k : 1;
  const int pps = (pairs + sk - 1) / sk;
  if ((int64_t)(sk - 1) * pps >= pairs) return VS_ERR_BAD_ARG;           // an empty K slice
  if (d.a_scale && pps * 32 > GRN_KCAP) return VS_ERR_UNSUPPORTED;       // the GRN rows of a K slice are staged in LDS
  if (d.a_scale && (((uintptr_t)d.a_scale | (uintptr_t)d.a_shift) & 15 || (d.a_scale_ld & 3))) return VS_ERR_UNSUPPORTED;
  if (d.grn_part && (!d.a_scale || !d.grn_gamma || d.grn_nchunk < 1 || d.grn_nchunk > 16 || d.grn_nchunk * 32 != d.H * d.W)) return VS_ERR_BAD_ARG;
  if (mt * nt * sk > 0x7fffffffLL || M > 0x7fffffffLL) return VS_ERR_UNSUPPORTED;
  if (TN * WN >= 6 && sk == 1 && d.act == VS_ACT_TANH) return VS_ERR_UNSUPPORTED;      // (the 128 x 192 register tile is compiled without the tanh epilogue)
  // store_tile_full / store_tile_guarded address a tile with 32-bit byte offsets from its first element (conv_gemm.hip / gemm_pl.hip check the
  // same bound and keep a 64-bit path; this kernel has only the 32-bit one)
  if ((int64_t)BM * std::max<int64_t>(std::max<int64_t>(d.out_ld, d.res ? d.res_ld : 0), sk > 1 ? d.splitk_ld : 0) * 4 >= (1LL << 31)) return VS_ERR_UNSUPPORTED;
  const int ntot = (int)(mt * nt * sk);
  static const int force_grid = [] { const char* e = getenv("VS_GEMM_GRID"); return e ? atoi(e) : 0; }();      // experiments: 0 = one workgroup per CU
  const int grid = std::min(ntot, force_grid > 0 ? force_grid : vs_num_cus());
  if (d.arith == 2) {
    if (d.a_scale && pps * 32 <= GRN_KCAP / 2)
      hipLaunchKernelGGL((gemm1x1_pc_kernel<TN, true, 2, WN, GRN_KCAP / 2>), dim3(grid), dim3(512), 0, st, d, (int)M, (int)mt, (int)nt, pps, ntot);
    else if (d.a_scale)
      hipLaunchKernelGGL((gemm1x1_pc_kernel<TN, true, 2, WN>), dim3(grid), dim3(512), 0, st, d, (int)M, (int)mt, (int)nt, pps, ntot);
    else
      hipLaunchKernelGGL((gemm1x1_pc_kernel<TN, false, 2, WN>), dim3(grid), dim3(512), 0, st, d, (int)M, (int)mt, (int)nt, pps, ntot);
  } else if constexpr (WN == 1) {
    return VS_ERR_UNSUPPORTED;          // 128 x 96 tiles: 2 x f16 arithmetic only (3 planes x 3 column groups do not divide over the producer threads)
  } else if (d.a_scale)
    hipLaunchKernelGGL((gemm1x1_pc_kernel<TN, true, 3, WN>), dim3(grid), dim3(512), 0, st, d, (int)M, (int)mt, (int)nt, pps, ntot);
  else
    hipLaunchKernelGGL((gemm1x1_pc_kernel<TN, false, 3, WN>), dim3(grid), dim3(512), 0, st, d, (int)M, (int)mt, (int)nt, pps, ntot);
  int rc = vs_launch_status();
  if (rc != VS_OK || sk == 1) return rc;
  return vs_splitk_epilogue(d, (int)M, st);
}

}  // namespace

int vs_splitk_epilogue(const vs_conv_desc_t& d, int M, hipStream_t st) {
  const int64_t items = (int64_t)M * ((d.n_store + 3) / 4);
  if ((d.splitk_ld & 3) || ((uintptr_t)d.splitk_ws & 15)) return VS_ERR_BAD_ARG;
  const int vec = ((d.out_ld & 3) == 0 && (d.out_coff & 3) == 0 && ((uintptr_t)d.out & 15) == 0) ? 1 : 0;
  // every operand in whole 16-byte pieces: the form with all loads of an item in flight at once (VS_SPLITK_EPI=scalar keeps the first form; experiments)
  static const bool scalar_only = [] { const char* e = getenv("VS_SPLITK_EPI"); return e && e[0] == 's'; }();
  const bool wide = !scalar_only && vs_debug_get(VS_DBG_SPLITK_EPI) != 1 && (vec || !d.out) && (d.N & 3) == 0 && (d.n_store & 3) == 0 && (!d.bias || ((uintptr_t)d.bias & 15) == 0) &&
                    (!d.bias2 || ((uintptr_t)d.bias2 & 15) == 0) && (!d.res || (((uintptr_t)d.res & 15) == 0 && (d.res_ld & 3) == 0));
  if (wide) {
    const dim3 grid((unsigned)cdiv64((int64_t)M * (d.n_store / 4), 256));
    switch (d.split_k) {
      case 2: hipLaunchKernelGGL(splitk_epilogue_vec_kernel<2>, grid, dim3(256), 0, st, d, M); break;
      case 3: hipLaunchKernelGGL(splitk_epilogue_vec_kernel<3>, grid, dim3(256), 0, st, d, M); break;
      case 4: hipLaunchKernelGGL(splitk_epilogue_vec_kernel<4>, grid, dim3(256), 0, st, d, M); break;
      case 6: hipLaunchKernelGGL(splitk_epilogue_vec_kernel<6>, grid, dim3(256), 0, st, d, M); break;
      case 8: hipLaunchKernelGGL(splitk_epilogue_vec_kernel<8>, grid, dim3(256), 0, st, d, M); break;
      default: hipLaunchKernelGGL(splitk_epilogue_vec_kernel<0>, grid, dim3(256), 0, st, d, M); break;
    }
    return vs_launch_status();
  }
  hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)cdiv64(items, 256)), dim3(256), 0, st, d, M, vec);
  return vs_launch_status();
}

// tile 17 = 128 x 128, tile 18 = 128 x 192, tile 26 = 128 x 96 (four consumer waves stacked over the rows).  Preconditions are checked by vs_conv_gemm.
int vs_gemm1x1_pc_dispatch(const vs_conv_desc_t& d, int tile, hipStream_t st) {
  switch (tile) {
    case 17: return launch_g<2>(d, st);
    case 18: return launch_g<3>(d, st);
    case 26: return launch_g<3, 1>(d, st);
    default: return VS_ERR_UNSUPPORTED;
  }
}
