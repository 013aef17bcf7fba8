// 3x3 stride-1 convolution with the input patch staged ONCE per 16-channel chunk ("patch" kernel), 3 x bf16 split.
//
// The generic implicit-GEMM kernel (conv_gemm.hip) re-gathers the A tile from global memory for every filter tap: nine
// shifted reads of the same pixels per channel chunk and nine operand splits.  For the U-Net (all 3x3, stride 1, most of
// the embed FLOPs, several layers with only 16-64 channels at 128^2 / 256^2 where the conv is bandwidth-bound) this is
// the dominant cost.  Here a workgroup owns an 8 x 16 block of output pixels of one frame; per 16-channel chunk it loads
// the 10 x 18 input patch (zero or reflect padded) once, splits it once into three bf16 planes in LDS, and the nine taps
// read their A fragments from that patch with a tap offset (one ds_read_b128 per fragment, per-lane base + immediate).
// Global A traffic drops from 9x to (10*18)/(8*16) = 1.4x of the input, the split VALU work by 6.4x.
// B (pre-split weights, [3][N][tap][CinP] bf16) is streamed per (chunk, tap) step through a double-buffered LDS tile
// exactly like the generic kernel.  K order is (chunk, tap): identical for every tile shape of this kernel.
#include "conv_common.h"

namespace {

using namespace vsconv;

constexpr int TH = 8, TW = 16, PW = TW + 2, PH = TH + 2, PROWS = PW * PH;   // 180 patch pixels
constexpr int BM = TH * TW;                                                  // 128 output pixels
constexpr int NTH = 256;
constexpr int PITEMS = PROWS * 4;                                            // float4 items of a patch chunk
constexpr int NPI = (PITEMS + NTH - 1) / NTH;                                // per thread: 3

template <int WM, int WN, int TM, int TN, int NP>
__global__ __launch_bounds__(NTH, 2) void conv3x3_patch_kernel(const vs_conv_desc_t d, const int tiles_x, const int tiles_y,
                                                               const int mtiles) {
  constexpr int BN = WN * TN * 32;
  static_assert(WM * WN == 4 && WM * TM * 32 == BM, "4 waves cover the 128-pixel tile");
  constexpr int BSLOTS = BN * 2;                        // 16-byte B slots per plane per step
  constexpr int NB = (BSLOTS + NTH - 1) / NTH;
  using AR = Arith<NP>;
  constexpr int P_BYTES = NP * PROWS * ROWB;            // patch, NP 16-bit planes
  constexpr int B_BYTES = NP * BN * ROWB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[P_BYTES + 2 * B_BYTES];
  unsigned char* const Ps = smem;
  unsigned char* const Bs0 = smem + P_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, g = lane >> 5;

  const int bm = blockIdx.x % mtiles;
  const int bn = blockIdx.x / mtiles;
  const int n0 = bn * BN;
  const int tx = bm % tiles_x;
  const int ty = (bm / tiles_x) % tiles_y;
  const int fb = bm / (tiles_x * tiles_y);             // frame
  const int y0 = ty * TH, x0 = tx * TW;

  const int spt = d.CinP / BK;
  const int n1 = 9 * spt;
  const int n2 = d.in2 ? d.Cin2P / BK : 0;
  const int64_t Ktot = (int64_t)9 * d.CinP;
  const bool reflect = d.pad_mode == VS_PAD_REFLECT;
  const int abl = VS_KERNEL_ABL(d);   // debug ablation (tools/bench_conv.py): 1 no B loads, 2 no MFMA, 4 no B store, 8 no barrier

  // ---- patch items of this thread: (patch pixel, 4-channel group); addresses are constant across chunks
  unsigned p_off[NPI];
  int p_lds[NPI];
  bool p_ok[NPI], p_have[NPI];
#pragma unroll
  for (int i = 0; i < NPI; ++i) {
    const int item = tid + i * NTH;
    p_have[i] = item < PITEMS;
    const int prow = p_have[i] ? item >> 2 : 0;
    const int k4 = (item & 3) * 4;
    int iy = y0 - 1 + prow / PW, ix = x0 - 1 + prow % PW;
    bool ok = true;
    if (reflect) {
      iy = iy < 0 ? -iy : (iy >= d.H ? 2 * d.H - 2 - iy : iy);
      ix = ix < 0 ? -ix : (ix >= d.W ? 2 * d.W - 2 - ix : ix);
      ok = iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;   // (tiles hanging over the image edge)
    } else {
      ok = iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
    }
    p_ok[i] = ok && p_have[i];
    p_off[i] = ok ? (unsigned)(((int64_t)fb * d.in_sb + (int64_t)iy * d.in_sy + (int64_t)ix * d.in_sx + k4) * 4) : 0u;
    p_lds[i] = prow * ROWB + k4 * 2;
  }
  // ---- phase-2 rows (1x1 conv on in2): thread -> (pixel, 4-channel group) x 2
  unsigned q_off[2];
  bool q_ok[2];
  const int qk4 = (tid & 3) * 4;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = (tid + i * NTH) >> 2;
    const int y = y0 + (p >> 4), x = x0 + (p & 15);
    q_ok[i] = y < d.H && x < d.W;
    const int64_t m = ((int64_t)fb * d.H + (q_ok[i] ? y : 0)) * d.W + (q_ok[i] ? x : 0);
    q_off[i] = (unsigned)((m * d.in2_ld + qk4) * 4);
  }
  // ---- B slots
  unsigned b_off[NB], b_off2[NB];
  int b_lds[NB];
  bool b_have[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int idx = tid + i * NTH;
    b_have[i] = idx < BSLOTS;
    const int row = idx >> 1, sub = idx & 1;
    int nrow = n0 + row;
    nrow = nrow < d.N ? nrow : d.N - 1;
    b_off[i] = (unsigned)(((int64_t)nrow * Ktot + sub * 8) * 2);
    b_off2[i] = (unsigned)(((int64_t)nrow * d.Cin2P + sub * 8) * 2);
    b_lds[i] = row * ROWB + sub * 16;
  }
  const int64_t plane1 = (int64_t)d.N * Ktot * 2;
  const int64_t plane2 = (int64_t)d.N * d.Cin2P * 2;
  const char* const wbase = reinterpret_cast<const char*>(d.wt_split);
  const char* const wbase2 = reinterpret_cast<const char*>(d.wt2_split);

  const float amul = NP == 2 ? d.a_mul : 1.f;
  f32x4 rp[NPI];
  u32x4 rb[NB][NP];

  auto load_patch = [&, tid](const int cc) __attribute__((always_inline)) {      // channel chunk cc -> registers
    const bool cok = (cc * BK + (tid & 3) * 4) < d.Cin;
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      // (branch-free since round 6, as the weight loads below: a load inside an `if` region is waited for at the join.  Lanes without a slot or
      //  beyond Cin read the tensor's first 16 bytes -- an offset select, always inside the allocation -- and are zeroed as before)
      const bool ld_ok = p_have[i] && cok;
      const unsigned off = ld_ok ? p_off[i] + (unsigned)cc * (BK * 4) : 0u;
      f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(d.in) + off);
      if (!(ld_ok && p_ok[i])) v = f32x4{0.f, 0.f, 0.f, 0.f};
      rp[i] = v;
    }
  };
  auto store_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NPI; ++i)
      if (p_have[i]) {
        u32x2 pl[NP];
        split4n<NP>(rp[i], amul, pl);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(Ps + p * PROWS * ROWB + p_lds[i]) = pl[p];
      }
  };
  auto load_b = [&](const int cc, const int tap) __attribute__((always_inline)) {
    const char* bbase = wbase + ((int64_t)tap * d.CinP + (int64_t)cc * BK) * 2;
#pragma unroll
    for (int i = 0; i < NB; ++i) {      // (unconditional since round 6: b_off of a thread without a slot is a clamped, valid weight row -- a load inside an
#pragma unroll                          //  `if (b_have)` region made hipcc wait vmcnt(0) for it BEFORE the step's MFMAs; only the LDS store stays guarded)
      for (int p = 0; p < NP; ++p) rb[i][p] = *reinterpret_cast<const u32x4*>(bbase + p * plane1 + b_off[i]);
    }
  };
  auto load_b2 = [&](const int c2) __attribute__((always_inline)) {
    const char* bbase = wbase2 + (int64_t)c2 * (BK * 2);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
#pragma unroll
      for (int p = 0; p < NP; ++p) rb[i][p] = *reinterpret_cast<const u32x4*>(bbase + p * plane2 + b_off2[i]);
    }
  };
  auto store_b = [&](const int buf) __attribute__((always_inline)) {
    unsigned char* Bb = Bs0 + buf * B_BYTES;
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (b_have[i]) {
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(Bb + p * BN * ROWB + b_lds[i]) = rb[i][p];
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
  // A fragment base of MFMA tile i: output pixel p = (wm*TM + i)*32 + r  ->  patch row (p>>4)*PW + (p&15)  [tap (0,0)]
  int a_frag[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int p = (wm * TM + i) * 32 + r;
    a_frag[i] = ((p >> 4) * PW + (p & 15)) * ROWB + g * 16;
  }
  const int b_frag = ((wn * TN) * 32 + r) * ROWB + g * 16;

  auto mfma_step = [&](const int buf, const int a_tap_off) __attribute__((always_inline)) {
    const unsigned char* Bb = Bs0 + buf * B_BYTES;
    bf16x8 af[TM][NP], bf[TN][NP];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int p = 0; p < NP; ++p) af[i][p] = *reinterpret_cast<const bf16x8*>(Ps + p * PROWS * ROWB + a_frag[i] + a_tap_off);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < NP; ++p) bf[j][p] = *reinterpret_cast<const bf16x8*>(Bb + p * BN * ROWB + b_frag + j * 32 * ROWB);
      // product term outermost, tiles innermost: consecutive MFMAs hit different accumulators (no dependent-issue stall);
      // smallest terms first
#pragma unroll
      for (int q = 0; q < AR::NPROD; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = AR::mfma(af[i][AR::PA[q]], bf[j][AR::PB[q]], acc[i][j]);
      }
  };

  // ---- phase 1: chunks x 9 taps.  B step s = cc*9 + tap lives in buffer s & 1
  load_patch(0);
  load_b(0, 0);
  store_patch();
  store_b(0);
  __syncthreads();
  int step = 0;
  for (int cc = 0; cc < spt; ++cc) {
    if (cc + 1 < spt) load_patch(cc + 1);                // in flight during the nine taps of this chunk
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap, ++step) {
      const int s1 = step + 1;
      if (s1 < n1) {
        const int ncc = tap == 8 ? cc + 1 : cc, ntap = tap == 8 ? 0 : tap + 1;
        if (!(abl & 1)) load_b(ncc, ntap);
      } else if (n2 > 0) {
        load_b2(0);
      }
      const int ky = tap / 3, kx = tap - ky * 3;
      if (!(abl & 2)) mfma_step(step & 1, (ky * PW + kx) * ROWB);
      if ((s1 < n1 || n2 > 0) && !(abl & 4)) store_b(s1 & 1);
      if (!(abl & 8)) __syncthreads();
    }
    if (cc + 1 < spt) {                                   // every wave is past tap 8: the patch may be replaced
      store_patch();
      __syncthreads();
    }
  }
  // ---- phase 2: the ResnetBlock's 1x1 res_conv on the block input, accumulated on top of relu(bn(conv3x3))
  if constexpr (NP == 2) scale_all<TM, TN>(acc, d.acc_mul);      // back to real units (exact: a power of two)
  if (n2 > 0) {
    apply_act_all<TM, TN>(acc, bias1, bias2, d.act);
    if constexpr (NP == 2) scale_all<TM, TN>(acc, 1.f / d.acc_mul2);   // into the units of the phase-2 products
    int a2_frag[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) a2_frag[i] = ((wm * TM + i) * 32 + r) * ROWB + g * 16 - a_frag[i];   // rows are plain pixels
    for (int c2 = 0; c2 < n2; ++c2, ++step) {
      // A chunk of in2 -> patch rows 0..127 (the patch buffer is free: phase 1 is over)
      const char* abase = reinterpret_cast<const char*>(d.in2) + (int64_t)c2 * (BK * 4);
      const bool cok = (c2 * BK + qk4) < d.Cin2;
      f32x4 qa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        qa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (q_ok[i] && cok) qa[i] = *reinterpret_cast<const f32x4*>(abase + q_off[i]);
      }
      if (c2 + 1 < n2) load_b2(c2 + 1);
      __syncthreads();                                    // previous readers of the patch rows are done
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        u32x2 pl[NP];
        split4n<NP>(qa[i], amul, pl);
        const int off = ((tid + i * NTH) >> 2) * ROWB + qk4 * 2;
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(Ps + p * PROWS * ROWB + off) = pl[p];
      }
      __syncthreads();
      {
        const unsigned char* Bb = Bs0 + (step & 1) * B_BYTES;
        bf16x8 af[TM][NP], bf[TN][NP];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int p = 0; p < NP; ++p) af[i][p] = *reinterpret_cast<const bf16x8*>(Ps + p * PROWS * ROWB + a_frag[i] + a2_frag[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int p = 0; p < NP; ++p) bf[j][p] = *reinterpret_cast<const bf16x8*>(Bb + p * BN * ROWB + b_frag + j * 32 * ROWB);
      // product term outermost, tiles innermost: consecutive MFMAs hit different accumulators (no dependent-issue stall);
      // smallest terms first
#pragma unroll
      for (int q = 0; q < AR::NPROD; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = AR::mfma(af[i][AR::PA[q]], bf[j][AR::PB[q]], acc[i][j]);
      }
      }
      if (c2 + 1 < n2) store_b((step + 1) & 1);
    }
    if constexpr (NP == 2) scale_all<TM, TN>(acc, d.acc_mul2);
  } else {
    float zero[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) zero[j] = 0.f;
    apply_act_all<TM, TN>(acc, bias1, zero, d.act);
  }

  // ---- epilogue: C/D layout col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5); row = pixel inside the 8 x 16 tile
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int p = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
      const int y = y0 + (p >> 4), x = x0 + (p & 15);
      if (y >= d.H || x >= d.W) continue;
      const int64_t m = ((int64_t)fb * d.H + y) * d.W + x;
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

template <int WM, int WN, int TM, int TN>
int launch_patch(const vs_conv_desc_t& d, hipStream_t st) {
  const bool h2 = d.arith == 2;
  constexpr int BN = WN * TN * 32;
  const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
  const int64_t mt = (int64_t)d.B * tiles_x * tiles_y, nt = cdiv64(d.n_store, BN);
  if (mt * nt > 0x7fffffffLL) return VS_ERR_UNSUPPORTED;
  if (h2) hipLaunchKernelGGL((conv3x3_patch_kernel<WM, WN, TM, TN, 2>), dim3((unsigned)(mt * nt)), dim3(NTH), 0, st, d, tiles_x, tiles_y, (int)mt);
  else hipLaunchKernelGGL((conv3x3_patch_kernel<WM, WN, TM, TN, 3>), dim3((unsigned)(mt * nt)), dim3(NTH), 0, st, d, tiles_x, tiles_y, (int)mt);
  return vs_launch_status();
}

}  // namespace

// 3x3 / stride 1 / pad 1 / same-size output, split weights present (checked by the caller, conv_gemm.hip).
// tile: 10 = 128 x 32, 11 = 128 x 64, 12 = 128 x 128
int vs_conv3x3_patch_dispatch(const vs_conv_desc_t& d, int tile, hipStream_t st) {
  switch (tile) {
    case 10: return launch_patch<4, 1, 1, 1>(d, st);
    case 11: return launch_patch<2, 2, 2, 1>(d, st);
    case 12: return launch_patch<2, 2, 2, 2>(d, st);
    default: return VS_ERR_UNSUPPORTED;
  }
}
