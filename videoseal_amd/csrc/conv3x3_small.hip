// 3x3 stride-1 convolution for the thin, full-resolution layers of the U-Net (16 input channels, <= 32 output channels at
// 256^2 / 128^2: unet.py:24-39 `inc` / last `ups` ResnetBlocks), optional fused 1x1 res_conv on a 16-channel block input.
// Same arithmetic as conv3x3_patch_kernel (3 x bf16 split, six v_mfma_f32_32x32x16_bf16 per tap, K order = taps) so the
// results are bit-identical to it; what differs is the schedule.  These layers have K = 9 (+1) steps in total: streaming
// one weight tile per step through LDS with a barrier each (the patch kernel) spends the time on barriers and on
// ~10 dependent global round trips per 128-pixel tile.  Here
//   * the whole weight set of a wave (27 + 3 fragments) is loaded ONCE into registers -- workgroups are persistent and
//     walk over `tiles_per_wg` pixel tiles;
//   * the only LDS traffic is the input patch (10 x 18 pixels x 16 channels, split once), double-buffered: the patch of
//     tile t+1 is loaded into registers before, and split + stored after, the MFMAs of tile t; one barrier per tile;
//   * the nine taps of a tile run back to back (3 ds_read_b128 + 6 MFMAs each, fully unrolled).
#include "conv_common.h"

namespace {

using namespace vsconv;

constexpr int TH = 8, TW = 16, PW = TW + 2, PH = TH + 2, PROWS = PW * PH;   // 180 patch pixels
constexpr int BMP = TH * TW;                                                 // 128 output pixels per tile
constexpr int PITEMS = PROWS * 4;
constexpr int NPI = (PITEMS + 255) / 256;                                    // 3 patch float4 items per thread

template <bool IN2, int NP>
__global__ __launch_bounds__(256, 2) void conv3x3_small_kernel(const vs_conv_desc_t d, const int tiles_x, const int tiles_y,
                                                               const int ntiles_total, const int tiles_per_wg) {
  using AR = Arith<NP>;
  constexpr int P_BYTES = NP * PROWS * ROWB;                                   // one patch buffer (NP 16-bit planes)
  constexpr int Q_BYTES = NP * BMP * ROWB;                                     // in2 rows of a tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * P_BYTES + (IN2 ? Q_BYTES : 0)];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 31, g = lane >> 5;
  const bool reflect = d.pad_mode == VS_PAD_REFLECT;
  const int abl = VS_KERNEL_ABL(d);      // debug ablation (tools/bench_small.py): 1 no MFMA, 2 no output stores, 4 no patch loads / staging

  // ---- weights of this wave: B fragment (tap, plane) = 16 bytes of row n = r at k = tap*16 + g*8  (wt_split: [3][N][144])
  const int nrow = r < d.N ? r : d.N - 1;                 // N < 32: lanes beyond N re-read a valid row (columns discarded)
  const int64_t plane1 = (int64_t)d.N * 144 * 2;
  const char* wb = reinterpret_cast<const char*>(d.wt_split) + ((int64_t)nrow * 144 + g * 8) * 2;
  bf16x8 bw[9][NP];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int p = 0; p < NP; ++p) bw[t][p] = *reinterpret_cast<const bf16x8*>(wb + p * plane1 + t * 32);
  bf16x8 bw2[NP];
  if (IN2) {
    const char* wb2 = reinterpret_cast<const char*>(d.wt2_split) + ((int64_t)nrow * 16 + g * 8) * 2;
#pragma unroll
    for (int p = 0; p < NP; ++p) bw2[p] = *reinterpret_cast<const bf16x8*>(wb2 + p * ((int64_t)d.N * 16 * 2));
  }
  // MFMA operands are swapped (weights = A operand, pixels = B operand): the accumulator is C[channel][pixel], i.e. lane =
  // pixel (lane & 31) and element e = channel (e&3) + 8*(e>>2) + 4*g.  Every lane then owns runs of 4 consecutive
  // channels of ONE pixel -> 16-byte stores / residual loads (with C[pixel][channel] a store instruction writes 4 bytes
  // per lane into 64-byte pieces; for N = 16 that alone cost as much as the MFMAs).
  float bias1[16], bias2[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int ch = (e & 3) + 8 * (e >> 2) + 4 * g;
    bias1[e] = (ch < d.N && d.bias) ? d.bias[ch] : 0.f;
    bias2[e] = (IN2 && ch < d.N && d.bias2) ? d.bias2[ch] : 0.f;
  }

  // ---- per-thread patch items: position relative to the tile origin (constant), byte offset for interior tiles
  int p_dy[NPI], p_dx[NPI], p_lds[NPI];
  unsigned p_rel[NPI];
  bool p_have[NPI];
  const int k4 = (tid & 3) * 4;
  const bool cok = k4 < d.Cin;
#pragma unroll
  for (int i = 0; i < NPI; ++i) {
    const int item = tid + i * 256;
    p_have[i] = item < PITEMS;
    const int prow = p_have[i] ? item >> 2 : 0;
    p_dy[i] = prow / PW - 1;
    p_dx[i] = prow % PW - 1;
    p_lds[i] = prow * ROWB + k4 * 2;
    p_rel[i] = (unsigned)(((int64_t)(p_dy[i] + 1) * d.in_sy + (int64_t)(p_dx[i] + 1) * d.in_sx + (cok ? k4 : 0)) * 4);   // from pixel (y0-1, x0-1)
  }
  const bool cok2 = IN2 && k4 < d.Cin2;
  // A fragment of this wave: output pixel p = wave*32 + r  ->  patch row (p>>4)*PW + (p&15)  [tap (0,0)]
  const int pw = wave * 32 + r;
  const int a_frag = ((pw >> 4) * PW + (pw & 15)) * ROWB + g * 16;
  const int a2_frag = pw * ROWB + g * 16;
  const int py = pw >> 4, px = pw & 15;                   // this lane's output pixel inside the tile
  const bool vec_io = (d.out_ld & 3) == 0 && (d.out_coff & 3) == 0 && ((uintptr_t)d.out & 15) == 0 &&
                      (!d.res || ((d.res_ld & 3) == 0 && ((uintptr_t)d.res & 15) == 0));

  const float amul = NP == 2 ? d.a_mul : 1.f;
  const float acc_mul = NP == 2 ? d.acc_mul : 1.f, acc_mul2 = (NP == 2 && IN2) ? d.acc_mul2 : 1.f, inv_mul2 = 1.f / acc_mul2;
  f32x4 rp[NPI], rq[2];
  auto load_tile = [&](const int fb, const int y0, const int x0) __attribute__((always_inline)) {   // patch (+ in2 rows) -> registers
    const bool interior = y0 >= 1 && x0 >= 1 && y0 + TH + 1 <= d.H && x0 + TW + 1 <= d.W;           // workgroup-uniform
    if (interior) {       // no padding, no clamping: one 64-bit base + constant 32-bit lane offsets
      const char* base = reinterpret_cast<const char*>(d.in + (int64_t)fb * d.in_sb + (int64_t)(y0 - 1) * d.in_sy + (int64_t)(x0 - 1) * d.in_sx);
#pragma unroll
      for (int i = 0; i < NPI; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (p_have[i] ? p_rel[i] : 0u));
        rp[i] = cok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
#pragma unroll
      for (int i = 0; i < NPI; ++i) {
        int iy = y0 + p_dy[i], ix = x0 + p_dx[i];
        if (reflect) {
          iy = iy < 0 ? -iy : (iy >= d.H ? 2 * d.H - 2 - iy : iy);
          ix = ix < 0 ? -ix : (ix >= d.W ? 2 * d.W - 2 - ix : ix);
        }
        const bool ok = p_have[i] && cok && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
        const int cy = iy < 0 ? 0 : (iy >= d.H ? d.H - 1 : iy), cx = ix < 0 ? 0 : (ix >= d.W ? d.W - 1 : ix);
        const f32x4 v = *reinterpret_cast<const f32x4*>(d.in + (int64_t)fb * d.in_sb + (int64_t)cy * d.in_sy + (int64_t)cx * d.in_sx + (cok ? k4 : 0));
        rp[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};          // unconditional load from a clamped address, zeroed afterwards
      }
    }
    if (IN2) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int p = (tid + i * 256) >> 2;
        const int y = y0 + (p >> 4), x = x0 + (p & 15);
        const bool ok = cok2 && y < d.H && x < d.W;
        const int64_t m = ((int64_t)fb * d.H + (y < d.H ? y : d.H - 1)) * d.W + (x < d.W ? x : d.W - 1);
        const f32x4 v = *reinterpret_cast<const f32x4*>(d.in2 + m * d.in2_ld + (cok2 ? k4 : 0));
        rq[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto store_tile = [&](const int buf) __attribute__((always_inline)) {       // registers -> split -> LDS buffer `buf`
    unsigned char* Ps = smem + buf * P_BYTES;
#pragma unroll
    for (int i = 0; i < NPI; ++i)
      if (p_have[i]) {
        u32x2 pl[NP];
        split4n<NP>(rp[i], amul, pl);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(Ps + p * PROWS * ROWB + p_lds[i]) = pl[p];
      }
    if (IN2) {
      unsigned char* Qs = smem + 2 * P_BYTES;              // single buffer (LDS: two workgroups per CU), see the barrier below
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        u32x2 pl[NP];
        split4n<NP>(rq[i], amul, pl);
        const int off = ((tid + i * 256) >> 2) * ROWB + k4 * 2;
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(Qs + p * BMP * ROWB + off) = pl[p];
      }
    }
  };

  const int t_begin = blockIdx.x * tiles_per_wg;
  const int t_end = min(ntiles_total, t_begin + tiles_per_wg);
  if (t_begin >= t_end) return;
  // tile coordinates are carried along instead of being re-derived by division for every tile
  int tx = t_begin % tiles_x, ty = (t_begin / tiles_x) % tiles_y, fb = t_begin / (tiles_x * tiles_y);
  load_tile(fb, ty * TH, tx * TW);
  store_tile(0);
  __syncthreads();
  for (int t = t_begin; t < t_end; ++t) {
    const int buf = (t - t_begin) & 1;
    int ntx = tx + 1, nty = ty, nfb = fb;
    if (ntx == tiles_x) { ntx = 0; if (++nty == tiles_y) { nty = 0; ++nfb; } }
    if (t + 1 < t_end && !(abl & 4)) load_tile(nfb, nty * TH, ntx * TW);                  // in flight during this tile's MFMAs
    const unsigned char* Ps = smem + buf * P_BYTES + a_frag;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int toff = ((tap / 3) * PW + (tap % 3)) * ROWB;
      bf16x8 af[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) af[p] = *reinterpret_cast<const bf16x8*>(Ps + p * PROWS * ROWB + toff);
      if (abl & 1) { acc[tap] += (float)af[0][0] + (float)af[1][1]; continue; }
      // smallest partial products first (same order as every other split kernel); weights are the A operand
#pragma unroll
      for (int q = 0; q < AR::NPROD; ++q) acc = AR::mfma(bw[tap][AR::PB[q]], af[AR::PA[q]], acc);
    }
    if constexpr (NP == 2) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] *= acc_mul;       // back to real units (exact: a power of two)
    }
    // epilogue order of vs_conv_gemm: v = act(acc + bias); [phase 2: v += in2 (1x1) wt2 + bias2]; v += res
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = vs_apply_act(acc[e] + bias1[e], d.act) + bias2[e];
    if (IN2) {
      const unsigned char* Qs = smem + 2 * P_BYTES + a2_frag;
      bf16x8 af[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) af[p] = *reinterpret_cast<const bf16x8*>(Qs + p * BMP * ROWB);
      if constexpr (NP == 2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] *= inv_mul2;    // into the units of the phase-2 products
      }
#pragma unroll
      for (int q = 0; q < AR::NPROD; ++q) acc = AR::mfma(bw2[AR::PB[q]], af[AR::PA[q]], acc);
      if constexpr (NP == 2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] *= acc_mul2;
      }
    }
    const int y = ty * TH + py, x = tx * TW + px;
    if (y < d.H && x < d.W && !((abl & 2) && acc[0] != 123.25f)) {
      const int64_t m = ((int64_t)fb * d.H + y) * d.W + x;
      float* orow = d.out + m * d.out_ld + d.out_coff;
      const float* rrow = d.res ? d.res + m * d.res_ld : nullptr;
#pragma unroll
      for (int q = 0; q < 4; ++q) {                        // channels 8q + 4g .. + 3
        const int c0 = 8 * q + 4 * g;
        if (c0 >= d.n_store) continue;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = (c0 + c < d.N) ? acc[4 * q + c] : 0.f;      // columns in [N, n_store) are zero padding
        if (vec_io && c0 + 4 <= d.n_store) {
          if (rrow && c0 + 4 <= d.N) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(rrow + c0);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] += rv[c];
          } else if (rrow) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c0 + c < d.N) v[c] += rrow[c0 + c];
          }
          *reinterpret_cast<f32x4*>(orow + c0) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (c0 + c < d.n_store) {
              if (rrow && c0 + c < d.N) v[c] += rrow[c0 + c];
              orow[c0 + c] = v[c];
            }
        }
      }
    }
    if (t + 1 < t_end) {
      if (IN2) __syncthreads();              // the in2 rows have one buffer: every wave must be done reading tile t's
      if (!(abl & 4)) store_tile(buf ^ 1);   // the other patch buffer was last read during tile t-1: every wave is past that barrier
      __syncthreads();
    }
    tx = ntx; ty = nty; fb = nfb;
  }
}

}  // namespace

// tile code 20.  Preconditions (3x3 / stride 1 / pad 1, CinP == 16, N <= 32, n_store <= 32, in2 with Cin2P == 16) are checked by vs_conv_gemm.
int vs_conv3x3_small_dispatch(const vs_conv_desc_t& d, hipStream_t st) {
  const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
  const int64_t nt = (int64_t)d.B * tiles_x * tiles_y;
  if (nt > 0x7fffffffLL) return VS_ERR_UNSUPPORTED;
  // persistent workgroups: 2 per CU x 256 CUs, each walks a contiguous run of tiles (weights are loaded once per workgroup)
  const int64_t want = 512 * 2;
  const int per = (int)((nt + want - 1) / want);
  const int tpw = per < 1 ? 1 : per;
  const unsigned grid = (unsigned)((nt + tpw - 1) / tpw);
  if (d.arith == 2) {
    if (d.in2) hipLaunchKernelGGL((conv3x3_small_kernel<true, 2>), dim3(grid), dim3(256), 0, st, d, tiles_x, tiles_y, (int)nt, tpw);
    else hipLaunchKernelGGL((conv3x3_small_kernel<false, 2>), dim3(grid), dim3(256), 0, st, d, tiles_x, tiles_y, (int)nt, tpw);
  } else if (d.in2)
    hipLaunchKernelGGL((conv3x3_small_kernel<true, 3>), dim3(grid), dim3(256), 0, st, d, tiles_x, tiles_y, (int)nt, tpw);
  else
    hipLaunchKernelGGL((conv3x3_small_kernel<false, 3>), dim3(grid), dim3(256), 0, st, d, tiles_x, tiles_y, (int)nt, tpw);
  return vs_launch_status();
}
