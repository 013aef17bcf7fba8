// ResnetBlock of the thin, full-resolution U-Net levels in ONE kernel (unet.py:24-39, eval BatchNorm folded):
//     out = relu(conv3x3_1(t) + b1) + (conv1x1_res(x) + b_res),     t = relu(conv3x3_0(x) + b0)
// for <= 16 input channels and 16 mid / output channels at 256^2 (VideoSeal 1.0: `inc` 1 -> 16 -> 16 and the last `ups` block 16 -> 16 -> 16).
// As two launches these layers are HBM-bound at a third of what the memory delivers: every conv reads a 134 MB map and writes one (32 frames),
// and `t` makes the round trip for nothing (profiles/r04b_stage_times_image.json: 137 + 168 us per block).  Here `t` never leaves the chip:
//   * workgroup = 8 x 16 output pixels; the 12 x 20 input patch (two-pixel halo) is split once into operand planes in LDS (double-buffered,
//     persistent workgroups: the next tile's patch is in flight during this tile's products);
//   * conv0 is evaluated on the 10 x 18 tile of `t` the second conv needs (one-pixel halo recomputed: 180 instead of 128 pixels), bias + ReLU +
//     operand split happen on the accumulator registers and the planes of `t` go to LDS -- positions outside the image are the zero padding of
//     conv1, not conv0 of padded data;
//   * conv1 and the 1x1 res_conv read `t` / the patch centre from LDS; every lane ends with four consecutive channels of one pixel: 16-byte stores,
//     a wave writes one contiguous KiB.
// Matrix instruction: v_mfma_f32_16x16x32 (16 channels x 16 pixels x K = 32): the 16 output channels fill the tile (the 32x32x16 form of
// conv3x3_small_kernel wastes half of it on N = 16), and K = 32 is TWO taps of 16 channels -- lanes 0-31 read the patch shifted by the first tap,
// lanes 32-63 by the second, so the nine taps are five instructions per partial product.  Per 128 output pixels and wave: 81 instructions of
// ~16 cycles against 108 of 32 for the two separate launches.  Same operand splits, scales and partial-product order as every other kernel of the
// arithmetic (Arith<NP>); the K order differs from the 32x32x16 kernels (tap pairs), so results agree with them to fp32 rounding, not bit for bit.
#include "conv_common.h"

namespace {

using namespace vsconv;

constexpr int TH = 8, TW = 16;
constexpr int T_W = TW + 2, T_H = TH + 2, T_PX = T_W * T_H;          // 18 x 10 = 180 pixels of t
constexpr int X_W = TW + 4, X_H = TH + 4, X_PX = X_W * X_H;          // 20 x 12 = 240 pixels of x
constexpr int XITEMS = X_PX * 4;                                     // float4 items of the patch (16 channels per pixel)
constexpr int NXI = (XITEMS + 255) / 256;                            // 4 per thread
constexpr int G0_PER_WAVE = 3;                                       // conv0: 12 groups of 16 t pixels (192 >= 180) over 4 waves
constexpr int G1_PER_WAVE = 2;                                       // conv1: 8 output rows of 16 pixels over 4 waves
// LDS row pitch of a pixel's 16 channels (one f16 plane).  ds_read_b128 serves a wave in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- over 64 banks.  With the 16x16x32 operand layout (lane = pixel n + 16 x k-block) a group
// therefore holds pixels {0-3, 12-15} of one k-block and pixels {4-11} of its neighbour: the 16 addresses cover all banks exactly once iff the
// pitch is 32 bytes mod 64 (checked exhaustively: 32, 96, 160 ...).  The 48-byte pitch of the 32x32x16 kernels (lane = row, conflict-free THERE)
// gave this kernel 55 % conflict cycles (profiles/r04q_pmc_sq_resblock_thin.json).
constexpr int ROWB16 = 32;

template <int NP> struct Arith16;
template <> struct Arith16<3> {
  static __device__ __forceinline__ f32x4 mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Arith16<2> {
  static __device__ __forceinline__ f32x4 mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

struct RbArgs {
  const float* x; int64_t x_sb, x_sy, x_sx; int B, H, W, Cin;
  const void *w0, *w1, *wr;                 // split planes [NP][16][144], [NP][16][144], [NP][16][16]
  const float *b0, *b1, *br;
  float a_mul, acc_mul0, acc_mul1, acc_mulr;
  float* out; int64_t out_ld;
};

__device__ __forceinline__ bf16x8 zero_frag() {
  const u32x4 z = {0u, 0u, 0u, 0u};
  return __builtin_bit_cast(bf16x8, z);
}

template <int NP, int NBUF, int OCC>
__global__ __launch_bounds__(256, OCC) void resblock_thin_kernel(const RbArgs d, const int tiles_x, const int tiles_y, const int ntiles_total,
                                                                 const int tiles_per_wg) {
  using AR = Arith<NP>;
  using A16 = Arith16<NP>;
  constexpr int XP_BYTES = NP * X_PX * ROWB16;                         // one patch buffer
  constexpr int TT_BYTES = NP * T_PX * ROWB16;
  // NBUF = 2: the next tile's patch lands in the other buffer (63 KB: two workgroups per CU); NBUF = 1 (three planes: static LDS stays below
  // 64 KB): one buffer + one more barrier per tile.
  __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * XP_BYTES + TT_BYTES];
  unsigned char* const Tt = smem + NBUF * XP_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int n = lane & 15, kb = lane >> 4;                           // operand row (channel / pixel) and 8-wide K block of this lane
  const int hi = kb >> 1;                                            // which tap of the pair
  const int kh = (kb & 1) * 8;                                       // channel offset inside the tap

  // ---- weights: A fragment of step s = row n, k = (tap 2s + hi, channels kh .. kh + 7); tap 9 does not exist -> zero
  bf16x8 w0f[5][NP], w1f[5][NP], wrf[NP];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int tap = 2 * s + hi;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (tap < 9) {
        w0f[s][p] = *reinterpret_cast<const bf16x8*>(static_cast<const char*>(d.w0) + ((int64_t)p * 16 * 144 + n * 144 + tap * 16 + kh) * 2);
        w1f[s][p] = *reinterpret_cast<const bf16x8*>(static_cast<const char*>(d.w1) + ((int64_t)p * 16 * 144 + n * 144 + tap * 16 + kh) * 2);
      } else {
        w0f[s][p] = zero_frag();
        w1f[s][p] = zero_frag();
      }
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p)
    wrf[p] = hi == 0 ? *reinterpret_cast<const bf16x8*>(static_cast<const char*>(d.wr) + ((int64_t)p * 16 * 16 + n * 16 + kh) * 2) : zero_frag();
  // accumulator layout D[channel][pixel]: lane = pixel n, registers e = channels 4 kb + e
  float b0v[4], b1v[4], brv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    b0v[e] = d.b0 ? d.b0[4 * kb + e] : 0.f;
    b1v[e] = d.b1 ? d.b1[4 * kb + e] : 0.f;
    brv[e] = d.br ? d.br[4 * kb + e] : 0.f;
  }

  // ---- patch items of this thread (positions relative to the tile origin are constant)
  int p_dy[NXI], p_dx[NXI], p_lds[NXI];
  bool p_have[NXI];
  const int k4 = (tid & 3) * 4;
  const bool cok = k4 < d.Cin;
#pragma unroll
  for (int i = 0; i < NXI; ++i) {
    const int item = tid + i * 256;
    p_have[i] = item < XITEMS;
    const int prow = p_have[i] ? item >> 2 : 0;
    p_dy[i] = prow / X_W - 2;
    p_dx[i] = prow % X_W - 2;
    p_lds[i] = prow * ROWB16 + k4 * 2;
  }
  // tap offsets of this lane's half of every tap pair (the ninth tap has no partner: its second half re-reads tap 8 against zero weights)
  int toffx[5], tofft[5];
#pragma unroll
  for (int s5 = 0; s5 < 5; ++s5) {
    const int tap = 2 * s5 + hi < 9 ? 2 * s5 + hi : 8;
    toffx[s5] = ((tap / 3) * X_W + tap % 3) * ROWB16;
    tofft[s5] = ((tap / 3) * T_W + tap % 3) * ROWB16;
  }
  // ---- fragment addresses (bytes inside a plane): conv0 group gi of this wave = t pixels 16 (3 wave + gi) + n
  int a0[G0_PER_WAVE], t_i[G0_PER_WAVE];
#pragma unroll
  for (int gi = 0; gi < G0_PER_WAVE; ++gi) {
    int i = 16 * (wave * G0_PER_WAVE + gi) + n;
    t_i[gi] = i;
    i = i < T_PX ? i : T_PX - 1;
    a0[gi] = ((i / T_W) * X_W + (i % T_W)) * ROWB16 + kh * 2;
  }
  int a1[G1_PER_WAVE], ar[G1_PER_WAVE];
#pragma unroll
  for (int gi = 0; gi < G1_PER_WAVE; ++gi) {
    const int row = wave * G1_PER_WAVE + gi;
    a1[gi] = (row * T_W + n) * ROWB16 + kh * 2;
    ar[gi] = ((row + 2) * X_W + (n + 2)) * ROWB16 + kh * 2;
  }
  const float amul = NP == 2 ? d.a_mul : 1.f;
  const float am0 = NP == 2 ? d.acc_mul0 : 1.f, am1 = NP == 2 ? d.acc_mul1 : 1.f, amr = NP == 2 ? d.acc_mulr : 1.f;

  f32x4 rp[NXI];
  auto load_tile = [&](const int fb, const int y0, const int x0) __attribute__((always_inline)) {
    // (round 6: ONE branch-free form.  The interior fast path of round 5 -- a workgroup-uniform `if` around its own loads -- and the `if (t + 1 < t_end)`
    //  around this call made hipcc wait for the prefetch at the join, in front of the tile's MFMAs: "in flight during this tile's products" was not true
    //  in the emitted code, tools/isa_scan.py order.  The clamped form costs a few integer operations per item and nothing else.)
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int iy = y0 + p_dy[i], ix = x0 + p_dx[i];
      const bool ok = p_have[i] && cok && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
      const int cy = iy < 0 ? 0 : (iy >= d.H ? d.H - 1 : iy), cx = ix < 0 ? 0 : (ix >= d.W ? d.W - 1 : ix);
      const f32x4 v = *reinterpret_cast<const f32x4*>(d.x + (int64_t)fb * d.x_sb + (int64_t)cy * d.x_sy + (int64_t)cx * d.x_sx + (cok ? k4 : 0));
      rp[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};               // unconditional load from a clamped address, zeroed afterwards (zero padding)
    }
  };
  auto store_tile = [&](const int buf) __attribute__((always_inline)) {
    unsigned char* Ps = smem + buf * XP_BYTES;
#pragma unroll
    for (int i = 0; i < NXI; ++i)
      if (p_have[i]) {
        u32x2 pl[NP];
        split4n<NP>(rp[i], amul, pl);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(Ps + p * X_PX * ROWB16 + p_lds[i]) = pl[p];
      }
  };

  const int t_begin = blockIdx.x * tiles_per_wg;
  const int t_end = min(ntiles_total, t_begin + tiles_per_wg);
  if (t_begin >= t_end) return;
  int tx = t_begin % tiles_x, ty = (t_begin / tiles_x) % tiles_y, fb = t_begin / (tiles_x * tiles_y);
  load_tile(fb, ty * TH, tx * TW);
  store_tile(0);
  __syncthreads();
  for (int t = t_begin; t < t_end; ++t) {
    const int buf = NBUF == 2 ? (t - t_begin) & 1 : 0;
    int ntx = tx + 1, nty = ty, nfb = fb;
    if (ntx == tiles_x) { ntx = 0; if (++nty == tiles_y) { nty = 0; ++nfb; } }
    {                                                           // in flight during this tile's products; the last tile re-reads itself (values unused)
      const bool more = t + 1 < t_end;
      load_tile(more ? nfb : fb, (more ? nty : ty) * TH, (more ? ntx : tx) * TW);
    }
    const unsigned char* Ps = smem + buf * XP_BYTES;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- conv0 on the 10 x 18 tile of t
#pragma unroll
    for (int gi = 0; gi < G0_PER_WAVE; ++gi) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int off = a0[gi] + toffx[s];
        bf16x8 xf[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) xf[p] = *reinterpret_cast<const bf16x8*>(Ps + p * X_PX * ROWB16 + off);
#pragma unroll
        for (int q = 0; q < AR::NPROD; ++q) acc = A16::mfma(w0f[s][AR::PB[q]], xf[AR::PA[q]], acc);
      }
      // t = relu(acc + b0) inside the image, 0 outside (conv1's zero padding); split into operand planes -> LDS
      const int i = t_i[gi];
      if (i < T_PX) {
        const int gy = y0 - 1 + i / T_W, gx = x0 - 1 + i % T_W;
        const bool in = gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = NP == 2 ? acc[e] * am0 : acc[e];
          v[e] = in ? vs_relu(a + b0v[e]) : 0.f;
        }
        u32x2 pl[NP];
        split4n<NP>(v, amul, pl);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(Tt + p * T_PX * ROWB16 + i * ROWB16 + 8 * kb) = pl[p];
      }
    }
    // two patch buffers: the next patch goes to LDS HERE, before this tile's output stores are issued -- behind them its `s_waitcnt vmcnt` would
    // also wait for the stores' completion (loads and stores share the counter); the other buffer was last read in tile t - 1
    if (NBUF == 2 && t + 1 < t_end) store_tile(buf ^ 1);
    __syncthreads();                                              // t complete

    // ---- conv1 + res_conv on the 8 x 16 output pixels
#pragma unroll
    for (int gi = 0; gi < G1_PER_WAVE; ++gi) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int off = a1[gi] + tofft[s];
        bf16x8 tf[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) tf[p] = *reinterpret_cast<const bf16x8*>(Tt + p * T_PX * ROWB16 + off);
#pragma unroll
        for (int q = 0; q < AR::NPROD; ++q) acc = A16::mfma(w1f[s][AR::PB[q]], tf[AR::PA[q]], acc);
      }
      {
        bf16x8 xf[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) xf[p] = *reinterpret_cast<const bf16x8*>(Ps + p * X_PX * ROWB16 + ar[gi]);
#pragma unroll
        for (int q = 0; q < AR::NPROD; ++q) accr = A16::mfma(wrf[AR::PB[q]], xf[AR::PA[q]], accr);
      }
      const int y = y0 + wave * G1_PER_WAVE + gi, x = x0 + n;
      if (y < d.H && x < d.W) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = NP == 2 ? acc[e] * am1 : acc[e];
          const float r = NP == 2 ? accr[e] * amr : accr[e];
          v[e] = vs_relu(a + b1v[e]) + (r + brv[e]);
        }
        *reinterpret_cast<f32x4*>(d.out + (((int64_t)fb * d.H + y) * d.W + x) * d.out_ld + 4 * kb) = v;
      }
    }
    if (NBUF == 1) {                              // single patch buffer: every wave must be done reading it
      __syncthreads();
      if (t + 1 < t_end) store_tile(0);
    }
    __syncthreads();                              // next patch visible; t may be overwritten
    tx = ntx; ty = nty; fb = nfb;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same block for 32 mid / output channels (<= 32 input channels): VideoSeal 1.0's 128^2 level (`downs.0.conv`, `ups.1.conv`), PixelSeal's
// `inc`.  K = 32 of the matrix instruction is ONE tap x 32 channels, the 32 output channels are two 16-row halves, and the weights
// (2 x 36 KB + 4 KB of f16 planes) no longer fit the register file next to the accumulators: they live in LDS, row-swizzled (16-byte unit
// kb ^ 2 (n >> 3) of the 64-byte row of output channel n: conflict-free fragment reads without padding), and a weight fragment is read once per
// tap and used for every pixel group of the wave.  155 KB of LDS: one persistent workgroup of EIGHT waves per CU (two per SIMD), one patch
// buffer (the next tile's patch waits in registers), three barriers per tile.  2 x f16 arithmetic only (three planes do not fit).
constexpr int ROWB32 = 96;                                  // 32 halves + 32 bytes of padding: pitch = 32 mod 64 (see ROWB16)
constexpr int NW32 = 8, NT32 = NW32 * 64;                   // eight waves: two per SIMD cover each other's LDS latencies (four waves: 122 us per block)
constexpr int XITEMS32 = X_PX * 8;                          // float4 items of the patch (32 channels per pixel)
constexpr int NXI32 = (XITEMS32 + NT32 - 1) / NT32;         // 4 per thread
constexpr int G0N = (12 + NW32 - 1) / NW32, G1N = 8 / NW32; // pixel groups per wave: conv0 groups w, w + 8 (< 12); conv1 row w
constexpr int XP32_BYTES = 2 * X_PX * ROWB32, TT32_BYTES = 2 * T_PX * ROWB32;
constexpr int W32_CONV = 9 * 2 * 2 * 16 * 64, W32_RES = 2 * 2 * 16 * 64;       // [tap][half][plane][n][64 B]

__global__ __launch_bounds__(NT32) void resblock_thin32_kernel(const RbArgs d, const int tiles_x, const int tiles_y, const int ntiles_total,
                                                               const int tiles_per_wg) {
  using AR = Arith<2>;
  using A16 = Arith16<2>;
  __shared__ __attribute__((aligned(16))) unsigned char smem[XP32_BYTES + TT32_BYTES + 2 * W32_CONV + W32_RES];
  unsigned char* const Ps = smem;
  unsigned char* const Tt = smem + XP32_BYTES;
  unsigned char* const Wl = Tt + TT32_BYTES;                // conv0 | conv1 | res
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kb = lane >> 4;

  // ---- weights -> LDS (once per workgroup)
  for (int u = tid; u < 2 * 2304; u += NT32) {
    const int conv = u / 2304, r = u - conv * 2304;
    const int tap = r >> 8, r2 = r & 255;
    const int half = r2 >> 7, plane = (r2 >> 6) & 1, nn = (r2 >> 2) & 15, kbu = r2 & 3;
    const char* src = static_cast<const char*>(conv ? d.w1 : d.w0) + ((int64_t)(plane * 32 + half * 16 + nn) * 288 + tap * 32 + kbu * 8) * 2;
    unsigned char* dst = Wl + conv * W32_CONV + ((((tap * 2 + half) * 2 + plane) * 16 + nn) * 64) + ((kbu ^ ((nn >> 3) << 1)) * 16);
    *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(src);
  }
  if (tid < 256) {
    const int u = tid;
    const int half = u >> 7, plane = (u >> 6) & 1, nn = (u >> 2) & 15, kbu = u & 3;
    const char* src = static_cast<const char*>(d.wr) + ((int64_t)(plane * 32 + half * 16 + nn) * 32 + kbu * 8) * 2;
    *reinterpret_cast<u32x4*>(Wl + 2 * W32_CONV + (((half * 2 + plane) * 16 + nn) * 64) + ((kbu ^ ((nn >> 3) << 1)) * 16)) = *reinterpret_cast<const u32x4*>(src);
  }
  const int aw = n * 64 + ((kb ^ ((n >> 3) << 1)) * 16);    // this lane's 16 bytes inside a [16][64 B] weight fragment (swizzle found by
                                                            // exhaustive search over the four b128 lane groups: conflict-free)
  float b0v[2][4], b1v[2][4], brv[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = h * 16 + 4 * kb + e;
      b0v[h][e] = d.b0 ? d.b0[c] : 0.f;
      b1v[h][e] = d.b1 ? d.b1[c] : 0.f;
      brv[h][e] = d.br ? d.br[c] : 0.f;
    }

  // ---- patch items of this thread: 8 float4 per pixel, item = tid + NT32 i -> pixel (tid >> 3) + 64 i, channels 4 (tid & 7) ..
  const int k4 = (tid & 7) * 4;
  const bool cok = k4 < d.Cin;
  int p_dy[NXI32], p_dx[NXI32], p_lds[NXI32];
  bool p_have[NXI32];
#pragma unroll
  for (int i = 0; i < NXI32; ++i) {
    const int prow = (tid >> 3) + (NT32 / 8) * i;
    p_have[i] = prow < X_PX;
    const int pr = p_have[i] ? prow : 0;
    p_dy[i] = pr / X_W - 2;
    p_dx[i] = pr % X_W - 2;
    p_lds[i] = pr * ROWB32 + k4 * 2;
  }
  int a0[G0N], t_i[G0N];
  bool g0_have[G0N];
#pragma unroll
  for (int gi = 0; gi < G0N; ++gi) {
    const int g = wave + gi * NW32;
    g0_have[gi] = g < 12;                                   // wave-uniform
    int i = 16 * (g0_have[gi] ? g : 0) + n;
    t_i[gi] = i;
    i = i < T_PX ? i : T_PX - 1;
    a0[gi] = ((i / T_W) * X_W + (i % T_W)) * ROWB32 + kb * 16;
  }
  const int row1 = wave;                                    // conv1: one output row of 16 pixels per wave (G1N == 1)
  const int a1 = (row1 * T_W + n) * ROWB32 + kb * 16;
  const int ar = ((row1 + 2) * X_W + (n + 2)) * ROWB32 + kb * 16;
  static_assert(G1N == 1, "one output row per wave");
  const float amul = d.a_mul, am0 = d.acc_mul0, am1 = d.acc_mul1, amr = d.acc_mulr;

  f32x4 rp[NXI32];
  auto load_tile = [&](const int fb, const int y0, const int x0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NXI32; ++i) {
      const int iy = y0 + p_dy[i], ix = x0 + p_dx[i];
      const bool ok = p_have[i] && cok && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
      const int cy = iy < 0 ? 0 : (iy >= d.H ? d.H - 1 : iy), cx = ix < 0 ? 0 : (ix >= d.W ? d.W - 1 : ix);
      const f32x4 v = *reinterpret_cast<const f32x4*>(d.x + (int64_t)fb * d.x_sb + (int64_t)cy * d.x_sy + (int64_t)cx * d.x_sx + (cok ? k4 : 0));
      rp[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NXI32; ++i)
      if (p_have[i]) {
        u32x2 pl[2];
        split4n<2>(rp[i], amul, pl);
#pragma unroll
        for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x2*>(Ps + p * X_PX * ROWB32 + p_lds[i]) = pl[p];
      }
  };

  const int t_begin = blockIdx.x * tiles_per_wg;
  const int t_end = min(ntiles_total, t_begin + tiles_per_wg);
  if (t_begin >= t_end) return;
  int tx = t_begin % tiles_x, ty = (t_begin / tiles_x) % tiles_y, fb = t_begin / (tiles_x * tiles_y);
  load_tile(fb, ty * TH, tx * TW);
  store_tile();
  __syncthreads();                                            // patch + weights visible
  for (int t = t_begin; t < t_end; ++t) {
    int ntx = tx + 1, nty = ty, nfb = fb;
    if (ntx == tiles_x) { ntx = 0; if (++nty == tiles_y) { nty = 0; ++nfb; } }
    {                                                           // (branch-free, as in resblock_thin_kernel: the last tile re-reads itself)
      const bool more = t + 1 < t_end;
      load_tile(more ? nfb : fb, (more ? nty : ty) * TH, (more ? ntx : tx) * TW);
    }
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- conv0 on the 10 x 18 tile of t: pixel groups wave, wave + 8 x 2 channel halves
    {
      f32x4 acc[G0N][2];
#pragma unroll
      for (int gi = 0; gi < G0N; ++gi)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[gi][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        bf16x8 wf[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int p = 0; p < 2; ++p) wf[h][p] = *reinterpret_cast<const bf16x8*>(Wl + ((tap * 2 + h) * 2 + p) * 1024 + aw);
        const int toff = ((tap / 3) * X_W + tap % 3) * ROWB32;
#pragma unroll
        for (int gi = 0; gi < G0N; ++gi) {
          if (!g0_have[gi]) continue;
          bf16x8 xf[2];
#pragma unroll
          for (int p = 0; p < 2; ++p) xf[p] = *reinterpret_cast<const bf16x8*>(Ps + p * X_PX * ROWB32 + a0[gi] + toff);
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < AR::NPROD; ++q) acc[gi][h] = A16::mfma(wf[h][AR::PB[q]], xf[AR::PA[q]], acc[gi][h]);
        }
      }
#pragma unroll
      for (int gi = 0; gi < G0N; ++gi) {
        const int i = t_i[gi];
        if (g0_have[gi] && i < T_PX) {
          const int gy = y0 - 1 + i / T_W, gx = x0 - 1 + i % T_W;
          const bool in = gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = in ? vs_relu(acc[gi][h][e] * am0 + b0v[h][e]) : 0.f;
            u32x2 pl[2];
            split4n<2>(v, amul, pl);
#pragma unroll
            for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x2*>(Tt + p * T_PX * ROWB32 + i * ROWB32 + (h * 16 + 4 * kb) * 2) = pl[p];
          }
        }
      }
    }
    __syncthreads();                                            // t complete

    // ---- conv1 + res_conv: output row `wave` x 2 channel halves
    f32x4 outv[2];
    {
      f32x4 acc[2], accr[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) { acc[h] = f32x4{0.f, 0.f, 0.f, 0.f}; accr[h] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int toff = ((tap / 3) * T_W + tap % 3) * ROWB32;
        bf16x8 tf[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) tf[p] = *reinterpret_cast<const bf16x8*>(Tt + p * T_PX * ROWB32 + a1 + toff);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          bf16x8 wf[2];
#pragma unroll
          for (int p = 0; p < 2; ++p) wf[p] = *reinterpret_cast<const bf16x8*>(Wl + W32_CONV + ((tap * 2 + h) * 2 + p) * 1024 + aw);
#pragma unroll
          for (int q = 0; q < AR::NPROD; ++q) acc[h] = A16::mfma(wf[AR::PB[q]], tf[AR::PA[q]], acc[h]);
        }
      }
      {
        bf16x8 xf[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) xf[p] = *reinterpret_cast<const bf16x8*>(Ps + p * X_PX * ROWB32 + ar);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          bf16x8 wf[2];
#pragma unroll
          for (int p = 0; p < 2; ++p) wf[p] = *reinterpret_cast<const bf16x8*>(Wl + 2 * W32_CONV + (h * 2 + p) * 1024 + aw);
#pragma unroll
          for (int q = 0; q < AR::NPROD; ++q) accr[h] = A16::mfma(wf[AR::PB[q]], xf[AR::PA[q]], accr[h]);
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) outv[h][e] = vs_relu(acc[h][e] * am1 + b1v[h][e]) + (accr[h][e] * amr + brv[h][e]);
    }
    __syncthreads();                              // every wave is done reading the patch
    if (t + 1 < t_end) store_tile();              // (waits for the patch loads only: this tile's output stores are issued below, behind it --
    __syncthreads();                              //  loads and stores share `vmcnt`)  next patch visible; t may be overwritten
    {
      const int y = y0 + row1, x = x0 + n;
      if (y < d.H && x < d.W) {
        float* orow = d.out + (((int64_t)fb * d.H + y) * d.W + x) * d.out_ld;
#pragma unroll
        for (int h = 0; h < 2; ++h) *reinterpret_cast<f32x4*>(orow + h * 16 + 4 * kb) = outv[h];
      }
    }
    tx = ntx; ty = nty; fb = nfb;
  }
}

}  // namespace

extern "C" int vs_resblock_thin_supported(int cin_ld, int cmid, int cout) {
  if (cin_ld < 4 || (cin_ld & 3)) return 0;
  // (32 channels: arith 2 only, and 155 KB of static LDS -- a device whose workgroups get 64 KiB takes the two-launch form the callers fall back to)
  if (cmid == 32 && cout == 32 && cin_ld <= 32) return vs_max_lds_bytes() >= (int)(XP32_BYTES + TT32_BYTES + 2 * W32_CONV + W32_RES) ? 1 : 0;
  return (cmid == 16 && cout == 16 && cin_ld <= 16) ? 1 : 0;
}

extern "C" int vs_resblock_thin(const vs_resblock_thin_desc_t* d, void* stream) {
  VS_REQUIRE(d && d->x && d->out && d->w0_split && d->w1_split && d->wr_split && d->B > 0 && d->H > 0 && d->W > 0);
  const int C = d->Cout == 32 ? 32 : 16;
  VS_REQUIRE(d->Cout == 0 || d->Cout == 16 || d->Cout == 32);
  VS_REQUIRE(vs_resblock_thin_supported(d->Cin, C, C) && d->x_ld >= d->Cin && (d->x_ld & 3) == 0 && d->out_ld >= C && (d->out_ld & 3) == 0);
  VS_REQUIRE(d->arith == 2 || (d->arith == 3 && C == 16));
  VS_REQUIRE((((uintptr_t)d->x | (uintptr_t)d->out | (uintptr_t)d->w0_split | (uintptr_t)d->w1_split | (uintptr_t)d->wr_split) & 15) == 0);
  RbArgs a{d->x, (int64_t)d->H * d->W * d->x_ld, (int64_t)d->W * d->x_ld, d->x_ld, d->B, d->H, d->W, d->Cin,
           d->w0_split, d->w1_split, d->wr_split, d->b0, d->b1, d->br, d->a_mul, d->acc_mul0, d->acc_mul1, d->acc_mulr, d->out, d->out_ld};
  const int tiles_x = (d->W + TW - 1) / TW, tiles_y = (d->H + TH - 1) / TH;
  const int64_t nt = (int64_t)d->B * tiles_x * tiles_y;
  if (nt > 0x7fffffffLL) return VS_ERR_UNSUPPORTED;
  // persistent workgroups, each walks a contiguous run of tiles (weights are loaded once per workgroup): two per CU for 16 channels (63 KB of LDS
  // each; a single patch buffer + three per CU measured slower: embed 6.33 against 6.14 ms, profiles/r04g_*), one per CU for 32 (145 KB)
  const int64_t want = (int64_t)vs_num_cus() * (C == 32 ? 1 : 2);
  const int per = (int)((nt + want - 1) / want);
  const int tpw = per < 1 ? 1 : per;
  const unsigned grid = (unsigned)((nt + tpw - 1) / tpw);
  if (C == 32) hipLaunchKernelGGL(resblock_thin32_kernel, dim3(grid), dim3(NT32), 0, (hipStream_t)stream, a, tiles_x, tiles_y, (int)nt, tpw);
  else if (d->arith == 2) hipLaunchKernelGGL((resblock_thin_kernel<2, 2, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a, tiles_x, tiles_y, (int)nt, tpw);
  else hipLaunchKernelGGL((resblock_thin_kernel<3, 1, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a, tiles_x, tiles_y, (int)nt, tpw);
  return vs_launch_status();
}
