// Weight operand packing on the device: one launch turns a packed fp32 weight matrix [N][K] (k = (tap, channel), engine.pack_conv) into the two
// 16-bit images the split conv / GEMM kernels read -- the planes [P][N][K] and their LDS-image order [ceil(N/32)][K/16][P][64 slots][8]
// (engine.pack_blocked: chunk order (channel chunk, tap), slot = 2 r + (h ^ ((r >> 3) & 1))).  A training step re-packs every layer twice (forward
// weights + the transposed / flipped backward-data weights): as a dozen ATen ops per layer that was ~2000 launches and most of the step's host time.
#include "vs_common.h"

namespace {

template <int ARITH>
__global__ __launch_bounds__(256) void split_block_kernel(const float* __restrict__ wt, int N, int64_t K, int ntaps, float w_mul,
                                                          unsigned short* __restrict__ split, unsigned short* __restrict__ blk) {
  constexpr int P = ARITH == 2 ? 2 : 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int G = (N + 31) / 32;
  if (idx >= (int64_t)G * 32 * K) return;
  const int64_t k = idx % K;
  const int row = (int)(idx / K);
  const float v = row < N ? wt[(int64_t)row * K + k] : 0.f;
  unsigned short pl[P];
  if constexpr (ARITH == 2) {           // hi = f16(w * w_mul), lo = f16(w * w_mul - hi), round to nearest (engine.split_f16x2)
    const float ws = v * w_mul;
    const _Float16 hi = (_Float16)ws;
    const _Float16 lo = (_Float16)(ws - (float)hi);
    pl[0] = __builtin_bit_cast(unsigned short, hi);
    pl[1] = __builtin_bit_cast(unsigned short, lo);
  } else {                              // three bf16 terms by truncation, exact (engine.split_bf16x3)
    float r = v;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const unsigned b = __builtin_bit_cast(unsigned, r) & 0xffff0000u;
      pl[p] = (unsigned short)(b >> 16);
      r = r - __builtin_bit_cast(float, b);
    }
  }
  const int64_t cinp = K / ntaps;
  const int tap = (int)(k / cinp);
  const int c = (int)(k - (int64_t)tap * cinp);
  const int chunk = c >> 4, within = c & 15;
  const int64_t nch = K >> 4, ch = (int64_t)chunk * ntaps + tap;
  const int h = within >> 3, e = within & 7;
  const int g = row >> 5, r5 = row & 31;
  const int slot = 2 * r5 + (h ^ ((r5 >> 3) & 1));
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if (row < N) split[((int64_t)p * N + row) * K + k] = pl[p];
    blk[((((int64_t)g * nch + ch) * P + p) * 64 + slot) * 8 + e] = pl[p];
  }
}

// fp32 conv weights [N][Cin][KH][KW] -> the packed matrix [N'][KH*KW*CinP] the conv / GEMM kernels read, k = (ky * KW + kx) * CinP + c, channels
// >= Cin' zero (engine.pack_conv) -- one launch instead of zeros + permute + copy.  transpose = 1 packs the BACKWARD-DATA weights directly:
// rows = input channels, columns = output channels, taps flipped (W'[ci][co][ky][kx] = W[co][ci][KH-1-ky][KW-1-kx]; for a 1x1 layer the plain
// transpose of a Linear / GEMM matrix).
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ w, int Co, int Ci, int KH, int KW, int cinp, int transpose,
                                                        int64_t total, float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % cinp);
  const int64_t r = idx / cinp;
  const int t = (int)(r % (KH * KW));
  const int n = (int)(r / (KH * KW));
  const int ky = t / KW, kx = t - ky * KW;
  float v = 0.f;
  if (!transpose) {
    if (c < Ci) v = w[(((int64_t)n * Ci + c) * KH + ky) * KW + kx];
  } else {
    if (c < Co) v = w[(((int64_t)c * Ci + n) * KH + (KH - 1 - ky)) * KW + (KW - 1 - kx)];
  }
  out[idx] = v;
}

}  // namespace

extern "C" int vs_split_block(const float* wt, int N, int64_t K, int ntaps, int arith, float w_mul, void* split, void* blk, void* stream) {
  VS_REQUIRE(wt && split && blk && N > 0 && K > 0 && ntaps > 0 && (K % (16 * (int64_t)ntaps)) == 0 && (arith == 2 || arith == 3));
  const int64_t total = (int64_t)((N + 31) / 32) * 32 * K;
  const unsigned grid = (unsigned)cdiv64(total, 256);
  if (arith == 2)
    hipLaunchKernelGGL(split_block_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, wt, N, K, ntaps, w_mul, (unsigned short*)split,
                       (unsigned short*)blk);
  else
    hipLaunchKernelGGL(split_block_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, wt, N, K, ntaps, w_mul, (unsigned short*)split,
                       (unsigned short*)blk);
  return vs_launch_status();
}

extern "C" int vs_pack_conv(const float* w, int Co, int Ci, int KH, int KW, int cinp, int transpose, float* out, void* stream) {
  VS_REQUIRE(w && out && Co > 0 && Ci > 0 && KH > 0 && KW > 0 && cinp >= (transpose ? Co : Ci) && (transpose == 0 || transpose == 1));
  const int64_t total = (int64_t)(transpose ? Ci : Co) * KH * KW * cinp;
  hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, w, Co, Ci, KH, KW, cinp, transpose, total, out);
  return vs_launch_status();
}
