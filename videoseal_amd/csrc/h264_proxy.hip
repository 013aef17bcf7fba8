// H.264-style transform-coding PROXY for the video-compression augmentations (augmentation/video.py:20-205: VideoCompression, H264,
// H264rgb, H265).  The reference shells out to libx264 / libx265 through PyAV on the CPU (video.py:48-86); that codec is not part of
// the reference's source, is not installed offline and has no arithmetic to restate.  What this file
// implements instead is DEFINED here (and restated in oracle/h264_proxy.py, against which it is bit-exact):
//   frames -> clamp -> uint8 (truncation, video.py:38-40) -> [yuv420: integer BT.601 limited-range YCbCr, 2x2 rounded-mean chroma]
//   -> every 4x4 block of every plane: residual against a flat 128 prediction -> H.264 forward core transform (Cf X Cf^T) ->
//   H.264 quantisation at QP (MF table, intra dead zone f = 2^qbits / 3) -> H.264 de-quantisation (V table) -> inverse core transform
//   ((x + 32) >> 6) -> + 128, clip -> [yuv420: nearest chroma up-sampling, integer YCbCr -> RGB] -> / 255.
// QP = clamp(crf, 0, 51) for luma (x264's CRF is a QP-domain target; for intra frames of average complexity QP ~ crf), the chroma QP
// follows the standard's QPc table.  There is NO intra/inter prediction, NO deblocking filter and NO rate control: the proxy injects
// the codec's characteristic distortion (4x4 integer-DCT quantisation noise at the CRF's step size, 4:2:0 chroma loss) and stays
// on the GPU; it is not a bit-stream-exact model of libx264.  Frames are padded to multiples of 8 by edge replication.
#include <algorithm>

#include "vs_common.h"

namespace {

__constant__ int kMF[6][3] = {{13107, 5243, 8066}, {11916, 4660, 7490}, {10082, 4194, 6554}, {9362, 3647, 5825}, {8192, 3355, 5243}, {7282, 2893, 4559}};
__constant__ int kV[6][3] = {{10, 16, 13}, {11, 18, 14}, {13, 20, 16}, {14, 23, 18}, {16, 25, 20}, {18, 29, 23}};

__device__ __forceinline__ int clip255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int to_u8(float v) { return (int)(fminf(fmaxf(v, 0.f), 1.f) * 255.0f); }     // clamp, *255, truncation

// mode 0: Y [F][H8][W8], Cb / Cr [F][H8/2][W8/2];  mode 1 (libx264rgb): R, G, B planes [F][H8][W8]
__global__ __launch_bounds__(256) void h264_to_planes_kernel(const float* __restrict__ src, int H, int W, int H8, int W8, int mode,
                                                             unsigned char* __restrict__ p0, unsigned char* __restrict__ p1,
                                                             unsigned char* __restrict__ p2, int64_t total) {
  const int hw2 = (H8 / 2) * (W8 / 2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int f = (int)(i / hw2), r = (int)(i % hw2);
    const int cy = r / (W8 / 2), cx = r % (W8 / 2);
    const float* fr = src + (int64_t)f * 3 * H * W;
    int cb = 0, cr = 0;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int y = 2 * cy + dy, x = 2 * cx + dx;
        const int64_t sp = (int64_t)min(y, H - 1) * W + min(x, W - 1);           // edge replication into the padding
        const int R = to_u8(fr[sp]), G = to_u8(fr[(int64_t)H * W + sp]), B = to_u8(fr[2 * (int64_t)H * W + sp]);
        const int64_t dp = ((int64_t)f * H8 + y) * W8 + x;
        if (mode == 1) {
          p0[dp] = (unsigned char)R; p1[dp] = (unsigned char)G; p2[dp] = (unsigned char)B;
        } else {
          p0[dp] = (unsigned char)(((66 * R + 129 * G + 25 * B + 128) >> 8) + 16);
          cb += ((-38 * R - 74 * G + 112 * B + 128) >> 8) + 128;
          cr += ((112 * R - 94 * G - 18 * B + 128) >> 8) + 128;
        }
      }
    if (mode == 0) {
      p1[(int64_t)f * hw2 + r] = (unsigned char)((cb + 2) >> 2);
      p2[(int64_t)f * hw2 + r] = (unsigned char)((cr + 2) >> 2);
    }
  }
}

// one thread per 4x4 block of a plane set [planes][ph][pw] (in place)
__global__ __launch_bounds__(256) void h264_block_tq_kernel(unsigned char* __restrict__ plane, int ph, int pw, int qp, int64_t nblocks) {
  const int bw = pw / 4, bpp = (ph / 4) * bw;
  const int qm = qp % 6, qd = qp / 6, qbits = 15 + qd, f = (1 << qbits) / 3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nblocks; i += (int64_t)gridDim.x * 256) {
    const int64_t pl = i / bpp;
    const int r = (int)(i % bpp);
    unsigned char* b0 = plane + pl * ph * pw + (int64_t)(r / bw) * 4 * pw + (r % bw) * 4;
    int x[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const unsigned int v = *reinterpret_cast<const unsigned int*>(b0 + (int64_t)a * pw);
#pragma unroll
      for (int c = 0; c < 4; ++c) x[a][c] = (int)((v >> (8 * c)) & 255u) - 128;
    }
    // forward core transform W = Cf X Cf^T, Cf = [1 1 1 1; 2 1 -1 -2; 1 -1 -1 1; 1 -2 2 -1]
    int t[4][4], w[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {             // rows of X: t = X Cf^T
      const int s0 = x[a][0] + x[a][3], s1 = x[a][1] + x[a][2], d0 = x[a][0] - x[a][3], d1 = x[a][1] - x[a][2];
      t[a][0] = s0 + s1; t[a][1] = 2 * d0 + d1; t[a][2] = s0 - s1; t[a][3] = d0 - 2 * d1;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {             // columns: w = Cf t
      const int s0 = t[0][c] + t[3][c], s1 = t[1][c] + t[2][c], d0 = t[0][c] - t[3][c], d1 = t[1][c] - t[2][c];
      w[0][c] = s0 + s1; w[1][c] = 2 * d0 + d1; w[2][c] = s0 - s1; w[3][c] = d0 - 2 * d1;
    }
    // quantise + de-quantise; position class: both indices even -> 0, both odd -> 1, else 2
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int cls = ((a & 1) == 0 && (c & 1) == 0) ? 0 : (((a & 1) == 1 && (c & 1) == 1) ? 1 : 2);
        const int av = w[a][c] < 0 ? -w[a][c] : w[a][c];
        const int z = (av * kMF[qm][cls] + f) >> qbits;
        const int dq = (z * kV[qm][cls]) << qd;
        w[a][c] = w[a][c] < 0 ? -dq : dq;
      }
    // inverse core transform (rows, then columns), (x + 32) >> 6
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int e0 = w[a][0] + w[a][2], e1 = w[a][0] - w[a][2], e2 = (w[a][1] >> 1) - w[a][3], e3 = w[a][1] + (w[a][3] >> 1);
      t[a][0] = e0 + e3; t[a][1] = e1 + e2; t[a][2] = e1 - e2; t[a][3] = e0 - e3;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e0 = t[0][c] + t[2][c], e1 = t[0][c] - t[2][c], e2 = (t[1][c] >> 1) - t[3][c], e3 = t[1][c] + (t[3][c] >> 1);
      x[0][c] = e0 + e3; x[1][c] = e1 + e2; x[2][c] = e1 - e2; x[3][c] = e0 - e3;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      unsigned int v = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) v |= (unsigned int)clip255(((x[a][c] + 32) >> 6) + 128) << (8 * c);
      *reinterpret_cast<unsigned int*>(b0 + (int64_t)a * pw) = v;
    }
  }
}

__global__ __launch_bounds__(256) void h264_to_rgb_kernel(const unsigned char* __restrict__ p0, const unsigned char* __restrict__ p1,
                                                          const unsigned char* __restrict__ p2, int H, int W, int H8, int W8, int mode,
                                                          float* __restrict__ dst, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int64_t f = i / ((int64_t)W * H);
    const int64_t lp = (f * H8 + y) * W8 + x;
    int R, G, B;
    if (mode == 1) {
      R = p0[lp]; G = p1[lp]; B = p2[lp];
    } else {
      const int64_t cp = (f * (H8 / 2) + y / 2) * (W8 / 2) + x / 2;
      const int C = (int)p0[lp] - 16, D = (int)p1[cp] - 128, E = (int)p2[cp] - 128;
      R = clip255((298 * C + 409 * E + 128) >> 8);
      G = clip255((298 * C - 100 * D - 208 * E + 128) >> 8);
      B = clip255((298 * C + 516 * D + 128) >> 8);
    }
    float* o = dst + f * 3 * H * W + (int64_t)y * W + x;
    o[0] = (float)R / 255.0f; o[(int64_t)H * W] = (float)G / 255.0f; o[2 * (int64_t)H * W] = (float)B / 255.0f;
  }
}

static const int kQPcHost[22] = {29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};

inline unsigned gridn(int64_t n) { return (unsigned)std::min<int64_t>(cdiv64(n, 256), 1 << 16); }

}  // namespace

extern "C" int64_t vs_h264_proxy_workspace_bytes(int F, int H, int W) {
  const int64_t H8 = (H + 7) / 8 * 8, W8 = (W + 7) / 8 * 8;
  return (int64_t)F * H8 * W8 * 3 + 256;
}

extern "C" int vs_h264_proxy_roundtrip(const float* src, float* dst, int F, int H, int W, int qp, int rgb_mode, void* workspace,
                                       void* stream) {
  VS_REQUIRE(src && dst && workspace && F > 0 && H > 0 && W > 0 && qp >= 0 && qp <= 51);
  const int H8 = (H + 7) / 8 * 8, W8 = (W + 7) / 8 * 8;
  hipStream_t st = (hipStream_t)stream;
  unsigned char* p0 = static_cast<unsigned char*>(workspace);
  const int64_t lum = (int64_t)F * H8 * W8, chr = lum / 4;
  unsigned char* p1 = p0 + lum;
  unsigned char* p2 = p1 + (rgb_mode ? lum : chr);
  const int64_t ncell = (int64_t)F * (H8 / 2) * (W8 / 2);
  hipLaunchKernelGGL(h264_to_planes_kernel, dim3(gridn(ncell)), dim3(256), 0, st, src, H, W, H8, W8, rgb_mode ? 1 : 0, p0, p1, p2, ncell);
  if (rgb_mode) {
    const int64_t nb = 3 * lum / 16;
    hipLaunchKernelGGL(h264_block_tq_kernel, dim3(gridn(nb)), dim3(256), 0, st, p0, H8, W8, qp, nb);
  } else {
    const int qpc = qp < 30 ? qp : kQPcHost[qp - 30];
    hipLaunchKernelGGL(h264_block_tq_kernel, dim3(gridn(lum / 16)), dim3(256), 0, st, p0, H8, W8, qp, lum / 16);
    hipLaunchKernelGGL(h264_block_tq_kernel, dim3(gridn(2 * chr / 16)), dim3(256), 0, st, p1, H8 / 2, W8 / 2, qpc, 2 * chr / 16);
  }
  const int64_t npx = (int64_t)F * H * W;
  hipLaunchKernelGGL(h264_to_rgb_kernel, dim3(gridn(npx)), dim3(256), 0, st, p0, p1, p2, H, W, H8, W8, rgb_mode ? 1 : 0, dst, npx);
  return vs_launch_status();
}
