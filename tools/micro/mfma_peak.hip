// Micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate with register-only operands (the ceiling the conv kernels are priced against).
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC, int MODE>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, int lds_reads, int rnd) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = rnd ? ((i * 2654435761u) >> 3 & 0x3fff3fffu) | 0x30003000u : 0x3f803f80u;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  bf16x8 x, y;
  for (int e = 0; e < 8; ++e) {
    if (rnd) {   // random mantissas / signs in [-1, 1): realistic toggle rate (data-dependent power -> clocks)
      unsigned h1 = (lane * 8 + e + 1) * 2654435761u, h2 = (lane * 8 + e + 77) * 2246822519u;
      h1 ^= h1 >> 15; h1 *= 2246822519u; h1 ^= h1 >> 13; h2 ^= h2 >> 15; h2 *= 2654435761u; h2 ^= h2 >> 13;
      x[e] = (__bf16)(((int)(h1 >> 8) - (1 << 23)) * (1.0f / (1 << 23)));
      y[e] = (__bf16)(((int)(h2 >> 8) - (1 << 23)) * (1.0f / (1 << 23)) * 0.01f);
    } else { x[e] = (__bf16)(float)(lane + e); y[e] = (__bf16)(float)(lane - e); }
  }
  bf16x8 fr[12];
  for (int q = 0; q < 12; ++q) fr[q] = x;
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 1) {
#pragma unroll
      for (int q = 0; q < 12; ++q)
        if (q < lds_reads) fr[q] = *reinterpret_cast<const bf16x8*>(smem + ((it * 12 + q) * 1024 & 0xffff) + lane * 16);
    }
#pragma unroll
    for (int q = 0; q < 24 / NACC; ++q)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MODE >= 1 ? fr[(q + a) % 12] : x, MODE >= 1 ? fr[(q * 5 + a + 3) % 12] : y, acc[a], 0, 0, 0);
    if (MODE == 2) __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  if (s == 123.456f) out[threadIdx.x] = s;
}

extern "C" int run_mfma(int mode, int blocks, int threads, int iters, int lds_reads, int rnd, float* out, float* ms_out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    if (mode == 0) hipLaunchKernelGGL((mfma_loop<4, 0>), dim3(blocks), dim3(threads), 0, 0, out, iters, lds_reads, rnd);
    else if (mode == 1) hipLaunchKernelGGL((mfma_loop<4, 1>), dim3(blocks), dim3(threads), 0, 0, out, iters, lds_reads, rnd);
    else hipLaunchKernelGGL((mfma_loop<4, 2>), dim3(blocks), dim3(threads), 0, 0, out, iters, lds_reads, rnd);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
  }
  hipEventElapsedTime(ms_out, e0, e1);
  return (int)hipGetLastError();
}
