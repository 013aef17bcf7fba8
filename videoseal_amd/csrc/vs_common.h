// Internal helpers shared by the HIP translation units of libvideoseal_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "videoseal_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VS_REQUIRE(cond)                  \
  do {                                    \
    if (!(cond)) return VS_ERR_BAD_ARG;   \
  } while (0)

static inline int vs_launch_status() { return hipGetLastError() == hipSuccess ? VS_OK : VS_ERR_LAUNCH; }

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float vs_apply_act(float v, int act) {
  switch (act) {
    case VS_ACT_RELU: return v > 0.f ? v : 0.f;
    case VS_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    case VS_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// 64-lane butterfly sum (every lane ends with the total)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
