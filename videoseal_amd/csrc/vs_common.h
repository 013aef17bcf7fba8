// Internal helpers shared by the HIP translation units of libvideoseal_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "videoseal_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VS_REQUIRE(cond)                  \
  do {                                    \
    if (!(cond)) return VS_ERR_BAD_ARG;   \
  } while (0)

static inline int vs_launch_status() { return hipGetLastError() == hipSuccess ? VS_OK : VS_ERR_LAUNCH; }

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Default arithmetic of the split conv / GEMM path (vs_conv_desc_t::arith): 3 = "3 x bf16" (exact operand split, 6 products),
// 2 = "2 x f16" (round-to-nearest split with power-of-two range scaling, 3 products; conv_common.h).  VIDEOSEAL_CONV=f16x2 | bf16x3
// overrides it for the model-level entry points (the operator level takes it from the descriptor).
#define VS_DEFAULT_ARITH 2
static inline int vs_default_arith() {
  static const int a = [] {
    const char* e = getenv("VIDEOSEAL_CONV");
    if (e && !strcmp(e, "f16x2")) return 2;
    if (e && !strcmp(e, "bf16x3")) return 3;
    return VS_DEFAULT_ARITH;
  }();
  return a;
}

// Development switches of the launchers (tests / tools select a kernel form or a strip height PER CALL): set through the exported
// vs_debug_set(key, value) -- an atomic the launch path reads; nothing on a launch path calls getenv (not thread-safe against setenv, and a
// test hook in production code).  0 = the library's own choice.
enum { VS_DBG_RESIZE_FORM = 0 /* 1 = 32 x 8 tile kernel */, VS_DBG_RESIZE_STRIP = 1 /* output rows */, VS_DBG_TAIL_STRIP = 2 /* rows */, VS_DBG_JPEG_FORM = 3 /* 1 = scalar kernels */, VS_DBG_CROP_RESIZE_FORM = 4 /* 1 = 32 x 8 tile kernel */, VS_DBG_TO_PLANES_FORM = 5 /* 1 = one row per wave */, VS_DBG_DWCONV_ROWS = 6 /* rows per strip of the one-row depthwise kernel */, VS_DBG_LN_FORM = 7 /* 1 = wave per row */, VS_DBG_SPLITK_EPI = 8 /* 1 = the first (scalar-load) K-slice epilogue kernel */, VS_DBG_COUNT = 9 };
int vs_debug_get(int key);     // api.hip

// LDS a workgroup may ask for on the current device (160 KiB on gfx950, 64 KiB on earlier CDNA parts): launchers whose kernels want more than
// 64 KiB check it and fall back to their small-LDS form instead of failing the launch
static inline int vs_max_lds_bytes() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) v = 64 * 1024;
    return v;
  }();
  return n;
}

// compute units of the current device (256 on MI355X): grid size of the persistent kernels
static inline int vs_num_cus() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}

// Exact-erf GELU (nn.GELU default, convnext.py:49) in ~16 VALU instructions, branch-free (ocml's erff is two divergent
// branches, ~40 instructions, and made the pwconv1 epilogues VALU-bound):
//   gelu(v) = max(v, 0) - 0.5 |v| erfc(|v|/sqrt2),   erfc(t) = exp(-P(t)),  P(t) = t * Q8(t) fitted to -ln erfc on [0, 4]
// with the 1/sqrt2 and -log2(e) factors folded into the coefficients (tools/micro/fit_gelu.py).  Max |error| vs the exact
// function 2.4e-7 over [-9, 9] -- ATen's own fp32 GELU is off by up to 1.2e-6 there -- and exactly v / 0 beyond |v| = 5.66.
// erfc(a / sqrt2) for a >= 0 (the fitted factor of vs_gelu; shared with its derivative in bwd_ops.hip)
__device__ __forceinline__ float vs_erfc_sqrt2(float a) {
  const float u = fminf(a, 5.65685424949238f);
  float q = 5.128553084e-07f;
  q = __builtin_fmaf(q, u, -9.560153558e-06f);
  q = __builtin_fmaf(q, u, 7.497344632e-05f);
  q = __builtin_fmaf(q, u, -2.843466646e-04f);
  q = __builtin_fmaf(q, u, 1.498938855e-05f);
  q = __builtin_fmaf(q, u, 6.931120995e-03f);
  q = __builtin_fmaf(q, u, -5.243476480e-02f);
  q = __builtin_fmaf(q, u, -4.592214525e-01f);
  q = __builtin_fmaf(q, u, -1.151104212e+00f);
  return __builtin_amdgcn_exp2f(q * u);
}
__device__ __forceinline__ float vs_gelu(float v) {
  const float a = fabsf(v);
  const float e = vs_erfc_sqrt2(a);
  return fmaxf(v, 0.f) - (0.5f * a) * e;
}
// vs_gelu on two values at once (round 6: the GEMM epilogues; the fused ConvNeXt block has had its own copy since round 4): polynomial and blends as
// 2-wide fp32 operations -- v_pk_fma_f32 / v_pk_mul_f32, two results per VALU slot, 23 instead of 40 instructions per pair --, only min / max / exp2
// stay scalar.  Same coefficients and operation order; every multiply-add is an explicit fma with contraction off, which is what hipcc makes of
// vs_gelu's `max - (0.5 a) e` -> the same values bit for bit (tests/test_gpu_kernels.py compares the epilogues that use it with those that do not).
typedef float vs_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ vs_f32x2 vs_gelu2(const vs_f32x2 v) {
#pragma clang fp contract(off)
  const vs_f32x2 a = {fabsf(v[0]), fabsf(v[1])};
  const vs_f32x2 u = {fminf(a[0], 5.65685424949238f), fminf(a[1], 5.65685424949238f)};
  vs_f32x2 q = {5.128553084e-07f, 5.128553084e-07f};
  q = __builtin_elementwise_fma(q, u, vs_f32x2{-9.560153558e-06f, -9.560153558e-06f});
  q = __builtin_elementwise_fma(q, u, vs_f32x2{7.497344632e-05f, 7.497344632e-05f});
  q = __builtin_elementwise_fma(q, u, vs_f32x2{-2.843466646e-04f, -2.843466646e-04f});
  q = __builtin_elementwise_fma(q, u, vs_f32x2{1.498938855e-05f, 1.498938855e-05f});
  q = __builtin_elementwise_fma(q, u, vs_f32x2{6.931120995e-03f, 6.931120995e-03f});
  q = __builtin_elementwise_fma(q, u, vs_f32x2{-5.243476480e-02f, -5.243476480e-02f});
  q = __builtin_elementwise_fma(q, u, vs_f32x2{-4.592214525e-01f, -4.592214525e-01f});
  q = __builtin_elementwise_fma(q, u, vs_f32x2{-1.151104212e+00f, -1.151104212e+00f});
  const vs_f32x2 t = q * u;
  const vs_f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  const vs_f32x2 m = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)};
  return __builtin_elementwise_fma(a * -0.5f, e, m);
}
// d gelu / dv = Phi(v) + v phi(v),  Phi(v) = 1 - erfc(v / sqrt2) / 2
__device__ __forceinline__ float vs_gelu_grad(float v) {
  const float e = vs_erfc_sqrt2(fabsf(v));
  const float Phi = v >= 0.f ? 1.f - 0.5f * e : 0.5f * e;
  return Phi + v * (0.3989422804014327f * __expf(-0.5f * v * v));
}

// ReLU that lets NaN through (fmaxf / `v > 0 ? v : 0` turn it into 0): an out-of-range operand of the 2 x f16 arithmetic must surface as
// NaN in the output, not as a silently wrong finite number
__device__ __forceinline__ float vs_relu(float v) { return v <= 0.f ? 0.f : v; }

__device__ __forceinline__ float vs_apply_act(float v, int act) {
  switch (act) {
    case VS_ACT_RELU: return vs_relu(v);
    case VS_ACT_GELU: return vs_gelu(v);
    case VS_ACT_TANH: return tanhf(v);
    case VS_ACT_SILU: return v / (1.0f + __expf(-v));       // nn.SiLU: x * sigmoid(x)
    default: return v;
  }
}

// 64-lane butterfly sum (every lane ends with the total)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
