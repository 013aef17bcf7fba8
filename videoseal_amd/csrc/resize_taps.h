// ATen-compatible 1-D interpolation taps shared by the resize kernels (shell.hip, aug.hip) and their adjoints (bwd_shell.hip): forward and
// backward evaluate the SAME weight expressions, so the adjoint is the exact transpose of the forward operator in fp32.
#pragma once
#include <hip/hip_runtime.h>

namespace vs_taps {

// ---- ATen-compatible 1-D interpolation taps (aten/native/cpu/UpSampleKernel.cpp semantics) -------------------
// antialias: triangle filter, support = max(scale,1), weights normalised over the taps that fall inside the input.
// plain    : 2-tap bilinear, align_corners=False, source index clamped at 0.
struct Taps {
  int lo, n;          // first input index, tap count
  float center, inv, total;   // antialias parameters
  float l1;           // plain: weight of the second tap
  bool aa;
};

__device__ __forceinline__ float tri(float x) { x = fabsf(x); return x < 1.f ? 1.f - x : 0.f; }

__device__ __forceinline__ Taps make_taps(int i, int in, int out, bool antialias) {
  Taps t;
  t.aa = antialias;
  const float scale = (float)in / (float)out;
  if (antialias) {
    const float support = scale >= 1.f ? scale : 1.f;
    t.center = scale * (i + 0.5f);
    t.inv = scale >= 1.f ? 1.f / scale : 1.f;
    int lo = (int)(t.center - support + 0.5f);
    lo = lo < 0 ? 0 : lo;
    int hi = (int)(t.center + support + 0.5f);
    hi = hi > in ? in : hi;
    t.lo = lo;
    t.n = hi - lo;
    float tot = 0.f;
    for (int j = 0; j < t.n; ++j) tot += tri((j + lo - t.center + 0.5f) * t.inv);
    t.total = tot;
    t.l1 = 0.f;
  } else {
    float src = scale * (i + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    int i0 = (int)src;
    i0 = i0 > in - 1 ? in - 1 : i0;
    t.lo = i0;
    t.n = 1 + (i0 < in - 1);
    t.l1 = src - i0;
    t.center = t.inv = t.total = 0.f;
  }
  return t;
}
// first input index and tap count only (what make_taps computes, without the weight total)
__device__ __forceinline__ void tap_range(int i, int in, int out, bool antialias, int& lo, int& n) {
  const float scale = (float)in / (float)out;
  if (antialias) {
    const float support = scale >= 1.f ? scale : 1.f;
    const float center = scale * (i + 0.5f);
    int l = (int)(center - support + 0.5f);
    l = l < 0 ? 0 : l;
    int h = (int)(center + support + 0.5f);
    h = h > in ? in : h;
    lo = l;
    n = h - l;
  } else {
    float src = scale * (i + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    int i0 = (int)src;
    i0 = i0 > in - 1 ? in - 1 : i0;
    lo = i0;
    n = 1 + (i0 < in - 1);
  }
}
__device__ __forceinline__ float tap_w(const Taps& t, int j) {
  if (t.aa) {
    const float w = tri((j + t.lo - t.center + 0.5f) * t.inv);
    return t.total != 0.f ? w / t.total : w;
  }
  if (t.n == 1) return 1.f;   // last row/column: ATen blends the border pixel with itself
  return j == 0 ? 1.f - t.l1 : t.l1;
}

// Adjoint side: the outputs o whose tap window can contain input index i lie in [o_lo, o_hi] (a conservative superset; the caller
// rebuilds make_taps(o) and tests lo <= i < lo + n, which keeps the weights bit-identical to the forward pass).
__device__ __forceinline__ void adjoint_range(int i, int in, int out, bool antialias, int& o_lo, int& o_hi) {
  const float scale = (float)in / (float)out;
  const float support = antialias ? (scale >= 1.f ? scale : 1.f) : 1.f;
  // forward: |i + 0.5 - scale * (o + 0.5)| <= support + 1 (one extra input pixel of slack for the integer rounding of lo / hi)
  float lo = ((float)i + 0.5f - support - 1.5f) / scale - 0.5f;
  float hi = ((float)i + 0.5f + support + 1.5f) / scale - 0.5f;
  int l = (int)floorf(lo) - 1, h = (int)ceilf(hi) + 1;
  if (!antialias && i <= 1) l = 0;          // plain bilinear clamps the source coordinate at 0: every early output reads pixels 0 (and 1)
  o_lo = l < 0 ? 0 : l;
  o_hi = h > out - 1 ? out - 1 : h;
}

}  // namespace vs_taps
