// Row-streaming separable anti-aliased / bilinear resize shared by shell.hip (`vs_resize_pre`: NHWC rgb + key frames) and aug.hip
// (`vs_aug_crop_resize_color`: Crop -> Resize -> colour ops, planar output).  See shell.hip for the history of the form (round 4) --
// a workgroup owns OW output columns x a strip of output rows and walks the INPUT rows of the strip top to bottom, eight at a time:
//   1. the eight rows x three planes x (OW scale + support) columns go to LDS with coalesced loads (the next eight are requested first);
//   2. horizontal pass: every (input row, output column) pair is filtered ONCE into a ring of horizontally resized rows;
//   3. vertical pass: every output row whose tap window is complete is combined from the ring and handed to the epilogue.
// The expressions and their order are those of the per-pixel kernels (r += wx * pixel over the taps, then acc += wy * r): bit-identical outputs.
// Round 6: (1) the rows go global -> registers -> LDS as 16-byte vectors from a 16-byte-aligned window start (12 loads + 12 `ds_write_b128` per
// thread and group instead of 48 + 48 scalar ones; frames whose width is not a multiple of four keep the scalar form); (2) a template over the
// tile: 128 output columns / <= 8 taps / 16 ring rows (scales up to 3.4: 768 -> 256) or 64 columns / <= 10 taps / 32 ring rows (scales up to 4:
// 1024 -> 256, BASELINE configs[4]); (3) the source may be a crop window (i0, j0, H x W) of a larger frame (srcH x srcW) -- the crop is an index
// offset, the cropped clip is never written; (4) the vertical tap weights of the strip's rows are tabulated ONCE per workgroup in LDS (a weight is a
// triangle and an IEEE division: evaluated per thread and output row they were ~a third of the kernel's instructions); (5) the store is a functor.
#pragma once
#include <hip/hip_runtime.h>
#include "resize_taps.h"
#include "vs_common.h"

namespace vs_rs {
using namespace vs_taps;

constexpr int RS_GI = 8;                                 // input rows per group
constexpr int RS_WTAB_ROWS = 64;                         // strip rows whose vertical weights are tabulated (taller strips evaluate them per row)
template <int OW_> struct RsGeo;
template <> struct RsGeo<128> { static constexpr int INW = 448, MT = 8, RING = 16; };
template <> struct RsGeo<64> { static constexpr int INW = 288, MT = 10, RING = 32; };
template <int OW> constexpr size_t rs_lds_bytes() {
  return (size_t)(RS_GI * 3 * RsGeo<OW>::INW + RsGeo<OW>::RING * 3 * OW + RS_WTAB_ROWS * (RsGeo<OW>::MT + 2)) * sizeof(float);
}
// which tile shape serves a resize (0: none -- the caller keeps its per-pixel kernel)
inline int rs_pick(int H, int W, int oh, int ow, int antialias) {
  const float sx = (float)W / (float)ow, sy = (float)H / (float)oh;
  const float supx = antialias ? (sx >= 1.f ? sx : 1.f) : 1.f, supy = antialias ? (sy >= 1.f ? sy : 1.f) : 1.f;
  auto fits = [&](int OW, int INW, int MT, int RING) {
    return 2.f * supx + 2.f <= (float)MT && 2.f * supy + 2.f <= (float)MT && (float)OW * sx + 2.f * supx + 4.f + 3.f <= (float)INW &&
           2.f * supy + 2.f + (float)RS_GI <= (float)RING;
  };
  if (fits(128, RsGeo<128>::INW, RsGeo<128>::MT, RsGeo<128>::RING)) return 128;
  if (fits(64, RsGeo<64>::INW, RsGeo<64>::MT, RsGeo<64>::RING)) return 64;
  return 0;
}

// src: frame b's three planes start at src + b * 3 * srcH * srcW; the resize reads the window rows [i0, i0 + H) x columns [j0, j0 + W) of them.
// epi(b, oy, ox, acc[3]) receives every output pixel of the tile exactly once.
template <int OW, class Epi>
__device__ __forceinline__ void resize_stream_body(float* __restrict__ rs_smem, const float* __restrict__ src, const int srcH, const int srcW,
                                                   const int i0, const int j0, const int H, const int W, const int oh, const int ow,
                                                   const int antialias, const int strip, const Epi& epi) {
  constexpr int INW = RsGeo<OW>::INW, MT = RsGeo<OW>::MT, RING = RsGeo<OW>::RING, NPART = 256 / OW, RPP = RS_GI / NPART;
  float* In = rs_smem;                                   // [RS_GI][3][INW]
  float* Hr = rs_smem + RS_GI * 3 * INW;                 // [RING][3][OW]
  float* Wy = Hr + RING * 3 * OW;                        // [RS_WTAB_ROWS][MT] vertical weights | [RS_WTAB_ROWS] first row | [RS_WTAB_ROWS] tap count
  int* WyLo = reinterpret_cast<int*>(Wy + RS_WTAB_ROWS * MT);
  int* WyN = WyLo + RS_WTAB_ROWS;
  const int ox0 = blockIdx.x * OW, oy0 = blockIdx.y * strip, b = blockIdx.z;
  const int oy_end = min(oh, oy0 + strip);
  const int64_t plane = (int64_t)srcH * srcW;
  const float* base = src + (int64_t)b * 3 * plane + (int64_t)i0 * srcW + j0;      // window origin inside the frame
  const int tid = threadIdx.x;
  const int oxl = tid & (OW - 1), part = tid / OW;
  const int ox = min(ox0 + oxl, ow - 1);
  const bool ox_ok = ox0 + oxl < ow;
  const bool wtab = strip <= RS_WTAB_ROWS;               // block-uniform
  if (wtab && tid < oy_end - oy0) {
    const Taps t = make_taps(oy0 + tid, H, oh, antialias);
    WyLo[tid] = t.lo;
    WyN[tid] = t.n;
    for (int q = 0; q < MT; ++q) Wy[tid * MT + q] = q < t.n ? tap_w(t, q) : 0.f;
  }
  // input window of the tile (window coordinates): columns [x_lo, x_lo + ww), rows [y_lo, y_end)
  int x_lo, x_hi, n_;
  tap_range(ox0, W, ow, antialias, x_lo, n_);
  tap_range(min(ox0 + OW - 1, ow - 1), W, ow, antialias, x_hi, n_);
  const bool vec = (srcW & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;    // block-uniform: 16-byte row pieces
  // LDS column 0 <-> window column xa: the 16-byte-aligned ABSOLUTE frame column at or below the window's first one (xa may lie up to three columns
  // left of the tile's window -- still inside the frame row: srcW % 4 == 0 keeps every piece of a row inside the row)
  const int xa = vec ? ((j0 + x_lo) & ~3) - j0 : x_lo;
  const int ww = x_hi + n_ - xa;
  const int nvec = (ww + 3) >> 2;                        // (ww <= INW - 3: rs_pick)
  int y_lo, y_hi;
  tap_range(oy0, H, oh, antialias, y_lo, n_);
  tap_range(oy_end - 1, H, oh, antialias, y_hi, n_);
  const int y_end = y_hi + n_;
  const Taps tx = make_taps(ox, W, ow, antialias);
  float wxs[MT];
#pragma unroll
  for (int jx = 0; jx < MT; ++jx) wxs[jx] = jx < tx.n ? tap_w(tx, jx) : 0.f;
  const int xoff = tx.lo - xa;

  // loads of one group: row-plane rp = 0 .. 23 (row = rp / 3, channel = rp % 3).  vec: thread (v = tid & 127, r2 = tid >> 7) owns the 16-byte
  // piece v of the row-planes r2, r2 + 2, ...; scalar: elements tid and tid + 256 of every row-plane
  const int vv = tid & 127, r2 = tid >> 7;
  const int vcl = vv < nvec ? vv : nvec - 1;              // lanes past the window: a valid address, never stored
  f32x4 prev[RS_GI * 3 / 2];
  float pres[RS_GI * 3][2];
  auto load_group = [&](const int r0) __attribute__((always_inline)) {
    if (vec) {
#pragma unroll
      for (int k = 0; k < RS_GI * 3 / 2; ++k) {
        const int rp = r2 + 2 * k;
        const int row = r0 + rp / 3, c = rp % 3;
        const int cy = row < y_end ? row : y_end - 1;                       // rows past the window: a valid address, never used
        const float* p = base + c * plane + (int64_t)cy * srcW + xa;
        prev[k] = *reinterpret_cast<const f32x4*>(p + 4 * vcl);
      }
    } else {
#pragma unroll
      for (int rp = 0; rp < RS_GI * 3; ++rp) {
        const int row = r0 + rp / 3, c = rp % 3;
        const int cy = row < y_end ? row : y_end - 1;
        const float* p = base + c * plane + (int64_t)cy * srcW + xa;
        pres[rp][0] = p[tid < ww ? tid : ww - 1];
        pres[rp][1] = p[tid + 256 < ww ? tid + 256 : ww - 1];
      }
    }
  };
  int oy_next = oy0 + part;                               // next output row this thread finalises (the thread parts take rows in turn)
  load_group(y_lo);
  for (int r0 = y_lo; r0 < y_end; r0 += RS_GI) {
    if (vec) {
      if (vv < nvec) {
#pragma unroll
        for (int k = 0; k < RS_GI * 3 / 2; ++k) *reinterpret_cast<f32x4*>(In + (r2 + 2 * k) * INW + 4 * vv) = prev[k];
      }
    } else {
#pragma unroll
      for (int rp = 0; rp < RS_GI * 3; ++rp) {
        if (tid < ww) In[rp * INW + tid] = pres[rp][0];
        if (tid + 256 < ww) In[rp * INW + tid + 256] = pres[rp][1];
      }
    }
    if (r0 + RS_GI < y_end) load_group(r0 + RS_GI);       // in flight during the two passes below
    __syncthreads();
    // horizontal pass: this thread's column, rows part * RPP .. + RPP - 1 of the group
#pragma unroll
    for (int q = 0; q < RPP; ++q) {
      const int rl = part * RPP + q;
      const int row = r0 + rl;
      if (row < y_end) {
        float r[3] = {0.f, 0.f, 0.f};
        const float* Lp = In + rl * 3 * INW + xoff;
#pragma unroll
        for (int jx = 0; jx < MT; ++jx)
          if (jx < tx.n) {
#pragma unroll
            for (int c = 0; c < 3; ++c) r[c] = __builtin_fmaf(wxs[jx], Lp[c * INW + jx], r[c]);
          }
#pragma unroll
        for (int c = 0; c < 3; ++c) Hr[((row & (RING - 1)) * 3 + c) * OW + oxl] = r[c];
      }
    }
    __syncthreads();
    // vertical pass: output rows whose window ends inside the rows processed so far
    const int done = min(r0 + RS_GI, y_end);
    while (oy_next < oy_end) {
      float acc[3] = {0.f, 0.f, 0.f};
      if (wtab) {
        const int tl = WyLo[oy_next - oy0], tn = WyN[oy_next - oy0];
        if (tl + tn > done) break;
        const float* wyp = Wy + (oy_next - oy0) * MT;
        for (int jy = 0; jy < tn; ++jy) {
          const float wy = wyp[jy];
          const float* hp = Hr + (((tl + jy) & (RING - 1)) * 3) * OW + oxl;
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[c] = __builtin_fmaf(wy, hp[c * OW], acc[c]);
        }
      } else {
        const Taps ty = make_taps(oy_next, H, oh, antialias);
        if (ty.lo + ty.n > done) break;
        for (int jy = 0; jy < ty.n; ++jy) {
          const float wy = tap_w(ty, jy);
          const float* hp = Hr + (((ty.lo + jy) & (RING - 1)) * 3) * OW + oxl;
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[c] = __builtin_fmaf(wy, hp[c * OW], acc[c]);
        }
      }
      if (ox_ok) epi(b, oy_next, ox, acc);
      oy_next += NPART;
    }
  }
}

}  // namespace vs_rs
