// Bilinear sampling taps shared by the warp kernels (warp.hip) and the fused census + warp kernels (photo.hip):
// the arithmetic of torch's grid_sample as the reference calls it (ATen/native/GridSampler.h:27-83 through
// utils/warp_utils.py:83-90 and utils/uflow_utils.py:53-77), computed once per pixel.
#pragma once
#include "common.hpp"

namespace {

struct Taps {
  int x0, y0;           // north-west corner
  float wx0, wx1, wy0, wy1;
  bool vx0, vx1, vy0, vy1;  // corner inside the source
  float dx, dy;         // d coord / d flow (0 where border clamping is active)
};

__device__ __forceinline__ Taps make_taps(float px, float py, float u, float v, int H, int W, int Hs,
                                          int Ws, int pad, bool align, int norm) {
  Taps t;
  float ix = af_sample_coord(px, u, W, Ws, norm, align, &t.dx);
  float iy = af_sample_coord(py, v, H, Hs, norm, align, &t.dy);
  if (pad == ARFLOW_PAD_BORDER) {
    ix = af_clip_border(ix, Ws, &t.dx);
    iy = af_clip_border(iy, Hs, &t.dy);
  }
  const float fx = floorf(ix), fy = floorf(iy);
  t.wx1 = ix - fx;
  t.wx0 = (fx + 1.f) - ix;
  t.wy1 = iy - fy;
  t.wy0 = (fy + 1.f) - iy;
  // comparisons in float first: NaN / huge coordinates fall out as "outside"
  t.vx0 = fx >= 0.f && fx <= (float)(Ws - 1);
  t.vx1 = fx + 1.f >= 0.f && fx + 1.f <= (float)(Ws - 1);
  t.vy0 = fy >= 0.f && fy <= (float)(Hs - 1);
  t.vy1 = fy + 1.f >= 0.f && fy + 1.f <= (float)(Hs - 1);
  t.x0 = (t.vx0 || t.vx1) ? (int)fx : 0;
  t.y0 = (t.vy0 || t.vy1) ? (int)fy : 0;
  return t;
}

// Branch-free channel loop: the four tap addresses are clamped into the source once (always
// dereferenceable) and taps outside the image are selected to zero after the load, so all loads of
// several unrolled channels are in flight together (the per-tap `if` form serialised them).
struct TapPlan {
  int o[4];     // element offsets inside one source plane (clamped)
  float w[4];   // bilinear weights
  bool ok[4];   // tap inside the source
};
__device__ __forceinline__ TapPlan plan_taps(const Taps& t, int Hs, int Ws) {
  TapPlan p;
  const int xa = min(max(t.x0, 0), Ws - 1), xb = min(max(t.x0 + 1, 0), Ws - 1);
  const int ya = min(max(t.y0, 0), Hs - 1), yb = min(max(t.y0 + 1, 0), Hs - 1);
  p.o[0] = ya * Ws + xa, p.o[1] = ya * Ws + xb, p.o[2] = yb * Ws + xa, p.o[3] = yb * Ws + xb;
  p.w[0] = t.wx0 * t.wy0, p.w[1] = t.wx1 * t.wy0, p.w[2] = t.wx0 * t.wy1, p.w[3] = t.wx1 * t.wy1;
  p.ok[0] = t.vx0 && t.vy0, p.ok[1] = t.vx1 && t.vy0, p.ok[2] = t.vx0 && t.vy1, p.ok[3] = t.vx1 && t.vy1;
  return p;
}

// Source rows / weights of output index d of the x2 bilinear upsample n_in -> n_out = 2 n_in:
// ATen/native/UpSample.h area_pixel_compute_source_index + the index / lambda arithmetic of upsample_bilinear2d.
__device__ __forceinline__ void up2_source(int d, int n_in, int n_out, bool align, int& i0, int& i1, float& l0,
                                           float& l1) {
  float src;
  if (align) {
    const float scale = n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f;
    src = scale * (float)d;
  } else {
    src = 0.5f * ((float)d + 0.5f) - 0.5f;  // scale_factor = 2 given: scale = 1 / 2
    src = src < 0.f ? 0.f : src;
  }
  i0 = min((int)src, n_in - 1);
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

__device__ __forceinline__ Taps no_taps() {
  Taps t;
  t.vx0 = t.vx1 = t.vy0 = t.vy1 = false;
  t.x0 = t.y0 = 0;
  t.wx0 = t.wx1 = t.wy0 = t.wy1 = t.dx = t.dy = 0.f;
  return t;
}

}  // namespace
