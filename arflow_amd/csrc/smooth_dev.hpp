// Device helpers of the edge-aware smoothness term (losses/loss_blocks.py:87-124, losses/uflow_loss.py:56-102), shared by
// smooth.hip and the fused range-map + smoothness launch of warp.hip.
#pragma once
#include "common.hpp"

namespace {

struct SmoothArgs {
  const float* flow;
  const float* img;
  int Ci, H, W;
  long fbs;
  float fscale, alpha;
  int order, wmode, penalty;
};

__device__ __forceinline__ float pen(float v, int penalty) {
  return penalty == 0 ? fabsf(v) : sqrtf(fmaf(v, v, 1e-6f));
}
__device__ __forceinline__ float dpen(float v, int penalty) {
  if (penalty == 0) return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
  return v / sqrtf(fmaf(v, v, 1e-6f));
}

// Sum over channels of |I(p) - I(q)| with the element offsets p, q inside one plane.  CI > 0: the channel
// count is a compile-time constant (3 for every caller) and all 2*CI loads are issued before the first
// use; CI == 0 keeps the runtime loop.  (With the runtime loop hipcc waits for each pair of loads: a
// dozen serialised round trips per pixel made the full-resolution launches latency-bound.)
template <int CI>
__device__ __forceinline__ float chan_absdiff(const SmoothArgs& a, const float* ib, long p, long q) {
  const long cs = (long)a.H * a.W;
  float s = 0.f;
  if (CI > 0) {
    float u[CI > 0 ? CI : 1], v[CI > 0 ? CI : 1];
#pragma unroll
    for (int c = 0; c < CI; ++c) u[c] = ib[c * cs + p], v[c] = ib[c * cs + q];
#pragma unroll
    for (int c = 0; c < CI; ++c) s += fabsf(u[c] - v[c]);
  } else {
    for (int c = 0; c < a.Ci; ++c) s += fabsf(ib[c * cs + p] - ib[c * cs + q]);
  }
  return s;
}
// edge weight for the x-difference anchored at (y,x); caller guarantees validity.
template <int CI>
__device__ __forceinline__ float edge_wx(const SmoothArgs& a, const float* ib, int y, int x) {
  const int xa = a.order == 1 ? x + 1 : x + 2;
  const int xb = a.order == 1 ? x : (a.wmode == 0 ? x + 1 : x);
  const float s = chan_absdiff<CI>(a, ib, (long)y * a.W + xa, (long)y * a.W + xb);
  return __expf(-(s / (float)a.Ci) * a.alpha);
}
template <int CI>
__device__ __forceinline__ float edge_wy(const SmoothArgs& a, const float* ib, int y, int x) {
  const int ya = a.order == 1 ? y + 1 : y + 2;
  const int yb = a.order == 1 ? y : (a.wmode == 0 ? y + 1 : y);
  const float s = chan_absdiff<CI>(a, ib, (long)ya * a.W + x, (long)yb * a.W + x);
  return __expf(-(s / (float)a.Ci) * a.alpha);
}
// flow differences anchored at (y,x)
__device__ __forceinline__ float diff_x(const SmoothArgs& a, const float* f, int y, int x) {
  const float* r = f + (long)y * a.W + x;
  return a.order == 1 ? (r[1] - r[0]) * a.fscale : ((r[2] - r[1]) - (r[1] - r[0])) * a.fscale;
}
__device__ __forceinline__ float diff_y(const SmoothArgs& a, const float* f, int y, int x) {
  const float* r = f + (long)y * a.W + x;
  const int W = a.W;
  return a.order == 1 ? (r[W] - r[0]) * a.fscale : ((r[2 * W] - r[W]) - (r[W] - r[0])) * a.fscale;
}


// the smoothness partial sums of ONE pixel (y, x): (x-term, y-term), as smooth_fwd_kernel accumulates them
template <int CI>
__device__ __forceinline__ void smooth_pixel(const SmoothArgs& a, const float* ib, const float* fb, int y, int x, float& px,
                                             float& py) {
  const long cs = (long)a.H * a.W;
  const int o = a.order;
  if (x < a.W - o) {
    const float w = edge_wx<CI>(a, ib, y, x);
    px += w * (pen(diff_x(a, fb, y, x), a.penalty) + pen(diff_x(a, fb + cs, y, x), a.penalty));
  }
  if (y < a.H - o) {
    const float w = edge_wy<CI>(a, ib, y, x);
    py += w * (pen(diff_y(a, fb, y, x), a.penalty) + pen(diff_y(a, fb + cs, y, x), a.penalty));
  }
}

// gradient of the smoothness sums w.r.t. the two flow planes at pixel (y, x) of sample b (smooth_bwd_kernel's work item)
template <int CI>
__device__ __forceinline__ void smooth_bwd_pixel(const SmoothArgs& a, const float* __restrict__ coef,
                                                 float* __restrict__ gflow, int b, int y, int x) {
  const int o = a.order;
  const float* ib = a.img + (long)b * a.Ci * a.H * a.W;
  const float* fb = a.flow + (long)b * a.fbs;
  const long cs = (long)a.H * a.W;
  const float cx = coef[0] * a.fscale, cy = coef[1] * a.fscale;
  float g[2] = {0.f, 0.f};
  // pixel (y,x) enters the difference anchored at x-k with stencil coefficient st[k]
  // order 1: {-1, +1}; order 2: {+1, -2, +1}
  const float st1[2] = {-1.f, 1.f};
  const float st2[3] = {1.f, -2.f, 1.f};
  for (int k = 0; k <= o; ++k) {
    const float s = o == 1 ? st1[k] : st2[k];
    const int xa = x - k;
    if (xa >= 0 && xa < a.W - o) {
      const float w = edge_wx<CI>(a, ib, y, xa) * s * cx;
      g[0] = fmaf(w, dpen(diff_x(a, fb, y, xa), a.penalty), g[0]);
      g[1] = fmaf(w, dpen(diff_x(a, fb + cs, y, xa), a.penalty), g[1]);
    }
    const int ya = y - k;
    if (ya >= 0 && ya < a.H - o) {
      const float w = edge_wy<CI>(a, ib, ya, x) * s * cy;
      g[0] = fmaf(w, dpen(diff_y(a, fb, ya, x), a.penalty), g[0]);
      g[1] = fmaf(w, dpen(diff_y(a, fb + cs, ya, x), a.penalty), g[1]);
    }
  }
  float* go = gflow + (long)b * 2 * cs + (long)y * a.W + x;
  go[0] = g[0];
  go[cs] = g[1];
}

}  // namespace
