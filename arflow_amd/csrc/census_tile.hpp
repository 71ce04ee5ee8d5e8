// The 16 x 64 pixel tile of the 4-pixels-per-lane census kernels (photo.hip, census_warp.hip): grey (x255) tiles
// with a MAXR halo in LDS at a conflict-free pitch, window rows read as three ds_read_b128.
#pragma once
#include "common.hpp"

namespace {
namespace census4 {
constexpr int TXW = 64, TYH = 16, NT = 256, PITCH = 128, MAXR = 3;
constexpr int ROWS = TYH + 2 * MAXR;  // 22 tile rows (R = 3); smaller radii use the top-left part
typedef float f32x4 __attribute__((ext_vector_type(4)));

// grey tile covering image rows [ty0-R, ty0+TYH+R) and columns [tx0-4, tx0+TXW+4), zero outside
template <int R>
__device__ __forceinline__ void load_gray(float* __restrict__ tile, const float* __restrict__ im, int H, int W,
                                          int ty0, int tx0) {
  constexpr int NR = TYH + 2 * R, NQ = (TXW + 8) / 4;  // 18 float4 per row
  const long cs = (long)H * W;
  for (int i = threadIdx.x; i < NR * NQ; i += NT) {
    const int r = i / NQ, q = i - r * NQ;
    const int gy = ty0 - R + r, gx = tx0 - 4 + 4 * q;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      const float* p = im + (long)gy * W + gx;
      const float4 a = *reinterpret_cast<const float4*>(p);
      const float4 b = *reinterpret_cast<const float4*>(p + cs);
      const float4 c = *reinterpret_cast<const float4*>(p + 2 * cs);
      g.x = ((a.x * 0.2989f + b.x * 0.5870f) + c.x * 0.1140f) * 255.f;
      g.y = ((a.y * 0.2989f + b.y * 0.5870f) + c.y * 0.1140f) * 255.f;
      g.z = ((a.z * 0.2989f + b.z * 0.5870f) + c.z * 0.1140f) * 255.f;
      g.w = ((a.w * 0.2989f + b.w * 0.5870f) + c.w * 0.1140f) * 255.f;
    }
    *reinterpret_cast<float4*>(tile + r * PITCH + 4 * q) = g;
  }
}

__device__ __forceinline__ void read12(const float* row, float (&w)[12]) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    f32x4 t = *reinterpret_cast<const f32x4*>(row + 4 * q);
    asm volatile("" : "+v"(t));  // keep it one ds_read_b128
    w[4 * q] = t.x, w[4 * q + 1] = t.y, w[4 * q + 2] = t.z, w[4 * q + 3] = t.w;
  }
}

}  // namespace census4
}  // namespace
