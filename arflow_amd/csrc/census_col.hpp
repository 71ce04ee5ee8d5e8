// Pair-shared census kernels ("column" layout) of the fused photometric direction (census_warp.hip).
//
// The soft census distance is symmetric in its pixel pair: with t(p, q) = d / sqrt(0.81 + d^2), d = grey(q) - grey(p),
// both transforms are odd under p <-> q, so dist(p, p + o) = dist(p + o, p) (utils/uflow_utils.py:241-279) and its
// derivative is odd.  The 4-pixels-per-lane kernels evaluate every ORDERED pair -- 48 evaluations per pixel, 3
// transcendentals each: they are VALU/transcendental-bound (0.92 VALU-busy, profiles/r02_pmc_valu.json).  Here every
// UNORDERED pair is evaluated once, by the pixel that sees its partner at an offset o in the half plane
//     HP = { dx > 0 } + { dx = 0, dy > 0 },
// and the value is added to BOTH pixels' sums.  The hand-over costs (almost) nothing because of the layout:
//   * a lane owns ONE COLUMN of K = 8 consecutive rows, a wave 64 adjacent columns, a workgroup 4 waves stacked
//     vertically (tile = 64 columns x 32 rows; grey tiles with an R halo in LDS, read column-wise: 64 lanes x 4 B
//     consecutive, conflict-free);
//   * the partner of (column l, row i) at offset (dx, dy) is (column l + dx, row i + dy): its ROW is a compile-time
//     register index of the receiving lane, its COLUMN is dx lanes to the right.  Contributions travelling dx lanes are
//     accumulated in T[K] and moved by ONE lane after each dx = R .. 1 (Horner: 3 x K wave shifts per tile for 21 K + 36
//     evaluations), dx = 0 stays in the lane;
//   * only left neighbours send, so only the R leftmost lanes of a wave are halo (their sums are incomplete and dropped):
//     64 - R useful columns per wave; rows R above / below the strip are evaluated as senders only.
// Evaluations per owned pixel: (24 K + 42) / K * 64 / 61 = 30.7 (K = 8, R = 3) instead of 48 (+ the centre tap).
// The ROUNDING differs from the ordered kernels only in the order the 48 terms of a pixel are added.
#pragma once
#include "census_tile.hpp"
#include "taps.hpp"

namespace {
namespace census_col {
constexpr int K = 8, NWV = 4, NT = 64 * NWV, TROWS = K * NWV, MAXR = 3;
constexpr int PITCH = 64 + MAXR + 1;          // tile columns: the 64 lanes' own + R to the right (+1: even pitch)
constexpr int LROWS = TROWS + 2 * MAXR;       // 38 tile rows
constexpr int TILE = LROWS * PITCH;

template <int R>
struct Geo {
  static constexpr int UX = 64 - R;           // useful columns per tile
  static constexpr int NR = TROWS + 2 * R, NC = 64 + R;
};
__host__ __device__ inline int tiles_x(int W, int R) { return (W + (64 - R) - 1) / (64 - R); }
__host__ __device__ inline int tiles_y(int H) { return (H + TROWS - 1) / TROWS; }

// plain plane tile: rows [y0-R, y0+TROWS+R), columns [x0, x0+64+R), zero outside the image
template <int R>
__device__ __forceinline__ void load_plane(float* __restrict__ tile, const float* __restrict__ g, int H, int W, int y0,
                                           int x0) {
  constexpr int NR = Geo<R>::NR, NC = Geo<R>::NC, ITER = (NR * NC + NT - 1) / NT;
  float v[ITER];
#pragma unroll
  for (int k = 0; k < ITER; ++k) {
    const int i = threadIdx.x + k * NT;
    const int r = i / NC, c = i - r * NC;
    const int gy = y0 - R + r, gx = x0 + c;
    const bool in = i < NR * NC && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const float t = g[in ? (long)gy * W + gx : 0];
    v[k] = in ? t : 0.f;
  }
#pragma unroll
  for (int k = 0; k < ITER; ++k) {
    const int i = threadIdx.x + k * NT;
    const int r = i / NC, c = i - r * NC;
    if (i < NR * NC) tile[r * PITCH + c] = v[k];
  }
}

__device__ __forceinline__ float sample1(const TapPlan& p, const float (&a)[4]) {
  float r = p.ok[0] ? a[0] * p.w[0] : 0.f;  // the per-channel expression of warp_fwd_kernel
  r = p.ok[1] ? fmaf(a[1], p.w[1], r) : r;
  r = p.ok[2] ? fmaf(a[2], p.w[2], r) : r;
  r = p.ok[3] ? fmaf(a[3], p.w[3], r) : r;
  return r;
}

// grey tile of the WARPED image b over the same rectangle (zero outside the image: the census transform zero-pads).
// Two batches: all flow loads of a batch are issued together, then all its 4 x taps.
template <int R>
__device__ __forceinline__ void load_gray_warped(float* __restrict__ tile, const float* __restrict__ gsrc,
                                                 const float* __restrict__ flow, int H, int W, int y0, int x0) {
  constexpr int NR = Geo<R>::NR, NC = Geo<R>::NC, ITER = (NR * NC + NT - 1) / NT, HB = (ITER + 1) / 2;
  const long cs = (long)H * W;
#pragma unroll
  for (int k0 = 0; k0 < ITER; k0 += HB) {
    float u[HB], v[HB];
    int gx[HB], gy[HB];
    bool in[HB];
#pragma unroll
    for (int k = 0; k < HB; ++k) {
      const int i = threadIdx.x + (k0 + k) * NT;
      const int r = i / NC, c = i - r * NC;
      gy[k] = y0 - R + r;
      gx[k] = x0 + c;
      in[k] = k0 + k < ITER && i < NR * NC && gy[k] >= 0 && gy[k] < H && gx[k] >= 0 && gx[k] < W;
      const long o = in[k] ? (long)gy[k] * W + gx[k] : 0;
      u[k] = flow[o];
      v[k] = flow[o + cs];
    }
    float a[HB][4];
    TapPlan p[HB];
#pragma unroll
    for (int k = 0; k < HB; ++k) {
      const Taps t = make_taps((float)gx[k], (float)gy[k], u[k], v[k], H, W, H, W, ARFLOW_PAD_ZEROS, true, ARFLOW_NORM_UFLOW);
      p[k] = plan_taps(t, H, W);
#pragma unroll
      for (int q = 0; q < 4; ++q) a[k][q] = gsrc[p[k].o[q]];
    }
#pragma unroll
    for (int k = 0; k < HB; ++k) {
      const int i = threadIdx.x + (k0 + k) * NT;
      const int r = i / NC, c = i - r * NC;
      if (k0 + k < ITER && i < NR * NC) tile[r * PITCH + c] = in[k] ? sample1(p[k], a[k]) : 0.f;
    }
  }
}

// EVERYTHING a tile needs from global memory in one pass over its positions (thread-strided): all loads that do not
// depend on another load are issued together, then the gathers, so a workgroup pays two memory round trips per batch
// before its pair loop and NONE after it (the epilogues used to reload the flow and gather the mask / the warp's corner
// taps per owned pixel: with every workgroup of the launch resident at once and in the same phase, that latency was not
// hidden by anybody's arithmetic -- it ADDED 22 us to the 28 us pair loop at 8 x 384 x 640).
//   ta: grey a;  tb: grey b warped by the position's flow (zero outside the image: the census transform zero-pads);
//   forward  (BWD = false): t2 = mask_invalid(flow) at the position (utils/uflow_utils.py:35-50); the range-map factor of
//                           the mask comes from stage_occ / up4_lds below;
//   backward (BWD = true):  t2 = dham (p2, plain), t3 / t4 = d sample / d flow_x, _y of the warp at the position (the
//                           bilinear corner differences of grey b -- the four taps are the sample's own).
template <int R, bool BWD>
__device__ __forceinline__ void fill_tiles(float* __restrict__ ta, float* __restrict__ tb, float* __restrict__ t2,
                                           float* __restrict__ t3, float* __restrict__ t4, const float* __restrict__ ga,
                                           const float* __restrict__ gsrc, const float* __restrict__ flow,
                                           const float* __restrict__ p2, int H, int W, int y0, int x0) {
  constexpr int NR = Geo<R>::NR, NC = Geo<R>::NC, ITER = (NR * NC + NT - 1) / NT, HB = (ITER + 1) / 2;
  const long cs = (long)H * W;
#pragma unroll
  for (int k0 = 0; k0 < ITER; k0 += HB) {
    float u[HB], v[HB], pa[HB], pg[HB];
    int gx[HB], gy[HB];
    bool in[HB];
#pragma unroll
    for (int k = 0; k < HB; ++k) {
      const int i = threadIdx.x + (k0 + k) * NT;
      const int r = i / NC, c = i - r * NC;
      gy[k] = y0 - R + r;
      gx[k] = x0 + c;
      in[k] = k0 + k < ITER && i < NR * NC && gy[k] >= 0 && gy[k] < H && gx[k] >= 0 && gx[k] < W;
      const long o = in[k] ? (long)gy[k] * W + gx[k] : 0;
      u[k] = flow[o];
      v[k] = flow[o + cs];
      pa[k] = ga[o];
      pg[k] = BWD ? p2[o] : 0.f;
    }
    float a[HB][4];
    Taps t[HB];
    TapPlan p[HB];
#pragma unroll
    for (int k = 0; k < HB; ++k) {
      t[k] = make_taps((float)gx[k], (float)gy[k], u[k], v[k], H, W, H, W, ARFLOW_PAD_ZEROS, true, ARFLOW_NORM_UFLOW);
      p[k] = plan_taps(t[k], H, W);
#pragma unroll
      for (int q = 0; q < 4; ++q) a[k][q] = gsrc[p[k].o[q]];
    }
#pragma unroll
    for (int k = 0; k < HB; ++k) {
      const int i = threadIdx.x + (k0 + k) * NT;
      const int r = i / NC, c = i - r * NC;
      if (k0 + k < ITER && i < NR * NC) {
        const int o = r * PITCH + c;
        ta[o] = in[k] ? pa[k] : 0.f;
        tb[o] = in[k] ? sample1(p[k], a[k]) : 0.f;
        if (BWD) {
          t2[o] = in[k] ? pg[k] : 0.f;
          const float nw = p[k].ok[0] ? a[k][0] : 0.f, ne = p[k].ok[1] ? a[k][1] : 0.f;
          const float sw = p[k].ok[2] ? a[k][2] : 0.f, se = p[k].ok[3] ? a[k][3] : 0.f;
          t3[o] = ((ne - nw) * t[k].wy0 + (se - sw) * t[k].wy1) * t[k].dx;
          t4[o] = ((sw - nw) * t[k].wx0 + (se - ne) * t[k].wx1) * t[k].dy;
        } else {
          // mask_invalid(flow_to_warp(flow)), utils/uflow_utils.py:35-50 (as warp_fwd_kernel's `valid`)
          const float cx = (float)gx[k] + u[k], cy = (float)gy[k] + v[k];
          const float mv = (cx >= 0.f && cx <= (float)(W - 1) && cy >= 0.f && cy <= (float)(H - 1)) ? 1.f : 0.f;
          t2[o] = in[k] ? mv : 0.f;
        }
      }
    }
  }
}

// upsample(clamp(range map, 0, 1), x4) for the tile's owned pixels out of LDS: the <= OCC_R x OCC_C cells of the [H/4, W/4]
// range map under the tile are staged once per workgroup, already clamped (one load per thread instead of 4 gathers per
// pixel), and up4_lds applies torch's bilinear arithmetic (align_corners=False, as up4_clamped) to them.
constexpr int OCC_R = TROWS / 4 + 2, OCC_C = 64 / 4 + 3, OCC_P = OCC_C + 1;
__device__ __forceinline__ int up4_first(int p) {  // first source cell of output index p (>= 0)
  return (int)fmaxf(0.25f * ((float)p + 0.5f) - 0.5f, 0.f);
}
__device__ __forceinline__ void stage_occ(float* __restrict__ tocc, const float* __restrict__ occ, int h4, int w4, int y0,
                                          int x0) {  // y0, x0: first OWNED row / column of the tile (>= 0)
  const int oy0 = up4_first(y0), ox0 = up4_first(x0);
  for (int i = threadIdx.x; i < OCC_R * OCC_P; i += NT) {
    const int r = i / OCC_P, c = i - r * OCC_P;
    const float v = occ[(long)min(oy0 + r, h4 - 1) * w4 + min(ox0 + c, w4 - 1)];
    tocc[i] = fminf(fmaxf(v, 0.f), 1.f);
  }
}
__device__ __forceinline__ float up4_lds(const float* __restrict__ tocc, int h4, int w4, int y0, int x0, int y, int x) {
  const float sy = fmaxf(0.25f * ((float)y + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.25f * ((float)x + 0.5f) - 0.5f, 0.f);
  const int ya = (int)sy, xa = (int)sx;
  const int yb = ya + (ya < h4 - 1 ? 1 : 0), xb = xa + (xa < w4 - 1 ? 1 : 0);
  const float ly = sy - (float)ya, lx = sx - (float)xa;
  const int oy0 = up4_first(y0), ox0 = up4_first(x0);
  const float v00 = tocc[(ya - oy0) * OCC_P + xa - ox0], v01 = tocc[(ya - oy0) * OCC_P + xb - ox0];
  const float v10 = tocc[(yb - oy0) * OCC_P + xa - ox0], v11 = tocc[(yb - oy0) * OCC_P + xb - ox0];
  return (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

// column `col` of a tile, rows [row0, row0 + K + 2R) -> registers
template <int R>
__device__ __forceinline__ void load_col(const float* __restrict__ tile, int row0, int col, float (&v)[K + 2 * R]) {
#pragma unroll
  for (int r = 0; r < K + 2 * R; ++r) v[r] = tile[(row0 + r) * PITCH + col];
}

template <int NV>
__device__ __forceinline__ void shift_right_1(float (&t)[NV]) {
#pragma unroll
  for (int k = 0; k < NV; ++k) t[k] = __shfl_up(t[k], 1, 64);  // lane l <- lane l-1 (lane 0 keeps its own: a halo lane)
}

// upsample(clamp(occ, 0, 1), x4)[y, x]: torch bilinear, align_corners=False (as up4_clamp_mul_kernel, smooth.hip)
__device__ __forceinline__ float up4_clamped(const float* __restrict__ occ, int h, int w, int y, int x) {
  const float sy = fmaxf(0.25f * ((float)y + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.25f * ((float)x + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  auto cl = [](float v) { return fminf(fmaxf(v, 0.f), 1.f); };
  const float v00 = cl(occ[(long)y0 * w + x0]), v01 = cl(occ[(long)y0 * w + x1]);
  const float v10 = cl(occ[(long)y1 * w + x0]), v11 = cl(occ[(long)y1 * w + x1]);
  return (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

template <int NV>
__device__ __forceinline__ void pin(float (&t)[NV]) {
#pragma unroll
  for (int k = 0; k < NV; ++k) asm volatile("" : "+v"(t[k]));
}

// soft census distance of one pixel pair: (t_a - t_b)^2 / (0.1 + (t_a - t_b)^2), t = d / sqrt(0.81 + d^2)
__device__ __forceinline__ float pair_dist(float ac, float bc, float an, float bn) {
  const float da = an - ac, db = bn - bc;
  const float tb = db * __builtin_amdgcn_rsqf(fmaf(db, db, 0.81f));
  const float e = fmaf(da, __builtin_amdgcn_rsqf(fmaf(da, da, 0.81f)), -tb), sq = e * e;
  return sq * __builtin_amdgcn_rcpf(0.1f + sq);  // (folding this product into the two accumulations as FMAs: forward 81 -> 84 us)
}
// its derivative w.r.t. the warped grey value of the FIRST pixel (before the constant 0.1 * -2 * 0.81), as
// census_warp::bwd_body evaluates it; odd under exchange of the two pixels
__device__ __forceinline__ float pair_grad(float ac, float bc, float an, float bn) {
  const float da = ac - an, db = bc - bn;
  const float ua = __builtin_amdgcn_rsqf(fmaf(da, da, 0.81f));
  const float ub = __builtin_amdgcn_rsqf(fmaf(db, db, 0.81f));
  const float e = fmaf(da, ua, -(db * ub));
  const float q = __builtin_amdgcn_rcpf(fmaf(e, e, 0.1f));
  const float w = q * ub;  // q^2 e ub^3 = (q ub)^2 (e ub): 4 multiplications instead of 5
  return (w * w) * (e * ub);
}

// Sum over the (2R+1)^2 - 1 neighbours of every pixel of the lane's K-row strip: s[i] = sum_o dist(pixel i, pixel i + o).
// ta / tb: the two grey tiles; w: wave (strip) index; lane: column.
template <int R>
__device__ __forceinline__ void pair_sums_fwd(const float* __restrict__ ta, const float* __restrict__ tb, int w, int lane,
                                              float (&s)[K]) {
  constexpr int NRR = K + 2 * R;
  float ac[NRR], bc[NRR], T[K];
  load_col<R>(ta, w * K, lane, ac);
  load_col<R>(tb, w * K, lane, bc);
#pragma unroll
  for (int i = 0; i < K; ++i) s[i] = 0.f, T[i] = 0.f;
#pragma unroll 1  // (rolled: unrolled, the kernel is 50 KB of straight-line code every wave walks once -- instruction fetch bound)
  for (int dx = R; dx >= 1; --dx) {
    float an[NRR], bn[NRR];
    load_col<R>(ta, w * K, lane + dx, an);
    load_col<R>(tb, w * K, lane + dx, bn);
#pragma unroll
    for (int dy = -R; dy <= R; ++dy) {
#pragma unroll
      for (int i = -R; i < K + R; ++i) {  // sender row i, partner row i + dy (strip-relative)
        const int j = i + dy;
        const bool own = i >= 0 && i < K, oth = j >= 0 && j < K;
        if (j < -R || j >= K + R || !(own || oth)) continue;
        const float v = pair_dist(ac[i + R], bc[i + R], an[j + R], bn[j + R]);
        if (own) s[i] += v;
        if (oth) T[j] += v;
      }
      pin(s);  // (the sums are only read under the epilogue's branch: left alone, LLVM sinks all their additions there
               //  and keeps every pair value alive until then -- 246 VGPRs)
      __builtin_amdgcn_sched_barrier(0);  // one (dx, dy) group at a time: bounds the live ranges
    }
    shift_right_1(T);
    asm volatile("" ::: "memory");  // the next dx's columns are loaded THEN (prefetched across the back edge they cost 42 VGPRs)
  }
#pragma unroll
  for (int i = 0; i < K; ++i) s[i] += T[i];
#pragma unroll
  for (int dy = 1; dy <= R; ++dy) {
#pragma unroll
    for (int i = -R; i < K; ++i) {
      const int j = i + dy;
      const bool own = i >= 0, oth = j >= 0 && j < K;
      if (!(own || oth)) continue;
      const float v = pair_dist(ac[i + R], bc[i + R], ac[j + R], bc[j + R]);
      if (own) s[i] += v;
      if (oth) s[j] += v;
    }
    pin(s);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// acc[i] = sum_o (g(i) + g(i + o)) * grad(pixel i, pixel i + o): the census4 backward's per-pixel value
template <int R>
__device__ __forceinline__ void pair_sums_bwd(const float* __restrict__ ta, const float* __restrict__ tb,
                                              const float* __restrict__ tg, int w, int lane, float (&acc)[K]) {
  constexpr int NRR = K + 2 * R;
  float ac[NRR], bc[NRR], gc[NRR], T[K];
  load_col<R>(ta, w * K, lane, ac);
  load_col<R>(tb, w * K, lane, bc);
  load_col<R>(tg, w * K, lane, gc);
#pragma unroll
  for (int i = 0; i < K; ++i) acc[i] = 0.f, T[i] = 0.f;
#pragma unroll 1  // (rolled: unrolled, the kernel is 50 KB of straight-line code every wave walks once -- instruction fetch bound)
  for (int dx = R; dx >= 1; --dx) {
    float an[NRR], bn[NRR], gn[NRR];
    load_col<R>(ta, w * K, lane + dx, an);
    load_col<R>(tb, w * K, lane + dx, bn);
    load_col<R>(tg, w * K, lane + dx, gn);
#pragma unroll
    for (int dy = -R; dy <= R; ++dy) {
#pragma unroll
      for (int i = -R; i < K + R; ++i) {
        const int j = i + dy;
        const bool own = i >= 0 && i < K, oth = j >= 0 && j < K;
        if (j < -R || j >= K + R || !(own || oth)) continue;
        const float gs = gc[i + R] + gn[j + R], hd = pair_grad(ac[i + R], bc[i + R], an[j + R], bn[j + R]);
        if (own) acc[i] = fmaf(gs, hd, acc[i]);
        if (oth) T[j] = fmaf(-gs, hd, T[j]);
        if (((i + R) & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // <= 4 evaluations interleaved (VGPRs: 203 -> 3 waves/SIMD)
      }
      pin(acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    shift_right_1(T);
    asm volatile("" ::: "memory");  // the next dx's columns are loaded THEN (prefetched across the back edge they cost 42 VGPRs)
  }
#pragma unroll
  for (int i = 0; i < K; ++i) acc[i] += T[i];
#pragma unroll
  for (int dy = 1; dy <= R; ++dy) {
#pragma unroll
    for (int i = -R; i < K; ++i) {
      const int j = i + dy;
      const bool own = i >= 0, oth = j >= 0 && j < K;
      if (!(own || oth)) continue;
      const float gs = gc[i + R] + gc[j + R], hd = pair_grad(ac[i + R], bc[i + R], ac[j + R], bc[j + R]);
      if (own) acc[i] = fmaf(gs, hd, acc[i]);
      if (oth) acc[j] = fmaf(-gs, hd, acc[j]);
      if (((i + R) & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    pin(acc);
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace census_col
}  // namespace
