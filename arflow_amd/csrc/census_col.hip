// The pair-shared ("column") census kernels of the fused photometric direction: see census_col.hpp for the scheme and
// census_warp.hip for the operation (losses/uflow_loss.py:30-54, utils/uflow_utils.py:241-293).
#include <cstdlib>

#include "census_col.hpp"
#include "smooth_dev.hpp"

namespace {
namespace census_warp {

template <int R>
__global__ __launch_bounds__(census_col::NT) void fwd_col_kernel(const float* __restrict__ gray_a, const float* __restrict__ gray_b,
                                                                 const float* __restrict__ flow, long fbs,
                                                                 const float* __restrict__ occ_small,
                                                                 float* __restrict__ mask_out, float* __restrict__ dham_out,
                                                                 float* __restrict__ sums, int nrows, int nimg, int H, int W,
                                                                 int pair) {
  namespace cc = census_col;
  __shared__ float ta[cc::TILE];
  __shared__ float tb[cc::TILE];
  __shared__ float tm[cc::TILE];
  __shared__ float tocc[cc::OCC_R * cc::OCC_P];
  __shared__ float red[2 * (cc::NT / 64)];
  int btx, bty, b;
  if (!af_tile_of_block(cc::tiles_x(W, R), cc::tiles_y(H), nimg, btx, bty, b)) {
    if (threadIdx.x == 0) af_store_partial(sums, nrows, 0.f, 0.f, 0.f);  // padding workgroup: its row must be defined
    return;
  }
  const int y0 = bty * cc::TROWS, x0 = btx * cc::Geo<R>::UX - R;
  const long cs = (long)H * W;
  const int bp = pair ? (b ^ 1) : b;
  const float* occ = occ_small ? occ_small + (long)bp * (H / 4) * (W / 4) : nullptr;
  const int ox = max(x0 + R, 0);  // first owned column
  if (occ) cc::stage_occ(tocc, occ, H / 4, W / 4, y0, ox);
  cc::fill_tiles<R, false>(ta, tb, tm, nullptr, nullptr, gray_a + b * cs, gray_b + bp * cs, flow + b * fbs, nullptr, H, W, y0,
                           x0);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float s[cc::K];
  cc::pair_sums_fwd<R>(ta, tb, w, lane, s);
  const int x = x0 + lane;
  float part[2] = {0.f, 0.f};
  if (lane >= R && x < W) {
    const bool colin = x >= R && x < W - R;
#pragma unroll
    for (int i = 0; i < cc::K; ++i) {
      const int y = y0 + w * cc::K + i;
      if (y < H) {
        float mv = tm[(R + w * cc::K + i) * cc::PITCH + lane];  // mask_invalid; x the upsampled range map
        if (occ) mv *= cc::up4_lds(tocc, H / 4, W / 4, y0, ox, y, x);
        const float pm = (colin && y >= R && y < H - R) ? mv : 0.f;
        const float lg = __log2f(fabsf(s[i]) + 0.01f);
        part[0] += exp2f(0.4f * lg) * pm;
        part[1] += pm;
        const long o = (long)b * cs + (long)y * W + x;
        if (mask_out) mask_out[o] = mv;
        dham_out[o] = pm * 0.4f * exp2f(-0.6f * lg);
      }
    }
  }
  af_block_sum<2>(part, red);
  if (threadIdx.x == 0) {
    if (pair && (b & 1))
      af_store_partial(sums, nrows, 0.f, 0.f, part[0], part[1]);
    else
      af_store_partial(sums, nrows, part[0], part[1], 0.f);
  }
}

template <int R>
__device__ __forceinline__ void bwd_col_body(const float* __restrict__ gray_a, const float* __restrict__ gray_b,
                                             const float* __restrict__ flow, long fbs, const float* __restrict__ dham,
                                             const float* __restrict__ scale, float* __restrict__ gflow, int nimg, int H,
                                             int W, int pair) {
  namespace cc = census_col;
  __shared__ float ta[cc::TILE];
  __shared__ float tb[cc::TILE];
  __shared__ float tg[cc::TILE];
  __shared__ float tcx[cc::TILE];
  __shared__ float tcy[cc::TILE];
  int btx, bty, b;
  if (!af_tile_of_block(cc::tiles_x(W, R), cc::tiles_y(H), nimg, btx, bty, b)) return;
  const int y0 = bty * cc::TROWS, x0 = btx * cc::Geo<R>::UX - R;
  const long cs = (long)H * W;
  cc::fill_tiles<R, true>(ta, tb, tg, tcx, tcy, gray_a + b * cs, gray_b + (pair ? (b ^ 1) : b) * cs, flow + b * fbs,
                          dham + b * cs, H, W, y0, x0);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float acc[cc::K];
  cc::pair_sums_bwd<R>(ta, tb, tg, w, lane, acc);
  const int x = x0 + lane;
  if (lane < R || x >= W) return;
  // d loss / d grey_b(p) = sc * acc (census4::bwd_kernel's value before the colour weights; the x255 lives in the
  // grey plane), times the warp's flow gradient (tcx, tcy)
  const float sc = (scale ? scale[pair ? (b & 1) : 0] : 1.f) * (0.1f * -2.f * 0.81f);  // pair: one scale per direction
#pragma unroll
  for (int i = 0; i < cc::K; ++i) {
    const int y = y0 + w * cc::K + i;
    if (y < H) {
      const int o = (R + w * cc::K + i) * cc::PITCH + lane;
      float* gf = gflow + (long)b * 2 * cs + (long)y * W + x;
      gf[0] = sc * acc[i] * tcx[o];
      gf[cs] = sc * acc[i] * tcy[o];
    }
  }
}

template <int R>
__global__ __launch_bounds__(census_col::NT) void bwd_col_kernel(const float* __restrict__ gray_a, const float* __restrict__ gray_b,
                                                                 const float* __restrict__ flow, long fbs,
                                                                 const float* __restrict__ dham, const float* __restrict__ scale,
                                                                 float* __restrict__ gflow, int nimg, int H, int W, int pair) {
  bwd_col_body<R>(gray_a, gray_b, flow, fbs, dham, scale, gflow, nimg, H, W, pair);
}


// The WHOLE backward of UFlowLoss as one launch (census_warp.hip pair_bwd_smooth_kernel, column form): workgroups
// [0, census_blocks) run the census + warp backward of both directions, the rest the smoothness backward of the level-2
// flows (one row of 256 columns per workgroup).
template <int R>
__global__ __launch_bounds__(census_col::NT) void pair_bwd_smooth_col_kernel(const float* __restrict__ gray,
                                                                             const float* __restrict__ flow, long fbs,
                                                                             const float* __restrict__ dham,
                                                                             const float* __restrict__ scale2,
                                                                             float* __restrict__ gflow, int nimg, int H, int W,
                                                                             unsigned census_blocks, SmoothArgs sa,
                                                                             const float* __restrict__ coef,
                                                                             float* __restrict__ gflow2) {
  if (blockIdx.x < census_blocks) {
    bwd_col_body<R>(gray, gray, flow, fbs, dham, scale2, gflow, nimg, H, W, 1);
    return;
  }
  const unsigned i = blockIdx.x - census_blocks;  // (b, y, x-block) of the level-2 grid
  const unsigned nxb = (unsigned)((sa.W + 255) / 256);
  const int xb = (int)(i % nxb), y = (int)((i / nxb) % (unsigned)sa.H), b = (int)(i / (nxb * (unsigned)sa.H));
  const int x = xb * 256 + (int)threadIdx.x;
  if (b < nimg && x < sa.W) smooth_bwd_pixel<3>(sa, coef, gflow2, b, y, x);
}

}  // namespace census_warp
}  // namespace

static unsigned col_grid(int B, int H, int W, int R) {
  return af_grid_for_tiles((long)census_col::tiles_x(W, R) * census_col::tiles_y(H) * B);
}

int census_col_fwd(const float* gray_a, const float* gray_b, const float* flow, long fbs, const float* occ_small,
                   float* mask_out, float* dham, float* sums, int nrows, int B, int H, int W, int radius, int pair,
                   hipStream_t st) {
  namespace cw = census_warp;
  dim3 g(col_grid(B, H, W, radius)), t(census_col::NT);
  switch (radius) {
    case 1: hipLaunchKernelGGL(cw::fwd_col_kernel<1>, g, t, 0, st, gray_a, gray_b, flow, fbs, occ_small, mask_out, dham, sums, nrows, B, H, W, pair); break;
    case 2: hipLaunchKernelGGL(cw::fwd_col_kernel<2>, g, t, 0, st, gray_a, gray_b, flow, fbs, occ_small, mask_out, dham, sums, nrows, B, H, W, pair); break;
    default: hipLaunchKernelGGL(cw::fwd_col_kernel<3>, g, t, 0, st, gray_a, gray_b, flow, fbs, occ_small, mask_out, dham, sums, nrows, B, H, W, pair); break;
  }
  return af_launch_status();
}

int census_col_bwd(const float* gray_a, const float* gray_b, const float* flow, long fbs, const float* dham,
                   const float* scale, float* gflow, int B, int H, int W, int radius, int pair, hipStream_t st) {
  namespace cw = census_warp;
  dim3 g(col_grid(B, H, W, radius)), t(census_col::NT);
  switch (radius) {
    case 1: hipLaunchKernelGGL(cw::bwd_col_kernel<1>, g, t, 0, st, gray_a, gray_b, flow, fbs, dham, scale, gflow, B, H, W, pair); break;
    case 2: hipLaunchKernelGGL(cw::bwd_col_kernel<2>, g, t, 0, st, gray_a, gray_b, flow, fbs, dham, scale, gflow, B, H, W, pair); break;
    default: hipLaunchKernelGGL(cw::bwd_col_kernel<3>, g, t, 0, st, gray_a, gray_b, flow, fbs, dham, scale, gflow, B, H, W, pair); break;
  }
  return af_launch_status();
}

int census_col_pair_bwd_smooth(const float* gray, const float* flow, long fbs, const float* dham, const float* scale2,
                               float* gflow, int B2, int H, int W, int radius, const float* flow2, long flow2_bstride,
                               const float* img2, int h2, int w2, float flow_scale, float alpha, int order, int wmode,
                               int penalty, const float* coef2, float* gflow2, hipStream_t st) {
  namespace cw = census_warp;
  const SmoothArgs sa{flow2, img2, 3, h2, w2, flow2_bstride, flow_scale, alpha, order, wmode, penalty};
  const unsigned cb = col_grid(B2, H, W, radius);
  const unsigned sb = (unsigned)af_cdiv(sa.W, 256) * (unsigned)sa.H * (unsigned)B2;
  dim3 g(cb + sb), t(census_col::NT);
  switch (radius) {
    case 1: hipLaunchKernelGGL(cw::pair_bwd_smooth_col_kernel<1>, g, t, 0, st, gray, flow, fbs, dham, scale2, gflow, B2, H, W, cb, sa, coef2, gflow2); break;
    case 2: hipLaunchKernelGGL(cw::pair_bwd_smooth_col_kernel<2>, g, t, 0, st, gray, flow, fbs, dham, scale2, gflow, B2, H, W, cb, sa, coef2, gflow2); break;
    default: hipLaunchKernelGGL(cw::pair_bwd_smooth_col_kernel<3>, g, t, 0, st, gray, flow, fbs, dham, scale2, gflow, B2, H, W, cb, sa, coef2, gflow2); break;
  }
  return af_launch_status();
}
