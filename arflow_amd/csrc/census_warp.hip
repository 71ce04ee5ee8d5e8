// Fused photometric direction of UFlowLoss (losses/uflow_loss.py:30-54) for gfx950:
//
//     recons = resample(im_b, flow_to_warp(flow))                     utils/uflow_utils.py:6-32,53-77
//     valid  = mask_invalid(flow_to_warp(flow))                       utils/uflow_utils.py:35-50
//     mask   = upsample(clamp(range_map, 0, 1), x4) * valid           losses/uflow_loss.py:41-48
//     loss   = census_loss(im_a, recons, mask)                        utils/uflow_utils.py:282-293
//
// as ONE forward launch and ONE backward launch.  Unfused this was warp_fwd (24-30 us at 8x3x384x640) ->
// up4_clamp_mul -> census_fwd forward and census_bwd -> warp_bwd_flow (25-27 us) backward, with the warped image,
// the validity mask and the image gradient making a round trip through HBM in between.
//
// The census transform only sees the GREY image (rgb_to_grayscale, utils/uflow_utils.py:227-231) and both the
// bilinear sample and the grey conversion are linear, so grey(resample(im_b)) = resample(grey(im_b)): the kernels
// take the grey planes (x255) of the two images -- written once per step by arflow_down4_gray, which reads the
// images anyway for the x1/4 copies of the smoothness term -- and sample ONE plane (4 taps) while the tile is
// filled instead of three.  The warped image, its gradient and the validity mask never exist in memory; the
// backward turns the per-pixel grey gradient straight into d loss / d flow with the bilinear corner differences
// of the grey plane.  (The re-association changes results at fp32 rounding level only; the unfused entry points
// keep the reference's operation order and tests/test_hip_parity.py compares the two and the oracle.)
//
// The census arithmetic (census4 of photo.hip: 16 x 64 pixel tile, 4 pixels per lane, window rows as
// ds_read_b128) is transcendental/VALU-bound; the ~6 extra dword gathers per pixel of the fill stage (tile + 3 px
// halo = 1.5 x the tile, 4 taps each) are served by L2 and overlap other workgroups' arithmetic.
#include <cstdlib>

#include "census_col.hpp"
#include "census_tile.hpp"
#include "smooth_dev.hpp"
#include "taps.hpp"

namespace {
namespace census_warp {
using namespace census4;
using census_col::up4_clamped;

// plain grey tile from a [H,W] plane: rows [ty0-R, ty0+TYH+R), columns [tx0-4, tx0+TXW+4), zero outside
template <int R>
__device__ __forceinline__ void load_plane(float* __restrict__ tile, const float* __restrict__ g, int H, int W, int ty0,
                                           int tx0) {
  constexpr int NR = TYH + 2 * R, NQ = (TXW + 8) / 4;
  for (int i = threadIdx.x; i < NR * NQ; i += NT) {
    const int r = i / NQ, q = i - r * NQ;
    const int gy = ty0 - R + r, gx = tx0 - 4 + 4 * q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *reinterpret_cast<const float4*>(g + (long)gy * W + gx);
    *reinterpret_cast<float4*>(tile + r * PITCH + 4 * q) = v;
  }
}

__device__ __forceinline__ float sample1(const TapPlan& p, const float (&a)[4]) {
  float r = p.ok[0] ? a[0] * p.w[0] : 0.f;  // the per-channel expression of warp_fwd_kernel
  r = p.ok[1] ? fmaf(a[1], p.w[1], r) : r;
  r = p.ok[2] ? fmaf(a[2], p.w[2], r) : r;
  r = p.ok[3] ? fmaf(a[3], p.w[3], r) : r;
  return r;
}

// Grey tile of the WARPED image b: rows [ty0-R, ty0+TYH+R), columns [tx0-R, tx0+TXW+R); zero outside the image
// (the census transform zero-pads).  All flow loads of a thread are issued together, then all its 4 x ITER taps:
// two round trips for the whole fill instead of a dependent chain per pixel.
template <int R>
__device__ __forceinline__ void load_gray_warped(float* __restrict__ tile, const float* __restrict__ gsrc,
                                                 const float* __restrict__ flow, int H, int W, int ty0, int tx0) {
  constexpr int NR = TYH + 2 * R, NC = TXW + 2 * R, ITER = (NR * NC + NT - 1) / NT;
  const long cs = (long)H * W;
  float u[ITER], v[ITER];
  int gx[ITER], gy[ITER];
  bool in[ITER];
#pragma unroll
  for (int k = 0; k < ITER; ++k) {
    const int i = threadIdx.x + k * NT;
    const int r = i / NC, c = i - r * NC;
    gy[k] = ty0 - R + r;
    gx[k] = tx0 - R + c;
    in[k] = i < NR * NC && gy[k] >= 0 && gy[k] < H && gx[k] >= 0 && gx[k] < W;
    const long o = in[k] ? (long)gy[k] * W + gx[k] : 0;
    u[k] = flow[o];
    v[k] = flow[o + cs];
  }
  float a[ITER][4];
  TapPlan p[ITER];
#pragma unroll
  for (int k = 0; k < ITER; ++k) {
    const Taps t = make_taps((float)gx[k], (float)gy[k], u[k], v[k], H, W, H, W, ARFLOW_PAD_ZEROS, true, ARFLOW_NORM_UFLOW);
    p[k] = plan_taps(t, H, W);
#pragma unroll
    for (int q = 0; q < 4; ++q) a[k][q] = gsrc[p[k].o[q]];
  }
#pragma unroll
  for (int k = 0; k < ITER; ++k) {
    const int i = threadIdx.x + k * NT;
    const int r = i / NC, c = i - r * NC;
    if (i < NR * NC) tile[r * PITCH + (c + 4 - R)] = in[k] ? sample1(p[k], a[k]) : 0.f;
  }
}

// zero the tile columns no gather fills (tile col 0 .. 3-R and 4+TXW+R .. TXW+7): read12 loads them
template <int R>
__device__ __forceinline__ void zero_margins(float* __restrict__ tile) {
  constexpr int NR = TYH + 2 * R, M = 4 - R;
  for (int i = threadIdx.x; i < NR * 2 * M; i += NT) {
    const int r = i / (2 * M), c = i - r * (2 * M);
    tile[r * PITCH + (c < M ? c : TXW + 4 + R + (c - M))] = 0.f;
  }
}

template <int R>
__global__ __launch_bounds__(NT) void fwd_kernel(const float* __restrict__ gray_a, const float* __restrict__ gray_b,
                                                 const float* __restrict__ flow, long fbs,
                                                 const float* __restrict__ occ_small, float* __restrict__ mask_out,
                                                 float* __restrict__ dham_out, float* __restrict__ sums, int nrows,
                                                 int nimg, int H, int W, int pair) {
  // pair: the batch interleaves the two DIRECTIONS of UFlowLoss (sample s = 2 b + dir; losses/uflow_loss.py:30-54 runs them
  // one after the other): image b of sample s is plane s ^ 1 of `gray_b` (= gray_a), its range map plane s ^ 1 of
  // `occ_small`, and the partial sums of direction 1 go to columns 2, 3 of the row instead of 0, 1.
  __shared__ __attribute__((aligned(16))) float ga[ROWS * PITCH];
  __shared__ __attribute__((aligned(16))) float gb[ROWS * PITCH];
  __shared__ float red[2 * (NT / 64)];
  int btx, bty, b;
  if (!af_tile_of_block((W + TXW - 1) / TXW, (H + TYH - 1) / TYH, nimg, btx, bty, b)) {
    if (threadIdx.x == 0) af_store_partial(sums, nrows, 0.f, 0.f, 0.f);  // padding workgroup: its row must be defined
    return;
  }
  const int ty0 = bty * TYH, tx0 = btx * TXW;
  const long cs = (long)H * W;
  const float* fl = flow + b * fbs;
  const int bp = pair ? (b ^ 1) : b;
  zero_margins<R>(gb);
  load_gray_warped<R>(gb, gray_b + bp * cs, fl, H, W, ty0, tx0);
  load_plane<R>(ga, gray_a + b * cs, H, W, ty0, tx0);
  __syncthreads();
  const int xg = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int x0 = tx0 + 4 * xg, y = ty0 + ly;
  float ca[4], cb[4], s[4] = {0.f, 0.f, 0.f, 0.f};
  {
    float wa[12], wb[12];
    read12(ga + (ly + R) * PITCH + 4 * xg, wa);
    read12(gb + (ly + R) * PITCH + 4 * xg, wb);
#pragma unroll
    for (int p = 0; p < 4; ++p) ca[p] = wa[4 + p], cb[p] = wb[4 + p];
  }
#pragma unroll 1
  for (int dy = 0; dy <= 2 * R; ++dy) {
    float wa[12], wb[12];
    read12(ga + (ly + dy) * PITCH + 4 * xg, wa);
    read12(gb + (ly + dy) * PITCH + 4 * xg, wb);
#pragma unroll
    for (int dx = 0; dx <= 2 * R; ++dx)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int k = p + dx + 4 - R;
        const float da = wa[k] - ca[p], db = wb[k] - cb[p];
        const float tb = db * __builtin_amdgcn_rsqf(fmaf(db, db, 0.81f));
        const float e = fmaf(da, __builtin_amdgcn_rsqf(fmaf(da, da, 0.81f)), -tb), sq = e * e;
        s[p] = fmaf(sq, __builtin_amdgcn_rcpf(0.1f + sq), s[p]);
      }
  }
  float part[2] = {0.f, 0.f};
  if (y < H && x0 < W) {  // W % 4 == 0: the 4 pixels are inside together
    const long o = (long)y * W + x0;
    const float4 fu = *reinterpret_cast<const float4*>(fl + o);
    const float4 fv = *reinterpret_cast<const float4*>(fl + cs + o);
    const float uu[4] = {fu.x, fu.y, fu.z, fu.w}, vv[4] = {fv.x, fv.y, fv.z, fv.w};
    const float* occ = occ_small ? occ_small + (long)bp * (H / 4) * (W / 4) : nullptr;
    float mv[4], dh[4];
    const bool rowin = y >= R && y < H - R;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int xx = x0 + p;
      // mask_invalid(flow_to_warp(flow)), utils/uflow_utils.py:35-50 (as warp_fwd_kernel's `valid`)
      const float cx = (float)xx + uu[p], cy = (float)y + vv[p];
      const float val = (cx >= 0.f && cx <= (float)(W - 1) && cy >= 0.f && cy <= (float)(H - 1)) ? 1.f : 0.f;
      mv[p] = occ ? up4_clamped(occ, H / 4, W / 4, y, xx) * val : val;
      const float pm = (rowin && xx >= R && xx < W - R) ? mv[p] : 0.f;
      const float lg = __log2f(fabsf(s[p]) + 0.01f);
      part[0] += exp2f(0.4f * lg) * pm;
      part[1] += pm;
      dh[p] = pm * 0.4f * exp2f(-0.6f * lg);
    }
    if (mask_out) *reinterpret_cast<float4*>(mask_out + (long)b * cs + o) = make_float4(mv[0], mv[1], mv[2], mv[3]);
    *reinterpret_cast<float4*>(dham_out + (long)b * cs + o) = make_float4(dh[0], dh[1], dh[2], dh[3]);
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
__device__ __forceinline__ void bwd_body(const float* __restrict__ gray_a, const float* __restrict__ gray_b,
                                         const float* __restrict__ flow, long fbs,
                                         const float* __restrict__ dham, const float* __restrict__ scale,
                                         float* __restrict__ gflow, int nimg, int H, int W, int pair) {
  __shared__ __attribute__((aligned(16))) float ga[ROWS * PITCH];
  __shared__ __attribute__((aligned(16))) float gb[ROWS * PITCH];
  __shared__ __attribute__((aligned(16))) float gg[ROWS * PITCH];
  int btx, bty, b;
  if (!af_tile_of_block((W + TXW - 1) / TXW, (H + TYH - 1) / TYH, nimg, btx, bty, b)) return;
  const int ty0 = bty * TYH, tx0 = btx * TXW;
  const long cs = (long)H * W;
  const float* fl = flow + b * fbs;
  const float* sb = gray_b + (pair ? (b ^ 1) : b) * cs;
  zero_margins<R>(gb);
  load_gray_warped<R>(gb, sb, fl, H, W, ty0, tx0);
  load_plane<R>(ga, gray_a + b * cs, H, W, ty0, tx0);
  load_plane<R>(gg, dham + b * cs, H, W, ty0, tx0);
  __syncthreads();
  const int xg = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int x0 = tx0 + 4 * xg, y = ty0 + ly;
  if (y >= H || x0 >= W) return;
  // the taps of this lane's 4 pixels (corner differences of the grey plane): issued before the census loop so
  // the 16 gathers are in flight while it runs
  const long o = (long)y * W + x0;
  const float4 fu = *reinterpret_cast<const float4*>(fl + o);
  const float4 fv = *reinterpret_cast<const float4*>(fl + cs + o);
  const float uu[4] = {fu.x, fu.y, fu.z, fu.w}, vv[4] = {fv.x, fv.y, fv.z, fv.w};
  float cdx[4], cdy[4];  // (d sample / d coordinate) * (d coordinate / d flow) per pixel
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const Taps t = make_taps((float)(x0 + p), (float)y, uu[p], vv[p], H, W, H, W, ARFLOW_PAD_ZEROS, true, ARFLOW_NORM_UFLOW);
    const TapPlan pl = plan_taps(t, H, W);
    float a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = sb[pl.o[q]];
    const float nw = pl.ok[0] ? a[0] : 0.f, ne = pl.ok[1] ? a[1] : 0.f;
    const float sw = pl.ok[2] ? a[2] : 0.f, se = pl.ok[3] ? a[3] : 0.f;
    cdx[p] = ((ne - nw) * t.wy0 + (se - sw) * t.wy1) * t.dx;
    cdy[p] = ((sw - nw) * t.wx0 + (se - ne) * t.wx1) * t.dy;
  }
  float ca[4], cb[4], cg[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
  {
    float wa[12], wb[12], wg[12];
    read12(ga + (ly + R) * PITCH + 4 * xg, wa);
    read12(gb + (ly + R) * PITCH + 4 * xg, wb);
    read12(gg + (ly + R) * PITCH + 4 * xg, wg);
#pragma unroll
    for (int p = 0; p < 4; ++p) ca[p] = wa[4 + p], cb[p] = wb[4 + p], cg[p] = wg[4 + p];
  }
#pragma unroll 1
  for (int dy = 0; dy <= 2 * R; ++dy) {
    float wa[12], wb[12], wg[12];
    read12(ga + (ly + dy) * PITCH + 4 * xg, wa);
    read12(gb + (ly + dy) * PITCH + 4 * xg, wb);
    read12(gg + (ly + dy) * PITCH + 4 * xg, wg);
#pragma unroll
    for (int dx = 0; dx <= 2 * R; ++dx) {
      if (dy == R && dx == R) continue;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int k = p + dx + 4 - R;
        const float da = ca[p] - wa[k], db = cb[p] - wb[k];
        const float ua = __builtin_amdgcn_rsqf(fmaf(da, da, 0.81f));
        const float ub = __builtin_amdgcn_rsqf(fmaf(db, db, 0.81f));
        const float e = fmaf(da, ua, -(db * ub));
        const float q = __builtin_amdgcn_rcpf(fmaf(e, e, 0.1f));
        const float hd = ((q * e) * q) * ((ub * ub) * ub);
        acc[p] = fmaf(wg[k] + cg[p], hd, acc[p]);
      }
    }
  }
  // d loss / d grey_b(p) = sc * acc (census4::bwd_kernel's value before the colour weights; the x255 lives in the
  // grey plane), times the warp's flow gradient
  const float sc = (scale ? scale[pair ? (b & 1) : 0] : 1.f) * (0.1f * -2.f * 0.81f);  // pair: one scale per direction
  float* gf = gflow + (long)b * 2 * cs + o;
  *reinterpret_cast<float4*>(gf) =
      make_float4(sc * acc[0] * cdx[0], sc * acc[1] * cdx[1], sc * acc[2] * cdx[2], sc * acc[3] * cdx[3]);
  *reinterpret_cast<float4*>(gf + cs) =
      make_float4(sc * acc[0] * cdy[0], sc * acc[1] * cdy[1], sc * acc[2] * cdy[2], sc * acc[3] * cdy[3]);
}

template <int R>
__global__ __launch_bounds__(NT) void bwd_kernel(const float* __restrict__ gray_a, const float* __restrict__ gray_b,
                                                 const float* __restrict__ flow, long fbs,
                                                 const float* __restrict__ dham, const float* __restrict__ scale,
                                                 float* __restrict__ gflow, int nimg, int H, int W, int pair) {
  bwd_body<R>(gray_a, gray_b, flow, fbs, dham, scale, gflow, nimg, H, W, pair);
}

// The WHOLE backward of UFlowLoss as one launch: workgroups [0, census_blocks) run the census + warp backward of both
// directions (bwd_body, pair form), the rest the smoothness backward of the level-2 flows (smooth_bwd_pixel: one row of
// 256 columns per workgroup) -- two independent kernels that used to pay two launch latencies back to back.
template <int R>
__global__ __launch_bounds__(NT) void pair_bwd_smooth_kernel(const float* __restrict__ gray, const float* __restrict__ flow,
                                                             long fbs, const float* __restrict__ dham,
                                                             const float* __restrict__ scale2, float* __restrict__ gflow,
                                                             int nimg, int H, int W, unsigned census_blocks, SmoothArgs sa,
                                                             const float* __restrict__ coef, float* __restrict__ gflow2) {
  if (blockIdx.x < census_blocks) {
    bwd_body<R>(gray, gray, flow, fbs, dham, scale2, gflow, nimg, H, W, 1);
    return;
  }
  const unsigned i = blockIdx.x - census_blocks;  // (b, y, x-block) of the level-2 grid
  const unsigned nxb = (unsigned)((sa.W + 255) / 256);
  const int xb = (int)(i % nxb), y = (int)((i / nxb) % (unsigned)sa.H), b = (int)(i / (nxb * (unsigned)sa.H));
  const int x = xb * 256 + (int)threadIdx.x;
  if (b < nimg && x < sa.W) smooth_bwd_pixel<3>(sa, coef, gflow2, b, y, x);
}

// grey plane (x255) of an RGB image and, optionally, its bilinear x1/4 copy (align_corners=False on a multiple-of-4
// grid = the mean of the central 2 x 2 of every 4 x 4 block, as down4_kernel of smooth.hip): one thread per 4 x 4
// block, every load and store a float4.
// zero (nullable): a [B, H/4, W/4] plane cleared by the same launch (the range-map accumulation target of the splat that
// follows: saves its fill launch)
__global__ __launch_bounds__(256) void down4_gray_kernel(const float* __restrict__ im, float* __restrict__ small,
                                                        float* __restrict__ gray, int H, int W,
                                                        float* __restrict__ zero = nullptr) {
  const int h = H / 4, w = W / 4;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= w) return;
  if (zero) zero[((long)b * h + y) * w + x] = 0.f;
  const long cs = (long)H * W;
  const float* p = im + (long)b * 3 * cs + (long)(4 * y) * W + 4 * x;
  float4 px[3][4];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) px[c][r] = *reinterpret_cast<const float4*>(p + c * cs + (long)r * W);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float4 g;
    g.x = ((px[0][r].x * 0.2989f + px[1][r].x * 0.5870f) + px[2][r].x * 0.1140f) * 255.f;
    g.y = ((px[0][r].y * 0.2989f + px[1][r].y * 0.5870f) + px[2][r].y * 0.1140f) * 255.f;
    g.z = ((px[0][r].z * 0.2989f + px[1][r].z * 0.5870f) + px[2][r].z * 0.1140f) * 255.f;
    g.w = ((px[0][r].w * 0.2989f + px[1][r].w * 0.5870f) + px[2][r].w * 0.1140f) * 255.f;
    *reinterpret_cast<float4*>(gray + (long)b * cs + (long)(4 * y + r) * W + 4 * x) = g;
  }
  if (small) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      small[(((long)b * 3 + c) * h + y) * w + x] =
          0.5f * (0.5f * px[c][1].y + 0.5f * px[c][1].z) + 0.5f * (0.5f * px[c][2].y + 0.5f * px[c][2].z);
  }
}

}  // namespace census_warp
}  // namespace

// Pair-symmetric kernels (census_sym.hip): every unordered pixel pair evaluated once, the value handed to the other
// end through LDS -- half the transcendentals.  Built, parity-green and MEASURED slower than the ordered-pair kernels
// of this file at the BASELINE shape (8x384x640: forward 76 us vs 58 us, backward 89 us vs 69 us), so
// they are opt-in: ARFLOW_CENSUS_SYM=1 selects them (tests run both, tools/kbench.py times both).
int census_sym_fwd(const float* gray_a, const float* gray_b, const float* flow, long fbs, const float* occ_small,
                   float* mask_out, float* dham, float* sums, int nrows, int B, int H, int W, int radius, hipStream_t st);
int census_sym_bwd(const float* gray_a, const float* gray_b, const float* flow, long fbs, const float* dham,
                   const float* scale, float* gflow, int B, int H, int W, int radius, hipStream_t st);
static bool use_sym() {
  const char* e = getenv("ARFLOW_CENSUS_SYM");
  return e && e[0] == '1';
}
// The pair-shared column kernels (census_col.hpp) are the default; ARFLOW_CENSUS_COL=0 selects the ordered-pair kernels of
// this file (tests run both, tools/kbench.py times both).
static bool use_col(bool backward = false) {
  const char* e = getenv("ARFLOW_CENSUS_COL");  // 0: neither, f: forward only, 1 / unset: both
  if (e && e[0] == '0') return false;
  if (e && e[0] == 'f') return !backward;
  return true;
}
// census_col.hip
int census_col_fwd(const float* gray_a, const float* gray_b, const float* flow, long fbs, const float* occ_small,
                   float* mask_out, float* dham, float* sums, int nrows, int B, int H, int W, int radius, int pair,
                   hipStream_t st);
int census_col_bwd(const float* gray_a, const float* gray_b, const float* flow, long fbs, const float* dham,
                   const float* scale, float* gflow, int B, int H, int W, int radius, int pair, hipStream_t st);
int census_col_pair_bwd_smooth(const float* gray, const float* flow, long fbs, const float* dham, const float* scale2,
                               float* gflow, int B2, int H, int W, int radius, const float* flow2, long flow2_bstride,
                               const float* img2, int h2, int w2, float flow_scale, float alpha, int order, int wmode,
                               int penalty, const float* coef2, float* gflow2, hipStream_t st);

extern "C" int arflow_census_warp_supported(int H, int W) { return (W % 4 == 0 && H % 4 == 0 && H >= 8 && W >= 8) ? 1 : 0; }

extern "C" int arflow_down4_gray_z(const float* im, float* small, float* gray, float* zero_plane, int B, int H, int W,
                                   arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(im);
  AF_REQUIRE_PTR(gray);
  AF_REQUIRE(B > 0 && B <= 65535 && H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0 && H / 4 <= 65535, ARFLOW_ESHAPE);
  hipLaunchKernelGGL(census_warp::down4_gray_kernel, dim3(af_cdiv(W / 4, 256), H / 4, B), dim3(256), 0,
                     (hipStream_t)stream, im, small, gray, H, W, zero_plane);
  return af_launch_status();
}
extern "C" int arflow_down4_gray(const float* im, float* small, float* gray, int B, int H, int W, arflow_stream_t stream) {
  return arflow_down4_gray_z(im, small, gray, nullptr, B, H, W, stream);
}

static int census_warp_fwd_impl(const float* gray_a, const float* gray_b, const float* flow, long flow_bstride,
                                const float* occ_small, float* mask_out, float* dham, float* sums, int B, int H,
                                int W, int radius, arflow_stream_t stream, int pair) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(gray_a);
  AF_REQUIRE_PTR(gray_b);
  AF_REQUIRE_PTR(flow);
  AF_REQUIRE_PTR(dham);
  AF_REQUIRE_PTR(sums);
  AF_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535 && H <= 8 * 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(arflow_census_warp_supported(H, W), ARFLOW_ESHAPE);
  AF_REQUIRE(flow_bstride >= 2L * H * W, ARFLOW_ESHAPE);
  AF_REQUIRE(radius >= 1 && radius <= 3, ARFLOW_EPARAM);
  hipStream_t st = (hipStream_t)stream;
  const int nrows = af_sums_rows(B, H, W);
  if (use_sym() && !pair) return census_sym_fwd(gray_a, gray_b, flow, flow_bstride, occ_small, mask_out, dham, sums, nrows, B, H, W, radius, st);
  namespace cw = census_warp;
  if (use_col())
    return census_col_fwd(gray_a, gray_b, flow, flow_bstride, occ_small, mask_out, dham, sums, nrows, B, H, W, radius, pair, st);
  dim3 g(af_grid_for_tiles((long)af_cdiv(W, cw::TXW) * af_cdiv(H, cw::TYH) * B));
  switch (radius) {
    case 1: hipLaunchKernelGGL(cw::fwd_kernel<1>, g, dim3(cw::NT), 0, st, gray_a, gray_b, flow, flow_bstride, occ_small, mask_out, dham, sums, nrows, B, H, W, pair); break;
    case 2: hipLaunchKernelGGL(cw::fwd_kernel<2>, g, dim3(cw::NT), 0, st, gray_a, gray_b, flow, flow_bstride, occ_small, mask_out, dham, sums, nrows, B, H, W, pair); break;
    default: hipLaunchKernelGGL(cw::fwd_kernel<3>, g, dim3(cw::NT), 0, st, gray_a, gray_b, flow, flow_bstride, occ_small, mask_out, dham, sums, nrows, B, H, W, pair); break;
  }
  return af_launch_status();
}

extern "C" int arflow_census_warp_fwd(const float* gray_a, const float* gray_b, const float* flow, long flow_bstride,
                                      const float* occ_small, float* mask_out, float* dham, float* sums, int B, int H,
                                      int W, int radius, arflow_stream_t stream) {
  return census_warp_fwd_impl(gray_a, gray_b, flow, flow_bstride, occ_small, mask_out, dham, sums, B, H, W, radius, stream, 0);
}
// Both directions of UFlowLoss in one launch: B = 2 x image pairs, sample s = 2 b + direction (what the model's [B,4,H,W]
// (fw, bw) flow tensor IS when viewed as [2B,2,H,W], and the [B,6,H,W] image pair viewed as [2B,3,H,W]); `gray` holds the 2B
// grey planes: image a of sample s is plane s, image b plane s ^ 1; occ_small plane s ^ 1 masks sample s; sums columns
// (0,1) belong to direction 0, (2,3) to direction 1; `scale` of the backward holds one factor per direction.
extern "C" int arflow_census_warp_pair_fwd(const float* gray, const float* flow, long flow_bstride, const float* occ_small,
                                           float* mask_out, float* dham, float* sums, int B2, int H, int W, int radius,
                                           arflow_stream_t stream) {
  AF_REQUIRE(B2 % 2 == 0, ARFLOW_ESHAPE);
  return census_warp_fwd_impl(gray, gray, flow, flow_bstride, occ_small, mask_out, dham, sums, B2, H, W, radius, stream, 1);
}

static int census_warp_bwd_impl(const float* gray_a, const float* gray_b, const float* flow, long flow_bstride,
                                const float* dham, const float* scale, float* gflow, int B, int H, int W,
                                int radius, arflow_stream_t stream, int pair) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(gray_a);
  AF_REQUIRE_PTR(gray_b);
  AF_REQUIRE_PTR(flow);
  AF_REQUIRE_PTR(dham);
  AF_REQUIRE_PTR(gflow);
  AF_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535 && H <= 8 * 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(arflow_census_warp_supported(H, W), ARFLOW_ESHAPE);
  AF_REQUIRE(flow_bstride >= 2L * H * W, ARFLOW_ESHAPE);
  AF_REQUIRE(radius >= 1 && radius <= 3, ARFLOW_EPARAM);
  hipStream_t st = (hipStream_t)stream;
  if (use_sym() && !pair) return census_sym_bwd(gray_a, gray_b, flow, flow_bstride, dham, scale, gflow, B, H, W, radius, st);
  namespace cw = census_warp;
  if (use_col(true)) return census_col_bwd(gray_a, gray_b, flow, flow_bstride, dham, scale, gflow, B, H, W, radius, pair, st);
  dim3 g(af_grid_for_tiles((long)af_cdiv(W, cw::TXW) * af_cdiv(H, cw::TYH) * B));
  switch (radius) {
    case 1: hipLaunchKernelGGL(cw::bwd_kernel<1>, g, dim3(cw::NT), 0, st, gray_a, gray_b, flow, flow_bstride, dham, scale, gflow, B, H, W, pair); break;
    case 2: hipLaunchKernelGGL(cw::bwd_kernel<2>, g, dim3(cw::NT), 0, st, gray_a, gray_b, flow, flow_bstride, dham, scale, gflow, B, H, W, pair); break;
    default: hipLaunchKernelGGL(cw::bwd_kernel<3>, g, dim3(cw::NT), 0, st, gray_a, gray_b, flow, flow_bstride, dham, scale, gflow, B, H, W, pair); break;
  }
  return af_launch_status();
}

extern "C" int arflow_census_warp_bwd(const float* gray_a, const float* gray_b, const float* flow, long flow_bstride,
                                      const float* dham, const float* scale, float* gflow, int B, int H, int W,
                                      int radius, arflow_stream_t stream) {
  return census_warp_bwd_impl(gray_a, gray_b, flow, flow_bstride, dham, scale, gflow, B, H, W, radius, stream, 0);
}
extern "C" int arflow_census_warp_pair_bwd(const float* gray, const float* flow, long flow_bstride, const float* dham,
                                           const float* scale2, float* gflow, int B2, int H, int W, int radius,
                                           arflow_stream_t stream) {
  AF_REQUIRE(B2 % 2 == 0, ARFLOW_ESHAPE);
  AF_REQUIRE_PTR(scale2);
  return census_warp_bwd_impl(gray, gray, flow, flow_bstride, dham, scale2, gflow, B2, H, W, radius, stream, 1);
}

// Backward of BOTH loss terms of UFlowLoss in one launch: arflow_census_warp_pair_bwd + arflow_smooth_bwd (3-channel image,
// level-2 flows [B2,2,h2,w2] with batch stride flow2_bstride); coef2 = the two smoothness-sum gradients.
extern "C" int arflow_uflow_pair_bwd(const float* gray, const float* flow, long flow_bstride, const float* dham,
                                     const float* scale2, float* gflow, int B2, int H, int W, int radius, const float* flow2,
                                     long flow2_bstride, const float* img2, const float* coef2, float* gflow2, int h2, int w2,
                                     float flow_scale, float alpha, int order, int wmode, int penalty,
                                     arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(gray);
  AF_REQUIRE_PTR(flow);
  AF_REQUIRE_PTR(dham);
  AF_REQUIRE_PTR(scale2);
  AF_REQUIRE_PTR(gflow);
  AF_REQUIRE_PTR(flow2);
  AF_REQUIRE_PTR(img2);
  AF_REQUIRE_PTR(coef2);
  AF_REQUIRE_PTR(gflow2);
  AF_REQUIRE(B2 > 0 && B2 % 2 == 0 && H > 0 && W > 0 && B2 <= 65535 && H <= 8 * 65535 && h2 > 0 && w2 > 0, ARFLOW_ESHAPE);
  AF_REQUIRE(arflow_census_warp_supported(H, W), ARFLOW_ESHAPE);
  AF_REQUIRE(flow_bstride >= 2L * H * W && flow2_bstride >= 2L * h2 * w2, ARFLOW_ESHAPE);
  AF_REQUIRE(radius >= 1 && radius <= 3, ARFLOW_EPARAM);
  AF_REQUIRE((order == 1 || order == 2) && (wmode == 0 || wmode == 1) && (penalty == 0 || penalty == 1), ARFLOW_EPARAM);
  namespace cw = census_warp;
  const SmoothArgs sa{flow2, img2, 3, h2, w2, flow2_bstride, flow_scale, alpha, order, wmode, penalty};
  hipStream_t st = (hipStream_t)stream;
  if (use_col(true)) return census_col_pair_bwd_smooth(gray, flow, flow_bstride, dham, scale2, gflow, B2, H, W, radius, flow2, flow2_bstride, img2, h2, w2,
                                                         flow_scale, alpha, order, wmode, penalty, coef2, gflow2, st);
  const unsigned cb = af_grid_for_tiles((long)af_cdiv(W, cw::TXW) * af_cdiv(H, cw::TYH) * B2);
  const unsigned sb = (unsigned)af_cdiv(w2, 256) * (unsigned)h2 * (unsigned)B2;
  switch (radius) {
    case 1: hipLaunchKernelGGL(cw::pair_bwd_smooth_kernel<1>, dim3(cb + sb), dim3(cw::NT), 0, st, gray, flow, flow_bstride, dham, scale2, gflow, B2, H, W, cb, sa, coef2, gflow2); break;
    case 2: hipLaunchKernelGGL(cw::pair_bwd_smooth_kernel<2>, dim3(cb + sb), dim3(cw::NT), 0, st, gray, flow, flow_bstride, dham, scale2, gflow, B2, H, W, cb, sa, coef2, gflow2); break;
    default: hipLaunchKernelGGL(cw::pair_bwd_smooth_kernel<3>, dim3(cb + sb), dim3(cw::NT), 0, st, gray, flow, flow_bstride, dham, scale2, gflow, B2, H, W, cb, sa, coef2, gflow2); break;
  }
  return af_launch_status();
}
