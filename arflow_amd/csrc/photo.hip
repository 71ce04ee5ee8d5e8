// Census kernels for gfx950: soft census (ternary) distance with fused census_loss reduction, forward and
// backward (the SSIM + L1 term of the ARFlow pyramid loss lives in ssim.hip).
//
// Reference arithmetic: utils/uflow_utils.py (rgb_to_grayscale :227-231, census_transform :241-261,
// soft_hamming :264-279, zero_mask_border :234-238, abs_robust_loss :213-214, census_loss :282-293),
// losses/loss_blocks.py (TernaryLoss :12-62, SSIM :65-84), losses/flow_loss.py:13-27.
//
// The reference materialises ~25 full-resolution 49-channel temporaries per census_loss call
// (385 MB each at B=8, 384x640).  Here the grey tiles (+3 px halo) live in LDS, the 49 neighbour
// comparisons run in registers, and the backward pass recomputes them instead of storing anything:
// HBM traffic is 7 floats in + 1-2 floats out per pixel forward, 8 in + 3 out backward.  These
// kernels are transcendental-bound (2 rsq + 1 rcp per neighbour), not HBM-bound.
#include "common.hpp"
#include "census_tile.hpp"

namespace {

constexpr int TX = 32, TY = 8;  // pixel tile = 256 threads, lanes run along x

__device__ __forceinline__ float gray255(const float* __restrict__ im, long cs, long off) {
  // ((r*0.2989 + g*0.5870) + b*0.1140) * 255, the reference's operation order
  return ((im[off] * 0.2989f + im[off + cs] * 0.5870f) + im[off + 2 * cs] * 0.1140f) * 255.f;
}

template <int R>
__device__ __forceinline__ void load_gray_tile(float (*tile)[TX + 2 * R + 1], const float* __restrict__ im,
                                               int H, int W, int ty0, int tx0) {
  constexpr int TR = TY + 2 * R, TC = TX + 2 * R;
  const long cs = (long)H * W;
  for (int idx = threadIdx.x; idx < TR * TC; idx += TX * TY) {
    const int r = idx / TC, c = idx - r * TC;
    const int gy = ty0 + r - R, gx = tx0 + c - R;
    float v = 0.f;  // zero padding of the intensities (conv2d padding, uflow_utils.py:257)
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = gray255(im, cs, (long)gy * W + gx);
    tile[r][c] = v;
  }
}

// ham = sum_k sq/(0.1+sq); optional fused census_loss partial sums.
template <int R>
__global__ __launch_bounds__(TX* TY) void census_fwd_kernel(const float* __restrict__ im_a,
                                                            const float* __restrict__ im_b,
                                                            const float* __restrict__ mask,
                                                            float* __restrict__ ham_out,
                                                            float* __restrict__ dham_out,
                                                            float* __restrict__ sums, int nrows, int nimg, int H, int W) {
  __shared__ float ga[TY + 2 * R][TX + 2 * R + 1];
  __shared__ float gb[TY + 2 * R][TX + 2 * R + 1];
  __shared__ float red[2 * (TX * TY / 64)];
  int btx_, bty_, b;
  if (!af_tile_of_block((W + TX - 1) / TX, (H + TY - 1) / TY, nimg, btx_, bty_, b)) {
    if (mask && threadIdx.x == 0) af_store_partial(sums, nrows, 0.f, 0.f, 0.f);  // padding workgroup: its row must be defined
    return;
  }
  const int ty0 = bty_ * TY, tx0 = btx_ * TX;
  const long ims = 3L * H * W;
  load_gray_tile<R>(ga, im_a + b * ims, H, W, ty0, tx0);
  load_gray_tile<R>(gb, im_b + b * ims, H, W, ty0, tx0);
  __syncthreads();
  const int lx = threadIdx.x % TX, ly = threadIdx.x / TX;
  const int x = tx0 + lx, y = ty0 + ly;
  const bool inside = x < W && y < H;
  const float ca = ga[ly + R][lx + R], cb = gb[ly + R][lx + R];
  float s = 0.f;
#pragma unroll
  for (int dy = 0; dy <= 2 * R; ++dy)
#pragma unroll
    for (int dx = 0; dx <= 2 * R; ++dx) {
      const float da = ga[ly + dy][lx + dx] - ca, db = gb[ly + dy][lx + dx] - cb;
      const float ta = da * __builtin_amdgcn_rsqf(fmaf(da, da, 0.81f));
      const float tb = db * __builtin_amdgcn_rsqf(fmaf(db, db, 0.81f));
      const float e = ta - tb, sq = e * e;
      s = fmaf(sq, __builtin_amdgcn_rcpf(0.1f + sq), s);
    }
  float part[2] = {0.f, 0.f};
  if (inside) {
    const long o = ((long)b * H + y) * W + x;
    if (ham_out) ham_out[o] = s;
    if (mask) {
      const bool interior = x >= R && x < W - R && y >= R && y < H - R;
      const float pm = interior ? mask[o] : 0.f;
      const float base = fabsf(s) + 0.01f;
      const float lg = __log2f(base);
      part[0] = exp2f(0.4f * lg) * pm;  // (|ham|+0.01)^0.4
      part[1] = pm;
      if (dham_out) dham_out[o] = pm * 0.4f * exp2f(-0.6f * lg);
    }
  }
  if (mask) {
    af_block_sum<2>(part, red);
    if (threadIdx.x == 0) af_store_partial(sums, nrows, part[0], part[1], 0.f);
  }
}

// g_im_b = scale * 255 * (0.2989,0.587,0.114) * sum_{k!=0} (G(r-k)+G(r)) * Hd(A[r]-A[r-k], B[r]-B[r-k])
// (see DESIGN.md "census backward": the centre term of pixel r for neighbour -k equals the
// neighbour term of pixel r for centre r-k because d ham / d d_b is odd in (d_a, d_b)).
template <int R>
__global__ __launch_bounds__(TX* TY) void census_bwd_kernel(const float* __restrict__ im_a,
                                                            const float* __restrict__ im_b,
                                                            const float* __restrict__ gham,
                                                            const float* __restrict__ scale,
                                                            float* __restrict__ g_im_b, int nimg, int H, int W) {
  __shared__ float ga[TY + 2 * R][TX + 2 * R + 1];
  __shared__ float gb[TY + 2 * R][TX + 2 * R + 1];
  __shared__ float gg[TY + 2 * R][TX + 2 * R + 1];
  int btx_, bty_, b;
  if (!af_tile_of_block((W + TX - 1) / TX, (H + TY - 1) / TY, nimg, btx_, bty_, b)) return;
  const int ty0 = bty_ * TY, tx0 = btx_ * TX;
  const long ims = 3L * H * W;
  load_gray_tile<R>(ga, im_a + b * ims, H, W, ty0, tx0);
  load_gray_tile<R>(gb, im_b + b * ims, H, W, ty0, tx0);
  {
    constexpr int TR = TY + 2 * R, TC = TX + 2 * R;
    const float* g = gham + (long)b * H * W;
    for (int idx = threadIdx.x; idx < TR * TC; idx += TX * TY) {
      const int r = idx / TC, c = idx - r * TC;
      const int gy = ty0 + r - R, gx = tx0 + c - R;
      gg[r][c] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? g[(long)gy * W + gx] : 0.f;
    }
  }
  __syncthreads();
  const int lx = threadIdx.x % TX, ly = threadIdx.x / TX;
  const int x = tx0 + lx, y = ty0 + ly;
  if (x >= W || y >= H) return;
  const float ca = ga[ly + R][lx + R], cb = gb[ly + R][lx + R], cg = gg[ly + R][lx + R];
  float acc = 0.f;
#pragma unroll
  for (int dy = 0; dy <= 2 * R; ++dy)
#pragma unroll
    for (int dx = 0; dx <= 2 * R; ++dx) {
      if (dy == R && dx == R) continue;
      // tile index (ly+dy, lx+dx) is pixel r - k with k = (R-dy, R-dx)
      const float da = ca - ga[ly + dy][lx + dx], db = cb - gb[ly + dy][lx + dx];
      const float ua = __builtin_amdgcn_rsqf(fmaf(da, da, 0.81f));
      const float ub = __builtin_amdgcn_rsqf(fmaf(db, db, 0.81f));
      const float e = da * ua - db * ub, sq = e * e;
      const float q = __builtin_amdgcn_rcpf(0.1f + sq);
      // d h/d sq = 0.1 q^2 ; d sq/d tb = -2e ; d tb/d db = 0.81 ub^3
      const float hd = (0.1f * q * q) * (-2.f * e) * (0.81f * ub * ub * ub);
      acc = fmaf(gg[ly + dy][lx + dx] + cg, hd, acc);
    }
  const float sc = (scale ? scale[0] : 1.f) * 255.f * acc;
  float* o = g_im_b + b * ims + (long)y * W + x;
  const long cs = (long)H * W;
  o[0] = sc * 0.2989f;
  o[cs] = sc * 0.5870f;
  o[2 * cs] = sc * 0.1140f;
}

// ------------------------------------------------------------------------------------------------
// Census, 4 pixels per lane (used when W % 4 == 0): tile 16 rows x 64 columns per 256-thread workgroup.
//   * tiles are filled with aligned float4 global loads (3 planes -> grey in registers -> one float4 LDS
//     store) instead of one bounds-checked dword per element;
//   * a lane reads each of the 2R+1 window rows as three ds_read_b128 (12 floats starting 4 columns left
//     of its pixel group) and reuses them for its 4 pixels: 9x fewer LDS instructions than one b32 per
//     (pixel, neighbour), and 4 independent accumulation chains per lane;
//   * row pitch 128 floats (a multiple of the 64-bank row) keeps those b128 reads conflict-free for the
//     lane -> (pixel group = lane % 16, row = lane / 16) map.
// ------------------------------------------------------------------------------------------------
namespace census4 {
template <int R>
__global__ __launch_bounds__(NT) void fwd_kernel(const float* __restrict__ im_a, const float* __restrict__ im_b,
                                                 const float* __restrict__ mask, float* __restrict__ ham_out,
                                                 float* __restrict__ dham_out, float* __restrict__ sums, int nrows, int nimg,
                                                 int H, int W) {
  __shared__ __attribute__((aligned(16))) float ga[ROWS * PITCH];
  __shared__ __attribute__((aligned(16))) float gb[ROWS * PITCH];
  __shared__ float red[2 * (NT / 64)];
  int btx, bty, b;
  if (!af_tile_of_block((W + TXW - 1) / TXW, (H + TYH - 1) / TYH, nimg, btx, bty, b)) {
    if (mask && threadIdx.x == 0) af_store_partial(sums, nrows, 0.f, 0.f, 0.f);  // padding workgroup: its row must be defined
    return;
  }
  const int ty0 = bty * TYH, tx0 = btx * TXW;
  const long ims = 3L * H * W;
  load_gray<R>(ga, im_a + b * ims, H, W, ty0, tx0);
  load_gray<R>(gb, im_b + b * ims, H, W, ty0, tx0);
  __syncthreads();
  const int xg = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int x0 = tx0 + 4 * xg, y = ty0 + ly;
  // window row `dy` of this lane: tile row ly+dy, columns 4*xg .. 4*xg+11 (= image x0-4 .. x0+7)
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
        const int k = p + dx + 4 - R;  // neighbour column x0+p+dx-R  <->  window index (x0+p+dx-R) - (x0-4)
        const float da = wa[k] - ca[p], db = wb[k] - cb[p];
        const float tb = db * __builtin_amdgcn_rsqf(fmaf(db, db, 0.81f));
        const float e = fmaf(da, __builtin_amdgcn_rsqf(fmaf(da, da, 0.81f)), -tb), sq = e * e;
        s[p] = fmaf(sq, __builtin_amdgcn_rcpf(0.1f + sq), s[p]);
      }
  }
  float part[2] = {0.f, 0.f};
  if (y < H && x0 < W) {  // W % 4 == 0: the 4 pixels are inside together
    const long o = ((long)b * H + y) * W + x0;
    if (ham_out) *reinterpret_cast<float4*>(ham_out + o) = make_float4(s[0], s[1], s[2], s[3]);
    if (mask) {
      const float4 mk = *reinterpret_cast<const float4*>(mask + o);
      const float mv[4] = {mk.x, mk.y, mk.z, mk.w};
      float dh[4];
      const bool rowin = y >= R && y < H - R;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int xx = x0 + p;
        const float pm = (rowin && xx >= R && xx < W - R) ? mv[p] : 0.f;
        const float lg = __log2f(fabsf(s[p]) + 0.01f);
        part[0] += exp2f(0.4f * lg) * pm;
        part[1] += pm;
        dh[p] = pm * 0.4f * exp2f(-0.6f * lg);
      }
      if (dham_out) *reinterpret_cast<float4*>(dham_out + o) = make_float4(dh[0], dh[1], dh[2], dh[3]);
    }
  }
  if (mask) {
    af_block_sum<2>(part, red);
    if (threadIdx.x == 0) af_store_partial(sums, nrows, part[0], part[1], 0.f);
  }
}

template <int R>
__global__ __launch_bounds__(NT) void bwd_kernel(const float* __restrict__ im_a, const float* __restrict__ im_b,
                                                 const float* __restrict__ gham, const float* __restrict__ scale,
                                                 float* __restrict__ g_im_b, int nimg, int H, int W) {
  __shared__ __attribute__((aligned(16))) float ga[ROWS * PITCH];
  __shared__ __attribute__((aligned(16))) float gb[ROWS * PITCH];
  __shared__ __attribute__((aligned(16))) float gg[ROWS * PITCH];
  int btx, bty, b;
  if (!af_tile_of_block((W + TXW - 1) / TXW, (H + TYH - 1) / TYH, nimg, btx, bty, b)) return;
  const int ty0 = bty * TYH, tx0 = btx * TXW;
  const long ims = 3L * H * W;
  load_gray<R>(ga, im_a + b * ims, H, W, ty0, tx0);
  load_gray<R>(gb, im_b + b * ims, H, W, ty0, tx0);
  {
    constexpr int NR = TYH + 2 * R, NQ = (TXW + 8) / 4;
    const float* g = gham + (long)b * H * W;
    for (int i = threadIdx.x; i < NR * NQ; i += NT) {
      const int r = i / NQ, q = i - r * NQ;
      const int gy = ty0 - R + r, gx = tx0 - 4 + 4 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *reinterpret_cast<const float4*>(g + (long)gy * W + gx);
      *reinterpret_cast<float4*>(gg + r * PITCH + 4 * q) = v;
    }
  }
  __syncthreads();
  const int xg = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int x0 = tx0 + 4 * xg, y = ty0 + ly;
  if (y >= H || x0 >= W) return;
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
        const int k = p + dx + 4 - R;  // tile pixel r - kk, kk = (R-dy, R-dx)
        // d ham / d t_b = (0.1 q^2) (-2 e) (0.81 ub^3), q = 1/(0.1 + e^2): the constant -0.162 is applied
        // once at the store; 17 VALU slots per tap (3 of them transcendental) -- the kernel is VALU-bound
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
  const float sc = (scale ? scale[0] : 1.f) * 255.f * (0.1f * -2.f * 0.81f);
  float* o = g_im_b + b * ims + (long)y * W + x0;
  const long cs = (long)H * W;
  *reinterpret_cast<float4*>(o) =
      make_float4(sc * acc[0] * 0.2989f, sc * acc[1] * 0.2989f, sc * acc[2] * 0.2989f, sc * acc[3] * 0.2989f);
  *reinterpret_cast<float4*>(o + cs) =
      make_float4(sc * acc[0] * 0.5870f, sc * acc[1] * 0.5870f, sc * acc[2] * 0.5870f, sc * acc[3] * 0.5870f);
  *reinterpret_cast<float4*>(o + 2 * cs) =
      make_float4(sc * acc[0] * 0.1140f, sc * acc[1] * 0.1140f, sc * acc[2] * 0.1140f, sc * acc[3] * 0.1140f);
}
}  // namespace census4

}  // namespace

// radius > 3: the one-thread-per-pixel kernels of generic.hip (TernaryLoss(max_distance > 3), no shipped config)
int census_any_fwd(const float* im_a, const float* im_b, const float* mask, float* ham, float* dham, float* sums, int nrows,
                   int B, int H, int W, int R, hipStream_t st);
int census_any_bwd(const float* im_a, const float* im_b, const float* gham, const float* scale, float* g_im_b, int B, int H,
                   int W, int R, hipStream_t st);

extern "C" int arflow_census_fwd(const float* im_a, const float* im_b, const float* mask, float* ham,
                                 float* dham, float* sums, int B, int H, int W, int radius,
                                 arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(im_a);
  AF_REQUIRE_PTR(im_b);
  AF_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535 && H <= 8 * 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(radius >= 1 && radius <= 16, ARFLOW_EPARAM);
  if (mask) AF_REQUIRE_PTR(sums);
  hipStream_t st = (hipStream_t)stream;
  const int nrows = af_sums_rows(B, H, W);
  if (radius > 3) {
    AF_REQUIRE(H <= 65535, ARFLOW_ESHAPE);
    return census_any_fwd(im_a, im_b, mask, ham, dham, sums, nrows, B, H, W, radius, st);
  }
  if ((W & 3) == 0) {
    namespace c4 = census4;
    dim3 g4(af_grid_for_tiles((long)af_cdiv(W, c4::TXW) * af_cdiv(H, c4::TYH) * B));
    switch (radius) {
      case 1: hipLaunchKernelGGL(c4::fwd_kernel<1>, g4, dim3(c4::NT), 0, st, im_a, im_b, mask, ham, dham, sums, nrows, B, H, W); break;
      case 2: hipLaunchKernelGGL(c4::fwd_kernel<2>, g4, dim3(c4::NT), 0, st, im_a, im_b, mask, ham, dham, sums, nrows, B, H, W); break;
      default: hipLaunchKernelGGL(c4::fwd_kernel<3>, g4, dim3(c4::NT), 0, st, im_a, im_b, mask, ham, dham, sums, nrows, B, H, W); break;
    }
    return af_launch_status();
  }
  dim3 grid(af_grid_for_tiles((long)af_cdiv(W, TX) * af_cdiv(H, TY) * B));
  switch (radius) {
    case 1: hipLaunchKernelGGL(census_fwd_kernel<1>, grid, dim3(TX * TY), 0, st, im_a, im_b, mask, ham, dham, sums, nrows, B, H, W); break;
    case 2: hipLaunchKernelGGL(census_fwd_kernel<2>, grid, dim3(TX * TY), 0, st, im_a, im_b, mask, ham, dham, sums, nrows, B, H, W); break;
    default: hipLaunchKernelGGL(census_fwd_kernel<3>, grid, dim3(TX * TY), 0, st, im_a, im_b, mask, ham, dham, sums, nrows, B, H, W); break;
  }
  return af_launch_status();
}

extern "C" int arflow_census_bwd(const float* im_a, const float* im_b, const float* gham, const float* scale,
                                 float* g_im_b, int B, int H, int W, int radius, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(im_a);
  AF_REQUIRE_PTR(im_b);
  AF_REQUIRE_PTR(gham);
  AF_REQUIRE_PTR(g_im_b);
  AF_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535 && H <= 8 * 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(radius >= 1 && radius <= 16, ARFLOW_EPARAM);
  hipStream_t st = (hipStream_t)stream;
  if (radius > 3) {
    AF_REQUIRE(H <= 65535, ARFLOW_ESHAPE);
    return census_any_bwd(im_a, im_b, gham, scale, g_im_b, B, H, W, radius, st);
  }
  if ((W & 3) == 0) {
    namespace c4 = census4;
    dim3 g4(af_grid_for_tiles((long)af_cdiv(W, c4::TXW) * af_cdiv(H, c4::TYH) * B));
    switch (radius) {
      case 1: hipLaunchKernelGGL(c4::bwd_kernel<1>, g4, dim3(c4::NT), 0, st, im_a, im_b, gham, scale, g_im_b, B, H, W); break;
      case 2: hipLaunchKernelGGL(c4::bwd_kernel<2>, g4, dim3(c4::NT), 0, st, im_a, im_b, gham, scale, g_im_b, B, H, W); break;
      default: hipLaunchKernelGGL(c4::bwd_kernel<3>, g4, dim3(c4::NT), 0, st, im_a, im_b, gham, scale, g_im_b, B, H, W); break;
    }
    return af_launch_status();
  }
  dim3 grid(af_grid_for_tiles((long)af_cdiv(W, TX) * af_cdiv(H, TY) * B));
  switch (radius) {
    case 1: hipLaunchKernelGGL(census_bwd_kernel<1>, grid, dim3(TX * TY), 0, st, im_a, im_b, gham, scale, g_im_b, B, H, W); break;
    case 2: hipLaunchKernelGGL(census_bwd_kernel<2>, grid, dim3(TX * TY), 0, st, im_a, im_b, gham, scale, g_im_b, B, H, W); break;
    default: hipLaunchKernelGGL(census_bwd_kernel<3>, grid, dim3(TX * TY), 0, st, im_a, im_b, gham, scale, g_im_b, B, H, W); break;
  }
  return af_launch_status();
}

