// Photometric loss kernels for gfx950: soft census (ternary) distance with fused census_loss
// reduction, and the SSIM + L1 term of the ARFlow pyramid loss -- forward and backward.
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
                                                            float* __restrict__ sums, int nimg, int H, int W) {
  __shared__ float ga[TY + 2 * R][TX + 2 * R + 1];
  __shared__ float gb[TY + 2 * R][TX + 2 * R + 1];
  __shared__ float red[2 * (TX * TY / 64)];
  int btx_, bty_, b;
  if (!af_tile_of_block((W + TX - 1) / TX, (H + TY - 1) / TY, nimg, btx_, bty_, b)) return;
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
    if (threadIdx.x == 0) {
      float* slot = af_sum_slot(sums);
      atomicAdd(slot, part[0]);
      atomicAdd(slot + 1, part[1]);
    }
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
                                                 float* __restrict__ dham_out, float* __restrict__ sums, int nimg,
                                                 int H, int W) {
  __shared__ __attribute__((aligned(16))) float ga[ROWS * PITCH];
  __shared__ __attribute__((aligned(16))) float gb[ROWS * PITCH];
  __shared__ float red[2 * (NT / 64)];
  int btx, bty, b;
  if (!af_tile_of_block((W + TXW - 1) / TXW, (H + TYH - 1) / TYH, nimg, btx, bty, b)) return;
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
    if (threadIdx.x == 0) {
      float* slot = af_sum_slot(sums);
      atomicAdd(slot, part[0]);
      atomicAdd(slot + 1, part[1]);
    }
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

// ------------------------------------------------------------------------------------------------
// SSIM (3x3, un-padded) + L1
// ------------------------------------------------------------------------------------------------
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;

struct Win {
  float mx, my, sx, sy, sxy;
};

// x / 9 exactly as IEEE division rounds it, in 3 VALU instructions instead of the ~12 of the generic
// sequence: q = x*c; r = fma(-9, q, x); q = fma(r, c, q) with c = RN(1/9).  Bit-identical to x / 9.0f for
// every finite float (all 2^32 patterns compared on the GPU, tools/ubench/div9_check.hip; only +-inf and one
// value next to overflow differ).  The SSIM kernels are VALU-bound and did six divisions per window.
__device__ __forceinline__ float div9(float x) {
  const float c = 1.0f / 9.0f;
  float q = x * c;
  const float r = fmaf(-9.0f, q, x);
  return fmaf(r, c, q);
}
// n / d and 1 / d for d > 0 (the SSIM denominators are >= C1*C2 > 0): hardware reciprocal + one Newton
// step, within 1 ulp of the IEEE quotient (enters (1 - n/d)/2 with absolute error <= 6e-8).
__device__ __forceinline__ float fdiv_pos(float n, float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  const float q = n * r;
  return fmaf(fmaf(-d, q, n), r, q);
}
__device__ __forceinline__ float frcp_pos(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}

template <int PITCH>
__device__ __forceinline__ Win window_stats(const float (*tx)[PITCH], const float (*ty)[PITCH], int r, int c) {
  float sxv = 0.f, syv = 0.f, sxx = 0.f, syy = 0.f, sxyv = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float a = tx[r + i][c + j], b = ty[r + i][c + j];
      // the reference pools the already-rounded products x*x, y*y, x*y (AvgPool2d of a product
      // tensor, loss_blocks.py:76-78): round each product, add in row-major order, divide by 9.
      // sigma = E[x^2]-mu^2 cancels catastrophically, so the operation order is kept.
      sxv += a;
      syv += b;
      sxx += a * a;
      syy += b * b;
      sxyv += a * b;
    }
  Win w;
  w.mx = div9(sxv);
  w.my = div9(syv);
  w.sx = div9(sxx) - w.mx * w.mx;
  w.sy = div9(syy) - w.my * w.my;
  w.sxy = div9(sxyv) - w.mx * w.my;
  return w;
}

__global__ __launch_bounds__(TX* TY) void photo_fwd_kernel(const float* __restrict__ im,
                                                           const float* __restrict__ rec,
                                                           const float* __restrict__ mask,
                                                           float* __restrict__ ssim_map,
                                                           float* __restrict__ sums, int nimg, int C, int H, int W) {
  __shared__ float tx[TY + 2][TX + 3];  // x = recons*mask
  __shared__ float ty[TY + 2][TX + 3];  // y = im*mask
  __shared__ float red[3 * (TX * TY / 64)];
  int btx_, bty_, b;
  if (!af_tile_of_block((W + TX - 1) / TX, (H + TY - 1) / TY, nimg, btx_, bty_, b)) return;
  const int ty0 = bty_ * TY, tx0 = btx_ * TX;
  const long cs = (long)H * W;
  const int lx = threadIdx.x % TX, ly = threadIdx.x / TX;
  const int x = tx0 + lx, y = ty0 + ly;
  float part[3] = {0.f, 0.f, 0.f};
  if (x < W && y < H) part[2] = mask ? mask[(long)b * cs + (long)y * W + x] : 1.f;
  for (int c = 0; c < C; ++c) {
    const float* imc = im + ((long)b * C + c) * cs;
    const float* rc = rec + ((long)b * C + c) * cs;
    __syncthreads();
    for (int idx = threadIdx.x; idx < (TY + 2) * (TX + 2); idx += TX * TY) {
      const int r = idx / (TX + 2), cc = idx - r * (TX + 2);
      const int gy = ty0 + r, gx = tx0 + cc;
      float a = 0.f, bb = 0.f;
      if (gy < H && gx < W) {
        const long o = (long)gy * W + gx;
        const float m = mask ? mask[(long)b * cs + o] : 1.f;
        const float iv = imc[o], rv = rc[o];
        a = rv * m;
        bb = iv * m;
        if (r < TY && cc < TX) part[0] += fabsf(iv - rv) * m;  // each pixel owned by exactly one tile slot
      }
      tx[r][cc] = a;
      ty[r][cc] = bb;
    }
    __syncthreads();
    if (x < W - 2 && y < H - 2) {
      const Win w = window_stats<TX + 3>(tx, ty, ly, lx);
      const float n = (2.f * w.mx * w.my + SSIM_C1) * (2.f * w.sxy + SSIM_C2);
      const float d = (w.mx * w.mx + w.my * w.my + SSIM_C1) * (w.sx + w.sy + SSIM_C2);
      const float dist = fminf(fmaxf((1.f - fdiv_pos(n, d)) / 2.f, 0.f), 1.f);
      part[1] += dist;
      if (ssim_map) ssim_map[(((long)b * C + c) * (H - 2) + y) * (W - 2) + x] = dist;
    }
  }
  af_block_sum<3>(part, red);
  if (threadIdx.x == 0) {
    float* slot = af_sum_slot(sums);
    atomicAdd(slot, part[0]);
    atomicAdd(slot + 1, part[1]);
    atomicAdd(slot + 2, part[2]);
  }
}

// d dist_w / d x_r = -(1/2) (alpha_w + beta_w x_r + gamma_w y_r) where 0 <= (1-S)/2 <= 1, else 0.
__global__ __launch_bounds__(TX* TY) void photo_bwd_kernel(const float* __restrict__ im,
                                                           const float* __restrict__ rec,
                                                           const float* __restrict__ mask,
                                                           const float* __restrict__ gmap,
                                                           const float* __restrict__ coef,
                                                           float* __restrict__ g_rec, int nimg, int C, int H, int W) {
  // data region: pixels (ty0-2 .. ty0+TY+1) x (tx0-2 .. tx0+TX+1); windows anchored at
  // (ty0-2 .. ty0+TY-1) x (tx0-2 .. tx0+TX-1)
  __shared__ float dx_[TY + 4][TX + 5];
  __shared__ float dy_[TY + 4][TX + 5];
  __shared__ float wa[TY + 2][TX + 3];
  __shared__ float wb[TY + 2][TX + 3];
  __shared__ float wc[TY + 2][TX + 3];
  int btx_, bty_, b;
  if (!af_tile_of_block((W + TX - 1) / TX, (H + TY - 1) / TY, nimg, btx_, bty_, b)) return;
  const int ty0 = bty_ * TY, tx0 = btx_ * TX;
  const long cs = (long)H * W;
  const int lx = threadIdx.x % TX, ly = threadIdx.x / TX;
  const int x = tx0 + lx, y = ty0 + ly;
  const float c_l1 = coef[0], c_ss = coef[1];
  for (int c = 0; c < C; ++c) {
    const float* imc = im + ((long)b * C + c) * cs;
    const float* rc = rec + ((long)b * C + c) * cs;
    __syncthreads();
    for (int idx = threadIdx.x; idx < (TY + 4) * (TX + 4); idx += TX * TY) {
      const int r = idx / (TX + 4), cc = idx - r * (TX + 4);
      const int gy = ty0 + r - 2, gx = tx0 + cc - 2;
      float a = 0.f, bb = 0.f;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        const long o = (long)gy * W + gx;
        const float m = mask ? mask[(long)b * cs + o] : 1.f;
        a = rc[o] * m;
        bb = imc[o] * m;
      }
      dx_[r][cc] = a;
      dy_[r][cc] = bb;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < (TY + 2) * (TX + 2); idx += TX * TY) {
      const int r = idx / (TX + 2), cc = idx - r * (TX + 2);
      const int wy = ty0 + r - 2, wx = tx0 + cc - 2;  // window anchor
      float A = 0.f, Bc = 0.f, Cc = 0.f;
      if (wy >= 0 && wy < H - 2 && wx >= 0 && wx < W - 2) {
        const Win w = window_stats<TX + 5>(dx_, dy_, r, cc);
        const float n1 = 2.f * w.mx * w.my + SSIM_C1, n2 = 2.f * w.sxy + SSIM_C2;
        const float d1 = w.mx * w.mx + w.my * w.my + SSIM_C1, d2 = w.sx + w.sy + SSIM_C2;
        const float n = n1 * n2, d = d1 * d2;
        const float v = (1.f - fdiv_pos(n, d)) / 2.f;
        if (v >= 0.f && v <= 1.f) {  // torch.clamp passes the gradient on the closed interval
          const float up = gmap ? gmap[(((long)b * C + c) * (H - 2) + wy) * (W - 2) + wx] : c_ss;
          const float k = -0.5f * up * (2.f / 9.f);
          const float id = frcp_pos(d), nd2 = n * id * id;
          Cc = k * n1 * id;                                                     // * y_r
          Bc = -k * nd2 * d1;                                                   // * x_r
          A = k * ((w.my * n2 - n1 * w.my) * id - nd2 * (w.mx * d2 - d1 * w.mx));  // constant
        }
      }
      wa[r][cc] = A;
      wb[r][cc] = Bc;
      wc[r][cc] = Cc;
    }
    __syncthreads();
    if (x < W && y < H) {
      float sa = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          // window anchored at (y-i, x-j) = tile slot (ly+2-i, lx+2-j)
          sa += wa[ly + 2 - i][lx + 2 - j];
          sb += wb[ly + 2 - i][lx + 2 - j];
          sc += wc[ly + 2 - i][lx + 2 - j];
        }
      const long o = (long)y * W + x;
      const float m = mask ? mask[(long)b * cs + o] : 1.f;
      const float xv = dx_[ly + 2][lx + 2], yv = dy_[ly + 2][lx + 2];
      const float diff = rc[o] - imc[o];
      const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
      g_rec[((long)b * C + c) * cs + o] = m * (c_l1 * sg + sa + sb * xv + sc * yv);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// SSIM + L1, 4 pixels per lane (rows 16-byte aligned: W % 4 == 0).  The 1-px kernels above spend their time
// on scalar LDS reads (18 per window, 27 more per pixel in the backward): here a 16 x 64 pixel tile is
// staged with float4 loads (all in flight, none branched around), a lane owns 4 consecutive pixels and
// reads each window row as ds_read_b128 + ds_read_b64 (6 values serve its 4 windows).  Same arithmetic, in
// the same order, as the 1-px kernels (which remain for unaligned widths).
// ------------------------------------------------------------------------------------------------
namespace photo4 {
constexpr int TXW = 64, TYH = 16, NT = 256, P = 128;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void read6(const float* row, float (&v)[6]) {
  f32x4 t = *reinterpret_cast<const f32x4*>(row);
  f32x2 u = *reinterpret_cast<const f32x2*>(row + 4);
  asm volatile("" : "+v"(t), "+v"(u));
  v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w, v[4] = u.x, v[5] = u.y;
}
__device__ __forceinline__ void read8(const float* row, float (&v)[8]) {
  f32x4 t = *reinterpret_cast<const f32x4*>(row);
  f32x4 u = *reinterpret_cast<const f32x4*>(row + 4);
  asm volatile("" : "+v"(t), "+v"(u));
  v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w, v[4] = u.x, v[5] = u.y, v[6] = u.z, v[7] = u.w;
}
// statistics of the 3x3 window whose left column is `e` of the 6-wide strips (same order as window_stats)
__device__ __forceinline__ Win stats6(const float (&a)[3][6], const float (&b)[3][6], int e) {
  float sxv = 0.f, syv = 0.f, sxx = 0.f, syy = 0.f, sxyv = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float x = a[i][e + j], y = b[i][e + j];
      sxv += x;
      syv += y;
      sxx += x * x;
      syy += y * y;
      sxyv += x * y;
    }
  Win w;
  w.mx = div9(sxv);
  w.my = div9(syv);
  w.sx = div9(sxx) - w.mx * w.mx;
  w.sy = div9(syy) - w.my * w.my;
  w.sxy = div9(sxyv) - w.mx * w.my;
  return w;
}

// masked tiles x = recons*mask, y = im*mask: `rows` x `nq` float4 starting at image (gy0, gx0) (gx0 % 4 == 0)
template <int ROWS, int NQ, bool L1>
__device__ __forceinline__ float stage(float* __restrict__ X, float* __restrict__ Y, const float* __restrict__ imc,
                                       const float* __restrict__ rc, const float* __restrict__ mb, int H, int W,
                                       int gy0, int gx0, int own_r0, int own_q0) {
  constexpr int NS = ROWS * NQ, ITER = (NS + NT - 1) / NT;
  float4 iv[ITER], rv[ITER], mv[ITER];
  int r[ITER], q[ITER];
  bool ok[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int s = threadIdx.x + it * NT;
    r[it] = s / NQ, q[it] = s - r[it] * NQ;
    const int gy = gy0 + r[it], gx = gx0 + 4 * q[it];
    ok[it] = s < NS && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const long o = ok[it] ? (long)gy * W + gx : 0;
    iv[it] = *reinterpret_cast<const float4*>(imc + o);
    rv[it] = *reinterpret_cast<const float4*>(rc + o);
    mv[it] = mb ? *reinterpret_cast<const float4*>(mb + o) : make_float4(1.f, 1.f, 1.f, 1.f);
  }
  float l1 = 0.f;
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    if (threadIdx.x + it * NT < NS) {
      const float4 i4 = iv[it], r4 = rv[it], m4 = mv[it];
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(X + r[it] * P + 4 * q[it]) =
          ok[it] ? make_float4(r4.x * m4.x, r4.y * m4.y, r4.z * m4.z, r4.w * m4.w) : z;
      *reinterpret_cast<float4*>(Y + r[it] * P + 4 * q[it]) =
          ok[it] ? make_float4(i4.x * m4.x, i4.y * m4.y, i4.z * m4.z, i4.w * m4.w) : z;
      if (L1 && ok[it] && r[it] >= own_r0 && r[it] < own_r0 + TYH && q[it] >= own_q0 && q[it] < own_q0 + TXW / 4)
        l1 += ((fabsf(i4.x - r4.x) * m4.x + fabsf(i4.y - r4.y) * m4.y) + fabsf(i4.z - r4.z) * m4.z) +
              fabsf(i4.w - r4.w) * m4.w;
    }
  }
  return l1;
}

__global__ __launch_bounds__(NT) void fwd_kernel(const float* __restrict__ im, const float* __restrict__ rec,
                                                 const float* __restrict__ mask, float* __restrict__ ssim_map,
                                                 float* __restrict__ sums, int nimg, int C, int H, int W) {
  __shared__ __attribute__((aligned(16))) float X[(TYH + 2) * P];
  __shared__ __attribute__((aligned(16))) float Y[(TYH + 2) * P];
  __shared__ float red[3 * (NT / 64)];
  int btx, bty, b;
  if (!af_tile_of_block((W + TXW - 1) / TXW, (H + TYH - 1) / TYH, nimg, btx, bty, b)) return;
  const int ty0 = bty * TYH, tx0 = btx * TXW;
  const long cs = (long)H * W;
  const int xg = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int x0 = tx0 + 4 * xg, y = ty0 + ly;
  const float* mb = mask ? mask + (long)b * cs : nullptr;
  float part[3] = {0.f, 0.f, 0.f};
  if (y < H && x0 < W) {
    if (mb) {
      const float4 m = *reinterpret_cast<const float4*>(mb + (long)y * W + x0);
      part[2] = (m.x + m.y) + (m.z + m.w);
    } else {
      part[2] = 4.f;
    }
  }
  for (int c = 0; c < C; ++c) {
    if (c) __syncthreads();
    part[0] += stage<TYH + 2, TXW / 4 + 1, true>(X, Y, im + ((long)b * C + c) * cs, rec + ((long)b * C + c) * cs, mb, H,
                                                  W, ty0, tx0, 0, 0);
    __syncthreads();
    if (y < H - 2) {
      float a[3][6], bb[3][6];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        read6(X + (ly + i) * P + 4 * xg, a[i]);
        read6(Y + (ly + i) * P + 4 * xg, bb[i]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (x0 + e < W - 2) {
          const Win w = stats6(a, bb, e);
          const float n = (2.f * w.mx * w.my + SSIM_C1) * (2.f * w.sxy + SSIM_C2);
          const float d = (w.mx * w.mx + w.my * w.my + SSIM_C1) * (w.sx + w.sy + SSIM_C2);
          const float dist = fminf(fmaxf((1.f - fdiv_pos(n, d)) / 2.f, 0.f), 1.f);
          part[1] += dist;
          if (ssim_map) ssim_map[(((long)b * C + c) * (H - 2) + y) * (W - 2) + x0 + e] = dist;
        }
      }
    }
  }
  af_block_sum<3>(part, red);
  if (threadIdx.x == 0) {
    float* slot = af_sum_slot(sums);
    atomicAdd(slot, part[0]);
    atomicAdd(slot + 1, part[1]);
    atomicAdd(slot + 2, part[2]);
  }
}

__global__ __launch_bounds__(NT) void bwd_kernel(const float* __restrict__ im, const float* __restrict__ rec,
                                                 const float* __restrict__ mask, const float* __restrict__ gmap,
                                                 const float* __restrict__ coef, float* __restrict__ g_rec, int nimg,
                                                 int C, int H, int W) {
  // tile coordinates: row r <-> image row ty0 - 2 + r (20 rows), column q <-> image column tx0 - 4 + q (72);
  // window anchors live at rows 0..17, columns 2..65 of the same coordinates
  __shared__ __attribute__((aligned(16))) float X[(TYH + 4) * P];
  __shared__ __attribute__((aligned(16))) float Y[(TYH + 4) * P];
  __shared__ __attribute__((aligned(16))) float WA[(TYH + 2) * P];
  __shared__ __attribute__((aligned(16))) float WB[(TYH + 2) * P];
  __shared__ __attribute__((aligned(16))) float WC[(TYH + 2) * P];
  int btx, bty, b;
  if (!af_tile_of_block((W + TXW - 1) / TXW, (H + TYH - 1) / TYH, nimg, btx, bty, b)) return;
  const int ty0 = bty * TYH, tx0 = btx * TXW;
  const long cs = (long)H * W;
  const int xg = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int x0 = tx0 + 4 * xg, y = ty0 + ly;
  const float* mb = mask ? mask + (long)b * cs : nullptr;
  const float c_l1 = coef[0], c_ss = coef[1];
  constexpr int NG = TXW / 4 + 1, NTASK = (TYH + 2) * NG;  // 18 anchor rows x 17 groups of 4 anchors
  for (int c = 0; c < C; ++c) {
    const float* imc = im + ((long)b * C + c) * cs;
    const float* rc = rec + ((long)b * C + c) * cs;
    if (c) __syncthreads();
    stage<TYH + 4, TXW / 4 + 2, false>(X, Y, imc, rc, mb, H, W, ty0 - 2, tx0 - 4, 0, 0);
    __syncthreads();
    // per-window coefficients: d dist_w / d x_r = -(1/2)(A + B x_r + C y_r)  (see photo_bwd_kernel)
    for (int t = threadIdx.x; t < NTASK; t += NT) {
      const int r = t / NG, g = t - r * NG;
      float a[3][6], bb[3][6];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        read6(X + (r + i) * P + 4 * g, a[i]);
        read6(Y + (r + i) * P + 4 * g, bb[i]);
      }
      const int wy = ty0 - 2 + r;
      float A[4], Bc[4], Cc[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int wx = tx0 - 4 + 4 * g + e;
        A[e] = Bc[e] = Cc[e] = 0.f;
        if (wy >= 0 && wy < H - 2 && wx >= 0 && wx < W - 2) {
          const Win w = stats6(a, bb, e);
          const float n1 = 2.f * w.mx * w.my + SSIM_C1, n2 = 2.f * w.sxy + SSIM_C2;
          const float d1 = w.mx * w.mx + w.my * w.my + SSIM_C1, d2 = w.sx + w.sy + SSIM_C2;
          const float n = n1 * n2, d = d1 * d2;
          const float v = (1.f - fdiv_pos(n, d)) / 2.f;
          if (v >= 0.f && v <= 1.f) {  // torch.clamp passes the gradient on the closed interval
            const float up = gmap ? gmap[(((long)b * C + c) * (H - 2) + wy) * (W - 2) + wx] : c_ss;
            const float k = -0.5f * up * (2.f / 9.f);
            const float id = frcp_pos(d), nd2 = n * id * id;
            Cc[e] = k * n1 * id;                                                        // * y_r
            Bc[e] = -k * nd2 * d1;                                                      // * x_r
            A[e] = k * ((w.my * n2 - n1 * w.my) * id - nd2 * (w.mx * d2 - d1 * w.mx));  // constant
          }
        }
      }
      *reinterpret_cast<float4*>(WA + r * P + 4 * g) = make_float4(A[0], A[1], A[2], A[3]);
      *reinterpret_cast<float4*>(WB + r * P + 4 * g) = make_float4(Bc[0], Bc[1], Bc[2], Bc[3]);
      *reinterpret_cast<float4*>(WC + r * P + 4 * g) = make_float4(Cc[0], Cc[1], Cc[2], Cc[3]);
    }
    __syncthreads();
    if (y < H && x0 < W) {
      // pixel (y, x0+e) = tile (ly+2, 4xg+4+e); the window anchored at (y-i, x-j) sits at tile (ly+2-i, 4xg+4+e-j)
      float ca[3][8], cb[3][8], cc[3][8], xc[8], yc[8];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        read8(WA + (ly + i) * P + 4 * xg, ca[i]);
        read8(WB + (ly + i) * P + 4 * xg, cb[i]);
        read8(WC + (ly + i) * P + 4 * xg, cc[i]);
      }
      read8(X + (ly + 2) * P + 4 * xg, xc);
      read8(Y + (ly + 2) * P + 4 * xg, yc);
      const long o = (long)y * W + x0;
      const float4 i4 = *reinterpret_cast<const float4*>(imc + o), r4 = *reinterpret_cast<const float4*>(rc + o);
      const float4 m4 = mb ? *reinterpret_cast<const float4*>(mb + o) : make_float4(1.f, 1.f, 1.f, 1.f);
      const float iv[4] = {i4.x, i4.y, i4.z, i4.w}, rv[4] = {r4.x, r4.y, r4.z, r4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w};
      float out[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float sa = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            sa += ca[2 - i][4 + e - j];
            sb += cb[2 - i][4 + e - j];
            sc += cc[2 - i][4 + e - j];
          }
        const float diff = rv[e] - iv[e];
        const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        out[e] = mv[e] * (c_l1 * sg + sa + sb * xc[4 + e] + sc * yc[4 + e]);
      }
      *reinterpret_cast<float4*>(g_rec + ((long)b * C + c) * cs + o) = make_float4(out[0], out[1], out[2], out[3]);
    }
  }
}
}  // namespace photo4

}  // namespace

extern "C" int arflow_census_fwd(const float* im_a, const float* im_b, const float* mask, float* ham,
                                 float* dham, float* sums, int B, int H, int W, int radius,
                                 arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(im_a);
  AF_REQUIRE_PTR(im_b);
  AF_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535 && H <= 8 * 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(radius >= 1 && radius <= 3, ARFLOW_EPARAM);
  if (mask) AF_REQUIRE_PTR(sums);
  hipStream_t st = (hipStream_t)stream;
  if (mask) {
    hipError_t e = hipMemsetAsync(sums, 0, AF_SUMS_BYTES, st);
    if (e != hipSuccess) return af_hip_status(e);
  }
  if ((W & 3) == 0) {
    namespace c4 = census4;
    dim3 g4(af_grid_for_tiles((long)af_cdiv(W, c4::TXW) * af_cdiv(H, c4::TYH) * B));
    switch (radius) {
      case 1: hipLaunchKernelGGL(c4::fwd_kernel<1>, g4, dim3(c4::NT), 0, st, im_a, im_b, mask, ham, dham, sums, B, H, W); break;
      case 2: hipLaunchKernelGGL(c4::fwd_kernel<2>, g4, dim3(c4::NT), 0, st, im_a, im_b, mask, ham, dham, sums, B, H, W); break;
      default: hipLaunchKernelGGL(c4::fwd_kernel<3>, g4, dim3(c4::NT), 0, st, im_a, im_b, mask, ham, dham, sums, B, H, W); break;
    }
    return af_launch_status();
  }
  dim3 grid(af_grid_for_tiles((long)af_cdiv(W, TX) * af_cdiv(H, TY) * B));
  switch (radius) {
    case 1: hipLaunchKernelGGL(census_fwd_kernel<1>, grid, dim3(TX * TY), 0, st, im_a, im_b, mask, ham, dham, sums, B, H, W); break;
    case 2: hipLaunchKernelGGL(census_fwd_kernel<2>, grid, dim3(TX * TY), 0, st, im_a, im_b, mask, ham, dham, sums, B, H, W); break;
    default: hipLaunchKernelGGL(census_fwd_kernel<3>, grid, dim3(TX * TY), 0, st, im_a, im_b, mask, ham, dham, sums, B, H, W); break;
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
  AF_REQUIRE(radius >= 1 && radius <= 3, ARFLOW_EPARAM);
  hipStream_t st = (hipStream_t)stream;
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

extern "C" int arflow_photo_fwd(const float* im, const float* recons, const float* mask, float* ssim_map,
                                float* sums, int B, int C, int H, int W, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(im);
  AF_REQUIRE_PTR(recons);
  AF_REQUIRE_PTR(sums);
  AF_REQUIRE(B > 0 && C > 0 && H >= 3 && W >= 3 && B <= 65535, ARFLOW_ESHAPE);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(sums, 0, AF_SUMS_BYTES, st);
  if (e != hipSuccess) return af_hip_status(e);
  if ((W & 3) == 0) {
    namespace p4 = photo4;
    dim3 g4(af_grid_for_tiles((long)af_cdiv(W, p4::TXW) * af_cdiv(H, p4::TYH) * B));
    hipLaunchKernelGGL(p4::fwd_kernel, g4, dim3(p4::NT), 0, st, im, recons, mask, ssim_map, sums, B, C, H, W);
    return af_launch_status();
  }
  dim3 grid(af_grid_for_tiles((long)af_cdiv(W, TX) * af_cdiv(H, TY) * B));
  hipLaunchKernelGGL(photo_fwd_kernel, grid, dim3(TX * TY), 0, st, im, recons, mask, ssim_map, sums, B, C, H, W);
  return af_launch_status();
}

extern "C" int arflow_photo_bwd(const float* im, const float* recons, const float* mask, const float* gmap,
                                const float* coef, float* g_recons, int B, int C, int H, int W,
                                arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(im);
  AF_REQUIRE_PTR(recons);
  AF_REQUIRE_PTR(coef);
  AF_REQUIRE_PTR(g_recons);
  AF_REQUIRE(B > 0 && C > 0 && H >= 3 && W >= 3 && B <= 65535, ARFLOW_ESHAPE);
  if ((W & 3) == 0) {
    namespace p4 = photo4;
    dim3 g4(af_grid_for_tiles((long)af_cdiv(W, p4::TXW) * af_cdiv(H, p4::TYH) * B));
    hipLaunchKernelGGL(p4::bwd_kernel, g4, dim3(p4::NT), 0, (hipStream_t)stream, im, recons, mask, gmap, coef, g_recons, B,
                       C, H, W);
    return af_launch_status();
  }
  dim3 grid(af_grid_for_tiles((long)af_cdiv(W, TX) * af_cdiv(H, TY) * B));
  hipLaunchKernelGGL(photo_bwd_kernel, grid, dim3(TX * TY), 0, (hipStream_t)stream, im, recons, mask, gmap,
                     coef, g_recons, B, C, H, W);
  return af_launch_status();
}
