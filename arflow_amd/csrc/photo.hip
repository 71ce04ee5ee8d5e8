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
// SSIM (3x3, un-padded) + L1
// ------------------------------------------------------------------------------------------------
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;

struct Win {
  float mx, my, sx, sy, sxy;
};

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
  w.mx = sxv / 9.f;
  w.my = syv / 9.f;
  w.sx = sxx / 9.f - w.mx * w.mx;
  w.sy = syy / 9.f - w.my * w.my;
  w.sxy = sxyv / 9.f - w.mx * w.my;
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
      const float dist = fminf(fmaxf((1.f - n / d) / 2.f, 0.f), 1.f);
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
        const float v = (1.f - n / d) / 2.f;
        if (v >= 0.f && v <= 1.f) {  // torch.clamp passes the gradient on the closed interval
          const float up = gmap ? gmap[(((long)b * C + c) * (H - 2) + wy) * (W - 2) + wx] : c_ss;
          const float k = -0.5f * up * (2.f / 9.f);
          const float id = 1.f / d, nd2 = n * id * id;
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

}  // namespace

extern "C" int arflow_census_fwd(const float* im_a, const float* im_b, const float* mask, float* ham,
                                 float* dham, float* sums, int B, int H, int W, int radius,
                                 arflow_stream_t stream) {
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
  AF_REQUIRE_PTR(im_a);
  AF_REQUIRE_PTR(im_b);
  AF_REQUIRE_PTR(gham);
  AF_REQUIRE_PTR(g_im_b);
  AF_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535 && H <= 8 * 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(radius >= 1 && radius <= 3, ARFLOW_EPARAM);
  hipStream_t st = (hipStream_t)stream;
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
  AF_REQUIRE_PTR(im);
  AF_REQUIRE_PTR(recons);
  AF_REQUIRE_PTR(sums);
  AF_REQUIRE(B > 0 && C > 0 && H >= 3 && W >= 3 && B <= 65535, ARFLOW_ESHAPE);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(sums, 0, AF_SUMS_BYTES, st);
  if (e != hipSuccess) return af_hip_status(e);
  dim3 grid(af_grid_for_tiles((long)af_cdiv(W, TX) * af_cdiv(H, TY) * B));
  hipLaunchKernelGGL(photo_fwd_kernel, grid, dim3(TX * TY), 0, st, im, recons, mask, ssim_map, sums, B, C, H, W);
  return af_launch_status();
}

extern "C" int arflow_photo_bwd(const float* im, const float* recons, const float* mask, const float* gmap,
                                const float* coef, float* g_recons, int B, int C, int H, int W,
                                arflow_stream_t stream) {
  AF_REQUIRE_PTR(im);
  AF_REQUIRE_PTR(recons);
  AF_REQUIRE_PTR(coef);
  AF_REQUIRE_PTR(g_recons);
  AF_REQUIRE(B > 0 && C > 0 && H >= 3 && W >= 3 && B <= 65535, ARFLOW_ESHAPE);
  dim3 grid(af_grid_for_tiles((long)af_cdiv(W, TX) * af_cdiv(H, TY) * B));
  hipLaunchKernelGGL(photo_bwd_kernel, grid, dim3(TX * TY), 0, (hipStream_t)stream, im, recons, mask, gmap,
                     coef, g_recons, B, C, H, W);
  return af_launch_status();
}
