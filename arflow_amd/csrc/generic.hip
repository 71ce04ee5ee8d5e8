// The REST of the reference's parameter space for the hot-path functions -- values no shipped config uses but the
// functions accept -- as plain one-thread-per-element gfx950 kernels (correct first; nothing here is on a measured
// path, the tuned kernels of the other files serve every configuration the reference ships):
//   * flow_warp(mode='nearest' | 'bicubic')                      utils/warp_utils.py:83-90 -> grid_sample nearest / bicubic
//   * TernaryLoss(max_distance > 3) / census_loss(patch_size > 7) losses/loss_blocks.py:12-62, utils/uflow_utils.py:241-293
//   * SSIM(md != 1)                                              losses/loss_blocks.py:65-84
//   * Correlation(kernel_size, stride1, stride2, pad_size != d)  models/correlation_package/correlation_cuda_kernel.cu:41-114
#include "common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------ nearest warp
// torch grid_sample(mode='nearest') (ATen/native/GridSampler.cpp, nearest branch): border padding clips the
// un-normalised coordinate, the index is nearbyint (round half to even), out-of-range indices read zero.
__device__ __forceinline__ bool nearest_index(float px, float py, float u, float v, int H, int W, int Hs, int Ws, int pad,
                                              bool align, int norm, int& xi, int& yi) {
  float dummy;
  float ix = af_sample_coord(px, u, W, Ws, norm, align, &dummy);
  float iy = af_sample_coord(py, v, H, Hs, norm, align, &dummy);
  if (pad == ARFLOW_PAD_BORDER) {
    ix = fminf(fmaxf(ix, 0.f), (float)(Ws - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(Hs - 1));
  }
  const float rx = nearbyintf(ix), ry = nearbyintf(iy);
  const bool ok = rx >= 0.f && rx <= (float)(Ws - 1) && ry >= 0.f && ry <= (float)(Hs - 1);  // NaN -> false
  xi = ok ? (int)rx : 0;
  yi = ok ? (int)ry : 0;
  return ok;
}

__global__ __launch_bounds__(256) void warp_nearest_fwd_kernel(const float* __restrict__ src, const float* __restrict__ flow,
                                                               float* __restrict__ out, int C, int Hs, int Ws, int H,
                                                               int W, long fbs, int pad, int align, int norm) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= W) return;
  const float* fb = flow + (long)b * fbs + (long)y * W + x;
  int xi, yi;
  const bool ok = nearest_index((float)x, (float)y, fb[0], fb[(long)H * W], H, W, Hs, Ws, pad, align != 0, norm, xi, yi);
  const long ss = (long)Hs * Ws, os = (long)H * W;
  const float* sp = src + (long)b * C * ss + (long)yi * Ws + xi;
  float* op = out + (long)b * C * os + (long)y * W + x;
  for (int c = 0; c < C; ++c) op[c * os] = ok ? sp[c * ss] : 0.f;
}

__global__ __launch_bounds__(256) void warp_nearest_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ flow,
                                                               float* __restrict__ gsrc, int C, int Hs, int Ws, int H,
                                                               int W, long fbs, int pad, int align, int norm) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= W) return;
  const float* fb = flow + (long)b * fbs + (long)y * W + x;
  int xi, yi;
  if (!nearest_index((float)x, (float)y, fb[0], fb[(long)H * W], H, W, Hs, Ws, pad, align != 0, norm, xi, yi)) return;
  const long ss = (long)Hs * Ws, os = (long)H * W;
  float* gp = gsrc + (long)b * C * ss + (long)yi * Ws + xi;
  const float* gop = gout + (long)b * C * os + (long)y * W + x;
  for (int c = 0; c < C; ++c) atomicAdd(gp + c * ss, gop[c * os]);
}

// ------------------------------------------------------------------------------------------------ bicubic warp
// torch grid_sample(mode='bicubic') (ATen/native/GridSampler.h get_cubic_upsample_coefficients / get_value_bounded,
// ATen/native/cuda/GridSampler.cu bicubic branch): A = -0.75; the 4 x 4 taps around floor(coordinate) are read at
// BOUNDED positions (border padding clips each tap's position, zeros padding reads 0 outside); rows are interpolated
// first, then the column; the coordinate gradient uses the derivative of the coefficients and ignores the tap clipping.
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
  const float A = -0.75f;
  const float x1 = t, x2 = 1.0f - t;
  c[0] = ((A * (x1 + 1.0f) - 5.0f * A) * (x1 + 1.0f) + 8.0f * A) * (x1 + 1.0f) - 4.0f * A;
  c[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
  c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  c[3] = ((A * (x2 + 1.0f) - 5.0f * A) * (x2 + 1.0f) + 8.0f * A) * (x2 + 1.0f) - 4.0f * A;
}
__device__ __forceinline__ void cubic_coeffs_grad(float t, float (&g)[4]) {
  const float A = -0.75f;
  float x = -1.0f - t;
  g[0] = (-3.0f * A * x - 10.0f * A) * x - 8.0f * A;
  x = -t;
  g[1] = (-3.0f * (A + 2.0f) * x - 2.0f * (A + 3.0f)) * x;
  x = 1.0f - t;
  g[2] = (3.0f * (A + 2.0f) * x - 2.0f * (A + 3.0f)) * x;
  x = 2.0f - t;
  g[3] = (3.0f * A * x - 10.0f * A) * x + 8.0f * A;
}
struct BicubicTaps {
  int off[4][4];  // element offset of tap (row j, column i) inside a source plane, or -1 when it reads zero
  float cx[4], cy[4], tx, ty, dx, dy;
};
__device__ __forceinline__ BicubicTaps bicubic_taps(float px, float py, float u, float v, int H, int W, int Hs, int Ws, int pad,
                                                    bool align, int norm) {
  BicubicTaps t;
  const float ix = af_sample_coord(px, u, W, Ws, norm, align, &t.dx);
  const float iy = af_sample_coord(py, v, H, Hs, norm, align, &t.dy);
  const float fx = floorf(ix), fy = floorf(iy);
  t.tx = ix - fx, t.ty = iy - fy;
  cubic_coeffs(t.tx, t.cx);
  cubic_coeffs(t.ty, t.cy);
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float x = fx - 1.0f + (float)i, y = fy - 1.0f + (float)j;  // get_value_bounded: compute_coordinates, then the cast
      if (pad == ARFLOW_PAD_BORDER) {
        x = fminf(fmaxf(x, 0.f), (float)(Ws - 1));
        y = fminf(fmaxf(y, 0.f), (float)(Hs - 1));
      }
      const bool ok = x >= 0.f && x <= (float)(Ws - 1) && y >= 0.f && y <= (float)(Hs - 1);  // NaN -> false
      t.off[j][i] = ok ? (int)y * Ws + (int)x : -1;
    }
  return t;
}

__global__ __launch_bounds__(256) void warp_bicubic_fwd_kernel(const float* __restrict__ src, const float* __restrict__ flow,
                                                               float* __restrict__ out, int C, int Hs, int Ws, int H, int W,
                                                               long fbs, int pad, int align, int norm) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= W) return;
  const float* fb = flow + (long)b * fbs + (long)y * W + x;
  const BicubicTaps t = bicubic_taps((float)x, (float)y, fb[0], fb[(long)H * W], H, W, Hs, Ws, pad, align != 0, norm);
  const long ss = (long)Hs * Ws, os = (long)H * W;
  const float* sp = src + (long)b * C * ss;
  float* op = out + (long)b * C * os + (long)y * W + x;
  for (int c = 0; c < C; ++c) {
    float rows[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = t.off[j][i] >= 0 ? sp[c * ss + t.off[j][i]] : 0.f;
      rows[j] = ((v[0] * t.cx[0] + v[1] * t.cx[1]) + v[2] * t.cx[2]) + v[3] * t.cx[3];  // cubic_interp1d
    }
    op[c * os] = ((rows[0] * t.cy[0] + rows[1] * t.cy[1]) + rows[2] * t.cy[2]) + rows[3] * t.cy[3];
  }
}

// gsrc (nullable) arrives zero-filled; gflow (nullable) [B,2,H,W] is written
__global__ __launch_bounds__(256) void warp_bicubic_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ src,
                                                               const float* __restrict__ flow, float* __restrict__ gsrc,
                                                               float* __restrict__ gflow, int C, int Hs, int Ws, int H, int W,
                                                               long fbs, int pad, int align, int norm) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= W) return;
  const float* fb = flow + (long)b * fbs + (long)y * W + x;
  const BicubicTaps t = bicubic_taps((float)x, (float)y, fb[0], fb[(long)H * W], H, W, Hs, Ws, pad, align != 0, norm);
  float gxc[4], gyc[4];
  cubic_coeffs_grad(t.tx, gxc);
  cubic_coeffs_grad(t.ty, gyc);
  const long ss = (long)Hs * Ws, os = (long)H * W;
  const float* sp = src + (long)b * C * ss;
  const float* gop = gout + (long)b * C * os + (long)y * W + x;
  float gix = 0.f, giy = 0.f;
  for (int c = 0; c < C; ++c) {
    const float g = gop[c * os];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (t.off[j][i] < 0) continue;
        if (gsrc) atomicAdd(gsrc + (long)b * C * ss + c * ss + t.off[j][i], g * t.cx[i] * t.cy[j]);
        if (gflow) {
          const float v = sp[c * ss + t.off[j][i]];
          gix -= v * gxc[i] * t.cy[j] * g;
          giy -= v * gyc[j] * t.cx[i] * g;
        }
      }
  }
  if (gflow) {
    float* gf = gflow + (long)b * 2 * os + (long)y * W + x;
    gf[0] = t.dx * gix;
    gf[os] = t.dy * giy;
  }
}

// ------------------------------------------------------------------------------------------------ census, any radius
__device__ __forceinline__ float gray255_at(const float* __restrict__ im, int H, int W, int y, int x) {
  if (y < 0 || y >= H || x < 0 || x >= W) return 0.f;  // the census transform zero-pads
  const long cs = (long)H * W, o = (long)y * W + x;
  return ((im[o] * 0.2989f + im[o + cs] * 0.5870f) + im[o + 2 * cs] * 0.1140f) * 255.f;
}

// same outputs as census4::fwd_kernel (photo.hip): ham and / or the fused census_loss pieces
__global__ __launch_bounds__(256) void census_any_fwd_kernel(const float* __restrict__ im_a, const float* __restrict__ im_b,
                                                             const float* __restrict__ mask, float* __restrict__ ham_out,
                                                             float* __restrict__ dham_out, float* __restrict__ sums, int nrows,
                                                             int H, int W, int R) {
  __shared__ float red[2 * 4];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  float part[2] = {0.f, 0.f};
  if (x < W) {
    const float* pa = im_a + (long)b * 3 * H * W;
    const float* pb = im_b + (long)b * 3 * H * W;
    const float ca = gray255_at(pa, H, W, y, x), cb = gray255_at(pb, H, W, y, x);
    float s = 0.f;
    for (int dy = -R; dy <= R; ++dy)
      for (int dx = -R; dx <= R; ++dx) {
        const float da = gray255_at(pa, H, W, y + dy, x + dx) - ca, db = gray255_at(pb, H, W, y + dy, x + dx) - cb;
        const float tb = db * __builtin_amdgcn_rsqf(fmaf(db, db, 0.81f));
        const float e = fmaf(da, __builtin_amdgcn_rsqf(fmaf(da, da, 0.81f)), -tb), sq = e * e;
        s = fmaf(sq, __builtin_amdgcn_rcpf(0.1f + sq), s);
      }
    const long o = ((long)b * H + y) * W + x;
    if (ham_out) ham_out[o] = s;
    if (mask) {
      const float pm = (y >= R && y < H - R && x >= R && x < W - R) ? mask[o] : 0.f;
      const float lg = __log2f(fabsf(s) + 0.01f);
      part[0] = exp2f(0.4f * lg) * pm;
      part[1] = pm;
      if (dham_out) dham_out[o] = pm * 0.4f * exp2f(-0.6f * lg);
    }
  }
  if (mask) {
    af_block_sum<2>(part, red);
    if (threadIdx.x == 0) af_store_partial(sums, nrows, part[0], part[1], 0.f);
  }
}

// d / d im_b, the arithmetic of census4::bwd_kernel: centre and neighbour roles folded into one term per offset
__global__ __launch_bounds__(256) void census_any_bwd_kernel(const float* __restrict__ im_a, const float* __restrict__ im_b,
                                                             const float* __restrict__ gham, const float* __restrict__ scale,
                                                             float* __restrict__ g_im_b, int H, int W, int R) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= W) return;
  const float* pa = im_a + (long)b * 3 * H * W;
  const float* pb = im_b + (long)b * 3 * H * W;
  const float* gg = gham + (long)b * H * W;
  const float ca = gray255_at(pa, H, W, y, x), cb = gray255_at(pb, H, W, y, x), cg = gg[(long)y * W + x];
  float acc = 0.f;
  for (int dy = -R; dy <= R; ++dy)
    for (int dx = -R; dx <= R; ++dx) {
      if (dy == 0 && dx == 0) continue;
      const int yy = y - dy, xx = x - dx;  // the pixel at r - kk
      const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
      const float wg = in ? gg[(long)yy * W + xx] : 0.f;
      const float da = ca - gray255_at(pa, H, W, yy, xx), db = cb - gray255_at(pb, H, W, yy, xx);
      const float ua = __builtin_amdgcn_rsqf(fmaf(da, da, 0.81f));
      const float ub = __builtin_amdgcn_rsqf(fmaf(db, db, 0.81f));
      const float e = fmaf(da, ua, -(db * ub));
      const float q = __builtin_amdgcn_rcpf(fmaf(e, e, 0.1f));
      const float hd = ((q * e) * q) * ((ub * ub) * ub);
      acc = fmaf(wg + cg, hd, acc);
    }
  const float sc = (scale ? scale[0] : 1.f) * 255.f * (0.1f * -2.f * 0.81f);
  float* o = g_im_b + (long)b * 3 * H * W + (long)y * W + x;
  const long cs = (long)H * W;
  o[0] = sc * acc * 0.2989f;
  o[cs] = sc * acc * 0.5870f;
  o[2 * cs] = sc * acc * 0.1140f;
}

// ------------------------------------------------------------------------------------------------ SSIM, any window
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;
struct WinStats {
  float mx, my, sx, sy, sxy;
};
// un-padded k x k window anchored at (wy, wx): the reference pools x, y and the rounded products x*x, y*y, x*y
// (losses/loss_blocks.py:70-78), sum in row-major order, then divides by k^2
__device__ __forceinline__ WinStats window_stats(const float* __restrict__ X, const float* __restrict__ Y, int W, int wy,
                                                 int wx, int k) {
  float sa = 0.f, sb = 0.f, saa = 0.f, sbb = 0.f, sab = 0.f;
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) {
      const float a = X[(long)(wy + i) * W + wx + j], b = Y[(long)(wy + i) * W + wx + j];
      sa += a, sb += b, saa += a * a, sbb += b * b, sab += a * b;
    }
  const float n = (float)(k * k);
  WinStats w;
  w.mx = sa / n, w.my = sb / n;
  w.sx = saa / n - w.mx * w.mx, w.sy = sbb / n - w.my * w.my, w.sxy = sab / n - w.mx * w.my;
  return w;
}

__global__ __launch_bounds__(256) void ssim_any_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           float* __restrict__ out, int H, int W, int md) {
  const int k = 2 * md + 1, Ho = H - 2 * md, Wo = W - 2 * md;
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
  const long plane = blockIdx.z;  // b * C + c
  if (ox >= Wo) return;
  const WinStats w = window_stats(x + plane * H * W, y + plane * H * W, W, oy, ox, k);
  const float n = (2.f * w.mx * w.my + SSIM_C1) * (2.f * w.sxy + SSIM_C2);
  const float d = (w.mx * w.mx + w.my * w.my + SSIM_C1) * (w.sx + w.sy + SSIM_C2);
  out[(plane * Ho + oy) * Wo + ox] = fminf(fmaxf((1.f - n / d) / 2.f, 0.f), 1.f);
}

// d / d x of sum_w gmap[w] * dist[w]; with S = A1 A2 / (B1 B2) over a window of N pixels,
//   dS/dx_p = [2 my A2 + 2 A1 (y_p - my)] / (N B1 B2) - S [2 mx / (N B1) + 2 (x_p - mx) / (N B2)],  d dist = -dS/2
// inside 0 < (1-S)/2 < 1, else 0.
__global__ __launch_bounds__(256) void ssim_any_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ gmap, float* __restrict__ gx, int H,
                                                           int W, int md) {
  const int k = 2 * md + 1, Ho = H - 2 * md, Wo = W - 2 * md;
  const int px = blockIdx.x * blockDim.x + threadIdx.x, py = blockIdx.y;
  const long plane = blockIdx.z;
  if (px >= W) return;
  const float* X = x + plane * H * W;
  const float* Y = y + plane * H * W;
  const float xp = X[(long)py * W + px], yp = Y[(long)py * W + px];
  const float N = (float)(k * k);
  float acc = 0.f;
  for (int wy = max(0, py - k + 1); wy <= min(py, Ho - 1); ++wy)
    for (int wx = max(0, px - k + 1); wx <= min(px, Wo - 1); ++wx) {
      const float g = gmap[(plane * Ho + wy) * Wo + wx];
      const WinStats w = window_stats(X, Y, W, wy, wx, k);
      const float A1 = 2.f * w.mx * w.my + SSIM_C1, A2 = 2.f * w.sxy + SSIM_C2;
      const float B1 = w.mx * w.mx + w.my * w.my + SSIM_C1, B2 = w.sx + w.sy + SSIM_C2;
      const float S = (A1 * A2) / (B1 * B2);
      const float dist = (1.f - S) / 2.f;
      if (!(dist > 0.f && dist < 1.f)) continue;  // clamp inactive only strictly inside (torch.clamp's gradient)
      const float dS = (2.f * w.my * A2 + 2.f * A1 * (yp - w.my)) / (N * B1 * B2) -
                       S * (2.f * w.mx / (N * B1) + 2.f * (xp - w.mx) / (N * B2));
      acc = fmaf(g, -0.5f * dS, acc);
    }
  gx[plane * H * W + (long)py * W + px] = acc;
}

// ------------------------------------------------------------------------------------------------ general correlation
// The forward of correlation_cuda_kernel.cu:41-114 for any (pad_size, kernel_size K, max_displacement md, stride1 s1,
// stride2 s2), on NCHW without the padded NHWC copies:
//   out[n, tc, oy, ox] = 1/(K^2 C) sum_{j,i in [-kr,kr]} sum_c p1[c, y1+j, x1+i] * p2[c, y1+j+tj*s2, x1+i+ti*s2]
//   y1 = oy*s1 + md, x1 = ox*s1 + md (coordinates in the zero-padded frame), tc = (tj+dr)*(2dr+1) + (ti+dr), dr = md/s2.
struct CorrGen {
  int C, H, W, pad, kr, md, s1, s2, dr, Ho, Wo;
};
__device__ __forceinline__ float padded(const float* __restrict__ p, const CorrGen& g, int c, int Y, int X) {
  const int y = Y - g.pad, x = X - g.pad;  // padded frame -> image
  return (y >= 0 && y < g.H && x >= 0 && x < g.W) ? p[((long)c * g.H + y) * g.W + x] : 0.f;
}

__global__ __launch_bounds__(256) void corr_general_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                               float* __restrict__ out, CorrGen g, long total) {
  const int ds = 2 * g.dr + 1, K = 2 * g.kr + 1;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int ox = idx % g.Wo, oy = (idx / g.Wo) % g.Ho, tc = (idx / ((long)g.Wo * g.Ho)) % (ds * ds);
    const int n = idx / ((long)g.Wo * g.Ho * ds * ds);
    const int tj = tc / ds - g.dr, ti = tc % ds - g.dr;
    const int y1 = oy * g.s1 + g.md, xx1 = ox * g.s1 + g.md;
    const float* p1 = x1 + (long)n * g.C * g.H * g.W;
    const float* p2 = x2 + (long)n * g.C * g.H * g.W;
    float s = 0.f;
    for (int j = -g.kr; j <= g.kr; ++j)
      for (int i = -g.kr; i <= g.kr; ++i)
        for (int c = 0; c < g.C; ++c)
          s = fmaf(padded(p1, g, c, y1 + j, xx1 + i), padded(p2, g, c, y1 + j + tj * g.s2, xx1 + i + ti * g.s2), s);
    out[idx] = s / (float)(K * K * g.C);
  }
}

// The exact gradients of that forward (one thread per input element, gather form):
//   gx1[c,y,x] = 1/(K^2 C) sum_{j,i} [oy = (y+pad-md-j)/s1, ox = (x+pad-md-i)/s1 integral and in range]
//                sum_tc gout[tc,oy,ox] * p2[c, y+pad+tj*s2, x+pad+ti*s2]
//   gx2[c,y,x] = 1/(K^2 C) sum_tc sum_{j,i} [oy = (y+pad-tj*s2-j-md)/s1, ox likewise, integral and in range]
//                gout[tc,oy,ox] * p1[c, y+pad-tj*s2, x+pad-ti*s2]
// (The CUDA extension's backward bounds its output window with floor divisions, correlation_cuda_kernel.cu:148-151,
// which for stride1 > 1 admits offsets outside the kernel window; this is the gradient of the forward.)
__global__ __launch_bounds__(256) void corr_general_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ x1,
                                                               const float* __restrict__ x2, float* __restrict__ gx1,
                                                               float* __restrict__ gx2, CorrGen g, long total) {
  const int ds = 2 * g.dr + 1, K = 2 * g.kr + 1;
  const float inv = 1.f / (float)(K * K * g.C);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % g.W, y = (idx / g.W) % g.H, c = (idx / ((long)g.W * g.H)) % g.C;
    const int n = idx / ((long)g.W * g.H * g.C);
    const float* go = gout + (long)n * ds * ds * g.Ho * g.Wo;
    const float* p1 = x1 + (long)n * g.C * g.H * g.W;
    const float* p2 = x2 + (long)n * g.C * g.H * g.W;
    const int Y = y + g.pad, X = x + g.pad;
    float s1 = 0.f, s2 = 0.f;
    for (int tc = 0; tc < ds * ds; ++tc) {
      const int tj = (tc / ds - g.dr) * g.s2, ti = (tc % ds - g.dr) * g.s2;
      const float* gt = go + (long)tc * g.Ho * g.Wo;
      for (int j = -g.kr; j <= g.kr; ++j)
        for (int i = -g.kr; i <= g.kr; ++i) {
          // role of x1: (Y, X) = (y1 + j, x1 + i)
          int ny = Y - g.md - j, nx = X - g.md - i;
          if (ny >= 0 && nx >= 0 && ny % g.s1 == 0 && nx % g.s1 == 0 && ny / g.s1 < g.Ho && nx / g.s1 < g.Wo)
            s1 = fmaf(gt[(long)(ny / g.s1) * g.Wo + nx / g.s1], padded(p2, g, c, Y + tj, X + ti), s1);
          // role of x2: (Y, X) = (y1 + j + tj, x1 + i + ti)
          ny = Y - tj - j - g.md, nx = X - ti - i - g.md;
          if (ny >= 0 && nx >= 0 && ny % g.s1 == 0 && nx % g.s1 == 0 && ny / g.s1 < g.Ho && nx / g.s1 < g.Wo)
            s2 = fmaf(gt[(long)(ny / g.s1) * g.Wo + nx / g.s1], padded(p1, g, c, Y - tj, X - ti), s2);
        }
    }
    if (gx1) gx1[idx] = s1 * inv;
    if (gx2) gx2[idx] = s2 * inv;
  }
}

}  // namespace

extern "C" int arflow_warp_nearest_fwd(const float* src, const float* flow, float* out, int B, int C, int Hs, int Ws, int H,
                                       int W, long flow_bstride, int pad_mode, int align_corners, int norm_mode,
                                       arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(src);
  AF_REQUIRE_PTR(flow);
  AF_REQUIRE_PTR(out);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0 && B <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(flow_bstride >= 2L * H * W, ARFLOW_ESHAPE);
  AF_REQUIRE(pad_mode == ARFLOW_PAD_ZEROS || pad_mode == ARFLOW_PAD_BORDER, ARFLOW_EPARAM);
  AF_REQUIRE(norm_mode >= ARFLOW_NORM_ARFLOW && norm_mode <= ARFLOW_NORM_UFLOW_ABS, ARFLOW_EPARAM);
  hipLaunchKernelGGL(warp_nearest_fwd_kernel, dim3(af_cdiv(W, 256), H, B), dim3(256), 0, (hipStream_t)stream, src, flow, out,
                     C, Hs, Ws, H, W, flow_bstride, pad_mode, align_corners, norm_mode);
  return af_launch_status();
}

extern "C" int arflow_warp_nearest_bwd(const float* gout, const float* flow, float* gsrc, int B, int C, int Hs, int Ws, int H,
                                       int W, long flow_bstride, int pad_mode, int align_corners, int norm_mode,
                                       arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(gout);
  AF_REQUIRE_PTR(flow);
  AF_REQUIRE_PTR(gsrc);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0 && B <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(flow_bstride >= 2L * H * W, ARFLOW_ESHAPE);
  AF_REQUIRE(pad_mode == ARFLOW_PAD_ZEROS || pad_mode == ARFLOW_PAD_BORDER, ARFLOW_EPARAM);
  AF_REQUIRE(norm_mode >= ARFLOW_NORM_ARFLOW && norm_mode <= ARFLOW_NORM_UFLOW_ABS, ARFLOW_EPARAM);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(gsrc, 0, sizeof(float) * (size_t)B * C * Hs * Ws, st);
  if (e != hipSuccess) return af_hip_status(e);
  hipLaunchKernelGGL(warp_nearest_bwd_kernel, dim3(af_cdiv(W, 256), H, B), dim3(256), 0, st, gout, flow, gsrc, C, Hs, Ws, H,
                     W, flow_bstride, pad_mode, align_corners, norm_mode);
  return af_launch_status();
}

extern "C" int arflow_warp_bicubic_fwd(const float* src, const float* flow, float* out, int B, int C, int Hs, int Ws, int H,
                                       int W, long flow_bstride, int pad_mode, int align_corners, int norm_mode,
                                       arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(src);
  AF_REQUIRE_PTR(flow);
  AF_REQUIRE_PTR(out);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0 && B <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(flow_bstride >= 2L * H * W, ARFLOW_ESHAPE);
  AF_REQUIRE(pad_mode == ARFLOW_PAD_ZEROS || pad_mode == ARFLOW_PAD_BORDER, ARFLOW_EPARAM);
  AF_REQUIRE(norm_mode >= ARFLOW_NORM_ARFLOW && norm_mode <= ARFLOW_NORM_UFLOW_ABS, ARFLOW_EPARAM);
  hipLaunchKernelGGL(warp_bicubic_fwd_kernel, dim3(af_cdiv(W, 256), H, B), dim3(256), 0, (hipStream_t)stream, src, flow, out,
                     C, Hs, Ws, H, W, flow_bstride, pad_mode, align_corners, norm_mode);
  return af_launch_status();
}

extern "C" int arflow_warp_bicubic_bwd(const float* gout, const float* src, const float* flow, float* gsrc, float* gflow, int B,
                                       int C, int Hs, int Ws, int H, int W, long flow_bstride, int pad_mode, int align_corners,
                                       int norm_mode, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(gout);
  AF_REQUIRE_PTR(src);
  AF_REQUIRE_PTR(flow);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0 && B <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(flow_bstride >= 2L * H * W, ARFLOW_ESHAPE);
  AF_REQUIRE(pad_mode == ARFLOW_PAD_ZEROS || pad_mode == ARFLOW_PAD_BORDER, ARFLOW_EPARAM);
  AF_REQUIRE(norm_mode >= ARFLOW_NORM_ARFLOW && norm_mode <= ARFLOW_NORM_UFLOW_ABS, ARFLOW_EPARAM);
  if (!gsrc && !gflow) return ARFLOW_OK;
  hipStream_t st = (hipStream_t)stream;
  if (gsrc) {
    hipError_t e = hipMemsetAsync(gsrc, 0, sizeof(float) * (size_t)B * C * Hs * Ws, st);
    if (e != hipSuccess) return af_hip_status(e);
  }
  hipLaunchKernelGGL(warp_bicubic_bwd_kernel, dim3(af_cdiv(W, 256), H, B), dim3(256), 0, st, gout, src, flow, gsrc, gflow, C, Hs,
                     Ws, H, W, flow_bstride, pad_mode, align_corners, norm_mode);
  return af_launch_status();
}

// called by arflow_census_fwd / arflow_census_bwd (photo.hip) for radius > 3
int census_any_fwd(const float* im_a, const float* im_b, const float* mask, float* ham, float* dham, float* sums, int nrows,
                   int B, int H, int W, int R, hipStream_t st) {
  hipLaunchKernelGGL(census_any_fwd_kernel, dim3(af_cdiv(W, 256), H, B), dim3(256), 0, st, im_a, im_b, mask, ham, dham, sums,
                     nrows, H, W, R);
  return af_launch_status();
}
int census_any_bwd(const float* im_a, const float* im_b, const float* gham, const float* scale, float* g_im_b, int B, int H,
                   int W, int R, hipStream_t st) {
  hipLaunchKernelGGL(census_any_bwd_kernel, dim3(af_cdiv(W, 256), H, B), dim3(256), 0, st, im_a, im_b, gham, scale, g_im_b, H,
                     W, R);
  return af_launch_status();
}

extern "C" int arflow_ssim_fwd(const float* x, const float* y, float* out, int B, int C, int H, int W, int md,
                               arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(x);
  AF_REQUIRE_PTR(y);
  AF_REQUIRE_PTR(out);
  AF_REQUIRE(md >= 1 && md <= 16, ARFLOW_EPARAM);
  AF_REQUIRE(B > 0 && C > 0 && H > 2 * md && W > 2 * md && (long)B * C <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  hipLaunchKernelGGL(ssim_any_fwd_kernel, dim3(af_cdiv(W - 2 * md, 256), H - 2 * md, B * C), dim3(256), 0,
                     (hipStream_t)stream, x, y, out, H, W, md);
  return af_launch_status();
}

extern "C" int arflow_ssim_bwd(const float* x, const float* y, const float* gmap, float* gx, int B, int C, int H, int W, int md,
                               arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(x);
  AF_REQUIRE_PTR(y);
  AF_REQUIRE_PTR(gmap);
  AF_REQUIRE_PTR(gx);
  AF_REQUIRE(md >= 1 && md <= 16, ARFLOW_EPARAM);
  AF_REQUIRE(B > 0 && C > 0 && H > 2 * md && W > 2 * md && (long)B * C <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  hipLaunchKernelGGL(ssim_any_bwd_kernel, dim3(af_cdiv(W, 256), H, B * C), dim3(256), 0, (hipStream_t)stream, x, y, gmap, gx,
                     H, W, md);
  return af_launch_status();
}

static int corr_general_cfg(CorrGen& g, int C, int H, int W, int pad_size, int kernel_size, int max_disp, int stride1,
                            int stride2) {
  AF_REQUIRE(kernel_size >= 1 && (kernel_size & 1) == 1 && max_disp >= 0 && stride1 >= 1 && stride2 >= 1 && pad_size >= 0,
             ARFLOW_EPARAM);
  g.C = C, g.H = H, g.W = W, g.pad = pad_size, g.kr = (kernel_size - 1) / 2, g.md = max_disp, g.s1 = stride1, g.s2 = stride2;
  g.dr = max_disp / stride2;
  const int border = g.kr + max_disp;
  const int ph = H + 2 * pad_size - 2 * border, pw = W + 2 * pad_size - 2 * border;
  AF_REQUIRE(ph > 0 && pw > 0, ARFLOW_ESHAPE);
  g.Ho = (ph + stride1 - 1) / stride1;  // ceil, correlation_cuda.cc:33-34
  g.Wo = (pw + stride1 - 1) / stride1;
  return ARFLOW_OK;
}

extern "C" int arflow_corr_general_out_size(int H, int W, int pad_size, int kernel_size, int max_disp, int stride1,
                                            int stride2, int* out_channels, int* out_h, int* out_w) {
  CorrGen g;
  const int rc = corr_general_cfg(g, 1, H, W, pad_size, kernel_size, max_disp, stride1, stride2);
  if (rc) return rc;
  if (out_channels) *out_channels = (2 * g.dr + 1) * (2 * g.dr + 1);
  if (out_h) *out_h = g.Ho;
  if (out_w) *out_w = g.Wo;
  return ARFLOW_OK;
}

extern "C" int arflow_corr_general_fwd(const float* x1, const float* x2, float* out, int B, int C, int H, int W, int pad_size,
                                       int kernel_size, int max_disp, int stride1, int stride2, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(x1);
  AF_REQUIRE_PTR(x2);
  AF_REQUIRE_PTR(out);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, ARFLOW_ESHAPE);
  CorrGen g;
  const int rc = corr_general_cfg(g, C, H, W, pad_size, kernel_size, max_disp, stride1, stride2);
  if (rc) return rc;
  const long total = (long)B * (2 * g.dr + 1) * (2 * g.dr + 1) * g.Ho * g.Wo;
  const unsigned blocks = (unsigned)((total + 255) / 256 > 65535 * 16 ? 65535 * 16 : (total + 255) / 256);
  hipLaunchKernelGGL(corr_general_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x1, x2, out, g, total);
  return af_launch_status();
}

extern "C" int arflow_corr_general_bwd(const float* gout, const float* x1, const float* x2, float* gx1, float* gx2, int B,
                                       int C, int H, int W, int pad_size, int kernel_size, int max_disp, int stride1,
                                       int stride2, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(gout);
  AF_REQUIRE_PTR(x1);
  AF_REQUIRE_PTR(x2);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, ARFLOW_ESHAPE);
  if (!gx1 && !gx2) return ARFLOW_OK;
  CorrGen g;
  const int rc = corr_general_cfg(g, C, H, W, pad_size, kernel_size, max_disp, stride1, stride2);
  if (rc) return rc;
  const long total = (long)B * C * H * W;
  const unsigned blocks = (unsigned)((total + 255) / 256 > 65535 * 16 ? 65535 * 16 : (total + 255) / 256);
  hipLaunchKernelGGL(corr_general_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, gout, x1, x2, gx1, gx2, g,
                     total);
  return af_launch_status();
}
