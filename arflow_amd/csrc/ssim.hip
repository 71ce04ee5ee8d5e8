// SSIM (3x3, un-padded) + L1 photometric term of the ARFlow pyramid loss for gfx950 -- forward and backward
// (losses/loss_blocks.py:65-84 SSIM, losses/flow_loss.py:13-27).  Split from photo.hip (census kernels) because the two
// want different compiler settings: these kernels are built WITHOUT the SLP vectoriser (Makefile: its v_pk_* pairing
// costs more v_mov than it saves here: forward 28.4 -> 23.5 us, backward 53.3 -> 45.6 us at 8x3x384x640), the census
// kernels with it (backward 59 -> 69 us without).
#include "common.hpp"

namespace {

constexpr int TX = 32, TY = 8;  // pixel tile = 256 threads, lanes run along x

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
                                                           float* __restrict__ sums, int nrows, int nimg, int C, int H, int W) {
  __shared__ float tx[TY + 2][TX + 3];  // x = recons*mask
  __shared__ float ty[TY + 2][TX + 3];  // y = im*mask
  __shared__ float red[3 * (TX * TY / 64)];
  int btx_, bty_, b;
  if (!af_tile_of_block((W + TX - 1) / TX, (H + TY - 1) / TY, nimg, btx_, bty_, b)) {
    if (threadIdx.x == 0) af_store_partial(sums, nrows, 0.f, 0.f, 0.f);  // padding workgroup: its row must be defined
    return;
  }
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
  if (threadIdx.x == 0) af_store_partial(sums, nrows, part[0], part[1], part[2]);
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
                                                 float* __restrict__ sums, int nrows, int nimg, int C, int H, int W) {
  __shared__ __attribute__((aligned(16))) float X[(TYH + 2) * P];
  __shared__ __attribute__((aligned(16))) float Y[(TYH + 2) * P];
  __shared__ float red[3 * (NT / 64)];
  int btx, bty, b;
  if (!af_tile_of_block((W + TXW - 1) / TXW, (H + TYH - 1) / TYH, nimg, btx, bty, b)) {
    if (threadIdx.x == 0) af_store_partial(sums, nrows, 0.f, 0.f, 0.f);  // padding workgroup: its row must be defined
    return;
  }
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
  if (threadIdx.x == 0) af_store_partial(sums, nrows, part[0], part[1], part[2]);
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

extern "C" int arflow_photo_fwd(const float* im, const float* recons, const float* mask, float* ssim_map,
                                float* sums, int B, int C, int H, int W, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(im);
  AF_REQUIRE_PTR(recons);
  AF_REQUIRE_PTR(sums);
  AF_REQUIRE(B > 0 && C > 0 && H >= 3 && W >= 3 && B <= 65535, ARFLOW_ESHAPE);
  hipStream_t st = (hipStream_t)stream;
  const int nrows = af_sums_rows(B, H, W);
  if ((W & 3) == 0) {
    namespace p4 = photo4;
    dim3 g4(af_grid_for_tiles((long)af_cdiv(W, p4::TXW) * af_cdiv(H, p4::TYH) * B));
    hipLaunchKernelGGL(p4::fwd_kernel, g4, dim3(p4::NT), 0, st, im, recons, mask, ssim_map, sums, nrows, B, C, H, W);
    return af_launch_status();
  }
  dim3 grid(af_grid_for_tiles((long)af_cdiv(W, TX) * af_cdiv(H, TY) * B));
  hipLaunchKernelGGL(photo_fwd_kernel, grid, dim3(TX * TY), 0, st, im, recons, mask, ssim_map, sums, nrows, B, C, H, W);
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
