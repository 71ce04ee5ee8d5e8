// Bilinear flow warp (forward / backward), forward-splat occlusion maps and coordinate masks for
// gfx950.  Reference arithmetic: utils/warp_utils.py (flow_warp :83-90, get_corresponding_map
// :26-80, get_occu_mask_bidirection :93-100, border_mask :119-134), utils/uflow_utils.py
// (flow_to_warp :6-32, mask_invalid :35-50, resample :53-77, compute_range_map :80-160) and torch's
// grid_sample (ATen/native/GridSampler.h) which the reference calls.
//
// Layout: one lane per output pixel, consecutive lanes along x, so flow reads, every output
// channel plane and (for small displacements) the four source taps are coalesced.  Coordinates
// and bilinear weights are computed once per pixel and reused for all channels.  HBM traffic is
// the compulsory 4*px*(2C+2) bytes forward, 4*px*(3C+4) backward.
#include "common.hpp"
#include "taps.hpp"
#include "featnorm_stats.hpp"
#include "smooth_dev.hpp"

namespace {

// Forward warp.  One workgroup = an 8 x 32 tile of output pixels (XCD-aware tile order).  The tile's taps
// land in a small source window; that window is staged in LDS with aligned 16-byte loads (full
// coalesced rows: dword-granular gathers straight from global spent 2.4x the bytes in L1->L2 traffic
// and ran at 25 % of the HBM roofline) and the four taps of every pixel are then read from LDS.
// Falls back to direct (batched) gathers when the window does not fit or rows are not 16-byte aligned.
// source storage: float, or bf16 bit patterns (opt-in bf16 storage of the warped features, SURVEY section 8(f)-4;
// sampling arithmetic and outputs stay fp32)
typedef unsigned short bf16_t;
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16_t* p) {  // 4 bf16 = one 8-byte load
  const uint2 r = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                     __uint_as_float(r.y & 0xffff0000u));
}

namespace fwd_win {
constexpr int TX = 32, TY = 8, NT = 256, HMAX = 24;
// CCH = channels staged per chunk.  These kernels are latency-bound chains (flow -> taps -> box -> window -> taps),
// so what counts is how many workgroups a CU holds, i.e. the LDS window: 4 channels (27.6 KB, 5 per CU) ran the
// B16 C32 96x160 forward in 24.5 us, 2 channels (13.8 KB, 8 per CU = the wave limit) in 18.8 us; the 3-channel
// image warps of the losses use 3 (20.7 KB, 7 per CU): 30.0 -> 24.7 us

// Stage CCH channels of the source window (bh rows x WQ aligned float4 each) into LDS.  All loads of a
// thread are issued before the first LDS write and none is branched around (a slot outside the window
// reads the plane's first 16 bytes instead): with `if (inside) win[..] = load` per slot hipcc waits for
// every load before issuing the next -- 2 x CCH serialised round trips per chunk.
// The window of a chunk travels in two halves so that the NEXT chunk's global loads are in flight while the current
// chunk is computed (round 2: the kernels are latency chains -- 16 chunks x (load -> barrier -> 4 LDS taps -> barrier)
// per workgroup; with the loads issued one chunk ahead the chain is one round trip shorter per chunk):
//   window_load  : all loads of a thread, none branched around (a slot outside the window reads the plane's first
//                  16 bytes instead: with `if (inside) load` hipcc waits for every load before issuing the next);
//   window_store : registers -> LDS.
template <int WQ>
struct WindowPlan {
  static constexpr int ITER = (HMAX * WQ + NT - 1) / NT;
  bool ok[ITER];
  int dst[ITER];
  long off[ITER];
};
template <int WQ>
__device__ __forceinline__ WindowPlan<WQ> window_plan(int Ws, int ax0, int by0, int bh) {
  constexpr int WP = 4 * WQ;
  WindowPlan<WQ> pl;
  const int per = bh * WQ;
#pragma unroll
  for (int it = 0; it < WindowPlan<WQ>::ITER; ++it) {
    const int i = threadIdx.x + it * NT;
    const int r = i / WQ, xs = i - r * WQ;
    pl.ok[it] = i < per && ax0 + 4 * xs < Ws;
    pl.dst[it] = r * WP + 4 * xs;
    pl.off[it] = pl.ok[it] ? (long)(by0 + r) * Ws + ax0 + 4 * xs : 0;
  }
  return pl;
}
template <int WQ, int CCH, typename TS>
__device__ __forceinline__ void window_load(float4 (&v)[CCH][WindowPlan<WQ>::ITER], const WindowPlan<WQ>& pl,
                                            const TS* __restrict__ sp, int c0, int C, int ss) {
#pragma unroll
  for (int it = 0; it < WindowPlan<WQ>::ITER; ++it)
#pragma unroll
    for (int c = 0; c < CCH; ++c) v[c][it] = load4(sp + (long)min(max(c0 + c, 0), C - 1) * ss + pl.off[it]);
}
template <int WQ, int CCH>
__device__ __forceinline__ void window_store(float* __restrict__ win, const float4 (&v)[CCH][WindowPlan<WQ>::ITER],
                                             const WindowPlan<WQ>& pl, int c0, int C) {
  constexpr int WP = 4 * WQ;
#pragma unroll
  for (int it = 0; it < WindowPlan<WQ>::ITER; ++it)
#pragma unroll
    for (int c = 0; c < CCH; ++c)
      if (pl.ok[it] && c0 + c < C) *reinterpret_cast<float4*>(win + c * HMAX * WP + pl.dst[it]) = v[c][it];
}

// MOM: also accumulate this thread's share of the moments the feature normalisation needs -- (sum, sum of squares) of
// the first feature map at the thread's pixel (`x1p`, read one chunk ahead like the window) and of the warped values
// it has just produced -- into mom[0..3] (fp32 over <= C values per thread; the caller continues in double).
template <int WQ, int CCH, typename TS, bool MOM = false>  // window row = WQ float4
__device__ __forceinline__ void run(float* __restrict__ win, const TS* __restrict__ sp, float* __restrict__ op,
                                    const TapPlan& p, bool inside, int C, int ss, int os, int Ws, int ax0, int by0,
                                    int bh, int l0, int l1, int l2, int l3, const float* __restrict__ x1p = nullptr,
                                    float* mom = nullptr) {
  constexpr int WP = 4 * WQ;
  const WindowPlan<WQ> pl = window_plan<WQ>(Ws, ax0, by0, bh);
  float4 v[CCH][WindowPlan<WQ>::ITER];
  float xn[CCH];
  auto fetch_x1 = [&](int cc) {
#pragma unroll
    for (int c = 0; c < CCH; ++c) xn[c] = (MOM && x1p && inside && cc + c < C) ? x1p[(long)(cc + c) * os] : 0.f;
  };
  const int step = gridDim.y * CCH;  // channels are independent: at small levels they are spread over gridDim.y workgroups
  int c0 = blockIdx.y * CCH;
  if (c0 < C) {
    if (MOM) fetch_x1(c0);
    window_load<WQ, CCH, TS>(v, pl, sp, c0, C, ss);
  }
  for (; c0 < C; c0 += step) {
    float xa[CCH];
#pragma unroll
    for (int c = 0; c < CCH; ++c) xa[c] = MOM ? xn[c] : 0.f;
    window_store<WQ, CCH>(win, v, pl, c0, C);
    if (c0 + step < C) {  // next chunk: in flight during this one
      if (MOM) fetch_x1(c0 + step);
      window_load<WQ, CCH, TS>(v, pl, sp, c0 + step, C, ss);
    }
    __syncthreads();
    if (inside) {
#pragma unroll
      for (int c = 0; c < CCH; ++c) {
        if (c0 + c < C) {
          const float* w = win + c * HMAX * WP;
          float r = p.ok[0] ? w[l0] * p.w[0] : 0.f;
          r = p.ok[1] ? fmaf(w[l1], p.w[1], r) : r;
          r = p.ok[2] ? fmaf(w[l2], p.w[2], r) : r;
          r = p.ok[3] ? fmaf(w[l3], p.w[3], r) : r;
          op[(long)(c0 + c) * os] = r;
          if (MOM) {
            mom[0] += xa[c], mom[1] = fmaf(xa[c], xa[c], mom[1]);
            mom[2] += r, mom[3] = fmaf(r, r, mom[3]);
          }
        }
      }
    }
    __syncthreads();
  }
}
}  // namespace fwd_win

template <int CCH, typename TS = float>
__global__ __launch_bounds__(256) void warp_fwd_kernel(const TS* __restrict__ src,
                                                       const float* __restrict__ flow,
                                                       float* __restrict__ out, float* __restrict__ valid,
                                                       int nimg, int C, int Hs, int Ws, int H, int W, long fbs,
                                                       int pad, int align, int norm) {
  using namespace fwd_win;
  __shared__ __attribute__((aligned(16))) float win[CCH * HMAX * 72];
  __shared__ int red[4][NT / 64];
  __shared__ int box[4];
  int btx, bty, b;
  if (!af_tile_of_block((W + TX - 1) / TX, (H + TY - 1) / TY, nimg, btx, bty, b)) return;
  const int x = btx * TX + (int)(threadIdx.x & 31), y = bty * TY + (int)(threadIdx.x >> 5);
  const bool inside = x < W && y < H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  Taps t;
  if (inside) {
    const float* fb = flow + (long)b * fbs + (long)y * W + x;
    const float u = fb[0], v = fb[(long)H * W];
    t = make_taps((float)x, (float)y, u, v, H, W, Hs, Ws, pad, align != 0, norm);
    if (valid) {
      const bool abs_in = norm == ARFLOW_NORM_UFLOW_ABS;
      const float cx = abs_in ? u : (float)x + u, cy = abs_in ? v : (float)y + v;
      valid[((long)b * H + y) * W + x] =
          (cx >= 0.f && cx <= (float)(W - 1) && cy >= 0.f && cy <= (float)(H - 1)) ? 1.f : 0.f;
    }
  } else {
    t.vx0 = t.vx1 = t.vy0 = t.vy1 = false;
    t.x0 = t.y0 = 0;
    t.wx0 = t.wx1 = t.wy0 = t.wy1 = t.dx = t.dy = 0.f;
  }
  const TapPlan p = plan_taps(t, Hs, Ws);
  // bounding box of the valid taps over the workgroup
  const bool any = (t.vx0 || t.vx1) && (t.vy0 || t.vy1);
  int lo_x = any ? t.x0 + (t.vx0 ? 0 : 1) : 0x7fffffff, hi_x = any ? t.x0 + (t.vx1 ? 1 : 0) : -0x7fffffff;
  int lo_y = any ? t.y0 + (t.vy0 ? 0 : 1) : 0x7fffffff, hi_y = any ? t.y0 + (t.vy1 ? 1 : 0) : -0x7fffffff;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo_x = min(lo_x, __shfl_xor(lo_x, off, 64));
    lo_y = min(lo_y, __shfl_xor(lo_y, off, 64));
    hi_x = max(hi_x, __shfl_xor(hi_x, off, 64));
    hi_y = max(hi_y, __shfl_xor(hi_y, off, 64));
  }
  if (lane == 0) red[0][wave] = lo_x, red[1][wave] = lo_y, red[2][wave] = hi_x, red[3][wave] = hi_y;
  __syncthreads();
  if (threadIdx.x == 0) {
    int a = red[0][0], bb = red[1][0], c = red[2][0], d = red[3][0];
    for (int w = 1; w < NT / 64; ++w)
      a = min(a, red[0][w]), bb = min(bb, red[1][w]), c = max(c, red[2][w]), d = max(d, red[3][w]);
    box[0] = a, box[1] = bb, box[2] = c, box[3] = d;
  }
  __syncthreads();
  const int bx0 = box[0], by0 = box[1];
  const int bh = box[3] - by0 + 1;
  const int ax0 = bx0 & ~3;                   // 16-byte aligned window start
  const int aw = box[2] - ax0 + 1;            // floats needed from ax0
  const bool empty = box[2] < bx0;
  const int ss = Hs * Ws, os = H * W;
  const TS* sp = src + (long)b * C * ss;
  float* op = out + (long)b * C * os + (long)y * W + x;

  if (!empty && (Ws & 3) == 0 && bh <= HMAX && aw <= 72) {
    // LDS offsets of the four taps (clamped like the global ones, relative to the window)
    const int xa = min(max(t.x0, 0), Ws - 1) - ax0, xb = min(max(t.x0 + 1, 0), Ws - 1) - ax0;
    const int ya = min(max(t.y0, 0), Hs - 1) - by0, yb = min(max(t.y0 + 1, 0), Hs - 1) - by0;
    // taps of pixels without any valid tap may point outside the window: clamp (their weight is unused)
    const int cxa = min(max(xa, 0), 71), cxb = min(max(xb, 0), 71);
    const int cya = min(max(ya, 0), HMAX - 1), cyb = min(max(yb, 0), HMAX - 1);
    if (aw <= 48)
      run<12, CCH, TS>(win, sp, op, p, inside, C, ss, os, Ws, ax0, by0, bh, cya * 48 + cxa, cya * 48 + cxb, cyb * 48 + cxa,
              cyb * 48 + cxb);
    else
      run<18, CCH, TS>(win, sp, op, p, inside, C, ss, os, Ws, ax0, by0, bh, cya * 72 + cxa, cya * 72 + cxb, cyb * 72 + cxa,
              cyb * 72 + cxb);
    return;
  }
  if (!inside) return;
  if (empty) {
    for (int c = blockIdx.y; c < C; c += gridDim.y) op[(long)c * os] = 0.f;
    return;
  }
  // fallback: direct gathers, U channels per round (all 4*U loads in flight together)
  constexpr int U = CCH;  // (the channel split over gridDim.y counts chunks of CCH)
  for (int c0 = blockIdx.y * U; c0 < C; c0 += gridDim.y * U) {
    float a[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const TS* s = sp + (long)min(c0 + u, C - 1) * ss;  // clamped: the loads stay unconditional
#pragma unroll
      for (int k = 0; k < 4; ++k) a[u][k] = to_f32(s[p.o[k]]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      asm volatile("" : "+v"(a[u][0]), "+v"(a[u][1]), "+v"(a[u][2]), "+v"(a[u][3]));  // loads may not be predicated away
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float r = p.ok[0] ? a[u][0] * p.w[0] : 0.f;
      r = p.ok[1] ? fmaf(a[u][1], p.w[1], r) : r;
      r = p.ok[2] ? fmaf(a[u][2], p.w[2], r) : r;
      r = p.ok[3] ? fmaf(a[u][3], p.w[3], r) : r;
      if (c0 + u < C) op[(long)(c0 + u) * os] = r;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Level forward, first launch: everything a pyramid level does in front of its cost volume
// (models/pwclite_uflow.py:203-214, models/uflow_model.py:160-172) in ONE pass --
//   UP:  flow = interpolate(flow_coarse * 2, x2, bilinear)   (ATen upsample_bilinear2d arithmetic, either align flag),
//        written to `flow_up` and (optionally) into its slot of the decoder's concatenated input `flow_up2`;
//   x2w = bilinear warp of the second feature map by that flow (the warp_fwd_kernel above, same LDS window);
//   the partial moments of normalize_features -- (sum x1, sum x1^2, sum x2w, sum x2w^2) of this tile -- as one row
//   of 4 doubles per workgroup in `acc` ([B][rows][4], rows = tiles per sample x gridDim.y), which the correlation
//   launch folds into its epilogue: the normalised maps are never written (the moment pass and the apply pass of
//   featnorm.hip and ATen's interpolate + mul launches disappear).
// ------------------------------------------------------------------------------------------------
template <bool UP, int CCH = 2>
__global__ __launch_bounds__(256) void level_warp_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ src,
                                                             const float* __restrict__ flow, float* __restrict__ flow_up,
                                                             float* __restrict__ flow_up2, long fu2_bs,
                                                             float* __restrict__ out, double* __restrict__ acc, int nimg,
                                                             int C, int H, int W, long fbs, int pad, int align, int norm,
                                                             int up_align) {
  using namespace fwd_win;
  __shared__ __attribute__((aligned(16))) float win[CCH * HMAX * 72];
  __shared__ int red[4][NT / 64];
  __shared__ int box[4];
  __shared__ double dscratch[4 * (NT / 64)];
  int btx, bty, b;
  const int ntx = (W + TX - 1) / TX, nty = (H + TY - 1) / TY;
  if (!af_tile_of_block(ntx, nty, nimg, btx, bty, b)) return;
  const int x = btx * TX + (int)(threadIdx.x & 31), y = bty * TY + (int)(threadIdx.x >> 5);
  const bool inside = x < W && y < H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Hs = H, Ws = W;
  Taps t = no_taps();
  if (inside) {
    float u, v;
    if (UP) {
      const int Hc = H / 2, Wc = W / 2;
      int xa, xb, ya, yb;
      float wx0, wx1, wy0, wy1;
      up2_source(x, Wc, W, up_align != 0, xa, xb, wx0, wx1);
      up2_source(y, Hc, H, up_align != 0, ya, yb, wy0, wy1);
      const float* fc = flow + (long)b * fbs;
      const long cs = (long)Hc * Wc;
      const float a00 = fc[ya * Wc + xa], a01 = fc[ya * Wc + xb], a10 = fc[yb * Wc + xa], a11 = fc[yb * Wc + xb];
      const float b00 = fc[cs + ya * Wc + xa], b01 = fc[cs + ya * Wc + xb], b10 = fc[cs + yb * Wc + xa],
                  b11 = fc[cs + yb * Wc + xb];
      // interpolate(2 f) = 2 interpolate(f) exactly in fp32 (a power-of-two scale commutes with every rounding)
      u = 2.f * (wy0 * (wx0 * a00 + wx1 * a01) + wy1 * (wx0 * a10 + wx1 * a11));
      v = 2.f * (wy0 * (wx0 * b00 + wx1 * b01) + wy1 * (wx0 * b10 + wx1 * b11));
      if (blockIdx.y == 0) {
        const long o = (long)y * W + x, os2 = (long)H * W;
        if (flow_up) flow_up[(long)b * 2 * os2 + o] = u, flow_up[(long)b * 2 * os2 + os2 + o] = v;
        if (flow_up2) flow_up2[(long)b * fu2_bs + o] = u, flow_up2[(long)b * fu2_bs + os2 + o] = v;
      }
    } else {
      const float* fb = flow + (long)b * fbs + (long)y * W + x;
      u = fb[0], v = fb[(long)H * W];
    }
    t = make_taps((float)x, (float)y, u, v, H, W, Hs, Ws, pad, align != 0, norm);
  }
  const TapPlan p = plan_taps(t, Hs, Ws);
  const bool any = (t.vx0 || t.vx1) && (t.vy0 || t.vy1);
  int lo_x = any ? t.x0 + (t.vx0 ? 0 : 1) : 0x7fffffff, hi_x = any ? t.x0 + (t.vx1 ? 1 : 0) : -0x7fffffff;
  int lo_y = any ? t.y0 + (t.vy0 ? 0 : 1) : 0x7fffffff, hi_y = any ? t.y0 + (t.vy1 ? 1 : 0) : -0x7fffffff;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo_x = min(lo_x, __shfl_xor(lo_x, off, 64));
    lo_y = min(lo_y, __shfl_xor(lo_y, off, 64));
    hi_x = max(hi_x, __shfl_xor(hi_x, off, 64));
    hi_y = max(hi_y, __shfl_xor(hi_y, off, 64));
  }
  if (lane == 0) red[0][wave] = lo_x, red[1][wave] = lo_y, red[2][wave] = hi_x, red[3][wave] = hi_y;
  __syncthreads();
  if (threadIdx.x == 0) {
    int a = red[0][0], bb = red[1][0], c = red[2][0], d = red[3][0];
    for (int w = 1; w < NT / 64; ++w)
      a = min(a, red[0][w]), bb = min(bb, red[1][w]), c = max(c, red[2][w]), d = max(d, red[3][w]);
    box[0] = a, box[1] = bb, box[2] = c, box[3] = d;
  }
  __syncthreads();
  const int bx0 = box[0], by0 = box[1];
  const int bh = box[3] - by0 + 1;
  const int ax0 = bx0 & ~3;
  const int aw = box[2] - ax0 + 1;
  const bool empty = box[2] < bx0;
  const int ss = Hs * Ws, os = H * W;
  const float* sp = src + (long)b * C * ss;
  float* op = out + (long)b * C * os + (long)y * W + x;
  const float* x1p = x1 ? x1 + (long)b * C * os + (long)y * W + x : nullptr;  // null: the first map's moments come from elsewhere
  float mom[4] = {0.f, 0.f, 0.f, 0.f};

  if (!empty && (Ws & 3) == 0 && bh <= HMAX && aw <= 72) {
    const int xa = min(max(t.x0, 0), Ws - 1) - ax0, xb = min(max(t.x0 + 1, 0), Ws - 1) - ax0;
    const int ya = min(max(t.y0, 0), Hs - 1) - by0, yb = min(max(t.y0 + 1, 0), Hs - 1) - by0;
    const int cxa = min(max(xa, 0), 71), cxb = min(max(xb, 0), 71);
    const int cya = min(max(ya, 0), HMAX - 1), cyb = min(max(yb, 0), HMAX - 1);
    if (aw <= 48)
      run<12, CCH, float, true>(win, sp, op, p, inside, C, ss, os, Ws, ax0, by0, bh, cya * 48 + cxa, cya * 48 + cxb,
                                cyb * 48 + cxa, cyb * 48 + cxb, x1p, mom);
    else
      run<18, CCH, float, true>(win, sp, op, p, inside, C, ss, os, Ws, ax0, by0, bh, cya * 72 + cxa, cya * 72 + cxb,
                                cyb * 72 + cxa, cyb * 72 + cxb, x1p, mom);
  } else if (inside) {
    // no tap of the tile inside the source (zeros out), or a window too large / unaligned rows (direct gathers)
    for (int c = blockIdx.y * CCH; c < C; c += gridDim.y * CCH) {
#pragma unroll
      for (int u = 0; u < CCH; ++u) {
        if (c + u >= C) continue;
        float r = 0.f;
        if (!empty) {
          const float* s = sp + (long)(c + u) * ss;
          const float a0 = s[p.o[0]], a1 = s[p.o[1]], a2 = s[p.o[2]], a3 = s[p.o[3]];
          r = p.ok[0] ? a0 * p.w[0] : 0.f;
          r = p.ok[1] ? fmaf(a1, p.w[1], r) : r;
          r = p.ok[2] ? fmaf(a2, p.w[2], r) : r;
          r = p.ok[3] ? fmaf(a3, p.w[3], r) : r;
        }
        op[(long)(c + u) * os] = r;
        const float xv = x1p ? x1p[(long)(c + u) * os] : 0.f;
        mom[0] += xv, mom[1] = fmaf(xv, xv, mom[1]);
        mom[2] += r, mom[3] = fmaf(r, r, mom[3]);
      }
    }
  }
  double dm[4] = {(double)mom[0], (double)mom[1], (double)mom[2], (double)mom[3]};
  featnorm::block_sum_f64<4, NT>(dm, dscratch);
  if (threadIdx.x == 0) {
    const int rows = ntx * nty * gridDim.y;
    double* row = acc + 4 * ((long)b * rows + (long)(bty * ntx + btx) * gridDim.y + blockIdx.y);
    row[0] = dm[0], row[1] = dm[1], row[2] = dm[2], row[3] = dm[3];
  }
}

// Shared prologue of the tile kernels: bounding box of the valid taps of the workgroup's 256 pixels.
struct TileBox {
  int x0, y0, x1, y1;  // inclusive; empty when x1 < x0
};
__device__ __forceinline__ TileBox tile_bbox(const Taps& t, int (*red)[4], int* box) {
  const bool any = (t.vx0 || t.vx1) && (t.vy0 || t.vy1);
  int lo_x = any ? t.x0 + (t.vx0 ? 0 : 1) : 0x7fffffff, hi_x = any ? t.x0 + (t.vx1 ? 1 : 0) : -0x7fffffff;
  int lo_y = any ? t.y0 + (t.vy0 ? 0 : 1) : 0x7fffffff, hi_y = any ? t.y0 + (t.vy1 ? 1 : 0) : -0x7fffffff;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo_x = min(lo_x, __shfl_xor(lo_x, off, 64));
    lo_y = min(lo_y, __shfl_xor(lo_y, off, 64));
    hi_x = max(hi_x, __shfl_xor(hi_x, off, 64));
    hi_y = max(hi_y, __shfl_xor(hi_y, off, 64));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[0][wave] = lo_x, red[1][wave] = lo_y, red[2][wave] = hi_x, red[3][wave] = hi_y;
  __syncthreads();
  if (threadIdx.x == 0) {
    int a = red[0][0], bb = red[1][0], c = red[2][0], d = red[3][0];
    for (int w = 1; w < 4; ++w) a = min(a, red[0][w]), bb = min(bb, red[1][w]), c = max(c, red[2][w]), d = max(d, red[3][w]);
    box[0] = a, box[1] = bb, box[2] = c, box[3] = d;
  }
  __syncthreads();
  TileBox r;
  r.x0 = box[0], r.y0 = box[1], r.x1 = box[2], r.y1 = box[3];
  return r;
}

// ------------------------------------------------------------------------------------------------
// d loss / d flow of the warp: per pixel  sum_c gout_c * (bilinear corner differences of src_c), times
// d coord / d flow.  Same tile / LDS source-window structure as the forward kernel; this is the whole
// backward of the loss-side image warps (their source is detached, losses/uflow_loss.py:31,34).
// ------------------------------------------------------------------------------------------------
// (inverse-window gather of d/d src, inv_gather below: does that kernel take the pair (target p, source q)?  qinfo = window
// centre of q (13 + 13 bits, biased by 4096) + bit 30 = "q is left to the atomics")
__device__ __forceinline__ bool af_gather_takes(int qinfo, int px, int py, int m) {
  const int cx = (qinfo & 0x1fff) - 4096, cy = ((qinfo >> 13) & 0x1fff) - 4096;
  return !(qinfo & (1 << 30)) && abs(px - cx) <= m && abs(py - cy) <= m;
}

namespace flow_grad {
using fwd_win::HMAX;
// NB (level backward): `gop` holds the gradient of the NORMALISED warped map; the normalisation's backward
//     d = r g - r G / N - r^3 Q (x - c) / (N - 1)         (featnorm.hip, bwd_apply_kernel)
// is applied on load -- x, the warped value, is the bilinear sum of the four taps this kernel reads anyway -- so the
// gradient of the raw warped map is never written; the same pass applies it to the first map's gradient
// (g1p + gdp, with x1p) and stores d/d x1 (d1p), all at the thread's pixel.
struct NormBwd {
  float rf, cg, cq, c1, c2;
};
__device__ __forceinline__ NormBwd norm_bwd_coeffs(const double* rows, int nrows, const float* st, long n, int mode) {
  double gq[2];
  featnorm::sum_rows<2>(rows, nrows, gq);
  const double r = 1.0 / (double)st[3], dn = (double)n;
  NormBwd nb;
  nb.rf = (float)r;
  nb.cg = (float)(r * gq[0] / (2.0 * dn));
  nb.cq = (float)(mode == ARFLOW_FEATNORM_JOINT ? r * r * r * gq[1] / (2.0 * dn - 1.0) : r * r * r * gq[1] / (2.0 * (dn - 1.0)));
  nb.c1 = mode == ARFLOW_FEATNORM_JOINT ? st[2] : st[0];
  nb.c2 = mode == ARFLOW_FEATNORM_JOINT ? st[2] : st[1];
  return nb;
}
__device__ __forceinline__ float norm_bwd_apply(const NormBwd& nb, float g, float x, float c) {
  return fmaf(nb.rf, g, -nb.cg) - nb.cq * (x - c);
}

template <int WQ, int CCH, typename TS, bool NB = false, bool FIX = false>
__device__ __forceinline__ void run(float* __restrict__ win, const TS* __restrict__ sp,
                                    const float* __restrict__ gop, const TapPlan& p, const Taps& t, bool inside, int C,
                                    int ss, int os, int Ws, int ax0, int by0, int bh, int l0, int l1, int l2, int l3,
                                    float& gix, float& giy, const NormBwd* nb = nullptr,
                                    const float* __restrict__ g1p = nullptr, const float* __restrict__ gdp = nullptr,
                                    const float* __restrict__ x1p = nullptr, float* __restrict__ d1p = nullptr,
                                    float* __restrict__ fixp = nullptr, unsigned fixmask = 0u) {
  // fixp / fixmask (level backward with the gather form of d/d src): taps of this pixel that the gather kernel did NOT
  // take (bit k of fixmask) are added to d/d src here, with float atomics -- fixp = that tensor at this sample
  constexpr int WP = 4 * WQ;
  using fwd_win::WindowPlan;
  const WindowPlan<WQ> pl = fwd_win::window_plan<WQ>(Ws, ax0, by0, bh);
  float4 v[CCH][WindowPlan<WQ>::ITER];
  float gn[CCH];  // the next chunk's output gradients travel with its window
  float an[CCH], xn[CCH];  // NB: first map's gradient (both parts added) and value
  auto fetch_g = [&](int c0) {
#pragma unroll
    for (int c = 0; c < CCH; ++c) {
      const bool ok = inside && c0 + c < C;
      gn[c] = ok ? gop[(long)(c0 + c) * os] : 0.f;
      if (NB) {
        an[c] = ok ? g1p[(long)(c0 + c) * os] : 0.f;
        if (gdp) an[c] += ok ? gdp[(long)(c0 + c) * os] : 0.f;
        xn[c] = ok ? x1p[(long)(c0 + c) * os] : 0.f;
      }
    }
  };
  const int step = gridDim.y * CCH;
  int c0 = blockIdx.y * CCH;
  if (c0 < C) {
    fetch_g(c0);
    fwd_win::window_load<WQ, CCH, TS>(v, pl, sp, c0, C, ss);
  }
  for (; c0 < C; c0 += step) {
    float g[CCH];
#pragma unroll
    for (int c = 0; c < CCH; ++c) g[c] = gn[c];
    if (NB) {
#pragma unroll
      for (int c = 0; c < CCH; ++c)
        if (inside && c0 + c < C) d1p[(long)(c0 + c) * os] = norm_bwd_apply(*nb, an[c], xn[c], nb->c1);
    }
    fwd_win::window_store<WQ, CCH>(win, v, pl, c0, C);
    if (c0 + step < C) {
      fetch_g(c0 + step);
      fwd_win::window_load<WQ, CCH, TS>(v, pl, sp, c0 + step, C, ss);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CCH; ++c) {
      const float* w = win + c * HMAX * WP;
      const float nw = p.ok[0] ? w[l0] : 0.f, ne = p.ok[1] ? w[l1] : 0.f;
      const float sw = p.ok[2] ? w[l2] : 0.f, se = p.ok[3] ? w[l3] : 0.f;
      float gc = g[c];
      if (NB) {
        const float xw = fmaf(se, p.w[3], fmaf(sw, p.w[2], fmaf(ne, p.w[1], nw * p.w[0])));
        gc = norm_bwd_apply(*nb, gc, xw, nb->c2);
      }
      gix = fmaf(gc, (ne - nw) * t.wy0 + (se - sw) * t.wy1, gix);
      giy = fmaf(gc, (sw - nw) * t.wx0 + (se - ne) * t.wx1, giy);
      if (NB && FIX && fixmask && c0 + c < C) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if ((fixmask >> k) & 1u) atomicAdd(fixp + (long)(c0 + c) * ss + p.o[k], gc * p.w[k]);
      }
    }
    __syncthreads();
  }
}
}  // namespace flow_grad

// add1 / add2 (nullable): further gradients of the same flow tensor that are added on the way out (sample b at
// add1 + b * add1_bs resp. add2 + b * 2 * H * W) -- the level backward folds the gradient the decoder's concatenation
// and the residual sum return for the upsampled flow in here instead of two ATen add passes.
template <int CCH, typename TS = float>
__global__ __launch_bounds__(256) void warp_bwd_flow_kernel(const float* __restrict__ gout,
                                                            const TS* __restrict__ src,
                                                            const float* __restrict__ flow, float* __restrict__ gflow,
                                                            int nimg, int C, int Hs, int Ws, int H, int W, long fbs,
                                                            int pad, int align, int norm,
                                                            const float* __restrict__ add1 = nullptr, long add1_bs = 0,
                                                            const float* __restrict__ add2 = nullptr) {
  using namespace fwd_win;
  __shared__ __attribute__((aligned(16))) float win[CCH * HMAX * 72];
  __shared__ int red[4][4];
  __shared__ int box[4];
  int btx, bty, b;
  if (!af_tile_of_block((W + TX - 1) / TX, (H + TY - 1) / TY, nimg, btx, bty, b)) return;
  const int x = btx * TX + (int)(threadIdx.x & 31), y = bty * TY + (int)(threadIdx.x >> 5);
  const bool inside = x < W && y < H;
  Taps t = no_taps();
  if (inside) {
    const float* fb = flow + (long)b * fbs + (long)y * W + x;
    t = make_taps((float)x, (float)y, fb[0], fb[(long)H * W], H, W, Hs, Ws, pad, align != 0, norm);
  }
  const TapPlan p = plan_taps(t, Hs, Ws);
  const TileBox bb = tile_bbox(t, red, box);
  const int bh = bb.y1 - bb.y0 + 1, ax0 = bb.x0 & ~3, aw = bb.x1 - ax0 + 1;
  const bool empty = bb.x1 < bb.x0;
  const int ss = Hs * Ws, os = H * W;
  const TS* sp = src + (long)b * C * ss;
  const float* gop = gout + (long)b * C * os + (long)y * W + x;
  float gix = 0.f, giy = 0.f;
  if (!empty && (Ws & 3) == 0 && bh <= HMAX && aw <= 72) {
    const int xa = min(max(t.x0, 0), Ws - 1) - ax0, xb = min(max(t.x0 + 1, 0), Ws - 1) - ax0;
    const int ya = min(max(t.y0, 0), Hs - 1) - bb.y0, yb = min(max(t.y0 + 1, 0), Hs - 1) - bb.y0;
    const int cxa = min(max(xa, 0), 71), cxb = min(max(xb, 0), 71);
    const int cya = min(max(ya, 0), HMAX - 1), cyb = min(max(yb, 0), HMAX - 1);
    if (aw <= 48)
      flow_grad::run<12, CCH, TS>(win, sp, gop, p, t, inside, C, ss, os, Ws, ax0, bb.y0, bh, cya * 48 + cxa, cya * 48 + cxb,
                         cyb * 48 + cxa, cyb * 48 + cxb, gix, giy);
    else
      flow_grad::run<18, CCH, TS>(win, sp, gop, p, t, inside, C, ss, os, Ws, ax0, bb.y0, bh, cya * 72 + cxa, cya * 72 + cxb,
                         cyb * 72 + cxa, cyb * 72 + cxb, gix, giy);
  } else if (inside && !empty) {
    for (int c = blockIdx.y; c < C; c += gridDim.y) {  // direct gathers (window too large or unaligned rows)
      const float g = gop[(long)c * os];
      const TS* s = sp + (long)c * ss;
      float a0 = to_f32(s[p.o[0]]), a1 = to_f32(s[p.o[1]]), a2 = to_f32(s[p.o[2]]), a3 = to_f32(s[p.o[3]]);
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      const float nw = p.ok[0] ? a0 : 0.f, ne = p.ok[1] ? a1 : 0.f, sw = p.ok[2] ? a2 : 0.f, se = p.ok[3] ? a3 : 0.f;
      gix = fmaf(g, (ne - nw) * t.wy0 + (se - sw) * t.wy1, gix);
      giy = fmaf(g, (sw - nw) * t.wx0 + (se - ne) * t.wx1, giy);
    }
  }
  if (inside) {
    float* gf = gflow + (long)b * 2 * os + (long)y * W + x;
    float rx = gix * t.dx, ry = giy * t.dy;
    if (blockIdx.y == 0) {
      const long o = (long)y * W + x;
      if (add1) rx += add1[(long)b * add1_bs + o], ry += add1[(long)b * add1_bs + os + o];
      if (add2) rx += add2[(long)b * 2 * os + o], ry += add2[(long)b * 2 * os + os + o];
    }
    if (gridDim.y == 1) {
      gf[0] = rx;
      gf[os] = ry;
    } else {  // channel-split launch: partial sums meet in the pre-zeroed gflow
      atomicAdd(gf, rx);
      atomicAdd(gf + os, ry);
    }
  }
}

// Level backward, the warp's flow gradient with the normalisation's backward folded in (flow_grad::run<.., NB>): reads
// the gradients of the NORMALISED maps (g2n for the warped one, g1n + gdir for the first one), writes d/d x1 and the
// flow gradient (+ add1 + add2).  Coefficients from the partial rows of featnorm's bwd_sum_kernel.
struct LevelBwdArgs {
  const double* rows;
  int nrows, mode;
  const float* stats;
  const float* g1n;
  const float* gdir;  // nullable
  long gdir_bs;
  const float* x1;
  float* d1;
  float* gcoarse;  // non-null: the flow gradient goes straight through the adjoint of the x2 upsample into this PRE-ZEROED
  int up_align;    // [B,2,H/2,W/2] tensor with float atomics (coarse levels: one launch less than up2_bwd_kernel)
  const int* qinfo = nullptr;  // gather form of d/d src (inv_gather): per source pixel, written by gather_kernel
  float* gfix = nullptr;       // d/d src [B,C,H,W]: the pairs the gather kernel did not take are added here
};
template <bool FIX = false>  // FIX: the gather form of d/d src is in front (la.qinfo): add the pairs it left
__device__ __forceinline__ void level_warp_bwd_flow_body(const float* __restrict__ g2n, const float* __restrict__ src,
                                                         const float* __restrict__ flow, float* __restrict__ gflow,
                                                         int nimg, int C, int H, int W, long fbs, int pad, int align,
                                                         int norm, const float* __restrict__ add1, long add1_bs,
                                                         const float* __restrict__ add2, const LevelBwdArgs& la,
                                                         unsigned bx) {
  using namespace fwd_win;
  constexpr int CCH = 2;
  __shared__ __attribute__((aligned(16))) float win[CCH * HMAX * 72];
  __shared__ int red[4][4];
  __shared__ int box[4];
  int btx, bty, b;
  if (!af_tile_of_block((W + TX - 1) / TX, (H + TY - 1) / TY, nimg, btx, bty, b, bx)) return;
  const int x = btx * TX + (int)(threadIdx.x & 31), y = bty * TY + (int)(threadIdx.x >> 5);
  const bool inside = x < W && y < H;
  const int Hs = H, Ws = W;
  const int ss = Hs * Ws, os = H * W;
  const flow_grad::NormBwd nb = flow_grad::norm_bwd_coeffs(la.rows + 4L * la.nrows * b, la.nrows, la.stats + 4 * b,
                                                            (long)C * os, la.mode);
  Taps t = no_taps();
  if (inside) {
    const float* fb = flow + (long)b * fbs + (long)y * W + x;
    t = make_taps((float)x, (float)y, fb[0], fb[(long)H * W], H, W, Hs, Ws, pad, align != 0, norm);
  }
  const TapPlan p = plan_taps(t, Hs, Ws);
  const TileBox bb = tile_bbox(t, red, box);
  const int bh = bb.y1 - bb.y0 + 1, ax0 = bb.x0 & ~3, aw = bb.x1 - ax0 + 1;
  const bool empty = bb.x1 < bb.x0;
  const float* sp = src + (long)b * C * ss;
  const long po = (long)b * C * os + (long)y * W + x;
  const float* gop = g2n + po;
  const float* g1p = la.g1n + po;
  const float* gdp = la.gdir ? la.gdir + (long)b * la.gdir_bs + (long)y * W + x : nullptr;
  const float* x1p = la.x1 + po;
  float* d1p = la.d1 + po;
  float gix = 0.f, giy = 0.f;
  // gather form of d/d src: which of this pixel's taps the gather kernel did not take (they are added here)
  unsigned fixmask = 0u;
  float* fixp = nullptr;
  if (FIX && la.qinfo && inside) {
    const int* qi = la.qinfo + (long)b * ss;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (p.ok[k] && !af_gather_takes(qi[p.o[k]], x, y, 3)) fixmask |= 1u << k;
    fixp = la.gfix + (long)b * C * ss;
  }
  if (!empty && (Ws & 3) == 0 && bh <= HMAX && aw <= 72) {
    const int xa = min(max(t.x0, 0), Ws - 1) - ax0, xb = min(max(t.x0 + 1, 0), Ws - 1) - ax0;
    const int ya = min(max(t.y0, 0), Hs - 1) - bb.y0, yb = min(max(t.y0 + 1, 0), Hs - 1) - bb.y0;
    const int cxa = min(max(xa, 0), 71), cxb = min(max(xb, 0), 71);
    const int cya = min(max(ya, 0), HMAX - 1), cyb = min(max(yb, 0), HMAX - 1);
    if (aw <= 48)
      flow_grad::run<12, CCH, float, true, FIX>(win, sp, gop, p, t, inside, C, ss, os, Ws, ax0, bb.y0, bh, cya * 48 + cxa,
                                                cya * 48 + cxb, cyb * 48 + cxa, cyb * 48 + cxb, gix, giy, &nb, g1p, gdp, x1p, d1p,
                                                fixp, fixmask);
    else
      flow_grad::run<18, CCH, float, true, FIX>(win, sp, gop, p, t, inside, C, ss, os, Ws, ax0, bb.y0, bh, cya * 72 + cxa,
                                                cya * 72 + cxb, cyb * 72 + cxa, cyb * 72 + cxb, gix, giy, &nb, g1p, gdp, x1p, d1p,
                                                fixp, fixmask);
  } else if (inside) {
    for (int c = blockIdx.y; c < C; c += gridDim.y) {  // no tap inside the source, or a window too large: direct gathers
      const float gsum = g1p[(long)c * os] + (gdp ? gdp[(long)c * os] : 0.f);
      d1p[(long)c * os] = flow_grad::norm_bwd_apply(nb, gsum, x1p[(long)c * os], nb.c1);
      if (empty) continue;
      const float* s = sp + (long)c * ss;
      float a0 = s[p.o[0]], a1 = s[p.o[1]], a2 = s[p.o[2]], a3 = s[p.o[3]];
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      const float nw = p.ok[0] ? a0 : 0.f, ne = p.ok[1] ? a1 : 0.f, sw = p.ok[2] ? a2 : 0.f, se = p.ok[3] ? a3 : 0.f;
      const float xw = fmaf(se, p.w[3], fmaf(sw, p.w[2], fmaf(ne, p.w[1], nw * p.w[0])));
      const float g = flow_grad::norm_bwd_apply(nb, gop[(long)c * os], xw, nb.c2);
      gix = fmaf(g, (ne - nw) * t.wy0 + (se - sw) * t.wy1, gix);
      giy = fmaf(g, (sw - nw) * t.wx0 + (se - ne) * t.wx1, giy);
      if (FIX && fixmask) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if ((fixmask >> k) & 1u) atomicAdd(fixp + (long)c * ss + p.o[k], g * p.w[k]);
      }
    }
  }
  if (inside) {
    float* gf = gflow + (long)b * 2 * os + (long)y * W + x;
    float rx = gix * t.dx, ry = giy * t.dy;
    if (blockIdx.y == 0) {
      const long o = (long)y * W + x;
      if (add1) rx += add1[(long)b * add1_bs + o], ry += add1[(long)b * add1_bs + os + o];
      if (add2) rx += add2[(long)b * 2 * os + o], ry += add2[(long)b * 2 * os + os + o];
    }
    if (la.gcoarse) {
      const int Hc = H / 2, Wc = W / 2;
      int xa, xb, ya, yb;
      float wx0, wx1, wy0, wy1;
      up2_source(x, Wc, W, la.up_align != 0, xa, xb, wx0, wx1);
      up2_source(y, Hc, H, la.up_align != 0, ya, yb, wy0, wy1);
      float* gc = la.gcoarse + (long)b * 2 * Hc * Wc;
      const long cs = (long)Hc * Wc;
      rx *= 2.f, ry *= 2.f;  // interpolate(2 f)
      atomicAdd(gc + ya * Wc + xa, wy0 * wx0 * rx), atomicAdd(gc + ya * Wc + xb, wy0 * wx1 * rx);
      atomicAdd(gc + yb * Wc + xa, wy1 * wx0 * rx), atomicAdd(gc + yb * Wc + xb, wy1 * wx1 * rx);
      atomicAdd(gc + cs + ya * Wc + xa, wy0 * wx0 * ry), atomicAdd(gc + cs + ya * Wc + xb, wy0 * wx1 * ry);
      atomicAdd(gc + cs + yb * Wc + xa, wy1 * wx0 * ry), atomicAdd(gc + cs + yb * Wc + xb, wy1 * wx1 * ry);
    } else if (gridDim.y == 1) {
      gf[0] = rx;
      gf[os] = ry;
    } else {
      atomicAdd(gf, rx);
      atomicAdd(gf + os, ry);
    }
  }
}

template <bool FIX = false>
__global__ __launch_bounds__(256) void level_warp_bwd_flow_kernel(const float* __restrict__ g2n, const float* __restrict__ src,
                                                                  const float* __restrict__ flow, float* __restrict__ gflow,
                                                                  int nimg, int C, int H, int W, long fbs, int pad, int align,
                                                                  int norm, const float* __restrict__ add1, long add1_bs,
                                                                  const float* __restrict__ add2, LevelBwdArgs la) {
  level_warp_bwd_flow_body<FIX>(g2n, src, flow, gflow, nimg, C, H, W, fbs, pad, align, norm, add1, add1_bs, add2, la, blockIdx.x);
}

// Adjoint of the x2 bilinear flow upsample of the level forward (up2_source), times the factor 2 of
// interpolate(flow * 2): one thread per coarse cell gathers the fine pixels that read it -- rows 2i-2 .. 2i+3 cover
// either align flag -- in a fixed order (ATen's backward scatters with atomics; this one is reproducible).
__global__ __launch_bounds__(256) void up2_bwd_kernel(const float* __restrict__ gfine, float* __restrict__ gcoarse,
                                                      int planes, int H, int W, int up_align) {
  const int Hc = H / 2, Wc = W / 2;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)planes * Hc * Wc) return;
  const int j = (int)(idx % Wc), i = (int)((idx / Wc) % Hc);
  const long pl = idx / ((long)Wc * Hc);
  const float* g = gfine + pl * H * W;
  float wy[6], wx[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    int a0, a1;
    float l0, l1;
    const int y = 2 * i - 2 + k;
    wy[k] = 0.f;
    if (y >= 0 && y < H) {
      up2_source(y, Hc, H, up_align != 0, a0, a1, l0, l1);
      wy[k] = (a0 == i ? l0 : 0.f) + (a1 == i ? l1 : 0.f);
    }
    const int x = 2 * j - 2 + k;
    wx[k] = 0.f;
    if (x >= 0 && x < W) {
      up2_source(x, Wc, W, up_align != 0, a0, a1, l0, l1);
      wx[k] = (a0 == j ? l0 : 0.f) + (a1 == j ? l1 : 0.f);
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int ky = 0; ky < 6; ++ky) {
    const int y = min(max(2 * i - 2 + ky, 0), H - 1);
    float row = 0.f;
#pragma unroll
    for (int kx = 0; kx < 6; ++kx) {
      const int x = min(max(2 * j - 2 + kx, 0), W - 1);
      row = fmaf(wx[kx], g[(long)y * W + x], row);
    }
    acc = fmaf(wy[ky], row, acc);
  }
  gcoarse[idx] = 2.f * acc;
}

// ------------------------------------------------------------------------------------------------
// d loss / d src of the warp: the 4-tap scatter turned into a gather through an index list in LDS.
//
// One workgroup = an 8 x 32 tile of output pixels.  The taps of its pixels land in a source window whose
// geometry does not depend on the channel.  So, ONCE per tile, the workgroup builds a CSR list
// "window cell -> (pixel, weight) contributors" with INTEGER LDS atomics (ds_add_u32: ~8 cycles per
// wave instruction; the float form ds_add_f32 measured ~200 cycles on gfx950, tools/ubench) and compacts
// the non-empty cells.  Then, per chunk of CCH channels, every thread stages its gout values in LDS and
// each non-empty cell sums its contributors with plain fp32 FMAs and issues ONE global atomic per cell
// and channel -- consecutive lanes on consecutive addresses, no same-address collisions inside an
// instruction (those made the direct 4-taps-per-pixel scatter 5-8x slower on displaced fields), ~2x
// fewer atomics.  The window may be as large as 128 x 64 (a violently divergent field); beyond that the
// tile falls back to direct global atomics.
// ------------------------------------------------------------------------------------------------
namespace lds_scatter {
constexpr int TX = 32, TY = 8, NT = TX * TY;
constexpr int WMAX = 128, HMAX = 64, NCELL = WMAX * HMAX, CCH = 4;
static_assert(CCH == 4, "gt holds one float4 per pixel");

// NB (level backward): gout is the gradient of the NORMALISED warped map; the normalisation's backward is applied while
// the gradients are staged (flow_grad::norm_bwd_apply with x = the saved warped map `x2w`), see level_warp_bwd_flow_kernel.
// SLAB (level backward, two-pass form): instead of adding its window sums into gsrc with float atomics (~0.7 TB/s of added
// bytes for these ragged 10 x 34 windows, tools/ubench/atomic_shape.hip; plain stores of the same shape run 4x faster) the
// tile STORES them, dense, into its own slab -- slab[(tile * C + c) * cap + r * wq + cx], wq = the window width rounded up to
// 4 -- with the window rectangle in meta[tile]; slab_gather_kernel then sums, per 8 x 32 block of gsrc, the slabs whose
// rectangles meet it, in tile order.  A window that does not fit `cap` cells falls back to the atomics (gsrc arrives
// zero-filled) and raises ovf[b], which makes the gather ADD to gsrc for that sample instead of overwriting it.
struct SlabArgs {
  float* base;
  int4* meta;
  int* ovf;
  int cap;
};

template <bool NB, bool SLAB = false>
__device__ __forceinline__ void warp_bwd_src_body(const float* __restrict__ gout,
                                                  const float* __restrict__ flow, float* __restrict__ gsrc,
                                                  int nimg, int C, int Hs, int Ws, int H, int W, long fbs,
                                                  int pad, int align, int norm,
                                                  const float* __restrict__ x2w,
                                                  const double* __restrict__ rows, int nrows,
                                                  const float* __restrict__ stats, int mode, unsigned bx,
                                                  SlabArgs sl = SlabArgs{}) {
  // cell c lives at halfword cell_ptr[c + (c >> 5)]: the scan walks a lane-private run of consecutive cells,
  // and the +1-per-32 skew keeps 64 lanes with a stride that is a multiple of 32 off a common bank.
  // 16-bit cells (a workgroup has at most 4 * 256 = 1024 list entries) halve the array: 17 KB instead of
  // 34 KB lets 4 workgroups share a CU instead of 2 -- the kernel is latency-bound, occupancy is what hides it.
  // Counting / filling use 32-bit LDS atomics on the containing word (1 << 16 for the upper cell; a cell
  // never exceeds 1024, so nothing carries into its neighbour); the scan uses 16-bit loads / stores.
  __shared__ __attribute__((aligned(4))) unsigned short cell_ptr[NCELL + NCELL / 32 + 2];  // counts, then fill pointers
  __shared__ unsigned short ne_cell[4 * NT];    // compacted non-empty cells, packed (row << 7 | col), row-major
  __shared__ unsigned short ne_beg[4 * NT + 1]; // first entry of each non-empty cell; [n] = total
  __shared__ float2 entry[4 * NT];              // (pixel index as float bits, weight)
  __shared__ float4 gt[2][NT];                  // staged output gradients (4 channels per pixel), double-buffered
  __shared__ int red[4][4];
  __shared__ int box[4];
  __shared__ int wave_tot[2][NT / 64];
  const int lx = threadIdx.x % TX, ly = threadIdx.x / TX;
  int btx, bty, b;
  if (!af_tile_of_block((W + TX - 1) / TX, (H + TY - 1) / TY, nimg, btx, bty, b, bx)) return;  // whole workgroup
  const int x = btx * TX + lx, y = bty * TY + ly;
  const bool inside = x < W && y < H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  Taps t = no_taps();
  if (inside) {
    const float* fb = flow + (long)b * fbs + (long)y * W + x;
    t = make_taps((float)x, (float)y, fb[0], fb[(long)H * W], H, W, Hs, Ws, pad, align != 0, norm);
  }
  const TileBox bb = tile_bbox(t, red, box);
  const int bx0 = bb.x0, by0 = bb.y0;
  const int bw = bb.x1 - bx0 + 1, bh = bb.y1 - by0 + 1;
  const bool empty = bb.x1 < bx0;
  const int wq = (bw + 3) & ~3;  // SLAB: row pitch of the tile's slab
  const bool priv = bw <= WMAX && bh <= HMAX;
  const bool fits = SLAB && !empty && priv && bh * wq <= sl.cap;
  const long tile_g = ((long)b * ((H + TY - 1) / TY) + bty) * ((W + TX - 1) / TX) + btx;
  if (SLAB && blockIdx.y == 0 && threadIdx.x == 0) {
    sl.meta[tile_g] = empty ? make_int4(0, 0, 0, 0) : (fits ? make_int4(bx0, by0, bw, bh) : make_int4(0, 0, -1, 0));
    if (!empty && !fits) atomicOr(sl.ovf + b, 1);
  }
  if (empty) return;  // no pixel of the tile samples inside the source: nothing to add (uniform)

  const float wgt[4] = {t.wx0 * t.wy0, t.wx1 * t.wy0, t.wx0 * t.wy1, t.wx1 * t.wy1};
  const bool ok[4] = {t.vx0 && t.vy0, t.vx1 && t.vy0, t.vx0 && t.vy1, t.vx1 && t.vy1};
  const long o00 = (long)t.y0 * Ws + t.x0;
  const long ss = (long)Hs * Ws, os = (long)H * W;
  float* gp = gsrc + (long)b * C * ss;
  const float* gop = gout + (long)b * C * os + (long)y * W + x;
  const float* xwp = NB ? x2w + (long)b * C * os + (long)y * W + x : nullptr;
  flow_grad::NormBwd nb;
  if (NB) nb = flow_grad::norm_bwd_coeffs(rows + 4L * nrows * b, nrows, stats + 4 * b, (long)C * os, mode);

  if (!priv) {
    if (inside)
      for (int c = blockIdx.y; c < C; c += gridDim.y) {
        float g = gop[c * os];
        if (NB) g = flow_grad::norm_bwd_apply(nb, g, xwp[c * os], nb.c2);
        float* d = gp + c * ss + o00;
        if (ok[0]) atomicAdd(d, g * wgt[0]);
        if (ok[1]) atomicAdd(d + 1, g * wgt[1]);
        if (ok[2]) atomicAdd(d + Ws, g * wgt[2]);
        if (ok[3]) atomicAdd(d + Ws + 1, g * wgt[3]);
      }
    return;
  }
  // window pitch: the box width rounded up to 32; only bh * wp cells are zeroed / scanned
  const int wp = (bw + 31) & ~31;
  const int ncell = bh * wp;
  const int per = (ncell + NT - 1) / NT;  // consecutive cells owned by one thread in the scan
  auto slot_of = [](int c) { return c + (c >> 5); };
  auto cell_add = [&](int c) {  // returns the cell's value before the increment
    const int sl = slot_of(c);
    const unsigned old = atomicAdd(reinterpret_cast<unsigned*>(cell_ptr) + (sl >> 1), (sl & 1) ? 0x10000u : 1u);
    return (int)((sl & 1) ? old >> 16 : old & 0xffffu);
  };
  {
    unsigned* cw = reinterpret_cast<unsigned*>(cell_ptr);
    const int nhalf = ncell + (ncell >> 5) + 1;
    for (int i = threadIdx.x; i < (nhalf + 1) / 2; i += NT) cw[i] = 0u;
  }
  __syncthreads();
  const int rx = t.x0 - bx0, ry = t.y0 - by0;
  const int cell[4] = {ry * wp + rx, ry * wp + rx + 1, (ry + 1) * wp + rx, (ry + 1) * wp + rx + 1};
  // 1. count contributors per cell
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (ok[k]) cell_add(cell[k]);
  __syncthreads();
  // 2. scan: entries and non-empty cells are numbered in cell order
  const int c_lo = min((int)threadIdx.x * per, ncell), c_hi = min(c_lo + per, ncell);
  int run = 0, nz = 0;
  for (int ci = c_lo; ci < c_hi; ++ci) {
    const int c = cell_ptr[slot_of(ci)];
    run += c;
    nz += c != 0;
  }
  int incl = run, incz = nz;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64), z = __shfl_up(incz, off, 64);
    if (lane >= off) incl += v, incz += z;
  }
  if (lane == 63) wave_tot[0][wave] = incl, wave_tot[1][wave] = incz;
  __syncthreads();
  int base = incl - run, basez = incz - nz, total = 0, totalz = 0;
  for (int w = 0; w < NT / 64; ++w) {
    if (w < wave) base += wave_tot[0][w], basez += wave_tot[1][w];
    total += wave_tot[0][w], totalz += wave_tot[1][w];
  }
  {
    int r = c_lo / wp, cc = c_lo - r * wp;  // running (row, col) of the cell
    for (int ci = c_lo; ci < c_hi; ++ci) {
      const int c = cell_ptr[slot_of(ci)];
      if (c != 0) {
        ne_cell[basez] = (unsigned short)((r << 7) | cc);
        ne_beg[basez] = (unsigned short)base;
        ++basez;
      }
      cell_ptr[slot_of(ci)] = (unsigned short)base;  // becomes the fill pointer
      base += c;
      if (++cc == wp) cc = 0, ++r;
    }
  }
  if (threadIdx.x == 0) ne_beg[totalz] = (unsigned short)total;
  __syncthreads();
  // 3. fill the lists
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (ok[k]) {
      const int slot = cell_add(cell[k]);
      entry[slot] = make_float2(__int_as_float((int)threadIdx.x), wgt[k]);
    }
  __syncthreads();
  // 4. per channel chunk: stage gout pixel-major (one ds_write_b128 per pixel, one ds_read_b128 per list
  //    entry brings its 4 channels), gather per non-empty cell, one global atomic per cell and channel
  int buf = 0;
  auto fetch = [&](int c0, float (&v)[CCH]) {
    float xv[CCH];
#pragma unroll
    for (int c = 0; c < CCH; ++c) {
      const bool ok = c0 + c < C && inside;
      v[c] = ok ? gop[(c0 + c) * os] : 0.f;
      if (NB) xv[c] = ok ? xwp[(c0 + c) * os] : 0.f;
    }
    if (NB) {
#pragma unroll
      for (int c = 0; c < CCH; ++c) v[c] = (c0 + c < C && inside) ? flow_grad::norm_bwd_apply(nb, v[c], xv[c], nb.c2) : 0.f;
    }
  };
  float nv[CCH];  // the next chunk's gradients travel while the current chunk is gathered
  fetch(blockIdx.y * CCH, nv);
  for (int c0 = blockIdx.y * CCH; c0 < C; c0 += gridDim.y * CCH, buf ^= 1) {
    gt[buf][threadIdx.x] = make_float4(nv[0], nv[1], nv[2], nv[3]);
    fetch(c0 + gridDim.y * CCH, nv);
    __syncthreads();
    if (SLAB && fits) {  // dense sweep of the window: every cell (zeros included) into the tile's slab, plain stores
      float* sb = sl.base + (tile_g * C + c0) * (long)sl.cap;
      for (int i = threadIdx.x; i < bh * wq; i += NT) {
        const int r = i / wq, cx = i - r * wq;
        float acc[CCH] = {0.f, 0.f, 0.f, 0.f};
        if (cx < bw) {
          const int ci = r * wp + cx;
          const int end = cell_ptr[slot_of(ci)], beg = ci > 0 ? cell_ptr[slot_of(ci - 1)] : 0;  // fill pointers: ends
          for (int e = beg; e < end; e += 4) {
            float2 en[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              en[u] = entry[min(e + u, end - 1)];
              if (e + u >= end) en[u].y = 0.f;
            }
            float4 gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) gv[u] = gt[buf][__float_as_int(en[u].x)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              acc[0] = fmaf(gv[u].x, en[u].y, acc[0]);
              acc[1] = fmaf(gv[u].y, en[u].y, acc[1]);
              acc[2] = fmaf(gv[u].z, en[u].y, acc[2]);
              acc[3] = fmaf(gv[u].w, en[u].y, acc[3]);
            }
          }
        }
#pragma unroll
        for (int c = 0; c < CCH; ++c)
          if (c0 + c < C) sb[(long)c * sl.cap + i] = acc[c];
      }
      continue;
    }
    for (int i = threadIdx.x; i < totalz; i += NT) {
      const int ci = ne_cell[i];
      const int beg = ne_beg[i], end = ne_beg[i + 1];
      float acc[CCH];
#pragma unroll
      for (int c = 0; c < CCH; ++c) acc[c] = 0.f;
      // four list entries per trip (most cells have <= 4 contributors): the entry reads go out together,
      // then the four gradient reads -- two dependent LDS round trips per cell instead of two per entry.
      // Slots past the end re-read the last entry with weight 0 (same sum, same order).
      for (int e = beg; e < end; e += 4) {
        float2 en[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          en[u] = entry[min(e + u, end - 1)];
          if (e + u >= end) en[u].y = 0.f;
        }
        float4 gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) gv[u] = gt[buf][__float_as_int(en[u].x)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[0] = fmaf(gv[u].x, en[u].y, acc[0]);
          acc[1] = fmaf(gv[u].y, en[u].y, acc[1]);
          acc[2] = fmaf(gv[u].z, en[u].y, acc[2]);
          acc[3] = fmaf(gv[u].w, en[u].y, acc[3]);
        }
      }
      float* d = gp + (long)c0 * ss + (long)(by0 + (ci >> 7)) * Ws + bx0 + (ci & 127);
#pragma unroll
      for (int c = 0; c < CCH; ++c)
        if (c0 + c < C) atomicAdd(d + c * ss, acc[c]);
    }
    // gt is double-buffered: the next chunk stages into the other buffer; its barrier orders this chunk's
    // reads before this buffer is overwritten two chunks later
  }
}
template <bool NB = false, bool SLAB = false>
__global__ __launch_bounds__(NT) void warp_bwd_src_kernel(const float* __restrict__ gout,
                                                          const float* __restrict__ flow, float* __restrict__ gsrc,
                                                          int nimg, int C, int Hs, int Ws, int H, int W, long fbs,
                                                          int pad, int align, int norm,
                                                          const float* __restrict__ x2w = nullptr,
                                                          const double* __restrict__ rows = nullptr, int nrows = 0,
                                                          const float* __restrict__ stats = nullptr, int mode = 0,
                                                          SlabArgs sl = SlabArgs{}) {
  warp_bwd_src_body<NB, SLAB>(gout, flow, gsrc, nimg, C, Hs, Ws, H, W, fbs, pad, align, norm, x2w, rows, nrows, stats, mode,
                              blockIdx.x, sl);
}

// Second pass of the SLAB form: one workgroup per 8 x 32 block of gsrc (x a channel split): the rectangles of the sample's
// T tiles are scanned (meta in LDS), and every pixel sums, in TILE ORDER, the slab cells of the tiles whose window
// contains it -- plain loads, one plain store per pixel and channel; no atomics, no zero-fill of gsrc needed.
constexpr int MAXT = 1024;
__global__ __launch_bounds__(NT) void slab_gather_kernel(const float* __restrict__ slab, const int4* __restrict__ meta,
                                                         const int* __restrict__ ovf, float* __restrict__ gsrc, int nimg,
                                                         int C, int Hs, int Ws, int T, int cap) {
  __shared__ int4 m[MAXT];
  __shared__ unsigned char hit[MAXT];
  __shared__ unsigned short list[MAXT];
  __shared__ int nhit;
  int bx, by, b;
  if (!af_tile_of_block((Ws + TX - 1) / TX, (Hs + TY - 1) / TY, nimg, bx, by, b)) return;
  const int sx0 = bx * TX, sy0 = by * TY;
  for (int t = threadIdx.x; t < T; t += NT) {
    const int4 mt = meta[(long)b * T + t];
    m[t] = mt;
    hit[t] = mt.z > 0 && mt.x < sx0 + TX && mt.x + mt.z > sx0 && mt.y < sy0 + TY && mt.y + mt.w > sy0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // the tiles whose window meets this block, compacted in TILE ORDER
    int n = 0;
    for (int t = 0; t < T; ++t)
      if (hit[t]) list[n++] = (unsigned short)t;
    nhit = n;
  }
  __syncthreads();
  const int n = nhit;
  const int x = sx0 + (int)(threadIdx.x % TX), y = sy0 + (int)(threadIdx.x / TX);
  const bool inside = x < Ws && y < Hs;
  const bool add = ovf[b] != 0;  // some tile of this sample went through the atomics: keep what they left
  const long ss = (long)Hs * Ws;
  float* gp = gsrc + (long)b * C * ss + (long)y * Ws + x;
  constexpr int CG = 8, EU = 4;  // 8 channels x 4 tiles = 32 independent loads in flight per thread
  for (int c0 = blockIdx.y * CG; c0 < C; c0 += gridDim.y * CG) {
    float acc[CG];
#pragma unroll
    for (int k = 0; k < CG; ++k) acc[k] = (add && inside && c0 + k < C) ? gp[(long)(c0 + k) * ss] : 0.f;
    for (int e0 = 0; e0 < n; e0 += EU) {
      float v[EU][CG];
      bool in[EU];
#pragma unroll
      for (int u = 0; u < EU; ++u) {
        const int t = list[min(e0 + u, n - 1)];
        const int4 mt = m[t];
        const int rx = x - mt.x, ry = y - mt.y;
        in[u] = e0 + u < n && inside && rx >= 0 && rx < mt.z && ry >= 0 && ry < mt.w;
        // every load is issued (a pixel outside the rectangle re-reads the slab's first cell, weight 0)
        const float* sp = slab + (((long)b * T + t) * C + c0) * (long)cap + (in[u] ? ry * ((mt.z + 3) & ~3) + rx : 0);
#pragma unroll
        for (int k = 0; k < CG; ++k) v[u][k] = sp[(long)min(k, C - 1 - c0) * cap];
      }
#pragma unroll
      for (int u = 0; u < EU; ++u)  // tile order
#pragma unroll
        for (int k = 0; k < CG; ++k) acc[k] += in[u] ? v[u][k] : 0.f;
    }
    if (inside) {
#pragma unroll
      for (int k = 0; k < CG; ++k)
        if (c0 + k < C) gp[(long)(c0 + k) * ss] = acc[k];
    }
  }
}
}  // namespace lds_scatter

// The two independent gradient kernels of the warp as ONE launch.  The roles alternate in runs of 8 workgroups along
// blockIdx.x (bit 3 picks the role; the low 3 bits -- the XCD a workgroup lands on -- and the tile index are those of the
// plain kernels), so the two workgroups of a tile are dispatched next to each other on the SAME XCD: the second one finds
// the tile's g2n in that L2, and the atomics-bound role runs beside the gather-bound one instead of after it (roles in
// blockIdx.z were dispatched one after the other: the launch took the SUM of the two kernels' times).
__global__ __launch_bounds__(256) void level_warp_bwd_both_kernel(const float* __restrict__ g2n, const float* __restrict__ src,
                                                                  const float* __restrict__ x2w,
                                                                  const float* __restrict__ flow, float* __restrict__ gsrc,
                                                                  float* __restrict__ gflow, int nimg, int C, int H, int W,
                                                                  long fbs, int pad, int align, int norm,
                                                                  const float* __restrict__ add1, long add1_bs,
                                                                  const float* __restrict__ add2, LevelBwdArgs la) {
  const unsigned bx = (blockIdx.x & 7u) | ((blockIdx.x >> 4) << 3);
  if ((blockIdx.x & 8u) == 0)
    lds_scatter::warp_bwd_src_body<true>(g2n, flow, gsrc, nimg, C, H, W, H, W, fbs, pad, align, norm, x2w, la.rows, la.nrows,
                                         la.stats, la.mode, bx);
  else
    level_warp_bwd_flow_body<false>(g2n, src, flow, gflow, nimg, C, H, W, fbs, pad, align, norm, add1, add1_bs, add2, la, bx);
}

// ------------------------------------------------------------------------------------------------
// d loss / d src of the warp as a GATHER over the inverse-flow window (level backward, fine level; round 3).
//
// The scatter form above ends in ~10 M float atomics (180-260 G/s on MI355X whatever their shape, tools/ubench/
// atomic_shape.hip): >= 50 us at B16 C32 96x160.  For the flows a pyramid level sees (the x2 upsample of the coarser
// estimate: smooth except at motion boundaries) the targets p whose taps hit a source pixel q lie within a few pixels of
// q - flow(q).  So every source pixel q scans the (2M+1)^2 targets p around c(q) = q - round(flow(q)) against the taps of
// those targets (computed once per workgroup, staged in LDS), keeps the (p, weight) pairs whose tap IS q, and sums
// w * g[c, p] per channel out of an LDS-staged tile of gradients: plain stores, every element of gsrc written (no
// zero-fill), a fixed summation order.
// Exactness for ANY flow: a pair (p, q) outside q's window -- or a q whose list overflows, or a tile whose window box does
// not fit LDS -- is added by the flow-gradient role instead, which visits every p with its four taps anyway: it reads
// qinfo[q] (window centre + flag, written here) and issues the float atomic iff THIS kernel did not take the pair (same
// predicate on both sides: the pairs are partitioned, none is lost or counted twice).  On the flows of a training step
// those atomics are rare.
// ------------------------------------------------------------------------------------------------
namespace inv_gather {
constexpr int TX = 32, TY = 8, NT = 256, M = 3, RW = 48, RH = 16, RMAX = RW * RH, K = 8, CCH = 4;
constexpr int QFLAG = 1 << 30;
__device__ __forceinline__ int pack_q(int cx, int cy, bool flag) {  // cx, cy in [-4096, 4095] after the clamp below
  return ((cx + 4096) & 0x1fff) | (((cy + 4096) & 0x1fff) << 13) | (flag ? QFLAG : 0);
}
__device__ __forceinline__ void unpack_q(int v, int& cx, int& cy, bool& flag) {
  cx = (v & 0x1fff) - 4096, cy = ((v >> 13) & 0x1fff) - 4096, flag = (v & QFLAG) != 0;
}
// does the gather kernel take the pair (target p, source q)?  (the ONE predicate both kernels use)
__device__ __forceinline__ bool taken(int qinfo, int px, int py) {
  int cx, cy;
  bool flag;
  unpack_q(qinfo, cx, cy, flag);
  return !flag && abs(px - cx) <= M && abs(py - cy) <= M;
}

__global__ __launch_bounds__(NT) void gather_kernel(const float* __restrict__ g2n, const float* __restrict__ x2w,
                                                    const float* __restrict__ flow, float* __restrict__ gsrc,
                                                    int* __restrict__ qinfo, int nimg, int C, int H, int W, long fbs, int pad,
                                                    int align, int norm, const double* __restrict__ rows, int nrows,
                                                    const float* __restrict__ stats, int mode) {
  __shared__ int tinfo[RMAX];                                   // x0 + 1 | (y0 + 1) << 13 | valid bits << 26
  __shared__ __attribute__((aligned(16))) float4 twts[RMAX];    // wx0, wx1, wy0, wy1
  __shared__ float gbuf[CCH][RMAX];
  __shared__ int red[4][4];
  __shared__ int box[4];
  int btx, bty, b;
  if (!af_tile_of_block((W + TX - 1) / TX, (H + TY - 1) / TY, nimg, btx, bty, b)) return;
  const int x = btx * TX + (int)(threadIdx.x & 31), y = bty * TY + (int)(threadIdx.x >> 5);
  const bool inside = x < W && y < H;
  const long os = (long)H * W;
  const float* fb = flow + (long)b * fbs;
  // window centre of this source pixel
  int cx = 0, cy = 0;
  bool bad = !inside;
  if (inside) {
    const float u = fb[(long)y * W + x], v = fb[os + (long)y * W + x];
    if (fabsf(u) < 4000.f && fabsf(v) < 4000.f)
      cx = x - (int)rintf(u), cy = y - (int)rintf(v);
    else
      bad = true;  // NaN / absurd flow: leave the pixel to the atomics
  }
  // box of the windows (clamped to the image) over the tile
  int lo_x = bad ? 0x7fffffff : max(cx - M, 0), hi_x = bad ? -0x7fffffff : min(cx + M, W - 1);
  int lo_y = bad ? 0x7fffffff : max(cy - M, 0), hi_y = bad ? -0x7fffffff : min(cy + M, H - 1);
  if (!bad && (lo_x > hi_x || lo_y > hi_y)) lo_x = 0x7fffffff, hi_x = -0x7fffffff, lo_y = 0x7fffffff, hi_y = -0x7fffffff;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo_x = min(lo_x, __shfl_xor(lo_x, off, 64)), lo_y = min(lo_y, __shfl_xor(lo_y, off, 64));
    hi_x = max(hi_x, __shfl_xor(hi_x, off, 64)), hi_y = max(hi_y, __shfl_xor(hi_y, off, 64));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[0][wave] = lo_x, red[1][wave] = lo_y, red[2][wave] = hi_x, red[3][wave] = hi_y;
  __syncthreads();
  if (threadIdx.x == 0) {
    int a = red[0][0], bb = red[1][0], c = red[2][0], d = red[3][0];
    for (int w = 1; w < 4; ++w) a = min(a, red[0][w]), bb = min(bb, red[1][w]), c = max(c, red[2][w]), d = max(d, red[3][w]);
    box[0] = a, box[1] = bb, box[2] = c, box[3] = d;
  }
  __syncthreads();
  const int rx0 = box[0], ry0 = box[1];
  const int rw = box[2] - rx0 + 1, rh = box[3] - ry0 + 1;
  const bool none = box[2] < rx0;                          // no pixel of the tile has a window inside the image
  const bool oversize = !none && (rw > RW || rh > RH);     // workgroup-uniform
  const int nr = (none || oversize) ? 0 : rw * rh;
  // taps of the targets in the box, once per workgroup
  for (int i = threadIdx.x; i < nr; i += NT) {
    const int r = i / rw, c = i - r * rw;
    const int px = rx0 + c, py = ry0 + r;
    const Taps t = make_taps((float)px, (float)py, fb[(long)py * W + px], fb[os + (long)py * W + px], H, W, H, W, pad,
                             align != 0, norm);
    const int vb = (t.vx0 ? 1 : 0) | (t.vx1 ? 2 : 0) | (t.vy0 ? 4 : 0) | (t.vy1 ? 8 : 0);
    tinfo[i] = ((t.x0 + 1) & 0x1fff) | (((t.y0 + 1) & 0x1fff) << 13) | (vb << 26);
    twts[i] = make_float4(t.wx0, t.wx1, t.wy0, t.wy1);
  }
  __syncthreads();
  // this pixel's contributors
  int lidx[K];
  float lw[K];
  int n = 0;
  bool flag = bad || oversize;
#pragma unroll
  for (int k = 0; k < K; ++k) lidx[k] = 0, lw[k] = 0.f;
  if (!flag && !none) {
    for (int dy = -M; dy <= M; ++dy) {
      const int py = cy + dy;
      if (py < 0 || py >= H) continue;
      for (int dx = -M; dx <= M; ++dx) {
        const int px = cx + dx;
        if (px < 0 || px >= W) continue;
        const int i = (py - ry0) * rw + (px - rx0);
        const int ti = tinfo[i];
        const int ddx = x - ((ti & 0x1fff) - 1), ddy = y - (((ti >> 13) & 0x1fff) - 1);
        if ((unsigned)ddx > 1u || (unsigned)ddy > 1u) continue;
        const int vb = ti >> 26;
        if (!((vb >> ddx) & 1) || !((vb >> (2 + ddy)) & 1)) continue;
        const float4 w4 = twts[i];
        const float w = (ddx ? w4.y : w4.x) * (ddy ? w4.w : w4.z);
        if (n < K) {
#pragma unroll
          for (int k = 0; k < K; ++k)
            if (k == n) lidx[k] = i, lw[k] = w;
        }
        ++n;
      }
    }
    if (n > K) flag = true, n = 0;  // (convergent flow: more contributors than the list holds)
  }
  if (flag) n = 0;
  if (inside && blockIdx.y == 0) qinfo[(long)b * os + (long)y * W + x] = pack_q(max(min(cx, 4095), -4096), max(min(cy, 4095), -4096), flag);
  // (a clamped centre can only belong to a pixel whose window lies outside the image: nothing is taken for it either way)
  const flow_grad::NormBwd nb = flow_grad::norm_bwd_coeffs(rows + 4L * nrows * b, nrows, stats + 4 * b, (long)C * os, mode);
  const float* gp = g2n + (long)b * C * os;
  const float* xp = x2w + (long)b * C * os;
  float* op = gsrc + (long)b * C * os + (long)y * W + x;
  constexpr int ITER = RMAX / NT;  // 3
  const int step = gridDim.y * CCH;
  int c0 = blockIdx.y * CCH;
  float gv[CCH][ITER], xv[CCH][ITER];
  auto fetch = [&](int cc) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int i = threadIdx.x + it * NT;
      const int r = i / max(rw, 1), c = i - r * max(rw, 1);
      const long o = i < nr ? (long)(ry0 + r) * W + rx0 + c : 0;
#pragma unroll
      for (int k = 0; k < CCH; ++k) {
        const long co = (long)min(cc + k, C - 1) * os + o;
        gv[k][it] = gp[co], xv[k][it] = xp[co];
      }
    }
  };
  if (c0 < C) fetch(c0);
  for (; c0 < C; c0 += step) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int i = threadIdx.x + it * NT;
      if (i < nr) {
#pragma unroll
        for (int k = 0; k < CCH; ++k) gbuf[k][i] = flow_grad::norm_bwd_apply(nb, gv[k][it], xv[k][it], nb.c2);
      }
    }
    if (c0 + step < C) fetch(c0 + step);
    __syncthreads();
    if (inside) {
#pragma unroll
      for (int k = 0; k < CCH; ++k) {
        if (c0 + k >= C) break;
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < K; ++e)
          if (e < n) sum = fmaf(lw[e], gbuf[k][lidx[e]], sum);
        op[(long)(c0 + k) * os] = sum;
      }
    }
    __syncthreads();
  }
}
}  // namespace inv_gather

// ------------------------------------------------------------------------------------------------
// Forward splat of the 4 bilinear weights of every pixel's target position (compute_range_map /
// get_corresponding_map).  One workgroup = an 8 x 32 tile of source pixels; their targets fall into a
// small window that is accumulated in LDS with INTEGER atomics on 2^-22 fixed-point weights
// (ds_add_u32 is ~25x faster than ds_add_f32 on gfx950, and integer sums do not depend on the order),
// then every touched cell is flushed with one global float atomic.  The direct form -- four global
// atomics per pixel, with same-address collisions inside a wave wherever the flow compresses -- took
// 128 us for 8 x 384 x 640.  Weight quantisation error <= 1.2e-7 per tap.  Windows larger than
// 128 x 64 fall back to direct atomics.
struct SplatTaps {
  int xi[4], yi[4];
  float w[4];
  bool ok[4];
};
__device__ __forceinline__ SplatTaps splat_taps(float cx, float cy, int H, int W, int variant) {
  SplatTaps t;
  const float fx = floorf(cx), fy = floorf(cy);
  if ((variant & 1) == 0) {
    const float ox = cx - fx, oy = cy - fy;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int di = k >> 1, dj = k & 1;
      const float yi = fy + di, xj = fx + dj;
      t.ok[k] = yi >= 0.f && yi < (float)H && xj >= 0.f && xj < (float)W;
      t.w[k] = (di ? oy : 1.f - oy) * (dj ? ox : 1.f - ox);
      t.yi[k] = t.ok[k] ? (int)yi : 0;
      t.xi[k] = t.ok[k] ? (int)xj : 0;
    }
  } else {
    const float xw = (float)(W - 1), yh = (float)(H - 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int di = k >> 1, dj = k & 1;
      const float yr = fy + di, xr = fx + dj;
      const float yc = fminf(fmaxf(yr, 0.f), yh), xc = fminf(fmaxf(xr, 0.f), xw);
      t.ok[k] = yc == yr && xc == xr;
      t.w[k] = (1.f - fabsf(cx - xc)) * (1.f - fabsf(cy - yc));
      t.yi[k] = t.ok[k] ? (int)yc : 0;
      t.xi[k] = t.ok[k] ? (int)xc : 0;
    }
  }
  return t;
}

// SM: the same launch also takes the edge-aware smoothness partial sums of its tile's pixels (smooth_fwd_kernel's
// arithmetic, smooth_dev.hpp) -- UFlowLoss needs the range map AND the smoothness term of the same level-2 flows
// (losses/uflow_loss.py:43,62-102): one launch instead of two at a size where a launch costs more than either.
template <bool SM = false>
__global__ __launch_bounds__(256) void splat_kernel(const float* __restrict__ flow, float* __restrict__ out,
                                                    int nimg, int H, int W, long fbs, int variant,
                                                    SmoothArgs sa = SmoothArgs{}, float* __restrict__ sums = nullptr,
                                                    int nrows = 0) {
  constexpr int WMAX = 128, HMAX = 64;
  constexpr float FIX = 4194304.f;  // 2^22
  __shared__ unsigned cellv[WMAX * HMAX];
  __shared__ int red[4][4];
  __shared__ int box[4];
  __shared__ float sred[2 * 4];
  int btx, bty, b;
  if (!af_tile_of_block((W + 31) / 32, (H + 7) / 8, nimg, btx, bty, b)) {
    if (SM && threadIdx.x == 0) af_store_partial(sums, nrows, 0.f, 0.f, 0.f);  // padding workgroup: its row must be defined
    return;
  }
  const int x = btx * 32 + (int)(threadIdx.x & 31), y = bty * 8 + (int)(threadIdx.x >> 5);
  const bool inside = x < W && y < H;
  if (SM) {
    float part[2] = {0.f, 0.f};
    if (inside) smooth_pixel<3>(sa, sa.img + (long)b * 3 * H * W, sa.flow + (long)b * sa.fbs, y, x, part[0], part[1]);
    af_block_sum<2>(part, sred);
    if (threadIdx.x == 0) af_store_partial(sums, nrows, part[0], part[1], 0.f);
  }
  SplatTaps t;
#pragma unroll
  for (int k = 0; k < 4; ++k) t.ok[k] = false, t.xi[k] = t.yi[k] = 0, t.w[k] = 0.f;
  if (inside) {
    const float* fb = flow + (long)b * fbs + (long)y * W + x;
    const bool abs_in = (variant & ARFLOW_COORDS_ABS) != 0;
    const float cx = abs_in ? fb[0] : (float)x + fb[0], cy = abs_in ? fb[(long)H * W] : (float)y + fb[(long)H * W];
    t = splat_taps(cx, cy, H, W, variant);
  }
  // bounding box of the valid targets
  int lo_x = 0x7fffffff, lo_y = 0x7fffffff, hi_x = -0x7fffffff, hi_y = -0x7fffffff;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (t.ok[k]) {
      lo_x = min(lo_x, t.xi[k]), hi_x = max(hi_x, t.xi[k]);
      lo_y = min(lo_y, t.yi[k]), hi_y = max(hi_y, t.yi[k]);
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo_x = min(lo_x, __shfl_xor(lo_x, off, 64));
    lo_y = min(lo_y, __shfl_xor(lo_y, off, 64));
    hi_x = max(hi_x, __shfl_xor(hi_x, off, 64));
    hi_y = max(hi_y, __shfl_xor(hi_y, off, 64));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[0][wave] = lo_x, red[1][wave] = lo_y, red[2][wave] = hi_x, red[3][wave] = hi_y;
  __syncthreads();
  if (threadIdx.x == 0) {
    int a = red[0][0], bb = red[1][0], c = red[2][0], d = red[3][0];
    for (int w = 1; w < 4; ++w) a = min(a, red[0][w]), bb = min(bb, red[1][w]), c = max(c, red[2][w]), d = max(d, red[3][w]);
    box[0] = a, box[1] = bb, box[2] = c, box[3] = d;
  }
  __syncthreads();
  const int bx0 = box[0], by0 = box[1], bw = box[2] - bx0 + 1, bh = box[3] - by0 + 1;
  if (box[2] < bx0) return;  // nothing lands inside the image
  float* ob = out + (long)b * H * W;
  if (bw > WMAX || bh > HMAX) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (t.ok[k]) atomicAdd(ob + (long)t.yi[k] * W + t.xi[k], t.w[k]);
    return;
  }
  const int wp = (bw + 31) & ~31;
  for (int i = threadIdx.x; i < bh * wp; i += 256) cellv[i] = 0u;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (t.ok[k]) atomicAdd(&cellv[(t.yi[k] - by0) * wp + (t.xi[k] - bx0)], (unsigned)__float2int_rn(t.w[k] * FIX));
  __syncthreads();
  for (int r = wave; r < bh; r += 4)
    for (int c = lane; c < bw; c += 64) {
      const unsigned v = cellv[r * wp + c];
      if (v) atomicAdd(ob + (long)(by0 + r) * W + bx0 + c, (float)v * (1.f / FIX));
    }
}

__global__ __launch_bounds__(256) void coord_mask_kernel(const float* __restrict__ flow,
                                                         float* __restrict__ out, int H, int W, long fbs,
                                                         int mode) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, b = blockIdx.z;
  if (x >= W) return;
  const float* fb = flow + (long)b * fbs + (long)y * W + x;
  const bool abs_in = (mode & ARFLOW_COORDS_ABS) != 0;
  const float cx = abs_in ? fb[0] : (float)x + fb[0], cy = abs_in ? fb[(long)H * W] : (float)y + fb[(long)H * W];
  const float xw = (float)(W - 1), yh = (float)(H - 1);
  const bool ok = (mode & 1) == 0 ? (cx >= 0.f && cx <= xw && cy >= 0.f && cy <= yh)
                            : (cx > 0.f && cx < xw && cy > 0.f && cy < yh);
  out[((long)b * H + y) * W + x] = ok ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void occ_bidir_kernel(const float* __restrict__ f12,
                                                        const float* __restrict__ f21,
                                                        float* __restrict__ out, int H, int W, long bs12,
                                                        long bs21, float scale, float bias) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, b = blockIdx.z;
  if (x >= W) return;
  const long os = (long)H * W;
  const float* a = f12 + (long)b * bs12 + (long)y * W + x;
  const float u = a[0], v = a[os];
  const Taps t = make_taps((float)x, (float)y, u, v, H, W, H, W, ARFLOW_PAD_ZEROS, true, ARFLOW_NORM_ARFLOW);
  const float wnw = t.wx0 * t.wy0, wne = t.wx1 * t.wy0, wsw = t.wx0 * t.wy1, wse = t.wx1 * t.wy1;
  const bool bnw = t.vx0 && t.vy0, bne = t.vx1 && t.vy0, bsw = t.vx0 && t.vy1, bse = t.vx1 && t.vy1;
  const float* s = f21 + (long)b * bs21 + (long)t.y0 * W + t.x0;
  float w2[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float* sc = s + c * os;
    float r = 0.f;
    if (bnw) r = sc[0] * wnw;
    if (bne) r = fmaf(sc[1], wne, r);
    if (bsw) r = fmaf(sc[W], wsw, r);
    if (bse) r = fmaf(sc[W + 1], wse, r);
    w2[c] = r;
  }
  const float dx = u + w2[0], dy = v + w2[1];
  const float mag = (u * u + v * v) + (w2[0] * w2[0] + w2[1] * w2[1]);
  out[((long)b * H + y) * W + x] = (dx * dx + dy * dy) > (scale * mag + bias) ? 1.f : 0.f;
}

inline dim3 pixel_grid(int B, int H, int W, int bx) { return dim3(af_cdiv(W, bx), H, B); }
inline unsigned channel_split(long tiles, int C) { return af_channel_split(tiles, C); }
inline int pick_bx(int W) { return W >= 192 ? 256 : (W >= 96 ? 128 : 64); }

}  // namespace

extern "C" int arflow_warp_fwd(const float* src, const float* flow, float* out, float* valid, int B, int C,
                               int Hs, int Ws, int H, int W, long flow_bstride, int pad_mode,
                               int align_corners, int norm_mode, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(src);
  AF_REQUIRE_PTR(flow);
  AF_REQUIRE_PTR(out);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0, ARFLOW_ESHAPE);
  AF_REQUIRE(B <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(flow_bstride >= 2L * H * W, ARFLOW_ESHAPE);
  AF_REQUIRE(pad_mode == ARFLOW_PAD_ZEROS || pad_mode == ARFLOW_PAD_BORDER, ARFLOW_EPARAM);
  AF_REQUIRE(norm_mode >= ARFLOW_NORM_ARFLOW && norm_mode <= ARFLOW_NORM_UFLOW_ABS, ARFLOW_EPARAM);
  const long tiles = (long)af_cdiv(W, 32) * af_cdiv(H, 8) * B;
  if (C <= 3)  // image warps of the losses: 3 channels per chunk (smaller LDS window, more workgroups per CU)
    hipLaunchKernelGGL(warp_fwd_kernel<3>, dim3(af_grid_for_tiles(tiles), 1), dim3(256), 0, (hipStream_t)stream, src, flow,
                       out, valid, B, C, Hs, Ws, H, W, flow_bstride, pad_mode, align_corners, norm_mode);
  else
    hipLaunchKernelGGL(warp_fwd_kernel<2>, dim3(af_grid_for_tiles(tiles), channel_split(tiles, C)), dim3(256), 0,
                       (hipStream_t)stream, src, flow, out, valid, B, C, Hs, Ws, H, W, flow_bstride, pad_mode,
                       align_corners, norm_mode);
  return af_launch_status();
}

int af_level_warp_fwd_launch(const float* x1, const float* x2, const float* flow, long flow_bstride, int flow_is_coarse,
                             int up_align_corners, float* flow_up, float* flow_up2, long flow_up2_bstride, float* x2w,
                             double* acc, int B, int C, int H, int W, int pad_mode, int align_corners, int norm_mode,
                             hipStream_t st) {
  const long tiles = (long)af_cdiv(W, 32) * af_cdiv(H, 8) * B;
  const dim3 grid(af_grid_for_tiles(tiles), channel_split(tiles, C));
  // 4 channels per chunk at the fine level (half as many load -> barrier -> taps rounds per workgroup: 90.6 -> 85.7 us for the
  // level forward at B16 C32 96x160 from cold caches, tools/level_bench.py); 2 below (48x80: 43.9 vs 45.7 us)
  if (flow_is_coarse && tiles >= 768)
    hipLaunchKernelGGL((level_warp_fwd_kernel<true, 4>), grid, dim3(256), 0, st, x1, x2, flow, flow_up, flow_up2,
                       flow_up2_bstride, x2w, acc, B, C, H, W, flow_bstride, pad_mode, align_corners, norm_mode,
                       up_align_corners);
  else if (flow_is_coarse)
    hipLaunchKernelGGL(level_warp_fwd_kernel<true>, grid, dim3(256), 0, st, x1, x2, flow, flow_up, flow_up2,
                       flow_up2_bstride, x2w, acc, B, C, H, W, flow_bstride, pad_mode, align_corners, norm_mode,
                       up_align_corners);
  else
    hipLaunchKernelGGL(level_warp_fwd_kernel<false>, grid, dim3(256), 0, st, x1, x2, flow, nullptr, nullptr,
                       0L, x2w, acc, B, C, H, W, flow_bstride, pad_mode, align_corners, norm_mode, 0);
  return af_launch_status();
}

// ---- level entry points: see level_warp_fwd_kernel -------------------------------------------
extern "C" int arflow_level_acc_rows(int B, int C, int H, int W, int has_flow) {
  if (!(B > 0 && C > 0 && H > 0 && W > 0)) return ARFLOW_ESHAPE;
  if (has_flow) {
    const long per = (long)af_cdiv(W, 32) * af_cdiv(H, 8);
    return (int)(per * channel_split(per * B, C));
  }
  return (int)af_blocks_per_sample(B, (long)C * H * W, 256 * 16);
}

extern "C" int arflow_level_warp_fwd(const float* x1, const float* x2, const float* flow, long flow_bstride,
                                     int flow_is_coarse, int up_align_corners, float* flow_up, float* flow_up2,
                                     long flow_up2_bstride, float* x2w, double* acc, int B, int C, int H, int W,
                                     int pad_mode, int align_corners, int norm_mode, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(x2);
  AF_REQUIRE_PTR(flow);
  AF_REQUIRE_PTR(x2w);
  AF_REQUIRE_PTR(acc);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && B <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(pad_mode == ARFLOW_PAD_ZEROS || pad_mode == ARFLOW_PAD_BORDER, ARFLOW_EPARAM);
  AF_REQUIRE(norm_mode == ARFLOW_NORM_ARFLOW || norm_mode == ARFLOW_NORM_UFLOW, ARFLOW_EPARAM);
  if (flow_is_coarse) {
    AF_REQUIRE(H % 2 == 0 && W % 2 == 0 && H >= 2 && W >= 2, ARFLOW_ESHAPE);
    AF_REQUIRE(flow_bstride >= 2L * (H / 2) * (W / 2), ARFLOW_ESHAPE);
    AF_REQUIRE(flow_up2 == nullptr || flow_up2_bstride >= 2L * H * W, ARFLOW_ESHAPE);
  } else {
    AF_REQUIRE(flow_bstride >= 2L * H * W, ARFLOW_ESHAPE);
  }
  return af_level_warp_fwd_launch(x1, x2, flow, flow_bstride, flow_is_coarse, up_align_corners, flow_up, flow_up2,
                                  flow_up2_bstride, x2w, acc, B, C, H, W, pad_mode, align_corners, norm_mode,
                                  (hipStream_t)stream);
}

// internal launcher (also used by the level entry points, level.hip); add1 / add2: see warp_bwd_flow_kernel
int af_warp_bwd_launch(const float* gout, const float* src, const float* flow, float* gsrc, float* gflow, int B, int C,
                       int Hs, int Ws, int H, int W, long flow_bstride, int pad_mode, int align_corners, int norm_mode,
                       const float* add1, long add1_bs, const float* add2, hipStream_t st) {
  if (!gsrc && !gflow) return ARFLOW_OK;
  if (gsrc) {
    hipError_t e = hipMemsetAsync(gsrc, 0, sizeof(float) * (size_t)B * C * Hs * Ws, st);
    if (e != hipSuccess) return af_hip_status(e);
  }
  const long tiles = (long)af_cdiv(W, 32) * af_cdiv(H, 8) * B;
  const unsigned nsplit = channel_split(tiles, C);
  if (gsrc)
    hipLaunchKernelGGL(lds_scatter::warp_bwd_src_kernel<false>, dim3(af_grid_for_tiles(tiles), nsplit), dim3(256), 0, st, gout,
                       flow, gsrc, B, C, Hs, Ws, H, W, flow_bstride, pad_mode, align_corners, norm_mode);
  if (gsrc && gflow) AF_LAUNCH_CHECK();
  if (gflow) {
    if (nsplit > 1) {
      hipError_t e = hipMemsetAsync(gflow, 0, sizeof(float) * (size_t)B * 2 * H * W, st);
      if (e != hipSuccess) return af_hip_status(e);
    }
    if (C <= 3)
      hipLaunchKernelGGL(warp_bwd_flow_kernel<3>, dim3(af_grid_for_tiles(tiles), nsplit), dim3(256), 0, st, gout, src, flow,
                         gflow, B, C, Hs, Ws, H, W, flow_bstride, pad_mode, align_corners, norm_mode, add1, add1_bs, add2);
    else
      hipLaunchKernelGGL(warp_bwd_flow_kernel<2>, dim3(af_grid_for_tiles(tiles), nsplit), dim3(256), 0, st, gout, src, flow,
                         gflow, B, C, Hs, Ws, H, W, flow_bstride, pad_mode, align_corners, norm_mode, add1, add1_bs, add2);
  }
  return af_launch_status();
}

// level backward, warp part: the normalisation's backward folded into both warp-gradient kernels (no apply pass, the
// gradient of the raw warped map is never written).  rows: partial (sum g, sum g (x - mu)) rows of featnorm's sum pass.
int af_level_warp_bwd_launch(const float* g2n, const float* x2, const float* x2w, const float* flow, long flow_bstride,
                             float* gx2, float* gflow, int B, int C, int H, int W, int pad_mode, int align_corners,
                             int norm_mode, const double* rows, int nrows, const float* stats, int featnorm_mode,
                             const float* g1n, const float* gdir, long gdir_bs, const float* x1, float* gx1,
                             const float* add1, long add1_bs, const float* add2, float* gcoarse, int up_align,
                             float* slab, void* slab_meta, int* slab_ovf, int slab_cap, int* qinfo, hipStream_t st) {
  // gx2, gflow (and gcoarse, slab_ovf) arrive ZERO-FILLED (by the correlation backward launch in front of this one)
  // -- except gx2 in the gather form (qinfo != null): gather_kernel writes every element of it
  const long per = (long)af_cdiv(W, 32) * af_cdiv(H, 8);
  const long tiles = per * B;
  const unsigned nsplit = channel_split(tiles, C);
  const dim3 grid(af_grid_for_tiles(tiles), nsplit);
  LevelBwdArgs la{rows, nrows, featnorm_mode, stats, g1n, gdir, gdir_bs, x1, gx1, gcoarse, up_align};
  // the two-pass slab form of the source gradient is OPT-IN (arflow_level_bwd_ws_bytes only reserves the slabs under
  // ARFLOW_WARP_SLAB=1): built, parity-green, measured SLOWER at B16 C32 96x160 -- stores 39 us + gather 33 us vs 64 us for
  // the atomic flush; it is bit-reproducible across workgroup scheduling, the atomics are not
  const bool want_slab = slab != nullptr && per <= lds_scatter::MAXT;
  if (qinfo) {  // fine level: d/d src as a gather over the inverse-flow window, then the flow-gradient role + the pairs it left
    hipLaunchKernelGGL(inv_gather::gather_kernel, grid, dim3(256), 0, st, g2n, x2w, flow, gx2, qinfo, B, C, H, W, flow_bstride,
                       pad_mode, align_corners, norm_mode, rows, nrows, stats, featnorm_mode);
    AF_LAUNCH_CHECK();
    la.qinfo = qinfo, la.gfix = gx2;
    hipLaunchKernelGGL(level_warp_bwd_flow_kernel<true>, grid, dim3(256), 0, st, g2n, x2, flow, gflow, B, C, H, W, flow_bstride,
                       pad_mode, align_corners, norm_mode, add1, add1_bs, add2, la);
    return af_launch_status();
  }
  if (tiles * nsplit <= 2048 && !want_slab) {  // both roles in one launch
    hipLaunchKernelGGL(level_warp_bwd_both_kernel, dim3(2 * grid.x, grid.y), dim3(256), 0, st, g2n, x2, x2w, flow, gx2, gflow,
                       B, C, H, W, flow_bstride, pad_mode, align_corners, norm_mode, add1, add1_bs, add2, la);
    return af_launch_status();
  }
  // fine level: the source gradient in the two-pass slab form (stores + gather) when the workspace is there
  const bool use_slab = want_slab;
  if (use_slab) {
    const lds_scatter::SlabArgs sl{slab, (int4*)slab_meta, slab_ovf, slab_cap};
    hipLaunchKernelGGL((lds_scatter::warp_bwd_src_kernel<true, true>), grid, dim3(256), 0, st, g2n, flow, gx2, B, C, H, W, H, W,
                       flow_bstride, pad_mode, align_corners, norm_mode, x2w, rows, nrows, stats, featnorm_mode, sl);
  } else {
    hipLaunchKernelGGL(lds_scatter::warp_bwd_src_kernel<true>, grid, dim3(256), 0, st, g2n, flow, gx2, B, C, H, W, H, W,
                       flow_bstride, pad_mode, align_corners, norm_mode, x2w, rows, nrows, stats, featnorm_mode);
  }
  AF_LAUNCH_CHECK();
  hipLaunchKernelGGL(level_warp_bwd_flow_kernel<false>, grid, dim3(256), 0, st, g2n, x2, flow, gflow, B, C, H, W, flow_bstride,
                     pad_mode, align_corners, norm_mode, add1, add1_bs, add2, la);
  if (use_slab) {
    AF_LAUNCH_CHECK();
    hipLaunchKernelGGL(lds_scatter::slab_gather_kernel, dim3(af_grid_for_tiles(tiles), C >= 32 ? 2 : 1), dim3(256), 0, st, slab,
                       (const int4*)slab_meta, slab_ovf, gx2, B, C, H, W, (int)per, slab_cap);
  }
  return af_launch_status();
}

int af_up2_bwd_launch(const float* gfine, float* gcoarse, int planes, int H, int W, int up_align, hipStream_t st) {
  const long n = (long)planes * (H / 2) * (W / 2);
  hipLaunchKernelGGL(up2_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, gfine, gcoarse, planes, H, W,
                     up_align);
  return af_launch_status();
}

// Adjoint of `flow_up = interpolate(flow * 2, x2, bilinear)` as arflow_level_warp_fwd evaluates it (gather form, no
// atomics, gcoarse fully written): gfine [B,2,H,W] -> gcoarse [B,2,H/2,W/2].  models/pwclite.py:178-179 backward.
extern "C" int arflow_up2_bwd(const float* gfine, float* gcoarse, int B, int H, int W, int up_align_corners,
                              arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(gfine);
  AF_REQUIRE_PTR(gcoarse);
  AF_REQUIRE(B > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && B <= 65535, ARFLOW_ESHAPE);
  return af_up2_bwd_launch(gfine, gcoarse, B * 2, H, W, up_align_corners, (hipStream_t)stream);
}

extern "C" int arflow_warp_bwd(const float* gout, const float* src, const float* flow, float* gsrc,
                               float* gflow, int B, int C, int Hs, int Ws, int H, int W, long flow_bstride,
                               int pad_mode, int align_corners, int norm_mode, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(gout);
  AF_REQUIRE_PTR(src);
  AF_REQUIRE_PTR(flow);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0, ARFLOW_ESHAPE);
  AF_REQUIRE(B <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(flow_bstride >= 2L * H * W, ARFLOW_ESHAPE);
  AF_REQUIRE(pad_mode == ARFLOW_PAD_ZEROS || pad_mode == ARFLOW_PAD_BORDER, ARFLOW_EPARAM);
  AF_REQUIRE(norm_mode >= ARFLOW_NORM_ARFLOW && norm_mode <= ARFLOW_NORM_UFLOW_ABS, ARFLOW_EPARAM);
  return af_warp_bwd_launch(gout, src, flow, gsrc, gflow, B, C, Hs, Ws, H, W, flow_bstride, pad_mode, align_corners,
                            norm_mode, nullptr, 0, nullptr, (hipStream_t)stream);
}

extern "C" int arflow_splat_map(const float* flow, float* out, int B, int H, int W, long flow_bstride,
                                int variant, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(flow);
  AF_REQUIRE_PTR(out);
  AF_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(flow_bstride >= 2L * H * W, ARFLOW_ESHAPE);
  AF_REQUIRE(variant >= 0 && variant <= 3, ARFLOW_EPARAM);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * H * W, st);
  if (e != hipSuccess) return af_hip_status(e);
  const long tiles = (long)af_cdiv(W, 32) * af_cdiv(H, 8) * B;
  hipLaunchKernelGGL(splat_kernel<false>, dim3(af_grid_for_tiles(tiles)), dim3(256), 0, st, flow, out, B, H, W, flow_bstride,
                     variant);
  return af_launch_status();
}

// compute_range_map(flow) (variant 0 of arflow_splat_map) AND the smoothness sums of arflow_smooth_fwd(flow, img, ...) in
// ONE launch.  `out` must arrive ZERO-FILLED when prezeroed != 0 (arflow_down4_gray_z does it), else it is cleared here.
extern "C" int arflow_splat_smooth_fwd(const float* flow, const float* img, float* out, float* sums, int B, int H, int W,
                                       long flow_bstride, float flow_scale, float alpha, int order, int wmode, int penalty,
                                       int prezeroed, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(flow);
  AF_REQUIRE_PTR(img);
  AF_REQUIRE_PTR(out);
  AF_REQUIRE_PTR(sums);
  AF_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(flow_bstride >= 2L * H * W, ARFLOW_ESHAPE);
  AF_REQUIRE(order == 1 || order == 2, ARFLOW_EPARAM);
  AF_REQUIRE(wmode == 0 || wmode == 1, ARFLOW_EPARAM);
  AF_REQUIRE(penalty == 0 || penalty == 1, ARFLOW_EPARAM);
  hipStream_t st = (hipStream_t)stream;
  if (!prezeroed) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * H * W, st);
    if (e != hipSuccess) return af_hip_status(e);
  }
  const long tiles = (long)af_cdiv(W, 32) * af_cdiv(H, 8) * B;
  const SmoothArgs sa{flow, img, 3, H, W, flow_bstride, flow_scale, alpha, order, wmode, penalty};
  hipLaunchKernelGGL(splat_kernel<true>, dim3(af_grid_for_tiles(tiles)), dim3(256), 0, st, flow, out, B, H, W, flow_bstride,
                     0, sa, sums, af_sums_rows(B, H, W));
  return af_launch_status();
}

extern "C" int arflow_coord_mask(const float* flow, float* out, int B, int H, int W, long flow_bstride,
                                 int mode, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(flow);
  AF_REQUIRE_PTR(out);
  AF_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(flow_bstride >= 2L * H * W, ARFLOW_ESHAPE);
  AF_REQUIRE(mode >= 0 && mode <= 3, ARFLOW_EPARAM);
  const int bx = pick_bx(W);
  hipLaunchKernelGGL(coord_mask_kernel, pixel_grid(B, H, W, bx), dim3(bx), 0, (hipStream_t)stream, flow, out,
                     H, W, flow_bstride, mode);
  return af_launch_status();
}

extern "C" int arflow_occ_bidir(const float* flow12, const float* flow21, float* out, int B, int H, int W,
                                long bstride12, long bstride21, float scale, float bias,
                                arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(flow12);
  AF_REQUIRE_PTR(flow21);
  AF_REQUIRE_PTR(out);
  AF_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(bstride12 >= 2L * H * W && bstride21 >= 2L * H * W, ARFLOW_ESHAPE);
  const int bx = pick_bx(W);
  hipLaunchKernelGGL(occ_bidir_kernel, pixel_grid(B, H, W, bx), dim3(bx), 0, (hipStream_t)stream, flow12,
                     flow21, out, H, W, bstride12, bstride21, scale, bias);
  return af_launch_status();
}

// ---- bf16 STORAGE of the warped features (opt-in): src holds bf16 bit patterns; coordinates,
// bilinear arithmetic, the output and both gradients are fp32.  (d/d src never reads src: arflow_warp_bwd's kernel.)
extern "C" int arflow_warp_fwd_bf16(const unsigned short* src, const float* flow, float* out, float* valid, int B, int C,
                                    int Hs, int Ws, int H, int W, long flow_bstride, int pad_mode, int align_corners,
                                    int norm_mode, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(src);
  AF_REQUIRE_PTR(flow);
  AF_REQUIRE_PTR(out);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0, ARFLOW_ESHAPE);
  AF_REQUIRE(B <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(flow_bstride >= 2L * H * W, ARFLOW_ESHAPE);
  AF_REQUIRE(pad_mode == ARFLOW_PAD_ZEROS || pad_mode == ARFLOW_PAD_BORDER, ARFLOW_EPARAM);
  AF_REQUIRE(norm_mode >= ARFLOW_NORM_ARFLOW && norm_mode <= ARFLOW_NORM_UFLOW_ABS, ARFLOW_EPARAM);
  const long tiles = (long)af_cdiv(W, 32) * af_cdiv(H, 8) * B;
  hipLaunchKernelGGL((warp_fwd_kernel<2, bf16_t>), dim3(af_grid_for_tiles(tiles), channel_split(tiles, C)), dim3(256), 0,
                     (hipStream_t)stream, src, flow, out, valid, B, C, Hs, Ws, H, W, flow_bstride, pad_mode,
                     align_corners, norm_mode);
  return af_launch_status();
}

extern "C" int arflow_warp_bwd_bf16(const float* gout, const unsigned short* src, const float* flow, float* gsrc,
                                    float* gflow, int B, int C, int Hs, int Ws, int H, int W, long flow_bstride,
                                    int pad_mode, int align_corners, int norm_mode, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(gout);
  AF_REQUIRE_PTR(src);
  AF_REQUIRE_PTR(flow);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0, ARFLOW_ESHAPE);
  AF_REQUIRE(B <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(flow_bstride >= 2L * H * W, ARFLOW_ESHAPE);
  AF_REQUIRE(pad_mode == ARFLOW_PAD_ZEROS || pad_mode == ARFLOW_PAD_BORDER, ARFLOW_EPARAM);
  AF_REQUIRE(norm_mode >= ARFLOW_NORM_ARFLOW && norm_mode <= ARFLOW_NORM_UFLOW_ABS, ARFLOW_EPARAM);
  if (!gsrc && !gflow) return ARFLOW_OK;
  hipStream_t st = (hipStream_t)stream;
  if (gsrc) {
    hipError_t e = hipMemsetAsync(gsrc, 0, sizeof(float) * (size_t)B * C * Hs * Ws, st);
    if (e != hipSuccess) return af_hip_status(e);
  }
  const long tiles = (long)af_cdiv(W, 32) * af_cdiv(H, 8) * B;
  const unsigned nsplit = channel_split(tiles, C);
  if (gsrc)
    hipLaunchKernelGGL(lds_scatter::warp_bwd_src_kernel<false>, dim3(af_grid_for_tiles(tiles), nsplit), dim3(256), 0, st, gout,
                       flow, gsrc, B, C, Hs, Ws, H, W, flow_bstride, pad_mode, align_corners, norm_mode);
  if (gsrc && gflow) AF_LAUNCH_CHECK();
  if (gflow) {
    if (nsplit > 1) {
      hipError_t e = hipMemsetAsync(gflow, 0, sizeof(float) * (size_t)B * 2 * H * W, st);
      if (e != hipSuccess) return af_hip_status(e);
    }
    hipLaunchKernelGGL((warp_bwd_flow_kernel<2, bf16_t>), dim3(af_grid_for_tiles(tiles), nsplit), dim3(256), 0, st, gout,
                       src, flow, gflow, B, C, Hs, Ws, H, W, flow_bstride, pad_mode, align_corners, norm_mode);
  }
  return af_launch_status();
}
