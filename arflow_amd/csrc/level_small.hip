// The COARSE pyramid levels (H * W <= 1024 pixels per sample: 12x20 and 24x40 of the 384x640 workloads, 14x32 / 16x28 of
// the others) as ONE launch per direction (models/pwclite_uflow.py:203-222,
// models/uflow_model.py:160-198).  At these sizes the tiled level kernels are latency chains on < 1 MB of data and every
// launch costs 5-20 us inside a training step (cold instruction cache, cold TLB, the kernel-argument round trip) whatever
// it computes: seven launches backward, two forward.  Here one workgroup (768 threads) owns a whole sample:
//
//   forward  pass 1: x2 upsampled flow (ATen upsample_bilinear2d arithmetic) -> bilinear warp of the second map, its
//                    channels staged through LDS 8 at a time -> the sample's moments (block reduction, double);
//            pass 2: cost volume of the normalised pair, 4 channels per step through a double-buffered LDS tile with a
//                    zero halo (the first map is normalised while it is staged and written out once), one (4-pixel
//                    group, row shift) item per thread: 36 accumulators, three ds_read_b128 of a 12-float window row
//                    per 36 FMAs; fused LeakyReLU + sign words.
//            With many pixels the nine row shifts are dealt to gridDim.y = 3 workgroups, each of which repeats pass 1
//            (identical values: nothing is exchanged between workgroups).
// Arithmetic identical to the tiled kernels (same normalisation-in-the-epilogue form, see corr_v2::fwd_kernel<.., NORM>).
#include "common.hpp"
#include "featnorm_stats.hpp"
#include "level_internal.hpp"
#include "taps.hpp"

namespace {
namespace small {

constexpr int NT = 768, D = 4, N = 9, PX = 4, CCH = 4, CST = 8;
constexpr int MAXHW = 1024, MAXHALO = 1536;  // pixels per sample, (H + 8) * (W + 8)
constexpr int ARENA = 32768;                 // floats: C * H * W of the staged second map
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FwdArgs {
  const float* x1;
  const float* x2;
  const float* flow;  // coarse [B,2,H/2,W/2] / fine [B,2,H,W] (HAS_FLOW), batch stride fbs
  long fbs;
  int flow_is_coarse, up_align;
  float* flow_up;   // nullable
  float* flow_up2;  // nullable, batch stride fu2_bs
  long fu2_bs;
  float* x2w;  // [B,C,H,W] (HAS_FLOW)
  int mode;
  float* out;
  long obs;
  float* x1n;  // nullable
  long x1n_bs;
  unsigned* sign;  // nullable: [B,3,H,W]
  float* stats;
  int C, H, W;
  float slope;
  int pad, align, norm;
  const double* r1;  // optional [B][n1][2] partial moments of x1 (conv epilogue); with HAS_FLOW == false also r2 for x2
  int n1;
  const double* r2;
  int n2;
};

template <bool HAS_FLOW>
__global__ __launch_bounds__(NT) void fwd_kernel(FwdArgs a) {
  // one arena: pass 1 stages the WHOLE raw second map in it (C * H * W <= ARENA floats), pass 2 reuses it for the
  // double-buffered halo tiles (xw) and the first-map chunks (x1s)
  __shared__ __attribute__((aligned(16))) float arena[ARENA];
  float* const xw = arena;                        // [2][CCH][MAXHALO]
  float* const x1s = arena + 2 * CCH * MAXHALO;   // [2][CCH][MAXHW]
  __shared__ unsigned sg[3 * MAXHW];
  __shared__ double dscratch[4 * (NT / 64)];
  __shared__ double tot[4];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int C = a.C, H = a.H, W = a.W, HW = H * W, HW4 = HW / 4;
  const float* __restrict__ x1b = a.x1 + (long)b * C * HW;
  const float* __restrict__ x2b = a.x2 + (long)b * C * HW;
  float* __restrict__ x2wb = HAS_FLOW ? a.x2w + (long)b * C * HW : nullptr;

  // ---------------- pass 1: flow, warp, moments ----------------
  float mom[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (HAS_FLOW) {
    // the whole second map -> LDS, every load of a thread in flight at once
    for (int i = tid; i < C * HW4; i += NT) reinterpret_cast<float4*>(arena)[i] = reinterpret_cast<const float4*>(x2b)[i];
    if (!a.r1)
      for (int i = tid; i < C * HW4; i += NT) {
        const float4 u = reinterpret_cast<const float4*>(x1b)[i];
        mom[0] += (u.x + u.y) + (u.z + u.w);
        mom[1] = fmaf(u.x, u.x, fmaf(u.y, u.y, fmaf(u.z, u.z, fmaf(u.w, u.w, mom[1]))));
      }
    TapPlan tp[2];
    bool act[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int px = tid + k * NT;
      act[k] = px < HW;
      Taps t = no_taps();
      if (act[k]) {
        const int y = px / W, x = px - y * W;
        float u, v;
        if (a.flow_is_coarse) {
          const int Hc = H / 2, Wc = W / 2;
          int xa, xb, ya, yb;
          float wx0, wx1, wy0, wy1;
          up2_source(x, Wc, W, a.up_align != 0, xa, xb, wx0, wx1);
          up2_source(y, Hc, H, a.up_align != 0, ya, yb, wy0, wy1);
          const float* fc = a.flow + (long)b * a.fbs;
          const long cs = (long)Hc * Wc;
          const float a00 = fc[ya * Wc + xa], a01 = fc[ya * Wc + xb], a10 = fc[yb * Wc + xa], a11 = fc[yb * Wc + xb];
          const float b00 = fc[cs + ya * Wc + xa], b01 = fc[cs + ya * Wc + xb], b10 = fc[cs + yb * Wc + xa],
                      b11 = fc[cs + yb * Wc + xb];
          u = 2.f * (wy0 * (wx0 * a00 + wx1 * a01) + wy1 * (wx0 * a10 + wx1 * a11));
          v = 2.f * (wy0 * (wx0 * b00 + wx1 * b01) + wy1 * (wx0 * b10 + wx1 * b11));
          if (blockIdx.y == 0) {
            if (a.flow_up) a.flow_up[(long)b * 2 * HW + px] = u, a.flow_up[(long)b * 2 * HW + HW + px] = v;
            if (a.flow_up2) a.flow_up2[(long)b * a.fu2_bs + px] = u, a.flow_up2[(long)b * a.fu2_bs + HW + px] = v;
          }
        } else {
          const float* fb = a.flow + (long)b * a.fbs + px;
          u = fb[0], v = fb[HW];
        }
        t = make_taps((float)x, (float)y, u, v, H, W, H, W, a.pad, a.align != 0, a.norm);
      }
      tp[k] = plan_taps(t, H, W);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (!act[k]) continue;
      const int px = tid + k * NT;
#pragma unroll 8
      for (int c = 0; c < C; ++c) {
        const float* s = arena + c * HW;
        float r = tp[k].ok[0] ? s[tp[k].o[0]] * tp[k].w[0] : 0.f;
        r = tp[k].ok[1] ? fmaf(s[tp[k].o[1]], tp[k].w[1], r) : r;
        r = tp[k].ok[2] ? fmaf(s[tp[k].o[2]], tp[k].w[2], r) : r;
        r = tp[k].ok[3] ? fmaf(s[tp[k].o[3]], tp[k].w[3], r) : r;
        x2wb[(long)c * HW + px] = r;  // every row-shift workgroup writes the same values and reads its own
        mom[2] += r, mom[3] = fmaf(r, r, mom[3]);
      }
    }
    __syncthreads();  // the arena is reused below
  } else if (!(a.r1 && a.r2)) {
    for (int i = tid; i < C * HW4; i += NT) {
      const float4 u = reinterpret_cast<const float4*>(x1b)[i], v = reinterpret_cast<const float4*>(x2b)[i];
      mom[0] += (u.x + u.y) + (u.z + u.w);
      mom[1] = fmaf(u.x, u.x, fmaf(u.y, u.y, fmaf(u.z, u.z, fmaf(u.w, u.w, mom[1]))));
      mom[2] += (v.x + v.y) + (v.z + v.w);
      mom[3] = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, mom[3]))));
    }
  }
  double dm[4] = {(double)mom[0], (double)mom[1], (double)mom[2], (double)mom[3]};
  featnorm::block_sum_f64<4, NT>(dm, dscratch);
  if (tid == 0) tot[0] = dm[0], tot[1] = dm[1], tot[2] = dm[2], tot[3] = dm[3];
  // zero the halo tiles and the sign words while the totals settle
  for (int i = tid; i < 2 * CCH * MAXHALO / 4; i += NT) reinterpret_cast<float4*>(xw)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = tid; i < 3 * MAXHW; i += NT) sg[i] = 0u;
  __syncthreads();
  double ta[4] = {tot[0], tot[1], tot[2], tot[3]};
  if (a.r1) featnorm::sum_rows2(a.r1 + 2L * a.n1 * b, a.n1, ta[0], ta[1]);  // every wave adds the rows itself
  if (!HAS_FLOW && a.r1 && a.r2) featnorm::sum_rows2(a.r2 + 2L * a.n2 * b, a.n2, ta[2], ta[3]);
  const featnorm::Moments m = featnorm::moments_from_totals(ta, (long)C * HW, a.mode);
  const float sd = sqrtf(m.var + 1e-16f);
  const float mu = m.mu, rs = 1.0f / sd;
  if (blockIdx.y == 0 && tid == 0) {
    float* st = a.stats + 4 * b;
    st[0] = m.m1, st[1] = m.m2, st[2] = m.mu, st[3] = sd;
  }

  // ---------------- pass 2: cost volume of the normalised pair ----------------
  const float* s2b = HAS_FLOW ? x2wb : x2b;
  const int WP = W + 2 * D, W4 = W / 4, groups = H * W4;
  const int ndy = N / (int)gridDim.y, dy0 = (int)blockIdx.y * ndy;
  const int dyl = tid / groups, g = tid - dyl * groups;
  const bool active = dyl < ndy;
  const int gy = g / W4, gx = 4 * (g - gy * W4), dy = dy0 + dyl;
  float* x1nb = (a.x1n && blockIdx.y == 0) ? a.x1n + (long)b * a.x1n_bs : nullptr;

  float acc[N][PX];
#pragma unroll
  for (int j = 0; j < N; ++j)
#pragma unroll
    for (int p = 0; p < PX; ++p) acc[j][p] = 0.f;
  float s1n[PX] = {0.f, 0.f, 0.f, 0.f};

  const int HALO = (H + 2 * D) * WP;
  const bool oneshot = C * (HW + HALO) <= ARENA;  // both maps fit LDS whole (12x20 at C = 32): no chunk loop, no barriers in it
  if (oneshot) {
    // halo tiles [C][HALO] (zero-filled above as far as xw reaches; clear the rest) + normalised first map [C][HW]
    float* xh = arena;
    float* xn = arena + C * HALO;
    for (int i = 2 * CCH * MAXHALO / 4 + tid; i < C * HALO / 4; i += NT) reinterpret_cast<float4*>(arena)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    for (int i = tid; i < C * HW4; i += NT) {
      const int c = i / HW4, j = i - c * HW4;
      const int px = 4 * j, y = px / W, x = px - y * W;
      const float4 v2 = reinterpret_cast<const float4*>(s2b)[i], v1 = reinterpret_cast<const float4*>(x1b)[i];
      *reinterpret_cast<float4*>(xh + c * HALO + (y + D) * WP + x + D) = v2;
      const float4 n = make_float4((v1.x - mu) * rs, (v1.y - mu) * rs, (v1.z - mu) * rs, (v1.w - mu) * rs);
      *reinterpret_cast<float4*>(xn + c * HW + px) = n;
      if (x1nb) reinterpret_cast<float4*>(x1nb)[i] = n;
    }
    __syncthreads();
    if (active) {
#pragma unroll 2
      for (int c = 0; c < C; ++c) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(xn + c * HW + gy * W + gx);
        const float* wr = xh + c * HALO + (gy + dy) * WP + gx;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr), w1 = *reinterpret_cast<const f32x4*>(wr + 4),
                    w2 = *reinterpret_cast<const f32x4*>(wr + 8);
        const float win[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
        const float av4[PX] = {av.x, av.y, av.z, av.w};
#pragma unroll
        for (int p = 0; p < PX; ++p) s1n[p] += av4[p];
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
          for (int p = 0; p < PX; ++p) acc[j][p] = fmaf(av4[p], win[j + p], acc[j][p]);
      }
    }
  }

  float4 v2[2], v1[2];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + k * NT;  // float4 index inside the chunk: CCH * HW4 = HW of them
      if (i < CCH * HW4) {
        const int c = i / HW4, j = i - c * HW4;
        v2[k] = reinterpret_cast<const float4*>(s2b + (long)(c0 + c) * HW)[j];
        v1[k] = reinterpret_cast<const float4*>(x1b + (long)(c0 + c) * HW)[j];
      }
    }
  };
  auto store_chunk = [&](int c0, int buf) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + k * NT;
      if (i < CCH * HW4) {
        const int c = i / HW4, j = i - c * HW4;
        const int px = 4 * j, y = px / W, x = px - y * W;
        *reinterpret_cast<float4*>(xw + (buf * CCH + c) * MAXHALO + (y + D) * WP + x + D) = v2[k];
        const float4 n = make_float4((v1[k].x - mu) * rs, (v1[k].y - mu) * rs, (v1[k].z - mu) * rs, (v1[k].w - mu) * rs);
        *reinterpret_cast<float4*>(x1s + (buf * CCH + c) * MAXHW + px) = n;
        if (x1nb) reinterpret_cast<float4*>(x1nb + (long)(c0 + c) * HW)[j] = n;
      }
    }
  };

  // (the pass-1 stage aliased x1s: every thread is past it -- the barrier above)
  const int nchunk = oneshot ? 0 : C / CCH;
  if (!oneshot) {
    load_chunk(0);
    store_chunk(0, 0);
  }
  for (int ch = 0; ch < nchunk; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunk) load_chunk((ch + 1) * CCH);
    __syncthreads();  // chunk ch is in LDS; everybody is done with the other buffer
    if (active) {
#pragma unroll
      for (int c = 0; c < CCH; ++c) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(x1s + (buf * CCH + c) * MAXHW + gy * W + gx);
        const float* wr = xw + (buf * CCH + c) * MAXHALO + (gy + dy) * WP + gx;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr), w1 = *reinterpret_cast<const f32x4*>(wr + 4),
                    w2 = *reinterpret_cast<const f32x4*>(wr + 8);
        const float win[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
        const float av4[PX] = {av.x, av.y, av.z, av.w};
#pragma unroll
        for (int p = 0; p < PX; ++p) s1n[p] += av4[p];
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
          for (int p = 0; p < PX; ++p) acc[j][p] = fmaf(av4[p], win[j + p], acc[j][p]);
      }
    }
    if (ch + 1 < nchunk) store_chunk((ch + 1) * CCH, buf ^ 1);
  }

  // epilogue: normalisation term of the second map, 1/C, LeakyReLU, sign words
  if (active) {
    const float sc = rs / (float)C;
    const int yy = gy + dy - D;
    const bool rowin = yy >= 0 && yy < H;
    float* ob = a.out + (long)b * a.obs + (long)(dy * N) * HW + (long)gy * W + gx;
    unsigned bits[PX] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float v[PX];
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        const int xx = gx + p + j - D;
        const bool inb = rowin && xx >= 0 && xx < W;
        v[p] = (acc[j][p] - (inb ? mu * s1n[p] : 0.f)) * sc;
        bits[p] |= v[p] > 0.f ? 1u << ((dy % 3) * N + j) : 0u;
        v[p] = v[p] > 0.f ? v[p] : v[p] * a.slope;
      }
      *reinterpret_cast<float4*>(ob + (long)j * HW) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (a.sign) {
#pragma unroll
      for (int p = 0; p < PX; ++p) atomicOr(&sg[(dy / 3) * MAXHW + gy * W + gx + p], bits[p]);
    }
  }
  if (a.sign) {
    __syncthreads();
    const int pl0 = dy0 / 3, npl = (ndy + 2) / 3;
    for (int i = tid; i < npl * HW; i += NT) {
      const int pl = pl0 + i / HW, px = i % HW;
      a.sign[((long)b * 3 + pl) * HW + px] = sg[pl * MAXHW + px];
    }
  }
}

}  // namespace small
}  // namespace

bool af_level_small_ok(int C, int H, int W) {
  static const bool off = [] {
    const char* e = getenv("ARFLOW_LEVEL_SMALL");
    return e && e[0] == '0';
  }();
  // measured (tools/level_bench.py, in-step cold): 12x20 21.3 us vs 24.8 us for the two tiled launches; at 24x40 one CU per
  // sample is bandwidth-bound (51.8 vs 29.7 us: ~0.8 MB per sample through one CU) -> only the smallest level comes here
  return !off && H * W <= 256 && H * W <= small::MAXHW && (H + 8) * (W + 8) <= small::MAXHALO && W % 4 == 0 && C % small::CST == 0 &&
         H * (W / 4) * 3 <= small::NT && (long)C * H * W <= small::ARENA;
}

int af_level_small_fwd_launch(const float* x1, const float* x2, const float* flow, long flow_bstride, int flow_is_coarse,
                              int up_align_corners, float* flow_up, float* flow_up2, long flow_up2_bstride, float* x2w,
                              int norm_mode, float* out, long out_bstride, float* x1n, long x1n_bstride, unsigned* sign_bits,
                              float* stats, int B, int C, int H, int W, float negative_slope, int pad_mode, int align_corners,
                              int coord_norm, hipStream_t st, const double* r1, int n1, const double* r2, int n2) {
  small::FwdArgs a;
  a.r1 = r1, a.n1 = n1, a.r2 = r2, a.n2 = n2;
  a.x1 = x1, a.x2 = x2, a.flow = flow, a.fbs = flow_bstride, a.flow_is_coarse = flow_is_coarse, a.up_align = up_align_corners;
  a.flow_up = flow_up, a.flow_up2 = flow_up2, a.fu2_bs = flow_up2_bstride, a.x2w = x2w, a.mode = norm_mode;
  a.out = out, a.obs = out_bstride, a.x1n = x1n, a.x1n_bs = x1n_bstride, a.sign = sign_bits, a.stats = stats;
  a.C = C, a.H = H, a.W = W, a.slope = negative_slope, a.pad = pad_mode, a.align = align_corners, a.norm = coord_norm;
  const int groups = H * (W / 4);
  const dim3 grid(B, groups * 9 <= small::NT ? 1 : 3);
  if (flow)
    hipLaunchKernelGGL(small::fwd_kernel<true>, grid, dim3(small::NT), 0, st, a);
  else
    hipLaunchKernelGGL(small::fwd_kernel<false>, grid, dim3(small::NT), 0, st, a);
  return af_launch_status();
}
