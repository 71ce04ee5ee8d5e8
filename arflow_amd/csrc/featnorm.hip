// Feature normalisation in front of the cost volume ("next" rank 1) for gfx950.
// Reference arithmetic: normalize_features of models/pwclite_uflow.py:30-38 (moments of the
// channel-concatenated pair) and of models/uflow_model.py:8-50 as PWCFlow calls it (:167-172: per-tensor
// moments over (C,H,W), averaged across the two images, centre + normalise).  Both are
//     y_i = (x_i - mu) / sqrt(var + 1e-16),   i = 1, 2,
// with one (mu, var) per sample:
//     JOINT:  mu = sum(x1, x2) / 2n,          var = sum((x - mu)^2 over both) / (2n - 1)
//     AVG:    mu = (m1 + m2) / 2,             var = (v1 + v2) / 2,   v_i = sum((x_i - m_i)^2) / (n - 1)
//
// In eager PyTorch this is ~14 launches forward and ~30 backward per pyramid level, each a full pass
// over the feature tensors.  Here: forward = one moment pass (float4 loads, fp32 per-thread partials
// over <= 64 values, double from the wave reduction on -- sum(x^2) - sum(x)^2/n is then exact to fp32
// for any mean / spread ratio) + one apply pass; backward = one pass for the two sums the chain rule
// needs + one apply pass.  HBM-bound: 3 reads + 1 write of each tensor forward, 4 reads + 1 write backward.
#include "common.hpp"
#include "featnorm_stats.hpp"

namespace {

constexpr int NT = 256, VPT = 16;  // floats per thread per tensor and trip (4 float4)
// Partial sums: every workgroup of the reduction pass STORES its 4 doubles into its own row of `acc`
// ([B][rows][4], rows = its grid.x <= 2048/B + 1), the apply pass adds a sample's rows (one per lane, wave-reduced).
// No zero-fill and no atomics (round 1 added into 8 slotted rows per sample after a hipMemsetAsync -- a separate
// ~4 us GPU operation per call; ~120 workgroups adding into ONE address per sample had serialised at the L2).

using featnorm::Moments;
using featnorm::moments_from_totals;
using featnorm::moments_of;
using featnorm::sum_rows;
template <int NV, int NTH = NT>
__device__ __forceinline__ void block_sum_f64(double (&v)[NV], double* scratch) {
  featnorm::block_sum_f64<NV, NTH>(v, scratch);
}

// grid (blocks per sample, B).  acc[b] += (sum x1, sum x1^2, sum x2, sum x2^2) of this block's slice.
__global__ __launch_bounds__(NT) void moment_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                    double* __restrict__ acc, long n) {
  __shared__ double scratch[4 * (NT / 64)];
  const int b = blockIdx.y;
  const float* p1 = x1 + (long)b * n;
  const float* p2 = x2 + (long)b * n;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  const long n4 = (n % 4 == 0) ? n / 4 : 0;  // float4 part (rows are 16-byte aligned only when n % 4 == 0)
  for (long i0 = (long)blockIdx.x * NT * (VPT / 4); i0 < n4; i0 += (long)gridDim.x * NT * (VPT / 4)) {
    float a1 = 0.f, q1 = 0.f, a2 = 0.f, q2 = 0.f;
#pragma unroll
    for (int k = 0; k < VPT / 4; ++k) {
      const long i = i0 + (long)k * NT + threadIdx.x;
      if (i < n4) {
        const float4 u = reinterpret_cast<const float4*>(p1)[i];
        const float4 v = reinterpret_cast<const float4*>(p2)[i];
        a1 += (u.x + u.y) + (u.z + u.w);
        q1 = fmaf(u.x, u.x, fmaf(u.y, u.y, fmaf(u.z, u.z, fmaf(u.w, u.w, q1))));
        a2 += (v.x + v.y) + (v.z + v.w);
        q2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, q2))));
      }
    }
    s[0] += (double)a1, s[1] += (double)q1, s[2] += (double)a2, s[3] += (double)q2;
  }
  for (long i = 4 * n4 + (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {  // unaligned sizes
    const float u = p1[i], v = p2[i];
    s[0] += (double)u, s[1] += (double)u * (double)u, s[2] += (double)v, s[3] += (double)v * (double)v;
  }
  block_sum_f64<4>(s, scratch);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[4 * ((long)b * gridDim.x + blockIdx.x) + k] = s[k];
  }
}

// y_i = (x_i - mu) / std; block (0, b) records (m1, m2, mu, std) for the backward.
__global__ __launch_bounds__(NT) void apply_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                   float* __restrict__ y1, float* __restrict__ y2,
                                                   const double* __restrict__ acc, int nrows, float* __restrict__ stats,
                                                   long n, int mode) {
  const int b = blockIdx.y;
  const Moments m = moments_of(acc + 4L * nrows * b, nrows, n, mode);
  const float sd = sqrtf(m.var + 1e-16f);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float* st = stats + 4 * b;
    st[0] = m.m1, st[1] = m.m2, st[2] = m.mu, st[3] = sd;
  }
  const float* p1 = x1 + (long)b * n;
  const float* p2 = x2 + (long)b * n;
  float* o1 = y1 + (long)b * n;
  float* o2 = y2 + (long)b * n;
  const long n4 = (n % 4 == 0) ? n / 4 : 0;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long)gridDim.x * NT) {
    const float4 u = reinterpret_cast<const float4*>(p1)[i];
    const float4 v = reinterpret_cast<const float4*>(p2)[i];
    reinterpret_cast<float4*>(o1)[i] =
        make_float4((u.x - m.mu) / sd, (u.y - m.mu) / sd, (u.z - m.mu) / sd, (u.w - m.mu) / sd);
    reinterpret_cast<float4*>(o2)[i] =
        make_float4((v.x - m.mu) / sd, (v.y - m.mu) / sd, (v.z - m.mu) / sd, (v.w - m.mu) / sd);
  }
  for (long i = 4 * n4 + (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
    o1[i] = (p1[i] - m.mu) / sd;
    o2[i] = (p2[i] - m.mu) / sd;
  }
}

// acc[b] += (sum g, sum g (x - mu)) over both tensors (entries 0, 1)
// g1b (nullable): a second gradient of y1 that is added on load -- the slot of the decoder's concatenated input that
// holds y1 (sample b at g1b + b * g1b_bs, n contiguous floats); saves the separate add pass.
__device__ __forceinline__ float4 ld4_sum(const float* a, const float* b, long i) {
  float4 v = reinterpret_cast<const float4*>(a)[i];
  if (b) {
    const float4 w = reinterpret_cast<const float4*>(b)[i];
    v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
  }
  return v;
}
__global__ __launch_bounds__(NT) void bwd_sum_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                     const float* __restrict__ x1, const float* __restrict__ x2,
                                                     const float* __restrict__ stats, double* __restrict__ acc,
                                                     long n, const float* __restrict__ g1b, long g1b_bs) {
  __shared__ double scratch[2 * (NT / 64)];
  const int b = blockIdx.y;
  const float mu = stats[4 * b + 2];
  const long o = (long)b * n;
  const float* gbp = g1b ? g1b + (long)b * g1b_bs : nullptr;
  double s[2] = {0.0, 0.0};
  const long n4 = (n % 4 == 0) ? n / 4 : 0;
  for (long i0 = (long)blockIdx.x * NT * (VPT / 4); i0 < n4; i0 += (long)gridDim.x * NT * (VPT / 4)) {
    float sg = 0.f, sq = 0.f;
#pragma unroll
    for (int k = 0; k < VPT / 4; ++k) {
      const long i = i0 + (long)k * NT + threadIdx.x;
      if (i < n4) {
        const float4 ga = ld4_sum(g1 + o, gbp, i), xa = reinterpret_cast<const float4*>(x1 + o)[i];
        const float4 gb = reinterpret_cast<const float4*>(g2 + o)[i], xb = reinterpret_cast<const float4*>(x2 + o)[i];
        sg += ((ga.x + ga.y) + (ga.z + ga.w)) + ((gb.x + gb.y) + (gb.z + gb.w));
        sq = fmaf(ga.x, xa.x - mu, fmaf(ga.y, xa.y - mu, fmaf(ga.z, xa.z - mu, fmaf(ga.w, xa.w - mu, sq))));
        sq = fmaf(gb.x, xb.x - mu, fmaf(gb.y, xb.y - mu, fmaf(gb.z, xb.z - mu, fmaf(gb.w, xb.w - mu, sq))));
      }
    }
    s[0] += (double)sg, s[1] += (double)sq;
  }
  for (long i = 4 * n4 + (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
    const float ga = g1[o + i] + (gbp ? gbp[i] : 0.f), gb = g2[o + i];
    s[0] += (double)ga + (double)gb;
    s[1] += (double)ga * (double)(x1[o + i] - mu) + (double)gb * (double)(x2[o + i] - mu);
  }
  block_sum_f64<2>(s, scratch);
  if (threadIdx.x == 0) {
    acc[4 * ((long)b * gridDim.x + blockIdx.x)] = s[0];
    acc[4 * ((long)b * gridDim.x + blockIdx.x) + 1] = s[1];
  }
}

// With r = 1/std, G = sum g, Q = sum g (x - mu) (both tensors):
//   JOINT: dx = r g - r G / N - r^3 Q (x - mu) / (N - 1),           N = 2n
//   AVG:   dx_i = r g - r G / (2n) - r^3 Q (x - m_i) / (2 (n - 1))
__global__ __launch_bounds__(NT) void bwd_apply_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                       const float* __restrict__ x1, const float* __restrict__ x2,
                                                       const float* __restrict__ stats, const double* __restrict__ acc,
                                                       int nrows, float* __restrict__ d1, float* __restrict__ d2, long n,
                                                       int mode, const float* __restrict__ g1b, long g1b_bs) {
  const int b = blockIdx.y;
  const float* gbp = g1b ? g1b + (long)b * g1b_bs : nullptr;
  const float* st = stats + 4 * b;
  const double r = 1.0 / (double)st[3];
  double gq[2];
  sum_rows<2>(acc + 4L * nrows * b, nrows, gq);
  const double G = gq[0], Q = gq[1];
  const double dn = (double)n;
  const float rf = (float)r;
  const float cg = (float)(r * G / (2.0 * dn));
  const float cq = (float)(mode == ARFLOW_FEATNORM_JOINT ? r * r * r * Q / (2.0 * dn - 1.0) : r * r * r * Q / (2.0 * (dn - 1.0)));
  const float c1 = mode == ARFLOW_FEATNORM_JOINT ? st[2] : st[0], c2 = mode == ARFLOW_FEATNORM_JOINT ? st[2] : st[1];
  const long o = (long)b * n;
  auto f = [&](float g, float x, float c) { return fmaf(rf, g, -cg) - cq * (x - c); };
  const long n4 = (n % 4 == 0) ? n / 4 : 0;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long)gridDim.x * NT) {
    if (d1) {
      const float4 g = ld4_sum(g1 + o, gbp, i), x = reinterpret_cast<const float4*>(x1 + o)[i];
      reinterpret_cast<float4*>(d1 + o)[i] = make_float4(f(g.x, x.x, c1), f(g.y, x.y, c1), f(g.z, x.z, c1), f(g.w, x.w, c1));
    }
    if (d2) {
      const float4 g = reinterpret_cast<const float4*>(g2 + o)[i], x = reinterpret_cast<const float4*>(x2 + o)[i];
      reinterpret_cast<float4*>(d2 + o)[i] = make_float4(f(g.x, x.x, c2), f(g.y, x.y, c2), f(g.z, x.z, c2), f(g.w, x.w, c2));
    }
  }
  for (long i = 4 * n4 + (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
    if (d1) d1[o + i] = f(g1[o + i] + (gbp ? gbp[i] : 0.f), x1[o + i], c1);
    if (d2) d2[o + i] = f(g2[o + i], x2[o + i], c2);
  }
}

// Coarsest pyramid levels (n <= SMALL_N floats per tensor and sample; at n = 30720 the two-launch form is
// already faster, 9.7 vs 12.9 us): ONE launch, one 1024-thread workgroup
// per sample does the reduction and the apply pass back to back (the second read comes from L2) -- the
// two-launch form spends more time in launch latency and the accumulator zero-fill than in the kernels.
constexpr int NTS = 1024;
constexpr long SMALL_N = 16384;

__global__ __launch_bounds__(NTS) void fwd_small_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                        float* __restrict__ y1, float* __restrict__ y2,
                                                        float* __restrict__ stats, long n, int mode) {
  __shared__ double scratch[4 * (NTS / 64)];
  __shared__ double tot[4];
  const int b = blockIdx.x;
  const float* p1 = x1 + (long)b * n;
  const float* p2 = x2 + (long)b * n;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  const long n4 = (n % 4 == 0) ? n / 4 : 0;
  for (long i0 = 0; i0 < n4; i0 += 4 * NTS) {  // fp32 partials over <= 16 values, double beyond
    float a1 = 0.f, q1 = 0.f, a2 = 0.f, q2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long i = i0 + (long)k * NTS + threadIdx.x;
      if (i < n4) {
        const float4 u = reinterpret_cast<const float4*>(p1)[i];
        const float4 v = reinterpret_cast<const float4*>(p2)[i];
        a1 += (u.x + u.y) + (u.z + u.w);
        q1 = fmaf(u.x, u.x, fmaf(u.y, u.y, fmaf(u.z, u.z, fmaf(u.w, u.w, q1))));
        a2 += (v.x + v.y) + (v.z + v.w);
        q2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, q2))));
      }
    }
    s[0] += (double)a1, s[1] += (double)q1, s[2] += (double)a2, s[3] += (double)q2;
  }
  for (long i = 4 * n4 + threadIdx.x; i < n; i += NTS) {
    const float u = p1[i], v = p2[i];
    s[0] += (double)u, s[1] += (double)u * (double)u, s[2] += (double)v, s[3] += (double)v * (double)v;
  }
  block_sum_f64<4, NTS>(s, scratch);
  if (threadIdx.x == 0) tot[0] = s[0], tot[1] = s[1], tot[2] = s[2], tot[3] = s[3];
  __syncthreads();
  const double a[4] = {tot[0], tot[1], tot[2], tot[3]};
  const Moments m = moments_from_totals(a, n, mode);
  const float sd = sqrtf(m.var + 1e-16f);
  if (threadIdx.x == 0) {
    float* st = stats + 4 * b;
    st[0] = m.m1, st[1] = m.m2, st[2] = m.mu, st[3] = sd;
  }
  float* o1 = y1 + (long)b * n;
  float* o2 = y2 + (long)b * n;
  for (long i = threadIdx.x; i < n4; i += NTS) {
    const float4 u = reinterpret_cast<const float4*>(p1)[i];
    const float4 v = reinterpret_cast<const float4*>(p2)[i];
    reinterpret_cast<float4*>(o1)[i] =
        make_float4((u.x - m.mu) / sd, (u.y - m.mu) / sd, (u.z - m.mu) / sd, (u.w - m.mu) / sd);
    reinterpret_cast<float4*>(o2)[i] =
        make_float4((v.x - m.mu) / sd, (v.y - m.mu) / sd, (v.z - m.mu) / sd, (v.w - m.mu) / sd);
  }
  for (long i = 4 * n4 + threadIdx.x; i < n; i += NTS) {
    o1[i] = (p1[i] - m.mu) / sd;
    o2[i] = (p2[i] - m.mu) / sd;
  }
}

__global__ __launch_bounds__(NTS) void bwd_small_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                        const float* __restrict__ x1, const float* __restrict__ x2,
                                                        const float* __restrict__ stats, float* __restrict__ d1,
                                                        float* __restrict__ d2, long n, int mode,
                                                        const float* __restrict__ g1b, long g1b_bs) {
  __shared__ double scratch[2 * (NTS / 64)];
  __shared__ double tot[2];
  const int b = blockIdx.x;
  const float* gbp = g1b ? g1b + (long)b * g1b_bs : nullptr;
  const float* st = stats + 4 * b;
  const float mu = st[2];
  const long o = (long)b * n;
  double s[2] = {0.0, 0.0};
  const long n4 = (n % 4 == 0) ? n / 4 : 0;
  for (long i0 = 0; i0 < n4; i0 += 4 * NTS) {
    float sg = 0.f, sq = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long i = i0 + (long)k * NTS + threadIdx.x;
      if (i < n4) {
        const float4 ga = ld4_sum(g1 + o, gbp, i), xa = reinterpret_cast<const float4*>(x1 + o)[i];
        const float4 gb = reinterpret_cast<const float4*>(g2 + o)[i], xb = reinterpret_cast<const float4*>(x2 + o)[i];
        sg += ((ga.x + ga.y) + (ga.z + ga.w)) + ((gb.x + gb.y) + (gb.z + gb.w));
        sq = fmaf(ga.x, xa.x - mu, fmaf(ga.y, xa.y - mu, fmaf(ga.z, xa.z - mu, fmaf(ga.w, xa.w - mu, sq))));
        sq = fmaf(gb.x, xb.x - mu, fmaf(gb.y, xb.y - mu, fmaf(gb.z, xb.z - mu, fmaf(gb.w, xb.w - mu, sq))));
      }
    }
    s[0] += (double)sg, s[1] += (double)sq;
  }
  for (long i = 4 * n4 + threadIdx.x; i < n; i += NTS) {
    const float ga = g1[o + i] + (gbp ? gbp[i] : 0.f), gb = g2[o + i];
    s[0] += (double)ga + (double)gb;
    s[1] += (double)ga * (double)(x1[o + i] - mu) + (double)gb * (double)(x2[o + i] - mu);
  }
  block_sum_f64<2, NTS>(s, scratch);
  if (threadIdx.x == 0) tot[0] = s[0], tot[1] = s[1];
  __syncthreads();
  const double r = 1.0 / (double)st[3], G = tot[0], Q = tot[1], dn = (double)n;
  const float rf = (float)r;
  const float cg = (float)(r * G / (2.0 * dn));
  const float cq = (float)(mode == ARFLOW_FEATNORM_JOINT ? r * r * r * Q / (2.0 * dn - 1.0) : r * r * r * Q / (2.0 * (dn - 1.0)));
  const float c1 = mode == ARFLOW_FEATNORM_JOINT ? st[2] : st[0], c2 = mode == ARFLOW_FEATNORM_JOINT ? st[2] : st[1];
  auto f = [&](float g, float x, float c) { return fmaf(rf, g, -cg) - cq * (x - c); };
  for (long i = threadIdx.x; i < n4; i += NTS) {
    if (d1) {
      const float4 g = ld4_sum(g1 + o, gbp, i), x = reinterpret_cast<const float4*>(x1 + o)[i];
      reinterpret_cast<float4*>(d1 + o)[i] = make_float4(f(g.x, x.x, c1), f(g.y, x.y, c1), f(g.z, x.z, c1), f(g.w, x.w, c1));
    }
    if (d2) {
      const float4 g = reinterpret_cast<const float4*>(g2 + o)[i], x = reinterpret_cast<const float4*>(x2 + o)[i];
      reinterpret_cast<float4*>(d2 + o)[i] = make_float4(f(g.x, x.x, c2), f(g.y, x.y, c2), f(g.z, x.z, c2), f(g.w, x.w, c2));
    }
  }
  for (long i = 4 * n4 + threadIdx.x; i < n; i += NTS) {
    if (d1) d1[o + i] = f(g1[o + i] + (gbp ? gbp[i] : 0.f), x1[o + i], c1);
    if (d2) d2[o + i] = f(g2[o + i], x2[o + i], c2);
  }
}

inline unsigned blocks_per_sample(int B, long n, int floats_per_block) { return af_blocks_per_sample(B, n, floats_per_block); }

}  // namespace

extern "C" int arflow_featnorm_fwd(const float* x1, const float* x2, float* y1, float* y2, double* acc, float* stats,
                                   int B, long n, int mode, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(x1);
  AF_REQUIRE_PTR(x2);
  AF_REQUIRE_PTR(y1);
  AF_REQUIRE_PTR(y2);
  AF_REQUIRE_PTR(acc);
  AF_REQUIRE_PTR(stats);
  AF_REQUIRE(B > 0 && B <= 65535 && n >= 2, ARFLOW_ESHAPE);
  AF_REQUIRE(mode == ARFLOW_FEATNORM_JOINT || mode == ARFLOW_FEATNORM_AVG, ARFLOW_EPARAM);
  hipStream_t st = (hipStream_t)stream;
  if (n <= SMALL_N) {
    hipLaunchKernelGGL(fwd_small_kernel, dim3(B), dim3(NTS), 0, st, x1, x2, y1, y2, stats, n, mode);
    return af_launch_status();
  }
  const unsigned rows = blocks_per_sample(B, n, NT * VPT);  // rows * B <= 2048 + B: fits ARFLOW_FEATNORM_ACC_DOUBLES(B)
  hipLaunchKernelGGL(moment_kernel, dim3(rows, B), dim3(NT), 0, st, x1, x2, acc, n);
  AF_LAUNCH_CHECK();
  hipLaunchKernelGGL(apply_kernel, dim3(blocks_per_sample(B, n, NT * 4), B), dim3(NT), 0, st, x1, x2, y1, y2, acc, (int)rows,
                     stats, n, mode);
  return af_launch_status();
}

// internal launcher (also used by the level entry points, level.hip): g1b = optional second gradient of y1, added on load
int af_featnorm_bwd_launch(const float* g1, const float* g1b, long g1b_bs, const float* g2, const float* x1, const float* x2,
                           const float* stats, double* acc, float* gx1, float* gx2, int B, long n, int mode, hipStream_t st) {
  if (!gx1 && !gx2) return ARFLOW_OK;
  if (n <= SMALL_N) {
    hipLaunchKernelGGL(bwd_small_kernel, dim3(B), dim3(NTS), 0, st, g1, g2, x1, x2, stats, gx1, gx2, n, mode, g1b, g1b_bs);
    return af_launch_status();
  }
  const unsigned rows = blocks_per_sample(B, n, NT * VPT);
  hipLaunchKernelGGL(bwd_sum_kernel, dim3(rows, B), dim3(NT), 0, st, g1, g2, x1, x2, stats, acc, n, g1b, g1b_bs);
  AF_LAUNCH_CHECK();
  hipLaunchKernelGGL(bwd_apply_kernel, dim3(blocks_per_sample(B, n, NT * 4), B), dim3(NT), 0, st, g1, g2, x1, x2, stats, acc,
                     (int)rows, gx1, gx2, n, mode, g1b, g1b_bs);
  return af_launch_status();
}

extern "C" int arflow_featnorm_bwd(const float* g1, const float* g2, const float* x1, const float* x2,
                                   const float* stats, double* acc, float* gx1, float* gx2, int B, long n, int mode,
                                   arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(g1);
  AF_REQUIRE_PTR(g2);
  AF_REQUIRE_PTR(x1);
  AF_REQUIRE_PTR(x2);
  AF_REQUIRE_PTR(stats);
  AF_REQUIRE_PTR(acc);
  AF_REQUIRE(B > 0 && B <= 65535 && n >= 2, ARFLOW_ESHAPE);
  AF_REQUIRE(mode == ARFLOW_FEATNORM_JOINT || mode == ARFLOW_FEATNORM_AVG, ARFLOW_EPARAM);
  return af_featnorm_bwd_launch(g1, nullptr, 0, g2, x1, x2, stats, acc, gx1, gx2, B, n, mode, (hipStream_t)stream);
}

// apply pass alone: the two sums come as rows of 4 doubles ([B][nrows][4]: sum g, sum g (x - mu)) left by the level
// correlation backward (corr_v2::bwd_kernel<.., SUMS>)
int af_featnorm_bwd_apply_launch(const float* g1, const float* g1b, long g1b_bs, const float* g2, const float* x1,
                                 const float* x2, const float* stats, const double* rows, int nrows, float* gx1, float* gx2,
                                 int B, long n, int mode, hipStream_t st) {
  hipLaunchKernelGGL(bwd_apply_kernel, dim3(blocks_per_sample(B, n, NT * 4), B), dim3(NT), 0, st, g1, g2, x1, x2, stats, rows,
                     nrows, gx1, gx2, n, mode, g1b, g1b_bs);
  return af_launch_status();
}

// sum pass alone (any n): rows of (sum g, sum g (x - mu)) in `acc`; returns the rows per sample through *nrows
int af_featnorm_bwd_sums_launch(const float* g1, const float* g1b, long g1b_bs, const float* g2, const float* x1,
                                const float* x2, const float* stats, double* acc, int* nrows, int B, long n, hipStream_t st) {
  const unsigned rows = blocks_per_sample(B, n, NT * VPT);
  *nrows = (int)rows;
  hipLaunchKernelGGL(bwd_sum_kernel, dim3(rows, B), dim3(NT), 0, st, g1, g2, x1, x2, stats, acc, n, g1b, g1b_bs);
  return af_launch_status();
}

int af_featnorm_moments_launch(const float* x1, const float* x2, double* acc, int B, long n, hipStream_t st) {
  const unsigned rows = blocks_per_sample(B, n, NT * VPT);
  hipLaunchKernelGGL(moment_kernel, dim3(rows, B), dim3(NT), 0, st, x1, x2, acc, n);
  return af_launch_status();
}

// Level without a warp (the coarsest one, models/pwclite_uflow.py:206-207): the moment pass alone, rows in the layout
// the level correlation reads (arflow_level_acc_rows(..., has_flow = 0) rows per sample).
extern "C" int arflow_level_moments(const float* x1, const float* x2, double* acc, int B, long n, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(x1);
  AF_REQUIRE_PTR(x2);
  AF_REQUIRE_PTR(acc);
  AF_REQUIRE(B > 0 && B <= 65535 && n >= 2, ARFLOW_ESHAPE);
  return af_featnorm_moments_launch(x1, x2, acc, B, n, (hipStream_t)stream);
}
