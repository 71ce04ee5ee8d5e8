// Per-sample statistics of `normalize_features` (models/pwclite_uflow.py:30-38 'joint', models/uflow_model.py:8-50
// as PWCFlow calls it 'avg'), shared by featnorm.hip (the stand-alone op) and the level kernels that fold the
// normalisation into the warp / correlation launches (warp.hip, corr.hip).
//
// Partial sums travel as rows of 4 doubles (sum x1, sum x1^2, sum x2, sum x2^2): every workgroup of a reduction
// pass STORES its row ([B][rows][4]); a consumer adds a sample's rows (one row per lane, wave-reduced) -- no
// zero-fill, no atomics, a fixed summation order.
#pragma once
#include "common.hpp"

namespace featnorm {

struct Moments {
  float m1, m2, mu, rstd, var;
};

// (sum x1, sum x1^2, sum x2, sum x2^2) over n values per tensor -> the sample's statistics
__device__ __forceinline__ Moments moments_from_totals(const double (&a)[4], long n, int mode) {
  const double dn = (double)n;
  const double m1 = a[0] / dn, m2 = a[2] / dn;
  double mu, var;
  if (mode == ARFLOW_FEATNORM_JOINT) {
    const double N = 2.0 * dn;
    mu = (a[0] + a[2]) / N;
    var = ((a[1] + a[3]) - N * mu * mu) / (N - 1.0);
  } else {
    mu = 0.5 * (m1 + m2);
    var = 0.5 * ((a[1] - dn * m1 * m1) + (a[3] - dn * m2 * m2)) / (dn - 1.0);
  }
  var = var > 0.0 ? var : 0.0;
  Moments m;
  m.m1 = (float)m1, m.m2 = (float)m2, m.mu = (float)mu, m.var = (float)var;
  m.rstd = 0.f;
  return m;
}

// sum of the first NV entries of `nrows` rows of 4 doubles: one row per lane, then a wave reduction (every lane gets it)
template <int NV>
__device__ __forceinline__ void sum_rows(const double* rows, int nrows, double (&a)[NV]) {
#pragma unroll
  for (int k = 0; k < NV; ++k) a[k] = 0.0;
  for (int r = threadIdx.x & 63; r < nrows; r += 64) {
#pragma unroll
    for (int k = 0; k < NV; ++k) a[k] += rows[4 * r + k];
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a[k] += __shfl_xor(a[k], off, 64);
  }
}

__device__ __forceinline__ Moments moments_of(const double* rows, int nrows, long n, int mode) {
  double a[4];
  sum_rows<4>(rows, nrows, a);
  return moments_from_totals(a, n, mode);
}

// sum of `nrows` rows of 2 doubles (the rows arflow_bias_act_fwd_mom leaves): one row per lane + wave reduction
__device__ __forceinline__ void sum_rows2(const double* rows, int nrows, double& s, double& q) {
  s = 0.0, q = 0.0;
  for (int r = threadIdx.x & 63; r < nrows; r += 64) s += rows[2 * r], q += rows[2 * r + 1];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64), q += __shfl_xor(q, off, 64);
}

// The sample's statistics from up to three partial-moment sources: `acc` rows of 4 doubles (sum x1, sum x1^2, sum x2,
// sum x2^2; the warp launch / moment pass), and optional 2-column rows for the first (`r1`) and the second map (`r2`)
// taken in the conv epilogue that produced them (they REPLACE the corresponding columns of acc; acc may be null when both
// are given).
struct MomentSrc {
  const double* acc;
  int nrows;
  const double* r1;
  int n1;
  const double* r2;
  int n2;
};
__device__ __forceinline__ Moments moments_of(const MomentSrc& m, int b, long n, int mode) {
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  if (m.acc && m.nrows > 0) sum_rows<4>(m.acc + 4L * m.nrows * b, m.nrows, a);
  if (m.r1) sum_rows2(m.r1 + 2L * m.n1 * b, m.n1, a[0], a[1]);
  if (m.r2) sum_rows2(m.r2 + 2L * m.n2 * b, m.n2, a[2], a[3]);
  return moments_from_totals(a, n, mode);
}

// sum over the block of NV doubles per thread; result in thread 0.  scratch: NV * NTH / 64 doubles.
template <int NV, int NTH>
__device__ __forceinline__ void block_sum_f64(double (&v)[NV], double* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) scratch[k * (NTH / 64) + wave] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      double s = 0.0;
      for (int w = 0; w < NTH / 64; ++w) s += scratch[k * (NTH / 64) + w];
      v[k] = s;
    }
  }
}

}  // namespace featnorm
