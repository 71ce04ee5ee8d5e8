// Bias + LeakyReLU epilogue of the host models' convolutions, and its backward (LeakyReLU derivative and
// bias gradient in one pass), for gfx950.  Every conv block of the reference models is
// Sequential(Conv2d(bias=True), LeakyReLU(0.1)) (models/pwclite.py:10-23, models/uflow_model.py:271-287);
// in eager PyTorch on ROCm that is conv + bias add + LeakyReLU forward and LeakyReLU-backward + a
// full-tensor reduction for the bias gradient backward -- four extra passes over every activation.  MIOpen
// keeps the convolution (bias-free); these two kernels do the rest in one pass each, HBM-bound.
#include "common.hpp"

namespace {
constexpr int NT = 256, EPT = 16;  // elements per thread (4 float4)

// y = lrelu(x + bias[c]); x, y: [B, C, HW] (may alias).  grid (chunks of a plane, C, B).
// MOM: the workgroup also leaves (sum y, sum y^2) of its slice as one row of 2 doubles in `mom` ([B][C * gridDim.x][2]):
// the partial moments of normalize_features (models/pwclite_uflow.py:30-38), taken where the feature map is produced --
// the level kernels (level.hip) then need no pass over the first feature map for them.
template <bool MOM>
__global__ __launch_bounds__(NT) void bias_act_fwd_kernel(const float* x, const float* __restrict__ bias, float* y,
                                                          int C, long HW, float slope, double* __restrict__ mom) {
  __shared__ double dred[2 * (NT / 64)];
  const int c = blockIdx.y;
  const long base = ((long)blockIdx.z * C + c) * HW;
  const float bv = bias ? bias[c] : 0.f;
  float s = 0.f, q = 0.f;  // <= 16 values per thread in fp32, double from the wave reduction on
  auto f = [&](float v) {
    v += bv;
    v = v > 0.f ? v : v * slope;
    if (MOM) s += v, q = fmaf(v, v, q);
    return v;
  };
  if ((HW & 3) == 0) {
    const long n4 = HW / 4;
#pragma unroll
    for (int k = 0; k < EPT / 4; ++k) {
      const long i = ((long)blockIdx.x * (EPT / 4) + k) * NT + threadIdx.x;
      if (i < n4) {
        const float4 v = reinterpret_cast<const float4*>(x + base)[i];
        reinterpret_cast<float4*>(y + base)[i] = make_float4(f(v.x), f(v.y), f(v.z), f(v.w));
      }
    }
  } else {
    for (long i = (long)blockIdx.x * NT * EPT + threadIdx.x; i < min(HW, ((long)blockIdx.x + 1) * NT * EPT); i += NT)
      y[base + i] = f(x[base + i]);
  }
  if (MOM) {
    double ds = (double)s, dq = (double)q;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ds += __shfl_xor(ds, off, 64), dq += __shfl_xor(dq, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) dred[2 * wave] = ds, dred[2 * wave + 1] = dq;
    __syncthreads();
    if (threadIdx.x == 0) {
      double* row = mom + 2 * (((long)blockIdx.z * C + c) * gridDim.x + blockIdx.x);
      row[0] = (dred[0] + dred[2]) + (dred[4] + dred[6]);
      row[1] = (dred[1] + dred[3]) + (dred[5] + dred[7]);
    }
  }
}

// gin = gout * (y > 0 ? 1 : slope); gbias[c] += sum over the block's slice of gin
__global__ __launch_bounds__(NT) void bias_act_bwd_kernel(const float* gout, const float* __restrict__ y, float* gin,
                                                          float* __restrict__ gbias, int C, long HW, float slope) {
  __shared__ float red[NT / 64];
  const int c = blockIdx.y;
  const long base = ((long)blockIdx.z * C + c) * HW;
  float s[1] = {0.f};
  if ((HW & 3) == 0) {
    const long n4 = HW / 4;
#pragma unroll
    for (int k = 0; k < EPT / 4; ++k) {
      const long i = ((long)blockIdx.x * (EPT / 4) + k) * NT + threadIdx.x;
      if (i < n4) {
        const float4 g = reinterpret_cast<const float4*>(gout + base)[i];
        const float4 v = reinterpret_cast<const float4*>(y + base)[i];
        const float4 r = make_float4(v.x > 0.f ? g.x : g.x * slope, v.y > 0.f ? g.y : g.y * slope,
                                     v.z > 0.f ? g.z : g.z * slope, v.w > 0.f ? g.w : g.w * slope);
        reinterpret_cast<float4*>(gin + base)[i] = r;
        s[0] += (r.x + r.y) + (r.z + r.w);
      }
    }
  } else {
    for (long i = (long)blockIdx.x * NT * EPT + threadIdx.x; i < min(HW, ((long)blockIdx.x + 1) * NT * EPT); i += NT) {
      const float g = gout[base + i];
      const float r = y[base + i] > 0.f ? g : g * slope;
      gin[base + i] = r;
      s[0] += r;
    }
  }
  if (gbias) {
    af_block_sum<1>(s, red);
    if (threadIdx.x == 0) atomicAdd(gbias + c, s[0]);
  }
}
}  // namespace

extern "C" int arflow_bias_act_fwd(const float* x, const float* bias, float* y, int B, int C, long HW,
                                   float negative_slope, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(x);
  AF_REQUIRE_PTR(y);
  AF_REQUIRE(B > 0 && C > 0 && HW > 0 && B <= 65535 && C <= 65535, ARFLOW_ESHAPE);
  hipLaunchKernelGGL(bias_act_fwd_kernel<false>, dim3(af_cdiv(HW, NT * EPT), C, B), dim3(NT), 0, (hipStream_t)stream, x, bias, y, C,
                     HW, negative_slope, (double*)nullptr);
  return af_launch_status();
}

// As arflow_bias_act_fwd; additionally the partial moments (sum y, sum y^2) per workgroup: mom holds
// [B][arflow_bias_act_mom_rows(C, HW)][2] doubles (every row written by the call).
extern "C" int arflow_bias_act_mom_rows(int C, long HW) { return (C > 0 && HW > 0) ? C * af_cdiv(HW, NT * EPT) : ARFLOW_ESHAPE; }
extern "C" int arflow_bias_act_fwd_mom(const float* x, const float* bias, float* y, double* mom, int B, int C, long HW,
                                       float negative_slope, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(x);
  AF_REQUIRE_PTR(y);
  AF_REQUIRE_PTR(mom);
  AF_REQUIRE(B > 0 && C > 0 && HW > 0 && B <= 65535 && C <= 65535, ARFLOW_ESHAPE);
  hipLaunchKernelGGL(bias_act_fwd_kernel<true>, dim3(af_cdiv(HW, NT * EPT), C, B), dim3(NT), 0, (hipStream_t)stream, x, bias, y, C,
                     HW, negative_slope, mom);
  return af_launch_status();
}

extern "C" int arflow_bias_act_bwd(const float* gout, const float* y, float* gin, float* gbias, int B, int C, long HW,
                                   float negative_slope, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(gout);
  AF_REQUIRE_PTR(y);
  AF_REQUIRE_PTR(gin);
  AF_REQUIRE(B > 0 && C > 0 && HW > 0 && B <= 65535 && C <= 65535, ARFLOW_ESHAPE);
  hipStream_t st = (hipStream_t)stream;
  if (gbias) {
    hipError_t e = hipMemsetAsync(gbias, 0, sizeof(float) * (size_t)C, st);
    if (e != hipSuccess) return af_hip_status(e);
  }
  hipLaunchKernelGGL(bias_act_bwd_kernel, dim3(af_cdiv(HW, NT * EPT), C, B), dim3(NT), 0, st, gout, y, gin, gbias, C, HW,
                     negative_slope);
  return af_launch_status();
}
