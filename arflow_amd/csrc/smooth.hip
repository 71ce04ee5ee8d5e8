// Edge-aware flow smoothness (1st / 2nd order) forward + backward, and the two resize helpers of
// the UFlow loss, for gfx950.
//
// Reference arithmetic: losses/loss_blocks.py (gradient :87-90, smooth_grad_1st :93-109,
// smooth_grad_2nd :112-124, penalty_uflow :8-9), losses/uflow_loss.py:56-102 (image_grads,
// robust_l1 of utils/uflow_utils.py:207-210,337-338), utils/uflow_utils.py:163-204
// (upsample / downsample, bilinear, align_corners=False).
//
// One lane per pixel; every value a lane needs sits within 2 pixels of it, so the neighbouring
// reads hit L1/L2 and HBM sees each tensor once.  The reference runs ~12 elementwise ATen kernels
// plus 2 reductions per direction; here it is one launch forward and one backward.
#include "common.hpp"
#include "smooth_dev.hpp"

namespace {

// A workgroup covers 256 columns x `rows` rows (chosen at launch so that ~2000 workgroups remain): fewer
// partial sums meet in the slotted rows (9216 workgroups x 2 atomics over 64 rows serialised ~290 deep at
// 384 x 640).
template <int CI>
__global__ __launch_bounds__(256) void smooth_fwd_kernel(SmoothArgs a, float* __restrict__ sums, int nrows, int rows) {
  __shared__ float red[2 * 4];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.z;
  const int o = a.order;
  float part[2] = {0.f, 0.f};
  if (x < a.W) {
    const float* ib = a.img + (long)b * a.Ci * a.H * a.W;
    const float* fb = a.flow + (long)b * a.fbs;
    const long cs = (long)a.H * a.W;
    for (int r = 0; r < rows; ++r) {
      const int y = blockIdx.y * rows + r;
      if (y >= a.H) break;
      if (x < a.W - o) {
        const float w = edge_wx<CI>(a, ib, y, x);
        part[0] += w * (pen(diff_x(a, fb, y, x), a.penalty) + pen(diff_x(a, fb + cs, y, x), a.penalty));
      }
      if (y < a.H - o) {
        const float w = edge_wy<CI>(a, ib, y, x);
        part[1] += w * (pen(diff_y(a, fb, y, x), a.penalty) + pen(diff_y(a, fb + cs, y, x), a.penalty));
      }
    }
  }
  af_block_sum<2>(part, red);
  if (threadIdx.x == 0) af_store_partial(sums, nrows, part[0], part[1], 0.f);
}

template <int CI>
__global__ __launch_bounds__(256) void smooth_bwd_kernel(SmoothArgs a, const float* __restrict__ coef,
                                                        float* __restrict__ gflow) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= a.W) return;
  smooth_bwd_pixel<CI>(a, coef, gflow, b, y, x);
}

// bilinear x1/4, align_corners=False on a multiple-of-4 grid: source coordinate of output i is
// 4i+1.5, i.e. the mean of pixels 4i+1 and 4i+2 in each axis.
__global__ __launch_bounds__(256) void down4_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                   int H, int W) {
  const int h = H / 4, w = W / 4;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const float* p = in + (long)blockIdx.z * H * W + (long)(4 * y + 1) * W + 4 * x + 1;
  // F.interpolate evaluates (1-ly)*((1-lx)*a + lx*b) + ly*((1-lx)*c + lx*d) with lx = ly = 0.5
  out[((long)blockIdx.z * h + y) * w + x] = 0.5f * (0.5f * p[0] + 0.5f * p[1]) + 0.5f * (0.5f * p[W] + 0.5f * p[W + 1]);
}

__global__ __launch_bounds__(256) void up4_clamp_mul_kernel(const float* __restrict__ in,
                                                           const float* __restrict__ valid,
                                                           float* __restrict__ out, int h, int w) {
  const int H = 4 * h, W = 4 * w;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= W) return;
  // torch area_pixel_compute_source_index: max(0.25*(dst+0.5)-0.5, 0)
  const float sy = fmaxf(0.25f * ((float)y + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.25f * ((float)x + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const float* p = in + (long)b * h * w;
  auto cl = [](float v) { return fminf(fmaxf(v, 0.f), 1.f); };
  const float v00 = cl(p[(long)y0 * w + x0]), v01 = cl(p[(long)y0 * w + x1]);
  const float v10 = cl(p[(long)y1 * w + x0]), v11 = cl(p[(long)y1 * w + x1]);
  float r = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  const long o = ((long)b * H + y) * W + x;
  if (valid) r *= valid[o];
  out[o] = r;
}

int check_smooth(const float* flow, const float* img, int B, int Ci, int H, int W, long fbs, int order,
                 int wmode, int penalty) {
  AF_REQUIRE_PTR(flow);
  AF_REQUIRE_PTR(img);
  AF_REQUIRE(B > 0 && Ci > 0 && H > 0 && W > 0 && B <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(fbs >= 2L * H * W, ARFLOW_ESHAPE);
  AF_REQUIRE(order == 1 || order == 2, ARFLOW_EPARAM);
  AF_REQUIRE(wmode == 0 || wmode == 1, ARFLOW_EPARAM);
  AF_REQUIRE(penalty == 0 || penalty == 1, ARFLOW_EPARAM);
  return ARFLOW_OK;
}

}  // namespace

extern "C" int arflow_smooth_fwd(const float* flow, const float* img, float* sums, int B, int Ci, int H,
                                 int W, long flow_bstride, float flow_scale, float alpha, int order,
                                 int wmode, int penalty, arflow_stream_t stream) {
  af_clear_stale_error();
  int rc = check_smooth(flow, img, B, Ci, H, W, flow_bstride, order, wmode, penalty);
  if (rc) return rc;
  AF_REQUIRE_PTR(sums);
  hipStream_t st = (hipStream_t)stream;
  const int nrows = af_sums_rows(B, H, W);
  SmoothArgs a{flow, img, Ci, H, W, flow_bstride, flow_scale, alpha, order, wmode, penalty};
  long rows = (long)af_cdiv(W, 256) * H * B / 2048;
  rows = rows < 1 ? 1 : (rows > 8 ? 8 : rows);
  const dim3 grid(af_cdiv(W, 256), af_cdiv(H, rows), B);
  if (Ci == 3)
    hipLaunchKernelGGL(smooth_fwd_kernel<3>, grid, dim3(256), 0, st, a, sums, nrows, (int)rows);
  else
    hipLaunchKernelGGL(smooth_fwd_kernel<0>, grid, dim3(256), 0, st, a, sums, nrows, (int)rows);
  return af_launch_status();
}

extern "C" int arflow_smooth_bwd(const float* flow, const float* img, const float* coef, float* gflow, int B,
                                 int Ci, int H, int W, long flow_bstride, float flow_scale, float alpha,
                                 int order, int wmode, int penalty, arflow_stream_t stream) {
  af_clear_stale_error();
  int rc = check_smooth(flow, img, B, Ci, H, W, flow_bstride, order, wmode, penalty);
  if (rc) return rc;
  AF_REQUIRE_PTR(coef);
  AF_REQUIRE_PTR(gflow);
  SmoothArgs a{flow, img, Ci, H, W, flow_bstride, flow_scale, alpha, order, wmode, penalty};
  if (Ci == 3)
    hipLaunchKernelGGL(smooth_bwd_kernel<3>, dim3(af_cdiv(W, 256), H, B), dim3(256), 0, (hipStream_t)stream, a, coef, gflow);
  else
    hipLaunchKernelGGL(smooth_bwd_kernel<0>, dim3(af_cdiv(W, 256), H, B), dim3(256), 0, (hipStream_t)stream, a, coef, gflow);
  return af_launch_status();
}

extern "C" int arflow_down4(const float* in, float* out, int planes, int H, int W, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(in);
  AF_REQUIRE_PTR(out);
  AF_REQUIRE(planes > 0 && H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0 && planes <= 65535, ARFLOW_ESHAPE);
  hipLaunchKernelGGL(down4_kernel, dim3(af_cdiv(W / 4, 256), H / 4, planes), dim3(256), 0, (hipStream_t)stream,
                     in, out, H, W);
  return af_launch_status();
}

extern "C" int arflow_up4_clamp_mul(const float* in, const float* valid, float* out, int B, int h, int w,
                                    arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(in);
  AF_REQUIRE_PTR(out);
  AF_REQUIRE(B > 0 && h > 0 && w > 0 && B <= 65535 && 4 * h <= 65535, ARFLOW_ESHAPE);
  hipLaunchKernelGGL(up4_clamp_mul_kernel, dim3(af_cdiv(4 * w, 256), 4 * h, B), dim3(256), 0,
                     (hipStream_t)stream, in, valid, out, h, w);
  return af_launch_status();
}
