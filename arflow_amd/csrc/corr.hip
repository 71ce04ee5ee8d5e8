// Cost-volume correlation for gfx950 (MI355X): forward and both input gradients.
//
//   out[b, i*N+j, y, x] = (1/C) * sum_c x1[b,c,y,x] * x2[b,c,y+i-D,x+j-D]        N = 2D+1
//
// Reference arithmetic: models/correlation_native.py:13-23 (also models/uflow_model.py:53-92 and
// the legacy CUDA extension models/correlation_package/correlation_cuda_kernel.cu:41-300).  This
// is a new CDNA4 design, not a translation of that extension:
//   * NCHW is read directly -- no padded NHWC scratch copies (the extension's `channels_first`
//     pass and its rInput1/rInput2 buffers do not exist here);
//   * forward: one workgroup = N waves over one TH x TW pixel tile; wave i owns row shift i, each
//     lane keeps N x PX accumulators (9 x 8 for D=4) in VGPRs, the x2 tile + halo and the x1 tile
//     are staged through LDS per channel chunk and read back as 128-bit rows, so one LDS float
//     feeds ~3 FMAs;
//   * backward: each lane keeps the N*N output gradients of its PX pixels in VGPRs for the whole
//     kernel, so gout is read from HBM exactly once per gradient; the other operand is staged
//     through the same LDS tile + halo.  gx2 is computed in gather form (no atomics):
//       gx2[c,q] = (1/C) sum_{i',j'} gout[(N-1-i')*N + (N-1-j')][q+(i'-D, j'-D)] * x1[c][q+(i'-D, j'-D)]
//     i.e. the gx1 loop with a flipped channel index and gout read at the shifted position.
// HBM traffic is the compulsory 4*px*(2C+N*N) bytes forward and 4*px*(N*N+4C) backward (both
// gradients); halo re-reads are served by L2.
#include "common.hpp"
#include "corr_v2.hpp"

namespace {

// Feature storage type of the tiled kernels below: float, or bf16 bit patterns (opt-in bf16
// STORAGE of the correlation inputs -- the reference's native path dispatches half as well,
// correlation_cuda_kernel.cu:352,369; arithmetic and outputs stay fp32).
typedef unsigned short bf16_t;
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return __uint_as_float((unsigned)v << 16); }

// ------------------------------------------------------------------------------------------------
// cooperative tile loader: CC channels of a (ROWS x COLS) window whose top-left pixel is (gy0,gx0),
// zero outside the image / past the last channel.  dst layout [CC][ROWS][PITCH].
// ------------------------------------------------------------------------------------------------
template <int CC, int ROWS, int COLS, int PITCH, int NT, typename TX>
__device__ __forceinline__ void load_tile(float* __restrict__ dst, const TX* __restrict__ src,
                                          int c0, int C, int H, int W, int gy0, int gx0) {
  constexpr int PER = ROWS * COLS;
  for (int idx = threadIdx.x; idx < CC * PER; idx += NT) {
    const int c = idx / PER;
    const int rem = idx - c * PER;
    const int r = rem / COLS;
    const int x = rem - r * COLS;
    const int gy = gy0 + r, gx = gx0 + x;
    float v = 0.f;
    if (c0 + c < C && gy >= 0 && gy < H && gx >= 0 && gx < W)
      v = to_f32(src[((long)(c0 + c) * H + gy) * W + gx]);
    dst[(c * ROWS + r) * PITCH + x] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// forward: wave i <-> row shift i
// ------------------------------------------------------------------------------------------------
template <int D, int PX, int TW, int CC>
struct FwdCfg {
  static constexpr int N = 2 * D + 1;
  static constexpr int LPR = TW / PX;    // lanes per tile row
  static constexpr int TH = 64 / LPR;    // tile rows
  static constexpr int R2 = TH + 2 * D;  // x2 rows incl. halo
  static constexpr int C2 = TW + 2 * D;  // x2 cols incl. halo
  static constexpr int P2 = (C2 + 3) & ~3;
  static constexpr int NT = 64 * N;
  static constexpr int LDS_FLOATS = CC * (TH * TW + R2 * P2);
};

template <int D, int PX, int TW, int CC, typename TX>
// two workgroups (2N waves) per CU: ceil(2N/4) waves per SIMD bounds the VGPR budget
__global__ __launch_bounds__(64 * (2 * D + 1), (2 * (2 * D + 1) + 3) / 4) void corr_fwd_kernel(
    const TX* __restrict__ x1, const TX* __restrict__ x2, float* __restrict__ out, int C, int H,
    int W, float inv_c, float slope) {
  using K = FwdCfg<D, PX, TW, CC>;
  constexpr int N = K::N, TH = K::TH, R2 = K::R2, P2 = K::P2;
  __shared__ __attribute__((aligned(16))) float lds[K::LDS_FLOATS];
  float* s1 = lds;                 // [CC][TH][TW]
  float* s2 = lds + CC * TH * TW;  // [CC][R2][P2]

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ry = lane / K::LPR, sx = (lane % K::LPR) * PX;
  const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH, b = blockIdx.z;
  const TX* x1b = x1 + (long)b * C * H * W;
  const TX* x2b = x2 + (long)b * C * H * W;

  float acc[N][PX];
#pragma unroll
  for (int j = 0; j < N; ++j)
#pragma unroll
    for (int p = 0; p < PX; ++p) acc[j][p] = 0.f;

  for (int c0 = 0; c0 < C; c0 += CC) {
    load_tile<CC, TH, TW, TW, K::NT, TX>(s1, x1b, c0, C, H, W, ty0, tx0);
    load_tile<CC, R2, K::C2, P2, K::NT, TX>(s2, x2b, c0, C, H, W, ty0 - D, tx0 - D);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < CC; ++c) {
      float a[PX], w[PX + 2 * D];
      const float* pa = s1 + (c * TH + ry) * TW + sx;
      const float* pw = s2 + (c * R2 + ry + wave) * P2 + sx;
      if constexpr (PX % 4 == 0) {
#pragma unroll
        for (int q = 0; q < PX / 4; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(pa + 4 * q);
          a[4 * q] = t.x, a[4 * q + 1] = t.y, a[4 * q + 2] = t.z, a[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int q = 0; q < (PX + 2 * D) / 4; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(pw + 4 * q);
          w[4 * q] = t.x, w[4 * q + 1] = t.y, w[4 * q + 2] = t.z, w[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int q = ((PX + 2 * D) / 4) * 4; q < PX + 2 * D; ++q) w[q] = pw[q];
      } else {
#pragma unroll
        for (int q = 0; q < PX; ++q) a[q] = pa[q];
#pragma unroll
        for (int q = 0; q < PX + 2 * D; ++q) w[q] = pw[q];
      }
#pragma unroll
      for (int j = 0; j < N; ++j)
#pragma unroll
        for (int p = 0; p < PX; ++p) acc[j][p] = fmaf(a[p], w[j + p], acc[j][p]);
    }
    __syncthreads();
  }

  const int gy = ty0 + ry, gx = tx0 + sx;
  if (gy >= H) return;
#pragma unroll
  for (int j = 0; j < N; ++j)
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      const float v = acc[j][p] * inv_c;
      acc[j][p] = v > 0.f ? v : v * slope;
    }
  float* ob = out + (((long)b * N * N + wave * N) * H + gy) * W + gx;
  const long cs = (long)H * W;
  if ((PX % 4 == 0) && (W % 4 == 0) && gx + PX <= W) {
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
      for (int q = 0; q < PX / 4; ++q)
        *reinterpret_cast<float4*>(ob + j * cs + 4 * q) =
            make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
  } else {
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
      for (int p = 0; p < PX; ++p)
        if (gx + p < W) ob[j * cs + p] = acc[j][p];
  }
}

template <int D, int PX, int TW, int CC, typename TX>
int launch_fwd(const TX* x1, const TX* x2, float* out, int B, int C, int H, int W, float slope,
               hipStream_t st) {
  using K = FwdCfg<D, PX, TW, CC>;
  dim3 grid(af_cdiv(W, TW), af_cdiv(H, K::TH), B);
  hipLaunchKernelGGL((corr_fwd_kernel<D, PX, TW, CC, TX>), grid, dim3(K::NT), 0, st, x1, x2, out, C, H, W,
                     1.0f / (float)C, slope);
  return af_launch_status();
}

// ------------------------------------------------------------------------------------------------
// backward: lane keeps gout[N*N][PX] in registers; MODE 0 -> gx1 (src = x2), MODE 1 -> gx2 (src = x1)
// ------------------------------------------------------------------------------------------------
template <int D, int PX, int TW, int CC, int NT>
struct BwdCfg {
  static constexpr int N = 2 * D + 1;
  static constexpr int LPR = TW / PX;
  static constexpr int TH = NT / LPR;
  static constexpr int R2 = TH + 2 * D;
  static constexpr int C2 = TW + 2 * D;
  static constexpr int P2 = (C2 + 3) & ~3;
  static constexpr int LDS_FLOATS = CC * R2 * P2;
};

template <int D, int PX, int TW, int CC, int NT, typename TX>
__global__ __launch_bounds__(NT, 2) void corr_bwd_kernel(const float* __restrict__ gout,
                                                     const float* __restrict__ fout, float slope,
                                                     const TX* __restrict__ x1,
                                                     const TX* __restrict__ x2,
                                                     float* __restrict__ gx1, float* __restrict__ gx2,
                                                     int B, int C, int H, int W, float inv_c,
                                                     int mode_base) {
  using K = BwdCfg<D, PX, TW, CC, NT>;
  constexpr int N = K::N, R2 = K::R2, P2 = K::P2;
  __shared__ __attribute__((aligned(16))) float s2[K::LDS_FLOATS];

  const int mode = mode_base + (int)(blockIdx.z / B);  // 0: gx1, 1: gx2
  const int b = blockIdx.z % B;
  const int ry = threadIdx.x / K::LPR, sx = (threadIdx.x % K::LPR) * PX;
  const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * K::TH;
  const int gy = ty0 + ry, gx = tx0 + sx;
  const TX* srcb = (mode == 0 ? x2 : x1) + (long)b * C * H * W;
  float* dstb = (mode == 0 ? gx1 : gx2) + (long)b * C * H * W;
  const float* gb = gout + (long)b * N * N * H * W;
  const float* fb = fout ? fout + (long)b * N * N * H * W : nullptr;
  const long cs = (long)H * W;
  auto gval = [&](long off) {  // output gradient through the fused LeakyReLU
    const float v = gb[off];
    return (fb && !(fb[off] > 0.f)) ? v * slope : v;
  };

  // output gradients of this lane's pixels, read once
  float g[N][N][PX];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (mode == 0) {
#pragma unroll
        for (int p = 0; p < PX; ++p)
          g[i][j][p] = (gy < H && gx + p < W) ? gval((i * N + j) * cs + (long)gy * W + gx + p) : 0.f;
      } else {
        const int yy = gy + i - D;
        const int ch = (N - 1 - i) * N + (N - 1 - j);
#pragma unroll
        for (int p = 0; p < PX; ++p) {
          const int xx = gx + p + j - D;
          g[i][j][p] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? gval(ch * cs + (long)yy * W + xx) : 0.f;
        }
      }
    }

  for (int c0 = 0; c0 < C; c0 += CC) {
    __syncthreads();
    load_tile<CC, R2, K::C2, P2, NT, TX>(s2, srcb, c0, C, H, W, ty0 - D, tx0 - D);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < CC; ++c) {
      float acc[PX];
#pragma unroll
      for (int p = 0; p < PX; ++p) acc[p] = 0.f;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        float w[PX + 2 * D];
        const float* pw = s2 + (c * R2 + ry + i) * P2 + sx;
        if constexpr (PX % 2 == 0) {
#pragma unroll
          for (int q = 0; q < (PX + 2 * D) / 2; ++q) {
            const float2 t = *reinterpret_cast<const float2*>(pw + 2 * q);
            w[2 * q] = t.x, w[2 * q + 1] = t.y;
          }
        } else {
#pragma unroll
          for (int q = 0; q < PX + 2 * D; ++q) w[q] = pw[q];
        }
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
          for (int p = 0; p < PX; ++p) acc[p] = fmaf(g[i][j][p], w[j + p], acc[p]);
      }
      if (c0 + c < C && gy < H) {
        float* o = dstb + (long)(c0 + c) * cs + (long)gy * W + gx;
#pragma unroll
        for (int p = 0; p < PX; ++p)
          if (gx + p < W) o[p] = acc[p] * inv_c;
      }
    }
  }
}

template <int D, int PX, int TW, int CC, int NT, typename TX>
int launch_bwd(const float* gout, const float* fout, float slope, const TX* x1, const TX* x2, float* gx1,
               float* gx2, int B, int C, int H, int W, hipStream_t st) {
  using K = BwdCfg<D, PX, TW, CC, NT>;
  const int nmodes = (gx1 ? 1 : 0) + (gx2 ? 1 : 0);
  if (nmodes == 0) return ARFLOW_OK;
  dim3 grid(af_cdiv(W, TW), af_cdiv(H, K::TH), B * nmodes);
  hipLaunchKernelGGL((corr_bwd_kernel<D, PX, TW, CC, NT, TX>), grid, dim3(NT), 0, st, gout, fout, slope, x1, x2, gx1, gx2, B,
                     C, H, W, 1.0f / (float)C, gx1 ? 0 : 1);
  return af_launch_status();
}

// ------------------------------------------------------------------------------------------------
// generic fall-back for max_disp > 4: one thread per output element, runtime D.  Not a hot path
// (every model in the reference uses max_displacement = 4, models/pwclite.py:124-126).
// ------------------------------------------------------------------------------------------------
__global__ void corr_fwd_generic(const float* __restrict__ x1, const float* __restrict__ x2,
                                 float* __restrict__ out, int B, int C, int H, int W, int D, float slope) {
  const int N = 2 * D + 1;
  const long total = (long)B * N * N * H * W;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % W;
    const int y = (idx / W) % H;
    const int ch = (idx / ((long)W * H)) % (N * N);
    const int b = idx / ((long)W * H * N * N);
    const int yy = y + ch / N - D, xx = x + ch % N - D;
    float s = 0.f;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W)
      for (int c = 0; c < C; ++c)
        s = fmaf(x1[(((long)b * C + c) * H + y) * W + x], x2[(((long)b * C + c) * H + yy) * W + xx], s);
    const float v = s / (float)C;
    out[idx] = v > 0.f ? v : v * slope;
  }
}

__global__ void corr_bwd_generic(const float* __restrict__ gout, const float* __restrict__ fout, float slope,
                                 const float* __restrict__ x1,
                                 const float* __restrict__ x2, float* __restrict__ gx1,
                                 float* __restrict__ gx2, int B, int C, int H, int W, int D) {
  const int N = 2 * D + 1;
  const long total = (long)B * C * H * W;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % W;
    const int y = (idx / W) % H;
    const int c = (idx / ((long)W * H)) % C;
    const int b = idx / ((long)W * H * C);
    const float* gb = gout + (long)b * N * N * H * W;
    const float* fb = fout ? fout + (long)b * N * N * H * W : nullptr;
    auto gval = [&](long off) {
      const float v = gb[off];
      return (fb && !(fb[off] > 0.f)) ? v * slope : v;
    };
    const float* p1 = x1 + ((long)b * C + c) * H * W;
    const float* p2 = x2 + ((long)b * C + c) * H * W;
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        const int ya = y + i - D, xa = x + j - D;  // gx1: x2 at the shifted position
        if (ya >= 0 && ya < H && xa >= 0 && xa < W)
          s1 = fmaf(gval(((long)(i * N + j) * H + y) * W + x), p2[(long)ya * W + xa], s1);
        const int yb = y - i + D, xb = x - j + D;  // gx2: gout and x1 at the un-shifted position
        if (yb >= 0 && yb < H && xb >= 0 && xb < W)
          s2 = fmaf(gval(((long)(i * N + j) * H + yb) * W + xb), p1[(long)yb * W + xb], s2);
      }
    if (gx1) gx1[idx] = s1 / (float)C;
    if (gx2) gx2[idx] = s2 / (float)C;
  }
}

template <int D, typename TX>
int dispatch_fwd(const TX* x1, const TX* x2, float* out, int B, int C, int H, int W, float slope,
                 hipStream_t st) {
  // pick the widest lane strip that still yields enough workgroups to occupy the chip
  const long px = (long)B * H * W;
  if (W >= 24 && px >= 32768) return launch_fwd<D, 8, 32, 8, TX>(x1, x2, out, B, C, H, W, slope, st);
  if (W >= 12 && px >= 8192) return launch_fwd<D, 4, 16, 8, TX>(x1, x2, out, B, C, H, W, slope, st);
  return launch_fwd<D, 2, 16, 8, TX>(x1, x2, out, B, C, H, W, slope, st);
}

template <int D, typename TX>
int dispatch_bwd(const float* gout, const float* fout, float slope, const TX* x1, const TX* x2, float* gx1,
                 float* gx2, int B, int C, int H, int W, hipStream_t st) {
  const long px = (long)B * H * W;
  if (W >= 48 && px >= 65536) return launch_bwd<D, 2, 64, 8, 256, TX>(gout, fout, slope, x1, x2, gx1, gx2, B, C, H, W, st);
  if (W >= 24 && px >= 8192) return launch_bwd<D, 2, 32, 8, 128, TX>(gout, fout, slope, x1, x2, gx1, gx2, B, C, H, W, st);
  return launch_bwd<D, 1, 16, 8, 64, TX>(gout, fout, slope, x1, x2, gx1, gx2, B, C, H, W, st);
}

}  // namespace

extern "C" int arflow_corr_sign_planes(int C, int W, int max_disp) {
  return corr_v2::eligible(C, W, max_disp) ? corr_v2::NW : 0;
}

static int corr_fwd_impl(const float* x1, const float* x2, float* out, long out_bstride, unsigned* sign_bits, int B,
                         int C, int H, int W, int max_disp, float negative_slope, arflow_stream_t stream);
extern "C" int arflow_corr_fwd(const float* x1, const float* x2, float* out, unsigned* sign_bits, int B, int C,
                               int H, int W, int max_disp, float negative_slope, arflow_stream_t stream) {
  return corr_fwd_impl(x1, x2, out, 0, sign_bits, B, C, H, W, max_disp, negative_slope, stream);
}
extern "C" int arflow_corr_strided_supported(int C, int W, int max_disp) { return corr_v2::eligible(C, W, max_disp) ? 1 : 0; }
extern "C" int arflow_corr_fwd_strided(const float* x1, const float* x2, float* out, long out_bstride,
                                       unsigned* sign_bits, int B, int C, int H, int W, int max_disp,
                                       float negative_slope, arflow_stream_t stream) {
  AF_REQUIRE(corr_v2::eligible(C, W, max_disp), ARFLOW_EPARAM);
  AF_REQUIRE(out_bstride >= (long)(2 * max_disp + 1) * (2 * max_disp + 1) * H * W && out_bstride % 4 == 0, ARFLOW_ESHAPE);
  return corr_fwd_impl(x1, x2, out, out_bstride, sign_bits, B, C, H, W, max_disp, negative_slope, stream);
}
static int corr_fwd_impl(const float* x1, const float* x2, float* out, long out_bstride, unsigned* sign_bits, int B,
                         int C, int H, int W, int max_disp, float negative_slope, arflow_stream_t stream) {
  af_clear_stale_error();
  const float slope = negative_slope;
  AF_REQUIRE_PTR(x1);
  AF_REQUIRE_PTR(x2);
  AF_REQUIRE_PTR(out);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, ARFLOW_ESHAPE);
  AF_REQUIRE(max_disp >= 1, ARFLOW_EPARAM);
  AF_REQUIRE(B <= 65535, ARFLOW_ESHAPE);
  hipStream_t st = (hipStream_t)stream;
  if (corr_v2::eligible(C, W, max_disp)) return corr_v2::launch_fwd(x1, x2, out, sign_bits, B, C, H, W, slope, st, out_bstride);
  AF_REQUIRE(sign_bits == nullptr, ARFLOW_EPARAM);  // only the fast path records signs
  switch (max_disp) {
    case 1: return dispatch_fwd<1, float>(x1, x2, out, B, C, H, W, slope, st);
    case 2: return dispatch_fwd<2, float>(x1, x2, out, B, C, H, W, slope, st);
    case 3: return dispatch_fwd<3, float>(x1, x2, out, B, C, H, W, slope, st);
    case 4: return dispatch_fwd<4, float>(x1, x2, out, B, C, H, W, slope, st);
    default: {
      const int N = 2 * max_disp + 1;
      const long total = (long)B * N * N * H * W;
      hipLaunchKernelGGL(corr_fwd_generic, dim3((unsigned)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256)),
                         dim3(256), 0, st, x1, x2, out, B, C, H, W, max_disp, slope);
      return af_launch_status();
    }
  }
}

// internal launchers for the level entry points (level.hip)
int af_level_corr_fwd_launch(const float* x1, const float* x2w, const double* acc, int acc_rows, int norm_mode, float* out,
                             long out_bstride, float* x1n, long x1n_bstride, unsigned* sign_bits, float* stats, int B, int C,
                             int H, int W, float negative_slope, hipStream_t st, const double* r1, int n1, const double* r2,
                             int n2) {
  const corr_v2::NormArgs na{acc, acc_rows, norm_mode, x1n, x1n_bstride, stats, r1, n1, r2, n2};
  return corr_v2::launch_fwd(x1, x2w, out, sign_bits, B, C, H, W, negative_slope, st, out_bstride, &na);
}
int af_level_corr_bwd_launch(const float* gout, long gout_bstride, const unsigned* sign_bits, const float* x1n,
                             long x1n_bstride, const float* x2w, const float* stats, float* gx1n, float* gx2n, int B, int C,
                             int H, int W, float negative_slope, hipStream_t st, float* zero_c, float* zero_f,
                             float* zero_fc, int* zero_i) {
  return corr_v2::launch_bwd(gout, nullptr, negative_slope == 1.0f ? nullptr : sign_bits, negative_slope, x1n, x2w, gx1n,
                             gx2n, B, C, H, W, st, gout_bstride, 0, x1n_bstride, stats, zero_c, zero_f, zero_fc, zero_i);
}
extern "C" int arflow_level_supported(int C, int W, int max_disp) { return corr_v2::eligible(C, W, max_disp) ? 1 : 0; }

// Level forward, second launch: cost volume of the NORMALISED pair straight from the raw maps
// (corr_v2::fwd_kernel<.., NORM = true>): volume (+ fused LeakyReLU and sign words) into `out`, the normalised first
// map into `x1n` (both with a batch stride: slots of the decoder's concatenated input), the statistics into `stats`.
extern "C" int arflow_level_corr_fwd(const float* x1, const float* x2w, const double* acc, int acc_rows, int norm_mode,
                                     float* out, long out_bstride, float* x1n, long x1n_bstride, unsigned* sign_bits,
                                     float* stats, int B, int C, int H, int W, int max_disp, float negative_slope,
                                     arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(x1);
  AF_REQUIRE_PTR(x2w);
  AF_REQUIRE_PTR(acc);
  AF_REQUIRE_PTR(out);
  AF_REQUIRE_PTR(stats);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && B <= 65535 && acc_rows >= 0, ARFLOW_ESHAPE);
  AF_REQUIRE(corr_v2::eligible(C, W, max_disp), ARFLOW_EPARAM);
  AF_REQUIRE(norm_mode == ARFLOW_FEATNORM_JOINT || norm_mode == ARFLOW_FEATNORM_AVG, ARFLOW_EPARAM);
  const long vol = (long)corr_v2::N * corr_v2::N * H * W;
  AF_REQUIRE(out_bstride >= vol && out_bstride % 4 == 0, ARFLOW_ESHAPE);
  AF_REQUIRE(x1n == nullptr || (x1n_bstride >= (long)C * H * W && x1n_bstride % 4 == 0), ARFLOW_ESHAPE);
  const corr_v2::NormArgs na{acc, acc_rows, norm_mode, x1n, x1n_bstride, stats, nullptr, 0, nullptr, 0};
  return corr_v2::launch_fwd(x1, x2w, out, sign_bits, B, C, H, W, negative_slope, (hipStream_t)stream, out_bstride, &na);
}

// Level backward, first launch: gradients w.r.t. the NORMALISED maps from the raw second map and the saved normalised
// first map (corr_v2::bwd_kernel<.., NORM = true>).  gx1n / gx2n: [B,C,H,W] contiguous.
extern "C" int arflow_level_corr_bwd(const float* gout, long gout_bstride, const unsigned* sign_bits, const float* x1n,
                                     long x1n_bstride, const float* x2w, const float* stats, float* gx1n, float* gx2n,
                                     int B, int C, int H, int W, int max_disp, float negative_slope,
                                     arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(gout);
  AF_REQUIRE_PTR(x1n);
  AF_REQUIRE_PTR(x2w);
  AF_REQUIRE_PTR(stats);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && B <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(corr_v2::eligible(C, W, max_disp), ARFLOW_EPARAM);
  AF_REQUIRE(negative_slope == 1.0f || sign_bits != nullptr, ARFLOW_ENULL);
  const long vol = (long)corr_v2::N * corr_v2::N * H * W;
  AF_REQUIRE(gout_bstride >= vol && gout_bstride % 4 == 0, ARFLOW_ESHAPE);
  AF_REQUIRE(x1n_bstride >= (long)C * H * W && x1n_bstride % 4 == 0, ARFLOW_ESHAPE);
  return af_level_corr_bwd_launch(gout, gout_bstride, sign_bits, x1n, x1n_bstride, x2w, stats, gx1n, gx2n, B, C, H, W,
                                  negative_slope, (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr);
}

static int corr_bwd_impl(const float* gout, long gout_bstride, const float* out, long out_bstride,
                         const unsigned* sign_bits, const float* x1, const float* x2, float* gx1, float* gx2, int B, int C,
                         int H, int W, int max_disp, float negative_slope, arflow_stream_t stream);
extern "C" int arflow_corr_bwd(const float* gout, const float* out, const unsigned* sign_bits, const float* x1,
                               const float* x2, float* gx1, float* gx2, int B, int C, int H, int W, int max_disp,
                               float negative_slope, arflow_stream_t stream) {
  return corr_bwd_impl(gout, 0, out, 0, sign_bits, x1, x2, gx1, gx2, B, C, H, W, max_disp, negative_slope, stream);
}
extern "C" int arflow_corr_bwd_strided(const float* gout, long gout_bstride, const float* out, long out_bstride,
                                       const unsigned* sign_bits, const float* x1, const float* x2, float* gx1,
                                       float* gx2, int B, int C, int H, int W, int max_disp, float negative_slope,
                                       arflow_stream_t stream) {
  AF_REQUIRE(corr_v2::eligible(C, W, max_disp), ARFLOW_EPARAM);
  const long vol = (long)(2 * max_disp + 1) * (2 * max_disp + 1) * H * W;
  AF_REQUIRE(gout_bstride >= vol && gout_bstride % 4 == 0, ARFLOW_ESHAPE);
  AF_REQUIRE(out == nullptr || (out_bstride >= vol && out_bstride % 4 == 0), ARFLOW_ESHAPE);
  return corr_bwd_impl(gout, gout_bstride, out, out_bstride, sign_bits, x1, x2, gx1, gx2, B, C, H, W, max_disp,
                       negative_slope, stream);
}
static int corr_bwd_impl(const float* gout, long gout_bstride, const float* out, long out_bstride,
                         const unsigned* sign_bits, const float* x1, const float* x2, float* gx1, float* gx2, int B, int C,
                         int H, int W, int max_disp, float negative_slope, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(gout);
  const float slope = negative_slope;
  const bool act = negative_slope != 1.0f;
  const bool fast = corr_v2::eligible(C, W, max_disp);
  if (!(act && fast)) sign_bits = nullptr;  // the generic kernels read `out`
  const float* fout = (act && !sign_bits) ? out : nullptr;
  if (act && !sign_bits) {
    AF_REQUIRE_PTR(out);
    AF_REQUIRE(negative_slope >= 0.f, ARFLOW_EPARAM);  // out > 0 <=> pre-activation > 0 only then
  }
  AF_REQUIRE_PTR(x1);
  AF_REQUIRE_PTR(x2);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, ARFLOW_ESHAPE);
  AF_REQUIRE(max_disp >= 1, ARFLOW_EPARAM);
  AF_REQUIRE(2 * B <= 65535, ARFLOW_ESHAPE);
  hipStream_t st = (hipStream_t)stream;
  if (fast) return corr_v2::launch_bwd(gout, fout, sign_bits, slope, x1, x2, gx1, gx2, B, C, H, W, st, gout_bstride, out_bstride);
  switch (max_disp) {
    case 1: return dispatch_bwd<1, float>(gout, fout, slope, x1, x2, gx1, gx2, B, C, H, W, st);
    case 2: return dispatch_bwd<2, float>(gout, fout, slope, x1, x2, gx1, gx2, B, C, H, W, st);
    case 3: return dispatch_bwd<3, float>(gout, fout, slope, x1, x2, gx1, gx2, B, C, H, W, st);
    case 4: return dispatch_bwd<4, float>(gout, fout, slope, x1, x2, gx1, gx2, B, C, H, W, st);
    default: {
      const long total = (long)B * C * H * W;
      hipLaunchKernelGGL(corr_bwd_generic, dim3((unsigned)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256)),
                         dim3(256), 0, st, gout, fout, slope, x1, x2, gx1, gx2, B, C, H, W, max_disp);
      return af_launch_status();
    }
  }
}

// ---- bf16 STORAGE of the features (opt-in): x1, x2 hold bf16 bit patterns, the products are
// accumulated in fp32, the volume and both gradients are fp32.  The LDS-tiled kernels above with a converting loader.
extern "C" int arflow_corr_fwd_bf16(const unsigned short* x1, const unsigned short* x2, float* out, int B, int C, int H,
                                    int W, int max_disp, float negative_slope, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(x1);
  AF_REQUIRE_PTR(x2);
  AF_REQUIRE_PTR(out);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && B <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(max_disp == 4, ARFLOW_EPARAM);  // the only displacement the reference's models use
  return dispatch_fwd<4, bf16_t>(x1, x2, out, B, C, H, W, negative_slope, (hipStream_t)stream);
}

extern "C" int arflow_corr_bwd_bf16(const float* gout, const float* out, const unsigned short* x1,
                                    const unsigned short* x2, float* gx1, float* gx2, int B, int C, int H, int W,
                                    int max_disp, float negative_slope, arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(gout);
  AF_REQUIRE_PTR(x1);
  AF_REQUIRE_PTR(x2);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && 2 * B <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(max_disp == 4, ARFLOW_EPARAM);
  const bool act = negative_slope != 1.0f;
  if (act) {
    AF_REQUIRE_PTR(out);  // LeakyReLU derivative from the sign of the forward output
    AF_REQUIRE(negative_slope >= 0.f, ARFLOW_EPARAM);
  }
  return dispatch_bwd<4, bf16_t>(gout, act ? out : nullptr, negative_slope, x1, x2, gx1, gx2, B, C, H, W,
                                 (hipStream_t)stream);
}
