// Shared device/host helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/arflow_hip.h"

#define AF_WAVE 64

#define AF_REQUIRE_PTR(p) \
  do {                    \
    if ((p) == nullptr) return ARFLOW_ENULL; \
  } while (0)
#define AF_REQUIRE(cond, code) \
  do {                         \
    if (!(cond)) return (code); \
  } while (0)

// hipGetLastError() reports (and clears) the last error of ANY runtime call on this host thread.  An entry
// point must not blame its own launches for an error that was already pending when it was called, and must not
// silently drop that error either: the pending code is moved into a process-wide slot that the host can read
// (and clear) with arflow_take_stale_error(), and the first one is reported once on stderr.
extern "C" int arflow_take_stale_error(void);
void af_record_stale_error(int code);  // api.hip
static inline void af_clear_stale_error() {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) af_record_stale_error((int)e);
}
// after every launch of a multi-launch entry point except the last (HIP keeps only the LAST call's status)
#define AF_LAUNCH_CHECK()                \
  do {                                   \
    const int rc_ = af_launch_status();  \
    if (rc_ != ARFLOW_OK) return rc_;    \
  } while (0)

static inline int af_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ARFLOW_OK : (ARFLOW_ELAUNCH_BASE - (int)e);
}
static inline int af_hip_status(hipError_t e) {
  return e == hipSuccess ? ARFLOW_OK : (ARFLOW_ELAUNCH_BASE - (int)e);
}
static inline int af_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Launch-shape helpers shared by translation units (the level entry points size one accumulator for either producer).
// Few tiles (coarse pyramid levels): the 4-channel chunks of a tile are spread over up to C/4 workgroups (gridDim.y)
// until ~2048 workgroups are in flight, instead of one workgroup walking all channels serially.
static inline unsigned af_channel_split(long tiles, int C) {
  unsigned n = 1;
  while ((long)n * 2 <= C / 4 && tiles * n * 2 <= 2048) n *= 2;
  return n;
}
// featnorm reduction passes: enough workgroups to fill the chip (~2048) without slicing a sample finer than one trip per block
static inline unsigned af_blocks_per_sample(int B, long n, int floats_per_block) {
  long nb = (n + floats_per_block - 1) / floats_per_block;
  const long want = (2048 + B - 1) / B;
  if (nb > want) nb = want;
  return (unsigned)(nb < 1 ? 1 : nb);
}

// Sum over the 64 lanes of a wave; every lane gets the total.
__device__ __forceinline__ float af_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, AF_WAVE);
  return v;
}

// Block-wide sum of NV values per thread, result valid in thread 0.  `scratch` needs
// NV * (blockDim/64) floats of LDS.  All threads must call.
template <int NV>
__device__ __forceinline__ void af_block_sum(float (&v)[NV], float* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = af_wave_sum(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) scratch[k * nw + wave] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      float s = 0.f;
      for (int w = 0; w < nw; ++w) s += scratch[k * nw + w];
      v[k] = s;
    }
  }
}

// Reduction outputs ("sums"): every workgroup STORES its partial sums into its own row of ARFLOW_SUM_COLS floats
// (row = linear workgroup id) and zero-fills the rows beyond the grid, so after the launch every one of the `nrows`
// rows is defined and the caller adds them all -- no zero-fill launch, no atomics, and a result that does not depend
// on the order workgroups finish in.  (Round 1: hipMemsetAsync + atomicAdd into 256 slotted rows -- the memset was
// a separate ~4 us GPU operation in front of every reduction.)  nrows >= number of workgroups, host-checked.
__device__ __forceinline__ void af_store_partial(float* __restrict__ sums, int nrows, float a, float b, float c,
                                                 float d = 0.f) {
  const int nwg = gridDim.x * gridDim.y * gridDim.z;
  const int w = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  *reinterpret_cast<float4*>(sums + (long)w * ARFLOW_SUM_COLS) = make_float4(a, b, c, d);
  for (int r = w + nwg; r < nrows; r += nwg)
    *reinterpret_cast<float4*>(sums + (long)r * ARFLOW_SUM_COLS) = make_float4(0.f, 0.f, 0.f, 0.f);
}
int af_sums_rows(int B, int H, int W);  // api.hip: rows every `sums` buffer has (>= the grid of any reduction kernel)

// XCD-aware workgroup -> tile map for 1-D grids of 2-D tiles.  Workgroups are dealt round-robin over the
// 8 XCDs (private L2 each), so consecutive ids would spread neighbouring tiles -- which share halo rows
// and bilinear taps -- over different L2s (measured: 2.6x fabric read over-fetch in the warp forward).
// Give every XCD a contiguous run of tiles: id b runs tile (b % 8) * ceil(T/8) + b / 8.  Pure speed.
// Returns false for the padding ids of the rounded-up grid (launch with af_grid_for_tiles(T) blocks).
// bx: the workgroup's index in the 1-D tile grid (blockIdx.x, or what a role-fused kernel derives from it).
__device__ __forceinline__ bool af_tile_of_block(int ntx, int nty, int nimg, int& tx, int& ty, int& img, unsigned bx) {
  const int T = ntx * nty * nimg;
  const int per = (T + 7) >> 3;
  const int t = (int)(bx & 7) * per + (int)(bx >> 3);
  if ((int)(bx >> 3) >= per || t >= T) return false;
  tx = t % ntx;
  ty = (t / ntx) % nty;
  img = t / (ntx * nty);
  return true;
}
__device__ __forceinline__ bool af_tile_of_block(int ntx, int nty, int nimg, int& tx, int& ty, int& img) {
  return af_tile_of_block(ntx, nty, nimg, tx, ty, img, blockIdx.x);
}
static inline unsigned af_grid_for_tiles(long T) { return 8u * (unsigned)((T + 7) / 8); }

// torch grid_sample un-normalisation (ATen/native/GridSampler.h:27-36).
__device__ __forceinline__ float af_unnormalize(float g, int size, bool align) {
  return align ? ((g + 1.f) / 2.f) * (float)(size - 1) : ((g + 1.f) * (float)size - 1.f) / 2.f;
}

// Pixel coordinate sampled for output position p with displacement u, reproducing the reference's
// fp32 normalise -> un-normalise round trip.
//   norm 0 (ARFlow flow_warp, utils/warp_utils.py:16-23): g = 2*(p+u)/(n_flow-1) - 1, then
//           grid_sample's un-normalisation with the SOURCE size and the align flag;
//   norm 1 (UFlow resample, utils/uflow_utils.py:71-76): g = 2*(p+u)/max(n_src-1,1) - 1, align=True;
//   norm 2: as 1, but `u` is the absolute coordinate (resample(source, coords) called directly).
// *dcoord receives d(coordinate)/d(u).
// x / den for an INTEGER-valued divisor den >= 1 (the image extents minus one), bit-identical to the IEEE division
// wherever the quotient matters: r = RN(1/den), q = RN(x r), q' = fma(fma(-q, den, x), r, q) -- 3 VALU instructions
// instead of the ~11 of v_div_scale / v_rcp / 4 x v_fma / v_div_fmas / v_div_fixup (the reciprocal is wave-uniform), and
// no branch (the IEEE sequence's special-case handling split every tile-fill loop into basic blocks).  Correctly rounded
// by Markstein's theorem (r correctly rounded, q faithful, den's significand not all ones -- no integer below 2^24 - 1
// is); tools/ubench/div_exact.hip checks EVERY float |x| <= 2^15 against EVERY divisor 1 .. 16384 on the GPU: 0
// mismatches with |x / den| >= 1e-30.  Below that (signed zeros, denormal quotients: the residual underflows) the callers'
// `- 1.0f` absorbs the quotient entirely.
__device__ __forceinline__ float af_div_den(float x, float den) {
  const float r = 1.0f / den;
  const float q = x * r;
  return fmaf(fmaf(-q, den, x), r, q);
}

__device__ __forceinline__ float af_sample_coord(float p, float u, int n_flow, int n_src, int norm,
                                                 bool align, float* dcoord) {
  if (norm != ARFLOW_NORM_ARFLOW) {
    const float den = (float)(n_src - 1 > 1 ? n_src - 1 : 1);
    const float pos = norm == ARFLOW_NORM_UFLOW_ABS ? u : p + u;  // ABS: `u` already is the coordinate
    const float g = af_div_den(2.0f * pos, den) - 1.0f;
    *dcoord = (2.0f / den) * ((float)(n_src - 1) / 2.f);
    return ((g + 1.f) / 2.f) * (float)(n_src - 1);
  }
  const float den = (float)(n_flow - 1);
  const float g = (n_flow > 1 ? af_div_den(2.0f * (p + u), den) : 2.0f * (p + u) / den) - 1.0f;
  *dcoord = (2.0f / den) * (align ? (float)(n_src - 1) / 2.f : (float)n_src / 2.f);
  return af_unnormalize(g, n_src, align);
}

// border padding: clamp to [0,size-1]; the coordinate gradient is zeroed at/over the border
// (ATen/native/GridSampler.h:58-83).
__device__ __forceinline__ float af_clip_border(float c, int size, float* dmul) {
  if (c <= 0.f) {
    *dmul = 0.f;
    return 0.f;
  }
  const float mx = (float)(size - 1);
  if (c >= mx) {
    *dmul = 0.f;
    return mx;
  }
  return c;
}
