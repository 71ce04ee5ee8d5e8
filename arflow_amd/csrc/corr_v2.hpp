// Fast path of the cost-volume kernels for the configuration every model uses (max_disp = 4) when
// W % 4 == 0 and C % 4 == 0.  See corr.hip for the arithmetic and the generic kernels.
//
// One workgroup = 3 waves over one 8 x 32 pixel tile.  Wave w owns the row shifts i = 3w..3w+2; a
// lane owns 4 consecutive pixels of one tile row.
//   * staging: the operand tile (+4 px halo: 16 rows x 40 floats) of 4 channels is brought in by
//     LDS-DMA (global_load_lds_dwordx4: per-lane global address, linear LDS destination) into a ring
//     of NBUF buffers, issued NBUF-1 chunks ahead of the FMAs -- no staging VGPRs, one barrier per
//     chunk, counted s_waitcnt vmcnt(N) so younger chunks stay in flight across the barrier.
//     Pieces outside the image are sourced from a 16-byte block of zeros in device memory, so every
//     lane of every DMA instruction is active (exact instruction counts, no LDS pre-fill).
//   * the lane -> (pixel group, row) map is a bit permutation chosen so that every ds_read_b128 of a
//     12-float window row is bank-conflict free with the native 40-float pitch (searched offline
//     against the gfx950 b128 lane groups); SQ_LDS_BANK_CONFLICT = 0 measured.
//   * forward: 3 x 9 x 4 accumulators per lane; per channel 1 + 9 b128 LDS reads feed 108 FMAs.
//   * backward: the lane keeps its 3 x 9 x 4 output gradients in VGPRs for the whole kernel (gout is
//     read once); per channel it produces a 4-pixel partial sum over its 27 displacements, the three
//     waves' partials meet in LDS and are summed + stored by the workgroup.
//   * workgroup -> tile order is XCD-aware (tiles sharing halos share an L2).
#pragma once
#include "common.hpp"
#include "featnorm_stats.hpp"

namespace corr_v2 {

constexpr int D = 4, N = 9, PX = 4, TW = 32, TH = 8, CC = 4, NW = 3, NT = 64 * NW;
constexpr int SR = TH + 2 * D;           // 16 staged rows
constexpr int SP = TW + 2 * D;           // 40 floats: row pitch of both staged tiles
constexpr int SRC_FLOATS = CC * SR * SP;  // 2560 = 10 wave-wide DMA instructions
constexpr int X1_FLOATS = CC * TH * SP;   // 1280 = 5  (32 of 40 floats used: pitch kept for banking)
constexpr int SRC_DMA = SRC_FLOATS / 256, X1_DMA = X1_FLOATS / 256;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // 4-byte aligned: one global_load_dwordx4 all the same

// 16 zero bytes in device memory: DMA source of every piece outside the image.
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void lane_xy(int lane, int& xg, int& y) {
  xg = ((lane >> 2) & 1) | (((lane >> 3) & 1) << 1) | ((lane & 1) << 2);
  y = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 1) & 1) << 2);
}

__device__ __forceinline__ void dma16(const float* gsrc, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Element offset (inside one [C,H,W] image, relative to channel c0) of the 16-byte piece this lane
// moves in wave-wide DMA instruction `k` of a staged region with `rows` rows per channel, or -1 when
// the piece lies outside the image / in the pad columns.
__device__ __forceinline__ int piece_offset(int k, int lane, int rows, int used_slots, int gy0, int gx0,
                                            int H, int W) {
  const int s = k * 64 + lane;
  const int per_c = rows * (SP / 4);
  const int c = s / per_c, rem = s - c * per_c;
  const int r = rem / (SP / 4), xs = rem - r * (SP / 4);
  const int gy = gy0 + r, gx = gx0 + 4 * xs;
  if (xs >= used_slots || gy < 0 || gy >= H || gx < 0 || gx >= W) return -1;
  return (c * H + gy) * W + gx;
}

// wait until at most N_OUT of this wave's vector-memory operations are outstanding, then barrier.
// Raw s_barrier: __syncthreads() would drain every in-flight DMA (vmcnt(0)).
template <int N_OUT>
__device__ __forceinline__ void wait_dma_and_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N_OUT) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// A 12-float window row is read as THREE ds_read_b128, issued together one step ahead of their use
// (issue_window) and unpacked after the FMAs of the current step (land_window).  The empty asm in
// land_window pins each read as ONE ds_read_b128: left alone, hipcc re-splits the window into
// ds_read2_b32 pairs (to feed v_pk_fma_f32 operands at odd offsets), which bank-conflict (32-bank mode)
// and double the LDS instruction count (measured 57 % of LDS cycles lost in the backward kernel).  The pin
// sits at the point of USE, not at the load: pinned at the load, every read was followed by its own
// s_waitcnt lgkmcnt(0) -- ten serialised LDS round trips per channel.
__device__ __forceinline__ void issue_window(const float* row, f32x4 (&t)[3]) {
#pragma unroll
  for (int q = 0; q < 3; ++q) t[q] = *reinterpret_cast<const f32x4*>(row + 4 * q);
}
__device__ __forceinline__ void land_window(f32x4 (&t)[3]) {
  asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]));
}
#define CORR_WIN(t, i) ((t)[(i) >> 2][(i) & 3])  // element i (compile-time) of a landed window
__device__ __forceinline__ void load_vec4(const float* p, float (&v)[PX]) {
  f32x4 t = *reinterpret_cast<const f32x4*>(p);
  asm volatile("" : "+v"(t));
  v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
}

// Workgroup -> tile map.  Workgroups are dealt round-robin over the 8 XCDs (each with a private L2), so
// consecutive ids would put neighbouring tiles -- which share their 4-pixel halos -- on different L2s.
// Give every XCD a contiguous run of tiles instead: id b runs tile (b % 8) * ceil(T/8) + b / 8.  Pure
// speed: any placement is correct.  Returns false for the padding ids of the rounded-up grid.
__device__ __forceinline__ bool tile_of_block(int ntx, int nty, int nimg, int& tx, int& ty, int& img) {
  const int T = ntx * nty * nimg;
  const int per = (T + 7) >> 3;
  const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || t >= T) return false;
  tx = t % ntx;
  ty = (t / ntx) % nty;
  img = t / (ntx * nty);
  return true;
}
inline unsigned grid_for_tiles(int T) { return 8u * (unsigned)((T + 7) / 8); }

// ------------------------------------------------------------------------------------------------
// G > 1 (coarse levels with many channels, e.g. PWCLite's 6x10 level with C = 192: 16 tiles x 48 serial
// chunks left the chip idle for 66 us): the workgroup has G groups of 3 waves; group g runs the channel
// chunks g, g+G, ... through its own DMA ring, and the groups' accumulators meet through LDS at the end
// (float atomics into the volume were 6x slower than no split at all).  Needs (C / 4) % G == 0; one such
// workgroup (G = 4: 12 waves, 124 KB of LDS) fills a CU.
// NORM (level kernels): the feature normalisation in front of the cost volume
// (normalize_features, models/pwclite_uflow.py:30-38 / models/uflow_model.py:8-50) folded into this launch.  x1 and x2
// are the RAW first map and the raw (warped) second map; the per-sample (mu, sigma) come from the partial-moment rows
// the warp launch (or the moment pass) left in `acc`.  With x1n = (x1 - mu) / sigma formed in registers,
//     corr(x1n, (x2 - mu) / sigma)[p, d] = (sum_c x1n[c,p] x2[c,p+d]  -  mu * sum_c x1n[c,p]) / (C sigma)
// for displacements inside the image and 0 outside (models/correlation_native.py:13-23 zero-pads the NORMALISED map),
// so the second normalised map is never formed and the first one is written once (the decoder concatenates it,
// models/pwclite_uflow.py:218-222) from the registers that feed the FMAs: no moment pass, no apply pass.
struct NormArgs {
  const double* acc;  // [B][nrows][4] partial (sum x1, sum x1^2, sum x2, sum x2^2)
  int nrows, mode;    // ARFLOW_FEATNORM_JOINT / _AVG
  float* x1n;         // normalised first map out (batch stride x1n_bs floats), may be null
  long x1n_bs;
  float* stats;       // [B][4] (m1, m2, mu, sigma) for the backward
  const double* r1;   // optional [B][n1][2] rows of the first map's moments (conv epilogue), replacing acc's columns 0, 1
  int n1;
  const double* r2;   // the same for the second map (level without a warp)
  int n2;
};

template <int NBUF, int G, bool NORM = false>
// (the 4-buffer ring of the few-tiles form needs 2 waves/SIMD worth of registers: asking for 3 only earned a
// "failed to meet occupancy target" remark)
__global__ __launch_bounds__(NT * G, (NBUF == 4 && G == 1) ? 2 : 3) void fwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                        float* __restrict__ out, unsigned* __restrict__ sign_bits,
                                                        int nimg, int C, int H, int W, float inv_c, float slope,
                                                        long obs /* floats between the volumes of two samples */,
                                                        NormArgs na) {
  constexpr int BUF = SRC_FLOATS + X1_FLOATS, RING = NBUF * BUF;
  static_assert(G == 1 || G * RING >= NW * N * 64 * PX, "the accumulator exchange reuses the rings");
  __shared__ __attribute__((aligned(16))) float lds_all[G * RING + TH * SP];  // + pad: prefetch runs a channel ahead
  const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // wave id in an SGPR
  const int grp = wave_all / NW, wave = wave_all - grp * NW;
  float* lds = lds_all + grp * RING;
  int xg, y;
  lane_xy(lane, xg, y);
  int btx, bty, b;
  if (!tile_of_block((W + TW - 1) / TW, (H + TH - 1) / TH, nimg, btx, bty, b)) return;
  const int tx0 = btx * TW, ty0 = bty * TH;
  const long cs = (long)H * W;
  const float* x1b = x1 + (long)b * C * cs;
  const float* x2b = x2 + (long)b * C * cs;

  // this wave issues DMA instructions k = wave, wave+3, ... of the 15 per chunk (10 x2 + 5 x1)
  constexpr int NK = (SRC_DMA + X1_DMA) / NW;  // 5
  int off[NK];
#pragma unroll
  for (int m = 0; m < NK; ++m) {
    const int k = wave + NW * m;
    off[m] = k < SRC_DMA ? piece_offset(k, lane, SR, SP / 4, ty0 - D, tx0 - D, H, W)
                         : piece_offset(k - SRC_DMA, lane, TH, TW / 4, ty0, tx0, H, W);
  }
  auto issue = [&](int chunk) {
    float* buf = lds + (chunk % NBUF) * BUF;
#pragma unroll
    for (int m = 0; m < NK; ++m) {
      const int k = wave + NW * m;
      const float* g = (k < SRC_DMA ? x2b : x1b) + (long)(grp + chunk * G) * CC * cs;
      dma16(off[m] >= 0 ? g + off[m] : g_zero16, buf + k * 256);  // x1 region follows x2: block k carries over
    }
  };

  const int nchunk = C / CC / G;  // per group
#pragma unroll
  for (int pre = 0; pre < NBUF - 1; ++pre)
    if (pre < nchunk) issue(pre);

  float acc[3][N][PX];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
      for (int p = 0; p < PX; ++p) acc[k][j][p] = 0.f;

  // NORM: the sample's statistics while the first chunks are in flight (every wave adds the rows itself: no LDS)
  float mu = 0.f, rs = 1.f;
  float s1n[PX] = {0.f, 0.f, 0.f, 0.f};  // sum over channels of the normalised first map at the lane's 4 pixels
  const int gy = ty0 + y, gx = tx0 + 4 * xg;
  const bool lane_in = gy < H && gx < W;
  float* x1n_p = nullptr;
  if constexpr (NORM) {
    featnorm::Moments m;
    m.m1 = m.m2 = m.mu = 0.f, m.var = 1.f;
    if (na.nrows > 0 || na.r1)
      m = featnorm::moments_of(featnorm::MomentSrc{na.acc, na.nrows, na.r1, na.n1, na.r2, na.n2}, b, (long)C * cs, na.mode);
    const float sd = sqrtf(m.var + 1e-16f);
    mu = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m.mu)));
    rs = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.0f / sd)));
    if (na.stats && btx == 0 && bty == 0 && threadIdx.x == 0) {
      float* st = na.stats + 4 * b;
      st[0] = m.m1, st[1] = m.m2, st[2] = m.mu, st[3] = sd;
    }
    if (na.x1n) x1n_p = na.x1n + (long)b * na.x1n_bs + (long)gy * W + gx;
  }

  for (int ch = 0; ch < nchunk; ++ch) {
    // chunk ch must have landed; the NBUF-2 younger chunks may stay in flight
    if (ch + NBUF - 2 < nchunk)
      wait_dma_and_barrier<(NBUF - 2) * NK>();
    else
      wait_dma_and_barrier<0>();
    // every wave is past the FMAs of chunk ch-1, whose buffer the next DMA overwrites
    if (ch + NBUF - 1 < nchunk) issue(ch + NBUF - 1);
    const float* cur = lds + (ch % NBUF) * BUF;
    const float* s2 = cur + (y + 3 * wave) * SP + 4 * xg;
    const float* s1 = cur + SRC_FLOATS + y * SP + 4 * xg;
    // software-pipelined over the 12 (channel, row-shift) steps: the window of the next step is
    // requested before the 108 FMAs of the current one so the LDS latency hides behind them.  The
    // prefetch past the last channel reads the next region of the LDS array (in bounds, unused).
    f32x4 wc[3], wn[3];
    float a[PX];
    issue_window(s2, wc);
    land_window(wc);
    load_vec4(s1, a);
#pragma unroll 1
    for (int c = 0; c < CC; ++c) {
      const float* sc = s2 + c * SR * SP;
      f32x4 ta;
      if constexpr (NORM) {
#pragma unroll
        for (int p = 0; p < PX; ++p) {
          a[p] = (a[p] - mu) * rs;
          s1n[p] += a[p];
        }
        // channel c of a chunk is written by wave c % 3 (wave id and c are scalar: a uniform branch)
        if (x1n_p && lane_in && (c == wave || c == wave + NW))
          *reinterpret_cast<float4*>(x1n_p + (long)((grp + ch * G) * CC + c) * cs) = make_float4(a[0], a[1], a[2], a[3]);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        issue_window(k < 2 ? sc + (k + 1) * SP : sc + SR * SP, wn);
        if (k == 2) ta = *reinterpret_cast<const f32x4*>(s1 + (c + 1) * TH * SP);
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
          for (int p = 0; p < PX; ++p) acc[k][j][p] = fmaf(a[p], CORR_WIN(wc, j + p), acc[k][j][p]);
        land_window(wn);
        wc[0] = wn[0], wc[1] = wn[1], wc[2] = wn[2];
      }
      asm volatile("" : "+v"(ta));
      a[0] = ta.x, a[1] = ta.y, a[2] = ta.z, a[3] = ta.w;
    }
  }

  if (G > 1) {  // groups 1 .. G-1 hand their partial sums to group 0 through the rings
    // One row shift at a time, ALL foreign groups at once, each into its own region (3 waves x 9 columns x 64 lanes
    // x 16 B = 27.6 KB per group: 3 groups fit the 4 rings): 2 barriers per row shift = 6 in all.  (Round 1 handed
    // over one group at a time: 18 barriers of a 768-thread workgroup, ~6 us of the ~13 us a coarse level takes.)
    static_assert((G - 1) * NW * N * 64 * PX <= G * RING, "the hand-over regions live in the rings");
    f32x4* xch = reinterpret_cast<f32x4*>(lds_all) + (wave * N) * 64 + lane;
    constexpr int REGION = NW * N * 64;  // float4 per foreign group
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      __syncthreads();  // all FMAs done with the rings / group 0 has consumed the previous row shift
      if (grp != 0) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
          f32x4 t;
          t.x = acc[k][j][0], t.y = acc[k][j][1], t.z = acc[k][j][2], t.w = acc[k][j][3];
          xch[(grp - 1) * REGION + j * 64] = t;
        }
      }
      __syncthreads();
      if (grp == 0) {
#pragma unroll 1
        for (int g = 1; g < G; ++g) {
#pragma unroll
          for (int j = 0; j < N; ++j) {
            const f32x4 t = xch[(g - 1) * REGION + j * 64];
            acc[k][j][0] += t.x, acc[k][j][1] += t.y, acc[k][j][2] += t.z, acc[k][j][3] += t.w;
          }
        }
      }
    }
    if constexpr (NORM) {  // the groups' partial channel sums of the normalised first map meet the same way
      __syncthreads();
      f32x4* xs = reinterpret_cast<f32x4*>(lds_all) + wave * 64 + lane;
      if (grp != 0) {
        f32x4 t;
        t.x = s1n[0], t.y = s1n[1], t.z = s1n[2], t.w = s1n[3];
        xs[(grp - 1) * NW * 64] = t;
      }
      __syncthreads();
      if (grp == 0) {
#pragma unroll 1
        for (int g = 1; g < G; ++g) {
          const f32x4 t = xs[(g - 1) * NW * 64];
          s1n[0] += t.x, s1n[1] += t.y, s1n[2] += t.z, s1n[3] += t.w;
        }
      }
    }
    if (grp != 0) return;
  }
  if (!lane_in) return;
  float* ob = out + (long)b * obs + ((long)(3 * wave * N) * H + gy) * W + gx;
  unsigned sg[PX] = {0u, 0u, 0u, 0u};  // bit k*9+j: this wave's channel (3*wave+k)*9+j is positive
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float v[PX];
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        if constexpr (NORM) {
          const int yy = gy + 3 * wave + k - D, xx = gx + p + j - D;
          const bool inb = yy >= 0 && yy < H && xx >= 0 && xx < W;
          v[p] = (acc[k][j][p] - (inb ? mu * s1n[p] : 0.f)) * (inv_c * rs);
        } else {
          v[p] = acc[k][j][p] * inv_c;
        }
        sg[p] |= v[p] > 0.f ? 1u << (k * N + j) : 0u;
        v[p] = v[p] > 0.f ? v[p] : v[p] * slope;  // fused LeakyReLU (slope 1 = identity)
      }
      *reinterpret_cast<float4*>(ob + (k * N + j) * cs) = make_float4(v[0], v[1], v[2], v[3]);
    }
  if (sign_bits)
    *reinterpret_cast<uint4*>(sign_bits + (((long)b * NW + wave) * H + gy) * W + gx) = make_uint4(sg[0], sg[1], sg[2], sg[3]);
}

// ------------------------------------------------------------------------------------------------
// mode 0: gx1[c,p] = (1/C) sum_{i,j} g[i*9+j][p]        * x2[c][p+(i-4,j-4)]
// mode 1: gx2[c,q] = (1/C) sum_{i,j} g[80-(i*9+j)][q+(i-4,j-4)] * x1[c][q+(i-4,j-4)]
// ACT: how the fused LeakyReLU derivative is selected -- 0 none (slope 1), 1 sign of the forward output
// `fout`, 2 the forward's compact sign words.
// NORM (level kernels): the backward of fwd_kernel<.., NORM = true>.  `x1` holds the NORMALISED first map x1n (batch
// stride x1bs: the decoder's concatenated input keeps it), `x2` the RAW (warped) second map; (mu, sigma) come from
// `stats`.  With x2n = (x2 - mu) / sigma inside the image and 0 outside,
//   mode 0: d/d x1n[c,p] = (1 / (C sigma)) (sum_d g[d,p] x2[c,p+d]  -  mu * sum_{d: p+d inside} g[d,p])
//   mode 1: d/d x2n[c,q] = (1 / C) sum_d g[d,q-d] x1n[c,q-d]
// -- the gradients w.r.t. the NORMALISED maps; the normalisation's own backward follows in the warp launch.
template <int NBUF, int ACT, bool NORM = false>
__global__ __launch_bounds__(NT, 3) void bwd_kernel(const float* __restrict__ gout, const float* __restrict__ fout,
                                                    const unsigned* __restrict__ sign_bits, float slope,
                                                    const float* __restrict__ x1,
                                                    const float* __restrict__ x2, float* __restrict__ gx1,
                                                    float* __restrict__ gx2, int B, int C, int H, int W,
                                                    float inv_c, int mode_base, int nmodes,
                                                    long gbs /* batch stride of gout */, long fbs /* of fout */,
                                                    long x1bs /* batch stride of x1 */, const float* __restrict__ stats,
                                                    float* __restrict__ zero_c, float* __restrict__ zero_f,
                                                    float* __restrict__ zero_fc, int* __restrict__ zero_i) {
  // zero_c [B,C,H,W], zero_f [B,2,H,W], zero_fc [B,2,H/2,W/2] (level backward, each nullable): accumulation targets of the
  // warp backward that follows, zero-filled HERE -- mode-1 workgroups clear their tile of zero_c next to the gradient
  // they store, split-0 mode-0 workgroups their tile of the two flow-gradient buffers -- instead of by fill launches
  // (two of the seven launches a coarse level's backward consisted of)
  constexpr int BUF = SRC_FLOATS;
  // Cross-wave sum of the per-wave partials.  Channel c of a chunk is OWNED by wave c % 3: the other two waves
  // publish their 4-pixel partials for it in LDS, the owner keeps its own in registers and, one iteration later
  // (after the barrier that also waits for the next chunk's DMA), adds the three and stores.  `part` is double-
  // buffered so that this ONE barrier per chunk is enough -- the previous form (all three waves publish, barrier,
  // all threads reduce, and the next chunk's barrier protecting the reuse) had two.
  constexpr int PART1 = 2 * CC * 64 * PX;  // 2048 floats: two foreign partials per channel
  __shared__ __attribute__((aligned(16))) float lds[NBUF * BUF + 2 * PART1];
  float* part = lds + NBUF * BUF;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // wave id in an SGPR
  int xg, y;
  lane_xy(lane, xg, y);
  // both gradients of a tile sit next to each other in the XCD's tile run (mode is the fastest index):
  // the second pass over gout then hits the L2 the first one filled (fabric reads 217 -> ~150 MB at B16 96x160)
  int btx, bty, b, mode;
  {
    const int ntx = (W + TW - 1) / TW, nty = (H + TH - 1) / TH;
    const int T = ntx * nty * B * nmodes;
    const int per = (T + 7) >> 3;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || t >= T) return;
    mode = mode_base + t % nmodes;
    const int tt = t / nmodes;
    btx = tt % ntx;
    bty = (tt / ntx) % nty;
    b = tt / (ntx * nty);
  }
  const int tx0 = btx * TW, ty0 = bty * TH;
  const long cs = (long)H * W;
  const float* srcb = mode == 0 ? x2 + (long)b * C * cs : x1 + (long)b * x1bs;
  float* dstb = (mode == 0 ? gx1 : gx2) + (long)b * C * cs;
  const float* gb = gout + (long)b * gbs;
  const float* fb = ACT == 1 ? fout + (long)b * fbs : nullptr;  // forward output: LeakyReLU derivative
  const int gy = ty0 + y, gx = tx0 + 4 * xg;
  // Every load of the gradient phase is UNCONDITIONAL: a piece outside the image reads the 16 zero bytes
  // of g_zero16 instead of being branched around.  With branches hipcc waits for each load before the
  // next one is issued (s_waitcnt vmcnt(0) per displacement: 27+ serialised round trips per workgroup,
  // two thirds of the kernel's wave-time measured); without them a whole row shift is in flight at once.
  auto load4 = [&](bool ok, long off) {  // 4 output gradients through the fused LeakyReLU
    float4 t = *reinterpret_cast<const float4*>(ok ? gb + off : g_zero16);
    if (ACT == 1) {
      const float4 f = *reinterpret_cast<const float4*>(ok ? fb + off : g_zero16);
      t.x = f.x > 0.f ? t.x : t.x * slope, t.y = f.y > 0.f ? t.y : t.y * slope;
      t.z = f.z > 0.f ? t.z : t.z * slope, t.w = f.w > 0.f ? t.w : t.w * slope;
    }
    return t;
  };
  // compact alternative to `fout`: the forward's sign words (plane w holds the 27 channels of wave w)
  const unsigned* sb = ACT == 2 ? sign_bits + (long)b * NW * cs : nullptr;

  // 10 DMA instructions per chunk over 3 waves: every wave issues 4 (the surplus two re-send piece 9,
  // identical bytes to the same LDS block) so that the per-wave count is a compile-time constant
  constexpr int NK = (SRC_DMA + NW - 1) / NW;  // 4
  int off[NK], blk[NK];
#pragma unroll
  for (int m = 0; m < NK; ++m) {
    int k = wave + NW * m;
    if (k >= SRC_DMA) k = SRC_DMA - 1;
    blk[m] = k;
    off[m] = piece_offset(k, lane, SR, SP / 4, ty0 - D, tx0 - D, H, W);
  }
  // Channel chunks are independent in the backward (every output channel is its own sum), so when a level
  // has few tiles the chunks are spread over gridDim.y workgroups: chunk k of this workgroup is channel
  // chunk blockIdx.y + k * gridDim.y.  gout is then re-read per workgroup (from L2; small at such levels).
  const int nsplit = gridDim.y, split = blockIdx.y;
  auto issue = [&](int k) {
    float* buf = lds + (k % NBUF) * BUF;
    const float* g = srcb + (long)(split + k * nsplit) * CC * cs;
#pragma unroll
    for (int m = 0; m < NK; ++m) dma16(off[m] >= 0 ? g + off[m] : g_zero16, buf + blk[m] * 256);
  };
  const int nchunk = (C / CC - split + nsplit - 1) / nsplit;  // chunks handled by this workgroup
#pragma unroll
  for (int pre = 0; pre < NBUF - 1; ++pre)
    if (pre < nchunk) issue(pre);

  // the 27 x 4 output gradients this lane combines, read once.  The two modes are separate straight-line
  // regions under ONE workgroup-uniform branch, so that the loads of a row shift are scheduled together.
  float g[3][N][PX];
  const unsigned* zu = reinterpret_cast<const unsigned*>(g_zero16);
  if (mode == 0) {
    const bool ok = gy < H && gx < W;
    unsigned pk[PX] = {0u, 0u, 0u, 0u};  // ACT == 2: plane-`wave` sign words of the lane's own 4 pixels
    if (ACT == 2) {
      const uint4 t = *reinterpret_cast<const uint4*>(ok ? sb + (long)wave * cs + (long)gy * W + gx : zu);
      pk[0] = t.x, pk[1] = t.y, pk[2] = t.z, pk[3] = t.w;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = 3 * wave + k;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float4 t = load4(ok, (i * N + j) * cs + (long)gy * W + gx);
        g[k][j][0] = t.x, g[k][j][1] = t.y, g[k][j][2] = t.z, g[k][j][3] = t.w;
#pragma unroll
        for (int p = 0; p < PX; ++p)
          if (ACT == 2 && !((pk[p] >> (k * N + j)) & 1u)) g[k][j][p] *= slope;
        if (ACT == 1 && j % 3 == 2) asm volatile("" ::: "memory");  // (two loads per displacement: bound the batch)
      }
    }
  } else {
    // tiles whose shifted columns gx-4 .. gx+7 all lie inside the row (every tile but the first and last of a tile row)
    // read each displacement's 4 gradients as ONE 4-byte-aligned dwordx4 (27 loads instead of 45 aligned ones + register
    // shifts); the tiles at the left / right image border keep the aligned form, whose blocks are entirely in or out
    const bool xin = tx0 >= D && tx0 + TW + D <= W;  // workgroup-uniform
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = 3 * wave + k;
      const int yy = gy + i - D;
      const bool rowok = yy >= 0 && yy < H;
      // ACT == 2: channel 80-(i*9+j) is row 2-k, column 8-j of plane 2-wave; it is needed at pixel
      // (yy, gx+p+j-4): the 9-bit row-(2-k) fields of the 12 sign words of columns gx-4 .. gx+7, three
      // fields per register
      unsigned pk[4] = {0u, 0u, 0u, 0u};
      if (ACT == 2) {
        const unsigned* row = sb + (long)(NW - 1 - wave) * cs + (long)yy * W;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int xb = gx - D + 4 * q;
          const uint4 t = *reinterpret_cast<const uint4*>(rowok && xb >= 0 && xb < W ? row + xb : zu);
          const unsigned wd[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int w = 4 * q + e;
            pk[w / 3] |= ((wd[e] >> ((2 - k) * N)) & 0x1ffu) << ((w % 3) * N);
          }
        }
      }
      if (xin) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const long gro = (N * N - 1 - (i * N + j)) * cs + (long)yy * W + gx + (j - D);
          f32x4u t = *reinterpret_cast<const f32x4u*>(rowok ? gb + gro : g_zero16);
          if (ACT == 1) {
            const f32x4u f = *reinterpret_cast<const f32x4u*>(rowok ? fb + gro : g_zero16);
            t.x = f.x > 0.f ? t.x : t.x * slope, t.y = f.y > 0.f ? t.y : t.y * slope;
            t.z = f.z > 0.f ? t.z : t.z * slope, t.w = f.w > 0.f ? t.w : t.w * slope;
          }
          g[k][j][0] = t.x, g[k][j][1] = t.y, g[k][j][2] = t.z, g[k][j][3] = t.w;
#pragma unroll
          for (int p = 0; p < PX; ++p)
            if (ACT == 2 && !((pk[(j + p) / 3] >> (((j + p) % 3) * N + (N - 1 - j))) & 1u)) g[k][j][p] *= slope;
          if (ACT == 1 ? j % 3 == 2 : j == N - 1) asm volatile("" ::: "memory");  // a whole row shift (9 loads) in flight
        }
        continue;
      }
#pragma unroll
      for (int j = 0; j < N; ++j) {
        // 4 consecutive pixels starting at gx + (j-4): one or two ALIGNED float4 loads (gx and W are
        // multiples of 4, so an aligned block is entirely inside or outside a row) + a compile-time
        // register shift, instead of 4 bounds-checked scalar loads
        const long gro = (N * N - 1 - (i * N + j)) * cs + (long)yy * W;
        const int e = j - D;                                    // -4 .. 4 (compile-time after unrolling)
        const int blo = (e >= 0 ? e / 4 : -((3 - e) / 4)) * 4;  // 4*floor(e/4)
        const int sh = e - blo;                                 // 0..3
        const int xlo = gx + blo, xhi = xlo + 4;
        const float4 tl = load4(rowok && xlo >= 0 && xlo < W, gro + xlo);
        const float lo[4] = {tl.x, tl.y, tl.z, tl.w};
        float hi[4] = {0.f, 0.f, 0.f, 0.f};
        if (sh != 0) {
          const float4 th = load4(rowok && xhi >= 0 && xhi < W, gro + xhi);
          hi[0] = th.x, hi[1] = th.y, hi[2] = th.z, hi[3] = th.w;
        }
#pragma unroll
        for (int p = 0; p < PX; ++p) g[k][j][p] = (p + sh < 4) ? lo[p + sh] : hi[p + sh - 4];
#pragma unroll
        for (int p = 0; p < PX; ++p)
          if (ACT == 2 && !((pk[(j + p) / 3] >> (((j + p) % 3) * N + (N - 1 - j))) & 1u)) g[k][j][p] *= slope;
        // at most half a row shift's loads (8 x 16 B per lane) in flight at a time: more does not fit
        // the 168-VGPR budget (3 waves per SIMD) next to g[][][]
        if (ACT == 1 ? j % 3 == 2 : (j == 4 || j == N - 1)) asm volatile("" ::: "memory");
      }
    }
  }
  // the g loads above are younger than the prologue DMA: drain everything once, then count exactly
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float nmg[PX] = {0.f, 0.f, 0.f, 0.f};  // NORM, mode 0: -mu * (this wave's share of the in-image gradient sum)
  if constexpr (NORM) {
    const float mu = stats[4 * b + 2], rs = 1.0f / stats[4 * b + 3];
    if (mode == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int yy = gy + 3 * wave + k - D;
        const bool rowin = yy >= 0 && yy < H;
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
          for (int p = 0; p < PX; ++p) {
            const int xx = gx + p + j - D;
            nmg[p] += (rowin && xx >= 0 && xx < W) ? g[k][j][p] : 0.f;
          }
      }
#pragma unroll
      for (int p = 0; p < PX; ++p) nmg[p] *= -mu;
      inv_c *= rs;
    }
  }

  float own[2][PX] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // this wave's partials of the channels it owns
  const bool lane_in = gy < H && gx < W;
  if constexpr (NORM) {
    if (zero_i && mode == 0 && split == 0 && btx == 0 && bty == 0 && threadIdx.x == 0) zero_i[b] = 0;  // per-sample flag word
    if (mode == 0 && split == 0 && wave == 0 && lane_in) {
      if (zero_f) {
        float* z = zero_f + (long)b * 2 * cs + (long)gy * W + gx;
        *reinterpret_cast<float4*>(z) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(z + cs) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (zero_fc && (gy & 1) == 0) {  // the 2 coarse cells under the lane's 4 pixels, both planes
        const int Hc = H / 2, Wc = W / 2;
        float* z = zero_fc + (long)b * 2 * Hc * Wc + (long)(gy >> 1) * Wc + (gx >> 1);
        *reinterpret_cast<float2*>(z) = make_float2(0.f, 0.f);
        *reinterpret_cast<float2*>(z + (long)Hc * Wc) = make_float2(0.f, 0.f);
      }
    }
  }
  // add the three waves' partials of chunk `chunk` (in wave order, as before: bit-identical results) and store
  auto reduce_store = [&](int chunk, int par) {
    const float* pp = part + par * PART1;
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      if (c % NW != wave) continue;  // wave-uniform (wave id lives in an SGPR)
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(pp + ((c * 2 + 0) * 64 + lane) * PX);  // wave (owner+1) % 3
      const f32x4 s1 = *reinterpret_cast<const f32x4*>(pp + ((c * 2 + 1) * 64 + lane) * PX);  // wave (owner+2) % 3
      const float* o = own[c / NW];
      float r[PX];
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        const float w0 = wave == 0 ? o[p] : (wave == 1 ? s1[p] : s0[p]);
        const float w1 = wave == 1 ? o[p] : (wave == 2 ? s1[p] : s0[p]);
        const float w2 = wave == 2 ? o[p] : (wave == 0 ? s1[p] : s0[p]);
        r[p] = ((w0 + w1) + w2) * inv_c;
      }
      if (lane_in) {
        const long o = (long)((split + chunk * nsplit) * CC + c) * cs + (long)gy * W + gx;
        *reinterpret_cast<float4*>(dstb + o) = make_float4(r[0], r[1], r[2], r[3]);
        if constexpr (NORM) {
          if (mode == 1 && zero_c) *reinterpret_cast<float4*>(zero_c + (long)b * C * cs + o) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
  };
  int par = 0;
  for (int ch = 0; ch < nchunk; ++ch) {
    // chunk ch landed and the partials of chunk ch-1 are visible (the waits cover this wave's LDS writes too).
    // The stores of the previous reduce may still drain: they are older than the DMA of chunk ch+NBUF-2 only
    // when NBUF == 2, so count conservatively: wait for everything but the youngest (NBUF-2)*NK DMA instructions
    if (ch + NBUF - 2 < nchunk && ch > 0)
      wait_dma_and_barrier<(NBUF - 2) * NK>();
    else
      wait_dma_and_barrier<0>();
    if (ch + NBUF - 1 < nchunk) issue(ch + NBUF - 1);
    if (ch > 0) reduce_store(ch - 1, par ^ 1);
    const float* cur = lds + (ch % NBUF) * BUF;
    const float* s2 = cur + (y + 3 * wave) * SP + 4 * xg;
    float* pw = part + par * PART1;
    f32x4 wc[3], wn[3];
    issue_window(s2, wc);
    land_window(wc);
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      const float* sc = s2 + c * SR * SP;
      float pa[PX] = {nmg[0], nmg[1], nmg[2], nmg[3]};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        issue_window(k < 2 ? sc + (k + 1) * SP : sc + SR * SP, wn);  // next step (in-bounds past the end)
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
          for (int p = 0; p < PX; ++p) pa[p] = fmaf(g[k][j][p], CORR_WIN(wc, j + p), pa[p]);
        land_window(wn);
        wc[0] = wn[0], wc[1] = wn[1], wc[2] = wn[2];
      }
      if (c % NW == wave) {  // wave-uniform: the owner keeps its partial
#pragma unroll
        for (int p = 0; p < PX; ++p) own[c / NW][p] = pa[p];
      } else {
        const int slot = (wave - c % NW - 1 + NW) % NW;  // 0: wave (owner+1) % 3, 1: wave (owner+2) % 3
        *reinterpret_cast<float4*>(pw + ((c * 2 + slot) * 64 + lane) * PX) = make_float4(pa[0], pa[1], pa[2], pa[3]);
      }
    }
    par ^= 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (nchunk > 0) reduce_store(nchunk - 1, par ^ 1);
}

inline bool eligible(int C, int W, int max_disp) { return max_disp == 4 && (W % 4) == 0 && (C % CC) == 0; }

inline int launch_fwd(const float* x1, const float* x2, float* out, unsigned* sign_bits, int B, int C, int H, int W,
                      float slope, hipStream_t st, long obs = 0, const NormArgs* norm = nullptr) {
  if (obs == 0) obs = (long)N * N * H * W;
  const NormArgs na = norm ? *norm : NormArgs{nullptr, 0, 0, nullptr, 0, nullptr, nullptr, 0, nullptr, 0};
  const int tiles = af_cdiv(W, TW) * af_cdiv(H, TH) * B;
  dim3 grid(grid_for_tiles(tiles));
  // many tiles: 4 workgroups per CU hide each other's DMA latency, keep LDS small (2 buffers);
  // few tiles: one workgroup per CU -> deeper ring so its own DMA runs 3 chunks ahead
  const float inv_c = 1.0f / (float)C;
#define CORR_V2_FWD(NB, GG, NN, THREADS) \
  hipLaunchKernelGGL((fwd_kernel<NB, GG, NN>), grid, dim3(THREADS), 0, st, x1, x2, out, sign_bits, B, C, H, W, inv_c, slope, obs, na)
  if (tiles <= 160 && (C / CC) % 4 == 0 && C / CC >= 8) {  // few tiles, >= 2 chunks per group: 4 channel groups per workgroup
    if (norm) CORR_V2_FWD(2, 4, true, NT * 4); else CORR_V2_FWD(2, 4, false, NT * 4);
  // (2 groups at ~1 tile per CU, e.g. 48x80: 23.4 -> 26.1 us -- splitting only pays while CUs are idle)
  } else if (tiles >= 768) {
    if (norm) CORR_V2_FWD(2, 1, true, NT); else CORR_V2_FWD(2, 1, false, NT);
  } else {
    if (norm) CORR_V2_FWD(4, 1, true, NT); else CORR_V2_FWD(4, 1, false, NT);
  }
#undef CORR_V2_FWD
  return af_launch_status();
}

inline int launch_bwd(const float* gout, const float* fout, const unsigned* sign_bits, float slope, const float* x1,
                      const float* x2,
                      float* gx1, float* gx2, int B, int C, int H, int W, hipStream_t st, long gbs = 0, long fbs = 0,
                      long x1bs = 0, const float* stats = nullptr, float* zero_c = nullptr, float* zero_f = nullptr,
                      float* zero_fc = nullptr, int* zero_i = nullptr) {
  if (gbs == 0) gbs = (long)N * N * H * W;
  if (fbs == 0) fbs = (long)N * N * H * W;
  if (x1bs == 0) x1bs = (long)C * H * W;
  const int nmodes = (gx1 ? 1 : 0) + (gx2 ? 1 : 0);
  if (nmodes == 0) return ARFLOW_OK;
  const int tiles = af_cdiv(W, TW) * af_cdiv(H, TH) * B * nmodes;
  int nsplit = 1;  // spread channel chunks over workgroups until ~1024 are in flight
  while (nsplit * 2 <= C / CC && tiles * nsplit * 2 <= 1024) nsplit *= 2;
  dim3 grid(grid_for_tiles(tiles), nsplit);
  const int act = sign_bits ? 2 : (fout ? 1 : 0);
  const float inv_c = 1.0f / (float)C;
  const int mb = gx1 ? 0 : 1;
#define CORR_V2_BWD(NB, ACT, NN)                                                                                      \
  hipLaunchKernelGGL((bwd_kernel<NB, ACT, NN>), grid, dim3(NT), 0, st, gout, fout, sign_bits, slope, x1, x2, gx1, gx2, \
                     B, C, H, W, inv_c, mb, nmodes, gbs, fbs, x1bs, stats, zero_c, zero_f, zero_fc, zero_i)
  if (stats) {  // level kernels: always with the sign words of the fused LeakyReLU (ACT 2) or without activation (0)
    if (act == 1) return ARFLOW_EPARAM;
    if (tiles >= 768) {
      if (act == 2) CORR_V2_BWD(2, 2, true); else CORR_V2_BWD(2, 0, true);
    } else {
      if (act == 2) CORR_V2_BWD(3, 2, true); else CORR_V2_BWD(3, 0, true);
    }
  } else if (tiles >= 768) {
    if (act == 2) CORR_V2_BWD(2, 2, false); else if (act == 1) CORR_V2_BWD(2, 1, false); else CORR_V2_BWD(2, 0, false);
  } else {
    if (act == 2) CORR_V2_BWD(3, 2, false); else if (act == 1) CORR_V2_BWD(3, 1, false); else CORR_V2_BWD(3, 0, false);  // ring of 3: 46 KB of LDS, 3 workgroups per CU (4: 56 KB, 2)
  }
#undef CORR_V2_BWD
  return af_launch_status();
}

}  // namespace corr_v2
