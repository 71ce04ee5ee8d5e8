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

namespace corr_v2 {

constexpr int D = 4, N = 9, PX = 4, TW = 32, TH = 8, CC = 4, NW = 3, NT = 64 * NW;
constexpr int SR = TH + 2 * D;           // 16 staged rows
constexpr int SP = TW + 2 * D;           // 40 floats: row pitch of both staged tiles
constexpr int SRC_FLOATS = CC * SR * SP;  // 2560 = 10 wave-wide DMA instructions
constexpr int X1_FLOATS = CC * TH * SP;   // 1280 = 5  (32 of 40 floats used: pitch kept for banking)
constexpr int SRC_DMA = SRC_FLOATS / 256, X1_DMA = X1_FLOATS / 256;

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// The empty asm pins each read as ONE ds_read_b128.  Left alone, hipcc re-splits the window into
// ds_read2_b32 pairs (to feed v_pk_fma_f32 operands at odd offsets), which bank-conflict (32-bank
// mode) and double the LDS instruction count: measured 57 % of LDS cycles lost in the backward kernel.
__device__ __forceinline__ void load_window(const float* row, float (&w)[PX + 2 * D]) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    f32x4 t = *reinterpret_cast<const f32x4*>(row + 4 * q);
    asm volatile("" : "+v"(t));
    w[4 * q] = t.x, w[4 * q + 1] = t.y, w[4 * q + 2] = t.z, w[4 * q + 3] = t.w;
  }
}
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
template <int NBUF>
__global__ __launch_bounds__(NT, 3) void fwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                    float* __restrict__ out, int nimg, int C, int H, int W,
                                                    float inv_c, float slope) {
  constexpr int BUF = SRC_FLOATS + X1_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[NBUF * BUF + TH * SP];  // + pad: prefetch runs a channel ahead
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
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
      const float* g = (k < SRC_DMA ? x2b : x1b) + (long)chunk * CC * cs;
      dma16(off[m] >= 0 ? g + off[m] : g_zero16, buf + k * 256);  // x1 region follows x2: block k carries over
    }
  };

  const int nchunk = C / CC;
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
    float w[2][PX + 2 * D];
    float a[PX], an[PX];
    load_window(s2, w[0]);
    load_vec4(s1, a);
#pragma unroll 1
    for (int c = 0; c < CC; ++c) {
      const float* sc = s2 + c * SR * SP;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (k < 2) {
          load_window(sc + (k + 1) * SP, w[(k + 1) & 1]);
        } else {
          load_window(sc + SR * SP, w[(k + 1) & 1]);
          load_vec4(s1 + (c + 1) * TH * SP, an);
        }
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
          for (int p = 0; p < PX; ++p) acc[k][j][p] = fmaf(a[p], w[k & 1][j + p], acc[k][j][p]);
      }
      // after 3 steps the "next" window sits in w[1]; it becomes w[0] of the next channel
#pragma unroll
      for (int q = 0; q < PX + 2 * D; ++q) w[0][q] = w[1][q];
#pragma unroll
      for (int p = 0; p < PX; ++p) a[p] = an[p];
    }
  }

  const int gy = ty0 + y, gx = tx0 + 4 * xg;
  if (gy >= H || gx >= W) return;
  float* ob = out + (((long)b * N * N + 3 * wave * N) * H + gy) * W + gx;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float v[PX];
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        v[p] = acc[k][j][p] * inv_c;
        v[p] = v[p] > 0.f ? v[p] : v[p] * slope;  // fused LeakyReLU (slope 1 = identity)
      }
      *reinterpret_cast<float4*>(ob + (k * N + j) * cs) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ------------------------------------------------------------------------------------------------
// mode 0: gx1[c,p] = (1/C) sum_{i,j} g[i*9+j][p]        * x2[c][p+(i-4,j-4)]
// mode 1: gx2[c,q] = (1/C) sum_{i,j} g[80-(i*9+j)][q+(i-4,j-4)] * x1[c][q+(i-4,j-4)]
template <int NBUF>
__global__ __launch_bounds__(NT, 3) void bwd_kernel(const float* __restrict__ gout, const float* __restrict__ fout,
                                                    float slope, const float* __restrict__ x1,
                                                    const float* __restrict__ x2, float* __restrict__ gx1,
                                                    float* __restrict__ gx2, int B, int C, int H, int W,
                                                    float inv_c, int mode_base, int nmodes) {
  constexpr int BUF = SRC_FLOATS;
  constexpr int PART = NW * CC * 64 * PX;  // 3072 floats
  __shared__ __attribute__((aligned(16))) float lds[NBUF * BUF + PART];
  float* part = lds + NBUF * BUF;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
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
  const float* srcb = (mode == 0 ? x2 : x1) + (long)b * C * cs;
  float* dstb = (mode == 0 ? gx1 : gx2) + (long)b * C * cs;
  const float* gb = gout + (long)b * N * N * cs;
  const float* fb = fout ? fout + (long)b * N * N * cs : nullptr;  // forward output: LeakyReLU derivative
  const int gy = ty0 + y, gx = tx0 + 4 * xg;
  auto load4 = [&](long off) {  // 4 output gradients through the fused LeakyReLU
    float4 t = *reinterpret_cast<const float4*>(gb + off);
    if (fb) {
      const float4 f = *reinterpret_cast<const float4*>(fb + off);
      t.x = f.x > 0.f ? t.x : t.x * slope, t.y = f.y > 0.f ? t.y : t.y * slope;
      t.z = f.z > 0.f ? t.z : t.z * slope, t.w = f.w > 0.f ? t.w : t.w * slope;
    }
    return t;
  };

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

  // the 27 x 4 output gradients this lane combines, read once
  float g[3][N][PX];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int i = 3 * wave + k;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (mode == 0) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy < H && gx < W) t = load4((i * N + j) * cs + (long)gy * W + gx);
        g[k][j][0] = t.x, g[k][j][1] = t.y, g[k][j][2] = t.z, g[k][j][3] = t.w;
      } else {
        // 4 consecutive pixels starting at gx + (j-4): one or two ALIGNED float4 loads (gx and W are
        // multiples of 4, so an aligned block is entirely inside or outside a row) + a compile-time
        // register shift, instead of 4 bounds-checked scalar loads
        const int yy = gy + i - D;
        const long gro = (N * N - 1 - (i * N + j)) * cs + (long)yy * W;
        const bool rowok = yy >= 0 && yy < H;
        const int e = j - D;                       // -4 .. 4 (compile-time after unrolling)
        const int blo = (e >= 0 ? e / 4 : -((3 - e) / 4)) * 4;  // 4*floor(e/4)
        const int sh = e - blo;                    // 0..3
        float lo[4] = {0.f, 0.f, 0.f, 0.f}, hi[4] = {0.f, 0.f, 0.f, 0.f};
        const int xlo = gx + blo, xhi = xlo + 4;
        if (rowok && xlo >= 0 && xlo < W) {
          const float4 t = load4(gro + xlo);
          lo[0] = t.x, lo[1] = t.y, lo[2] = t.z, lo[3] = t.w;
        }
        if (sh != 0 && rowok && xhi >= 0 && xhi < W) {
          const float4 t = load4(gro + xhi);
          hi[0] = t.x, hi[1] = t.y, hi[2] = t.z, hi[3] = t.w;
        }
#pragma unroll
        for (int p = 0; p < PX; ++p) g[k][j][p] = (p + sh < 4) ? lo[p + sh] : hi[p + sh - 4];
      }
    }
  }
  // the g loads above are younger than the prologue DMA: drain everything once, then count exactly
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  for (int ch = 0; ch < nchunk; ++ch) {
    // chunk ch landed (stores of the previous reduce may still drain: they are older than the DMA of
    // chunk ch+NBUF-2 only when NBUF == 2, so count conservatively: wait for everything but the
    // youngest (NBUF-2)*NK DMA instructions)
    if (ch + NBUF - 2 < nchunk && ch > 0)
      wait_dma_and_barrier<(NBUF - 2) * NK>();
    else
      wait_dma_and_barrier<0>();
    if (ch + NBUF - 1 < nchunk) issue(ch + NBUF - 1);
    const float* cur = lds + (ch % NBUF) * BUF;
    const float* s2 = cur + (y + 3 * wave) * SP + 4 * xg;
    float w[2][PX + 2 * D];
    load_window(s2, w[0]);
#pragma unroll 1
    for (int c = 0; c < CC; ++c) {
      const float* sc = s2 + c * SR * SP;
      float pa[PX] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        load_window(k < 2 ? sc + (k + 1) * SP : sc + SR * SP, w[(k + 1) & 1]);  // next step (in-bounds past the end)
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
          for (int p = 0; p < PX; ++p) pa[p] = fmaf(g[k][j][p], w[k & 1][j + p], pa[p]);
      }
#pragma unroll
      for (int q = 0; q < PX + 2 * D; ++q) w[0][q] = w[1][q];
      *reinterpret_cast<float4*>(part + ((wave * CC + c) * 64 + lane) * PX) = make_float4(pa[0], pa[1], pa[2], pa[3]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // sum the three waves' partials: CC*64 float4 results over 192 threads
    for (int o = threadIdx.x; o < CC * 64; o += NT) {
      const int c = o >> 6, l = o & 63;
      int oxg, oy;
      lane_xy(l, oxg, oy);
      const float4 p0 = *reinterpret_cast<const float4*>(part + ((0 * CC + c) * 64 + l) * PX);
      const float4 p1 = *reinterpret_cast<const float4*>(part + ((1 * CC + c) * 64 + l) * PX);
      const float4 p2 = *reinterpret_cast<const float4*>(part + ((2 * CC + c) * 64 + l) * PX);
      const int oyy = ty0 + oy, oxx = tx0 + 4 * oxg;
      if (oyy < H && oxx < W)
        *reinterpret_cast<float4*>(dstb + (long)((split + ch * nsplit) * CC + c) * cs + (long)oyy * W + oxx) =
            make_float4((p0.x + p1.x + p2.x) * inv_c, (p0.y + p1.y + p2.y) * inv_c, (p0.z + p1.z + p2.z) * inv_c,
                        (p0.w + p1.w + p2.w) * inv_c);
    }
  }
}

inline bool eligible(int C, int W, int max_disp) { return max_disp == 4 && (W % 4) == 0 && (C % CC) == 0; }

inline int launch_fwd(const float* x1, const float* x2, float* out, int B, int C, int H, int W, float slope,
                      hipStream_t st) {
  const int tiles = af_cdiv(W, TW) * af_cdiv(H, TH) * B;
  dim3 grid(grid_for_tiles(tiles));
  // many tiles: 4 workgroups per CU hide each other's DMA latency, keep LDS small (2 buffers);
  // few tiles: one workgroup per CU -> deeper ring so its own DMA runs 3 chunks ahead
  if (tiles >= 768)
    hipLaunchKernelGGL(fwd_kernel<2>, grid, dim3(NT), 0, st, x1, x2, out, B, C, H, W, 1.0f / (float)C, slope);
  else
    hipLaunchKernelGGL(fwd_kernel<4>, grid, dim3(NT), 0, st, x1, x2, out, B, C, H, W, 1.0f / (float)C, slope);
  return af_launch_status();
}

inline int launch_bwd(const float* gout, const float* fout, float slope, const float* x1, const float* x2,
                      float* gx1, float* gx2, int B, int C, int H, int W, hipStream_t st) {
  const int nmodes = (gx1 ? 1 : 0) + (gx2 ? 1 : 0);
  if (nmodes == 0) return ARFLOW_OK;
  const int tiles = af_cdiv(W, TW) * af_cdiv(H, TH) * B * nmodes;
  int nsplit = 1;  // spread channel chunks over workgroups until ~1024 are in flight
  while (nsplit * 2 <= C / CC && tiles * nsplit * 2 <= 1024) nsplit *= 2;
  dim3 grid(grid_for_tiles(tiles), nsplit);
  if (tiles >= 768)
    hipLaunchKernelGGL(bwd_kernel<2>, grid, dim3(NT), 0, st, gout, fout, slope, x1, x2, gx1, gx2, B, C, H, W, 1.0f / (float)C,
                       gx1 ? 0 : 1, nmodes);
  else
    hipLaunchKernelGGL(bwd_kernel<4>, grid, dim3(NT), 0, st, gout, fout, slope, x1, x2, gx1, gx2, B, C, H, W, 1.0f / (float)C,
                       gx1 ? 0 : 1, nmodes);
  return af_launch_status();
}

}  // namespace corr_v2
