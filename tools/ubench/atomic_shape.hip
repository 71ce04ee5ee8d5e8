// Micro-benchmark: cost of global float atomics by the SHAPE of one wave instruction, for the flush of the feature-warp
// d/dsrc windows (warp.hip lds_scatter): a window of 10 rows x 34 cells somewhere in a [C*H][W = 160] float image
// (31.5 MB), 4 channel planes per visit.  Every mode adds the same 340 cells per window and channel; they differ in how
// the cells are dealt to lanes.  Reports time per launch and cells/s.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int W = 160, H = 96, C = 32, B = 16, BH = 10, BW = 34;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* __restrict__ dst, int ntile_x, int ntile_y) {
  // one workgroup per 8 x 32 tile, as in the real kernel; the window starts 1 px left/up of the tile (clamped)
  const int t = blockIdx.x;
  const int tx = t % ntile_x, ty = (t / ntile_x) % ntile_y, b = t / (ntile_x * ntile_y);
  int x0 = tx * 32 - 1 + (t % 3), y0 = ty * 8 - 1;
  x0 = max(0, min(x0, W - BW));
  y0 = max(0, min(y0, H - BH));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long plane = (long)H * W;
  float* base = dst + (long)b * C * plane;
  for (int c0 = 0; c0 < C; c0 += 4) {
    if (MODE == 0) {  // current scheme: compacted cells in row-major order, 256 threads stride the 340 cells, 4 channels each
      for (int i = threadIdx.x; i < BH * BW; i += 256) {
        const int r = i / BW, cx = i - r * BW;
        float* d = base + (long)c0 * plane + (long)(y0 + r) * W + x0 + cx;
#pragma unroll
        for (int c = 0; c < 4; ++c) atomicAdd(d + c * plane, 1.0f);
      }
    } else if (MODE == 1) {  // one row per wave instruction (34 active lanes), wave w takes channel w
      float* d = base + (long)(c0 + wave) * plane + (long)y0 * W + x0 + lane;
      for (int r = 0; r < BH; ++r)
        if (lane < BW) atomicAdd(d + r * W, 1.0f);
    } else if (MODE == 2) {  // 64-byte aligned blocks: 4 blocks of 16 lanes per instruction, lanes outside the window masked
      const int xa = x0 & ~15, nblk = (x0 + BW - xa + 15) / 16;  // 3 (sometimes 4) blocks per row
      const int total = BH * nblk;
      float* d = base + (long)(c0 + wave) * plane;
      for (int q = lane >> 4; q < total; q += 4) {
        const int r = q / nblk, bq = q - r * nblk;
        const int x = xa + bq * 16 + (lane & 15);
        if (x >= x0 && x < x0 + BW) atomicAdd(d + (long)(y0 + r) * W + x, 1.0f);
      }
    } else if (MODE == 3) {  // 128-byte aligned half-rows: 2 segments of 32 lanes per instruction, masked
      const int xa = x0 & ~31, nblk = (x0 + BW - xa + 31) / 32;
      const int total = BH * nblk;
      float* d = base + (long)(c0 + wave) * plane;
      for (int q = lane >> 5; q < total; q += 2) {
        const int r = q / nblk, bq = q - r * nblk;
        const int x = xa + bq * 32 + (lane & 31);
        if (x >= x0 && x < x0 + BW) atomicAdd(d + (long)(y0 + r) * W + x, 1.0f);
      }
    } else if (MODE == 4) {  // reference: dense 256-byte aligned instructions covering the same number of cells
      float* d = base + (long)(c0 + wave) * plane + (long)y0 * W + (x0 & ~63);
      for (int r = 0; r < (BH * BW + 63) / 64; ++r) atomicAdd(d + r * W + lane, 1.0f);
    } else if (MODE == 5) {  // as 0 but the 4 channels of a cell by 4 consecutive lanes?  no: channel-major per wave, cells strided by 64
      float* d = base + (long)(c0 + wave) * plane;
      for (int i = lane; i < BH * BW; i += 64) {
        const int r = i / BW, cx = i - r * BW;
        atomicAdd(d + (long)(y0 + r) * W + x0 + cx, 1.0f);
      }
    } else if (MODE == 7 || MODE == 8) {  // shape 0 with WORKGROUP / WAVEFRONT-scope atomics: executed in the XCD's L2 instead of memory-side
      // (NOT coherent across XCDs: a throughput probe only -- what an L2-local accumulation could run at)
      for (int i = threadIdx.x; i < BH * BW; i += 256) {
        const int r = i / BW, cx = i - r * BW;
        float* d = base + (long)c0 * plane + (long)(y0 + r) * W + x0 + cx;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          __hip_atomic_fetch_add(d + c * plane, 1.0f, __ATOMIC_RELAXED,
                                 MODE == 7 ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    } else if (MODE == 6) {  // plain stores in shape 5 (what the bytes cost without the atomic unit)
      float* d = base + (long)(c0 + wave) * plane;
      for (int i = lane; i < BH * BW; i += 64) {
        const int r = i / BW, cx = i - r * BW;
        d[(long)(y0 + r) * W + x0 + cx] = 1.0f;
      }
    }
  }
}
template <int MODE>
void run(const char* name, float* dst) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int ntx = W / 32, nty = H / 8, grid = ntx * nty * B;
  float ms = 0;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, dst, ntx, nty);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double cells = (double)grid * C * BH * BW;
  printf("%-64s %8.1f us  %6.1f G cells/s  %5.2f TB/s\n", name, ms * 1e3, cells / ms / 1e6, cells * 4 / ms / 1e9);
}
int main() {
  float* dst;
  const size_t n = (size_t)B * C * H * W;
  (void)hipMalloc(&dst, n * sizeof(float) + 4096);
  (void)hipMemset(dst, 0, n * sizeof(float) + 4096);
  run<0>("0 compacted cells, thread-strided, 4 channels per thread (now)", dst);
  run<1>("1 one 34-cell row per instruction, wave = channel", dst);
  run<2>("2 64-B aligned blocks x4 per instruction, masked, wave = channel", dst);
  run<3>("3 128-B aligned half-rows x2 per instruction, masked", dst);
  run<4>("4 dense aligned 256-B instructions (reference)", dst);
  run<5>("5 compacted cells, lane-strided, wave = channel", dst);
  run<6>("6 plain stores, shape 5", dst);
  run<7>("7 shape 0, workgroup-scope atomics (L2-local; probe only)", dst);
  run<8>("8 shape 0, wavefront-scope atomics (probe only)", dst);
  return 0;
}
