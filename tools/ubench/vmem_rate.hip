// Micro-benchmark: L1/TA cost of global loads by access shape on gfx950 (data L2-resident, 4 MB buffer).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 512
template <int MODE>
__global__ void k(const float* __restrict__ src, float* out, int W) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  float acc = 0.f;
  const int nfl = 1 << 20;  // 4 MB of floats
  for (int it = 0; it < N_ITER; ++it) {
    const int base = ((wave * 977 + it * 131) * 64) & (nfl - 1024);
    if (MODE == 0) acc += src[base + lane];                                  // dword, aligned, contiguous
    if (MODE == 1) acc += src[base + lane + 1];                              // dword, contiguous, off by one
    if (MODE == 2) acc += src[base + (lane & 31) + (lane >> 5) * W];         // dword, two rows of 32
    if (MODE == 3) acc += src[base + (lane & 31) + (lane >> 5) * W + 3];     // two rows, misaligned
    if (MODE == 4) { const float4 v = *reinterpret_cast<const float4*>(src + base + 4 * lane); acc += v.x + v.y + v.z + v.w; }  // dwordx4
    if (MODE == 5) { const float2 v = *reinterpret_cast<const float2*>(src + base + 2 * lane); acc += v.x + v.y; }  // dwordx2
    if (MODE == 6) acc += src[base + ((lane * 5) & 63)];                     // dword, permuted within 256 B
    if (MODE == 7) acc += src[base + lane + ((lane >> 3) & 1)];             // dword, duplicates/gaps every 8 lanes (compressed flow)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE>
void run(const char* name, const float* src, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, src, out, 160);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
  }
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double winst = 8.0 * 4 * N_ITER;  // wave-instructions per CU
  printf("%-44s %.3f ms -> %6.1f ns per wave-load per CU (%5.1f cyc @2.4GHz)\n", name, ms, ms * 1e6 / winst,
         ms * 1e6 / winst * 2.4);
}
int main() {
  float *src, *out;
  (void)hipMalloc(&src, (1 << 20) * sizeof(float) + 4096);
  (void)hipMemset(src, 0, (1 << 20) * sizeof(float) + 4096);
  (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  run<0>("dword aligned contiguous", src, out);
  run<1>("dword contiguous +1", src, out);
  run<2>("dword two rows of 32", src, out);
  run<3>("dword two rows of 32, +3", src, out);
  run<4>("dwordx4 contiguous (1 KB/wave)", src, out);
  run<5>("dwordx2 contiguous", src, out);
  run<6>("dword permuted within 256 B", src, out);
  run<7>("dword with duplicate/gap every 8 lanes", src, out);
  return 0;
}
