// Micro-benchmark: LDS atomic throughput on gfx950 (float add vs integer add, 32/64 bit, with/without return)
// for a scatter-like address pattern (lane -> base + lane + small jitter).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 256
template <int MODE>
__global__ void k(float* out, int jitter) {
  __shared__ unsigned long long buf64[2048];
  float* bf = reinterpret_cast<float*>(buf64);
  unsigned* bu = reinterpret_cast<unsigned*>(buf64);
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) buf64[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x;
  int acc = 0;
  for (int it = 0; it < N_ITER; ++it) {
    const int idx = (lane + it * 7 + ((lane * jitter) & 3)) & 1023;
    if (MODE == 0) atomicAdd(&bf[idx], 1.0f);
    if (MODE == 1) atomicAdd(&bu[idx], 1u);
    if (MODE == 2) acc += atomicAdd(&bu[idx], 1u);
    if (MODE == 3) atomicAdd(&buf64[idx], 1ull);
    if (MODE == 4) bf[idx] += 1.0f;  // plain RMW (racy) as a reference for LDS access cost
  }
  __syncthreads();
  out[blockIdx.x * blockDim.x + threadIdx.x] = bf[lane] + acc;
}
template <int MODE>
void run(const char* name, float* out, int jitter) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(256), 0, 0, out, jitter);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
  }
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  // per CU: 4 blocks x 4 waves x N_ITER wave-instructions
  const double winst = 16.0 * N_ITER;
  printf("%-22s jitter=%d: %.3f ms -> %.1f ns per wave-instruction per CU (%.0f cycles @2.4GHz)\n", name, jitter, ms,
         ms * 1e6 / winst, ms * 1e6 / winst * 2.4);
}
int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
  for (int j = 0; j < 2; ++j) {
    run<0>("ds_add_f32", out, j);
    run<1>("ds_add_u32", out, j);
    run<2>("ds_add_rtn_u32", out, j);
    run<3>("ds_add_u64", out, j);
    run<4>("plain read+add+write", out, j);
  }
  return 0;
}
