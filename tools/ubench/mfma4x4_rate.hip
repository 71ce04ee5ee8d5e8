// Micro-benchmark: issue rate of v_mfma_f32_4x4x1_16b_f32 (sixteen independent 4x4 outer products = 256 MACs per wave
// instruction) against v_fma_f32 (64 MACs) on gfx950, at 1..4 waves per SIMD -- the premise of the banded-correlation
// note in DESIGN.md section 4.1 (a block = 4 pixels x 4 shifted columns; 36 of 48 products useful).
// build: hipcc --offload-arch=gfx950 -O3 mfma4x4_rate.hip -o mfma4x4_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 2000
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_fma(float* out, float a, float b) {
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = threadIdx.x + i;
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mfma(float* out, float a, float b) {
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float av = a + threadIdx.x, bv = b - threadIdx.x;
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[i], 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, int waves_per_simd, double macs_per_inst, int insts_per_iter) {
  float* out;
  const int blocks = 256 * 4 * waves_per_simd;  // one 64-thread block = one wave
  (void)hipMalloc(&out, (size_t)blocks * 64 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double inst = (double)blocks * N_ITER * insts_per_iter;
  const double cyc = ms * 1e-3 * 2.4e9 / (inst / (256.0 * 4));
  printf("%-28s waves/SIMD=%d: %.3f ms  %.1f T MAC/s  %.2f cycles per wave-instruction per SIMD (@2.4 GHz)\n", name, waves_per_simd,
         ms, inst * macs_per_inst / ms / 1e9, cyc);
  (void)hipFree(out);
}

int main() {
  for (int w = 1; w <= 4; ++w) {
    run("v_fma_f32", k_fma, w, 64.0, 16);
    run("v_mfma_f32_4x4x1_16b_f32", k_mfma, w, 256.0, 8);
  }
  return 0;
}
