// Micro-benchmark: issue rate of v_fma_f32, v_pk_fma_f32 and the transcendentals v_rsq_f32 / v_rcp_f32 on gfx950,
// at 1..4 waves per SIMD (bench.py prices VALU-bound kernels against these ceilings).
// build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 2000

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

__global__ void k_rsq(float* out, float a, float b) {
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = 1.0f + threadIdx.x + i;
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_rsq_f32 %0, %0" : "+v"(r[i]));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the census inner step: 2 rsq + 1 rcp among 10 plain ops (ratio of the real kernels)
__global__ void k_mix(float* out, float a, float b) {
  float r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = 1.0f + threadIdx.x + i;
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      asm volatile("v_rsq_f32 %0, %0" : "+v"(r[i]));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      asm volatile("v_rsq_f32 %0, %0" : "+v"(r[i]));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
      asm volatile("v_mul_f32 %0, %0, %0" : "+v"(r[i]));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
      asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float float2v __attribute__((ext_vector_type(2)));
__global__ void k_pkfma(float* out, float a, float b) {
  float2v r[16];
  float2v av = {a, a}, bv = {b, b};
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = float2v{(float)threadIdx.x + i, (float)i};
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(av), "v"(bv));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i].x + r[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 1024 * 1024 * 4 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wps = 1; wps <= 4; ++wps) {  // waves per SIMD
    const int threads = 64 * 4 * wps;   // one workgroup per CU
    for (int which = 0; which < 4; ++which) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k_fma, dim3(256), dim3(threads), 0, 0, out, 1.0001f, 0.5f);
        else if (which == 1) hipLaunchKernelGGL(k_pkfma, dim3(256), dim3(threads), 0, 0, out, 1.0001f, 0.5f);
        else if (which == 2) hipLaunchKernelGGL(k_rsq, dim3(256), dim3(threads), 0, 0, out, 1.0001f, 0.5f);
        else hipLaunchKernelGGL(k_mix, dim3(256), dim3(threads), 0, 0, out, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double insts = (double)N_ITER * (which == 3 ? 8 * 13 : 16);  // per wave
      const double flops = insts * 64 * (which == 1 ? 4 : 2) * wps * 4 * 256;
      const char* names[4] = {"v_fma_f32   ", "v_pk_fma_f32", "v_rsq_f32   ", "census mix  "};
      printf("%s waves/SIMD=%d: %.3f ms -> %.1f TFLOP/s-equivalent, %.2f T lane-instructions/s, %.2f ns per wave-instruction per SIMD (%.2f cyc @2.4GHz)\n",
             names[which], wps, ms, flops / ms / 1e9, insts * 64 * wps * 4 * 256 / ms / 1e9, ms * 1e6 / (insts * wps),
             ms * 1e6 / (insts * wps) * 2.4);
    }
  }
  return 0;
}
