// Exhaustive check (all 2^32 bit patterns) that x / 9.0f (IEEE, correctly rounded) equals the 3-instruction
// sequence  q = x * c;  r = fma(-9, q, x);  q = fma(r, c, q)  with c = RN(1/9) -- used by the SSIM kernels
// for the window means (AvgPool2d divides the window sum by 9).
// hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o div9_check div9_check.hip && ./div9_check
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* bad, unsigned* first) {
  const unsigned long long i0 = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 256ull;
  unsigned long long nb = 0;
  for (unsigned long long i = i0; i < i0 + 256ull; ++i) {
    const float x = __uint_as_float((unsigned)i);
    const float ref = x / 9.0f;
    const float c = 1.0f / 9.0f;
    float q = x * c;
    const float r = fmaf(-9.0f, q, x);
    q = fmaf(r, c, q);
    const bool same = (__float_as_uint(ref) == __float_as_uint(q)) || (ref != ref && q != q);
    if (!same) {
      ++nb;
      atomicMin(first, (unsigned)i);
    }
  }
  if (nb) atomicAdd(bad, nb);
}
int main() {
  unsigned long long* bad;
  unsigned* first;
  hipMalloc(&bad, 8);
  hipMalloc(&first, 4);
  hipMemset(bad, 0, 8);
  hipMemset(first, 0xff, 4);
  hipLaunchKernelGGL(k, dim3(65536), dim3(256), 0, 0, bad, first);
  unsigned long long hb;
  unsigned hf;
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
  hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
  printf("mismatches: %llu of 2^32 (first bit pattern 0x%08x)\n", hb, hf);
  return 0;
}
