// Exhaustive check of the 3-instruction division used for the warp coordinates (common.hpp af_div_den):
//     r = RN(1 / d);  q = RN(x r);  e = fma(-q, d, x);  q' = fma(e, r, q)
// against IEEE x / d for every integer divisor d in [1, DMAX] (the divisors are W-1 / H-1) and EVERY float x with
// |x| <= 2^15 (normal, denormal, both signs): prints the number of mismatching (x, d) pairs with |x / d| >= 1e-30 (expected 0)
// and, separately, those below (signed zeros and the denormal range, where a coordinate difference is immaterial).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench/div_exact.hip -o /tmp/div_exact && /tmp/div_exact [DMAX [DMIN [TWO_STEPS]]]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ void check(unsigned long long* bad, unsigned* first, int d0, int d1, unsigned xmax_bits, int two) {
  const unsigned stride = gridDim.x * blockDim.x;
  unsigned long long nb = 0, ntiny = 0;
  for (int d = d0 + blockIdx.y; d <= d1; d += gridDim.y) {
    const float den = (float)d, r = 1.0f / den;
    for (unsigned long long i = blockIdx.x * blockDim.x + threadIdx.x; i <= xmax_bits; i += stride) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float x = __uint_as_float((unsigned)i | (s ? 0x80000000u : 0u));
        const float ref = x / den;
        const float q = x * r;
        const float e = fmaf(-q, den, x);
        float q2 = fmaf(e, r, q);
        if (two) q2 = fmaf(fmaf(-q2, den, x), r, q2);  // second correction step
        if (__float_as_uint(q2) != __float_as_uint(ref)) {
          if (fabsf(ref) < 1e-30f) {  // (-0 vs +0, and quotients near the denormal range: the residual e underflows)
            ++ntiny;
          } else {
            ++nb;
            if (atomicCAS(first, 0u, 1u) == 0u) first[1] = __float_as_uint(x), first[2] = (unsigned)d;
          }
        }
      }
    }
  }
  if (nb) atomicAdd(bad, nb);
  if (ntiny) atomicAdd(bad + 1, ntiny);
}

int main(int argc, char** argv) {
  const int dmax = argc > 1 ? atoi(argv[1]) : 2047;
  const int dmin = argc > 2 ? atoi(argv[2]) : 1;
  const int two = argc > 3 ? atoi(argv[3]) : 0;
  unsigned long long* bad;
  unsigned* first;
  hipMalloc(&bad, 16);
  hipMalloc(&first, 16);
  hipMemset(bad, 0, 16);
  hipMemset(first, 0, 16);
  const float xmax = 32768.f;
  unsigned xb;
  std::memcpy(&xb, &xmax, 4);
  for (int d0 = dmin; d0 <= dmax; d0 += 64) {
    const int d1 = d0 + 63 < dmax ? d0 + 63 : dmax;
    hipLaunchKernelGGL(check, dim3(4096, d1 - d0 + 1), dim3(256), 0, 0, bad, first, d0, d1, xb, two);
    hipDeviceSynchronize();
    unsigned long long h;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    if (((d0 - dmin) / 64) % 16 == 0) {
      printf("divisors %d..%d done, mismatches so far %llu\n", d0, d1, h);
      fflush(stdout);
    }
  }
  unsigned long long h, hh[2];
  unsigned f[4];
  hipMemcpy(hh, bad, 16, hipMemcpyDeviceToHost);
  h = hh[0];
  hipMemcpy(f, first, 16, hipMemcpyDeviceToHost);
  printf("RESULT (%d correction step%s) divisors %d..%d, |x| <= 2^15 (all %u bit patterns x 2 signs): %llu mismatches", two ? 2 : 1, two ? "s" : "", dmin, dmax, xb + 1, h);
  if (h) printf(" (first: x bits 0x%08x, d = %u)", f[1], f[2]);
  printf("; %llu more where |x / d| < 1e-30 (signed zero / denormal range)\n", hh[1]);
  return h ? 1 : 0;
}
