/* Plain-C restatement of the cost-volume arithmetic (test infrastructure; never linked by the product).
 * Follows models/correlation_native.py:13-23 (forward) and the closed-form gradients verified in
 * SURVEY section 2.1 (same quantities as correlation_cuda_kernel.cu:116-300).  Scalar, single thread,
 * double accumulation: an independent third implementation next to oracle/ops.py and the HIP kernels. */
#include <stddef.h>

#define AT(p, b, c, y, x) (p)[(((size_t)(b) * C + (c)) * H + (y)) * W + (x)]

void corr_oracle_fwd(const float* x1, const float* x2, float* out, int B, int C, int H, int W, int d) {
  const int n = 2 * d + 1;
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        for (int y = 0; y < H; ++y)
          for (int x = 0; x < W; ++x) {
            const int yy = y + i - d, xx = x + j - d;
            double s = 0.0;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W)
              for (int c = 0; c < C; ++c) s += (double)AT(x1, b, c, y, x) * (double)AT(x2, b, c, yy, xx);
            out[(((size_t)b * n * n + i * n + j) * H + y) * W + x] = (float)(s / C);
          }
}

void corr_oracle_bwd(const float* g, const float* x1, const float* x2, float* gx1, float* gx2, int B, int C,
                     int H, int W, int d) {
  const int n = 2 * d + 1;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          double s1 = 0.0, s2 = 0.0;
          for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
              const size_t ch = (size_t)b * n * n + i * n + j;
              const int ya = y + i - d, xa = x + j - d;
              if (ya >= 0 && ya < H && xa >= 0 && xa < W)
                s1 += (double)g[(ch * H + y) * W + x] * (double)AT(x2, b, c, ya, xa);
              const int yb = y - i + d, xb = x - j + d;
              if (yb >= 0 && yb < H && xb >= 0 && xb < W)
                s2 += (double)g[(ch * H + yb) * W + xb] * (double)AT(x1, b, c, yb, xb);
            }
          AT(gx1, b, c, y, x) = (float)(s1 / C);
          AT(gx2, b, c, y, x) = (float)(s2 / C);
        }
}
