// accuracy of v_rsq_f64 and of one / two Newton steps on it (gfx950): max relative error against long-double 1/sqrt
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* y0, double* y1, double* y2, int n) {
  int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  double d = x[i], y = __builtin_amdgcn_rsq(d); y0[i] = y;
  double c = -0.5 * d, t = c * y; t = __builtin_fma(t, y, 0.5); y = __builtin_fma(y, t, y); y1[i] = y;
  t = c * y; t = __builtin_fma(t, y, 0.5); y = __builtin_fma(y, t, y); y2[i] = y;
  { double z = y0[i], r = __builtin_fma((-d) * z, z, 1.0), p = __builtin_fma(0.375, r, 0.5), zr = z * r; y1[i] = __builtin_fma(zr, p, z); }   // one cubic step
}
int main() {
  const int n = 1 << 22; std::vector<double> x(n), a(n), b(n), c(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); x[i] = std::exp((u - 0.5) * 60.0); }
  double *dx, *d0, *d1, *d2; hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
  long double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; i++) { long double r = 1.0L / sqrtl((long double)x[i]);
    e0 = fmaxl(e0, fabsl(a[i] - r) / r); e1 = fmaxl(e1, fabsl(b[i] - r) / r); e2 = fmaxl(e2, fabsl(c[i] - r) / r); }
  printf("max rel err: rsq %.3Le (2^%.1Lf)  one cubic step %.3Le (%.2Lf ulp)  +2 Newton %.3Le (%.2Lf ulp)\n", e0, log2l(e0), e1, e1 / 1.11e-16L, e2, e2 / 1.11e-16L);
  return 0;
}
