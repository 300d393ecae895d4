// dvm_slam_amd/csrc/f64_spec.h -- sin / cos / cube of a double as a SPECIFICATION: sequences of IEEE +, -, * (and one fma pair in
// the cube) whose result is the same bit pattern on the host and on the device.
//
// Why: g2o's SE3Quat::exp (Thirdparty/g2o/g2o/types/se3quat.h:212-240) calls sin(theta), cos(theta), pow(theta, 3) and the LM
// driver calls pow(2 rho - 1, 3) (optimization_algorithm_levenberg.cpp:131) from libm.  No device libm returns glibc's bits: glibc's
// sin / cos / pow are faithfully (< 1 ulp) but not correctly rounded -- pow(x, 3) differs from the correctly rounded cube on 0.08 % of
// random arguments (tools/dev note in DESIGN.md) -- and ROCm's ocml rounds differently again.  A bundle adjustment that is to be
// BIT-IDENTICAL to its CPU restatement (csrc/ba_window.hip) cannot call either; both sides evaluate this spec instead:
//   * f64_sin / f64_cos: fdlibm's __kernel_sin / __kernel_cos (Sun, 1993: < 1 ulp on |x| <= pi/4) behind fdlibm's medium-size
//     argument reduction by pi/2 in two pieces (first 33 bits + tail: exact for |x| < 2^19 pi/2), Horner form, every product and sum
//     written out -- compiled with -ffp-contract=off on both sides;
//   * f64_cube: t^3 rounded once -- the two products' rounding errors recovered with fma and added back before the final rounding.
// tests/test_f64_spec.py: host build == oracle restatement == device build bit for bit; against glibc: equal on > 99 % of the
// arguments, never more than 1 ulp apart on |x| <= pi/4.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DVM_F64_HD __host__ __device__ __forceinline__
#else
#define DVM_F64_HD inline
#endif

namespace dvm {

DVM_F64_HD uint32_t f64_hi_word(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
DVM_F64_HD double f64_from_hi_word(uint32_t hi) { const uint64_t u = (uint64_t)hi << 32; double x; __builtin_memcpy(&x, &u, 8); return x; }

// x^3, correctly rounded (up to a 2^-106 relative sliver around ties)
DVM_F64_HD double f64_cube(double t) {
  const double t2 = t * t, e2 = __builtin_fma(t, t, -t2);
  const double t3 = t2 * t, e3 = __builtin_fma(t2, t, -t3);
  return t3 + (e3 + e2 * t);
}

// fdlibm k_sin.c: sin(x + y) for |x| <= pi/4, y the tail of x
DVM_F64_HD double f64_kernel_sin(double x, double y, int iy) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  if ((f64_hi_word(x) & 0x7fffffffu) < 0x3e400000u) return x;                 // |x| < 2^-27
  const double z = x * x, v = z * x;
  const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  if (iy == 0) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
// fdlibm k_cos.c: cos(x + y) for |x| <= pi/4
DVM_F64_HD double f64_kernel_cos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const uint32_t ix = f64_hi_word(x) & 0x7fffffffu;
  if (ix < 0x3e400000u) return 1.0;                                            // |x| < 2^-27
  const double z = x * x;
  const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  if (ix < 0x3fd33333u) return 1.0 - (0.5 * z - (z * r - x * y));             // |x| < 0.3
  const double qx = ix > 0x3fe90000u ? 0.28125 : f64_from_hi_word(ix - 0x00200000u);   // ~ |x| / 4
  const double hz = 0.5 * z - qx, a = 1.0 - qx;
  return a - (hz - (z * r - x * y));
}
// fdlibm e_rem_pio2.c, medium path (first iteration only: exact products for |n| < 2^19, a 33 + 53 bit pi/2): x = n pi/2 + (y0 + y1)
DVM_F64_HD int f64_rem_pio2(double x, double* y0, double* y1) {
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
  const double ax = x < 0 ? -x : x;
  const int n = (int)(ax * invpio2 + 0.5);
  const double fn = (double)n;
  const double r = ax - fn * pio2_1, w = fn * pio2_1t;
  double a = r - w, b = (r - a) - w;
  if (x < 0) { *y0 = -a; *y1 = -b; return -n; }
  *y0 = a; *y1 = b;
  return n;
}
DVM_F64_HD double f64_sin(double x) {
  if ((f64_hi_word(x) & 0x7fffffffu) <= 0x3fe921fbu) return f64_kernel_sin(x, 0.0, 0);     // |x| <= ~pi/4
  double y0, y1;
  const int n = f64_rem_pio2(x, &y0, &y1) & 3;
  if (n == 0) return f64_kernel_sin(y0, y1, 1);
  if (n == 1) return f64_kernel_cos(y0, y1);
  if (n == 2) return -f64_kernel_sin(y0, y1, 1);
  return -f64_kernel_cos(y0, y1);
}
DVM_F64_HD double f64_cos(double x) {
  if ((f64_hi_word(x) & 0x7fffffffu) <= 0x3fe921fbu) return f64_kernel_cos(x, 0.0);
  double y0, y1;
  const int n = f64_rem_pio2(x, &y0, &y1) & 3;
  if (n == 0) return f64_kernel_cos(y0, y1);
  if (n == 1) return -f64_kernel_sin(y0, y1, 1);
  if (n == 2) return -f64_kernel_cos(y0, y1);
  return f64_kernel_sin(y0, y1, 1);
}

}  // namespace dvm
