// oracle/f64_spec.h -- TEST INFRASTRUCTURE (CPU oracle).  The double-precision sin / cos / cube SPEC the oracle's SE3 exponential
// and LM decision evaluate where the reference calls libm (Thirdparty/g2o/g2o/types/se3quat.h:212-240: sin, cos, pow(theta, 3);
// g2o/core/optimization_algorithm_levenberg.cpp:131: pow(2 rho - 1, 3)).  Restated here from the published fdlibm algorithms
// (k_sin.c, k_cos.c, e_rem_pio2.c medium path) so that the oracle does not include product code; the product's own statement is
// dvm_slam_amd/csrc/f64_spec.h and tests/test_f64_spec.py holds the two (and the device build) to the same bits, and this one
// to glibc within 1 ulp.  glibc itself is only faithfully rounded: std::pow(x, 3) is not the correctly rounded cube for 0.08 % of
// random x, so "what libm returns" is not a bit pattern another implementation can be asked to reproduce.
#pragma once
#include <cstdint>
#include <cstring>

namespace orc_spec {

inline uint32_t hi_word(double x) { uint64_t u; std::memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
inline double with_hi_word(uint32_t hi) { const uint64_t u = (uint64_t)hi << 32; double x; std::memcpy(&x, &u, 8); return x; }

// correctly rounded x^3: products' rounding errors via fma, one final rounding
#ifdef ORC_LIBM
}  // namespace orc_spec
#include <cmath>
namespace orc_spec {
inline double cube(double t) { return std::pow(t, 3); }
inline double cube_spec(double t) {
#else
inline double cube(double t) {
#endif
  const double sq = t * t, sq_err = __builtin_fma(t, t, -sq);
  const double cu = sq * t, cu_err = __builtin_fma(sq, t, -cu);
  return cu + (cu_err + sq_err * t);
}

inline double ksin(double x, double tail, bool has_tail) {
  static const double S[6] = {-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
                              2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10};
  if ((hi_word(x) & 0x7fffffffu) < 0x3e400000u) return x;
  const double z = x * x, v = z * x;
  const double r = S[1] + z * (S[2] + z * (S[3] + z * (S[4] + z * S[5])));
  if (!has_tail) return x + v * (S[0] + z * r);
  return x - ((z * (0.5 * tail - v * r) - tail) - v * S[0]);
}
inline double kcos(double x, double tail) {
  static const double C[6] = {4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
                              -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11};
  const uint32_t ix = hi_word(x) & 0x7fffffffu;
  if (ix < 0x3e400000u) return 1.0;
  const double z = x * x;
  const double r = z * (C[0] + z * (C[1] + z * (C[2] + z * (C[3] + z * (C[4] + z * C[5])))));
  if (ix < 0x3fd33333u) return 1.0 - (0.5 * z - (z * r - x * tail));
  const double qx = ix > 0x3fe90000u ? 0.28125 : with_hi_word(ix - 0x00200000u);
  const double hz = 0.5 * z - qx, a = 1.0 - qx;
  return a - (hz - (z * r - x * tail));
}
inline int reduce(double x, double& y0, double& y1) {
  const double ax = x < 0 ? -x : x;
  const int n = (int)(ax * 6.36619772367581382433e-01 + 0.5);
  const double fn = (double)n;
  const double r = ax - fn * 1.57079632673412561417e+00, w = fn * 6.07710050650619224932e-11;
  const double a = r - w, b = (r - a) - w;
  y0 = x < 0 ? -a : a;
  y1 = x < 0 ? -b : b;
  return x < 0 ? -n : n;
}
inline bool small(double x) { return (hi_word(x) & 0x7fffffffu) <= 0x3fe921fbu; }
#ifdef ORC_LIBM
// `make libm` (oracle/Makefile): the oracle as a glibc-linked g2o would compute -- std::sin / std::cos / std::pow(x, 3) exactly where
// the reference calls them -- for tests/test_oracle_libm.py, which measures how far "bit-identical to the spec oracle" can sit from that
}  // namespace orc_spec
#include <cmath>
namespace orc_spec {
inline double sin_spec(double x);
inline double cos_spec(double x);
inline double sin(double x) { return std::sin(x); }
inline double cos(double x) { return std::cos(x); }
#define ORC_SPEC_SIN sin_spec
#define ORC_SPEC_COS cos_spec
#else
#define ORC_SPEC_SIN sin
#define ORC_SPEC_COS cos
#endif
inline double ORC_SPEC_SIN(double x) {
  if (small(x)) return ksin(x, 0.0, false);
  double y0, y1;
  switch (reduce(x, y0, y1) & 3) {
    case 0: return ksin(y0, y1, true);
    case 1: return kcos(y0, y1);
    case 2: return -ksin(y0, y1, true);
    default: return -kcos(y0, y1);
  }
}
inline double ORC_SPEC_COS(double x) {
  if (small(x)) return kcos(x, 0.0);
  double y0, y1;
  switch (reduce(x, y0, y1) & 3) {
    case 0: return kcos(y0, y1);
    case 1: return -ksin(y0, y1, true);
    case 2: return -kcos(y0, y1);
    default: return ksin(y0, y1, true);
  }
}

}  // namespace orc_spec
