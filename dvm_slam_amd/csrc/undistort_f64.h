// dvm_slam_amd/csrc/undistort_f64.h -- cv::undistortPoints(src, dst, K, D, noArray(), P = K) for one point, as
// Frame::UndistortKeyPoints / ComputeImageBounds call it (reference src/Frame.cc:791-848): OpenCV 4.x
// cvUndistortPointsInternal with the default criteria (5 fixed-point iterations, no epsilon test), everything in double
// from the float inputs, one rounding to float at the end.  Written operation by operation in OpenCV's order (C precedence,
// no FMA contraction: the library is built with -ffp-contract=off); the tilt model (k[12], k[13] = 0: identity), R = I and
// the zero rational / thin-prism coefficients are exact identities and appear only where they could change a sign of zero.
#pragma once
#ifndef DVM_HD
#ifdef __HIPCC__
#define DVM_HD __host__ __device__ __forceinline__
#else
#define DVM_HD inline
#endif
#endif

namespace dvm_undistort {

struct Camera { float fx, fy, cx, cy, k1, k2, p1, p2, k3; };   // Pinhole::toK() and mDistCoef are CV_32F

DVM_HD void undistort_point(const Camera& c, float u_, float v_, float* xo, float* yo) {
  const double fx = (double)c.fx, fy = (double)c.fy, cx = (double)c.cx, cy = (double)c.cy;   // cvConvert(_cameraMatrix, &matA)
  const double k0 = (double)c.k1, k1 = (double)c.k2, k2 = (double)c.p1, k3 = (double)c.p2, k4 = (double)c.k3;
  const double ifx = 1. / fx, ify = 1. / fy;
  const double u = (double)u_, v = (double)v_;
  double x = (u - cx) * ifx, y = (v - cy) * ify;
  const double x0 = x, y0 = y;
  for (int j = 0; j < 5; j++) {
    const double r2 = x * x + y * y;
    const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
    if (icdist < 0) {   // OpenCV >= 4.1 (regression 14583): give up, keep the normalised input
      x = (u - cx) * ifx;
      y = (v - cy) * ify;
      break;
    }
    const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + 0. * r2 + 0. * r2 * r2;
    const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + 0. * r2 + 0. * r2 * r2;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  // RR = P * I: xx = RR[0][0] x + RR[0][1] y + RR[0][2], ww = 1 / (0 x + 0 y + 1)
  const double xx = fx * x + 0. * y + cx;
  const double yy = 0. * x + fy * y + cy;
  const double ww = 1. / (0. * x + 0. * y + 1.);
  *xo = (float)(xx * ww);
  *yo = (float)(yy * ww);
}

}  // namespace dvm_undistort
