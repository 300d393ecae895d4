// dvm_slam_amd/csrc/jacobi4.h -- cyclic Jacobi diagonalisation of a symmetric 4x4 matrix in double: 16 sweeps over the six
// rotations (0,1) (0,2) (0,3) (1,2) (1,3) (2,3), exact zeros skipped.  On return A is (numerically) diagonal and the columns of
// V are the eigenvectors.  Used where the reference calls Eigen's float eigen / singular value solvers on 4x4 matrices
// (Sim3Solver::ComputeSim3, GeometricTools::Triangulate): tolerance parity there, the same operation sequence as the oracle here.
#pragma once
#include <hip/hip_runtime.h>

namespace dvm {
__device__ __forceinline__ void jacobi4_dev(double A[4][4], double V[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 16; sweep++) {
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int q = p + 1; q < 4; q++) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
        for (int k = 0; k < 4; k++) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq; }
#pragma unroll
        for (int k = 0; k < 4; k++) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk; }
#pragma unroll
        for (int k = 0; k < 4; k++) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq; }
      }
  }
}
}  // namespace dvm
