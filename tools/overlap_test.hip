// tools/overlap_test.hip -- does a pinned H2D copy overlap a chip-filling kernel on another stream on this box?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(float* out, int iters) {
  float a = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; i++) a = a * 1.0001f + 0.5f;
  if (a == 12345.f) out[0] = a;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t N = 78643200;
  uint8_t *h, *d; float* o;
  hipHostMalloc(&h, N); hipMalloc(&d, N); hipMalloc(&o, 4);
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  auto kern = [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(256 * 8), dim3(256), 0, s, o, 400000); };
  kern(s1); hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, s2); hipDeviceSynchronize();
  double t0 = now(); kern(s1); hipDeviceSynchronize(); double tk = now() - t0;
  t0 = now(); hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, s2); hipDeviceSynchronize(); double tc = now() - t0;
  t0 = now(); kern(s1); hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, s2); hipDeviceSynchronize(); double tb = now() - t0;
  t0 = now(); hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, s2); kern(s1); hipDeviceSynchronize(); double tb2 = now() - t0;
  printf("kernel %.3f ms, copy %.3f ms, both (kernel first) %.3f ms, both (copy first) %.3f ms\n", tk * 1e3, tc * 1e3, tb * 1e3, tb2 * 1e3);
  return 0;
}
