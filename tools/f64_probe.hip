// tools/f64_probe.hip -- are the device's f64 sqrt / division / reciprocal the correctly rounded IEEE results (= what gcc's x86-64
// code produces)?  Reads N doubles pairs from stdin-file, writes sqrt(|a|), a / b, 1 / b, sin(a), cos(a) for the host to compare.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/f64_probe.hip -o tools/bin/f64_probe ; tools/bin/f64_probe in.bin out.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const double* a, const double* b, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = sqrt(fabs(a[i]));
  out[n + i] = a[i] / b[i];
  out[2 * n + i] = 1.0 / b[i];
  out[3 * n + i] = sin(a[i]);
  out[4 * n + i] = cos(a[i]);
}
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  fseek(f, 0, SEEK_END); const long bytes = ftell(f); fseek(f, 0, SEEK_SET);
  const int n = (int)(bytes / 16);
  std::vector<double> h(2 * (size_t)n), o(5 * (size_t)n);
  if (fread(h.data(), 8, 2 * (size_t)n, f) != 2 * (size_t)n) return 1;
  fclose(f);
  double *da, *dout;
  hipMalloc(&da, 16 * (size_t)n); hipMalloc(&dout, 40 * (size_t)n);
  hipMemcpy(da, h.data(), 16 * (size_t)n, hipMemcpyHostToDevice);
  probe<<<(n + 255) / 256, 256>>>(da, da + n, dout, n);
  if (hipMemcpy(o.data(), dout, 40 * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  f = fopen(argv[2], "wb"); fwrite(o.data(), 8, o.size(), f); fclose(f);
  return 0;
}
