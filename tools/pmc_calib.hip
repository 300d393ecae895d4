// tools/pmc_calib.hip -- known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 per ACCESS WIDTH
// (MI355X_MICROARCH.md, HBM: FETCH_SIZE reports half the bytes of a 16 B / lane stream; other widths uncalibrated).
//   hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/bin/pmc_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- tools/bin/pmc_calib      (and a second pass with WRITE_SIZE)
// Every kernel streams a 512 MiB buffer (twice the 256 MiB MALL) exactly once with one access width; tools/pmc_traffic.py divides
// the counter by the known byte count.  The buffers are re-initialised between kernels by a 1 GiB memset (evicts the MALL).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
template <typename T> __global__ void calib_read(const T* __restrict__ a, uint32_t* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i < n; i += st) {
    T v = a[i];
    const uint8_t* p = reinterpret_cast<const uint8_t*>(&v);
    for (size_t k = 0; k < sizeof(T); k++) acc += p[k];
  }
  if (acc == 0x12345678u) out[0] = acc;
}
template <typename T> __global__ void calib_write(T* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  T v;
  uint8_t* p = reinterpret_cast<uint8_t*>(&v);
  for (size_t k = 0; k < sizeof(T); k++) p[k] = (uint8_t)(k + 1);
  for (; i < n; i += st) b[i] = v;
}
// 16 B per lane at 4-byte alignment (the unaligned dwordx4 patch reads of k_orient_desc / k_fast_cells' tile loads)
struct __attribute__((packed, aligned(4))) U4 { uint32_t x, y, z, w; };
__global__ void calib_read_x4_unaligned(const uint32_t* __restrict__ a, uint32_t* out, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i + 1 < n16; i += st) { const U4 v = *reinterpret_cast<const U4*>(a + 4 * i + 1); acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
int main() {
  const size_t bytes = (size_t)512 << 20;
  void *a, *b, *scratch; uint32_t* o;
  hipMalloc(&a, bytes + 64); hipMalloc(&b, bytes + 64); hipMalloc(&scratch, (size_t)1 << 30); hipMalloc(&o, 4);
  hipMemset(a, 1, bytes);
  auto evict = [&] { hipMemset(scratch, 3, (size_t)1 << 30); hipDeviceSynchronize(); };
  const dim3 g(8192), t(256);
  evict(); hipLaunchKernelGGL(calib_read<uint4>, g, t, 0, 0, (const uint4*)a, o, bytes / 16); hipDeviceSynchronize();
  evict(); hipLaunchKernelGGL(calib_read<uint2>, g, t, 0, 0, (const uint2*)a, o, bytes / 8); hipDeviceSynchronize();
  evict(); hipLaunchKernelGGL(calib_read<uint32_t>, g, t, 0, 0, (const uint32_t*)a, o, bytes / 4); hipDeviceSynchronize();
  evict(); hipLaunchKernelGGL(calib_read<uint8_t>, g, t, 0, 0, (const uint8_t*)a, o, bytes / 4); hipDeviceSynchronize();   // a quarter of the buffer, 1 B / lane
  evict(); hipLaunchKernelGGL(calib_read_x4_unaligned, g, t, 0, 0, (const uint32_t*)a, o, bytes / 16); hipDeviceSynchronize();
  evict(); hipLaunchKernelGGL(calib_write<uint4>, g, t, 0, 0, (uint4*)b, bytes / 16); hipDeviceSynchronize();
  evict(); hipLaunchKernelGGL(calib_write<double>, g, t, 0, 0, (double*)b, bytes / 8); hipDeviceSynchronize();
  evict(); hipLaunchKernelGGL(calib_write<uint32_t>, g, t, 0, 0, (uint32_t*)b, bytes / 4); hipDeviceSynchronize();
  evict(); hipLaunchKernelGGL(calib_write<uint8_t>, g, t, 0, 0, (uint8_t*)b, bytes / 4); hipDeviceSynchronize();
  printf("{\"bytes\": {\"calib_read<uint4>\": %zu, \"calib_read<uint2>\": %zu, \"calib_read<unsigned int>\": %zu, \"calib_read<unsigned char>\": %zu, "
         "\"calib_read_x4_unaligned\": %zu, \"calib_write<uint4>\": %zu, \"calib_write<double>\": %zu, \"calib_write<unsigned int>\": %zu, \"calib_write<unsigned char>\": %zu}}\n",
         bytes, bytes, bytes, bytes / 4, bytes, bytes, bytes, bytes, bytes / 4);
  return 0;
}
