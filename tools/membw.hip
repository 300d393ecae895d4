// tools/membw.hip -- calibrates the achievable HBM copy / read / write rate on the GPU box
// (hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o /tmp/membw && /tmp/membw)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) b[i] = a[i];
}
__global__ void k_copy4(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {   // 4 loads in flight per thread
  size_t i = ((size_t)blockIdx.x * blockDim.x) * 4 + threadIdx.x;
  if (i + 3 * blockDim.x < n) {
    uint4 v0 = a[i], v1 = a[i + blockDim.x], v2 = a[i + 2 * blockDim.x], v3 = a[i + 3 * blockDim.x];
    b[i] = v0; b[i + blockDim.x] = v1; b[i + 2 * blockDim.x] = v2; b[i + 3 * blockDim.x] = v3;
  }
}
struct __attribute__((packed, aligned(4))) U4 { uint32_t x, y, z, w; };
// source misaligned by `off` dwords (dwordx4 loads at 4-byte alignment), destination aligned; rows of 43 lanes
__global__ void k_copy_mis(const uint32_t* __restrict__ a, uint4* __restrict__ b, size_t n, int off) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i + 1 < n) {
    const U4 v = *reinterpret_cast<const U4*>(a + 4 * i + off);
    b[i] = make_uint4(v.x, v.y, v.z, v.w);
  }
}
// like k_pyr_level0: rows of 640 B -> rows of 704 B at +19, 43 of 64 lanes active, 5-dword window + alignbyte
__global__ void k_rows(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows, size_t dst_frame) {
  const int X = 16 + (threadIdx.x & 63) * 16;
  const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int x0 = X - 19;
  if (x0 >= 640 || y >= rows) return;
  const uint8_t* S = src + (size_t)y * 640;
  const uintptr_t r0 = (uintptr_t)S, lo = r0 & ~(uintptr_t)3, hi = (r0 + 639) & ~(uintptr_t)3;
  const intptr_t aa = (intptr_t)r0 + x0;
  const uintptr_t q = (uintptr_t)(aa & ~(intptr_t)3);
  const uint32_t sh = (uint32_t)(aa & 3);
  uint32_t d[5];
#pragma unroll
  for (int j = 0; j < 5; j++) { uintptr_t p = q + 4 * j; p = p < lo ? lo : (p > hi ? hi : p); d[j] = *(const uint32_t*)p; }
  uint4 v;
  v.x = __builtin_amdgcn_alignbyte(d[1], d[0], sh); v.y = __builtin_amdgcn_alignbyte(d[2], d[1], sh);
  v.z = __builtin_amdgcn_alignbyte(d[3], d[2], sh); v.w = __builtin_amdgcn_alignbyte(d[4], d[3], sh);
  *(uint4*)(dst + (size_t)(y / 480) * dst_frame + (size_t)(y % 480) * 704 + X) = v;
}
// same traffic, aligned: rows of 640 B -> rows of 704 B at +16 (40 lanes active)
__global__ void k_rows_al(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows) {
  const int xg = threadIdx.x & 63;
  const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (xg >= 40 || y >= rows) return;
  *(uint4*)(dst + (size_t)y * 704 + 16 + xg * 16) = *(const uint4*)(src + (size_t)y * 640 + xg * 16);
}
__global__ void k_read(const uint4* __restrict__ a, uint32_t* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i < n; i += st) { uint4 v = a[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_write(uint4* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) b[i] = make_uint4(i, 1, 2, 3);
}
int main() {
  for (size_t mb : {80, 320}) {
    const size_t bytes = mb << 20, n = bytes / 16;
    uint4 *a, *b; uint32_t* o;
    hipMalloc(&a, bytes + (16 << 20)); hipMalloc(&b, bytes + (16 << 20)); hipMalloc(&o, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto fn, double moved) {
      for (int i = 0; i < 3; i++) fn();
      hipEventRecord(e0);
      for (int i = 0; i < 10; i++) fn();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%5zu MB %-22s %8.1f us  %7.2f TB/s (bytes moved)\n", mb, name, ms * 100, moved / (ms / 10 * 1e-3) / 1e12);
    };
    timeit("hipMemcpyDtoD", [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, 2.0 * bytes);
    timeit("copy grid-stride 2048x256", [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
    timeit("copy grid-stride 8192x256", [&] { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
    timeit("copy 1 elem/thread", [&] { hipLaunchKernelGGL(k_copy, dim3((unsigned)(n / 256)), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
    timeit("copy 4 elem/thread", [&] { hipLaunchKernelGGL(k_copy4, dim3((unsigned)(n / 1024)), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
    timeit("copy src +1 dword (x4 @4B)", [&] { hipLaunchKernelGGL(k_copy_mis, dim3((unsigned)(n / 256)), dim3(256), 0, 0, (const uint32_t*)a, b, n, 1); }, 2.0 * bytes);
    if (mb == 80) {
      const int rows = 256 * 480;   // 78.6 MB in, 86.5 MB out
      timeit("rows 640->704 +19 (level0)", [&] { hipLaunchKernelGGL(k_rows, dim3(rows / 4), dim3(256), 0, 0, (const uint8_t*)a, (uint8_t*)b, rows, (size_t)480 * 704); }, rows * (640.0 + 678.0));
      {  // pipeline-like footprint: frames 1.25 MB apart in the destination, 4 rotating source/destination sets (no MALL reuse)
        const size_t fb = 1310720;
        uint8_t *S4, *D4; hipMalloc(&S4, 4 * (size_t)rows * 640 + 4096); hipMalloc(&D4, 4 * 256 * fb + 4096);
        hipMemset(S4, 1, 4 * (size_t)rows * 640); hipMemset(D4, 1, 4 * 256 * fb);
        int rot = 0;
        timeit("level0, pyramid layout, rotating", [&] { hipLaunchKernelGGL(k_rows, dim3(rows / 4), dim3(256), 0, 0, S4 + (size_t)(rot & 3) * rows * 640, D4 + (size_t)(rot & 3) * 256 * fb, rows, fb); rot++; }, rows * (640.0 + 678.0));
        hipFree(S4); hipFree(D4);
      }
      timeit("rows 640->704 +16 aligned", [&] { hipLaunchKernelGGL(k_rows_al, dim3(rows / 4), dim3(256), 0, 0, (const uint8_t*)a, (uint8_t*)b, rows); }, rows * (640.0 + 640.0));
    }
    timeit("read only", [&] { hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, a, o, n); }, 1.0 * bytes);
    timeit("write only", [&] { hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, b, n); }, 1.0 * bytes);
    hipFree(a); hipFree(b); hipFree(o);
  }
  return 0;
}
