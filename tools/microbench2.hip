// access-pattern study for the per-cell stencil kernels (not part of the product)
#include "../elevation_mapping_cupy_amd/csrc/emap_device.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k_lin(const Cell* __restrict__ cells, float* __restrict__ out, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  const float4* cp = reinterpret_cast<const float4*>(&cells[i]); float4 a = cp[0], b = cp[1];
  out[i] = (a.z + b.z < 0.5f) ? 0.f : b.y;
}
template <int ROWS> __global__ void k_tile(const Cell* __restrict__ cells, float* __restrict__ out, int C) {
  const int tc = threadIdx.x & 63, wv = threadIdx.x >> 6, col = blockIdx.x * 64 + tc, r0 = blockIdx.y * ROWS;
#pragma unroll
  for (int k = 0; k < ROWS / 4; ++k) {
    long c = (long)(r0 + wv + 4 * k) * C + col;
    const float4* cp = reinterpret_cast<const float4*>(&cells[c]); float4 a = cp[0], b = cp[1];
    out[c] = (a.z + b.z < 0.5f) ? 0.f : b.y;
  }
}
// one wave per row segment of 256 cells: lanes read consecutive 16-B halves (fully coalesced 1 KB per instruction)
__global__ void k_rowwave(const Cell* __restrict__ cells, float* __restrict__ out, long n) {
  long base = ((long)blockIdx.x * 256);  // 256 cells per block
  const float4* p = reinterpret_cast<const float4*>(cells + base);
  __shared__ float4 sm[512];
  for (int k = threadIdx.x; k < 512; k += 256) sm[k] = p[k];
  __syncthreads();
  float4 a = sm[2 * threadIdx.x], b = sm[2 * threadIdx.x + 1];
  out[base + threadIdx.x] = (a.z + b.z < 0.5f) ? 0.f : b.y;
}
template <class F> float timeit(F f, int reps = 20) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int r = 0; r < reps; ++r) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / reps;
}
int main() {
  const int C = 1024; const long L = (long)C * C;
  Cell* cells; CK(hipMalloc(&cells, 32 * L)); CK(hipMemset(cells, 0x3f, 32 * L));
  float* out; CK(hipMalloc(&out, 4 * L));
  printf("linear 1 cell/thread        %8.1f us\n", timeit([&] { hipLaunchKernelGGL(k_lin, dim3(L / 256), dim3(256), 0, 0, cells, out, L); }));
  printf("tile 16x64 (4 rows/thread)  %8.1f us\n", timeit([&] { hipLaunchKernelGGL(k_tile<16>, dim3(C / 64, C / 16), dim3(256), 0, 0, cells, out, C); }));
  printf("tile 4x64 (1 row/thread)    %8.1f us\n", timeit([&] { hipLaunchKernelGGL(k_tile<4>, dim3(C / 64, C / 4), dim3(256), 0, 0, cells, out, C); }));
  printf("tile 32x64 (8 rows/thread)  %8.1f us\n", timeit([&] { hipLaunchKernelGGL(k_tile<32>, dim3(C / 64, C / 32), dim3(256), 0, 0, cells, out, C); }));
  printf("row-wave via LDS            %8.1f us\n", timeit([&] { hipLaunchKernelGGL(k_rowwave, dim3(L / 256), dim3(256), 0, 0, cells, out, L); }));
  return 0;
}
