// dispatch-rate probe: empty kernels, varying workgroup size / count / LDS.  hipcc --offload-arch=gfx950 -O3 tools/dbg/dispatch.hip -o /tmp/dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS> __global__ void k_empty(unsigned int* out) {
  __shared__ unsigned int s[LDS > 0 ? LDS / 4 : 1];
  if (LDS > 0) { s[threadIdx.x] = threadIdx.x; __syncthreads(); if (s[(threadIdx.x + 1) % blockDim.x] == 0xffffffffu) out[0] = 1; }
  else if (out == nullptr) out[1] = 2;
}
template <int LDS> float run(int grid, int block, unsigned int* d) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_empty<LDS>, dim3(grid), dim3(block), 0, 0, d);
  (void)hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_empty<LDS>, dim3(grid), dim3(block), 0, 0, d);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms = 0.f; (void)hipEventElapsedTime(&ms, a, b); return ms * 100.f;   // us per launch
}
int main() {
  unsigned int* d = nullptr; (void)hipMalloc(&d, 64);
  const int grids[] = {1024, 4096, 16384, 65536, 262144};
  const int blocks[] = {64, 256, 512, 1024};
  for (int g : grids) for (int b : blocks) {
    const float t0 = run<0>(g, b, d), t1 = run<16384>(g, b, d), t2 = run<49152>(g, b, d);
    printf("grid %6d block %4d waves %8d : no-lds %8.1f us (%.2f ns/WG, %.3f ns/wave)   lds16k %8.1f   lds48k %8.1f\n", g, b, g * (b / 64), t0, t0 * 1e3 / g, t0 * 1e3 / (g * (b / 64.0)), t1, t2);
  }
  return 0;
}
