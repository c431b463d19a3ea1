// probe: global_load_lds with 12-byte elements -- LDS stride per lane and source alignment rules on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
template <int OFF> __global__ void k(const float* __restrict__ src, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 64 * 4 + 16; i += 64) lds[i] = -1.f;
  __syncthreads();
  const float* g = src + 4 * threadIdx.x + OFF;          // lane i reads floats [4 i + OFF, 4 i + OFF + 3)
  __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)lds, 12, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 4 + 16; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<float> h(64 * 4 + 64);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
  float *d, *o; hipMalloc(&d, h.size() * 4); hipMalloc(&o, (64 * 4 + 16) * 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int off = 0; off < 2; ++off) {
    if (off == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 2048, 0, d, o); else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 2048, 0, d, o);
    hipError_t e = hipDeviceSynchronize();
    printf("offset %d floats: %s\n", off, hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    std::vector<float> r(64 * 4 + 16); hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    printf("  lds[0..15]: "); for (int i = 0; i < 16; ++i) printf("%g ", r[i]); printf("\n  lds[186..197]: "); for (int i = 186; i < 198; ++i) printf("%g ", r[i]); printf("\n");
  }
  return 0;
}
