// effective VALU issue rate of one SIMD under full occupancy: 8 waves per SIMD, each a chain of dependent fp32 FMAs / packed FMAs /
// conversions.  Prints instructions per nanosecond per SIMD (4-cycle issue at f GHz -> f / 4).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND> __global__ __launch_bounds__(1024, 8) void k(float* out, int iters, float a, float b) {
  float x = threadIdx.x * 1e-3f, y = x + 1.f;
  v2f p = {x, y};
  int sc = iters;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (KIND == 0) { x = __builtin_fmaf(x, a, b); y = __builtin_fmaf(y, a, b); }
      else if (KIND == 1) { p = __builtin_elementwise_fma(p, (v2f){a, a}, (v2f){b, b}); p = __builtin_elementwise_fma(p, (v2f){a, a}, (v2f){b, b}); }
      else if (KIND == 2) { x = (float)(_Float16)x + b; y = (float)(_Float16)y + b; }          // cvt f16, cvt f32, add
      else if (KIND == 3) { x = __builtin_floorf(x) + b; y = (float)(int)y + b; }               // floor, add, cvt i32, cvt f32, add
      else if (KIND == 4) { x = __builtin_fmaf(x, a, b); asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc) : : "scc"); y = __builtin_fmaf(y, a, b); asm volatile("s_xor_b32 %0, %0, 5" : "+s"(sc) : : "scc"); }   // VALU + SALU alternating
      else if (KIND == 5) { x = __builtin_fmaf(x, a, b); asm volatile("s_nop 0"); y = __builtin_fmaf(y, a, b); asm volatile("s_nop 0"); }
      else if (KIND == 6) { x = __builtin_fmaf(x, a, b); y = __builtin_fmaf(y, a, b); unsigned long long m = __builtin_amdgcn_ballot_w64(x > y); sc += (int)(m & 1); }   // v_fma x2, v_cmp -> SGPR -> s_and, s_add
      else if (KIND == 7) { x = __builtin_fmaf(x, a, b); if (__builtin_amdgcn_ballot_w64(x > 1e30f)) { y = __builtin_fmaf(y, a, b); } }   // fma, cmp, branch (never taken)
    }
  }
  out[blockIdx.x * 1024 + threadIdx.x] = x + y + p.x + p.y + (float)sc;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 512, iters = 2000;      // 512 x 16 waves = 8 waves on each of the 1024 SIMDs
  const char* names[8] = {"v_fma_f32 x2", "v_pk_fma_f32 x2", "cvt_f16+cvt_f32+add x2", "floor+add | cvt_i32+cvt_f32+add", "(fma, s_add, fma, s_xor): VALU only counted", "(fma, s_nop, fma, s_nop): VALU only counted", "(fma x2, v_cmp, s_and, s_add): 3 VALU counted", "(fma, v_cmp, s_cbranch): 2 VALU counted"};
  const int per_iter[8] = {32, 32, 96, 80, 32, 32, 48, 32};
  for (int kind = 0; kind < 8; ++kind) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1.0001f, 0.5f);
      if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1.0001f, 0.5f);
      if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1.0001f, 0.5f);
      if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1.0001f, 0.5f);
      if (kind == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1.0001f, 0.5f);
      if (kind == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1.0001f, 0.5f);
      if (kind == 6) hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1.0001f, 0.5f);
      if (kind == 7) hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1.0001f, 0.5f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double insts_per_simd = 8.0 * iters * per_iter[kind];        // wave instructions issued by one SIMD
      if (rep) printf("%-36s %.3f ms  %.3f wave-instr/ns/SIMD (= GHz/4 at full rate)\n", names[kind], ms, insts_per_simd / (ms * 1e6));
    }
  }
  return 0;
}
