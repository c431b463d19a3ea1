// Issue cost of single instruction kinds on one gfx950 SIMD at full occupancy (8 waves per SIMD, 2 independent chains per wave):
// nanoseconds of SIMD time per wave instruction.  v_fma_f32 is the unit.  (tools/: measurement aid, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(NAME, BODY) __global__ __launch_bounds__(1024, 8) void NAME(float* out, int iters, float a, float b) { \
  float x = threadIdx.x * 1e-3f + 1.f, y = x + 1.f; int ix = threadIdx.x, iy = threadIdx.x * 3; float2 px = {x, y}, py = {y, x}; int sc = 0; \
  for (int i = 0; i < iters; ++i) { REP16(BODY) } \
  out[blockIdx.x * 1024 + threadIdx.x] = x + y + (float)ix + (float)iy + px.x + px.y + py.x + py.y + (float)sc; }
KERNEL(k_fma,     asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(x), "+v"(y) : "v"(a), "v"(b));)
KERNEL(k_pkmul,   asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %2, %2, %1" : "+v"(px) : "v"(py), "v"(py));)
KERNEL(k_pkadd,   asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(px) : "v"(py)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(py) : "v"(px));)
KERNEL(k_cvt16,   asm volatile("v_cvt_f16_f32 %0, %0\n v_cvt_f16_f32 %1, %1" : "+v"(x), "+v"(y));)
KERNEL(k_fmamix,  asm volatile("v_fma_mix_f32 %0, %0, %2, %3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %1, %2, %3 op_sel_hi:[1,0,0]" : "+v"(x), "+v"(y) : "v"(a), "v"(b));)
KERNEL(k_floor,   asm volatile("v_floor_f32 %0, %0\n v_floor_f32 %1, %1" : "+v"(x), "+v"(y));)
KERNEL(k_med3,    asm volatile("v_med3_f32 %0, %0, %2, %3\n v_med3_f32 %1, %1, %2, %3" : "+v"(x), "+v"(y) : "v"(a), "v"(b));)
KERNEL(k_cvti,    asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1" : "+v"(x), "+v"(y));)
KERNEL(k_mad24,   asm volatile("v_mad_u32_u24 %0, %0, %2, %1\n v_mad_u32_u24 %1, %1, %2, %0" : "+v"(ix), "+v"(iy) : "v"(sc));)
KERNEL(k_bfe,     asm volatile("v_bfe_u32 %0, %0, %1, 1\n v_bfe_u32 %1, %1, %0, 1" : "+v"(ix), "+v"(iy));)
KERNEL(k_lshlor,  asm volatile("v_lshl_or_b32 %0, %0, 16, %1\n v_lshl_or_b32 %1, %1, 16, %0" : "+v"(ix), "+v"(iy));)
KERNEL(k_cmpcnd,  asm volatile("v_cmp_ne_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ix), "+v"(iy) : : "vcc");)
KERNEL(k_readlane, asm volatile("v_readlane_b32 %0, %1, 5\n v_readlane_b32 %0, %2, 7" : "+s"(sc) : "v"(ix), "v"(iy));)
KERNEL(k_rl_use,  asm volatile("v_readlane_b32 %0, %1, 5\n v_add_u32 %1, %1, %0" : "+s"(sc), "+v"(ix));)
KERNEL(k_min_s,   asm volatile("v_min_f32 %0, %2, %0\n v_min_f32 %1, %2, %1" : "+v"(x), "+v"(y) : "s"(a));)
KERNEL(k_salu,    asm volatile("s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 5" : "+s"(sc) : : "scc");)
KERNEL(k_nop,     asm volatile("s_nop 0\n s_nop 0");)
typedef void (*kern_t)(float*, int, float, float);
int main() {
  float* out; hipMalloc(&out, 4096 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 512, iters = 1000;
  struct { const char* n; kern_t k; } ks[] = {{"v_fma_f32", k_fma}, {"v_pk_mul_f32", k_pkmul}, {"v_pk_add_f32", k_pkadd}, {"v_cvt_f16_f32", k_cvt16},
    {"v_fma_mix_f32", k_fmamix}, {"v_floor_f32", k_floor}, {"v_med3_f32", k_med3}, {"v_cvt_i32_f32", k_cvti}, {"v_mad_u32_u24", k_mad24},
    {"v_bfe_u32", k_bfe}, {"v_lshl_or_b32", k_lshlor}, {"v_cmp + v_cndmask (pair)", k_cmpcnd}, {"v_readlane_b32", k_readlane},
    {"v_readlane -> VALU use (pair)", k_rl_use}, {"v_min_f32 sgpr,vgpr", k_min_s}, {"s_add / s_xor", k_salu}, {"s_nop", k_nop}};
  for (auto& e : ks) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(1024), 0, 0, out, iters, 1.0001f, 0.5f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double insts = 8.0 * iters * 32;      // wave instructions one SIMD issues (2 per body, 16 bodies, 8 waves)
    printf("%-32s %.3f ms  %.2f ns of SIMD time per wave instruction\n", e.n, best, best * 1e6 / insts);
  }
  return 0;
}
