// Micro-benchmarks behind DESIGN.md's kernel decisions (not part of the product).  hipcc --offload-arch=gfx950 -O3
// Measures the cost components of the point->cell scatter on MI355X: geometry ALU, cell gather, atomics by width /
// scope / record layout / spatial coherence.
#include "../elevation_mapping_cupy_amd/csrc/emap_device.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_geom(KP P, Pose T, const float* pts, long n, int* out) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  float x, y, z; load_point(pts, i, 3, x, y, z);
  Geo g = geometry<0>(P, T, x, y, z);
  out[i] = (g.valid && g.inside) ? P.C * g.ix + g.iy : -1;
}
__global__ void k_geom_gather(KP P, Pose T, const float* pts, long n, const Cell* cells, float* out) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  float x, y, z; load_point(pts, i, 3, x, y, z);
  Geo g = geometry<0>(P, T, x, y, z);
  float r = 0;
  if (g.valid && g.inside) { float4 m = *reinterpret_cast<const float4*>(&cells[(long)P.C * g.ix + g.iy]); r = m.x + m.y + m.z + m.w; }
  out[i] = r;
}
__global__ void k_idx_gather(const int* idx, long n, const Cell* cells, float* out) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  int c = idx[i]; float r = 0;
  if (c >= 0) { float4 m = *reinterpret_cast<const float4*>(&cells[c]); r = m.x + m.y + m.z + m.w; }
  out[i] = r;
}
template <int STRIDE_B, int SCOPE> __global__ void k_atomic64(const int* idx, long n, char* acc) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  int c = idx[i]; if (c < 0) return;
  unsigned long long* p = reinterpret_cast<unsigned long long*>(acc + (long)c * STRIDE_B);
  if (SCOPE == 0) atomicAdd(p, 1ull);
  else __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__global__ void k_atomic32(const int* idx, long n, unsigned int* acc) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  int c = idx[i]; if (c < 0) return;
  atomicAdd(&acc[c], 1u);
}
__global__ void k_store64(const int* idx, long n, char* acc) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  int c = idx[i]; if (c < 0) return;
  *reinterpret_cast<unsigned long long*>(acc + (long)c * 40) = (unsigned long long)i;
}
template <int STRIDE_B, int NATOM> __global__ void k_fuse_like(const int* idx, long n, char* acc) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  int c = idx[i]; if (c < 0) return;
  unsigned long long* p = reinterpret_cast<unsigned long long*>(acc + (long)c * STRIDE_B);
  atomicAdd(p + 1, 1ull);
  if (NATOM > 1) atomicAdd(p + 2, (unsigned long long)i);
  if (NATOM > 2) atomicAdd(p + 3, (unsigned long long)(i * 3));
  if (NATOM > 3) atomicMax(p + 4, (unsigned long long)i);
}
// SoA variant: 4 planes
__global__ void k_fuse_soa(const int* idx, long n, unsigned long long* a, long L) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  int c = idx[i]; if (c < 0) return;
  atomicAdd(a + c, 1ull); atomicAdd(a + L + c, (unsigned long long)i); atomicAdd(a + 2 * L + c, (unsigned long long)(i * 3)); atomicMax(a + 3 * L + c, (unsigned long long)i);
}
// full count kernel clone (geometry + gather + atomic)
__global__ void k_count_like(KP P, Pose T, const float* pts, long n, const Cell* cells, AccF* acc) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  float x, y, z; load_point(pts, i, 3, x, y, z);
  Geo g = geometry<0>(P, T, x, y, z);
  if (g.valid && g.inside) {
    long c = (long)P.C * g.ix + g.iy;
    float4 m = *reinterpret_cast<const float4*>(&cells[c]);
    unsigned inl = m.z > 0.5f && m.y < 0.05f;
    atomicAdd(&acc[c].pts_inl, 1ull | ((unsigned long long)inl << 32));
  }
}

template <class F> float timeit(F f, int reps = 20) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int r = 0; r < reps; ++r) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 1024; const long N = argc > 2 ? atol(argv[2]) : 1000000; const long L = (long)C * C;
  KP P; memset(&P, 0, sizeof P);
  P.C = C; P.mode = 0; P.row0 = 0; P.nrows = C; P.halo = 0; P.edge = 1; P.dil = 3; P.res = 0.04; P.half_w = 0.5 * C; P.snf = 0.05; P.mt = 2.0;
  P.mvd2 = 0.25; P.mhr = 1.0; P.ra = 0.3; P.rb = 1.0; P.rc = 0.2; P.q_wm1 = (float)(_Float16)(float)(C - 1);
  Pose T; memset(&T, 0, sizeof T); T.Rq[0] = T.Rq[4] = T.Rq[8] = 1.f; T.tq[2] = T.t[2] = 1.f;
  std::vector<float> h(3 * N); std::mt19937 rng(0); std::uniform_real_distribution<float> U(-C * 0.02f, C * 0.02f), Z(-0.5f, 0.5f);
  for (long i = 0; i < N; ++i) { h[3 * i] = U(rng); h[3 * i + 1] = U(rng); h[3 * i + 2] = Z(rng); }
  float* pts; CK(hipMalloc(&pts, 12 * N)); CK(hipMemcpy(pts, h.data(), 12 * N, hipMemcpyHostToDevice));
  int* idx; CK(hipMalloc(&idx, 4 * N)); float* outf; CK(hipMalloc(&outf, 4 * N));
  Cell* cells; CK(hipMalloc(&cells, 32 * L)); CK(hipMemset(cells, 0, 32 * L));
  char* acc; CK(hipMalloc(&acc, 64 * L)); CK(hipMemset(acc, 0, 64 * L));
  dim3 g((N + 255) / 256), b(256);
  printf("C=%d N=%ld (times in us per launch)\n", C, N);
  printf("geometry only            %8.1f\n", timeit([&] { hipLaunchKernelGGL(k_geom, g, b, 0, 0, P, T, pts, N, idx); }));
  printf("geometry + gather16      %8.1f\n", timeit([&] { hipLaunchKernelGGL(k_geom_gather, g, b, 0, 0, P, T, pts, N, cells, outf); }));
  printf("idx + gather16           %8.1f\n", timeit([&] { hipLaunchKernelGGL(k_idx_gather, g, b, 0, 0, idx, N, cells, outf); }));
  printf("count-like (geo+gather+atomic64 @40B) %8.1f\n", timeit([&] { hipLaunchKernelGGL(k_count_like, g, b, 0, 0, P, T, pts, N, cells, (AccF*)acc); }));
  printf("idx + atomic64 @40B      %8.1f\n", timeit([&] { hipLaunchKernelGGL((k_atomic64<40, 0>), g, b, 0, 0, idx, N, acc); }));
  printf("idx + atomic64 @64B      %8.1f\n", timeit([&] { hipLaunchKernelGGL((k_atomic64<64, 0>), g, b, 0, 0, idx, N, acc); }));
  printf("idx + atomic64 @8B       %8.1f\n", timeit([&] { hipLaunchKernelGGL((k_atomic64<8, 0>), g, b, 0, 0, idx, N, acc); }));
  printf("idx + atomic64 @40B wg-scope %8.1f\n", timeit([&] { hipLaunchKernelGGL((k_atomic64<40, 1>), g, b, 0, 0, idx, N, acc); }));
  printf("idx + atomic32 @4B       %8.1f\n", timeit([&] { hipLaunchKernelGGL(k_atomic32, g, b, 0, 0, idx, N, (unsigned*)acc); }));
  printf("idx + plain store64 @40B %8.1f\n", timeit([&] { hipLaunchKernelGGL(k_store64, g, b, 0, 0, idx, N, acc); }));
  printf("fuse-like 4 atomics @40B %8.1f\n", timeit([&] { hipLaunchKernelGGL((k_fuse_like<40, 4>), g, b, 0, 0, idx, N, acc); }));
  printf("fuse-like 4 atomics @64B %8.1f\n", timeit([&] { hipLaunchKernelGGL((k_fuse_like<64, 4>), g, b, 0, 0, idx, N, acc); }));
  printf("fuse-like 3 atomics @40B %8.1f\n", timeit([&] { hipLaunchKernelGGL((k_fuse_like<40, 3>), g, b, 0, 0, idx, N, acc); }));
  printf("fuse-like 2 atomics @40B %8.1f\n", timeit([&] { hipLaunchKernelGGL((k_fuse_like<40, 2>), g, b, 0, 0, idx, N, acc); }));
  printf("fuse-like 1 atomic  @40B %8.1f\n", timeit([&] { hipLaunchKernelGGL((k_fuse_like<40, 1>), g, b, 0, 0, idx, N, acc); }));
  printf("fuse-like 4 atomics SoA  %8.1f\n", timeit([&] { hipLaunchKernelGGL(k_fuse_soa, g, b, 0, 0, idx, N, (unsigned long long*)acc, L); }));
  // spatially coherent order: sort points by cell index on the host
  std::vector<int> hidx(N); CK(hipMemcpy(hidx.data(), idx, 4 * N, hipMemcpyDeviceToHost));
  std::vector<long> order(N); for (long i = 0; i < N; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](long a, long b2) { return hidx[a] < hidx[b2]; });
  std::vector<float> hs(3 * N); std::vector<int> sidx(N);
  for (long i = 0; i < N; ++i) { for (int k = 0; k < 3; ++k) hs[3 * i + k] = h[3 * order[i] + k]; sidx[i] = hidx[order[i]]; }
  CK(hipMemcpy(pts, hs.data(), 12 * N, hipMemcpyHostToDevice)); CK(hipMemcpy(idx, sidx.data(), 4 * N, hipMemcpyHostToDevice));
  printf("SORTED count-like        %8.1f\n", timeit([&] { hipLaunchKernelGGL(k_count_like, g, b, 0, 0, P, T, pts, N, cells, (AccF*)acc); }));
  printf("SORTED idx + atomic64 @40B %6.1f\n", timeit([&] { hipLaunchKernelGGL((k_atomic64<40, 0>), g, b, 0, 0, idx, N, acc); }));
  printf("SORTED fuse-like 4 @40B  %8.1f\n", timeit([&] { hipLaunchKernelGGL((k_fuse_like<40, 4>), g, b, 0, 0, idx, N, acc); }));
  printf("SORTED idx + gather16    %8.1f\n", timeit([&] { hipLaunchKernelGGL(k_idx_gather, g, b, 0, 0, idx, N, cells, outf); }));
  return 0;
}
