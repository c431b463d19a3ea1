// Would a ONE-PASS partition (block-local counting sort + decoupled look-back for the bin bases, round-3 verdict item 4) beat
// k_bin_hist + k_bin_scan + k_bin_scatter?  Its look-back step has no counterpart in the three-kernel form: every one of the B
// blocks (all co-resident: B <= 256 CUs) publishes its T bin counts and then needs, for every bin, the sum over ALL predecessor
// blocks -- with everybody publishing at the same moment there are no inclusive prefixes to stop at, so block b reads b rows of T
// counts (device-coherent loads: the rows come from other CUs).  This probe times exactly that step, stripped of everything else
// (no points, no geometry, no records): publish a row, ticket, read and sum the predecessors' rows.  The three-kernel form pays
// 7.1 us for k_bin_scan (rocprofv3, 245 x 1025) + one kernel boundary (~1.5 us).
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/lookback.hip -o /tmp/lookback && /tmp/lookback
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(1024) void k_lookback(unsigned int* __restrict__ rows, unsigned int* __restrict__ flags, unsigned int* __restrict__ out, int T, int pitch, unsigned int epoch) {
  const int b = blockIdx.x;
  // 1. publish this block's counts (stand-in values) as device-coherent stores, then the flag
  for (int t = threadIdx.x; t < T; t += 1024) __hip_atomic_store(&rows[(long)b * pitch + t], (unsigned int)(b + t) & 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&flags[b * 32], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // 2. look back: every predecessor's row, bin by bin (one thread per bin, 16 rows in flight)
  for (int t = threadIdx.x; t < T; t += 1024) {
    unsigned int sum = 0u;
    for (int p0 = 0; p0 < b; p0 += 16) {
      unsigned int v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int p = p0 + j;
        v[j] = 0u;
        if (p < b) {
          if (t < 64) { int spins = 0; while (__hip_atomic_load(&flags[p * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch && ++spins < 200000) __builtin_amdgcn_s_sleep(1); }     // (one wave polls, bounded; the others do not wait at all: a lower bound)
          v[j] = __hip_atomic_load(&rows[(long)p * pitch + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) sum += v[j];
    }
    out[(long)b * pitch + t] = sum;
  }
}
__global__ void k_empty() {}

int main() {
  for (int T : {1025, 4097, 16385}) {
    const int B = 245, pitch = (T + 3) & ~3;
    unsigned int *rows, *flags, *out;
    hipMalloc(&rows, sizeof(unsigned int) * (size_t)B * pitch); hipMalloc(&flags, sizeof(unsigned int) * B * 32); hipMalloc(&out, sizeof(unsigned int) * (size_t)B * pitch);
    hipMemset(flags, 0, sizeof(unsigned int) * B * 32);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ms;
    for (unsigned int it = 1; it <= 30; ++it) {
      hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_lookback, dim3(B), dim3(1024), 0, 0, rows, flags, out, T, pitch, it);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float m; hipEventElapsedTime(&m, e0, e1); ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    std::vector<unsigned int> h((size_t)B * pitch); hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    unsigned int want = 0; for (int p = 0; p < B - 1; ++p) want += (unsigned int)(p + 5) & 7u;
    printf("bins %5d blocks %d: look-back step alone %.1f us median (event spacing incl. ~4.5 us of event pair), check %s\n", T, B, ms[ms.size() / 2] * 1e3, h[(size_t)(B - 1) * pitch + 5] == want ? "ok" : "BAD");
    hipFree(rows); hipFree(flags); hipFree(out);
  }
  return 0;
}
