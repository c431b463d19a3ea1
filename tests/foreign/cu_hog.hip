// Test infrastructure (not product): a FOREIGN kernel that holds compute units for a while, launched on a stream of its own -- what a
// semantic-segmentation network running next to the map looks like to k_small_frame's grid barriers.  Each workgroup of 1024 threads
// takes half of a CU's wave slots; `groups` of them are launched and every one spins for `ms` milliseconds of device wall clock.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(1024) void k_hog(unsigned long long ticks, unsigned int* sink) {
  const unsigned long long t0 = wall_clock64();
  unsigned int x = threadIdx.x;
  while (wall_clock64() - t0 < ticks) x = x * 1664525u + 1013904223u;
  if (x == 0x12345u) *sink = x;                 // (keeps the loop alive)
}
static hipStream_t g_stream = nullptr;
static unsigned int* g_sink = nullptr;
extern "C" int hog_start(int device, int groups, double ms) {
  if (hipSetDevice(device) != hipSuccess) return 1;
  if (!g_stream && hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 2;
  if (!g_sink && hipMalloc((void**)&g_sink, 64) != hipSuccess) return 3;
  int rate_khz = 100000;                        // wall_clock64 ticks at 100 MHz on gfx9; ask the runtime when it knows
  if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || rate_khz <= 0) rate_khz = 100000;
  const unsigned long long ticks = (unsigned long long)(ms * (double)rate_khz);
  hipLaunchKernelGGL(k_hog, dim3(groups), dim3(1024), 0, g_stream, ticks, g_sink);
  return hipGetLastError() == hipSuccess ? 0 : 4;
}
extern "C" int hog_wait(void) { return g_stream && hipStreamSynchronize(g_stream) == hipSuccess ? 0 : 1; }
