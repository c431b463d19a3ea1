#!/bin/bash
# AddressSanitizer + UBSan sweep of the two HOST fills of the Inpainting plugin (emap_inpaint_telea_u8, emap_inpaint_ns_u8: plain C++
# behind the C ABI, no device code): random images, mask families, radii and degenerate arguments.  Runs on the CPU.
#   tools/asan_inpaint.sh            # images of >= 2 x 2 pixels (must be clean)
#   MIN_SIDE=1 tools/asan_inpaint.sh # also one-row / one-column images (emap_inpaint_telea_u8: known out-of-bounds read)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cat > $T/fuzz.cpp <<CPP
#include "$R/include/emap_hip.h"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
int main() {
  const int min_side = getenv("MIN_SIDE") ? atoi(getenv("MIN_SIDE")) : 2;
  std::mt19937 rng(12345);
  long calls = 0, rejected = 0, touched = 0;
  for (int it = 0; it < 4000; ++it) {
    int rows = min_side + rng() % 40, cols = min_side + rng() % 40;
    if (it % 50 == 0) { rows = min_side + rng() % 3; cols = min_side + rng() % 3; }
    if (it % 97 == 0) { rows = 64 + rng() % 64; cols = 64 + rng() % 64; }
    const int radius = (it % 11 == 0) ? (int)(rng() % 9) - 1 : 1 + (int)(rng() % 3);
    std::vector<uint8_t> img(rows * cols), mask(rows * cols), o1(rows * cols, 0xAB), o2(rows * cols, 0xCD);
    const int mode = rng() % 5;
    for (int i = 0; i < rows * cols; ++i) {
      const int r = i / cols, c = i % cols;
      img[i] = rng() & 255;
      mask[i] = mode == 0 ? (rng() % 4 == 0) : mode == 1 ? 1 : mode == 2 ? 0 : mode == 3 ? (r < 2 || c < 2 || r >= rows - 2 || c >= cols - 2)
                : ((r > rows / 4 && r < 3 * rows / 4 && c > cols / 4 && c < 3 * cols / 4) ? 255 : 0);
    }
    const int rc1 = emap_inpaint_telea_u8(img.data(), mask.data(), rows, cols, radius, o1.data());
    const int rc2 = emap_inpaint_ns_u8(img.data(), mask.data(), rows, cols, radius, o2.data());
    calls += 2; rejected += (rc1 != 0) + (rc2 != 0);
    for (int i = 0; i < rows * cols; ++i) if (!mask[i]) touched += (rc1 == 0 && o1[i] != img[i]) + (rc2 == 0 && o2[i] != img[i]);      // known pixels stay
  }
  uint8_t a[16] = {1, 2, 3, 4}, m[16] = {0, 1, 0, 0}, o[16];
  const int r[] = {emap_inpaint_telea_u8(nullptr, m, 2, 2, 1, o), emap_inpaint_telea_u8(a, nullptr, 2, 2, 1, o), emap_inpaint_telea_u8(a, m, 2, 2, 1, nullptr),
                   emap_inpaint_telea_u8(a, m, 0, 2, 1, o), emap_inpaint_telea_u8(a, m, 2, -1, 1, o), emap_inpaint_ns_u8(nullptr, m, 2, 2, 1, o),
                   emap_inpaint_ns_u8(a, m, 1, 4, 1, o), emap_inpaint_ns_u8(a, m, 4, 1, 1, o), emap_inpaint_ns_u8(a, m, 2, 2, 0, o), emap_inpaint_telea_u8(a, m, 2, 2, 0, o)};
  for (int x : r) printf("%d ", x);
  printf("\ncalls %ld, rejected %ld, known pixels changed %ld\n", calls, rejected, touched);
  return touched != 0;
}
CPP
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -x c++ $R/elevation_mapping_cupy_amd/csrc/emap_inpaint_host.hip -x c++ $R/elevation_mapping_cupy_amd/csrc/emap_inpaint_ns.cpp $T/fuzz.cpp -o $T/fuzz
$T/fuzz
rm -rf $T
