// Inpainting plugin, method "ns" (reference EM/plugins/inpainting.py:33-38,59: cv2.inpaint(h, mask, 1, cv2.INPAINT_NS)) -- HOST code,
// like the reference's (cp.asnumpy + OpenCV on the CPU).  The Navier-Stokes based method of Bertalmio, Bertozzi, Sapiro ("Navier-Stokes,
// Fluid Dynamics, and Image and Video Inpainting", CVPR 2001) in the fast-marching form OpenCV gives it: the region is filled in the
// order of the arrival time T of a front that starts at its boundary (the same march as Telea's method: narrow band, eikonal update
// from the four quadrant pairs, FIFO among equal T), and a pixel is the weighted mean of the known pixels within the radius with
//     w = 1 / (|r|^2 + 1)  x  |r . iso| / sqrt(|r| |iso|)
// where r points from the known pixel to the new one and iso = (-|dI/drow|, |dI/dcol|) is the isophote direction at the known pixel
// from one-sided / two-sided absolute differences of its known neighbours: pixels ALONG the isophote through the new pixel count,
// pixels across it do not -- the image's level lines are continued into the hole.  No gradient term is added (that is Telea's).
// OpenCV (requirements.txt: opencv-python, unpinned) is absent from this image and from /root/reference: this is a restatement of
// the published method in the form OpenCV's implementation is remembered to have, and PARITY WITH OPENCV'S VALUES IS NOT PINNED.
// tests/test_inpaint_ns.py pins it against a second, line-by-line restatement (oracle/ns_inpaint.py) and checks the properties any
// implementation must have.  Plain C++ (no device code): not part of the kernel sources the profiles are stamped with.
#include "../../include/emap_hip.h"
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

namespace {
enum : unsigned char { KNOWN = 0, BAND = 1, INSIDE = 2 };

struct Front {                                  // framed (rows + 2) x (cols + 2) flag / arrival-time planes + the queue of the march
  int R, C;
  std::vector<unsigned char> f;
  std::vector<float> t;
  std::multimap<float, std::pair<int, int>> q;  // equal keys leave in the order they came (insertion at the upper bound)
  unsigned char& F(int i, int j) { return f[(size_t)i * C + j]; }
  float& T(int i, int j) { return t[(size_t)i * C + j]; }
  void push(int i, int j, float T_) { q.emplace(T_, std::make_pair(i, j)); }
  bool pop(int& i, int& j) {
    if (q.empty()) return false;
    auto it = q.begin(); i = it->second.first; j = it->second.second; q.erase(it);
    return true;
  }
  // |grad T| = 1 from the pair of neighbours (i1, j1), (i2, j2)
  float solve(int i1, int j1, int i2, int j2) {
    const float a = T(i1, j1), b = T(i2, j2), m = a < b ? a : b;
    const bool ka = F(i1, j1) != INSIDE, kb = F(i2, j2) != INSIDE;
    if (ka && kb) return std::fabs(a - b) >= 1.0f ? 1.0f + m : (a + b + std::sqrt(2.0f - (a - b) * (a - b))) * 0.5f;
    if (ka) return 1.0f + a;
    if (kb) return 1.0f + b;
    return 1.0f + m;
  }
  float arrival(int i, int j) {
    const float a = solve(i - 1, j, i, j - 1), b = solve(i + 1, j, i, j - 1), c = solve(i - 1, j, i, j + 1), d = solve(i + 1, j, i, j + 1);
    const float ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
  }
};
}  // namespace

extern "C" int emap_inpaint_ns_u8(const uint8_t* image, const uint8_t* mask, int32_t rows, int32_t cols, int32_t radius, uint8_t* out) {
  if (!image || !mask || !out || rows < 2 || cols < 2 || (int64_t)rows * cols > (int64_t)1 << 30) return EMAP_ERR_INVALID;
  const int range = radius < 1 ? 1 : (radius > 100 ? 100 : radius);
  Front M; M.R = rows + 2; M.C = cols + 2;
  const size_t n = (size_t)M.R * M.C;
  M.f.assign(n, KNOWN); M.t.assign(n, 1.0e6f);
  std::vector<unsigned char> img(image, image + (size_t)rows * cols);
  auto O = [&](int i, int j) -> int { return (int)img[(size_t)i * cols + j]; };          // the un-framed image being filled
  for (int i = 0; i < rows; ++i) for (int j = 0; j < cols; ++j) if (mask[(size_t)i * cols + j]) M.F(i + 1, j + 1) = INSIDE;
  // narrow band: known pixels (off the frame) with a region pixel among their four neighbours; they start the march at T = 0, row-major
  for (int i = 1; i < M.R - 1; ++i) for (int j = 1; j < M.C - 1; ++j) {
    if (M.F(i, j) == INSIDE) continue;
    if (M.F(i - 1, j) == INSIDE || M.F(i + 1, j) == INSIDE || M.F(i, j - 1) == INSIDE || M.F(i, j + 1) == INSIDE) { M.T(i, j) = 0.0f; M.push(i, j, 0.0f); }
  }
  for (auto& e : M.q) M.F(e.second.first, e.second.second) = BAND;
  const int di[4] = {-1, 0, 1, 0}, dj[4] = {0, -1, 0, 1};
  int ii, jj;
  while (M.pop(ii, jj)) {
    M.F(ii, jj) = KNOWN;
    for (int q = 0; q < 4; ++q) {
      const int i = ii + di[q], j = jj + dj[q];
      if (i <= 0 || j <= 0 || i >= M.R - 1 || j >= M.C - 1 || M.F(i, j) != INSIDE) continue;
      const float dist = M.arrival(i, j);
      M.T(i, j) = dist;
      float Ia = 0.f, s = 1.0e-20f;
      for (int k = i - range; k <= i + range; ++k) {
        const int km = k - 1 + (k == 1), kp = k - 1 - (k == M.R - 2);            // image rows of the pixel / of its row neighbours, clamped at the frame
        for (int l = j - range; l <= j + range; ++l) {
          const int lm = l - 1 + (l == 1), lp = l - 1 - (l == M.C - 2);
          if (k <= 0 || l <= 0 || k >= M.R - 1 || l >= M.C - 1) continue;
          if (M.F(k, l) == INSIDE || (l - j) * (l - j) + (k - i) * (k - i) > range * range) continue;
          const float ry = (float)(i - k), rx = (float)(j - l);
          const float rlen = std::sqrt(rx * rx + ry * ry);
          const float dst = 1.0f / (rlen * rlen + 1.0f);
          float gr, gc;                                                          // |dI/drow|, |dI/dcol| at (k, l) from its known neighbours
          if (M.F(k + 1, l) != INSIDE) {
            if (M.F(k - 1, l) != INSIDE) gr = (float)(std::abs(O(kp + 1, lm) - O(kp, lm)) + std::abs(O(kp, lm) - O(km - 1, lm)));
            else gr = (float)std::abs(O(kp + 1, lm) - O(kp, lm)) * 2.0f;
          } else gr = M.F(k - 1, l) != INSIDE ? (float)std::abs(O(kp, lm) - O(km - 1, lm)) * 2.0f : 0.0f;
          if (M.F(k, l + 1) != INSIDE) {
            if (M.F(k, l - 1) != INSIDE) gc = (float)(std::abs(O(km, lp + 1) - O(km, lm)) + std::abs(O(km, lm) - O(km, lm - 1)));
            else gc = (float)std::abs(O(km, lp + 1) - O(km, lm)) * 2.0f;
          } else gc = M.F(k, l - 1) != INSIDE ? (float)std::abs(O(km, lm) - O(km, lm - 1)) * 2.0f : 0.0f;
          const float ix = -gr, iy = gc;                                         // the isophote direction (x: columns, y: rows)
          float dir = rx * ix + ry * iy;
          if (std::fabs(dir) <= 0.01f) dir = 0.000001f;
          else dir = (float)std::fabs((double)dir / std::sqrt((double)(rlen * std::sqrt(ix * ix + iy * iy))));
          const float w = dst * dir;
          Ia += w * (float)O(km, lm);
          s += w;
        }
      }
      const long r = std::lrint((double)Ia / (double)s);                         // one rounding to nearest (ties to even), then the clamp
      img[(size_t)(i - 1) * cols + (j - 1)] = (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
      M.F(i, j) = BAND;
      M.push(i, j, dist);
    }
  }
  memcpy(out, img.data(), (size_t)rows * cols);
  return EMAP_OK;
}
