// Telea's fast-marching inpainting (A. Telea, "An Image Inpainting Technique Based on the Fast Marching Method", J. Graphics
// Tools 9(1), 2004) for the Inpainting plugin -- HOST code, like the reference's: EM/plugins/inpainting.py:53-61 brings the
// elevation to the host (cp.asnumpy), quantises it to 8 bits and calls cv2.inpaint(h, mask, 1, cv2.INPAINT_TELEA), a serial
// priority-queue algorithm that OpenCV runs on the CPU.  OpenCV (requirements.txt: opencv-python, version not pinned) is absent
// from this image and its source is not part of the reference tree, so this is a restatement of the PUBLISHED algorithm in the form
// OpenCV's implementation is documented to have -- 1-pixel frame around the image, flags KNOWN / BAND / INSIDE, the narrow band =
// 4-neighbourhood dilation of the mask minus the mask, a FIFO-stable priority queue on the arrival time T, the eikonal update from
// the four quadrant pairs, the weighted first-order estimate with weights direction x distance x level over the known pixels within
// the radius, the distance field marched outwards (negative T) over the ring of known pixels within the radius -- and parity with
// OpenCV's values is NOT pinned (no golden vector exists offline).  tests/test_inpaint_telea.py pins it against a line-by-line
// Python restatement (oracle/telea.py) and checks what any correct implementation must do (known pixels untouched, constant and
// linear images reproduced).
#include "../../include/emap_hip.h"
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

namespace {
enum : unsigned char { KNOWN = 0, BAND = 1, INSIDE = 2, CHANGE = 3 };

struct Img {                      // (rows + 2) x (cols + 2) planes with a 1-pixel frame
  int R, C;
  std::vector<unsigned char> f, out;
  std::vector<float> t;
  unsigned char& F(int i, int j) { return f[(size_t)i * C + j]; }
  float& T(int i, int j) { return t[(size_t)i * C + j]; }
};

// priority queue on T, first-in first-out among equal keys (std::multimap inserts at the upper bound of an equal range)
struct Heap {
  std::multimap<float, std::pair<int, int>> q;
  void push(int i, int j, float T) { q.emplace(T, std::make_pair(i, j)); }
  bool pop(int& i, int& j) {
    if (q.empty()) return false;
    auto it = q.begin(); i = it->second.first; j = it->second.second; q.erase(it);
    return true;
  }
};

inline float min4(float a, float b, float c, float d) { a = a < b ? a : b; c = c < d ? c : d; return a < c ? a : c; }

// eikonal update |grad T| = 1 from the pair (i1, j1), (i2, j2) (Telea 2004, fig. 6 "solve")
float solve(Img& I, const std::vector<unsigned char>& f, int i1, int j1, int i2, int j2) {
  const float a11 = I.T(i1, j1), a22 = I.T(i2, j2), m12 = a11 < a22 ? a11 : a22;
  const bool k1 = f[(size_t)i1 * I.C + j1] != INSIDE, k2 = f[(size_t)i2 * I.C + j2] != INSIDE;
  if (k1) {
    if (k2) return std::fabs(a11 - a22) >= 1.0f ? 1.0f + m12 : (a11 + a22 + std::sqrt(2.0f - (a11 - a22) * (a11 - a22))) * 0.5f;
    return 1.0f + a11;
  }
  if (k2) return 1.0f + a22;
  return 1.0f + m12;
}

// distance field over the region flagged INSIDE in `f`, marched from the pixels on the heap
void calc_fmm(Img& I, std::vector<unsigned char>& f, Heap& H, bool negate) {
  int ii, jj;
  while (H.pop(ii, jj)) {
    f[(size_t)ii * I.C + jj] = negate ? CHANGE : KNOWN;
    const int di[4] = {-1, 0, 1, 0}, dj[4] = {0, -1, 0, 1};
    for (int q = 0; q < 4; ++q) {
      const int i = ii + di[q], j = jj + dj[q];
      if (i <= 0 || j <= 0 || i >= I.R - 1 || j >= I.C - 1) continue;
      if (f[(size_t)i * I.C + j] == INSIDE) {
        const float dist = min4(solve(I, f, i - 1, j, i, j - 1), solve(I, f, i + 1, j, i, j - 1), solve(I, f, i - 1, j, i, j + 1), solve(I, f, i + 1, j, i, j + 1));
        I.T(i, j) = dist;
        f[(size_t)i * I.C + j] = BAND;
        H.push(i, j, dist);
      }
    }
  }
  if (negate)
    for (size_t k = 0; k < f.size(); ++k)
      if (f[k] == CHANGE) { f[k] = KNOWN; I.t[k] = -I.t[k]; }
}

void telea(Img& I, int range, Heap& H) {
  int ii, jj;
  const int di[4] = {-1, 0, 1, 0}, dj[4] = {0, -1, 0, 1};
  auto O = [&](int i, int j) -> float { return (float)I.out[(size_t)i * (I.C - 2) + j]; };      // the un-framed output image
  while (H.pop(ii, jj)) {
    I.F(ii, jj) = KNOWN;
    for (int q = 0; q < 4; ++q) {
      const int i = ii + di[q], j = jj + dj[q];
      if (i <= 0 || j <= 0 || i >= I.R - 1 || j >= I.C - 1) continue;
      if (I.F(i, j) != INSIDE) continue;
      const float dist = min4(solve(I, I.f, i - 1, j, i, j - 1), solve(I, I.f, i + 1, j, i, j - 1), solve(I, I.f, i - 1, j, i, j + 1), solve(I, I.f, i + 1, j, i, j + 1));
      I.T(i, j) = dist;
      // gradient of the arrival time = direction of the front's normal
      float gx, gy;
      if (I.F(i, j + 1) != INSIDE) gx = I.F(i, j - 1) != INSIDE ? (I.T(i, j + 1) - I.T(i, j - 1)) * 0.5f : I.T(i, j + 1) - I.T(i, j);
      else gx = I.F(i, j - 1) != INSIDE ? I.T(i, j) - I.T(i, j - 1) : 0.0f;
      if (I.F(i + 1, j) != INSIDE) gy = I.F(i - 1, j) != INSIDE ? (I.T(i + 1, j) - I.T(i - 1, j)) * 0.5f : I.T(i + 1, j) - I.T(i, j);
      else gy = I.F(i - 1, j) != INSIDE ? I.T(i, j) - I.T(i - 1, j) : 0.0f;
      float Ia = 0.f, Jx = 0.f, Jy = 0.f, s = 1.0e-20f;
      for (int k = i - range; k <= i + range; ++k) {
        const int km = k - 1 + (k == 1), kp = k - 1 - (k == I.R - 2);           // image rows of k - 1 / k + 1 neighbours, clamped at the frame
        for (int l = j - range; l <= j + range; ++l) {
          const int lm = l - 1 + (l == 1), lp = l - 1 - (l == I.C - 2);
          if (k <= 0 || l <= 0 || k >= I.R - 1 || l >= I.C - 1) continue;
          if (I.F(k, l) == INSIDE || (l - j) * (l - j) + (k - i) * (k - i) > range * range) continue;
          const float ry = (float)(i - k), rx = (float)(j - l), len2 = rx * rx + ry * ry;
          const float dst = 1.0f / (len2 * std::sqrt(len2));                   // distance factor 1 / |r|^3 ... (the pixel itself is INSIDE: len2 > 0)
          const float lev = 1.0f / (1.0f + std::fabs(I.T(k, l) - I.T(i, j)));  // level-set factor
          float dir = rx * gx + ry * gy;                                       // direction factor
          if (std::fabs(dir) <= 0.01f) dir = 0.000001f;
          const float w = std::fabs(dst * lev * dir);
          float gIx, gIy;                                                      // image gradient at (k, l) from known neighbours
          if (I.F(k, l + 1) != INSIDE) gIx = I.F(k, l - 1) != INSIDE ? (O(km, lp + 1) - O(km, lm - 1)) * 2.0f : O(km, lp + 1) - O(km, lm);
          else gIx = I.F(k, l - 1) != INSIDE ? O(km, lp) - O(km, lm - 1) : 0.0f;
          if (I.F(k + 1, l) != INSIDE) gIy = I.F(k - 1, l) != INSIDE ? (O(kp + 1, lm) - O(km - 1, lm)) * 2.0f : O(kp + 1, lm) - O(km, lm);
          else gIy = I.F(k - 1, l) != INSIDE ? O(kp, lm) - O(km - 1, lm) : 0.0f;
          Ia += w * O(km, lm);
          Jx -= w * gIx * rx;
          Jy -= w * gIy * ry;
          s += w;
        }
      }
      // Telea eq. (3) with the normalised gradient term; stored with ONE rounding to nearest (ties to even) and a clamp.  (OpenCV's
      // source is remembered to add 0.5 in front of its rounding cast, which would turn a constant image of odd value v into v + 1;
      // its values cannot be pinned here, so the unbiased form is used: at most one of 255 levels apart.)
      const float sat = Ia / s + (Jx + Jy) / (std::sqrt(Jx * Jx + Jy * Jy) + 1.0e-20f);
      const long r = std::lrintf(sat);
      I.out[(size_t)(i - 1) * (I.C - 2) + (j - 1)] = (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
      I.F(i, j) = BAND;
      H.push(i, j, dist);
    }
  }
}
}  // namespace

extern "C" int emap_inpaint_telea_u8(const uint8_t* image, const uint8_t* mask, int32_t rows, int32_t cols, int32_t radius, uint8_t* out) {
  if (!image || !mask || !out || rows < 2 || cols < 2 || (int64_t)rows * cols > (int64_t)1 << 30) return EMAP_ERR_INVALID;      // (one row / one column: the clamped neighbours of the gradient term would leave the image)
  int range = radius < 1 ? 1 : (radius > 100 ? 100 : radius);
  Img I; I.R = rows + 2; I.C = cols + 2;
  const size_t n = (size_t)I.R * I.C;
  I.f.assign(n, KNOWN); I.t.assign(n, 1.0e6f);
  I.out.assign(image, image + (size_t)rows * cols);
  std::vector<unsigned char> m(n, 0), band(n, 0);
  for (int i = 0; i < rows; ++i) for (int j = 0; j < cols; ++j) if (mask[(size_t)i * cols + j]) m[(size_t)(i + 1) * I.C + j + 1] = INSIDE;
  auto dilate_cross = [&](const std::vector<unsigned char>& src, std::vector<unsigned char>& dst, int r) {   // (2 r + 1) cross: the city-block ball of radius r
    dst.assign(n, 0);
    for (int i = 0; i < I.R; ++i) for (int j = 0; j < I.C; ++j) {
      unsigned char v = 0;
      for (int k = -r; k <= r && !v; ++k) { const int a = i + k; if (a >= 0 && a < I.R && src[(size_t)a * I.C + j]) v = INSIDE; }
      for (int k = -r; k <= r && !v; ++k) { const int b = j + k; if (b >= 0 && b < I.C && src[(size_t)i * I.C + b]) v = INSIDE; }
      dst[(size_t)i * I.C + j] = v;
    }
  };
  auto clear_frame = [&](std::vector<unsigned char>& a) {
    for (int j = 0; j < I.C; ++j) { a[j] = 0; a[(size_t)(I.R - 1) * I.C + j] = 0; }
    for (int i = 0; i < I.R; ++i) { a[(size_t)i * I.C] = 0; a[(size_t)i * I.C + I.C - 1] = 0; }
  };
  dilate_cross(m, band, 1);
  for (size_t k = 0; k < n; ++k) band[k] = (band[k] && !m[k]) ? 1 : 0;           // narrow band = dilation minus the region
  clear_frame(band);
  Heap H;
  for (int i = 0; i < I.R; ++i) for (int j = 0; j < I.C; ++j) if (band[(size_t)i * I.C + j]) H.push(i, j, 0.0f);     // row-major, T = 0
  for (size_t k = 0; k < n; ++k) { if (band[k]) { I.f[k] = BAND; I.t[k] = 0.0f; } if (m[k]) I.f[k] = INSIDE; }
  {   // distances OUTSIDE the region (negative T) over the ring of known pixels within the radius
    std::vector<unsigned char> ring;
    dilate_cross(m, ring, range);
    for (size_t k = 0; k < n; ++k) ring[k] = (ring[k] && !m[k] && !band[k]) ? INSIDE : 0;
    clear_frame(ring);
    Heap Out;
    for (int i = 0; i < I.R; ++i) for (int j = 0; j < I.C; ++j) if (band[(size_t)i * I.C + j]) Out.push(i, j, 0.0f);
    calc_fmm(I, ring, Out, true);
  }
  telea(I, range, H);
  memcpy(out, I.out.data(), (size_t)rows * cols);
  return EMAP_OK;
}
