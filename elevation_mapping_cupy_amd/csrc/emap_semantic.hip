// RGB / semantic point-cloud fusion (BASELINE config 5: height + RGB + semantic layers).
// Reference: EM/kernels/custom_semantic_kernels.py -- sum_kernel :9-51, average_kernel :167-194, class_average_kernel
// :233-267, add_color_kernel :270-317, color_average_kernel :320-375 -- driven by EM/fusion/pointcloud_average.py:93-113,
// pointcloud_class_average.py:106-126, pointcloud_color.py:131-152 and EM/semantic_map.py:223-259.
// The reference re-reads a float-encoded cell index from the clobbered xyz columns (custom_kernels.py:260-262, exact
// only below 2^24); here the geometry is recomputed from xyz (a handful of VALU ops) and indices stay int32.
// One point pass for ALL float channels (one xyz read, K atomics), one cell pass that finalises and re-arms.
#include "emap_device.h"


template <int MODE>
__global__ __launch_bounds__(EM_BLOCK) void k_sem_sum(KP P, Pose T, SemSpec S, const float* __restrict__ pts, long n, int stride, ChanView V,
                                                       double* __restrict__ sums, long plane) {
  long i = (long)blockIdx.x * EM_BLOCK + threadIdx.x;
  if (i >= n) return;
  float rx, ry, rz;
  load_point(pts, i, stride, rx, ry, rz);
  Geo g = geometry<MODE>(P, T, rx, ry, rz);
  long c = (g.finite && g.valid && g.inside) ? owned_cell(P, g.ix, g.iy) : -1;   // valid && inside (:41-45)
  if (c < 0) return;
  const float* p = chan_row(V, i);
  for (int k = 0; k < S.n_sum; ++k) {
    const float v = p[S.sum_chan[k]];
    if (S.sum_kind[k] >= 2) {
      if (i * S.sum_K[k] + S.sum_q[k] >= n) continue;          // launch-size quirk of the compact kernels
      if (S.sum_kind[k] == 2 && !(v >= 0.0f)) continue;        // alpha_kernel: theta < 0 (or NaN) adds nothing (:36-41)
    }
    unsafeAtomicAdd(&sums[(long)S.sum_layer[k] * plane + c], (double)v);
  }
}

// add_color_kernel (:270-317).  The reference launches it with size = N while decoding id = i / K, layer = i % K
// (fusion/pointcloud_color.py:143): with K colour channels only the first N/K points contribute
// and the shared counter is incremented once per (point, layer).  Reproduced literally.
template <int MODE>
__global__ __launch_bounds__(EM_BLOCK) void k_sem_color(KP P, Pose T, SemSpec S, const float* __restrict__ pts, long n, int stride, ChanView V,
                                                         unsigned int* __restrict__ col, long plane) {
  long i = (long)blockIdx.x * EM_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int K = S.n_col;
  long id = i / K; int layer = (int)(i % K);
  float rx, ry, rz;
  load_point(pts, id, stride, rx, ry, rz);
  Geo g = geometry<MODE>(P, T, rx, ry, rz);
  long c = (g.finite && g.valid && g.inside) ? owned_cell(P, g.ix, g.iy) : -1;
  if (c < 0) return;
  unsigned int color = __float_as_uint(chan_row(V, id)[S.col_chan[layer]]);
  atomicAdd(&col[(long)(layer * 3) * plane + c], (color & 0xFF0000u) >> 16);
  atomicAdd(&col[(long)(layer * 3 + 1) * plane + c], (color & 0xFF00u) >> 8);
  atomicAdd(&col[(long)(layer * 3 + 2) * plane + c], color & 0xFFu);
  atomicAdd(&col[(long)(K * 3) * plane + c], 1u);
}

// average_kernel / class_average_kernel / color_average_kernel in one cell pass; accumulators are re-armed here
// (the reference zeroes new_map with a boolean-mask assignment and allocates a fresh colour map every frame).
__global__ __launch_bounds__(EM_BLOCK) void k_sem_finalize(KP P, SemSpec S, const unsigned int* __restrict__ cnt_plane,
                                                            double* __restrict__ sums, unsigned int* __restrict__ col,
                                                            float* __restrict__ sem, float* __restrict__ alpha_planes, long plane) {
  long li = (long)blockIdx.x * EM_BLOCK + threadIdx.x;
  if (li >= (long)P.nrows * P.C) return;
  long c = li + (long)P.halo * P.C;
  const unsigned int cnt = cnt_plane[c];          // accepted HEIGHT points of this frame (new_elmap plane 2, :185)
  float tot = 0.0f;
  for (int k = 0; k < S.n_sum; ++k) {
    const long j = (long)S.sum_layer[k] * plane + c;
    const double s = sums[j];
    if (S.sum_kind[k] == 2) {                      // class_bayesian: pseudo-counts persist (semantic_map.py:54-56), every cell
      const float a = (float)((double)alpha_planes[j] + s);
      alpha_planes[j] = a; tot += a;
    } else if (S.sum_kind[k] == 3) {               // bayesian_inference, literally (pointcloud_bayesian_inference.py:64-76):
      const int lrow_ = (int)(li / P.C);            // the prior variance layer is zeroed every frame => sigma_old = 0
      const long gcell = (long)logi_row(P, P.row0 + lrow_) * P.C + logi_col(P, (int)(li - (long)lrow_ * P.C));   // the reference's flat LOGICAL cell index
      if (cnt > 0 && gcell * S.sum_K[k] + S.sum_q[k] < (long)P.C * P.C) {
        const float cn = (float)cnt, feat_ml = (float)s / cn, sigma_old = 0.0f, sigma = 1.0f;
        sem[j] = sigma * sem[j] / (cn * sigma_old + sigma) + cn * sigma_old * feat_ml / (cn * sigma_old + sigma);
      }
    } else if (cnt > 0) {
      if (S.sum_kind[k] == 0) sem[j] = (float)(s / (double)cnt);
      else {
        const float prev = sem[j];
        sem[j] = (prev == 0.0f) ? (float)(s / (double)cnt)
                                : (float)(S.alpha * (double)prev + (1.0 - S.alpha) * s / (double)cnt);
      }
    }
    if (s != 0.0) sums[j] = 0.0;
  }
  if (S.any_bayes) {                               // theta = alpha / sum(alpha), 1 where the sum is 0 (pointcloud_class_bayesian.py:70-75)
    if (tot == 0.0f) tot = 1.0f;
    for (int k = 0; k < S.n_sum; ++k) if (S.sum_kind[k] == 2) { const long j = (long)S.sum_layer[k] * plane + c; sem[j] = alpha_planes[j] / tot; }
  }
  if (S.n_col > 0) {
    const int K = S.n_col;
    const unsigned int k = col[(long)(K * 3) * plane + c];
    if (k > 0) {
      for (int l = 0; l < K; ++l) {
        unsigned int r = col[(long)(l * 3) * plane + c] / k, g = col[(long)(l * 3 + 1) * plane + c] / k, b = col[(long)(l * 3 + 2) * plane + c] / k;
        sem[(long)S.col_layer[l] * plane + c] = __uint_as_float((r << 16) + (g << 8) + b);
      }
      for (int q = 0; q <= 3 * K; ++q) col[(long)q * plane + c] = 0u;
    }
  }
}

static inline unsigned int nblk_(long n) { return (unsigned int)((n + EM_BLOCK - 1) / EM_BLOCK); }

void launch_sem_points(hipStream_t s, const KP& P, const Pose& T, const SemSpec& S, const float* pts, long n, int stride, const ChanView& V,
                       double* sums, unsigned int* col, long plane) {
  if (n <= 0) return;
  if (S.n_sum > 0) {
    if (P.mode == 0) hipLaunchKernelGGL(k_sem_sum<0>, dim3(nblk_(n)), dim3(EM_BLOCK), 0, s, P, T, S, pts, n, stride, V, sums, plane);
    else hipLaunchKernelGGL(k_sem_sum<1>, dim3(nblk_(n)), dim3(EM_BLOCK), 0, s, P, T, S, pts, n, stride, V, sums, plane);
  }
  if (S.n_col > 0) {
    if (P.mode == 0) hipLaunchKernelGGL(k_sem_color<0>, dim3(nblk_(n)), dim3(EM_BLOCK), 0, s, P, T, S, pts, n, stride, V, col, plane);
    else hipLaunchKernelGGL(k_sem_color<1>, dim3(nblk_(n)), dim3(EM_BLOCK), 0, s, P, T, S, pts, n, stride, V, col, plane);
  }
}
void launch_sem_finalize(hipStream_t s, const KP& P, const SemSpec& S, const unsigned int* cnt_plane, double* sums, unsigned int* col,
                         float* sem, float* alpha_planes, long plane) {
  hipLaunchKernelGGL(k_sem_finalize, dim3(nblk_((long)P.nrows * P.C)), dim3(EM_BLOCK), 0, s, P, S, cnt_plane, sums, col, sem, alpha_planes, plane);
}

// ---------------------------------------------------------------------------------------------------------
// MinFilter plugin sweep (reference EM/plugins/min_filter.py:29-82): fill originally-invalid cells with the minimum
// of the already-filled values in a (2d+1)^2 window (flat-index neighbours incl. the row wrap).  The reference reads
// the buffers it writes (in-place, racy); the defined outcome here is the Jacobi one (every cell reads the previous
// sweep), which the racy kernel can produce.  One LDS-tiled launch per sweep, double buffered; a sweep that finds the
// previous sweep left no unfilled cell (host-free early exit of :114-115) degenerates into a copy.
// ---------------------------------------------------------------------------------------------------------
#define MF_R 16
#define MF_C 64
// MAX = true is the MaxFilter plugin (EM/plugins/max_filter.py:36-112): maximum instead of minimum, the cell's CURRENT mask decides
// whether it is (re)filled, and the reference itself runs that one out of place (it passes copies as inputs).
template <bool MAX>
__global__ __launch_bounds__(EM_BLOCK) void k_min_sweep(int C, int d, const float* __restrict__ orig_mask,
                                                         const float* __restrict__ val, const float* __restrict__ msk,
                                                         float* __restrict__ oval, float* __restrict__ omsk,
                                                         const unsigned int* __restrict__ prev_unfilled, unsigned int* __restrict__ unfilled) {
  extern __shared__ float lds[];
  const int W = MF_C + 2 * d, pitch = W + 1, H = MF_R + 2 * d;
  float* sval = lds;
  float* smsk = lds + (size_t)H * pitch;
  const int tile_r = blockIdx.y * MF_R, tile_c = blockIdx.x * MF_C;
  const int tc = threadIdx.x & 63, wv = threadIdx.x >> 6, col = tile_c + tc;
  const bool frozen = prev_unfilled && *prev_unfilled == 0u;     // everything was filled: later sweeps must not run
  if (!frozen) {
    for (int r = wv; r < H; r += EM_BLOCK / 64)
      for (int cc = tc; cc < W; cc += 64) {
        int lr = tile_r - d + r, cl = tile_c - d + cc;
        if (cl < 0) { cl += C; lr -= 1; } else if (cl >= C) { cl -= C; lr += 1; }
        float v = 0.f, m = 0.f;
        if (lr >= 1 && lr <= C - 2 && cl >= 1 && cl <= C - 2) { v = val[(long)lr * C + cl]; m = msk[(long)lr * C + cl]; }
        sval[r * pitch + cc] = v; smsk[r * pitch + cc] = m;
      }
    __syncthreads();
  }
  unsigned int open_cells = 0;
  if (col < C) {
    for (int k = 0; k < MF_R / 4; ++k) {
      const int tr = wv + 4 * k, row = tile_r + tr;
      if (row >= C) break;
      const long i = (long)row * C + col;
      float v = val[i], m = msk[i];
      if (!frozen && (MAX ? m : orig_mask[i]) < 0.5f) {
        float mn = MAX ? -1000000.0f : 1000000.0f;
        for (int dy = -d; dy <= d; ++dy)
          for (int dx = -d; dx <= d; ++dx) {
            const int o = (tr + d + dy) * pitch + (tc + d + dx);
            if (smsk[o] > 0.5f && (MAX ? sval[o] > mn : sval[o] < mn)) mn = sval[o];
          }
        if (MAX ? (mn > -1000000.0f + 1.0f) : (mn < 1000000.0f - 1.0f)) { v = mn; m = 0.6f; }
      }
      oval[i] = v; omsk[i] = m;
      open_cells += !(m > 0.5f);
    }
  }
  open_cells = (unsigned int)wave_sum_ll((long long)open_cells);
  if ((threadIdx.x & 63) == 0 && open_cells) atomicAdd(unfilled, open_cells);
}

void launch_min_sweep(hipStream_t s, int C, int d, const float* orig_mask, const float* val, const float* msk, float* oval, float* omsk,
                      const unsigned int* prev_unfilled, unsigned int* unfilled, bool is_max) {
  dim3 g((C + MF_C - 1) / MF_C, (C + MF_R - 1) / MF_R), b(EM_BLOCK);
  size_t lds = (size_t)2 * (MF_R + 2 * d) * (MF_C + 2 * d + 1) * sizeof(float);
  static LdsRaised raised_max, raised_min;     // window radii up to 32 need more than the default 64 KB of dynamic LDS (gfx950: 160 KB per CU)
  if (is_max) raise_lds(k_min_sweep<true>, raised_max, 158 * 1024); else raise_lds(k_min_sweep<false>, raised_min, 158 * 1024);
  if (is_max) hipLaunchKernelGGL(k_min_sweep<true>, g, b, lds, s, C, d, orig_mask, val, msk, oval, omsk, prev_unfilled, unfilled);
  else hipLaunchKernelGGL(k_min_sweep<false>, g, b, lds, s, C, d, orig_mask, val, msk, oval, omsk, prev_unfilled, unfilled);
}

// SmoothFilter plugin (EM/plugins/smooth_filter.py:56-58): scipy-style uniform_filter(size=3), i.e. a separable 3-tap mean along
// axis 0 then axis 1 with 'reflect' borders (index -1 -> 0, n -> n-1) and a float32 intermediate; one launch per 2-D pass.
__global__ __launch_bounds__(EM_BLOCK) void k_box3(int C, const float* __restrict__ in, float* __restrict__ out) {
  const long i = (long)blockIdx.x * EM_BLOCK + threadIdx.x;
  if (i >= (long)C * C) return;
  const int r = (int)(i / C), c = (int)(i % C);
  const int r0 = r > 0 ? r - 1 : 0, r2 = r < C - 1 ? r + 1 : C - 1, c0 = c > 0 ? c - 1 : 0, c2 = c < C - 1 ? c + 1 : C - 1;
  const int rows[3] = {r0, r, r2}, cols[3] = {c0, c, c2};
  double acc = 0.0;
  for (int b = 0; b < 3; ++b) {
    double col = 0.0;
    for (int a = 0; a < 3; ++a) col += (double)in[(long)rows[a] * C + cols[b]];
    acc += (double)(float)(col / 3.0);             // float32 intermediate of the axis-0 pass
  }
  out[i] = (float)(acc / 3.0);
}
void launch_box3(hipStream_t s, int C, const float* in, float* out) {
  hipLaunchKernelGGL(k_box3, dim3(nblk_((long)C * C)), dim3(EM_BLOCK), 0, s, C, in, out);
}

// ---------------------------------------------------------------------------------------------------------
// Inpainting plugin substitute.  The reference (EM/plugins/inpainting.py:53-61) quantises the elevation to 8 bits and
// calls OpenCV's cv2.inpaint(h, mask, 1, INPAINT_TELEA) -- a serial fast-marching method from an unpinned third-party
// library that is not available here and whose values no reference test pins.  DOCUMENTED SUBSTITUTE:
// the same mask / 8-bit semantics, the hole filled front-by-front from its boundary (the marching order of FMM) with
// the distance-weighted mean of the already known 8-neighbours (weights 1, 1/sqrt2), rounded to 8 bits like OpenCV's
// uchar output.  One Jacobi sweep per front layer, double buffered, device-side early exit.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EM_BLOCK) void k_inpaint_sweep(int C, const float* __restrict__ val, const float* __restrict__ msk,
                                                             float* __restrict__ oval, float* __restrict__ omsk,
                                                             const unsigned int* __restrict__ prev_unfilled, unsigned int* __restrict__ unfilled) {
  const long i = (long)blockIdx.x * EM_BLOCK + threadIdx.x;
  const bool frozen = prev_unfilled && *prev_unfilled == 0u;
  unsigned int open_cell = 0;
  if (i < (long)C * C) {
    float v = val[i], m = msk[i];
    if (!frozen && m < 0.5f) {
      const int r = (int)(i / C), c = (int)(i % C);
      float s = 0.f, w = 0.f;
      for (int dr = -1; dr <= 1; ++dr)
        for (int dc = -1; dc <= 1; ++dc) {
          if (!dr && !dc) continue;
          const int rr = r + dr, cc = c + dc;
          if (rr < 0 || rr >= C || cc < 0 || cc >= C) continue;
          const long j = (long)rr * C + cc;
          if (msk[j] > 0.5f) { const float wt = (dr && dc) ? 0.70710678f : 1.0f; s += wt * val[j]; w += wt; }
        }
      if (w > 0.f) { v = fminf(fmaxf(rintf(s / w), 0.f), 255.f); m = 1.f; }
    }
    oval[i] = v; omsk[i] = m;
    open_cell = !(m > 0.5f);
  }
  open_cell = (unsigned int)wave_sum_ll((long long)open_cell);
  if ((threadIdx.x & 63) == 0 && open_cell) atomicAdd(unfilled, open_cell);
}
void launch_inpaint_sweep(hipStream_t s, int C, const float* val, const float* msk, float* oval, float* omsk,
                          const unsigned int* prev_unfilled, unsigned int* unfilled) {
  hipLaunchKernelGGL(k_inpaint_sweep, dim3(nblk_((long)C * C)), dim3(EM_BLOCK), 0, s, C, val, msk, oval, omsk, prev_unfilled, unfilled);
}

// ---------------------------------------------------------------------------------------------------------
// pointcloud_class_max (reference EM/fusion/pointcloud_class_max.py:12-123): every class_max channel of a point carries one
// (probability, class id) pair packed into a float -- low 16 bits an IEEE half, high 16 bits the id (decode_max, :62-78).  Per
// frame the reference (1) takes the sorted set of ids seen in the cloud and in the map's id planes, (2) sums the probabilities per
// (class, cell) over the valid, inside points (sum_max_kernel, :12-47), (3) for the fusion's layers in turn stores the per-cell
// maximum over the classes and its class id, then sets the planes of ALL classes that were a maximum in SOME cell to zero (:119-121),
// (4) normalises the layers of a cell by their sum (:123-126).  The sums here are exact: a half is an integer multiple of 2^-24, so
// 64-bit integers in units of 2^-24 hold every partial sum and the result does not depend on the order of the atomics (the
// reference's float atomics round after every addition).  Host side (the set union, the whole-plane zeroing between layers):
// emap_api.hip emap_semantic_class_max.
// ---------------------------------------------------------------------------------------------------------
struct CmaxSpec { int n; int chan[8]; int layer[8]; };
__global__ __launch_bounds__(EM_BLOCK) void k_cmax_ids(ChanView V, long n, CmaxSpec S,
                                                        unsigned char* __restrict__ seen) {
  const long i = (long)blockIdx.x * EM_BLOCK + threadIdx.x;
  if (i >= n) return;
  for (int it = 0; it < S.n; ++it) seen[__float_as_uint(chan_row(V, i)[S.chan[it]]) >> 16] = 1;       // every point, valid or not (:82-84)
}
// ids stored in the map's id planes (elements_to_shift["id_max"], kept in the layers' persistent planes): values < 65536 only
__global__ __launch_bounds__(EM_BLOCK) void k_cmax_prev_ids(KP P, CmaxSpec S, const float* __restrict__ id_planes, long plane,
                                                             unsigned char* __restrict__ seen_prev) {
  const long i = (long)blockIdx.x * EM_BLOCK + threadIdx.x;
  if (i >= (long)P.nrows * P.C) return;
  const long c = (long)P.halo * P.C + i;
  for (int it = 0; it < S.n; ++it) { const unsigned int v = __float_as_uint(id_planes[(long)S.layer[it] * plane + c]); if (v < 65536u) seen_prev[v] = 1; }
}
template <int MODE>
__global__ __launch_bounds__(EM_BLOCK) void k_cmax_sum(KP P, Pose T, CmaxSpec S, const float* __restrict__ pts, long n, int stride, ChanView V,
                                                        const int* __restrict__ pos, long long* __restrict__ prob_sum, long plane) {
  const long i = (long)blockIdx.x * EM_BLOCK + threadIdx.x;
  if (i >= n) return;
  float rx, ry, rz;
  load_point(pts, i, stride, rx, ry, rz);
  const Geo g = geometry<MODE>(P, T, rx, ry, rz);
  const long c = (g.finite && g.valid && g.inside) ? owned_cell(P, g.ix, g.iy) : -1;
  if (c < 0) return;
  for (int it = 0; it < S.n; ++it) {
    const unsigned int bits = __float_as_uint(chan_row(V, i)[S.chan[it]]);
    const unsigned short hb = (unsigned short)(bits & 0xffffu);
    _Float16 h; __builtin_memcpy(&h, &hb, 2);
    const float prob = (float)h;
    if (!(fabsf(prob) <= 65504.0f)) continue;                      // inf / NaN probabilities add nothing (the reference would poison the sum)
    const long long fix = (long long)(prob * 16777216.0f);          // exact: halves are multiples of 2^-24
    atomicAdd(reinterpret_cast<unsigned long long*>(&prob_sum[(long)pos[bits >> 16] * plane + c]), (unsigned long long)fix);
  }
}
// one layer of step (3): maximum over the classes (planes already set to zero count as 0, like the reference's zeroed planes), FIRST
// class on ties (cp.argmax); the winning classes are flagged in `used` and zeroed for the next layer by k_cmax_merge
__global__ __launch_bounds__(EM_BLOCK) void k_cmax_top(KP P, int U, const long long* __restrict__ prob_sum, long plane,
                                                        const unsigned char* __restrict__ zeroed, unsigned char* __restrict__ used,
                                                        const unsigned int* __restrict__ unique_id, float* __restrict__ new_plane,
                                                        float* __restrict__ id_plane) {
  const long i = (long)blockIdx.x * EM_BLOCK + threadIdx.x;
  if (i >= (long)P.nrows * P.C) return;
  const long c = (long)P.halo * P.C + i;
  float best = 0.f; int arg = 0;
  for (int u = 0; u < U; ++u) {
    const float v = zeroed[u] ? 0.f : (float)((double)prob_sum[(long)u * plane + c] * (1.0 / 16777216.0));      // the exact sum, rounded once
    if (u == 0 || v > best) { best = v; arg = u; }
  }
  new_plane[c] = best;
  id_plane[c] = __uint_as_float(unique_id[arg]);
  used[arg] = 1;
}
__global__ void k_cmax_merge(int U, unsigned char* __restrict__ zeroed, unsigned char* __restrict__ used) {
  for (int u = threadIdx.x; u < U; u += blockDim.x) { if (used[u]) zeroed[u] = 1; used[u] = 0; }
}
__global__ __launch_bounds__(EM_BLOCK) void k_cmax_norm(KP P, CmaxSpec S, const float* __restrict__ newp, float* __restrict__ sem, long plane) {
  const long i = (long)blockIdx.x * EM_BLOCK + threadIdx.x;
  if (i >= (long)P.nrows * P.C) return;
  const long c = (long)P.halo * P.C + i;
  float sum = 0.f;
  for (int it = 0; it < S.n; ++it) sum += newp[(long)it * plane + c];
  if (sum == 0.f) sum = 1.f;
  for (int it = 0; it < S.n; ++it) sem[(long)S.layer[it] * plane + c] = newp[(long)it * plane + c] / sum;
}
void launch_cmax_ids(hipStream_t s, const KP& P, const CmaxSpec& S, const ChanView& V, long n, const float* id_planes, long plane,
                     unsigned char* seen, unsigned char* seen_prev) {
  if (n > 0) hipLaunchKernelGGL(k_cmax_ids, dim3(nblk_(n)), dim3(EM_BLOCK), 0, s, V, n, S, seen);
  hipLaunchKernelGGL(k_cmax_prev_ids, dim3(nblk_((long)P.nrows * P.C)), dim3(EM_BLOCK), 0, s, P, S, id_planes, plane, seen_prev);
}
void launch_cmax_sum(hipStream_t s, const KP& P, const Pose& T, const CmaxSpec& S, const float* pts, long n, int stride, const ChanView& V, const int* pos,
                     long long* prob_sum, long plane) {
  if (n <= 0) return;
  if (P.mode == 0) hipLaunchKernelGGL(k_cmax_sum<0>, dim3(nblk_(n)), dim3(EM_BLOCK), 0, s, P, T, S, pts, n, stride, V, pos, prob_sum, plane);
  else hipLaunchKernelGGL(k_cmax_sum<1>, dim3(nblk_(n)), dim3(EM_BLOCK), 0, s, P, T, S, pts, n, stride, V, pos, prob_sum, plane);
}
void launch_cmax_select(hipStream_t s, const KP& P, const CmaxSpec& S, int U, const long long* prob_sum, long plane, unsigned char* zeroed,
                        unsigned char* used, const unsigned int* unique_id, float* newp, float* id_planes, float* sem) {
  for (int it = 0; it < S.n; ++it) {
    hipLaunchKernelGGL(k_cmax_top, dim3(nblk_((long)P.nrows * P.C)), dim3(EM_BLOCK), 0, s, P, U, prob_sum, plane, zeroed, used, unique_id,
                       newp + (long)it * plane, id_planes + (long)S.layer[it] * plane);
    hipLaunchKernelGGL(k_cmax_merge, dim3(1), dim3(256), 0, s, U, zeroed, used);
  }
  hipLaunchKernelGGL(k_cmax_norm, dim3(nblk_((long)P.nrows * P.C)), dim3(EM_BLOCK), 0, s, P, S, newp, sem, plane);
}

// ---------------------------------------------------------------------------------------------------------
// Camera path.  image_to_map_correspondence_kernel (reference EM/kernels/custom_image_kernels.py:
// 9-157): every known cell is projected into the image (P = K [R|t], optional radtan distortion) and a Bresenham walk
// towards the camera cell rejects it when terrain in between rises above the line of sight.  Per cell, race free.
// exponential_/color_correspondences_to_map_kernel (:195-271) then sample the image per cell.
// ---------------------------------------------------------------------------------------------------------
struct CamArgs { float P[12], K[9], D[5], center[3]; float x1, y1, z1, ih, iw; double tol; };      // tol = tolerance_z_collision (:9; 0.10 in the reference's call)

__device__ __forceinline__ float l2_dist(int x0, int y0, int x1, int y1) { float dx = (float)(x0 - x1), dy = (float)(y0 - y1); return sqrtf(dx * dx + dy * dy); }

__global__ __launch_bounds__(EM_BLOCK) void k_image_corr(KP P, CamArgs A, Cells cells, float* __restrict__ uv,
                                                          unsigned char* __restrict__ valid) {
  const int W = P.C; const long L = (long)W * W;
  const long i = (long)blockIdx.x * EM_BLOCK + threadIdx.x;
  if (i >= L) return;
  float out_u = 0.f, out_v = 0.f; unsigned char out_ok = 0;
  do {
    int y0 = (int)(i % W), x0 = (int)(i / W);                              // logical cell; single-strip contexts: physical row = local row
    const Cell me = cells[(long)phys_row(P, x0) * W + phys_col(P, y0)];
    if (me.valid != 1.0f) break;                                           // only cells with is_valid == 1 (:40-42)
    float p1 = (float)((double)(x0 - (W / 2)) * P.res + (double)A.center[0]);
    float p2 = (float)((double)(y0 - (W / 2)) * P.res + (double)A.center[1]);
    const float z0 = me.h;
    float p3 = z0 + A.center[2];
    float u = p1 * A.P[0] + p2 * A.P[1] + p3 * A.P[2] + A.P[3];
    float v = p1 * A.P[4] + p2 * A.P[5] + p3 * A.P[6] + A.P[7];
    float d = p1 * A.P[8] + p2 * A.P[9] + p3 * A.P[10] + A.P[11];
    if (d <= 0.f) break;
    u = u / d; v = v / d;
    if (!(A.D[0] == 0.f && A.D[1] == 0.f && A.D[2] == 0.f && A.D[3] == 0.f && A.D[4] == 0.f)) {     // radtan (:66-86)
      const float k1 = A.D[0], k2 = A.D[1], q1 = A.D[2], q2 = A.D[3], k3 = A.D[4], fx = A.K[0], fy = A.K[4], cx = A.K[2], cy = A.K[5];
      float x = (u - cx) / fx, y = (v - cy) / fy;
      float r2 = x * x + y * y;
      float radial = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;
      float uc = x * radial + 2 * q1 * x * y + q2 * (r2 + 2 * x * x);
      float vc = y * radial + 2 * q2 * x * y + q1 * (r2 + 2 * y * y);
      u = fx * uc + cx; v = fy * vc + cy;
    }
    if ((u < 0.f) || (v < 0.f) || (u >= A.iw) || (v >= A.ih)) break;
    const int x0c = x0, y0c = y0;
    const float x1 = A.x1, y1 = A.y1;
    const float total_dis = l2_dist(x0c, y0c, (int)x1, (int)y1);
    const float delta_z = A.z1 - z0;
    const int dx = (int)fabsf(x1 - (float)x0), sx = (float)x0 < x1 ? 1 : -1, dy = -(int)fabsf(y1 - (float)y0), sy = (float)y0 < y1 ? 1 : -1;
    int error = dx + dy;
    bool ok = true;
    for (;;) {                                                               // Bresenham towards the camera cell (:103-147)
      if ((float)x0 == x1 && (float)y0 == y1) break;
      if (x0 >= 0 && y0 >= 0 && x0 < W && y0 < W) {
        const long idx = phys_col(P, y0) + (long)phys_row(P, x0) * W;
        const float4 hv = cells.hot[idx];     // h, (v), valid
        if (hv.z != 0.f) {
          float dis = l2_dist(x0c, y0c, x0, y0);
          float rayheight = z0 + (dis / total_dis * delta_z);
          if ((double)hv.x - A.tol > (double)rayheight) { ok = false; break; }
        }
      }
      const int e2 = 2 * error;
      if (e2 >= dy) { if ((float)x0 == x1) break; error += dy; x0 += sx; }
      if (e2 <= dx) { if ((float)y0 == y1) break; error += dx; y0 += sy; }
    }
    out_u = u; out_v = v; out_ok = ok ? 1 : 0;
  } while (false);
  uv[i] = out_u; uv[L + i] = out_v; valid[i] = out_ok;
}

__global__ __launch_bounds__(EM_BLOCK) void k_image_fuse(KP P, int kind, float* __restrict__ sem, const float* __restrict__ image,
                                                          const float* __restrict__ uv, const unsigned char* __restrict__ valid,
                                                          float ih, float iw, double alpha) {
  const long L = (long)P.C * P.C;
  const long i = (long)blockIdx.x * EM_BLOCK + threadIdx.x;
  if (i >= L || !valid[i]) return;
  const int idx = (int)((float)(int)uv[i] + (float)(int)uv[L + i] * iw);
  const int x0 = (int)(i / P.C), y0 = (int)(i - (long)x0 * P.C);
  const long pc = (long)phys_row(P, x0) * P.C + phys_col(P, y0);           // the layer is stored with the map's circular origin
  if (kind == 0) sem[pc] = (float)((double)sem[pc] * (1 - alpha) + alpha * (double)image[idx]);
  else if (kind == 2) sem[pc] = image[idx];                                 // average_correspondences_to_map_kernel (:160-192): the sample replaces the value
  else {
    const int ig = (int)(iw * ih + (float)idx), ib = (int)(iw * ih * 2 + (float)idx);
    const unsigned int r = (unsigned int)image[idx], g = (unsigned int)image[ig], b = (unsigned int)image[ib];
    sem[pc] = __uint_as_float((r << 16) + (g << 8) + b);
  }
}

void launch_image_corr(hipStream_t s, const KP& P, const CamArgs& A, Cells cells, float* uv, unsigned char* valid) {
  hipLaunchKernelGGL(k_image_corr, dim3(nblk_((long)P.C * P.C)), dim3(EM_BLOCK), 0, s, P, A, cells, uv, valid);
}
void launch_image_fuse(hipStream_t s, const KP& P, int kind, float* sem, const float* image, const float* uv, const unsigned char* valid,
                       float ih, float iw, double alpha) {
  hipLaunchKernelGGL(k_image_fuse, dim3(nblk_((long)P.C * P.C)), dim3(EM_BLOCK), 0, s, P, kind, sem, image, uv, valid, ih, iw, alpha);
}

// ---------------------------------------------------------------------------------------------------------
// Safety-polygon service: polygon_mask_kernel (reference EM/kernels/custom_kernels.py:509-651).  The vertex and bounding-box
// cell indices depend only on the polygon, so the host computes them once with the reference's arithmetic (float16 helper
// parameters, fp32 division; emap_api.hip: emap_polygon_mask); the per-cell test below is the reference's integer
// ray-crossing test, literally.
// ---------------------------------------------------------------------------------------------------------
struct PtI { int x, y; };
__device__ __forceinline__ bool pm_on_segment(PtI p, PtI q, PtI r) {
  return q.x <= max(p.x, r.x) && q.x >= min(p.x, r.x) && q.y <= max(p.y, r.y) && q.y >= min(p.y, r.y);
}
__device__ __forceinline__ int pm_orientation(PtI p, PtI q, PtI r) {
  int val = (q.y - p.y) * (r.x - q.x) - (q.x - p.x) * (r.y - q.y);
  if (val == 0) return 0;
  return (val > 0) ? 1 : 2;
}
__device__ __forceinline__ bool pm_intersect(PtI p1, PtI q1, PtI p2, PtI q2) {
  int o1 = pm_orientation(p1, q1, p2), o2 = pm_orientation(p1, q1, q2), o3 = pm_orientation(p2, q2, p1), o4 = pm_orientation(p2, q2, q1);
  if (o1 != o2 && o3 != o4) return true;
  if (o1 == 0 && pm_on_segment(p1, p2, q1)) return true;
  if (o2 == 0 && pm_on_segment(p1, q2, q1)) return true;
  if (o3 == 0 && pm_on_segment(p2, p1, q2)) return true;
  if (o4 == 0 && pm_on_segment(p2, q1, q2)) return true;
  return false;
}
__global__ __launch_bounds__(EM_BLOCK) void k_polygon_mask(int C, const int* __restrict__ vx, const int* __restrict__ vy, int n,
                                                            int bminx, int bminy, int bmaxx, int bmaxy, float* __restrict__ mask) {
  long i = (long)blockIdx.x * EM_BLOCK + threadIdx.x;
  if (i >= (long)C * C) return;
  PtI p = {(int)(i / C), (int)(i % C)}, extreme = {100000, p.y};
  if (p.x < bminx || p.x > bmaxx || p.y < bminy || p.y > bmaxy) { mask[i] = 0.f; return; }
  int cnt = 0;
  for (int j = 0; j < n; ++j) {
    const int j2 = (j + 1) % n;
    PtI p1 = {vx[j], vy[j]}, p2 = {vx[j2], vy[j2]};
    if (pm_intersect(p1, p2, p, extreme)) {
      if (pm_orientation(p1, p, p2) == 0) {
        if (pm_on_segment(p1, p, p2)) { mask[i] = 1.f; return; }
      } else if (((p1.y <= p.y) && (p2.y > p.y)) || ((p1.y > p.y) && (p2.y <= p.y))) cnt++;
    }
  }
  mask[i] = (cnt % 2 == 0) ? 0.f : 1.f;
}
void launch_polygon_mask(hipStream_t s, int C, const int* vx, const int* vy, int n, const int bbox[4], float* mask) {
  hipLaunchKernelGGL(k_polygon_mask, dim3(nblk_((long)C * C)), dim3(EM_BLOCK), 0, s, C, vx, vy, n, bbox[0], bbox[1], bbox[2], bbox[3], mask);
}

// ---------------------------------------------------------------------------------------------------------
// dilation_filter_kernel on caller-provided planes (reference custom_kernels.py:392-449), used by ElevationMap.initialize_map
// (elevation_mapping.py:914-921, radius dilation_size_initialize).  The reference runs it IN PLACE there (map == newmap,
// mask == newmask: a race on the GPU); this is the out-of-place (Jacobi) outcome.  Service rate: plain window search.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EM_BLOCK) void k_dilate_planes(int C, int d, const float* __restrict__ plane, const float* __restrict__ mask,
                                                             float* __restrict__ out, float* __restrict__ outmask) {
  const long L = (long)C * C, i = (long)blockIdx.x * EM_BLOCK + threadIdx.x;
  if (i >= L) return;
  float o = plane[i], om = mask[i];
  if (om < 0.5f) {
    float distance = 100.f, near_value = 0.f;
    for (int dy = -d; dy <= d; ++dy) for (int dx = -d; dx <= d; ++dx) {
      const long j = i + (long)C * dy + dx;
      if (j < 0 || j >= L) continue;
      const long jx = j / C, jy = j % C;
      if (jx <= 0 || jx >= C - 1 || jy <= 0 || jy >= C - 1) continue;
      if (mask[j] > 0.5f && (float)(dx + dy) < distance) { distance = (float)(dx + dy); near_value = plane[j]; }
    }
    if (distance < 100.f) { o = near_value; om = 1.0f; }
  }
  out[i] = o; outmask[i] = om;
}
void launch_dilate_planes(hipStream_t s, int C, int d, const float* plane, const float* mask, float* out, float* outmask) {
  hipLaunchKernelGGL(k_dilate_planes, dim3(nblk_((long)C * C)), dim3(EM_BLOCK), 0, s, C, d, plane, mask, out, outmask);
}

// ---------------------------------------------------------------------------------------------------------
// Erosion plugin (reference EM/plugins/erosion.py:96-104): the reference quantises the layer to 8 bits on the host and calls
// cv2.erode(img, ones((k, k)), iterations=n) -- OpenCV, an unpinned third-party dependency that is absent here.  Its published
// definition for a flat rectangular structuring element is the window minimum with the anchor at (k/2, k/2) and pixels outside
// the image ignored (morphologyDefaultBorderValue); one launch per iteration.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EM_BLOCK) void k_erode(int C, int k, const float* __restrict__ in, float* __restrict__ out) {
  const long i = (long)blockIdx.x * EM_BLOCK + threadIdx.x;
  if (i >= (long)C * C) return;
  const int r = (int)(i / C), c = (int)(i % C), a = k / 2;
  float mn = INFINITY;
  for (int dr = -a; dr < k - a; ++dr) {
    const int rr = r + dr;
    if (rr < 0 || rr >= C) continue;
    for (int dc = -a; dc < k - a; ++dc) {
      const int cc = c + dc;
      if (cc < 0 || cc >= C) continue;
      mn = fminf(mn, in[(long)rr * C + cc]);
    }
  }
  out[i] = mn;
}
void launch_erode(hipStream_t s, int C, int k, const float* in, float* out) {
  hipLaunchKernelGGL(k_erode, dim3(nblk_((long)C * C)), dim3(EM_BLOCK), 0, s, C, k, in, out);
}

// ---------------------------------------------------------------------------------------------------------
// The nine kernel factories of EM/kernels/custom_semantic_kernels.py as they are: raw-array elementwise kernels over `size`
// elements whose points carry (cell index, valid, inside) in their first three columns (what add_points_kernel leaves there,
// custom_kernels.py:260-262) -- the staged surface behind compat/elevation_mapping_cupy/kernels/custom_semantic_kernels.py.
// (The per-frame path fuses accumulate + finalise per tile in LDS: k_tile_semantic.)  Float accumulation uses float atomics like the
// reference (order dependent in the last bit there too).
// ---------------------------------------------------------------------------------------------------------
struct SemRaw { int op, stride, K, n_max; long size, cells; double alpha; };
enum { SR_SUM = 0, SR_SUM_COMPACT, SR_SUM_MAX, SR_ALPHA, SR_ADD_COLOR };                   // accumulate ops (:9-51, :54-86, :89-123, :126-164, :270-318)
enum { SF_AVERAGE = 0, SF_CLASS_AVERAGE, SF_BAYESIAN, SF_COLOR_AVERAGE };                  // finalise ops (:167-194, :233-267, :197-230, :320-375)
__global__ __launch_bounds__(EM_BLOCK) void k_semraw_acc(SemRaw A, const float* __restrict__ p, const int* __restrict__ pcl_chan, const int* __restrict__ map_lay,
                                                          const float* __restrict__ max_pt, const int* __restrict__ max_id, float* __restrict__ newmap,
                                                          unsigned int* __restrict__ color_map) {
  const long i = (long)blockIdx.x * EM_BLOCK + threadIdx.x;
  if (i >= A.size) return;
  if (A.op == SR_SUM_MAX) {
    const int idx = (int)p[i * A.stride];
    if ((unsigned long)idx >= (unsigned long)A.cells) return;                 // a cell index outside the planes (the reference would write out of bounds): dropped
    if (p[i * A.stride + 1] != 0.0f && p[i * A.stride + 2] != 0.0f)
      for (int it = 0; it < A.n_max; ++it) atomicAdd(&newmap[A.cells * max_id[i * A.n_max + it] + idx], max_pt[i * A.n_max + it]);
    return;
  }
  const long id = i / A.K; const int layer = (int)(i % A.K);
  const int idx = (int)p[id * A.stride];
  if ((unsigned long)idx >= (unsigned long)A.cells) return;
  if (!(p[id * A.stride + 1] != 0.0f && p[id * A.stride + 2] != 0.0f)) return;
  const float feat = p[id * A.stride + pcl_chan[layer]];
  switch (A.op) {
    case SR_SUM: atomicAdd(&newmap[A.cells * map_lay[layer] + idx], feat); break;
    case SR_SUM_COMPACT: atomicAdd(&newmap[A.cells * layer + idx], feat); break;
    case SR_ALPHA: { float theta_max = 0.f; int arg_max = 0; if (feat >= theta_max) { arg_max = map_lay[layer]; theta_max = feat; }
                     atomicAdd(&newmap[A.cells * arg_max + idx], theta_max); break; }
    default: { const unsigned int color = __float_as_uint(feat);
               atomicAdd(&color_map[A.cells * (layer * 3) + idx], (color & 0xFF0000u) >> 16);
               atomicAdd(&color_map[A.cells * (layer * 3 + 1) + idx], (color & 0xFF00u) >> 8);
               atomicAdd(&color_map[A.cells * (layer * 3 + 2) + idx], color & 0xFFu);
               atomicAdd(&color_map[A.cells * (A.K * 3) + idx], 1u); }
  }
}
__global__ __launch_bounds__(EM_BLOCK) void k_semraw_fin(SemRaw A, float* __restrict__ newmap, const unsigned int* __restrict__ color_map,
                                                          const int* __restrict__ map_lay, const float* __restrict__ new_elmap, const float* __restrict__ sum_mean,
                                                          float* __restrict__ map) {
  const long i = (long)blockIdx.x * EM_BLOCK + threadIdx.x;
  if (i >= A.size) return;
  const long id = i / A.K; const int layer = (int)(i % A.K);
  const long j = A.cells * map_lay[layer] + id;
  if (A.op == SF_COLOR_AVERAGE) {
    const unsigned int cnt = color_map[A.cells * (A.K * 3) + id];
    if (cnt > 0) {
      const unsigned int r = color_map[A.cells * (layer * 3) + id] / (1 * cnt), g = color_map[A.cells * (layer * 3 + 1) + id] / (1 * cnt),
                         b = color_map[A.cells * (layer * 3 + 2) + id] / (1 * cnt);
      map[j] = __uint_as_float((r << 16) + (g << 8) + b);
    }
    return;
  }
  const float cnt = new_elmap[A.cells * 2 + id];
  if (!(cnt > 0)) return;
  if (A.op == SF_AVERAGE) map[j] = newmap[j] / (1 * cnt);
  else if (A.op == SF_CLASS_AVERAGE) {
    const float prev = map[j];
    if (prev == 0) map[j] = newmap[j] / (1 * cnt);
    else map[j] = (float)(A.alpha * prev + (1 - A.alpha) * newmap[j] / (cnt));          // ${alpha} is a double literal in the reference's source
  } else {
    const float feat_ml = sum_mean[A.cells * layer + id] / cnt, feat_old = map[j], sigma_old = newmap[j], sigma = 1.0f;
    const float feat_new = sigma * feat_old / (cnt * sigma_old + sigma) + cnt * sigma_old * feat_ml / (cnt * sigma_old + sigma);
    const float sigma_new = sigma * sigma_old / (cnt * sigma_old + sigma);
    map[j] = feat_new; newmap[j] = sigma_new;
  }
}
void launch_semraw_acc(hipStream_t s, const SemRaw& A, const float* p, const int* pcl_chan, const int* map_lay, const float* max_pt, const int* max_id,
                       float* newmap, unsigned int* color_map) {
  if (A.size > 0) hipLaunchKernelGGL(k_semraw_acc, dim3(nblk_(A.size)), dim3(EM_BLOCK), 0, s, A, p, pcl_chan, map_lay, max_pt, max_id, newmap, color_map);
}
void launch_semraw_fin(hipStream_t s, const SemRaw& A, float* newmap, const unsigned int* color_map, const int* map_lay, const float* new_elmap,
                       const float* sum_mean, float* map) {
  if (A.size > 0) hipLaunchKernelGGL(k_semraw_fin, dim3(nblk_(A.size)), dim3(EM_BLOCK), 0, s, A, newmap, color_map, map_lay, new_elmap, sum_mean, map);
}
