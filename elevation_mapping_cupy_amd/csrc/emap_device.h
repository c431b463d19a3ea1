// Device-side data layout and arithmetic helpers of the MI355X elevation-map fusion core (gfx950 only).
//
// HBM layout (DESIGN.md "Data layout"):
//   Cell  cells[(rows+2*halo) * C]   32 B, one per map cell, cell-interleaved: the 7 planes of the reference's
//                                    planar (7,C,C) elevation_map (elevation_mapping.py:68-77) in one sector so a
//                                    point / ray step touches ONE 32-byte sector instead of 7 cache lines.
//   AccF  acc[...]                   40 B per-cell frame accumulators of the count + fuse passes (integer /
//                                    fixed-point atomics => bit-reproducible sums).
//   AccR  accr[...]                  16 B per-cell accumulators of the visibility pass.
//   float trav_in[...], normal[3][...]  planar (stencil inputs/outputs want unit stride).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EM_BLOCK 256

// A cell in REGISTERS.  In memory it is two 16-byte halves in two planes (struct Cells): the hot half (h, v, valid, trav) is what the
// point passes stage per tile and what the drift statistics read; the cold half (time, upper, is_upper, valid') is what the stencil
// pass reads -- valid' mirrors `valid` so that k_post needs ONE 16-byte load per cell.  A pass then moves only the half it needs
// (interleaved 32-byte cells cost a whole sector for either half: k_tile_count fetched 32 MB for 16, k_post 46 MB for 17 and
// wrote 4-byte `trav` fields into 32-byte sectors).
struct __attribute__((aligned(32))) Cell {
  float h, v, valid, trav, time, upper, is_upper, pad;
};
static_assert(sizeof(Cell) == 32, "cell");
struct CellRef {                       // lets `Cell m = cells[c]` / `cells[c] = m` read as before
  float4* hp; float4* cp;
  __device__ __forceinline__ operator Cell() const {
    const float4 a = *hp, b = *cp;
    Cell m; m.h = a.x; m.v = a.y; m.valid = a.z; m.trav = a.w; m.time = b.x; m.upper = b.y; m.is_upper = b.z; m.pad = b.w;
    return m;
  }
  __device__ __forceinline__ const CellRef& operator=(const Cell& m) const {
    *hp = make_float4(m.h, m.v, m.valid, m.trav);
    *cp = make_float4(m.time, m.upper, m.is_upper, m.valid);          // valid mirrored for the stencil pass
    return *this;
  }
};
struct Cells {
  float4* hot; float4* cold;
  __device__ __forceinline__ CellRef operator[](long i) const { return CellRef{hot + i, cold + i}; }
};

struct AccF {                      // zero == "nothing happened this frame"
  unsigned long long pts_inl;      // lo32: points per cell (newmap[4]); hi32: drift inliers per cell (newmap[3])
  unsigned long long cnt_out;      // lo32: accepted points (newmap[2]);  hi32: outliers (number of :174 hits)
  long long sum_h;                 // sum of new_h, Q31.32
  long long sum_v;                 // sum of new_v, Q23.40
  unsigned long long latest;       // ((point index + 1) << 32) | bits(new_h): largest index wins (= sequential :191)
};
static_assert(sizeof(AccF) == 40, "AccF");

struct AccR {
  long long dec;                   // sum of validity decrements (:251), Q23.40 (negative)
  unsigned int hits;               // number of penetrations (variance += outlier_variance each, :252)
  unsigned int upper_key;          // ~ordered(nz) maximised == min nz of qualifying visits; 0 = none
};
static_assert(sizeof(AccR) == 16, "AccR");

#define EM_ERR_SLOTS 256
struct __attribute__((aligned(128))) ErrSlot { long long sum; unsigned long long cnt; };  // Q27.36

struct FrameDev {                  // per-frame scalars that stay on the device (no D2H sync in the frame)
  long long err_sum_fix;           // Q27.36
  unsigned long long err_cnt;
  float shift, mean_error, additive_mean_error;
  int gate_fired;
  unsigned long long ray_visits;
  unsigned int n_points;
  // Which ray kernel the map's state asks for.  k_ray_apply zeroes the frame's inert bitmap (quiet = known + fresh, or border) anyway: on
  // the way it counts the set bits (one device atomic per workgroup that holds bitmap words -- 64 at 1024^2; counting in the tile
  // kernel, one atomic per tile workgroup on one line, cost that kernel 9 us: device atomics on a line serialise at ~10 ns) into
  // quiet_sum[parity of the launch]; thread 0 of the NEXT launch reads the finished sum, derives the class -- fewer than 30 % quiet:
  // a mostly unknown / stale map, where the march queues cell work at almost every step and is latency bound -- and tells the host
  // through a host-mapped word when it changes: such a map keeps the ray kernel's bitmap in global memory and runs two workgroups per
  // CU (terrain scene: 193 vs 219 us; on the uniform benchmark the LDS variant is 10 % ahead).
  unsigned int quiet_sum[2];
  unsigned int ray_class, pad_;
};

static_assert(sizeof(FrameDev) % 8 == 0, "FrameDev is copied as 64-bit words (k_small_frame)");
#define EM_SCALE_H 4294967296.0          /* 2^32 */
#define EM_SCALE_V 1099511627776.0       /* 2^40 */
#define EM_SCALE_E 68719476736.0         /* 2^36 */

// Kernel parameters: everything the reference bakes into its kernel strings as literals
// (custom_kernels.py:264-274), passed as kernargs instead (no JIT, parameters can change per launch).
// Map shifts that have not been written into the cells yet (DESIGN.md "Map shift"): move_to / move only rotate the circular origin
// and append an entry here; every kernel that reads cells before the next full rewrite replays the entries on the fly (cell_now),
// the next pass that rewrites every cell (the frame's tile kernel / commit pass, k_var_time, k_materialize) writes them out.
#define EM_MAX_MOVES 4
struct Moves { int n, pad_; int org_r[EM_MAX_MOVES], org_c[EM_MAX_MOVES], sr[EM_MAX_MOVES], sc[EM_MAX_MOVES]; float dz[EM_MAX_MOVES]; };

struct KP {
  int C, mode, row0, nrows, halo, edge, dil, idx_formula;   // idx_formula: reference_fp16 cell index by the float formula (host-proven exact, build_ray_tables)
  // k_rays only: the arrays it addresses hold `ncols` columns starting at column col0 with a row pitch of `pitch` cells -- (0, C, C) for
  // every map / strip context; the RAY WINDOW of a multi-GPU frame (rays marched by ray owner, emap_api.hip: rays_by_ray) is a small
  // rectangle of the map in logical coordinates with its own pitch (wmode = 1: normals are addressed like the cells)
  int col0, ncols, pitch, wmode;
  // row strips after a ROW shift: the normal planes handed to k_rays / k_win_pack are a row-aligned copy (emap_api.hip:
  // normal_exchange) -- row j holds the normals that belong to the cells of owned row j, columns still at the planes' own origin
  int nlag;
  int ray_pref;      // host side only (launch_rays): 1 = the map is mostly unknown / stale (FrameDev::ray_class): keep the ray kernel's bitmap in global memory
  // circular origin: logical cell (r, c) lives at physical row (r + org_r) mod C, column (c + org_c) mod C; row0 / nrows / halo
  // describe PHYSICAL rows (a strip keeps its rows when the map shifts).  norg_*: origin the stencil outputs (normal planes,
  // traversability_input) were written with -- the reference does not shift those (elevation_mapping.py:200-214).
  int org_r, org_c, norg_r, norg_c;
  Moves mv;
  double res, half_w, snf, mt, ov, dcvi_half, trav_inlier, wall, mrl, cs, cos_thresh, mvd2, mhr, ra, rb, rc;
  double max_var, ray_step;
  float init_var, ov_f, q_wm1, q_mrl, q_step, time_var, time_int, res_f, inv_res_f, half_w_f, cm1_f, hw_int_f, hw_frac_f, pad1;   // hw_int_f + hw_frac_f = cell_n / 2
};

// host-built tables of the visibility pass (emap_api.hip: build_ray_tables)
struct RayTab {
  const float* S; int nS;                   // s_k = Q(s_{k-1} + ray_step), all k with s_k < Q(max_ray_length)
  const unsigned short* lut; int lo, hi;    // reference_fp16: cell index by half bit pattern, magnitudes [lo, hi), 2 signs
  int small_pos, small_neg, big_pos, big_neg, nan_val;
  int formula_ok, pad_;                     // reference_fp16: trunc(clamp(fma(q, 1/res, C/2))) proven equal to the table for every half pattern
  float f_d_thresh, f_cos_thresh, f_wall;   // float thresholds equivalent to the reference's double comparisons
};

// Where k_rays accumulates its effects: the interleaved 16-byte AccR records of a context, or -- ray window of a multi-GPU frame --
// three planes that RCCL can reduce with typed operations (sum of int64 {dec, hits} pairs, max of uint32 keys).  Byte strides.
struct AccRView { char* dec; char* hits; char* key; int sd, sh, sk, pad_; };
__device__ __forceinline__ long long* accr_dec(const AccRView& A, unsigned int c) { return reinterpret_cast<long long*>(A.dec + (size_t)c * A.sd); }
__device__ __forceinline__ unsigned int* accr_hits(const AccRView& A, unsigned int c) { return reinterpret_cast<unsigned int*>(A.hits + (size_t)c * A.sh); }
__device__ __forceinline__ unsigned int* accr_key(const AccRView& A, unsigned int c) { return reinterpret_cast<unsigned int*>(A.key + (size_t)c * A.sk); }

// The ray window of a multi-GPU frame: logical rows [r0, r0 + nr) x logical columns [c0, c0 + nc) around the sensor (everything a ray of
// at most max_ray_length can reach), r0 / nr multiples of 8, c0 / nc multiples of 64.  One buffer of 32-bit words per rank -- hot half
// cells, cold half cells (w = "quiet" instead of valid'), three normal planes, the frame's inlier counts -- which the owners of the
// rows fill (k_win_pack) and an exact integer all-reduce (x + 0 + ... + 0) replicates; then every rank derives the bitmap and the block
// thresholds (k_win_prepare), marches the rays of ITS points over it (k_rays on a window KP), the effects are all-reduced (sum /
// max) and the owners take their rows back (k_win_unpack).
struct Win {
  int r0, c0, nr, nc;
  // what travels: ONE 32-byte record per cell {h, v, time, upper, n0, n1, n2, flags} -- flags bit 0: valid >= 0.5, 1: is_upper >= 0.5,
  // 2: quiet (snapshot S1), 3: this frame's drift inliers exceed wall_num_thresh -- everything the march and the window's bitmap /
  // thresholds read of a cell (round 5; until round 4 the 48 bytes of the two half cells, float normals and the inlier COUNT)
  unsigned int* rec;                                                 // 8 words per cell
  float4* hot; float4* cold; float* normal; unsigned int* inl;      // nr * nc elements each (normal: 3 planes), expanded from rec by k_win_prepare on every rank; inl = the flag
  unsigned long long* bits; float* thr;                              // nr * nc / 64 words (+ an all-ones word); (nr / 8) * (nc / 8) thresholds
  long long* dh; unsigned int* key;                                  // {dec, hits} pairs and keys of the window's cells
};

struct Pose {        // R, t of one frame; Rq/tq are the values after the float16 parameter rounding
  float Rq[9], tq[3], t[3];
};

template <int MODE> __device__ __forceinline__ float Qf(float x) {
  if constexpr (MODE == 0) return (float)(_Float16)x; else return x;   // v_cvt_f16_f32 (RNE) + v_cvt_f32_f16
}

__device__ __forceinline__ int sat_int(double v) {   // CUDA-style saturating conversion, NaN -> 0
  if (!(v == v)) return 0;
  v = fmin(fmax(v, -2147483648.0), 2147483647.0);
  return (int)v;
}

// get_x_idx/get_y_idx + clamp (custom_kernels.py:22-33,45-49) for an already-rounded coordinate; map centre is 0
// (the reference always passes center 0: elevation_mapping.py:337-338,359-360).
template <int MODE> __device__ __forceinline__ int axis_idx(const KP& P, float xq) {
  if constexpr (MODE == 0) {           // reference_fp16: the reference's own arithmetic (bit-exact indices)
    if (P.idx_formula) {               // (uniform) ... or the float formula the host proved equal to it for every half value: no fp64 division
      const float f = __builtin_floorf(__builtin_fmaf(xq, P.inv_res_f, P.hw_frac_f)) + P.hw_int_f;
      return (int)__builtin_amdgcn_fmed3f(f, 0.0f, P.cm1_f);
    }
    int i = sat_int((double)xq / P.res + P.half_w);
    float fi = Qf<MODE>((float)i);
    float r = fmaxf(fminf(fi, P.q_wm1), 0.0f);
    return (int)r;
  } else {
    // fp32 mode is defined by this project (the reference's float16 parameters break beyond 2049 cells): one float multiply,
    // one float add (no contraction), truncation, integer clamp -- no fp64 in the ray samples of the large maps; the oracle
    // evaluates the same two roundings.
    const float v = __fadd_rn(__fmul_rn(xq, P.inv_res_f), P.half_w_f);
    const int i = __float2int_rz(v);   // saturating, NaN -> 0
    return min(max(i, 0), P.C - 1);
  }
}

struct Geo { float x, y, z, v; int ix, iy; bool finite, valid, inside; };

// is_valid (custom_kernels.py:62-81); arguments already rounded
__device__ __forceinline__ bool is_valid_q(const KP& P, float x, float y, float z, float sx, float sy, float sz) {
  float dx = x - sx, dy = y - sy, dz = z - sz;
  float d = dx * dx + dy * dy + dz * dz;
  float dxy = (float)fmax((double)sqrtf(x * x + y * y) - P.rb, 0.0);
  if ((double)d < P.mvd2) return false;
  if ((double)dz > (double)dxy * P.ra + P.rc || (double)dz > P.mhr) return false;
  return true;
}

template <int MODE> __device__ __forceinline__ Geo geometry(const KP& P, const Pose& T, float rx, float ry, float rz) {
  Geo g;
  g.finite = !(isnan(rx) || isnan(ry) || isnan(rz));
  float qx = Qf<MODE>(rx), qy = Qf<MODE>(ry), qz = Qf<MODE>(rz);
  g.x = T.Rq[0] * qx + T.Rq[1] * qy + T.Rq[2] * qz + T.tq[0];   // transform_p :54-57 (contraction is off)
  g.y = T.Rq[3] * qx + T.Rq[4] * qy + T.Rq[5] * qz + T.tq[1];
  g.z = T.Rq[6] * qx + T.Rq[7] * qy + T.Rq[8] * qz + T.tq[2];
  double zz = (double)qz;
  g.v = (float)(P.snf * zz * zz);                                   // z_noise :58-60
  float xq = Qf<MODE>(g.x), yq = Qf<MODE>(g.y), zq = Qf<MODE>(g.z);
  g.ix = axis_idx<MODE>(P, xq);
  g.iy = axis_idx<MODE>(P, yq);
  g.valid = is_valid_q(P, xq, yq, zq, T.tq[0], T.tq[1], T.tq[2]);
  g.inside = !(g.ix == 0 || g.ix == P.C - 1 || g.iy == 0 || g.iy == P.C - 1);  // is_inside :34-44
  return g;
}

// !is_inside (custom_kernels.py:34-44): first / last row or column of the map (LOGICAL coordinates)
__device__ __forceinline__ bool border_cell(const KP& P, int r, int c) { return r <= 0 || r >= P.C - 1 || c <= 0 || c >= P.C - 1; }

// ---- circular origin: logical <-> physical ------------------------------------------------------------------------------
__device__ __forceinline__ int wrap_up(int v, int C) { return v >= C ? v - C : v; }       // v in [0, 2C)
__device__ __forceinline__ int wrap_dn(int v, int C) { return v < 0 ? v + C : v; }        // v in [-C, C)
__device__ __forceinline__ int phys_row(const KP& P, int r) { return wrap_up(r + P.org_r, P.C); }     // r in [0, C)
__device__ __forceinline__ int phys_col(const KP& P, int c) { return wrap_up(c + P.org_c, P.C); }
__device__ __forceinline__ int logi_row(const KP& P, int pr) { return wrap_dn(pr - P.org_r, P.C); }
__device__ __forceinline__ int logi_col(const KP& P, int pc) { return wrap_dn(pc - P.org_c, P.C); }
// row of the local arrays (halo + owned rows + halo) that holds physical row pr, or -1 if the strip does not hold it
__device__ __forceinline__ int local_row(const KP& P, int pr) {
  int rel = pr - P.row0;                                   // owned: [0, nrows); halos continue circularly on both sides
  if (rel < 0) rel += P.C;
  if (rel < P.nrows + P.halo) return P.halo + rel;
  if (rel >= P.C - P.halo) return P.halo - (P.C - rel);
  return -1;
}
// local cell index of LOGICAL (ix, iy), or -1 if this strip does not own the row
__device__ __forceinline__ long owned_cell(const KP& P, int ix, int iy) {
  int rel = phys_row(P, ix) - P.row0;
  if (rel < 0 || rel >= P.nrows) return -1;
  return (long)(rel + P.halo) * P.C + phys_col(P, iy);
}

// Index of the normal of LOGICAL cell (lix, liy) in the normal planes, or -1 if this context does not hold it.  The planes keep the
// origin they were written with (the reference does not shift normal_map, elevation_mapping.py:200-214): logical -> THEIR rows and
// columns.  lrow = the cell's owned row (physical row - row0), only used for the row-aligned copy of a strip (KP::nlag).
__device__ __forceinline__ long normal_index(const KP& P, int lrow, int lix, int liy) {
  if (P.nlag) return (long)lrow * P.C + wrap_up(liy + P.norg_c, P.C);
  const int nlr = local_row(P, wrap_up(lix + P.norg_r, P.C));
  return nlr < 0 ? -1 : (long)nlr * P.C + wrap_up(liy + P.norg_c, P.C);
}

// row of the inert bitmap for physical row prow: the logical row on single-strip contexts, else the local (physical) row
__device__ __forceinline__ int bitmap_row(const KP& P, int prow) { return P.nrows == P.C ? logi_row(P, prow) : prow - P.row0; }

struct Owned { long c; int prow, pcol; };
__device__ __forceinline__ Owned owned(const KP& P, int ix, int iy) {
  Owned o; o.prow = phys_row(P, ix); o.pcol = phys_col(P, iy);
  const int rel = o.prow - P.row0;
  o.c = (rel < 0 || rel >= P.nrows) ? -1 : (long)(rel + P.halo) * P.C + o.pcol;
  return o;
}

// A cell as it is NOW: the pending map shifts replayed on the stored value (shift_map_xy: cells of the entering band are reset,
// variance to initial_variance; shift_map_z: planes 0 and 5 += dz -- elevation_mapping.py:172-226), in the order they happened.
// (prow, pcol): PHYSICAL row / column of the cell.  Free when nothing is pending.
__device__ __forceinline__ void cell_now(const KP& P, float& h, float& v, float& valid, float& trav, float& time, float& upper,
                                         float& is_upper, int prow, int pcol) {
  for (int m = 0; m < P.mv.n; ++m) {
    const int r = wrap_dn(prow - P.mv.org_r[m], P.C), c = wrap_dn(pcol - P.mv.org_c[m], P.C);     // logical position after move m
    const int sr = P.mv.sr[m], sc = P.mv.sc[m];
    if ((sr > 0 && r < sr) || (sr < 0 && r >= P.C + sr) || (sc > 0 && c < sc) || (sc < 0 && c >= P.C + sc)) {
      h = 0.f; v = P.init_var; valid = 0.f; trav = 0.f; time = 0.f; upper = 0.f; is_upper = 0.f;
    }
    h += P.mv.dz[m]; upper += P.mv.dz[m];
  }
}

__device__ __forceinline__ unsigned int float_ord(float f) {       // monotone map float -> uint
  unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_float(unsigned int o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__device__ __forceinline__ void cell_now(const KP& P, Cell& m, int prow, int pcol) {
  if (P.mv.n) cell_now(P, m.h, m.v, m.valid, m.trav, m.time, m.upper, m.is_upper, prow, pcol);
}
// (h, v, valid, trav) part only (drift statistics, fusion)
__device__ __forceinline__ void cell_now(const KP& P, float4& m, int prow, int pcol) {
  if (P.mv.n) { float time = 0.f, upper = 0.f, is_upper = 0.f; cell_now(P, m.x, m.y, m.z, m.w, time, upper, is_upper, prow, pcol); }
}

// effects of custom_kernels.py:174 and :189-192 on one cell (-> snapshot S1)
__device__ __forceinline__ void commit_cell(const KP& P, Cell& c, const AccF& a) {
  unsigned int cnt = (unsigned int)(a.cnt_out & 0xffffffffull), n_out = (unsigned int)(a.cnt_out >> 32);
  if (n_out) c.v = c.v + P.ov_f * (float)n_out;
  if (cnt) { c.valid = 1.0f; c.time = 0.0f; c.upper = __uint_as_float((unsigned int)(a.latest & 0xffffffffull)); c.is_upper = 0.0f; }
}
// average_map_kernel (custom_kernels.py:365-389) on one committed cell
__device__ __forceinline__ void average_cell(const KP& P, Cell& m, const AccF& a) {
  const float valid0 = m.valid;
  const unsigned int cnt = (unsigned int)(a.cnt_out & 0xffffffffull);
  if (cnt > 0) {
    float nh = (float)(((double)a.sum_h / EM_SCALE_H) / (double)cnt);
    float nv = (float)(((double)a.sum_v / EM_SCALE_V) / (double)cnt);
    if ((double)nv > P.max_var) { m.h = 0.f; m.v = P.init_var; m.valid = 0.f; }
    else { m.h = nh; m.v = nv; m.valid = 1.f; }
  }
  if (valid0 < 0.5f) { m.h = 0.f; m.v = P.init_var; m.valid = 0.f; }
}

struct __attribute__((packed, aligned(4))) P3 { float x, y, z; };
__device__ __forceinline__ void load_point(const float* __restrict__ pts, long i, int stride, float& x, float& y, float& z) {
  if (stride == 3) { P3 p = reinterpret_cast<const P3*>(pts)[i]; x = p.x; y = p.y; z = p.z; }
  else { const float* p = pts + i * (long)stride; x = p[0]; y = p[1]; z = p[2]; }
}

// Where the EXTRA channels of the bound cloud live (RGB, semantic features: columns >= 3 of the caller's (N, 3 + K) matrix).  An
// interleaved device cloud: the same rows as xyz (p = the cloud, stride = 3 + K, col0 = 0).  A cloud uploaded through
// emap_upload_points (or bound with emap_set_points_device_split) is DE-INTERLEAVED: xyz as an (N, 3) matrix -- the point passes of
// a frame then stream 12 bytes per point instead of 12 + 4 K -- and the channels as an (N, K) matrix of their own (p = that matrix,
// stride = K, col0 = 3): with K = 4 every point's channels are ONE aligned 16-byte record.
struct ChanView { const float* p; int stride, col0; };
__device__ __forceinline__ const float* chan_row(const ChanView& V, long i) { return V.p + i * (long)V.stride - V.col0; }      // row[c] = channel column c of point i

// 64-lane sum (DPP-free portable form; executed once per wave and only when an inlier exists)
__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- "last workgroup" ticket --------------------------------------------------------------------------------------------
// One thread per workgroup calls this after its results are acknowledged; true for exactly one caller, the last.  Device-scope
// atomics on ONE address serialise at ~10 ns each on MI355X (1024 workgroups on a single counter: +10 us, measured), so the
// tickets form a two-level tree: sqrt(n) groups, each counter on its own 128-byte line.  Counters re-arm themselves.
// Memory model note: the data a workgroup publishes before it takes its ticket goes out as a RELAXED agent-scope atomic store followed
// by s_waitcnt (the store has been acknowledged by the memory side), and the winner reads it back with relaxed agent-scope atomic
// loads.  There is no release / acquire pair: an agent-scope release makes every workgroup write back its XCD's L2 (measured: scan
// 10 -> 30 us).  This ordering is a property of gfx950's device-coherent (sc1) accesses, not of the HIP memory model -- hence the
// guard below -- and it is pinned by tests/test_hip_large_maps.py::test_sort_offsets_under_stress (16384 sort bins, many frames,
// binned vs atomic scatter bit for bit).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "last_block_ticket / k_bin_scan order their hand-off with gfx950 device-coherent stores + s_waitcnt; re-derive it for another target"
#endif
#define EM_TICKET_WORDS (1025 * 32)
__device__ __forceinline__ bool last_block_ticket(unsigned int* __restrict__ sync, unsigned int bid, unsigned int nblocks) {
  unsigned int gs = 1; while (gs * gs < nblocks) gs <<= 1;
  const unsigned int g = bid / gs, ng = (nblocks + gs - 1) / gs, members = min(gs, nblocks - g * gs);
  unsigned int* c1 = sync + 32 * (1 + g);
  if (__hip_atomic_fetch_add(c1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != members - 1) return false;
  __hip_atomic_store(c1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (__hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ng - 1) return false;
  __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// ---- grid-wide barrier of the single-launch kernel (k_small_frame) --------------------------------------------------------------
// The host bounds the grid by a quarter of what the device holds (small_frame_grid), so every workgroup CAN be resident; whether it
// IS depends on what else runs on the device.  A barrier is the two-level ticket above + one 64-bit release word the waiting
// workgroups poll with device-coherent loads.  What a phase publishes for the next one goes out as device-scope atomics /
// device-coherent (sc1) stores BEFORE the arrival (s_waitcnt: acknowledged) and is read back with device-coherent loads after the
// release -- the hand-off of last_block_ticket, see the note there.
// Outcome of a barrier (round 6): ONE decision for the whole grid, taken by a compare-and-swap on the release word.  Its low half holds
// the epoch of the last launch that passed (distinct per launch, 1 .. 2^31 - 1), the high half that launch's payload (barrier 1: the
// drift shift).  The last workgroup to arrive swaps {old -> epoch | payload}: released.  A waiter whose patience runs out (SF spin
// limit, ~0.1 s: a foreign grid is holding the CUs the rest of this grid needs) swaps {old -> epoch | SF_ABORT}: aborted.  Whoever
// loses the swap reads the winner's word.  Either EVERY workgroup passes or EVERY workgroup -- those that arrive later included --
// takes the abort path: k_small_frame then puts its accumulators back and leaves the map as the launch found it; the host re-runs the frame on the chain of launches at its next call (emap_api.hip: sf_recover).  Nothing hangs, nothing is
// left undefined.
// What such a barrier costs on MI355X (measured, round 5): it is a chain of four to five DEPENDENT trips to the memory side
// (acknowledgements, group ticket, root ticket, release store, poll), 1.5-2 us each, i.e. 6-8 us -- more than the ~4.5 us a launch
// boundary occupies the stream, of which only ~1 us is not hidden behind the previous kernel's tail.  A barrier therefore only pays
// where it saves work: k_small_frame (the cloud read and transformed once instead of twice) 27.1 -> 25.3 us per frame; a sort
// front-end built the same way (histogram, scan and scatter in one launch, geometry kept in LDS: k_bin_sort) took 30 us against the
// 35 us event spacing of its three launches and left the 1024^2 / 1 M-point frame at 71.0 vs 70.3 us -- removed.  Variants measured on
// k_small_frame: a release word per ticket group instead of one (25.4 us: the polls are not what costs); one ever-growing counter that
// every workgroup adds to WITHOUT waiting and then polls, the gate evaluated redundantly by every workgroup (29.2 us: 196 atomics and
// 196 pollers on ONE address serialise at ~10 ns each, the late arrivals queue behind the early ones' polls).
#define SF_ABORT 0x80000000u
#define SF_SPIN_DEFAULT (1u << 15)      /* polls of ~4 us each under contention (measured: 2000 polls = 8.5 ms): ~0.13 s */
// every thread of the grid calls this; true in all threads of the LAST workgroup to arrive (which then calls sf_decide_last)
__device__ __forceinline__ bool sf_arrive(unsigned int* sync, bool* s_last) {
  __builtin_amdgcn_s_waitcnt(0);                               // this wave's atomics / device-coherent stores are acknowledged
  __syncthreads();
  if (threadIdx.x == 0) *s_last = last_block_ticket(sync, blockIdx.x, gridDim.x);
  __syncthreads();
  return *s_last;
}
// ONE thread of the last workgroup to arrive: release the grid with `payload`; returns the word that stands (low half epoch: released,
// epoch | SF_ABORT: a waiter gave up first)
__device__ __forceinline__ unsigned long long sf_decide_last(unsigned long long* rel, unsigned int epoch, unsigned int payload) {
  unsigned long long old = __hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long want = (unsigned long long)epoch | ((unsigned long long)payload << 32);
  if ((unsigned int)old == (epoch | SF_ABORT)) return old;
  if (__hip_atomic_compare_exchange_strong(rel, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return want;
  return old;                                                  // (the only other writer of this launch: a waiter's abort)
}
// ONE thread of every other workgroup: wait for the decision; `impatient`: give up at once (test hook).  err_host: host-mapped word
// that learns of the abort (code: the launch's epoch).
__device__ __forceinline__ unsigned long long sf_wait_decision(unsigned long long* rel, unsigned int epoch, unsigned int spin_limit, bool impatient,
                                                               unsigned int* err_host, unsigned int code) {
  unsigned int it = 0u;
  for (;;) {
    unsigned long long w = __hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (((unsigned int)w & ~SF_ABORT) == epoch) return w;      // released or aborted
    if (impatient || ++it > spin_limit) {
      const unsigned long long ab = (unsigned long long)(epoch | SF_ABORT);
      if (__hip_atomic_compare_exchange_strong(rel, &w, ab, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(err_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return ab;
      }
      return w;                                                // lost the swap: w is the last arriver's release (the only other writer)
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// ---- drift gate (elevation_mapping.py:346-352) --------------------------------------------------------------------------
// One wave: sums the error slots (integer => order independent), decides the shift, keeps additive_mean_error, re-arms the slots.
// (Folding it into the last workgroup of k_tile_count was measured: no gain -- a small dependent launch costs ~1 us in the frame.)
struct GateArgs {
  int enable, noise_ok, use_override, pad0_;
  double min_cnt, max_drift, sum_override;
  float alpha; unsigned int cnt_override, n_points, pad_;
};
__device__ __forceinline__ void gate_eval(const GateArgs& A, ErrSlot* __restrict__ slots, FrameDev* __restrict__ F, int lane, int reduce_only,
                                          double* __restrict__ dev_out, const double* __restrict__ dev_totals) {
  long long s = 0; unsigned long long k = 0;
  for (int j = lane; j < EM_ERR_SLOTS; j += 64) {
    s += __hip_atomic_load(&slots[j].sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    k += __hip_atomic_load(&slots[j].cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    slots[j].sum = 0; slots[j].cnt = 0;
  }
  s = wave_sum_ll(s); k = (unsigned long long)wave_sum_ll((long long)k);
  if (lane != 0) return;
  if (reduce_only) {
    F->err_sum_fix = s; F->err_cnt = k; F->n_points = A.n_points; F->ray_visits = 0;
    if (dev_out) { dev_out[0] = (double)s / EM_SCALE_E; dev_out[1] = (double)k; }
    return;
  }
  if (!A.use_override && !dev_totals) { F->err_sum_fix = s; F->err_cnt = k; }
  F->n_points = A.n_points; F->ray_visits = 0;
  double sum = dev_totals ? dev_totals[0] : (A.use_override ? A.sum_override : (double)s / EM_SCALE_E);
  float cnt = dev_totals ? (float)dev_totals[1] : (A.use_override ? (float)A.cnt_override : (float)k);
  float shift = 0.0f; int fired = 0;
  if (A.enable && (double)cnt > A.min_cnt && A.noise_ok) {
    float mean = (float)sum / cnt;
    fired = 1;
    F->mean_error = mean;
    F->additive_mean_error = F->additive_mean_error + mean;
    if ((double)fabsf(mean) < A.max_drift) shift = mean * A.alpha;
  }
  F->shift = shift; F->gate_fired = fired;
}

// The shift gate_eval decides, computed from the slots WITHOUT touching them or the frame record (k_small_frame: the last workgroup to
// arrive releases the others with the shift before it does the gate's bookkeeping).  Integer slot sums: the same value gate_eval derives
// afterwards.  Whole frames only (no overrides, no all-reduced totals).  One wave; every lane returns the shift.
__device__ __forceinline__ float gate_shift_only(const GateArgs& A, const ErrSlot* __restrict__ slots, int lane) {
  long long s = 0; unsigned long long k = 0;
  for (int j = lane; j < EM_ERR_SLOTS; j += 64) {
    s += __hip_atomic_load(&slots[j].sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    k += __hip_atomic_load(&slots[j].cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  s = wave_sum_ll(s); k = (unsigned long long)wave_sum_ll((long long)k);
  const double sum = (double)s / EM_SCALE_E;
  const float cnt = (float)k;
  float shift = 0.0f;
  if (A.enable && (double)cnt > A.min_cnt && A.noise_ok) {
    const float mean = (float)sum / cnt;
    if ((double)fabsf(mean) < A.max_drift) shift = mean * A.alpha;
  }
  return shift;
}

// Multi-GPU frames: the gate decision on the ALL-REDUCED drift totals rides in the head of the tile kernel (every workgroup reads
// the two doubles and decides for itself -- the same value everywhere; workgroup (0, 0) also keeps the frame record): one dependent
// launch less between the all-reduce and the fusion.  mode 0: not folded (the shift comes from FrameDev, written by k_gate).
// (Single-GPU frames keep k_gate: folding the reduction of the 256 error slots into every tile workgroup was measured in round 3 --
// k_gate's 4.6 us disappeared, k_tile_fuse grew by 3.0 us, the frame did not move.)
struct GateFold { int mode, pad_; const double* dev_totals; GateArgs A; };
__device__ __forceinline__ float gate_fold(const GateFold& g, FrameDev* __restrict__ F, bool keeper) {
  const double sum = g.dev_totals[0];
  const float cnt = (float)g.dev_totals[1];
  float shift = 0.0f, mean = 0.0f; int fired = 0;
  if (g.A.enable && (double)cnt > g.A.min_cnt && g.A.noise_ok) {
    mean = (float)sum / cnt;
    fired = 1;
    if ((double)fabsf(mean) < g.A.max_drift) shift = mean * g.A.alpha;
  }
  if (keeper) {
    F->n_points = g.A.n_points; F->ray_visits = 0;
    if (fired) { F->mean_error = mean; F->additive_mean_error = F->additive_mean_error + mean; }
    F->shift = shift; F->gate_fired = fired;
  }
  return shift;
}

// ---- clear_overlap_map (elevation_mapping.py:393-410) as an epilogue of the kernels that rewrite the cells anyway ----------
struct OverlapArgs { int on, cmin, cmax, pad_; float hmin, hmax; };
__device__ __forceinline__ bool overlap_window(const OverlapArgs& O, int lr, int lc) {      // logical row / column
  return O.on && lr >= O.cmin && lr < O.cmax && lc >= O.cmin && lc < O.cmax;
}
__device__ __forceinline__ bool overlap_cell(const KP& P, const OverlapArgs& O, Cell& m) {
  bool ch = false;
  if (m.h < O.hmin || m.h > O.hmax) { m.h = 0.f; m.v = P.init_var; m.valid = 0.f; ch = true; }
  if (m.upper < O.hmin || m.upper > O.hmax) { m.upper = 0.f; m.is_upper = 0.f; ch = true; }
  return ch;
}

// ---- tile-binned scatter (emap_binned.hip) --------------------------------------------------------------------------------
// a bin = `sub` stacked 16 x 64 tiles (sub > 1 only for maps beyond 16384 tiles); TB = T + 1 sort bins: the last one collects the
// valid points that fall OUTSIDE the owned cells when a visibility pass follows (raybin) -- they are not fused, but their rays are
// marched (k_rays walks the sorted records).  B blocks of `chunk` points; pitch = row pitch of the (block, tile) matrix in words
// (a multiple of 4).
struct BinGeo { int tiles_x, tiles_y, T, B; long chunk; int sub, TB, pitch, raybin; };
struct __attribute__((aligned(16))) BinRec { unsigned int lc_inl; float z, v; unsigned int i; };  // sorted by tile
struct BinStg;                                                                                     // staging record of the strip variants
// Records of a frame that CARRIES its semantic channels (round 6): the sorted record is 32 bytes -- the 16 above + the point's four
// carried channel columns -- so the semantic fusion reads a point's channels at the record's own position instead of gathering a
// 128-byte line per 16-byte channel row by point index (8192^2 / 16 M points: 2.0 GB fetched for 0.26 GB), and the scatter pass
// writes ONE full 32-byte sector per point (a 16-byte record already cost a whole sector on the way out; a SECOND stream of 16-byte
// stores was measured at 3.5x the pass in round 5).  Tile kernels address records as 16-byte units with a stride RS of 1 or 2.
struct __attribute__((aligned(32))) BinRec32 { unsigned int lc_inl; float z, v; unsigned int i; float c[4]; };
static_assert(sizeof(BinRec32) == 32, "BinRec32");
// what a carrying frame's scatter pass copies: columns [c0, c0 + 4) of the caller's (N, ncols) matrix (missing columns read as 0)
struct SemCarry { int on, c0, ncols, pad_; };
// the semantic fusion of a carrying frame inside the tile kernel (k_tile_fuse<.., SEM>): averaged channels (kind 0 average, 1
// class_average) and at most one colour channel, each addressed by its slot 0..3 in the record
struct SemMini { int n_sum, n_col; int slot[4], layer[4], kind[4]; int col_slot, col_layer; double alpha; float* sem; long plane; };

// ---- HEAVY tiles: the records of one tile reduced by SEVERAL workgroups --------------------------------------------------------
// One workgroup per tile is the right grain for a uniform cloud (1 M points / 1024 tiles: one trip of 1024 threads per tile).  A
// sensor delivers the opposite: the ground next to it is sampled hundreds of times per cell, and ONE tile of a scan-ordered,
// ray-cast cloud (tests/_fixtures.py: terrain_cloud) holds 300 k of its 1 M points -- that workgroup then issues ~80 instructions
// per record on one CU's four SIMDs while 255 CUs idle (k_tile_count 58 us, k_tile_fuse 148 us against 14 / 16 on the uniform
// cloud).  So a tile with more than SPLIT_CAP records is cut into parts of ~SPLIT_CAP consecutive records; part 0 runs in the
// tile's own workgroup, parts 1.. in EXTRA workgroups in front of the grid (k_bin_scan's tail lists them: the first place that
// knows the tile totals; emap_binned.hip: tile_work for how many the host launches).  Everything the parts accumulate is an integer sum or an ordered maximum, so the parts of a tile
// merge through device atomics into the tile's SLOT of a small scratch array, in any order, bit for bit:
//   k_tile_count  every part adds its error sums to the slots as before, and its per-cell point / drift-inlier counts (newmap[4],
//                 newmap[3] -- k_tile_fuse's first pass) to the slot;
//   k_tile_fuse   every part reads the whole tile's counts from the slot, runs the Kalman pass over ITS records in LDS, adds its
//                 partial sums to the slot and takes a ticket; the part that arrives last reads the totals back, clears the slot
//                 and the ticket, and runs the tile's epilogue (commit + average + bitmap + thresholds) as if it had seen every
//                 record.  Nobody waits for anybody.
// Tiles of <= SPLIT_CAP records (every tile of a uniform cloud) take the old path; what they pay is one word of k_bin_scan's
// tail per tile and one scalar load per workgroup.  Needs k_tile_count in the frame (the
// drift gate's statistics: on unless the caller rules the gate out) -- otherwise nothing is split.
#ifndef SPLIT_CAP
#define SPLIT_CAP 4096u          /* records per part: 4 trips of the 1024 threads.  Terrain scene, k_tile_count / k_tile_fuse: 28 / 39 us with 16384, 24 / 30 with 8192, 20 / 26 with 4096 (one workgroup per tile: 58 / 148) */
#endif
#ifndef SPLIT_MAX_PARTS
#define SPLIT_MAX_PARTS 128u      /* < 256: the part index is 8 bits of a list entry */
#endif
#define SPLIT_MAX_SLOTS 1024u    /* tiles (x stacked tiles of a bin) split in one frame */
#define SPLIT_MAX_EXTRA 4096u
#define SPLIT_NONE 0xffffffffu
#define SPLIT_CELLS 1024u        /* cells of a tile */
struct SplitView {
  int on;                                  // this frame's scan listed the heavy tiles
  unsigned int* tile_slot;                 // [T] first slot of the bin's tiles, SPLIT_NONE: not split
  unsigned int* extra;                     // [SPLIT_MAX_EXTRA] (bin << 8) | part of the extra workgroups' work
  unsigned int* n_extra;
  int cap;                                 // extra workgroups (x sub) in front of this frame's tile grids = parts the scan may list; a multiple of 8
  unsigned int* need_host;                 // host-mapped: the parts this frame would have listed with unlimited room
  unsigned int* tick;                      // [SPLIT_MAX_SLOTS] arrivals of k_tile_fuse's parts (zero between frames)
  unsigned int* pts; unsigned int* inl;    // [slot][cell] whole-tile counts, written by k_tile_count's parts     } all zero between
  unsigned long long* h; unsigned long long* v; unsigned long long* latest; unsigned int* cnt; unsigned int* out;   // } frames
};
__host__ __device__ __forceinline__ unsigned int split_parts(unsigned int n) {
  const unsigned int np = (n + SPLIT_CAP - 1u) / SPLIT_CAP;
  return np > SPLIT_MAX_PARTS ? SPLIT_MAX_PARTS : (np < 1u ? 1u : np);
}

// ---- gfx950 LDS-DMA: 64 lanes x 16 bytes from global memory straight into LDS (global_load_lds_dwordx4) -------------------------
// lane i's 16 bytes land at lds_base + 16 i (lds_base must be wave-uniform); the copy is ordered for readers by the vmcnt(0) the
// compiler places in front of the next __syncthreads().  For PURE copies of 16-byte records: no VGPR round trip, no ds_write.
typedef __attribute__((address_space(3))) void em_lds_void;
typedef __attribute__((address_space(1))) const void em_glb_void;
__device__ __forceinline__ void lds_dma16(const void* gsrc_lane, void* lds_base_uniform) {
  __builtin_amdgcn_global_load_lds((em_glb_void*)gsrc_lane, (em_lds_void*)lds_base_uniform, 16, 0, 0);
}
__device__ __forceinline__ void lds_dma16_at(const void* gsrc_lane, unsigned int lds_byte_offset_uniform) {      // destination given as an LDS byte offset
  __builtin_amdgcn_global_load_lds((em_glb_void*)gsrc_lane, (em_lds_void*)(size_t)lds_byte_offset_uniform, 16, 0, 0);
}

// ---- host: raising a kernel's dynamic LDS limit beyond the default 64 KB ---------------------------------------------------
// hipFuncSetAttribute applies to the CURRENT DEVICE, so the "already raised" flag is kept per (kernel instantiation, device) --
// a process may hold contexts on several devices (thread-mode strips) -- and only set once the call has succeeded.  Racing threads
// at worst repeat the (idempotent) call.
#define EM_MAX_DEV 64
struct LdsRaised { bool dev[EM_MAX_DEV]; };
template <class K> static inline bool raise_lds(K kern, LdsRaised& r, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= EM_MAX_DEV) dev = 0;
  if (r.dev[dev]) return true;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  r.dev[dev] = true;
  return true;
}

// Per-frame description of the RGB / semantic point fusion.  The leading members mirror emap_sem_spec (include/emap_hip.h);
// sum_K / sum_q are derived by emap_semantic_update: kinds 2 (class_bayesian) and 3 (bayesian_inference) reproduce the
// reference launch decode id = i / K, layer = i % K with size N (fusion/pointcloud_class_bayesian.py:28-29,67;
// fusion/pointcloud_bayesian_inference.py:28-29,111), i.e. element (point id, q-th channel of that fusion) exists only
// while id * K + q < N.  Kinds 0 / 1 carry K = 1, q = 0.
#define SEM_MAX_CH 16
struct SemSpec {
  int n_sum; int sum_chan[SEM_MAX_CH]; int sum_layer[SEM_MAX_CH]; int sum_kind[SEM_MAX_CH];   // kind 0 average, 1 class_average, 2 class_bayesian, 3 bayesian_inference
  int n_col; int col_chan[4]; int col_layer[4];
  double alpha;                                                                                // Parameter.average_weight
  int sum_K[SEM_MAX_CH]; int sum_q[SEM_MAX_CH]; int any_bayes; int pad0;
};
