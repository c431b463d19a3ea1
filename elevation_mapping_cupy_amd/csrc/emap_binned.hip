// Tile-binned scatter: the count + fuse passes WITHOUT per-point global atomics.
//
// Why: on MI355X a device-scope atomic costs ~44 us per 1 M operations whatever its width, scope, record layout or
// spatial coherence (tools/microbench.hip), and the straightforward count + fuse passes need up to 5 of them per point.
// An LDS atomic is ~100x cheaper, but LDS is per workgroup -- so the points are first counting-sorted by map TILE
// (16 rows x 64 columns = 1024 cells, the same tile the stencil kernels use) and each tile is then reduced by ONE
// workgroup in LDS:
//   k_bin_hist     per chunk of points: geometry (fp16-quirk transform, validity, cell), per-block LDS histogram over tiles; the
//                  block's histogram goes out as ONE contiguous row of the (block, tile) matrix
//   k_bin_scan     exclusive scan of every matrix COLUMN (64 columns per workgroup, staged through LDS in 256-byte row segments)
//                  -> every block's write cursor per tile; the last workgroup then scans the tile totals -> tile start offsets
//   k_bin_scatter  per chunk: geometry AGAIN from the 12-byte point (cheaper than carrying a 16-byte staging record per point
//                  through HBM: the round trip was 32 of the sort's 92 MB in round 2), LDS cursor -> 16-byte record at its
//                  sorted position (no map access)
//   k_tile_count   per tile: cells staged in LDS, drift-inlier test of every record (error_counting_kernel,
//                  custom_kernels.py:317-335), wave-reduced error sums
//   k_tile_fuse    per tile: pass 1 counts points/inliers per cell in LDS (newmap[4], newmap[3]); pass 2 is the Kalman
//                  update of custom_kernels.py:160-197 accumulated with LDS atomics (64-bit fixed point, ordered max);
//                  the epilogue commits + averages the cells (or writes the 40-byte AccF records on the staged path).
// The order of the records INSIDE a tile is not defined (LDS cursors) and nothing depends on it: every accumulation downstream is
// an integer / fixed-point sum or an ordered maximum.  The AccF contents are BIT-IDENTICAL to the atomic path of emap_kernels.hip,
// which stays as the path for clouds below ~130 k points (two launches with atomics have the lower latency there).  The LDS
// histogram holds at most 16384 bins: maps with more tiles (> 4096^2 cells per context) sort into bins of 2, 4, ... vertically
// stacked tiles and the tile kernels reduce one tile of the bin per workgroup (blockIdx.y), re-reading the bin's records from L2.
//
// Row strips (multi-GPU, the cloud is replicated): a strip owns 1/G of the rows, so 1/G of a uniform cloud.  k_bin_hist<.., STRIP>
// first runs a CHEAP ownership test on every point (the x row of the transform + one axis index: ~25 instructions instead of the
// ~120 of the full geometry with its square root and fp64 compares) and pushes the survivors on a per-WAVE LDS queue; whenever 64
// are waiting the wave works them off with all lanes busy (full geometry, histogram atomic) and appends their 16-byte staging
// records {bin | cell, z, noise, index} to the block's region of a staging array (1 KB coalesced bursts).  No block-wide barrier
// inside the loop: the waves of a workgroup hide each other's memory latency.  k_bin_scatter<.., STRIP> then reads only the staged
// records (N / G of them) and is a pure permutation.  So the full geometry, the atomics and the record traffic shrink with G; only
// ONE 12-byte stream over the cloud per frame does not.  With a visibility pass every VALID point marches a ray through the strip
// (custom_kernels.py:199-258), validity needs the full geometry, and the non-strip kernels run (same records, ray-only bin).
#include "emap_device.h"
#include <cstring>
#include <cstdlib>

#define BIN_TR 16
#define BIN_TC 64
#define BIN_MAX_T 16384   /* LDS histogram / cursor arrays are dynamic: 4 B per tile */
#define BIN_QCAP 128      /* entries of a wave's compaction queue (strip variants): < 64 pending + <= 64 pushed per step */

// (BinGeo, BinRec: emap_device.h)

// sort bin of a point (-1: none) and its cell inside the bin.  Tiles are PHYSICAL: 16 owned rows x 64 columns of memory.
__device__ __forceinline__ int bin_of(const KP& P, const BinGeo& G, const Geo& g, unsigned int& lc) {
  const int lrow = phys_row(P, g.ix) - P.row0, pcol = phys_col(P, g.iy);
  lc = 0u;
  if (!(g.finite && g.valid)) return -1;
  if (g.inside && lrow >= 0 && lrow < P.nrows) {
    const int BR = BIN_TR * G.sub;
    lc = (unsigned int)((lrow % BR) * BIN_TC + (pcol % BIN_TC));
    return (lrow / BR) * G.tiles_x + (pcol / BIN_TC);
  }
  return G.raybin ? G.T : -1;             // ray only (custom_kernels.py:199-258 marches every valid point)
}

// the cheap ownership test of the strip variants: only the x row of transform_p (custom_kernels.py:54-57) and its axis index
template <int MODE> __device__ __forceinline__ bool row_owned(const KP& P, const Pose& T, float rx, float ry, float rz) {
  const float qx = Qf<MODE>(rx), qy = Qf<MODE>(ry), qz = Qf<MODE>(rz);
  const float x = T.Rq[0] * qx + T.Rq[1] * qy + T.Rq[2] * qz + T.tq[0];          // same expression as geometry(): same bits
  const int ix = axis_idx<MODE>(P, Qf<MODE>(x));
  return (unsigned int)(phys_row(P, ix) - P.row0) < (unsigned int)P.nrows;        // (NaN coordinates index like geometry(): dropped there)
}

// staging record of the strip variants: what k_bin_scatter needs to place a point, computed once by k_bin_hist
struct __attribute__((aligned(16))) BinStg { unsigned int key; float z, v; unsigned int i; };      // key = (bin << 16) | cell in bin

// STRIP: the context owns a row strip and no visibility pass follows (see the header).  The staged records of block b are
// stg[b * chunk ...], their number stg_cnt[b].
#ifndef HIST_U
#define HIST_U 4
#endif
template <int MODE, int BLK, bool STRIP>
__global__ __launch_bounds__(BLK) void k_bin_hist(KP P, Pose T, BinGeo G, const float* __restrict__ pts, long n, int stride,
                                                  unsigned int* __restrict__ hist, BinStg* __restrict__ stg,
                                                  unsigned int* __restrict__ stg_cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned int h[];
  for (int t = threadIdx.x; t < G.TB; t += BLK) h[t] = 0u;
  const long base = (long)blockIdx.x * G.chunk;
  if (!STRIP) {
    __syncthreads();
    constexpr int U = HIST_U;                                  // loads in flight per thread (as in k_bin_scatter)
    for (long k0 = threadIdx.x; k0 < G.chunk; k0 += (long)U * BLK) {
      float rx[U], ry[U], rz[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long k = k0 + (long)u * BLK, i = base + k;
        rx[u] = ry[u] = rz[u] = NAN;                           // (a NaN row: no bin)
        if (k < G.chunk && i < n) load_point(pts, i, stride, rx[u], ry[u], rz[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        unsigned int lc;
        const int bin = bin_of(P, G, geometry<MODE>(P, T, rx[u], ry[u], rz[u]), lc);
        if (bin >= 0) atomicAdd(&h[bin], 1u);
      }
    }
  } else {
    float4* q = reinterpret_cast<float4*>(h + G.pitch) + (threadIdx.x >> 6) * BIN_QCAP;      // this wave's queue: (x, y, z, index)
    unsigned int* blk_cnt = h + G.pitch + (BLK / 64) * BIN_QCAP * 4;
    if (threadIdx.x == 0) *blk_cnt = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    BinStg* out = stg + base;
    int qn = 0;                                                    // wave-uniform number of queued points
    auto work = [&](int first, int count) {                        // queue entries [first, first + count), count <= 64: the whole wave
      bool keep = false;
      BinStg r; r.key = 0u; r.z = 0.f; r.v = 0.f; r.i = 0u;
      if (lane < count) {
        const float4 e = q[first + lane];
        const Geo g = geometry<MODE>(P, T, e.x, e.y, e.z);
        unsigned int lc;
        const int bin = bin_of(P, G, g, lc);
        if (bin >= 0) { atomicAdd(&h[bin], 1u); keep = true; r.key = ((unsigned int)bin << 16) | lc; r.z = g.z; r.v = g.v; r.i = __float_as_uint(e.w); }
      }
      const unsigned long long m = __ballot(keep);
      if (!m) return;
      unsigned int b0 = 0u;
      if (lane == 0) b0 = atomicAdd(blk_cnt, (unsigned int)__popcll(m));
      b0 = (unsigned int)__shfl((int)b0, 0, 64);
      if (keep) out[b0 + __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u))] = r;
    };
    auto push = [&](bool own, float x, float y, float z, long i) {
      const unsigned long long m = __ballot(own);
      if (!m) return;                                              // wave-uniform
      if (own) q[qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u))] = make_float4(x, y, z, __uint_as_float((unsigned int)i));
      qn += __popcll(m);
      __builtin_amdgcn_wave_barrier();
      if (qn >= 64) { qn -= 64; work(qn, 64); __builtin_amdgcn_wave_barrier(); }
    };
    for (long k0 = threadIdx.x; k0 < G.chunk; k0 += 4 * BLK) {     // uniform trip count (chunk is a multiple of 4 * BLK); four loads in flight
      float x[4], y[4], z[4]; bool in[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = base + k0 + (long)u * BLK;
        in[u] = i < n; x[u] = y[u] = z[u] = 0.f;
        if (in[u]) load_point(pts, i, stride, x[u], y[u], z[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) push(in[u] && row_owned<MODE>(P, T, x[u], y[u], z[u]), x[u], y[u], z[u], base + k0 + (long)u * BLK);
    }
    if (qn > 0) work(0, qn);
    __syncthreads();
    if (threadIdx.x == 0) stg_cnt[blockIdx.x] = *blk_cnt;
  }
  __syncthreads();
  unsigned int* row = hist + (long)blockIdx.x * G.pitch;
  for (int t = threadIdx.x; t < G.pitch; t += BLK) row[t] = t < G.TB ? h[t] : 0u;
}

// Column scan of the (block, tile) matrix hist[B][pitch]: workgroup g owns the 64 columns [64 g, 64 g + 64) -- one lane per column --
// and its 16 waves share the rows: in a slab of 256 rows wave w takes 16 consecutive rows, requests all 16 of its 256-byte row
// segments at once (the kernel is one memory round trip per slab, not bandwidth), sums them per column, exchanges the 16 partial
// sums through LDS and writes its rows back as exclusive prefixes: hist[b][t] becomes the number of points of tile t in blocks < b,
// tile_total[t] the column sum.  The LAST workgroup to finish (ticket counter `sync`, re-armed for the next frame) then scans the
// tile totals: tile_start[0..TB] -- one launch instead of two.
#define SCAN_TT 64
#define SCAN_BLK 1024
#define SCAN_RPW 16      /* rows per wave and slab */
__global__ __launch_bounds__(SCAN_BLK) void k_bin_scan(BinGeo G, unsigned int* __restrict__ hist, unsigned int* __restrict__ tile_total,
                                                        unsigned int* __restrict__ tile_start, unsigned int* __restrict__ sync, SplitView SV) {
  constexpr int NW = SCAN_BLK / 64;
  __shared__ unsigned int part[NW][SCAN_TT];
  __shared__ unsigned int s_nslot, s_nextra;                // heavy tiles of this frame (SplitView, emap_device.h)
  __shared__ bool s_last;
  if (threadIdx.x == 0) { s_nslot = 0u; s_nextra = 0u; }   // (published by the barriers of the slab loop / the ticket hand-off below)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, t = blockIdx.x * SCAN_TT + lane;
  const bool col_ok = t < G.pitch;
  unsigned int carry = 0u;                                 // per column (lane), identical in all waves
  for (int sl = 0; sl < G.B; sl += NW * SCAN_RPW) {
    const int r0 = sl + w * SCAN_RPW;
    unsigned int v[SCAN_RPW];
#pragma unroll
    for (int j = 0; j < SCAN_RPW; ++j) v[j] = (col_ok && r0 + j < G.B) ? hist[(long)(r0 + j) * G.pitch + t] : 0u;
    unsigned int sum = 0u;
#pragma unroll
    for (int j = 0; j < SCAN_RPW; ++j) sum += v[j];
    part[w][lane] = sum;
    __syncthreads();
    unsigned int run = carry, tot = 0u;
#pragma unroll
    for (int k = 0; k < NW; ++k) { const unsigned int x = part[k][lane]; if (k < w) run += x; tot += x; }
#pragma unroll
    for (int j = 0; j < SCAN_RPW; ++j) { if (col_ok && r0 + j < G.B) hist[(long)(r0 + j) * G.pitch + t] = run; run += v[j]; }
    carry += tot;
    __syncthreads();
  }
  // the totals go out as device-coherent stores and are read back with device-coherent loads; only their ORDER against the
  // ticket matters (wait for the stores' acknowledgement).  An agent-scope fence would write back / invalidate the XCD's whole L2
  // per workgroup (measured: 10 -> 30 us for this kernel).
  if (w == 0) {
    if (t < G.TB) __hip_atomic_store(&tile_total[t], carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) s_last = last_block_ticket(sync, blockIdx.x, gridDim.x);
  }
  __syncthreads();
  if (!s_last) return;
  // the tail: exclusive scan of the TB tile totals by all 1024 threads -- every thread a run of consecutive tiles (two passes over
  // its run around ONE block-wide scan of the run sums; the 64 rounds of a 256-thread scan cost 30 us at 16385 bins)
  const int per = (G.TB + SCAN_BLK - 1) / SCAN_BLK, t_lo = min(G.TB, (int)threadIdx.x * per), t_hi = min(G.TB, t_lo + per);
  unsigned int mine = 0u;
  for (int tt = t_lo; tt < t_hi; ++tt) mine += __hip_atomic_load(&tile_total[tt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned int inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned int y = __shfl_up(inc, o, 64); if (lane >= o) inc += y; }
  __syncthreads();                                            // part[][] is free again (every wave has left the slab loop)
  if (lane == 63) part[0][w] = inc;
  __syncthreads();
  unsigned int base = 0u, total = 0u;
#pragma unroll
  for (int k = 0; k < NW; ++k) { const unsigned int x = part[0][k]; if (k < w) base += x; total += x; }
  unsigned int run = base + inc - mine;
  for (int tt = t_lo; tt < t_hi; ++tt) {
    const unsigned int n_t = __hip_atomic_load(&tile_total[tt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tile_start[tt] = run; run += n_t;
    if (SV.on && tt < G.T) {                                  // heavy tile: a slot per stacked tile of the bin, an extra workgroup per further part
      if (n_t > SPLIT_CAP) {                                  // (tile_slot is read for such tiles only)
        // the tile WANTS split_parts(n_t) parts; it GETS as many as the extra workgroups of this frame's launches leave room for (cap:
        // sized by the last finished frame's need, never below the standing pool -- emap_api.hip) -- at least two, or it is reduced
        // by its own workgroup alone.  The number of parts travels in the high half of the tile's slot word.
        const unsigned int want = split_parts(n_t) - 1u;
        const unsigned int s0 = atomicAdd(&s_nslot, (unsigned int)G.sub), e0 = atomicAdd(&s_nextra, want);
        const unsigned int room = e0 < (unsigned int)SV.cap ? (unsigned int)SV.cap - e0 : 0u, take = min(want, room);
        const bool fits = s0 + (unsigned int)G.sub <= SPLIT_MAX_SLOTS && take > 0u;
        for (unsigned int q = 1u; q <= take; ++q) SV.extra[e0 + q - 1u] = fits ? (((unsigned int)tt << 8) | q) : SPLIT_NONE;
        SV.tile_slot[tt] = fits ? (s0 | ((take + 1u) << 16)) : SPLIT_NONE;
      }
    }
  }
  if (threadIdx.x == 0) tile_start[G.TB] = total;             // = number of sorted records
  if (SV.on) {
    __syncthreads();
    if (threadIdx.x == 0) {
      *SV.n_extra = min(s_nextra, (unsigned int)SV.cap);
      if (s_nextra != SV.n_extra[1]) {                       // what the NEXT frames' launches should provide (host-mapped: a store across PCIe, only when it changes)
        SV.n_extra[1] = s_nextra;
        __hip_atomic_store(SV.need_host, s_nextra, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

#ifndef SCATTER_U
#define SCATTER_U 4
#endif
// the four carried channel columns of point i (SemCarry, emap_device.h): one aligned 16-byte load when the channel matrix is the
// de-interleaved (N, 4) one (on == 2), else column by column (columns past the row read as 0)
__device__ __forceinline__ float4 load_carry(const ChanView& V, const SemCarry& SC, long i) {
  if (SC.on == 2) return reinterpret_cast<const float4*>(V.p)[i];
  const float* __restrict__ row = chan_row(V, i);
  float4 r;
  r.x = SC.c0 < SC.ncols ? row[SC.c0] : 0.f; r.y = SC.c0 + 1 < SC.ncols ? row[SC.c0 + 1] : 0.f;
  r.z = SC.c0 + 2 < SC.ncols ? row[SC.c0 + 2] : 0.f; r.w = SC.c0 + 3 < SC.ncols ? row[SC.c0 + 3] : 0.f;
  return r;
}
// CH: the frame carries its semantic channels -- 32-byte records (BinRec32), the channel row read coalesced next to the point
template <int MODE, int BLK, bool STRIP, bool CH>
__global__ __launch_bounds__(BLK) void k_bin_scatter(KP P, Pose T, BinGeo G, const float* __restrict__ pts, long n, int stride,
                                                     const unsigned int* __restrict__ hist, const unsigned int* __restrict__ tile_start,
                                                     BinRec* __restrict__ recs, const BinStg* __restrict__ stg,
                                                     const unsigned int* __restrict__ stg_cnt, ChanView V, SemCarry SC) {
  static_assert(!(STRIP && CH), "a strip's staged records carry no channels");
  extern __shared__ unsigned int cur[];
  const unsigned int* row = hist + (long)blockIdx.x * G.pitch;
  for (int t = threadIdx.x; t < G.TB; t += BLK) cur[t] = tile_start[t] + row[t];
  __syncthreads();
  const long base = (long)blockIdx.x * G.chunk;
  // SCATTER_U loads in flight per thread (round 5): a block is ONE workgroup of eight waves on its CU (245 blocks of 4096 points at
  // 1 M points; 253 blocks and a 64-KB cursor array at 16 M), and a thread that walks its eight points one dependent load at a time
  // spends the pass waiting -- 8 round trips of ~1.5 us were the 13 us the pass took.
  constexpr int U = SCATTER_U;
  if (!STRIP) {
    for (long k0 = threadIdx.x; k0 < G.chunk; k0 += (long)U * BLK) {      // geometry once more, LDS cursor -> sorted position; no map access
      float rx[U], ry[U], rz[U];
      float4 ch[CH ? U : 1];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long i = base + k0 + (long)u * BLK;
        rx[u] = ry[u] = rz[u] = NAN;                          // (a NaN row: no bin)
        if (CH) ch[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k0 + (long)u * BLK < G.chunk && i < n) { load_point(pts, i, stride, rx[u], ry[u], rz[u]); if (CH) ch[u] = load_carry(V, SC, i); }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long i = base + k0 + (long)u * BLK;
        const Geo g = geometry<MODE>(P, T, rx[u], ry[u], rz[u]);
        unsigned int lc;
        const int bin = bin_of(P, G, g, lc);
        if (bin < 0) continue;
        const unsigned int pos = atomicAdd(&cur[bin], 1u);
        if (CH) {
          // both halves of one 32-byte sector.  (What a random store costs on this part is the 16-byte lane store, not the instruction or
          // the line it lands in: a lane PAIR writing each record -- even lane the head, odd lane the channels, 32 sectors per store
          // instruction instead of 64 -- left the pass at 466 us for 16 M points, round 6; the 16-byte records of a plain frame: 258.)
          BinRec32 o; o.lc_inl = lc; o.z = g.z; o.v = g.v; o.i = (unsigned int)i; o.c[0] = ch[u].x; o.c[1] = ch[u].y; o.c[2] = ch[u].z; o.c[3] = ch[u].w;
          reinterpret_cast<BinRec32*>(recs)[pos] = o;
        } else {
          BinRec o; o.lc_inl = lc; o.z = g.z; o.v = g.v; o.i = (unsigned int)i;
          recs[pos] = o;
        }
      }
    }
  } else {
    const unsigned int m = stg_cnt[blockIdx.x];              // a pure permutation of the block's staged records
    for (unsigned int j0 = threadIdx.x; j0 < m; j0 += U * BLK) {
      BinStg r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) if (j0 + u * BLK < m) r[u] = stg[base + j0 + u * BLK];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (j0 + u * BLK >= m) continue;
        const unsigned int pos = atomicAdd(&cur[r[u].key >> 16], 1u);
        BinRec o; o.lc_inl = r[u].key & 0xffffu; o.z = r[u].z; o.v = r[u].v; o.i = r[u].i;
        recs[pos] = o;
      }
    }
  }
}

// Which (bin, tile of the bin) a workgroup of a tile kernel handles.  sub == 1: one workgroup per bin, in order.  sub > 1 (maps beyond
// 16384 tiles): the `sub` workgroups of a bin all walk the bin's records and keep those of their own tile -- so they are placed 8
// apart inside a group of 8 * sub consecutive workgroups: workgroup l runs on XCD l % 8 (observed dispatch rule, for speed only), the
// siblings of a bin then share ONE XCD's L2 and start within a few dozen dispatches of each other: the bin's records come from HBM
// once instead of `sub` times (round 4; before, the siblings were T workgroups apart: 16384 dispatches, nothing left in any cache).
__device__ __forceinline__ void bin_of_block(const BinGeo& G, int& t, int& sb) {
  const unsigned int l = blockIdx.x;
  if (G.sub == 1) { t = (int)l; sb = 0; return; }
  const unsigned int span = 8u * (unsigned int)G.sub, g = l / span, r = l - g * span;
  sb = (int)(r >> 3); t = (int)(g * 8u + (r & 7u));            // (t may reach past the last bin in the final group: the callers return)
}
__host__ __device__ __forceinline__ unsigned int tile_grid(const BinGeo& G) { return G.sub == 1 ? (unsigned int)G.T : (unsigned int)(((G.T + 7) / 8) * 8 * G.sub); }
// tile (bin t, stacked tile sb) of workgroup l of the tile grid
__device__ __forceinline__ void bin_of_index(const BinGeo& G, unsigned int l, int& t, int& sb) {
  if (G.sub == 1) { t = (int)l; sb = 0; return; }
  const unsigned int span = 8u * (unsigned int)G.sub, g = l / span, r = l - g * span;
  sb = (int)(r >> 3); t = (int)(g * 8u + (r & 7u));            // (t may reach past the last bin in the final group: the callers return)
}
// What a workgroup reduces: tile `sb` of bin t, records [r0, r1) = part `part` of the bin's np parts; slot: the tile's scratch slot
// when np > 1; first: the first workgroup of the tile grid proper.  false: nothing (a grid slot past the last bin / the last listed
// part).  Uniform per workgroup.
// The further parts of heavy tiles run in EXTRA workgroups IN FRONT of the tile grid (SV.cap x sub of them; they start first).  The
// tile kernels run in whole rounds of workgroups (1024 tiles = two rounds of 512 resident workgroups at 1024^2), and 64 more that only
// look at an empty list and leave cost the uniform benchmark 1.1-1.4 us per kernel wherever they sit in the grid (measured; so did
// letting every workgroup take listed parts off a shared counter after its own tile: the second copy of the body doubled the
// kernel's LDS).  So the host launches only as many as the LAST frame it has heard of needed (k_bin_scan reports the need through a
// host-mapped word; emap_api.hip: split_capacity) and the scan lists no more parts than that: a tile that does not fit is reduced
// by its own workgroup alone -- slower, same bits.  A cloud without heavy tiles launches none.
// SPLIT = false (no extra workgroups in this launch: nothing can be split) compiles to the plain one-workgroup-per-tile kernels.
struct TileWork { int t, sb; unsigned int r0, r1, slot, np, part; bool first; };
template <bool SPLIT>
__device__ __forceinline__ bool tile_work(const BinGeo& G, const SplitView& SV, const unsigned int* __restrict__ tile_start, TileWork& w) {
  const unsigned int eg = SPLIT ? (unsigned int)SV.cap * (unsigned int)G.sub : 0u;
  w.part = 0u; w.first = blockIdx.x == eg;
  if (blockIdx.x >= eg) { bin_of_index(G, blockIdx.x - eg, w.t, w.sb); if (w.t >= G.T) return false; }
  else {
    const unsigned int x = blockIdx.x, e = x / (unsigned int)G.sub;
    if (e >= *SV.n_extra) return false;
    const unsigned int item = SV.extra[e];
    if (item == SPLIT_NONE) return false;                          // (the part of a tile the lists had no room for)
    w.sb = (int)(x - e * (unsigned int)G.sub); w.t = (int)(item >> 8); w.part = item & 255u;
  }
  const unsigned int R0 = tile_start[w.t], R1 = tile_start[w.t + 1], n = R1 - R0;
  w.np = 1u; w.slot = SPLIT_NONE; w.r0 = R0; w.r1 = R1;
  if (SPLIT && n > SPLIT_CAP) {                                  // a heavy tile (uniform); the scan's tail listed it -- or found no room
    const unsigned int sw = SV.tile_slot[w.t];                   // first slot | parts << 16 (k_bin_scan's tail: as many parts as there was room for)
    if (sw != SPLIT_NONE) {
      w.np = sw >> 16; w.slot = (sw & 0xffffu) + (unsigned int)w.sb;
      const unsigned int len = (n + w.np - 1u) / w.np;
      w.r0 = min(R1, R0 + w.part * len); w.r1 = min(R1, w.r0 + len);
    }
  }
  return true;
}

// drift-inlier test of error_counting_kernel (custom_kernels.py:317-335) on the (h, v, valid, trav) of a cell
__device__ __forceinline__ bool drift_inlier(const KP& P, const float4 m, float z) {
  return m.z > 0.5f && (double)fabsf(m.x - z) < (double)m.y * P.mt && (double)m.y < P.dcvi_half && (double)m.w > P.trav_inlier;
}

#define TF_BLOCK 1024   /* threads per tile of the tile kernels */
#ifndef SPLIT_U
#define SPLIT_U 4         /* records per thread and trip in k_tile_count<true> (k_tile_fuse<.., true>: one -- four were slower, 76 -> 92 us on the terrain) */
#endif
#ifndef SPLIT_U_FUSE
#define SPLIT_U_FUSE 1
#endif
#ifndef TILE_PREFETCH
#define TILE_PREFETCH 1   /* tile kernels: first records requested before the staging barrier (A/B knob) */
#endif
// The hot halves (h, v, valid, trav) of one 16 x 64 tile -> LDS, one wave per tile row.  Without pending map moves this is a pure
// copy of 1 KB per row: gfx950's LDS-DMA (no VGPR round trip, no ds_write); cells beyond the map / strip read as zero.  With pending
// moves the cells pass through cell_now() in registers.  The caller's next __syncthreads() publishes the tile.
__device__ __forceinline__ void stage_hot_tile(const KP& P, Cells cells, float4* s_cell, int row_base, int tx) {
  const int tr = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), tc = threadIdx.x & 63, lrow = row_base + tr, col = tx * BIN_TC + tc;   // 1024 threads = 16 x 64 cells
  const bool in = lrow < P.nrows && col < P.C;
  const float4* src = cells.hot + ((long)(lrow + P.halo) * P.C + col);
  if (P.mv.n == 0) {                                       // (uniform)
    if (in) lds_dma16(src, s_cell + tr * BIN_TC);
    else s_cell[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in) { m = *src; cell_now(P, m, P.row0 + lrow, col); }
    s_cell[threadIdx.x] = m;
  }
}
// Error sums of the drift compensation, per tile: the tile's cells are staged ONCE, coalesced, in LDS and every sorted record of
// the tile is tested against its cell there -- the per-point gather of a random 32-byte cell (a whole 128-byte line per point,
// 144 MB fetched for 48 MB needed, profiles/r01f_pmc_cfg2.json) is gone.  Wave-reduced sums go to the 256 padded slots.
// RS: stride of the sorted records in 16-byte units (2: BinRec32 of a frame that carries its channels; only the leading 16 bytes are read here)
template <bool SPLIT, int RS>
__device__ __forceinline__ void tile_count_body(const KP& P, const BinGeo& G, const BinRec* __restrict__ recs, Cells cells,
                                                ErrSlot* __restrict__ slots, const SplitView& SV, const TileWork& w) {
  constexpr int NC = BIN_TR * BIN_TC;
  __shared__ float4 s_cell[NC];
  __shared__ unsigned int s_pts[NC], s_inl[NC];             // parts of a heavy tile only: per-cell counts for k_tile_fuse
  const int t = w.t, sb = w.sb;
  const int ty = t / G.tiles_x, tx = t - ty * G.tiles_x;
  const unsigned int r0 = w.r0, r1 = w.r1;
  const int row_base = (ty * G.sub + sb) * BIN_TR;
  if (row_base >= P.nrows || r0 == r1) return;
  const bool split = SPLIT && w.np > 1u;                             // (uniform)
  // the split kernels (frames with heavy tiles) keep SPLIT_U records per thread in flight: a part is up to SPLIT_CAP / 1024 dependent
  // trips of load -> LDS, each a full memory round trip when only one load per thread is outstanding
  constexpr int U = SPLIT ? SPLIT_U : 1;
  // (round 5) the first batch of records is requested BEFORE the tile's cells are staged: the two round trips overlap instead of
  // following each other behind the barrier -- the tile kernels are chains of dependent trips, 128 rounds of them at 8192^2
  BinRec rr[U];
#if TILE_PREFETCH
#pragma unroll
  for (int u = 0; u < U; ++u) { const unsigned int k = r0 + u * TF_BLOCK + threadIdx.x; if (k < r1) rr[u] = recs[(size_t)k * RS]; }
#endif
  stage_hot_tile(P, cells, s_cell, row_base, tx);
  if (split) { s_pts[threadIdx.x] = 0u; s_inl[threadIdx.x] = 0u; }
  __syncthreads();
  const unsigned int sel = (unsigned int)sb;
  for (unsigned int kb = r0; kb < r1; kb += TF_BLOCK * U) {     // uniform trip count: the wave reductions need all lanes
#if TILE_PREFETCH
    if (kb != r0)
#endif
#pragma unroll
    for (int u = 0; u < U; ++u) { const unsigned int k = kb + u * TF_BLOCK + threadIdx.x; if (k < r1) rr[u] = recs[(size_t)k * RS]; }
    long long e_fix = 0; unsigned long long cnt = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned int k = kb + u * TF_BLOCK + threadIdx.x;
      bool inl = false;
      if (k < r1) {
        const BinRec& r = rr[u];
        const unsigned int lcb = r.lc_inl & 0x7fffffffu;
        if ((lcb >> 10) == sel) {
          const float4 m = s_cell[lcb & 1023u];
          if (drift_inlier(P, m, r.z)) { inl = true; e_fix += __double2ll_rn((double)(r.z - m.x) * EM_SCALE_E); }
          if (split) { atomicAdd(&s_pts[lcb & 1023u], 1u); if (inl) atomicAdd(&s_inl[lcb & 1023u], 1u); }
        }
      }
      cnt += __popcll(__ballot(inl));
    }
    if (cnt) {                                                 // (wave-uniform)
      const long long s = wave_sum_ll(e_fix);
      if ((threadIdx.x & 63) == 0) {
        const unsigned int slot = (unsigned int)(((unsigned int)t * (TF_BLOCK / 64) + (threadIdx.x >> 6) + kb) & (EM_ERR_SLOTS - 1));
        atomicAdd(reinterpret_cast<unsigned long long*>(&slots[slot].sum), (unsigned long long)s);
        atomicAdd(&slots[slot].cnt, cnt);
      }
    }
  }
  if (split) {
    __syncthreads();
    const unsigned int i = w.slot * SPLIT_CELLS + threadIdx.x;
    if (s_pts[threadIdx.x]) atomicAdd(&SV.pts[i], s_pts[threadIdx.x]);
    if (s_inl[threadIdx.x]) atomicAdd(&SV.inl[i], s_inl[threadIdx.x]);
  }
}
template <bool SPLIT, int RS>
__global__ __launch_bounds__(TF_BLOCK) void k_tile_count(KP P, BinGeo G, const BinRec* __restrict__ recs,
                                                          const unsigned int* __restrict__ tile_start, Cells cells,
                                                          ErrSlot* __restrict__ slots, SplitView SV) {
  TileWork w;
  if (tile_work<SPLIT>(G, SV, tile_start, w)) tile_count_body<SPLIT, RS>(P, G, recs, cells, slots, SV, w);
}

// AVG = true (whole frames, emap_update): the epilogue commits AND averages the tile in registers and writes the 32-byte cells
// directly -- the AccF records never leave LDS and the separate k_commit / k_average passes disappear.  That is also valid in front
// of the visibility pass (RAYS = true): a cell fused this frame is known and fresh (valid = 1, time = 0), so every ray skips it
// (custom_kernels.py:236) and its averaged (or reset, :374-377) state is never looked at -- the kernel records it as inert in the
// bitmap k_rays tests first; a cell that is NOT fused is changed by commit / average only in ways the rays do not read (outlier
// variance is committed here; the reset of unknown cells touches h, v, valid of cells that already have valid < 0.5).  What the
// rays additionally need is written here too: the inert bitmap (one wave ballot = one 64-bit word per tile row) and newmap[3],
// the per-cell drift-inlier counts (`inl_plane`, read only when a ray penetrates a cell).  The ray effects are applied afterwards
// by k_ray_apply.  AVG = false keeps the staged contract: AccF records for k_commit / k_rays / k_average.
// RS: stride of the sorted records in 16-byte units (2: BinRec32, the frame carries its semantic channels).
// SEM (round 6; AVG frames without a visibility pass and without heavy-tile parts, RS = 2): the RGB / semantic point fusion of the
// frame (semantic_map.py:223-259; custom_semantic_kernels.py:9-51 sum, :167-194 average / class_average, :233-267 colour) runs HERE,
// as a third pass over the tile's records after the cells have been written: the accumulators of the Kalman pass and the staged
// cells are dead by then, their LDS holds the fp64 channel sums and the colour sums; the count the averages divide by (newmap[2],
// accepted height points) and the colour's point count (= the points per cell of pass 1) are still in LDS -- no count plane, no
// second kernel over the records, no gather of channel rows by point index (k_tile_semantic below stays for everything else).
template <bool AVG, bool RAYS, bool SPLIT, int RS, bool SEM>
__device__ __forceinline__ void tile_fuse_body(const KP& P, const BinGeo& G, const BinRec* __restrict__ recs, Cells cells,
                                               AccF* __restrict__ acc, FrameDev* __restrict__ F,
                                               unsigned int* __restrict__ cnt_plane, unsigned long long* __restrict__ inert,
                                               unsigned int* __restrict__ inl_plane, float* __restrict__ thr, const OverlapArgs& O, const GateFold& GF,
                                               const SplitView& SV, const TileWork& w, const SemMini& SM) {
  static_assert(!SEM || (AVG && !RAYS && !SPLIT && RS == 2), "the semantic pass rides on whole frames without rays / parts, on 32-byte records");
  constexpr int NC = BIN_TR * BIN_TC;
  // one block of LDS: [pts | cnt] stay alive through the semantic pass, everything behind them is re-used by it
  struct FuseLds { unsigned int pts[NC], cnt[NC], inl[NC], out[NC]; unsigned long long h[NC], v[NC], latest[NC]; float4 cell[NC]; };
  static_assert(offsetof(FuseLds, cell) == offsetof(FuseLds, h) + 3 * NC * 8 && sizeof(FuseLds) == 56 * NC, "FuseLds is packed");
  __shared__ __attribute__((aligned(16))) FuseLds L;
  unsigned int (&s_pts)[NC] = L.pts; unsigned int (&s_inl)[NC] = L.inl; unsigned int (&s_cnt)[NC] = L.cnt; unsigned int (&s_out)[NC] = L.out;
  unsigned long long (&s_h)[NC] = L.h; unsigned long long (&s_v)[NC] = L.v; unsigned long long (&s_latest)[NC] = L.latest;
  float4 (&s_cell)[NC] = L.cell;            // (h, v, valid, trav) of the tile's cells, staged once (coalesced): no per-record gather
  __shared__ float s_shift;
  __shared__ bool s_final;
  const int t = w.t, sb = w.sb;
  const int ty = t / G.tiles_x, tx = t - ty * G.tiles_x;
  const unsigned int r0 = w.r0, r1 = w.r1;
  const bool split = SPLIT && w.np > 1u;                   // (uniform)
  const int tc = threadIdx.x & 63, wv = threadIdx.x >> 6, col = tx * BIN_TC + tc;
  {
    const int row_base = (ty * G.sub + sb) * BIN_TR;
    if (row_base >= P.nrows) return;              // uniform
    const unsigned int sel = (unsigned int)sb;
    if (threadIdx.x == 0) s_shift = GF.mode ? gate_fold(GF, F, w.first) : F->shift;      // only pass 2 needs it (the first workgroup of the tile grid = bin 0, tile 0: always present)
    // (round 5) the thread's FIRST record is requested before the cells are staged and the accumulators zeroed, and serves both
    // passes: a tile of a uniform cloud holds about as many records as the workgroup has threads, so the second pass's trip to L2
    // disappears and the first one overlaps the staging
    const unsigned int k_first = r0 + threadIdx.x;
    BinRec first; first.lc_inl = 0u; first.z = 0.f; first.v = 0.f; first.i = 0u;
    float4 first_ch = make_float4(0.f, 0.f, 0.f, 0.f);      // (SEM: the record's channels, for the third pass)
    const bool have_first = TILE_PREFETCH && !SPLIT && k_first < r1;
    if (have_first) { first = recs[(size_t)k_first * RS]; if (SEM) first_ch = reinterpret_cast<const float4*>(recs)[(size_t)k_first * RS + 1]; }
    stage_hot_tile(P, cells, s_cell, row_base, tx);
    for (int k = threadIdx.x; k < NC; k += TF_BLOCK) { s_pts[k] = 0u; s_inl[k] = 0u; s_cnt[k] = 0u; s_out[k] = 0u; s_h[k] = 0ull; s_v[k] = 0ull; s_latest[k] = 0ull; }
    __syncthreads();
    if (split) {                                                               // a heavy tile: the WHOLE tile's counts, left in the slot by k_tile_count's parts
      static_assert(NC == TF_BLOCK && NC == (int)SPLIT_CELLS, "one thread per cell of a slot");
      s_pts[threadIdx.x] = SV.pts[w.slot * SPLIT_CELLS + threadIdx.x];
      if (!AVG || RAYS) s_inl[threadIdx.x] = SV.inl[w.slot * SPLIT_CELLS + threadIdx.x];     // (as below: only the ray pass reads newmap[3])
    } else
    for (unsigned int k = r0 + threadIdx.x; k < r1; k += TF_BLOCK) {          // pass 1: newmap[4] / newmap[3]
      const BinRec r = (have_first && k == k_first) ? first : recs[(size_t)k * RS];
      const unsigned int lcb = r.lc_inl & 0x7fffffffu;
      if ((lcb >> 10) != sel) continue;
      atomicAdd(&s_pts[lcb & 1023u], 1u);
      if ((!AVG || RAYS) && drift_inlier(P, s_cell[lcb & 1023u], r.z)) atomicAdd(&s_inl[lcb & 1023u], 1u);   // newmap[3]: only the ray pass reads it
    }
    __syncthreads();
    const float shift = s_shift;
    constexpr int U = SPLIT ? SPLIT_U_FUSE : 1;                               // (records per thread in flight: see tile_count_body)
    for (unsigned int kb = r0 + threadIdx.x; kb < r1; kb += TF_BLOCK * U) {   // pass 2: custom_kernels.py:160-197
     BinRec rr[U];
#pragma unroll
     for (int u = 0; u < U; ++u) if (kb + u * TF_BLOCK < r1) rr[u] = (have_first && u == 0 && kb == k_first) ? first : recs[(size_t)(kb + u * TF_BLOCK) * RS];
#pragma unroll
     for (int u = 0; u < U; ++u) {
      if (kb + u * TF_BLOCK >= r1) continue;
      const BinRec& r = rr[u];
      const unsigned int lcb = r.lc_inl & 0x7fffffffu;
      if ((lcb >> 10) != sel) continue;
      const unsigned int lc = lcb & 1023u;
      const float4 hv = s_cell[lc];
      const float map_h = hv.x + shift, map_v = hv.y;
      const float num_points = (float)s_pts[lc];
      if ((double)fabsf(map_h - r.z) > (double)map_v * P.mt) { atomicAdd(&s_out[lc], 1u); continue; }
      if (P.edge && (double)num_points > P.wall && (double)r.z < (double)map_h - (double)map_v * P.mt / (double)num_points) continue;
      const float new_h = (map_h * r.v + r.z * map_v) / (map_v + r.v);
      const float new_v = (map_v * r.v) / (map_v + r.v);
      atomicAdd(&s_h[lc], (unsigned long long)__double2ll_rn((double)new_h * EM_SCALE_H));
      atomicAdd(&s_v[lc], (unsigned long long)__double2ll_rn((double)new_v * EM_SCALE_V));
      atomicAdd(&s_cnt[lc], 1u);
      atomicMax(&s_latest[lc], ((unsigned long long)(r.i + 1u) << 32) | (unsigned long long)__float_as_uint(new_h));
     }
    }
    __syncthreads();
    if (split) {
      // This part's sums join the tile's slot (device atomics: integer adds and an ordered maximum, any order gives the same bits),
      // then a ticket: the part that arrives LAST owns the tile -- it reads the totals back (device-coherent loads: the atomics were
      // performed at the memory side, not in this XCD's L2), leaves slot and ticket zeroed for the next frame and runs the epilogue.
      const unsigned int i = w.slot * SPLIT_CELLS + threadIdx.x;
      if (s_cnt[threadIdx.x] | s_out[threadIdx.x]) {
        if (s_cnt[threadIdx.x]) {
          atomicAdd(&SV.h[i], s_h[threadIdx.x]); atomicAdd(&SV.v[i], s_v[threadIdx.x]);
          atomicAdd(&SV.cnt[i], s_cnt[threadIdx.x]); atomicMax(&SV.latest[i], s_latest[threadIdx.x]);
        }
        if (s_out[threadIdx.x]) atomicAdd(&SV.out[i], s_out[threadIdx.x]);
      }
      // Order: the atomics above are performed at the memory side (device scope); s_waitcnt 0 waits for their acknowledgements, the
      // barrier collects the workgroup, then the ticket -- the hand-off k_bin_scan uses (last_block_ticket).  A release FENCE here
      // writes back the XCD's whole L2 for every part: 1.6 us per part, measured (fuse 76 -> 163 us going from 19 to 73 parts).
#ifdef SPLIT_FENCE
      __threadfence();
#else
      __builtin_amdgcn_s_waitcnt(0);
#endif
      __syncthreads();
      if (threadIdx.x == 0) {
        const unsigned int arrived = __hip_atomic_fetch_add(&SV.tick[w.slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_final = arrived == w.np - 1u;
        if (s_final) __hip_atomic_store(&SV.tick[w.slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      if (!s_final) return;                                  // (uniform)
#ifdef SPLIT_FENCE
      __threadfence();
#endif
      s_h[threadIdx.x] = __hip_atomic_load(&SV.h[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_v[threadIdx.x] = __hip_atomic_load(&SV.v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_latest[threadIdx.x] = __hip_atomic_load(&SV.latest[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_cnt[threadIdx.x] = __hip_atomic_load(&SV.cnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_out[threadIdx.x] = __hip_atomic_load(&SV.out[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (s_cnt[threadIdx.x] | s_out[threadIdx.x]) {
        __hip_atomic_store(&SV.h[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(&SV.v[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&SV.latest[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&SV.cnt[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(&SV.out[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (s_pts[threadIdx.x]) { __hip_atomic_store(&SV.pts[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(&SV.inl[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      __syncthreads();
    }
    {
      static_assert(BIN_TR == TF_BLOCK / 64, "one wave per tile row");
      const int tr = wv, lrow = row_base + tr;
      const bool live = col < P.C && lrow < P.nrows;
      bool quiet = false, other = false;  // other: neither quiet nor unknown-without-a-bound (see the block thresholds below)
      float visit_thr = -INFINITY;        // a ray sample at or above this height cannot affect the cell (k_rays' block filter)
      if (live) {
        const int lc = tr * BIN_TC + tc;
        AccF a;
        a.pts_inl = (unsigned long long)s_pts[lc] | ((unsigned long long)s_inl[lc] << 32);
        a.cnt_out = (unsigned long long)s_cnt[lc] | ((unsigned long long)s_out[lc] << 32);
        a.sum_h = (long long)s_h[lc]; a.sum_v = (long long)s_v[lc]; a.latest = s_latest[lc];
        const long c = (long)(lrow + P.halo) * P.C + col;
        if (AVG) {
          // The hot half of the cell is in LDS already; the cold half (time, upper, is_upper) is READ only where something looks at it --
          // a cell without points in front of a visibility pass (stale / unknown tests below), a cell of the overlap window -- and
          // WRITTEN only where it changes: a fused cell (commit overwrites all of it), the overlap window, or the first frame after a
          // map move (the pending moves are written out with this rewrite: the whole cell then takes the slow way).
          const bool inwin = !RAYS && overlap_window(O, logi_row(P, P.row0 + lrow), logi_col(P, col));
          Cell m;
          bool wcold, whot = true;
          float4 hq0 = make_float4(0.f, 0.f, 0.f, 0.f);      // the hot half as it lies in memory (no pending moves)
          if (P.mv.n) {                                        // (uniform)
            m = cells[c];
            cell_now(P, m, P.row0 + lrow, col);
            wcold = true;
          } else {
            const float4 hq = s_cell[lc];
            hq0 = hq;
            const bool fused = s_cnt[lc] != 0u;
            const float4 cq = ((RAYS && !fused) || inwin) ? cells.cold[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            m.h = hq.x; m.v = hq.y; m.valid = hq.z; m.trav = hq.w; m.time = cq.x; m.upper = cq.y; m.is_upper = cq.z; m.pad = 0.f;
            wcold = fused;
          }
          m.h += shift;
          commit_cell(P, m, a);
          quiet = (!(m.valid < 0.5f) && m.time < 0.5f) || border_cell(P, logi_row(P, P.row0 + lrow), logi_col(P, col));   // snapshot S1 (what the rays are defined on); border cells: see k_commit
          // unknown cell: acts only while nz < upper_bound (or no bound yet, :229); known stale cell: the penetration test (:239)
          // h > nz + 0.01 - min(v, 1) * 0.05 implies nz < h + 0.04
          if (RAYS && !quiet) visit_thr = m.valid < 0.5f ? ((m.is_upper < 0.5f || !(m.upper <= 3.0e38f)) ? INFINITY : m.upper) : m.h + 0.05f;
          if (RAYS && !(visit_thr >= -INFINITY)) visit_thr = INFINITY;                          // NaN heights: never filter
          // a cell the rays can only give an upper bound to (unknown, no bound yet): the ray kernel needs nothing of it but its key
          if (RAYS && !quiet) other = !(m.valid < 0.5f && m.is_upper < 0.5f);
          average_cell(P, m, a);
          // clear_overlap_map (:372-375) of a frame WITHOUT a visibility pass rides on this rewrite (with rays: k_ray_apply)
          if (inwin) wcold = overlap_cell(P, O, m) || wcold;
          // A cell that comes out of the frame with the bits it went in with is not stored again: an unknown cell without points
          // stays (0, initial_variance, 0) whatever the drift shift -- 86 % of the cells of a robot's map, 60 % at 8192^2 / 16 M
          // points -- and its 16 bytes of write traffic stay away (round 5; the staged copy in LDS is what memory holds).
          if (P.mv.n == 0) whot = (__float_as_uint(m.h) ^ __float_as_uint(hq0.x)) | (__float_as_uint(m.v) ^ __float_as_uint(hq0.y)) |
                                  (__float_as_uint(m.valid) ^ __float_as_uint(hq0.z)) | (__float_as_uint(m.trav) ^ __float_as_uint(hq0.w));
          if (whot) cells.hot[c] = make_float4(m.h, m.v, m.valid, m.trav);
          if (wcold) cells.cold[c] = make_float4(m.time, m.upper, m.is_upper, m.valid);
          if (cnt_plane) cnt_plane[c] = s_cnt[lc];
          if (RAYS) inl_plane[c] = s_inl[lc];
        } else acc[c] = a;
      }
      if (AVG && RAYS) {                                       // every wave of the workgroup gets here (no early exit above)
        if (lrow < P.nrows) {                                  // wave-uniform condition: the ballot sees the whole row segment
          // the bitmap is indexed by LOGICAL column (k_commit): the 64 physical columns of this wave are one or two runs of logical
          // columns that straddle word boundaries -> OR the shifted pieces into the (pre-zeroed) words
          const unsigned long long bits = __ballot(quiet);
          if (tc == 0 && ((P.org_c | P.C) & 63) == 0) inert[(long)bitmap_row(P, P.row0 + lrow) * (P.C / 64) + logi_col(P, tx * BIN_TC) / 64] = bits;   // aligned: one whole word
          else if (tc == 0) {
            unsigned long long* row = inert + (long)bitmap_row(P, P.row0 + lrow) * ((P.C + 63) / 64);
            const int ncol = min(64, P.C - tx * BIN_TC);                      // physical columns of this segment
            int done = 0;
            while (done < ncol) {                                              // runs of consecutive logical columns (<= 2)
              const int l0 = logi_col(P, tx * BIN_TC + done), run = min(ncol - done, P.C - l0);
              const unsigned long long piece = (run == 64 ? bits : ((bits >> done) & ((1ull << run) - 1ull)));
              const int sh = l0 & 63;
              if (piece << sh) atomicOr(&row[l0 >> 6], piece << sh);
              if (sh && (piece >> (64 - sh))) atomicOr(&row[(l0 >> 6) + 1], piece >> (64 - sh));
              done += run;
            }
          }
        }
        // block thresholds: max over 8 columns (lanes) and 8 rows (waves) through the ordered-uint image of the float.  +INF is kept
        // for VIRGIN blocks only -- every cell quiet or unknown without a bound, the state of a cleared map and of the band a map
        // shift brings in: there a visit needs nothing but the cell's key (k_rays skips both cell loads); a block that ALSO holds
        // other cells gets the largest finite float instead (filters nothing either)
        unsigned int* s_thr = s_pts;                           // the counters are dead by now: 2 block rows x 8 block columns (+ 16 "other" flags)
        __syncthreads();
        if (threadIdx.x < 32) s_thr[threadIdx.x] = 0u;         // float_ord(x) > 0 for every x
        __syncthreads();
        unsigned int o = float_ord(visit_thr);
        o = max(o, (unsigned int)__shfl_xor((int)o, 1, 64)); o = max(o, (unsigned int)__shfl_xor((int)o, 2, 64)); o = max(o, (unsigned int)__shfl_xor((int)o, 4, 64));
        const unsigned long long ob = __ballot(other);
        if ((tc & 7) == 0) {
          atomicMax(&s_thr[(tr >> 3) * 8 + (tc >> 3)], o);
          if ((ob >> tc) & 0xffull) s_thr[16 + (tr >> 3) * 8 + (tc >> 3)] = 1u;       // (racing writers store the same value)
        }
        __syncthreads();
        if (threadIdx.x < 16) {
          const int br = (row_base >> 3) + (threadIdx.x >> 3), bc = tx * 8 + (threadIdx.x & 7);
          float bt = ord_float(s_thr[threadIdx.x]);
          if (bt == INFINITY && s_thr[16 + threadIdx.x]) bt = 3.4028234664e38f;
          if (br * 8 < P.nrows && bc * 8 < P.C) {
            thr[(long)br * ((P.C + 7) >> 3) + bc] = bt;
          }
        }
      }
      if (SEM) {
        if (r0 == r1) return;                                    // (uniform) a tile without records: no cell has a count, nothing is written
        // the semantic view of the block behind [pts | cnt]: colour sums r, g over inl / out, four fp64 channel sums over h, v, latest
        // and the first half of the staged cells, colour sum b over the third quarter of the staged cells
        unsigned int* const s_cr = L.inl; unsigned int* const s_cg = L.out;
        double* const s_sum = reinterpret_cast<double*>(L.h);                         // [4][NC]
        unsigned int* const s_cb = reinterpret_cast<unsigned int*>(L.cell) + 2 * NC;
        static_assert(4 * NC * sizeof(double) == 3 * NC * 8 + 2 * NC * 4, "s_sum ends where s_cb begins");
        __syncthreads();                                          // every wave is through its epilogue: accumulators and staged cells are dead
        const int ns = SM.n_sum;
        for (int k = threadIdx.x; k < ns * NC; k += TF_BLOCK) s_sum[k] = 0.0;
        if (SM.n_col) { s_cr[threadIdx.x] = 0u; s_cg[threadIdx.x] = 0u; s_cb[threadIdx.x] = 0u; }
        __syncthreads();
        auto pick = [](const float4& q, int j) { return j == 0 ? q.x : (j == 1 ? q.y : (j == 2 ? q.z : q.w)); };
        for (unsigned int k = r0 + threadIdx.x; k < r1; k += TF_BLOCK) {          // pass 3: custom_semantic_kernels.py:9-51, :233-267
          const bool pf = have_first && k == k_first;
          const unsigned int lcb = (pf ? first.lc_inl : recs[(size_t)k * RS].lc_inl) & 0x7fffffffu;
          if ((lcb >> 10) != sel) continue;
          const float4 ch = pf ? first_ch : reinterpret_cast<const float4*>(recs)[(size_t)k * RS + 1];
          const unsigned int lc = lcb & 1023u;
          for (int q = 0; q < ns; ++q) unsafeAtomicAdd(&s_sum[q * NC + lc], (double)pick(ch, SM.slot[q]));
          if (SM.n_col) {
            const unsigned int color = __float_as_uint(pick(ch, SM.col_slot));
            atomicAdd(&s_cr[lc], (color & 0xFF0000u) >> 16);
            atomicAdd(&s_cg[lc], (color & 0xFF00u) >> 8);
            atomicAdd(&s_cb[lc], color & 0xFFu);
          }
        }
        __syncthreads();
        if (live) {                                               // :167-194 (average / class_average), :254-267 (colour) on this thread's cell
          const int lc = tr * BIN_TC + tc;
          const long c = (long)(lrow + P.halo) * P.C + col;
          const unsigned int cnt = s_cnt[lc], cn = s_pts[lc];     // accepted height points (newmap[2]); points of the cell (the colour's own count)
          if (SM.n_col && cn) {
            const unsigned int rr = s_cr[lc] / cn, gg = s_cg[lc] / cn, bb = s_cb[lc] / cn;
            SM.sem[(long)SM.col_layer * SM.plane + c] = __uint_as_float((rr << 16) + (gg << 8) + bb);
          }
          if (cnt) for (int q = 0; q < ns; ++q) {
            const long j = (long)SM.layer[q] * SM.plane + c;
            const double sq = s_sum[q * NC + lc];
            if (SM.kind[q] == 0) SM.sem[j] = (float)(sq / (double)cnt);
            else {
              const float prev = SM.sem[j];
              SM.sem[j] = (prev == 0.0f) ? (float)(sq / (double)cnt) : (float)(SM.alpha * (double)prev + (1.0 - SM.alpha) * sq / (double)cnt);
            }
          }
        }
      }
    }
  }
}
template <bool AVG, bool RAYS, bool SPLIT, int RS, bool SEM>
__global__ __launch_bounds__(TF_BLOCK) void k_tile_fuse(KP P, BinGeo G, const BinRec* __restrict__ recs,
                                                         const unsigned int* __restrict__ tile_start, Cells cells,
                                                         AccF* __restrict__ acc, FrameDev* __restrict__ F,
                                                         unsigned int* __restrict__ cnt_plane, unsigned long long* __restrict__ inert,
                                                         unsigned int* __restrict__ inl_plane, float* __restrict__ thr, OverlapArgs O, GateFold GF, SplitView SV, SemMini SM) {
  TileWork w;                                     // sb = which 16 x 64 tile of the bin (sub == 1 up to 16384 tiles); for a heavy tile: a part of its records
  if (tile_work<SPLIT>(G, SV, tile_start, w)) tile_fuse_body<AVG, RAYS, SPLIT, RS, SEM>(P, G, recs, cells, acc, F, cnt_plane, inert, inl_plane, thr, O, GF, SV, w, SM);
}

static inline unsigned int nb(long n) { return (unsigned int)((n + EM_BLOCK - 1) / EM_BLOCK); }

// count stage of the binned path = hist, scans, scatter (error sums land in `slots` exactly as k_count leaves them)
// workgroup sizes of the two point passes (measured on MI355X, 1 M points / 1024 tiles: the histogram pass wants many
// loads in flight -> 1024 threads; the scatter pass 512; 4096-point chunks).  Tuning knobs only.
static int env_block(const char* name, int dflt) {
  if (const char* e = getenv(name)) { int v = atoi(e); if (v == 256 || v == 512 || v == 1024) return v; }
  return dflt;
}
// dynamic LDS of the histogram pass: the histogram row [+ the waves' compaction queues and the block's staging counter on strips]
static size_t bin_lds_bytes(const BinGeo& G, bool strip, int blk) { return sizeof(unsigned int) * G.pitch + (strip ? (size_t)(blk / 64) * BIN_QCAP * 16 + 16 : 0); }
template <int MODE, int BLK, bool STRIP>
static void launch_bin_hist_i(hipStream_t s, const KP& P, const Pose& T, const BinGeo& G, const float* pts, long n, int stride,
                              unsigned int* hist, BinStg* own, unsigned int* own_cnt) {
  static LdsRaised raised;                         // 16384 tiles + the ray-only bin (+ queues): past the default 64 KB window
  raise_lds(k_bin_hist<MODE, BLK, STRIP>, raised, 112 * 1024);
  hipLaunchKernelGGL((k_bin_hist<MODE, BLK, STRIP>), dim3(G.B), dim3(BLK), bin_lds_bytes(G, STRIP, BLK), s, P, T, G, pts, n, stride, hist, own, own_cnt);
}
template <int BLK>
static void launch_bin_hist_t(hipStream_t s, const KP& P, const Pose& T, const BinGeo& G, const float* pts, long n, int stride,
                              unsigned int* hist, BinStg* own, unsigned int* own_cnt) {
  if (own) { if (P.mode == 0) launch_bin_hist_i<0, BLK, true>(s, P, T, G, pts, n, stride, hist, own, own_cnt); else launch_bin_hist_i<1, BLK, true>(s, P, T, G, pts, n, stride, hist, own, own_cnt); }
  else { if (P.mode == 0) launch_bin_hist_i<0, BLK, false>(s, P, T, G, pts, n, stride, hist, own, own_cnt); else launch_bin_hist_i<1, BLK, false>(s, P, T, G, pts, n, stride, hist, own, own_cnt); }
}
// own != nullptr selects the strip variants (cheap ownership test + lane compaction)
void launch_bin_hist(hipStream_t s, const KP& P, const Pose& T, const BinGeo& G, const float* pts, long n, int stride,
                     unsigned int* hist, BinStg* own, unsigned int* own_cnt) {
  static const int blk = env_block("EMAP_HIST_BLOCK", 1024);
  switch (blk) {
    case 1024: launch_bin_hist_t<1024>(s, P, T, G, pts, n, stride, hist, own, own_cnt); break;
    case 512: launch_bin_hist_t<512>(s, P, T, G, pts, n, stride, hist, own, own_cnt); break;
    default: launch_bin_hist_t<256>(s, P, T, G, pts, n, stride, hist, own, own_cnt);
  }
}
void launch_bin_scan(hipStream_t s, const BinGeo& G, unsigned int* hist, unsigned int* tile_total, unsigned int* tile_start, unsigned int* sync, const SplitView& SV) {
  hipLaunchKernelGGL(k_bin_scan, dim3((G.TB + SCAN_TT - 1) / SCAN_TT), dim3(SCAN_BLK), 0, s, G, hist, tile_total, tile_start, sync, SV);
}
template <int MODE, int BLK, bool STRIP, bool CH>
static void launch_bin_scatter_i(hipStream_t s, const KP& P, const Pose& T, const BinGeo& G, const float* pts, long n, int stride,
                                 const unsigned int* hist, const unsigned int* tile_start, BinRec* recs, const BinStg* own, const unsigned int* own_cnt,
                                 const ChanView& V, const SemCarry& SC) {
  static LdsRaised raised;
  raise_lds(k_bin_scatter<MODE, BLK, STRIP, CH>, raised, 80 * 1024);
  hipLaunchKernelGGL((k_bin_scatter<MODE, BLK, STRIP, CH>), dim3(G.B), dim3(BLK), sizeof(unsigned int) * G.pitch, s, P, T, G, pts, n, stride, hist, tile_start, recs, own, own_cnt, V, SC);
}
template <int BLK>
static void launch_bin_scatter_t(hipStream_t s, const KP& P, const Pose& T, const BinGeo& G, const float* pts, long n, int stride,
                                 const unsigned int* hist, const unsigned int* tile_start, BinRec* recs, const BinStg* own, const unsigned int* own_cnt,
                                 const ChanView& V, const SemCarry& SC) {
  if (own) { if (P.mode == 0) launch_bin_scatter_i<0, BLK, true, false>(s, P, T, G, pts, n, stride, hist, tile_start, recs, own, own_cnt, V, SC); else launch_bin_scatter_i<1, BLK, true, false>(s, P, T, G, pts, n, stride, hist, tile_start, recs, own, own_cnt, V, SC); }
  else if (SC.on) { if (P.mode == 0) launch_bin_scatter_i<0, BLK, false, true>(s, P, T, G, pts, n, stride, hist, tile_start, recs, own, own_cnt, V, SC); else launch_bin_scatter_i<1, BLK, false, true>(s, P, T, G, pts, n, stride, hist, tile_start, recs, own, own_cnt, V, SC); }
  else { if (P.mode == 0) launch_bin_scatter_i<0, BLK, false, false>(s, P, T, G, pts, n, stride, hist, tile_start, recs, own, own_cnt, V, SC); else launch_bin_scatter_i<1, BLK, false, false>(s, P, T, G, pts, n, stride, hist, tile_start, recs, own, own_cnt, V, SC); }
}
// SC.on: the records are 32-byte BinRec32 (never together with the strip variants: own == nullptr then)
void launch_bin_scatter(hipStream_t s, const KP& P, const Pose& T, const BinGeo& G, const float* pts, long n, int stride,
                        const unsigned int* hist, const unsigned int* tile_start, BinRec* recs, const BinStg* own, const unsigned int* own_cnt,
                        const ChanView& V, const SemCarry& SC) {
  static const int blk = env_block("EMAP_SCATTER_BLOCK", 512);
  switch (blk) {
    case 1024: launch_bin_scatter_t<1024>(s, P, T, G, pts, n, stride, hist, tile_start, recs, own, own_cnt, V, SC); break;
    case 512: launch_bin_scatter_t<512>(s, P, T, G, pts, n, stride, hist, tile_start, recs, own, own_cnt, V, SC); break;
    default: launch_bin_scatter_t<256>(s, P, T, G, pts, n, stride, hist, tile_start, recs, own, own_cnt, V, SC);
  }
}
void launch_tile_count(hipStream_t s, const KP& P, const BinGeo& G, const BinRec* recs, int rs, const unsigned int* tile_start, Cells cells,
                       ErrSlot* slots, const SplitView& SV, long n) {
  static_assert(TF_BLOCK == BIN_TR * BIN_TC, "one thread per cell of a tile");
  (void)n;
  const bool split = SV.on && SV.cap > 0;
  const dim3 g((split ? (unsigned int)SV.cap * G.sub : 0u) + tile_grid(G)), b(TF_BLOCK);
  if (split) { if (rs == 2) hipLaunchKernelGGL((k_tile_count<true, 2>), g, b, 0, s, P, G, recs, tile_start, cells, slots, SV); else hipLaunchKernelGGL((k_tile_count<true, 1>), g, b, 0, s, P, G, recs, tile_start, cells, slots, SV); }
  else { if (rs == 2) hipLaunchKernelGGL((k_tile_count<false, 2>), g, b, 0, s, P, G, recs, tile_start, cells, slots, SV); else hipLaunchKernelGGL((k_tile_count<false, 1>), g, b, 0, s, P, G, recs, tile_start, cells, slots, SV); }
}
// fuse_average: commit + average in the tile kernel (whole frames); rays: the visibility pass follows (bitmap + inlier plane wanted)
// true when a launch with these arguments would run the semantic fusion inside the tile kernel (SEM): a carrying frame (rs == 2) that
// commits + averages itself, without a visibility pass and without heavy-tile parts in the launch
bool bin_fuse_takes_semantics(const SplitView& SV, bool fuse_average, bool rays, int rs) { return rs == 2 && fuse_average && !rays && !(SV.on && SV.cap > 0); }
// rs: record stride in 16-byte units; sm (may be null): the frame's semantic fusion, run inside the kernel when bin_fuse_takes_semantics()
void launch_bin_fuse(hipStream_t s, const KP& P, const BinGeo& G, const BinRec* recs, int rs, const unsigned int* tile_start, Cells cells,
                     AccF* acc, FrameDev* F, bool fuse_average, bool rays, unsigned int* cnt_plane, unsigned long long* inert,
                     unsigned int* inl_plane, float* thr, const OverlapArgs& O, const GateFold& GF, const SplitView& SV, long n, const SemMini* sm) {
  (void)n;
  const bool split = SV.on && SV.cap > 0;
  const dim3 g((split ? (unsigned int)SV.cap * G.sub : 0u) + tile_grid(G)), b(TF_BLOCK);
  SemMini SM; memset(&SM, 0, sizeof SM);
#define EM_FUSE_I(A, R, SP, RS_, SE) hipLaunchKernelGGL((k_tile_fuse<A, R, SP, RS_, SE>), g, b, 0, s, P, G, recs, tile_start, cells, acc, F, cnt_plane, inert, inl_plane, thr, O, GF, SV, SM)
#define EM_FUSE(A, R) do { if (split) EM_FUSE_I(A, R, true, 1, false); else EM_FUSE_I(A, R, false, 1, false); } while (0)
  if (rs == 2) {                                   // a carrying frame (emap_api.hip: only whole frames without a visibility pass carry)
    if (sm && bin_fuse_takes_semantics(SV, fuse_average, rays, rs)) { SM = *sm; EM_FUSE_I(true, false, false, 2, true); }
    else if (fuse_average && !rays) { if (split) EM_FUSE_I(true, false, true, 2, false); else EM_FUSE_I(true, false, false, 2, false); }
    else if (!fuse_average) { if (split) EM_FUSE_I(false, false, true, 2, false); else EM_FUSE_I(false, false, false, 2, false); }
    else abort();                                  // (32-byte records in front of a visibility pass: k_rays walks 16-byte records)
  }
  else if (fuse_average && rays) EM_FUSE(true, true);
  else if (fuse_average) EM_FUSE(true, false);
  else EM_FUSE(false, false);
#undef EM_FUSE
#undef EM_FUSE_I
}

// ---------------------------------------------------------------------------------------------------------
// Semantic / RGB point fusion on the tile-sorted records (reference custom_semantic_kernels.py:9-51,167-194,233-267,
// 270-375): the atomic path of emap_semantic.hip needs K + 4 global atomics per point (~44 us per million each); here one
// workgroup per tile accumulates its points in LDS (fp64 / uint32 LDS atomics), finalises the cells of the tile
// (average / class_average / packed colour) and writes the semantic planes directly -- no accumulator planes in HBM.
// Channels are processed in groups of 4 (32 KB of fp64 LDS accumulators per group).
// ---------------------------------------------------------------------------------------------------------
typedef SemSpec SemSpecB;
#define SEM_GROUP 4
#define SEM_BLK 1024     /* one wave per tile row: 32 waves per CU with two workgroups (256 threads left the CU at 12 waves: the kernel is a chain of short, latency-bound phases) */
// SPLIT (launches with extra workgroups in front, see tile_work): a heavy tile's parts each accumulate a channel group over their own
// records, add the partial sums (doubles -- sums of floats of one magnitude, exact in any grouping just as the LDS accumulation is -- and
// the colour sums) to the (tile, group) scratch of SemSplit and take that group's ticket; the part that arrives last reads the totals
// back and writes the group's planes.  Only for specs whose groups are independent of each other (no class_bayesian renormalisation
// over several layers, colour riding along or absent) and tiles whose slot lies below SemSplit::slots; everything else is reduced by
// the tile's own workgroup as before.
struct SemSplit { double* sum; unsigned int* col; unsigned int* tick; int slots, phases; };      // [slot][phase][4][cell], [slot][4][cell], [slot][phase]
// RS: record stride in 16-byte units; RS == 2 (the frame carried columns [cc0, cc0 + 4) of the cloud in its 32-byte records, emap_device.h:
// BinRec32): a group whose channels all lie in that window reads them from the record itself -- no gather
template <bool SPLIT, int RS>
__global__ __launch_bounds__(SEM_BLK) void k_tile_semantic(KP P, BinGeo G, SemSpecB S, const BinRec* __restrict__ recs,
                                                             const unsigned int* __restrict__ tile_start, ChanView V,
                                                             long n, const unsigned int* __restrict__ cnt_plane,
                                                             float* __restrict__ sem, float* __restrict__ alpha_planes, long plane,
                                                             SplitView SV, SemSplit X, int cc0) {
  constexpr int NC = BIN_TR * BIN_TC;
  const float* __restrict__ chp = V.p - V.col0;          // (a local restrict pointer: the gathers below must stay free to run ahead of the plane stores)
  const long chs = V.stride;
  __shared__ double s_sum[SEM_GROUP][NC];
  __shared__ unsigned int s_col[4][NC];                 // r, g, b, count of ONE colour layer at a time
  __shared__ bool s_fin;
  TileWork w;                                          // sb = which 16 x 64 tile of the bin
  if (!tile_work<SPLIT>(G, SV, tile_start, w)) return;
  if (SPLIT && w.np > 1u && w.slot >= (unsigned int)X.slots) {      // no scratch for this tile: its own workgroup takes all of its records
    if (w.part) return;
    w.np = 1u; w.r0 = tile_start[w.t]; w.r1 = tile_start[w.t + 1];
  }
  const bool split = SPLIT && w.np > 1u;               // (uniform)
  const int t = w.t, sb = w.sb;
  const int ty = t / G.tiles_x, tx = t - ty * G.tiles_x;
  const unsigned int r0 = w.r0, r1 = w.r1;
  const int tc = threadIdx.x & 63, wv = threadIdx.x >> 6, col = tx * BIN_TC + tc;
  const int row_base = (ty * G.sub + sb) * BIN_TR;
  if (row_base >= P.nrows) return;
  const unsigned int sel = (unsigned int)sb;
  // one colour channel next to averaged channels (the usual rgb + features cloud): its accumulation rides along the first
  // group's record loop, so every point row is gathered once
  const bool ride = S.n_col == 1 && S.n_sum > 0;
  for (int g0 = 0; g0 < S.n_sum; g0 += SEM_GROUP) {
    const int ng = min(SEM_GROUP, S.n_sum - g0);
    for (int k = threadIdx.x; k < SEM_GROUP * NC; k += SEM_BLK) (&s_sum[0][0])[k] = 0.0;
    if (ride && g0 == 0) for (int k = threadIdx.x; k < 4 * NC; k += SEM_BLK) (&s_col[0][0])[k] = 0u;
    __syncthreads();
    // The channel values of a point are gathered by point index from the cloud (rows of `stride` floats: a random row per record).
    // When the group's channels (and the colour channel riding along) lie within four consecutive columns -- the usual x y z rgb f1
    // f2 f3 cloud -- ONE 16-byte load fetches them all instead of one dword load per channel (each a request of its own to the same
    // one or two cache lines: the gathers were most of this kernel's time at 16 M points).
    int cmin = SEM_MAX_CH + 4096, cmax = 0;
    for (int q = 0; q < ng; ++q) { cmin = min(cmin, S.sum_chan[g0 + q]); cmax = max(cmax, S.sum_chan[g0 + q]); }
    if (ride && g0 == 0) { cmin = min(cmin, S.col_chan[0]); cmax = max(cmax, S.col_chan[0]); }
    const bool inrec = RS == 2 && cc0 >= 0 && cmin >= cc0 && cmax < cc0 + 4;                       // (uniform) the group's channels travelled in the record
    if (inrec) cmin = cc0;
    const bool wide = inrec || (cmax - cmin < 4 && cmin >= V.col0 && cmin - V.col0 + 4 <= V.stride);          // (uniform) the 16 bytes stay inside the point's row
    struct __attribute__((packed, aligned(4))) P4 { float a, b, c, d; };
    auto pick = [](const P4& w, int j) { return j == 0 ? w.a : (j == 1 ? w.b : (j == 2 ? w.c : w.d)); };
    for (unsigned int k = r0 + threadIdx.x; k < r1; k += SEM_BLK) {
      const BinRec r = recs[(size_t)k * RS];
      if (((r.lc_inl & 0x7fffffffu) >> 10) != sel) continue;
      const unsigned int lc = r.lc_inl & 1023u;
      const float* __restrict__ p = chp + (long)r.i * chs;
      P4 w4 = {0.f, 0.f, 0.f, 0.f};
      if (inrec) { const float4 q4 = reinterpret_cast<const float4*>(recs)[(size_t)k * RS + 1]; w4.a = q4.x; w4.b = q4.y; w4.c = q4.z; w4.d = q4.w; }
      else if (wide) w4 = *reinterpret_cast<const P4*>(p + cmin);
      for (int q = 0; q < ng; ++q) {
        const float v = wide ? pick(w4, S.sum_chan[g0 + q] - cmin) : p[S.sum_chan[g0 + q]];
        const int kind = S.sum_kind[g0 + q];
        if (kind >= 2) {                                                              // compact kernels: id * K + q < N, theta >= 0
          if ((long)r.i * S.sum_K[g0 + q] + S.sum_q[g0 + q] >= n) continue;
          if (kind == 2 && !(v >= 0.0f)) continue;
        }
        unsafeAtomicAdd(&s_sum[q][lc], (double)v);
      }
      if (ride && g0 == 0) {
        const unsigned int color = __float_as_uint(wide ? pick(w4, S.col_chan[0] - cmin) : p[S.col_chan[0]]);
        atomicAdd(&s_col[0][lc], (color & 0xFF0000u) >> 16);
        atomicAdd(&s_col[1][lc], (color & 0xFF00u) >> 8);
        atomicAdd(&s_col[2][lc], color & 0xFFu);
        atomicAdd(&s_col[3][lc], 1u);
      }
    }
    __syncthreads();
    bool fin = true;                                          // this workgroup writes the group's planes
    if (split) {
      static_assert(SEM_BLK == NC && NC == (int)SPLIT_CELLS, "one thread per cell of a slot");
      const int ph = g0 / SEM_GROUP;
      double* xs = X.sum + ((size_t)w.slot * X.phases + ph) * SEM_GROUP * NC;
      unsigned int* xc = X.col + (size_t)w.slot * 4 * NC;
      for (int q = 0; q < ng; ++q) { const double v = s_sum[q][threadIdx.x]; if (v != 0.0) unsafeAtomicAdd(&xs[q * NC + threadIdx.x], v); }
      if (ride && g0 == 0 && s_col[3][threadIdx.x])
        for (int q = 0; q < 4; ++q) atomicAdd(&xc[q * NC + threadIdx.x], s_col[q][threadIdx.x]);
      __builtin_amdgcn_s_waitcnt(0);                          // (the hand-off of k_tile_fuse's parts: acknowledgements, barrier, ticket)
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned int* tk = X.tick + (size_t)w.slot * X.phases + ph;
        s_fin = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == w.np - 1u;
        if (s_fin) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      fin = s_fin;
      if (fin) {                                              // totals back into LDS, scratch left zeroed for the next frame
        for (int q = 0; q < ng; ++q) {
          const double v = __hip_atomic_load(&xs[q * NC + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_sum[q][threadIdx.x] = v;
          if (v != 0.0) __hip_atomic_store(&xs[q * NC + threadIdx.x], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (ride && g0 == 0)
          for (int q = 0; q < 4; ++q) {
            const unsigned int v = __hip_atomic_load(&xc[q * NC + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_col[q][threadIdx.x] = v;
            if (v) __hip_atomic_store(&xc[q * NC + threadIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
      }
      __syncthreads();
    }
    if (fin && col < P.C) {
      for (int k = 0; k < BIN_TR / (SEM_BLK / 64); ++k) {
        const int tr = wv + (SEM_BLK / 64) * k, lrow = row_base + tr;
        if (lrow >= P.nrows) break;
        const long c = (long)(lrow + P.halo) * P.C + col;
        const unsigned int cnt = cnt_plane[c];               // accepted HEIGHT points (new_elmap plane 2, :185)
        if (ride && g0 == 0) {
          const int lc = tr * BIN_TC + tc;
          const unsigned int cn = s_col[3][lc];
          if (cn) {
            const unsigned int rr = s_col[0][lc] / cn, gg = s_col[1][lc] / cn, bb = s_col[2][lc] / cn;
            sem[(long)S.col_layer[0] * plane + c] = __uint_as_float((rr << 16) + (gg << 8) + bb);
          }
        }
        for (int q = 0; q < ng; ++q) {
          const long j = (long)S.sum_layer[g0 + q] * plane + c;
          const double s = s_sum[q][tr * BIN_TC + tc];
          const int kind = S.sum_kind[g0 + q];
          if (kind == 2) alpha_planes[j] = (float)((double)alpha_planes[j] + s);   // every cell; renormalised below
          else if (kind == 3) {
            const long gcell = (long)logi_row(P, P.row0 + lrow) * P.C + logi_col(P, col);      // the reference's flat LOGICAL cell index
            if (cnt > 0 && gcell * S.sum_K[g0 + q] + S.sum_q[g0 + q] < (long)P.C * P.C) {
              const float cn = (float)cnt, feat_ml = (float)s / cn, sigma_old = 0.0f, sigma = 1.0f;
              sem[j] = sigma * sem[j] / (cn * sigma_old + sigma) + cn * sigma_old * feat_ml / (cn * sigma_old + sigma);
            }
          } else if (cnt == 0) continue;
          else if (kind == 0) sem[j] = (float)(s / (double)cnt);
          else {
            const float prev = sem[j];
            sem[j] = (prev == 0.0f) ? (float)(s / (double)cnt) : (float)(S.alpha * (double)prev + (1.0 - S.alpha) * s / (double)cnt);
          }
        }
      }
    }
    __syncthreads();
  }
  if (S.any_bayes && col < P.C) {     // class_bayesian: theta = alpha / sum(alpha) over its layers, same thread <-> same cells as above
    for (int k = 0; k < BIN_TR / (SEM_BLK / 64); ++k) {
      const int tr = wv + (SEM_BLK / 64) * k, lrow = row_base + tr;
      if (lrow >= P.nrows) break;
      const long c = (long)(lrow + P.halo) * P.C + col;
      float tot = 0.0f;
      for (int q = 0; q < S.n_sum; ++q) if (S.sum_kind[q] == 2) tot += alpha_planes[(long)S.sum_layer[q] * plane + c];
      if (tot == 0.0f) tot = 1.0f;
      for (int q = 0; q < S.n_sum; ++q) if (S.sum_kind[q] == 2) { const long j = (long)S.sum_layer[q] * plane + c; sem[j] = alpha_planes[j] / tot; }
    }
  }
  if (S.n_col > 0 && !ride) {
    const int K = S.n_col;
    // the reference's launch-size quirk (fusion/pointcloud_color.py:143): element e = id * K + layer only exists for e < N,
    // and ONE counter plane is shared by all K layers
    for (int k = threadIdx.x; k < NC; k += SEM_BLK) s_col[3][k] = 0u;
    __syncthreads();
    for (unsigned int k = r0 + threadIdx.x; k < r1; k += SEM_BLK) {
      const BinRec r = recs[(size_t)k * RS];
      if (((r.lc_inl & 0x7fffffffu) >> 10) != sel) continue;
      const unsigned int lc = r.lc_inl & 1023u;
      for (int l = 0; l < K; ++l) if ((long)r.i * K + l < n) atomicAdd(&s_col[3][lc], 1u);
    }
    __syncthreads();
    for (int l = 0; l < K; ++l) {
      for (int k = threadIdx.x; k < 3 * NC; k += SEM_BLK) (&s_col[0][0])[k] = 0u;
      __syncthreads();
      for (unsigned int k = r0 + threadIdx.x; k < r1; k += SEM_BLK) {
        const BinRec r = recs[(size_t)k * RS];
        if ((long)r.i * K + l >= n || ((r.lc_inl & 0x7fffffffu) >> 10) != sel) continue;
        const unsigned int lc = r.lc_inl & 1023u;
        const unsigned int color = __float_as_uint(chp[(long)r.i * chs + S.col_chan[l]]);
        atomicAdd(&s_col[0][lc], (color & 0xFF0000u) >> 16);
        atomicAdd(&s_col[1][lc], (color & 0xFF00u) >> 8);
        atomicAdd(&s_col[2][lc], color & 0xFFu);
      }
      __syncthreads();
      if (col < P.C) {
        for (int k = 0; k < BIN_TR / (SEM_BLK / 64); ++k) {
          const int tr = wv + (SEM_BLK / 64) * k, lrow = row_base + tr, lc = tr * BIN_TC + tc;
          if (lrow >= P.nrows) break;
          const unsigned int cn = s_col[3][lc];
          if (cn == 0) continue;
          const unsigned int rr = s_col[0][lc] / cn, gg = s_col[1][lc] / cn, bb = s_col[2][lc] / cn;
          sem[(long)S.col_layer[l] * plane + (long)(lrow + P.halo) * P.C + col] = __uint_as_float((rr << 16) + (gg << 8) + bb);
        }
      }
      __syncthreads();
    }
  }
}
// scratch of the split semantic kernel: bytes for `slots` tiles (0: cannot split this spec)
size_t sem_split_bytes(int slots) { return (size_t)slots * ((size_t)(SEM_MAX_CH / SEM_GROUP) * SEM_GROUP * SPLIT_CELLS * 8 + 4 * SPLIT_CELLS * 4 + (SEM_MAX_CH / SEM_GROUP) * 4); }
bool sem_split_possible(const SemSpec& S) { return S.n_sum > 0 && !S.any_bayes && (S.n_col == 0 || S.n_col == 1); }     // independent groups; colour riding along or absent
void launch_tile_semantic(hipStream_t s, const KP& P, const BinGeo& G, const SemSpec& S, const BinRec* recs, int rs, int cc0, const unsigned int* tile_start,
                          const ChanView& V, long n, const unsigned int* cnt_plane, float* sem, float* alpha_planes, long plane,
                          const SplitView& SV, void* split_mem, int split_slots) {
  SemSplit X = {nullptr, nullptr, nullptr, 0, SEM_MAX_CH / SEM_GROUP};
  const bool split = SV.on && SV.cap > 0 && split_mem && split_slots > 0 && sem_split_possible(S);
  if (split) {
    X.slots = split_slots;
    X.sum = reinterpret_cast<double*>(split_mem);
    X.col = reinterpret_cast<unsigned int*>(X.sum + (size_t)split_slots * X.phases * SEM_GROUP * SPLIT_CELLS);
    X.tick = X.col + (size_t)split_slots * 4 * SPLIT_CELLS;
    const dim3 g((unsigned int)SV.cap * G.sub + tile_grid(G));
    if (rs == 2) hipLaunchKernelGGL((k_tile_semantic<true, 2>), g, dim3(SEM_BLK), 0, s, P, G, S, recs, tile_start, V, n, cnt_plane, sem, alpha_planes, plane, SV, X, cc0);
    else hipLaunchKernelGGL((k_tile_semantic<true, 1>), g, dim3(SEM_BLK), 0, s, P, G, S, recs, tile_start, V, n, cnt_plane, sem, alpha_planes, plane, SV, X, cc0);
  } else if (rs == 2) hipLaunchKernelGGL((k_tile_semantic<false, 2>), dim3(tile_grid(G)), dim3(SEM_BLK), 0, s, P, G, S, recs, tile_start, V, n, cnt_plane, sem, alpha_planes, plane, SV, X, cc0);
  else hipLaunchKernelGGL((k_tile_semantic<false, 1>), dim3(tile_grid(G)), dim3(SEM_BLK), 0, s, P, G, S, recs, tile_start, V, n, cnt_plane, sem, alpha_planes, plane, SV, X, cc0);
}
