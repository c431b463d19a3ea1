// C ABI of the MI355X elevation-map fusion core (declared in include/emap_hip.h).
// Host-side orchestration only: owns device memory, turns emap_params into kernargs, enqueues the kernels of
// emap_kernels.hip on ONE stream in the order of the reference's update_map_with_kernel
// (EM/elevation_mapping.py:316-391).  No per-frame allocation, no D2H sync inside a frame unless stats are asked for.
#include "emap_device.h"
#include <cstddef>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library itself is dlopen()ed by emap_comm_init
#include "../../include/emap_hip.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

// launchers (emap_kernels.hip)
bool launch_count(hipStream_t, const KP&, const Pose&, const float*, long, int, Cells, AccF*, ErrSlot*, const GateArgs*, FrameDev*, unsigned int*);
void launch_gate(hipStream_t, const GateArgs&, ErrSlot*, FrameDev*, int, double*, const double*);
int small_frame_grid(const KP&, long);
void launch_small_frame(hipStream_t, int, const KP&, const Pose&, const float*, long, int, Cells, AccF*, unsigned int*, const OverlapArgs&,
                        const GateArgs&, FrameDev*, FrameDev*, ErrSlot*, unsigned int*, unsigned int*, unsigned int*, unsigned int*, unsigned int, unsigned int, int);
void launch_fuse(hipStream_t, const KP&, const Pose&, const float*, long, int, Cells, AccF*, const FrameDev*);
void launch_commit(hipStream_t, const KP&, Cells, const AccF*, const FrameDev*, unsigned long long*);
void launch_rays(hipStream_t, const KP&, const Pose&, const RayTab&, const float*, long, int, Cells, const AccRView&, const float*, long, FrameDev*, bool, const unsigned long long*, const unsigned int*, int, const float*, const unsigned int*, const unsigned int*);
void launch_win_pack(hipStream_t, const KP&, const Win&, Cells, const float*, long, const unsigned int*, const unsigned long long*, float);
void launch_win_prepare(hipStream_t, const Win&, int);
void launch_win_unpack(hipStream_t, const KP&, const Win&, AccR*);
void launch_win_reduce(hipStream_t, long long*, unsigned int*, const long long*, const unsigned int*, int, long, long, long);
void launch_ray_apply(hipStream_t, const KP&, Cells, AccR*, unsigned long long*, const OverlapArgs&, FrameDev*, unsigned int*, int);
void launch_average(hipStream_t, const KP&, Cells, AccF*, AccR*, const FrameDev*, bool, bool, unsigned int*, const OverlapArgs&);
static_assert(offsetof(SemSpec, sum_K) == sizeof(emap_sem_spec), "emap_sem_spec is the leading part of SemSpec");
void launch_sem_points(hipStream_t, const KP&, const Pose&, const SemSpec&, const float*, long, int, const ChanView&, double*, unsigned int*, long);
void launch_sem_finalize(hipStream_t, const KP&, const SemSpec&, const unsigned int*, double*, unsigned int*, float*, float*, long);
struct SemRaw { int op, stride, K, n_max; long size, cells; double alpha; };
void launch_semraw_acc(hipStream_t, const SemRaw&, const float*, const int*, const int*, const float*, const int*, float*, unsigned int*);
void launch_semraw_fin(hipStream_t, const SemRaw&, float*, const unsigned int*, const int*, const float*, const float*, float*);
void launch_polygon_mask(hipStream_t, int, const int*, const int*, int, const int*, float*);
void launch_dilate_planes(hipStream_t, int, int, const float*, const float*, float*, float*);
struct CamArgs { float P[12], K[9], D[5], center[3]; float x1, y1, z1, ih, iw; double tol; };
struct CmaxSpec { int n; int chan[8]; int layer[8]; };
void launch_cmax_ids(hipStream_t, const KP&, const CmaxSpec&, const ChanView&, long, const float*, long, unsigned char*, unsigned char*);
void launch_cmax_sum(hipStream_t, const KP&, const Pose&, const CmaxSpec&, const float*, long, int, const ChanView&, const int*, long long*, long);
void launch_cmax_select(hipStream_t, const KP&, const CmaxSpec&, int, const long long*, long, unsigned char*, unsigned char*, const unsigned int*, float*, float*, float*);
void launch_image_corr(hipStream_t, const KP&, const CamArgs&, Cells, float*, unsigned char*);
void launch_image_fuse(hipStream_t, const KP&, int, float*, const float*, const float*, const unsigned char*, float, float, double);
void launch_inpaint_sweep(hipStream_t, int, const float*, const float*, float*, float*, const unsigned int*, unsigned int*);
void launch_min_sweep(hipStream_t, int, int, const float*, const float*, const float*, float*, float*, const unsigned int*, unsigned int*, bool);
void launch_box3(hipStream_t, int, const float*, float*);
void launch_erode(hipStream_t, int, int, const float*, float*);
void launch_overlap(hipStream_t, const KP&, Cells, int, int, float, float);
void launch_var_time(hipStream_t, const KP&, Cells, int, int);
int post_tile_rows(const KP&);
void launch_post(hipStream_t, const KP&, const float*, const float*, const float*, const float*, Cells, float*, float*, long, int, int, const int*, const int*, int, int);
void launch_get_plane(hipStream_t, const KP&, Cells, int, float*);
void launch_publish(hipStream_t, const KP&, Cells, const float*, long, int, float, int, float*);
void launch_set_plane(hipStream_t, const KP&, Cells, int, const float*);
void launch_fill_cells(hipStream_t, Cells, long, const Cell&);
void launch_point_index(hipStream_t, const KP&, const Pose&, const float*, long, int, int*, unsigned char*);
void launch_plane_view(hipStream_t, const KP&, int, int, float*, float*, int);
void launch_materialize(hipStream_t, const KP&, Cells);
void launch_band_clear(hipStream_t, const KP&, float*, int, long, int, int);

// tile-binned scatter (emap_binned.hip)
void launch_bin_hist(hipStream_t, const KP&, const Pose&, const BinGeo&, const float*, long, int, unsigned int*, BinStg*, unsigned int*);
void launch_bin_scan(hipStream_t, const BinGeo&, unsigned int*, unsigned int*, unsigned int*, unsigned int*, const SplitView&);
void launch_bin_scatter(hipStream_t, const KP&, const Pose&, const BinGeo&, const float*, long, int, const unsigned int*, const unsigned int*, BinRec*, const BinStg*, const unsigned int*, const ChanView&, const SemCarry&);
void launch_tile_count(hipStream_t, const KP&, const BinGeo&, const BinRec*, int, const unsigned int*, Cells, ErrSlot*, const SplitView&, long);
void launch_tile_semantic(hipStream_t, const KP&, const BinGeo&, const SemSpec&, const BinRec*, int, int, const unsigned int*, const ChanView&, long,
                          const unsigned int*, float*, float*, long, const SplitView&, void*, int);
size_t sem_split_bytes(int);
bool sem_split_possible(const SemSpec&);
#define SEM_SPLIT_SLOTS 128      /* heavy tiles whose semantic sums several workgroups may share (19 MB of scratch) */
void launch_bin_fuse(hipStream_t, const KP&, const BinGeo&, const BinRec*, int, const unsigned int*, Cells, AccF*, FrameDev*, bool, bool, unsigned int*, unsigned long long*, unsigned int*, float*, const OverlapArgs&, const GateFold&, const SplitView&, long, const SemMini*);
bool bin_fuse_takes_semantics(const SplitView&, bool, bool, int);
#define BIN_MAX_T 16384
#ifndef EMAP_SPLIT_POOL_DEFAULT
#define EMAP_SPLIT_POOL_DEFAULT 0       /* standing pool of extra tile workgroups (emap_count): off -- measured, see there */
#endif
#define BIN_MAX_B 2048

// timed stages of emap_update (emap_get_stage_times): hist+scan are 0 on the atomic path, where "scatter" is k_count
enum { ST_HIST = 0, ST_SCAN, ST_SCATTER, ST_GATE, ST_FUSE, ST_COMMIT, ST_RAYS, ST_AVERAGE, ST_OVERLAP, ST_POST, ST_N };

// the ten RCCL entry points of the path, bound with dlsym (no link-time dependency: the .so loads on machines without RCCL)
struct RcclApi {
  void* handle;
  decltype(&ncclGetUniqueId) GetUniqueId; decltype(&ncclCommInitRank) CommInitRank; decltype(&ncclCommDestroy) CommDestroy;
  decltype(&ncclAllReduce) AllReduce; decltype(&ncclSend) Send; decltype(&ncclRecv) Recv;
  decltype(&ncclGroupStart) GroupStart; decltype(&ncclGroupEnd) GroupEnd; decltype(&ncclGetErrorString) GetErrorString;
  decltype(&ncclCommCount) CommCount;
};

// ---- asynchronous cloud upload (a1: ElevationMap.input_pointcloud, EM/elevation_mapping.py:456-458) ------------------------------
// The ROS wrapper hands over a pageable float64 matrix.  It is converted to float32 on the HOST by a few worker threads straight
// into a pinned slot (so only 12 of the 24 bytes per point cross PCIe and no device-side cast pass is needed), chunk by chunk, each
// chunk's DMA on a copy stream overlapping the next chunk's conversion; the device buffer is double buffered so the upload of frame
// k+1 overlaps the kernels of frame k.  The call returns when the caller's buffer has been consumed (it is only borrowed).
struct Workers {
  std::vector<std::thread> th; std::mutex m; std::condition_variable cv, done_cv;
  std::function<void(int, int)> job; int gen = 0, pending = 0; bool stop = false;
  explicit Workers(int n) {
    for (int i = 0; i < n; ++i) th.emplace_back([this, i, n] {
      int seen = 0;
      for (;;) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return stop || gen != seen; });
        if (stop) return;
        seen = gen; auto f = job; l.unlock();
        f(i, n);
        l.lock(); if (--pending == 0) done_cv.notify_all();
      }
    });
  }
  void run(const std::function<void(int, int)>& f) {       // f(worker, n_workers) on every worker; returns when all are done
    std::unique_lock<std::mutex> l(m);
    job = f; pending = (int)th.size(); ++gen; cv.notify_all();
    done_cv.wait(l, [&] { return pending == 0; });
  }
  ~Workers() { { std::lock_guard<std::mutex> l(m); stop = true; } cv.notify_all(); for (auto& t : th) t.join(); }
};

struct emap_ctx {
  emap_params prm;
  emap_strip strip;
  KP kp;
  int device;
  hipStream_t stream;
  bool own_stream;
  long ncells_alloc;        // (rows + 2*halo) * C
  Cells cells;                     // two planes of 16-byte half cells (emap_device.h)
  int torg_r, torg_c;              // origin traversability_input was written with (kp.norg_*: the normal planes)
  AccF* acc; AccR* accr;
  float* trav_in; float* normal;   // normal: 3 planes of ncells_alloc
  float* scratch;                  // one plane (get/set staging)
  float* plug_buf; size_t plug_cap; unsigned int* plug_cnt; int plug_cnt_cap;   // plane scratch of the publish-time plugins (kept between calls)
  ErrSlot* slots; FrameDev* frame;
  RayTab rt; float* ray_S; unsigned short* ray_lut;
  unsigned long long* inert;       // 1 bit per owned cell (rows of ceil(C/64) words), written by k_commit / k_tile_fuse<true, true>
  unsigned int* inl_plane;         // newmap[3] of frames whose tile kernel commits itself (binned path + visibility pass), on demand
  float* ray_thr;                  // same frames: per 8 x 8 block height at or above which a ray sample cannot affect any cell of the block
  bool inert_zero;                 // the bitmap is all zero (k_ray_apply leaves it so; k_commit overwrites it)
  bool rays_fused;                 // this frame: the tile kernel committed + averaged, k_ray_apply follows the rays
  // tile-binned scatter buffers (allocated on demand)
  int scatter_mode;                // 0 auto, 1 atomic, 2 binned
  int force_sub;                   // test hook: minimum bin height factor (emap_set_scatter_mode bits 8..15)
  bool frame_binned;               // the count stage of the current frame used the binned path
  unsigned int* bin_sync;          // ticket counters of k_bin_scan (last_block_ticket), zero between launches
  SplitView split;                 // heavy tiles reduced by several workgroups (emap_device.h); split.on: this frame's scan listed them
  void* split_mem;                 // one allocation behind split's arrays
  void* sem_split_mem;             // scratch of the split semantic tile kernel (SemSplit), zero between launches
  volatile unsigned int* split_need;   // host-mapped word: the parts the last scan the device has finished would have listed
  bool split_dirty;                // k_tile_count has filled slots that no k_tile_fuse has cleared yet
  GateFold gate_fold;              // multi-GPU frames: gate decision on the all-reduced totals folded into the tile kernel (mode 0: k_gate ran)
  OverlapArgs ov_args;             // clear_overlap_map folded into the frame's rewriting kernels (on = 0: separate k_overlap launch)
  bool fold_gate, gate_folded;     // emap_update on the atomic path: the gate rides in k_count's last workgroup (cnt_sync: its ticket words)
  unsigned int* cnt_sync;
  // robot scale: count -> gate -> fuse -> commit + average in one launch (k_small_frame): its barriers' release words live behind the
  // ticket words of cnt_sync.  A launch whose grid barrier is ABORTED leaves map, accumulators and drift record exactly as it found
  // them, and the launches queued behind it do nothing (device-side poison word).  sf_host = two host-mapped words: [0] epoch of the
  // last launch that will be applied, [1] epoch of the first aborted launch.  The frames issued and not yet known to be applied wait
  // in sf_ring with everything needed to run them again; sf_settle (every entry point but the ones that only bind a cloud) learns
  // their fate and, after an abort, re-runs them in order on the chain of launches.
  unsigned int sf_epoch; volatile unsigned int* sf_host; unsigned int* sf_host_dev; unsigned int* sf_poison; bool sf_off;
  FrameDev* frame_save;            // the frame record as the gate of the last k_small_frame found it
  struct SfFrame { unsigned int epoch; float R[9], t[3]; double pn, on; Moves mv; const float* pts; long n_pts, n_pts_all; int stride; ChanView chan; int n_cols; };
  enum { SF_RING = 8 };
  SfFrame sf_ring[SF_RING]; int sf_head, sf_count;
  bool sf_redo;                    // inside sf_recover: frames take the chain of launches
  unsigned int sf_aborts;          // frames re-run so far (emap_small_frame_aborts)
  int update_path;                 // emap_last_update_path
  bool gate_possible;              // false inside emap_update when the host already knows that the drift gate cannot fire: the per-tile
                                   // error statistics are then skipped (they could not have any effect; err_sum / err_cnt report 0)
  BinGeo bg; BinRec* bin_recs; BinStg* bin_own; unsigned int* bin_own_cnt; long bin_own_cap; bool bin_strip;   // bin_own*: staged records of the owned points per block (strip contexts without a visibility pass)
  unsigned int* bin_hist; unsigned int* bin_tile_total; unsigned int* bin_tile_start; long bin_cap; size_t bin_hist_cap;      // bin_cap: 16-byte units
  // The semantic fusion declared for the NEXT whole frame (emap_frame_semantics): run inside the frame.  A frame that can CARRIES the
  // channels in 32-byte sorted records (bin_rs = 2, carry: which columns) and fuses them in the tile kernel itself (fsem_merged);
  // every other frame runs the stand-alone semantic kernels before it returns -- the result is the same either way.
  bool fsem_set, fsem_merged; int fsem_keep_counts; SemSpec fsem;
  bool carry_want;                 // this frame's emap_count may sort 32-byte records (decided by frame_sem_begin)
  int bin_rs; SemCarry carry;      // stride of the current frame's sorted records in 16-byte units; the carried columns (on = 0: none)
  // semantic layers (planar float planes + double / uint32 accumulators), allocated on demand
  float* img_uv; unsigned char* img_valid; float* img_buf; size_t img_cap;   // camera path
  double img_tol; bool img_tol_set;   // tolerance_z_collision of the occlusion walk (0.10 unless emap_image_set_tolerance was called)
  float* sem_alpha;   // class_bayesian pseudo-counts (the reference's persistent new_map layers), sem_layers planes, on demand
  int sem_layers; float* sem; double* sem_sums; unsigned int* sem_col; unsigned int* cnt_plane;
  // point cloud
  float* pts_dev[2]; long pts_cap[2];      // owned device buffers (floats), ping-pong between consecutive uploads
  float* pts_pin[2]; long pin_cap[2];      // pinned host slots the clouds are converted / copied into
  hipEvent_t ev_copied[2], ev_used[2];     // DMA of slot done (copy stream) / the frame that read the slot's device buffer done (main stream)
  int up_slot; bool up_used[2]; hipStream_t copy_stream; Workers* workers;
  const float* pts; long n_pts; int stride;         // xyz of the bound cloud: rows of `stride` floats (3 for a de-interleaved cloud)
  long n_pts_all;                                   // size of the cloud the caller handed over (> n_pts for a bucketed one): what every rank of a sharded map shares
  bool pts_bucketed; float bucket_R[9], bucket_t[3]; int bucket_org_r;   // the bound cloud only holds the points that can land in this strip's rows under this pose (emap_upload_points_strip)
  ChanView chan; int n_cols;                        // its extra channels (emap_device.h: ChanView); n_cols = columns of the caller's matrix (3 + K)
  int* tail_idx; unsigned char* tail_flags; long tail_cap;
  // frame state
  double pos_noise, ori_noise; bool use_override; double sum_override; unsigned int cnt_override;
  bool committed;
  bool stage_timing; hipEvent_t ev[ST_N + 1]; float stage_ms[ST_N];
  hipEvent_t t0, t1;
  bool want_ray_stats; bool in_update;
  // row-strip communicator (emap_comm_init): RCCL resolved at run time, exchange on its own stream so that it overlaps the interior stencils
  struct RcclApi* rccl; ncclComm_t comm; int comm_rank, comm_world;
  float* gather_buf;            // cell_n x cell_n plane of emap_comm_gather_layer (on demand)
  // rays by ray (multi-GPU frames with a visibility pass): the replicated ray window around the sensor (emap_device.h: Win)
  int ray_mode;                 // 0 auto (by ray from 2048^2 cells on), 1 always by row, 2 by ray whenever the frame allows it
  bool byray_frame;             // the current sharded frame marches its rays by ray
  size_t wire_bytes;            // payload of the last by-ray frame's three all-reduces (bytes per rank)
  int ray_par;                  // parity of the k_ray_apply launches (FrameDev::quiet_sum)
  unsigned int* win_state; unsigned int* win_rec; unsigned long long* win_bits; float* win_thr; long long* win_dh; unsigned int* win_key; long win_cap;
  long long* win_red_dh; unsigned int* win_red_key;      // by-ray effects reduced to the owners: (world - 1) parts of win_cap cells each
  hipStream_t comm_stream; hipEvent_t ev_ready, ev_done; double* comm_sums;   // [0..1] local err_sum / err_cnt, [2..3] totals, [4..36) emap_comm_allreduce_host
  // the un-shifted normal planes after a row shift (normal_exchange): a row-aligned copy of the rows this strip's cells belong to
  std::vector<int> cut_begin, cut_count;   // every rank's owned PHYSICAL rows (gathered by emap_comm_init)
  float* nlag_buf; long nlag_cap;          // 3 planes of row_count x cell_n
  bool nlag_ready;                         // filled for the current frame's visibility pass
  bool cuts_ok;                            // the gathered strips tile the map
  std::string err;
};

#define CK(call)                                                                                         \
  do { hipError_t e_ = (call);                                                                           \
       if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return EMAP_ERR_HIP; } } while (0)
#define CKARG(cond, msg) do { if (!(cond)) { if (ctx) ctx->err = msg; return EMAP_ERR_INVALID; } } while (0)

static float q16(float x) { return (float)(_Float16)x; }
// The k_small_frame launches that are issued and not yet known to be applied (emap_ctx::sf_ring).  sf_settle waits until the device
// has decided the fate of the last of them (a poll of a host-mapped word: the decision falls at the launch's second barrier, long
// before the stream is idle), forgets those that will be applied, and after an ABORT (the launch left everything as it found it, the
// launches behind it did nothing: emap_kernels.hip) drains the stream and re-runs the aborted frame and every one after it, in order,
// on the chain of launches, each on the cloud that was bound when it was issued.  Every entry point that reads or changes the map, the
// drift record or a cloud buffer passes through SF_CHECK first; binding another device cloud and emap_update's own small-frame path do
// not (frames pipeline), they only make room in the ring.
static int update_impl(emap_ctx* ctx, const float R[9], const float t[3], double position_noise, double orientation_noise, emap_stats* stats);
static void sf_forget_applied(emap_ctx* ctx) {
  const unsigned int passed = ctx->sf_host[0];
  while (ctx->sf_count > 0 && ctx->sf_ring[ctx->sf_head].epoch <= passed) { ctx->sf_head = (ctx->sf_head + 1) % emap_ctx::SF_RING; --ctx->sf_count; }
}
static int sf_recover(emap_ctx* ctx) {
  CK(hipSetDevice(ctx->device));
  CK(hipStreamSynchronize(ctx->stream));      // the aborted launch, the idle ones behind it and their stencil launches have drained
  const unsigned int first = ctx->sf_host[1];
  ctx->sf_host[1] = 0u;
  CK(hipMemset(ctx->sf_poison, 0, 4));
  // the binding of the moment (maybe a cloud for a frame that has not been issued yet) comes back afterwards
  const float* pts = ctx->pts; const long n_pts = ctx->n_pts, n_all = ctx->n_pts_all; const int stride = ctx->stride, n_cols = ctx->n_cols; const ChanView chan = ctx->chan;
  const bool fsem_set = ctx->fsem_set; ctx->fsem_set = false;      // (a fusion declared for the frame to come is not the re-run frames')
  int rc = EMAP_OK;
  ctx->sf_redo = true;
  bool first_done = false;
  while (ctx->sf_count > 0) {
    const emap_ctx::SfFrame f = ctx->sf_ring[ctx->sf_head];
    ctx->sf_head = (ctx->sf_head + 1) % emap_ctx::SF_RING; --ctx->sf_count;
    if (f.epoch < first) continue;            // applied before the abort
    if (!first_done) { ctx->kp.mv = f.mv; first_done = true; }      // the pending map shifts the aborted launch was to write out (later frames: none, no shift can lie between unsettled frames)
    ctx->pts = f.pts; ctx->n_pts = f.n_pts; ctx->n_pts_all = f.n_pts_all; ctx->stride = f.stride; ctx->chan = f.chan; ctx->n_cols = f.n_cols;
    ++ctx->sf_aborts;
    if (rc == EMAP_OK) rc = update_impl(ctx, f.R, f.t, f.pn, f.on, nullptr);
  }
  ctx->sf_redo = false; ctx->fsem_set = fsem_set;
  ctx->pts = pts; ctx->n_pts = n_pts; ctx->n_pts_all = n_all; ctx->stride = stride; ctx->chan = chan; ctx->n_cols = n_cols;
  return rc;
}
static int sf_settle(emap_ctx* ctx) {
  if (ctx->sf_redo || ctx->sf_count == 0) return EMAP_OK;
  const unsigned int last = ctx->sf_ring[(ctx->sf_head + ctx->sf_count - 1) % emap_ctx::SF_RING].epoch;
  for (long spins = 0; ctx->sf_host[1] == 0u && ctx->sf_host[0] < last; ++spins) {
    if (spins > (1L << 22)) {                 // (something else is very slow on this stream: wait for it the ordinary way)
      CK(hipSetDevice(ctx->device)); CK(hipStreamSynchronize(ctx->stream));
      if (ctx->sf_host[1] == 0u && ctx->sf_host[0] < last) { ctx->err = "k_small_frame finished without reporting its outcome"; return EMAP_ERR_HIP; }
      break;
    }
  }
  if (ctx->sf_host[1] != 0u) return sf_recover(ctx);
  sf_forget_applied(ctx);
  return EMAP_OK;
}
#define SF_CHECK() do { if (ctx->sf_count > 0) { int rc_sf_ = sf_settle(ctx); if (rc_sf_) return rc_sf_; } } while (0)
static int frame_sem_begin(emap_ctx* ctx, bool rays_on);                               // (the frame's semantic fusion: defined next to emap_semantic_update)
static int frame_sem_finish(emap_ctx* ctx, const float R[9], const float t[3]);

static void build_kp(emap_ctx* ctx) {
  const emap_params& p = ctx->prm;
  KP& k = ctx->kp;
  const KP old = k;
  memset(&k, 0, sizeof k);
  k.org_r = old.org_r; k.org_c = old.org_c; k.norg_r = old.norg_r; k.norg_c = old.norg_c; k.mv = old.mv;   // map-shift state survives a parameter update
  k.C = p.cell_n; k.mode = p.mode; k.row0 = ctx->strip.row_begin; k.nrows = ctx->strip.row_count; k.halo = ctx->strip.halo_rows;
  k.edge = p.enable_edge_sharpen; k.dil = p.dilation_size;
  k.res = p.resolution; k.half_w = 0.5 * p.cell_n; k.snf = p.sensor_noise_factor; k.mt = p.mahalanobis_thresh;
  k.ov = p.outlier_variance; k.dcvi_half = p.drift_compensation_variance_inlier / 2.0; k.trav_inlier = p.traversability_inlier;
  k.wall = p.wall_num_thresh; k.mrl = p.max_ray_length; k.cs = p.cleanup_step; k.cos_thresh = p.cleanup_cos_thresh;
  k.mvd2 = p.min_valid_distance * p.min_valid_distance; k.mhr = p.max_height_range;
  k.ra = p.ramped_height_range_a; k.rb = p.ramped_height_range_b; k.rc = p.ramped_height_range_c;
  k.max_var = p.max_variance; k.ray_step = p.ray_step;
  k.init_var = (float)p.initial_variance; k.ov_f = (float)p.outlier_variance;
  const bool h = p.mode == EMAP_MODE_REFERENCE_FP16;
  k.q_wm1 = h ? q16((float)(p.cell_n - 1)) : (float)(p.cell_n - 1);
  k.q_mrl = h ? q16((float)p.max_ray_length) : (float)p.max_ray_length;
  k.q_step = h ? q16((float)p.ray_step) : (float)p.ray_step;
  k.time_var = (float)p.time_variance; k.time_int = (float)p.time_interval; k.res_f = (float)p.resolution;
  k.inv_res_f = (float)(1.0 / p.resolution); k.half_w_f = 0.5f * (float)p.cell_n; k.cm1_f = (float)(p.cell_n - 1);
  k.hw_int_f = (float)(p.cell_n / 2); k.hw_frac_f = (p.cell_n & 1) ? 0.5f : 0.0f; k.pad1 = 0.f;
  k.col0 = 0; k.ncols = p.cell_n; k.pitch = p.cell_n; k.wmode = 0;
}

// smallest float >= c (a < c  <=>  a < up(c) for float a) / largest float <= c (a > c <=> a > dn(c))
static float f_up(double c) { float f = (float)c; if ((double)f < c) f = nextafterf(f, INFINITY); return f; }
static float f_dn(double c) { float f = (float)c; if ((double)f > c) f = nextafterf(f, -INFINITY); return f; }

static int host_axis_idx(const emap_params& p, float xq, float q_wm1, bool half_mode) {
  if (!half_mode) {   // fp32 mode: float multiply + float add, truncation, integer clamp (emap_device.h: axis_idx<1>)
    volatile float m = xq * (float)(1.0 / p.resolution);
    volatile float vf = m + 0.5f * (float)p.cell_n;
    const float f = vf;
    int i = !(f == f) ? 0 : (f >= 2147483648.0f ? 2147483647 : (f <= -2147483648.0f ? (int)0x80000000 : (int)f));
    return i < 0 ? 0 : (i > p.cell_n - 1 ? p.cell_n - 1 : i);
  }
  double v = (double)xq / p.resolution + 0.5 * p.cell_n;
  int i = !(v == v) ? 0 : (v >= 2147483647.0 ? 2147483647 : (v <= -2147483648.0 ? (int)0x80000000 : (int)v));
  float fi = half_mode ? q16((float)i) : (float)i;
  float r = fmaxf(fminf(fi, q_wm1), 0.0f);
  return (int)r;
}

// Tables of the visibility pass: step sequence (custom_kernels.py:203) and, for reference_fp16, the
// half-bits -> cell-index table (custom_kernels.py:22-33,45-49 evaluated exactly on the host).
static int build_ray_tables(emap_ctx* ctx) {
  const emap_params& p = ctx->prm;
  const bool h = p.mode == EMAP_MODE_REFERENCE_FP16;
  RayTab& rt = ctx->rt;
  if (ctx->ray_S) { CK(hipStreamSynchronize(ctx->stream)); CK(hipFree(ctx->ray_S)); ctx->ray_S = nullptr; }
  if (ctx->ray_lut) { CK(hipStreamSynchronize(ctx->stream)); CK(hipFree(ctx->ray_lut)); ctx->ray_lut = nullptr; }
  memset(&rt, 0, sizeof rt);
  std::vector<float> S;
  const float q_mrl = ctx->kp.q_mrl;
  float s = ctx->kp.q_step;
  while (s < q_mrl) {
    if (S.size() >= 8192u) {                  // the table is staged in LDS (<= 32 KB): refuse instead of silently shortening the rays
      ctx->err = "max_ray_length / ray_step needs more than 8192 ray samples"; return EMAP_ERR_INVALID; }
    S.push_back(s);
    float nx = (float)((double)s + p.ray_step);
    if (h) nx = q16(nx);
    if (!(nx > s)) break;   // half saturated: the reference's loop would never terminate here
    s = nx;
  }
  rt.nS = (int)S.size();
  if (rt.nS) {
    CK(hipMalloc((void**)&ctx->ray_S, sizeof(float) * S.size()));
    CK(hipMemcpyAsync(ctx->ray_S, S.data(), sizeof(float) * S.size(), hipMemcpyHostToDevice, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
  }
  rt.S = ctx->ray_S;
  rt.f_d_thresh = f_up(0.1); rt.f_cos_thresh = f_up(p.cleanup_cos_thresh); rt.f_wall = f_dn(p.wall_num_thresh);
  if (h) {
    std::vector<int> full(65536);
    for (int b = 0; b < 65536; ++b) {
      unsigned short us = (unsigned short)b; _Float16 hf; memcpy(&hf, &us, 2);
      full[b] = host_axis_idx(p, (float)hf, ctx->kp.q_wm1, true);
    }
    auto at = [&](int sg, int mag) { return full[(sg << 15) | mag]; };
    {   // AxisIdx<0, 2>: is the float formula exact for every half pattern (NaNs excluded: non-finite samples never march)?
      bool same = true;
      // index = floor(q / res [+ 0.5 for odd cell_n]) + cell_n / 2, clamped: the floor is taken BEFORE the half width is added (a tiny
      // negative coordinate would otherwise round up to exactly cell_n / 2 in fp32, where the reference's double truncates below it)
      const float inv = ctx->kp.inv_res_f, hwi = ctx->kp.hw_int_f, hwf = ctx->kp.hw_frac_f, cm1 = ctx->kp.cm1_f;
      for (int b = 0; b < 65536 && same; ++b) {
        if ((b & 0x7fff) > 0x7c00) continue;
        unsigned short us = (unsigned short)b; _Float16 hf; memcpy(&hf, &us, 2);
        const float q = (float)hf, f = floorf(fmaf(q, inv, hwf)) + hwi;
        const int got = (int)fminf(fmaxf(f, 0.0f), cm1);
        if (got != full[b]) same = false;
      }
      rt.formula_ok = same ? 1 : 0;
      ctx->kp.idx_formula = rt.formula_ok;       // the point passes index with the same formula (geometry())
      if (const char* e = getenv("EMAP_RAY_IDX")) { if (atoi(e) != 2) rt.formula_ok = 0; }     // test / tuning hook: force the table path
    }
    int lo = 0x7c00, hi = 1;
    for (int sg = 0; sg < 2; ++sg) {
      for (int m = 1; m < 0x7c00; ++m) if (at(sg, m) != at(sg, 1)) { if (m < lo) lo = m; break; }
      for (int m = 0x7bff; m >= 1; --m) if (at(sg, m) != at(sg, 0x7c00)) { if (m + 1 > hi) hi = m + 1; break; }
    }
    if (hi < lo) { hi = lo; }
    // verify the compact form against the full table (every pattern), else fall back to in-kernel arithmetic
    bool ok = at(0, 0) == at(0, 1) && at(1, 0) == at(0, 1);
    rt.small_pos = at(0, 1); rt.small_neg = at(1, 1); rt.big_pos = at(0, 0x7c00); rt.big_neg = at(1, 0x7c00); rt.nan_val = at(0, 0x7e00);
    for (int sg = 0; sg < 2 && ok; ++sg)
      for (int m = 1; m < 0x8000; ++m) {
        int want = at(sg, m), got;
        if (m >= lo && m < hi) continue;
        if (m < lo) got = sg ? rt.small_neg : rt.small_pos; else if (m > 0x7c00) got = rt.nan_val; else got = sg ? rt.big_neg : rt.big_pos;
        if (got != want) { ok = false; break; }
      }
    const size_t entries = (size_t)2 * (hi - lo);
    if (ok && entries > 0 && lo >= 1 && (entries + 4) * 2 <= 60 * 1024 && p.cell_n <= 65535) {
      const size_t span = (size_t)(hi - lo) + 2;   // per sign: [small, idx(lo..hi-1), big]
      std::vector<unsigned short> lut(2 * span + 2);
      for (int sg = 0; sg < 2; ++sg) {
        lut[sg * span] = (unsigned short)(sg ? rt.small_neg : rt.small_pos);
        for (int m = lo; m < hi; ++m) lut[sg * span + 1 + (m - lo)] = (unsigned short)at(sg, m);
        lut[sg * span + span - 1] = (unsigned short)(sg ? rt.big_neg : rt.big_pos);
      }
      CK(hipMalloc((void**)&ctx->ray_lut, sizeof(unsigned short) * lut.size()));
      CK(hipMemcpyAsync(ctx->ray_lut, lut.data(), sizeof(unsigned short) * lut.size(), hipMemcpyHostToDevice, ctx->stream));
      CK(hipStreamSynchronize(ctx->stream));
      rt.lut = ctx->ray_lut; rt.lo = lo; rt.hi = hi;
    }
  }
  return EMAP_OK;
}

static Pose make_pose(const emap_ctx* ctx, const float R[9], const float t[3]) {
  Pose T;
  const bool h = ctx->prm.mode == EMAP_MODE_REFERENCE_FP16;
  for (int i = 0; i < 9; ++i) T.Rq[i] = h ? q16(R[i]) : R[i];
  for (int i = 0; i < 3; ++i) { T.tq[i] = h ? q16(t[i]) : t[i]; T.t[i] = t[i] + 0.0f; }   // -0.0 -> +0.0 (same arithmetic, see IdxLut)
  return T;
}

// Map shifts are lazy (emap_shift): kernels of the frame replay them, everything else sees them written out first.
static int flush_moves(emap_ctx* ctx) {
  if (ctx->kp.mv.n == 0) return EMAP_OK;
  launch_materialize(ctx->stream, ctx->kp, ctx->cells);
  ctx->kp.mv.n = 0;
  CK(hipGetLastError());
  return EMAP_OK;
}
#define FLUSH() do { int rc_ = flush_moves(ctx); if (rc_) return rc_; } while (0)

// rows between the origin the normal planes were written with and the map's origin (signed, shortest way round)
static int normal_row_lag(const emap_ctx* ctx) {
  const int C = ctx->prm.cell_n;
  int d = ((ctx->kp.norg_r - ctx->kp.org_r) % C + C) % C;
  return d > C / 2 ? d - C : d;
}

static int plugin_scratch(emap_ctx* ctx, size_t planes, int counters) {
  const size_t need = planes * (size_t)ctx->prm.cell_n * ctx->prm.cell_n;
  if (need > ctx->plug_cap) {
    CK(hipStreamSynchronize(ctx->stream));
    if (ctx->plug_buf) CK(hipFree(ctx->plug_buf));
    ctx->plug_buf = nullptr; ctx->plug_cap = 0;
    CK(hipMalloc((void**)&ctx->plug_buf, sizeof(float) * need));
    ctx->plug_cap = need;
  }
  if (counters > ctx->plug_cnt_cap) {
    CK(hipStreamSynchronize(ctx->stream));
    if (ctx->plug_cnt) CK(hipFree(ctx->plug_cnt));
    ctx->plug_cnt = nullptr; ctx->plug_cnt_cap = 0;
    CK(hipMalloc((void**)&ctx->plug_cnt, sizeof(unsigned int) * counters));
    ctx->plug_cnt_cap = counters;
  }
  return EMAP_OK;
}

static int validate(const emap_params* p, const emap_strip* s, std::string* why) {
  if (!p) { *why = "params null"; return 0; }
  if (p->cell_n < 8 || p->cell_n > 46340) { *why = "cell_n out of range"; return 0; }
  if (p->mode != EMAP_MODE_REFERENCE_FP16 && p->mode != EMAP_MODE_FP32) { *why = "bad mode"; return 0; }
  if (p->mode == EMAP_MODE_REFERENCE_FP16 && p->cell_n > 2049) {
    *why = "reference_fp16 index mode is only defined for cell_n <= 2049 (half cannot hold larger indices)"; return 0; }
  if (!(p->resolution > 0)) { *why = "resolution must be > 0"; return 0; }
  if (p->dilation_size < 0 || p->dilation_size > 32) { *why = "dilation_size out of range"; return 0; }
  if (s) {
    if (s->row_begin < 0 || s->row_count <= 0 || s->row_begin + s->row_count > p->cell_n || s->halo_rows < 0) {
      *why = "bad strip"; return 0; }
  }
  return 1;
}

extern "C" {

int emap_abi_version(void) { return EMAP_ABI_VERSION; }

const char* emap_last_error(const emap_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int emap_destroy(emap_ctx* ctx) {
  if (!ctx) return EMAP_OK;
  hipSetDevice(ctx->device);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  hipFree(ctx->cells.hot); hipFree(ctx->cells.cold); hipFree(ctx->acc); hipFree(ctx->accr); hipFree(ctx->trav_in);
  hipFree(ctx->normal); hipFree(ctx->scratch); hipFree(ctx->plug_buf); hipFree(ctx->plug_cnt); hipFree(ctx->slots); hipFree(ctx->frame); hipFree(ctx->cnt_sync);
  for (int k = 0; k < 2; ++k) { hipFree(ctx->pts_dev[k]); if (ctx->pts_pin[k]) hipHostFree(ctx->pts_pin[k]); if (ctx->ev_copied[k]) hipEventDestroy(ctx->ev_copied[k]); if (ctx->ev_used[k]) hipEventDestroy(ctx->ev_used[k]); }
  if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
  delete ctx->workers;
  hipFree(ctx->tail_idx); hipFree(ctx->tail_flags); hipFree(ctx->ray_S); hipFree(ctx->ray_lut); hipFree(ctx->inert); hipFree(ctx->inl_plane); hipFree(ctx->ray_thr);
  hipFree(ctx->bin_recs); hipFree(ctx->bin_own); hipFree(ctx->bin_own_cnt); hipFree(ctx->bin_hist); hipFree(ctx->bin_tile_total); hipFree(ctx->bin_tile_start); hipFree(ctx->bin_sync); hipFree(ctx->split_mem); hipFree(ctx->sem_split_mem); if (ctx->split_need) hipHostFree((void*)ctx->split_need); if (ctx->sf_host) hipHostFree((void*)ctx->sf_host); hipFree(ctx->frame_save); hipFree(ctx->sf_poison);
  hipFree(ctx->img_uv); hipFree(ctx->img_valid); hipFree(ctx->img_buf);
  hipFree(ctx->sem_alpha); hipFree(ctx->sem); hipFree(ctx->sem_sums); hipFree(ctx->sem_col); hipFree(ctx->cnt_plane);
  emap_comm_destroy(ctx);
  hipFree(ctx->win_state); hipFree(ctx->win_rec); hipFree(ctx->win_bits); hipFree(ctx->win_thr); hipFree(ctx->win_dh); hipFree(ctx->win_key); hipFree(ctx->win_red_dh); hipFree(ctx->win_red_key);
  for (int i = 0; i <= ST_N; ++i) if (ctx->ev[i]) hipEventDestroy(ctx->ev[i]);
  if (ctx->t0) hipEventDestroy(ctx->t0);
  if (ctx->t1) hipEventDestroy(ctx->t1);
  if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return EMAP_OK;
}

int emap_clear(emap_ctx* ctx) {
  CKARG(ctx, "null ctx"); SF_CHECK();
  CK(hipSetDevice(ctx->device));
  // ElevationMap.clear (elevation_mapping.py:119-128): all planes 0, variance = initial_variance
  Cell z = {0.f, (float)ctx->prm.initial_variance, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  launch_fill_cells(ctx->stream, ctx->cells, ctx->ncells_alloc, z);
  ctx->kp.mv.n = 0;                   // every cell has just been written
  CK(hipMemsetAsync(ctx->acc, 0, sizeof(AccF) * ctx->ncells_alloc, ctx->stream));
  CK(hipMemsetAsync(ctx->accr, 0, sizeof(AccR) * ctx->ncells_alloc, ctx->stream));
  CK(hipMemsetAsync(ctx->slots, 0, sizeof(ErrSlot) * EM_ERR_SLOTS, ctx->stream));
  CK(hipMemsetAsync(ctx->frame, 0, sizeof(FrameDev), ctx->stream));
  if (ctx->split_need) {              // FrameDev::ray_class and its host-mapped mirror restart together (no frame is in flight after the wait)
    CK(hipStreamSynchronize(ctx->stream));
    ctx->split_need[1] = 0u;
  }
  ctx->committed = false;
  CK(hipGetLastError());
  return EMAP_OK;
}

int emap_create(const emap_params* params, const emap_strip* strip, int device, void* stream, emap_ctx** out) {
  if (!out) return EMAP_ERR_INVALID;
  *out = nullptr;
  std::string why;
  if (!validate(params, strip, &why)) { fprintf(stderr, "emap_create: %s\n", why.c_str()); return EMAP_ERR_INVALID; }
  emap_ctx* ctx = new (std::nothrow) emap_ctx();
  if (!ctx) return EMAP_ERR_INVALID;
  ctx->prm = *params;
  if (strip) ctx->strip = *strip; else { ctx->strip.row_begin = 0; ctx->strip.row_count = params->cell_n; ctx->strip.halo_rows = 0; ctx->strip.pad_ = 0; }
  ctx->device = device;
  ctx->bin_rs = 1;
  build_kp(ctx);
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) { fprintf(stderr, "emap_create: hipSetDevice(%d): %s\n", device, hipGetErrorString(e)); delete ctx; return EMAP_ERR_HIP; }
  if (stream) { ctx->stream = (hipStream_t)stream; ctx->own_stream = false; }
  else { e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking); ctx->own_stream = true;
         if (e != hipSuccess) { fprintf(stderr, "emap_create: stream: %s\n", hipGetErrorString(e)); delete ctx; return EMAP_ERR_HIP; } }
  const long C = params->cell_n;
  ctx->ncells_alloc = (long)(ctx->strip.row_count + 2 * ctx->strip.halo_rows) * C;
  const long n = ctx->ncells_alloc;
  int rc = EMAP_OK;
  auto alloc = [&](void** p, size_t bytes) { if (rc == EMAP_OK && hipMalloc(p, bytes) != hipSuccess) { rc = EMAP_ERR_HIP; fprintf(stderr, "emap_create: hipMalloc(%zu) failed\n", bytes); } };
  alloc((void**)&ctx->cells.hot, sizeof(float4) * n); alloc((void**)&ctx->cells.cold, sizeof(float4) * n); alloc((void**)&ctx->acc, sizeof(AccF) * n); alloc((void**)&ctx->accr, sizeof(AccR) * n);
  alloc((void**)&ctx->trav_in, sizeof(float) * n); alloc((void**)&ctx->normal, sizeof(float) * 3 * n);
  alloc((void**)&ctx->scratch, sizeof(float) * n); alloc((void**)&ctx->slots, sizeof(ErrSlot) * EM_ERR_SLOTS);
  alloc((void**)&ctx->frame, sizeof(FrameDev));
  alloc((void**)&ctx->cnt_sync, sizeof(unsigned int) * EM_TICKET_WORDS);      // tickets of k_count's folded gate (zero between launches)
  alloc((void**)&ctx->inert, sizeof(unsigned long long) * ((size_t)ctx->strip.row_count * ((C + 63) / 64) + 2));      // + an all-ones word behind the last row (k_rays)
  if (rc == EMAP_OK) {
    hipEventCreate(&ctx->t0); hipEventCreate(&ctx->t1);
    for (int i = 0; i <= ST_N; ++i) hipEventCreate(&ctx->ev[i]);
    hipMemsetAsync(ctx->trav_in, 0, sizeof(float) * n, ctx->stream);
    hipMemsetAsync(ctx->cnt_sync, 0, sizeof(unsigned int) * EM_TICKET_WORDS, ctx->stream);
    hipMemsetAsync(ctx->inert + (size_t)ctx->strip.row_count * ((C + 63) / 64), 0xff, 2 * sizeof(unsigned long long), ctx->stream);
    hipMemsetAsync(ctx->normal, 0, sizeof(float) * 3 * n, ctx->stream);
    rc = emap_clear(ctx);
    if (rc == EMAP_OK) rc = build_ray_tables(ctx);
    if (rc == EMAP_OK) {
      // ElevationMap.__init__: traversability plane starts at 1 (elevation_mapping.py:84)
      Cell z = {0.f, (float)params->initial_variance, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f};
      launch_fill_cells(ctx->stream, ctx->cells, n, z);
      if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = EMAP_ERR_HIP;
    }
  }
  if (rc != EMAP_OK) { emap_destroy(ctx); return rc; }
  *out = ctx;
  return EMAP_OK;
}

int emap_set_params(emap_ctx* ctx, const emap_params* params) {
  CKARG(ctx && params, "null argument"); SF_CHECK();
  CKARG(params->cell_n == ctx->prm.cell_n, "cell_n cannot change");
  std::string why;
  if (!validate(params, &ctx->strip, &why)) { ctx->err = why; return EMAP_ERR_INVALID; }
  ctx->prm = *params;
  build_kp(ctx);
  CK(hipSetDevice(ctx->device));
  return build_ray_tables(ctx);
}

int emap_sync(emap_ctx* ctx) {
  CKARG(ctx, "null ctx"); CK(hipSetDevice(ctx->device)); CK(hipStreamSynchronize(ctx->stream));
  if (ctx->sf_count > 0) { SF_CHECK(); CK(hipStreamSynchronize(ctx->stream)); }      // (an aborted small frame is re-run before the caller is told the stream is idle)
  return EMAP_OK;
}

// ---- point cloud --------------------------------------------------------------------------------------
// Which points of a cloud can land in THIS strip's rows under the pose (R, t map-centre relative)?  A conservative host-side test in
// double arithmetic: the kernels decide the row with the index mode's own rounding (emap_device.h: geometry / axis_idx -- in
// reference_fp16 mode coordinates, R and t pass through binary16), so the row estimate gets a margin that bounds every rounding on
// that path (2^-10 relative on every term in reference_fp16 mode, 2^-20 in fp32 mode) plus two cells; points outside the map clamp to
// its first / last row like the kernels' index does.  A point with a NaN coordinate is skipped by every kernel: never kept; one the
// test cannot place (non-finite row estimate, magnitudes beyond binary16) is kept everywhere.  The kept set is a SUPERSET of the
// points the strip's kernels act on, in the cloud's order -- so the strip's maps stay bit-identical (the kernels repeat the exact test).
struct StripKeep {
  double R0, R1, R2, t0, res, eps, half_c; int C, ls, nrows; bool half_mode;
  StripKeep(const emap_ctx* ctx, const float R[9], const float t[3]) {
    const emap_params& p = ctx->prm;
    R0 = R[0]; R1 = R[1]; R2 = R[2]; t0 = t[0]; res = p.resolution; C = p.cell_n; half_c = 0.5 * C;
    half_mode = p.mode == EMAP_MODE_REFERENCE_FP16;
    eps = half_mode ? 1.0 / 1024.0 : 1.0 / 1048576.0;
    ls = ((ctx->strip.row_begin - ctx->kp.org_r) % C + C) % C;      // logical row of the strip's first physical row
    nrows = ctx->strip.row_count;
  }
  bool operator()(double x, double y, double z) const {
    if (x != x || y != y || z != z) return false;
    const double a = R0 * x, b = R1 * y, c = R2 * z, xf = a + b + c + t0;
    if (!(std::fabs(xf) <= 1.0e30) || (half_mode && (std::fabs(x) > 60000.0 || std::fabs(y) > 60000.0 || std::fabs(z) > 60000.0 || std::fabs(xf) > 60000.0))) return true;
    const double m = eps * (2.0 * (std::fabs(a) + std::fabs(b) + std::fabs(c)) + std::fabs(t0) + std::fabs(xf)) + 2.0 * res;
    auto row = [&](double v) { const double r = std::floor(v / res + half_c); return r < 0.0 ? 0 : (r > C - 1 ? C - 1 : (int)r); };
    const int lo = row(xf - m), hi = row(xf + m);
    const int d = ((lo - ls) % C + C) % C;
    return d < nrows || d + (hi - lo) >= C;
  }
};

// conversion + (optional) order-preserving compaction of a host cloud into the pinned slot and out to the device, chunk by chunk
static int upload_impl(emap_ctx* ctx, const void* host, int64_t n, int64_t stride, int dtype, const StripKeep* keep, int64_t* n_kept) { SF_CHECK();
  CK(hipSetDevice(ctx->device));
  const long tot = (long)n * stride + (stride > 3 ? 64 : 0);      // (+ the padding in front of the channel matrix)
  if (!ctx->copy_stream) {
    CK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) { CK(hipEventCreateWithFlags(&ctx->ev_copied[k], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ctx->ev_used[k], hipEventDisableTiming)); }
    int nt = (int)std::thread::hardware_concurrency() / 2; if (nt > 8) nt = 8; if (nt < 1) nt = 1;
    if (const char* e = getenv("EMAP_UPLOAD_THREADS")) { int v = atoi(e); if (v >= 1 && v <= 64) nt = v; }
    ctx->workers = new Workers(nt);
  }
  // everything enqueued so far may read the buffer of the previous upload: mark the point after which that buffer is free again
  if (ctx->up_used[ctx->up_slot]) CK(hipEventRecord(ctx->ev_used[ctx->up_slot], ctx->stream));
  const int sl = ctx->up_slot ^= 1;
  // this slot's device buffer was read by the kernels before the previous upload, its pinned memory by the DMA two uploads ago
  if (ctx->up_used[sl]) { CK(hipEventSynchronize(ctx->ev_used[sl])); CK(hipEventSynchronize(ctx->ev_copied[sl])); }
  if (tot > ctx->pts_cap[sl]) {
    if (ctx->pts_dev[sl]) CK(hipFree(ctx->pts_dev[sl]));
    ctx->pts_dev[sl] = nullptr; ctx->pts_cap[sl] = 0;
    CK(hipMalloc((void**)&ctx->pts_dev[sl], sizeof(float) * tot));
    ctx->pts_cap[sl] = tot;
  }
  if (tot > ctx->pin_cap[sl]) {
    if (ctx->pts_pin[sl]) CK(hipHostFree(ctx->pts_pin[sl]));
    ctx->pts_pin[sl] = nullptr; ctx->pin_cap[sl] = 0;
    CK(hipHostMalloc((void**)&ctx->pts_pin[sl], sizeof(float) * tot, hipHostMallocDefault));
    ctx->pin_cap[sl] = tot;
  }
  // A cloud with extra channels is DE-INTERLEAVED on the way: the workers write xyz as an (n, 3) matrix and the K channels as an
  // (n, K) matrix behind it (at a 256-byte boundary) -- the conversion touches every element anyway -- so that the frame's point
  // passes stream 12 bytes per point instead of 12 + 4 K and the semantic fusion reads one K-float record per point.
  // With `keep` (row strips, emap_upload_points_strip) only the points that pass it are written, in order: every worker counts the
  // survivors of its share of the chunk, the shares' offsets follow, a second pass over the (cache-resident) share writes them.
  const long K = stride - 3, chan_off = K ? ((3 * (long)n + 63) & ~63L) : 0;
  long kept = 0;
  if (tot > 0) {
    float* pin = ctx->pts_pin[sl];
    const long cp = (1L << 19) / stride > 0 ? (1L << 19) / stride : 1;      // points per chunk: about 2 MB of float32 per DMA
    std::vector<long> cnt, off;
    for (long p0 = 0; p0 < (long)n; p0 += cp) {
      const long p1 = p0 + cp < (long)n ? p0 + cp : (long)n;
      long out0 = p0, out1 = p1;                                    // rows [out0, out1) of the device matrices this chunk fills
      auto coord = [&](long i, int k) -> double { return dtype == 0 ? (double)static_cast<const float*>(host)[i * stride + k] : static_cast<const double*>(host)[i * stride + k]; };
      if (keep) {
        cnt.assign(ctx->workers->th.size(), 0); off.assign(ctx->workers->th.size(), 0);
        ctx->workers->run([&](int w, int nw) {
          const long per = (p1 - p0 + nw - 1) / nw, a = p0 + (long)w * per, b = a + per < p1 ? a + per : p1;
          long c = 0;
          for (long i = a; i < b; ++i) c += (*keep)(coord(i, 0), coord(i, 1), coord(i, 2)) ? 1 : 0;
          cnt[w] = c;
        });
        long run = kept;
        for (size_t w = 0; w < cnt.size(); ++w) { off[w] = run; run += cnt[w]; }
        out0 = kept; out1 = run;
      }
      ctx->workers->run([&](int w, int nw) {
        const long per = (p1 - p0 + nw - 1) / nw, a = p0 + (long)w * per, b = a + per < p1 ? a + per : p1;
        if (b <= a) return;
        if (keep) {
          long o = off[w];
          for (long i = a; i < b; ++i) {
            const double x = coord(i, 0), y = coord(i, 1), z = coord(i, 2);
            if (!(*keep)(x, y, z)) continue;
            pin[3 * o] = (float)x; pin[3 * o + 1] = (float)y; pin[3 * o + 2] = (float)z;      // fp32 cast (:456)
            for (long k = 0; k < K; ++k) pin[chan_off + K * o + k] = (float)coord(i, 3 + (int)k);
            ++o;
          }
        } else if (K == 0) {
          if (dtype == 0) memcpy(pin + 3 * a, static_cast<const float*>(host) + 3 * a, sizeof(float) * 3 * (size_t)(b - a));
          else { const double* src = static_cast<const double*>(host); for (long i = 3 * a; i < 3 * b; ++i) pin[i] = (float)src[i]; }   // fp32 cast (:456)
        } else if (dtype == 0) {
          const float* src = static_cast<const float*>(host);
          for (long i = a; i < b; ++i) {
            const float* r = src + i * stride;
            pin[3 * i] = r[0]; pin[3 * i + 1] = r[1]; pin[3 * i + 2] = r[2];
            for (long k = 0; k < K; ++k) pin[chan_off + K * i + k] = r[3 + k];
          }
        } else {
          const double* src = static_cast<const double*>(host);
          for (long i = a; i < b; ++i) {
            const double* r = src + i * stride;
            pin[3 * i] = (float)r[0]; pin[3 * i + 1] = (float)r[1]; pin[3 * i + 2] = (float)r[2];
            for (long k = 0; k < K; ++k) pin[chan_off + K * i + k] = (float)r[3 + k];
          }
        }
      });
      kept = keep ? out1 : p1;
      if (out1 > out0) {
        CK(hipMemcpyAsync(ctx->pts_dev[sl] + 3 * out0, pin + 3 * out0, sizeof(float) * 3 * (size_t)(out1 - out0), hipMemcpyHostToDevice, ctx->copy_stream));
        if (K) CK(hipMemcpyAsync(ctx->pts_dev[sl] + chan_off + K * out0, pin + chan_off + K * out0, sizeof(float) * (size_t)K * (size_t)(out1 - out0), hipMemcpyHostToDevice, ctx->copy_stream));
      }
    }
    if (keep && kept == 0 && n > 0) {
      // none of the cloud's points can land in this strip's rows: bind ONE row of NaNs (skipped by every kernel, like any NaN row of a
      // cloud) so that the frame takes the same path -- launches, collectives -- as on the ranks that did get points
      pin[0] = pin[1] = pin[2] = NAN;
      for (long k = 0; k < K; ++k) pin[chan_off + k] = 0.f;
      CK(hipMemcpyAsync(ctx->pts_dev[sl], pin, sizeof(float) * 3, hipMemcpyHostToDevice, ctx->copy_stream));
      if (K) CK(hipMemcpyAsync(ctx->pts_dev[sl] + chan_off, pin + chan_off, sizeof(float) * (size_t)K, hipMemcpyHostToDevice, ctx->copy_stream));
      kept = 1;
    }
    CK(hipEventRecord(ctx->ev_copied[sl], ctx->copy_stream));
    CK(hipStreamWaitEvent(ctx->stream, ctx->ev_copied[sl], 0));     // kernels enqueued from now on see the cloud; nothing waits on the host
    ctx->up_used[sl] = true;
  }
  ctx->pts = ctx->pts_dev[sl]; ctx->n_pts = keep ? kept : (long)n; ctx->n_pts_all = (long)n; ctx->stride = 3; ctx->n_cols = (int)stride;
  ctx->chan.p = K ? ctx->pts_dev[sl] + chan_off : ctx->pts_dev[sl]; ctx->chan.stride = K ? (int)K : 3; ctx->chan.col0 = K ? 3 : 0;
  ctx->pts_bucketed = false;
  if (n_kept) *n_kept = ctx->n_pts;
  return EMAP_OK;
}

int emap_upload_points(emap_ctx* ctx, const void* host, int64_t n, int64_t stride, int dtype) {
  CKARG(ctx, "null ctx");
  CKARG(n >= 0 && stride >= 3 && stride < 4096 && (dtype == 0 || dtype == 1), "bad point buffer description");
  CKARG(n == 0 || host, "null host buffer");
  return upload_impl(ctx, host, n, stride, dtype, nullptr, nullptr);
}

// Row strips: every rank of a multi-GPU map is handed the SAME cloud (one sensor message), but only the points of its rows ever reach
// its tile kernels.  This entry point converts, uploads and binds only those (StripKeep: a conservative superset, order preserved), so a
// rank moves and streams 1 / G of the cloud: 8 x 192 MB over PCIe per 16 M-point frame on 8 GPUs become 192 MB.  The frame that follows
// must use the SAME pose (checked) and must not march its rays by row (every valid point marches a ray through every strip then:
// emap_update_sharded refuses); a cloud of a whole-map context is uploaded as by emap_upload_points.
int emap_upload_points_strip(emap_ctx* ctx, const void* host, int64_t n, int64_t stride, int dtype, const float R[9], const float t[3], int64_t* n_kept) {
  CKARG(ctx && R && t, "null argument");
  CKARG(n >= 0 && stride >= 3 && stride < 4096 && (dtype == 0 || dtype == 1), "bad point buffer description");
  CKARG(n == 0 || host, "null host buffer");
  if (ctx->strip.row_count >= ctx->prm.cell_n) return upload_impl(ctx, host, n, stride, dtype, nullptr, n_kept);
  const StripKeep keep(ctx, R, t);
  int rc = upload_impl(ctx, host, n, stride, dtype, &keep, n_kept);
  if (rc) return rc;
  ctx->pts_bucketed = true; ctx->bucket_org_r = ctx->kp.org_r;
  memcpy(ctx->bucket_R, R, sizeof ctx->bucket_R); memcpy(ctx->bucket_t, t, sizeof ctx->bucket_t);
  return EMAP_OK;
}

// the same predicate as data: keep[i] = 1 iff point i would be uploaded by emap_upload_points_strip (callers that keep their clouds
// on the device -- bench.py -- bucket them once with it; tests check the superset property against the exact cell indices)
int emap_strip_point_mask(emap_ctx* ctx, const void* host, int64_t n, int64_t stride, int dtype, const float R[9], const float t[3], uint8_t* keep_out) {
  CKARG(ctx && R && t && keep_out, "null argument");
  CKARG(n >= 0 && stride >= 3 && stride < 4096 && (dtype == 0 || dtype == 1), "bad point buffer description");
  CKARG(n == 0 || host, "null host buffer");
  const bool whole = ctx->strip.row_count >= ctx->prm.cell_n;
  const StripKeep keep(ctx, R, t);
  for (int64_t i = 0; i < n; ++i) {
    const double x = dtype == 0 ? (double)static_cast<const float*>(host)[i * stride] : static_cast<const double*>(host)[i * stride];
    const double y = dtype == 0 ? (double)static_cast<const float*>(host)[i * stride + 1] : static_cast<const double*>(host)[i * stride + 1];
    const double z = dtype == 0 ? (double)static_cast<const float*>(host)[i * stride + 2] : static_cast<const double*>(host)[i * stride + 2];
    keep_out[i] = (whole || keep(x, y, z)) ? 1 : 0;
  }
  return EMAP_OK;
}

// A device-resident cloud the caller bucketed itself (with emap_strip_point_mask's predicate, for THIS pose): declares it, so that the
// frame checks pose and ray mode as for an uploaded one.
int emap_declare_points_bucketed(emap_ctx* ctx, const float R[9], const float t[3], int64_t n_all) {
  CKARG(ctx && R && t && n_all >= ctx->n_pts, "bad argument");
  if (ctx->strip.row_count >= ctx->prm.cell_n) return EMAP_OK;
  ctx->pts_bucketed = true; ctx->n_pts_all = (long)n_all; ctx->bucket_org_r = ctx->kp.org_r;
  memcpy(ctx->bucket_R, R, sizeof ctx->bucket_R); memcpy(ctx->bucket_t, t, sizeof ctx->bucket_t);
  return EMAP_OK;
}

int emap_set_points_device_split(emap_ctx* ctx, const float* xyz_dev, const float* chan_dev, int64_t n, int64_t n_chan) {
  CKARG(ctx, "null ctx");
  CKARG(n >= 0 && n_chan >= 0 && n_chan < 4093 && (n == 0 || (xyz_dev && (n_chan == 0 || chan_dev))), "bad device point buffers");
  ctx->pts = xyz_dev; ctx->n_pts = (long)n; ctx->stride = 3; ctx->n_cols = 3 + (int)n_chan; ctx->pts_bucketed = false; ctx->n_pts_all = (long)n;
  ctx->chan.p = n_chan ? chan_dev : xyz_dev; ctx->chan.stride = n_chan ? (int)n_chan : 3; ctx->chan.col0 = n_chan ? 3 : 0;
  return EMAP_OK;
}

int emap_set_points_device(emap_ctx* ctx, const float* dev, int64_t n, int64_t stride) {
  CKARG(ctx, "null ctx");
  CKARG(n >= 0 && stride >= 3 && stride < 4096 && (n == 0 || dev), "bad device point buffer");
  ctx->pts = dev; ctx->n_pts = (long)n; ctx->stride = (int)stride; ctx->n_cols = (int)stride; ctx->pts_bucketed = false; ctx->n_pts_all = (long)n;
  ctx->chan.p = dev; ctx->chan.stride = (int)stride; ctx->chan.col0 = 0;            // interleaved rows: the channels sit behind xyz
  return EMAP_OK;
}

static int ensure_tail(emap_ctx* ctx) {
  if (ctx->n_pts > ctx->tail_cap) {
    CK(hipStreamSynchronize(ctx->stream));
    if (ctx->tail_idx) CK(hipFree(ctx->tail_idx));
    if (ctx->tail_flags) CK(hipFree(ctx->tail_flags));
    ctx->tail_idx = nullptr; ctx->tail_flags = nullptr; ctx->tail_cap = 0;
    CK(hipMalloc((void**)&ctx->tail_idx, sizeof(int) * ctx->n_pts));
    CK(hipMalloc((void**)&ctx->tail_flags, ctx->n_pts));
    ctx->tail_cap = ctx->n_pts;
  }
  return EMAP_OK;
}

int emap_point_index(emap_ctx* ctx, const float R[9], const float t[3], int32_t* idx, uint8_t* valid, uint8_t* inside) {
  CKARG(ctx && R && t && idx && valid && inside, "null argument"); SF_CHECK();
  if (!ctx->pts && ctx->n_pts) { ctx->err = "no point cloud bound"; return EMAP_ERR_NO_POINTS; }
  CK(hipSetDevice(ctx->device));
  if (ctx->n_pts == 0) return EMAP_OK;
  int rc = ensure_tail(ctx); if (rc) return rc;
  launch_point_index(ctx->stream, ctx->kp, make_pose(ctx, R, t), ctx->pts, ctx->n_pts, ctx->stride, ctx->tail_idx, ctx->tail_flags);
  CK(hipGetLastError());
  CK(hipMemcpyAsync(idx, ctx->tail_idx, sizeof(int) * ctx->n_pts, hipMemcpyDeviceToHost, ctx->stream));
  CK(hipMemcpyAsync(valid, ctx->tail_flags, ctx->n_pts, hipMemcpyDeviceToHost, ctx->stream));
  CK(hipStreamSynchronize(ctx->stream));
  for (long i = 0; i < ctx->n_pts; ++i) { uint8_t f = valid[i]; valid[i] = f & 1; inside[i] = (f >> 1) & 1; }
  return EMAP_OK;
}

// ---- stages -----------------------------------------------------------------------------------------------
#define NEED_POINTS() do { if (!ctx->pts && ctx->n_pts) { ctx->err = "no point cloud bound"; return EMAP_ERR_NO_POINTS; } } while (0)

// bins of `sub` stacked 16 x 64 tiles: the smallest power of two that keeps the bin count within the LDS histogram
static int bin_sub(const emap_ctx* ctx) {
  const long tx = (ctx->prm.cell_n + 63) / 64;
  for (int sub = ctx->force_sub > 0 ? ctx->force_sub : 1; sub <= 64; sub *= 2) {
    const long ty = (ctx->strip.row_count + 16 * sub - 1) / (16 * sub);
    if (tx * ty <= BIN_MAX_T) return sub;
  }
  return 0;
}
static bool bins_possible(const emap_ctx* ctx) { return bin_sub(ctx) > 0; }
static int ensure_bins(emap_ctx* ctx, bool raybin) {
  const long n = ctx->n_pts;
  BinGeo& g = ctx->bg;
  g.sub = bin_sub(ctx);
  g.tiles_x = (ctx->prm.cell_n + 63) / 64; g.tiles_y = (ctx->strip.row_count + 16 * g.sub - 1) / (16 * g.sub); g.T = g.tiles_x * g.tiles_y; g.TB = g.T + 1;
  g.pitch = (g.TB + 3) & ~3; g.raybin = raybin ? 1 : 0;
  // strip contexts without a visibility pass: cheap ownership test + lane compaction in the point passes (emap_binned.hip)
  // ... and strips whose frame marches its rays BY RAY: a rank then needs exactly the points of its rows (the valid ones that are not
  // is_inside ride in the ray-only bin) -- the other ranks march the rest
  // ... but not for a cloud that was BUCKETED for this strip on the host (emap_upload_points_strip): nearly every point is an owned one
  // then, the ownership pre-test and the staging record (16 B written + read per point) buy nothing -- the plain kernels sort 2 M
  // points in 42 us where the strip variants take 52
  // (a by-ray frame keeps the strip variants: there the EXACT ownership test decides which rank marches a point's ray -- a point the
  // host-side superset kept for two neighbouring strips must not ride in both ranks' ray-only bins)
  ctx->bin_strip = ctx->strip.row_count < ctx->prm.cell_n && (!raybin || ctx->byray_frame) && !(ctx->pts_bucketed && !raybin);
  if (const char* e = getenv("EMAP_BIN_STRIP")) { if (atoi(e) == 0) ctx->bin_strip = false; }      // test / tuning hook
  // 32-byte records that carry the frame's semantic channels (frame_sem_begin): plain point passes only (a strip's staging records
  // hold no channels), never in front of a visibility pass (k_rays walks 16-byte records)
  ctx->bin_rs = (ctx->carry_want && !ctx->bin_strip && !raybin) ? 2 : 1;
  if (ctx->bin_rs == 1) ctx->carry.on = 0;
  // Blocks: ~4096 points each, but every block carries a row of the (block, tile) matrix through three passes (written, scanned,
  // read): keep the matrix (4 B x TB x B, x4) below the cloud's own traffic (12 B x n, x2) -- B <= n / (3 TB) -- without dropping
  // under one block per CU.  (8192^2 / 16 M points: 2048 blocks of 16385 bins were 537 MB of matrix traffic per frame.)
  long target = n >= 1000000 ? 4096 : 2048;
  if (const char* e = getenv("EMAP_BIN_CHUNK")) { long v = atol(e); if (v >= 256 && v <= 65536) target = v; }   // tuning knob
  long B = (n + target - 1) / target;
  long bmax = n / (3L * g.TB); if (bmax < 256) bmax = 256; if (bmax > BIN_MAX_B) bmax = BIN_MAX_B;
  if (B > bmax) B = bmax;
  if (B > 256) B -= B % 256;       // whole rounds of workgroups on the 256 CUs (325 blocks = one and a quarter rounds ran as long as 512)
  if (B < 1) B = 1;
  const long unit = ctx->bin_strip ? 4096 : 1024;   /* a multiple of every hist / scatter block size [x 4 loads in flight on strips] */
  long chunk = (n + B - 1) / B; chunk = ((chunk + unit - 1) / unit) * unit;
  g.B = (int)((n + chunk - 1) / chunk); if (g.B < 1) g.B = 1;
  g.chunk = chunk;
  const size_t hist_need = (size_t)g.pitch * (size_t)g.B;
  if (hist_need > ctx->bin_hist_cap) {
    CK(hipStreamSynchronize(ctx->stream));
    if (ctx->bin_hist) CK(hipFree(ctx->bin_hist));
    ctx->bin_hist = nullptr; ctx->bin_hist_cap = 0;
    CK(hipMalloc((void**)&ctx->bin_hist, sizeof(unsigned int) * hist_need));
    ctx->bin_hist_cap = hist_need;
  }
  if (!ctx->bin_tile_total) {
    CK(hipMalloc((void**)&ctx->bin_tile_total, sizeof(unsigned int) * (BIN_MAX_T + 2)));
    CK(hipMalloc((void**)&ctx->bin_tile_start, sizeof(unsigned int) * (BIN_MAX_T + 2)));
    CK(hipMalloc((void**)&ctx->bin_sync, sizeof(unsigned int) * EM_TICKET_WORDS));
    CK(hipMemsetAsync(ctx->bin_sync, 0, sizeof(unsigned int) * EM_TICKET_WORDS, ctx->stream));
  }
  if (!ctx->split_mem) {           // scratch of the heavy tiles (SplitView): 40 bytes per cell of SPLIT_MAX_SLOTS tiles + the lists, zero between frames
    const size_t cells = (size_t)SPLIT_MAX_SLOTS * SPLIT_CELLS;
    const size_t words = (size_t)(BIN_MAX_T + 2) + SPLIT_MAX_EXTRA + 8 + SPLIT_MAX_SLOTS + 4 * cells, bytes = 4 * words + 8 * 3 * cells + 64;
    void* m = nullptr;
    CK(hipMalloc(&m, bytes));
    if (hipMemsetAsync(m, 0, bytes, ctx->stream) != hipSuccess) { hipFree(m); ctx->err = "hipMemsetAsync(split scratch)"; return EMAP_ERR_HIP; }
    ctx->split_mem = m;
    unsigned long long* q = reinterpret_cast<unsigned long long*>(m);          // the 64-bit planes first (alignment)
    ctx->split.h = q; ctx->split.v = q + cells; ctx->split.latest = q + 2 * cells;
    unsigned int* u = reinterpret_cast<unsigned int*>(q + 3 * cells);
    ctx->split.pts = u; ctx->split.inl = u + cells; ctx->split.cnt = u + 2 * cells; ctx->split.out = u + 3 * cells; u += 4 * cells;
    ctx->split.tick = u; u += SPLIT_MAX_SLOTS;
    ctx->split.tile_slot = u; u += BIN_MAX_T + 2;
    ctx->split.extra = u; u += SPLIT_MAX_EXTRA;
    ctx->split.n_extra = u;
    CK(hipHostMalloc((void**)&ctx->split_need, 64, hipHostMallocMapped));
    ctx->split_need[0] = 0u; ctx->split_need[1] = 0u;      // [0]: heavy-tile parts (k_bin_scan), [1]: the ray kernel the map's state asks for (k_ray_apply)
    CK(hipHostGetDevicePointer((void**)&ctx->split.need_host, const_cast<unsigned int*>(ctx->split_need), 0));
    ctx->split.on = 0;
  }
  if (n * ctx->bin_rs > ctx->bin_cap) {
    CK(hipStreamSynchronize(ctx->stream));
    if (ctx->bin_recs) CK(hipFree(ctx->bin_recs));
    ctx->bin_recs = nullptr; ctx->bin_cap = 0;
    CK(hipMalloc((void**)&ctx->bin_recs, sizeof(BinRec) * n * ctx->bin_rs));
    ctx->bin_cap = n * ctx->bin_rs;
  }
  if (ctx->bin_strip && (long)g.B * chunk > ctx->bin_own_cap) {
    CK(hipStreamSynchronize(ctx->stream));
    if (ctx->bin_own) CK(hipFree(ctx->bin_own));
    if (ctx->bin_own_cnt) CK(hipFree(ctx->bin_own_cnt));
    ctx->bin_own = nullptr; ctx->bin_own_cnt = nullptr; ctx->bin_own_cap = 0;
    CK(hipMalloc((void**)&ctx->bin_own, 16 * (size_t)g.B * chunk));           // 16-byte staging records; only the owned share is ever touched
    CK(hipMalloc((void**)&ctx->bin_own_cnt, sizeof(unsigned int) * (size_t)BIN_MAX_B));
    ctx->bin_own_cap = (long)g.B * chunk;
  }
  return EMAP_OK;
}

int emap_set_scatter_mode(emap_ctx* ctx, int32_t mode) {
  const int m = mode & 0xff, sub = (mode >> 8) & 0xff;
  CKARG(ctx && m >= 0 && m <= 2 && (mode >> 16) == 0, "scatter mode: 0 auto, 1 atomic, 2 binned"); SF_CHECK();
  CKARG(sub == 0 || (sub <= 64 && (sub & (sub - 1)) == 0), "bin height factor must be a power of two <= 64");
  ctx->force_sub = sub;
  CKARG(m != 2 || bins_possible(ctx), "binned scatter: too many bins for the LDS histogram");
  ctx->scatter_mode = m;
  return EMAP_OK;
}

int emap_set_ray_mode(emap_ctx* ctx, int32_t mode) {
  CKARG(ctx && mode >= 0 && mode <= 2, "ray mode: 0 auto, 1 by row, 2 by ray");
  ctx->ray_mode = mode;
  return EMAP_OK;
}

// the host-mapped word a grid barrier that gave up sets (k_small_frame: emap_device.h)
static int ensure_barrier_word(emap_ctx* ctx) {
  if (ctx->sf_host) return EMAP_OK;
  if (!ctx->frame_save) CK(hipMalloc((void**)&ctx->frame_save, sizeof(FrameDev)));
  if (!ctx->sf_poison) { CK(hipMalloc((void**)&ctx->sf_poison, 64)); CK(hipMemsetAsync(ctx->sf_poison, 0, 64, ctx->stream)); }
  volatile unsigned int* h = nullptr;
  CK(hipHostMalloc((void**)&h, 64, hipHostMallocMapped));
  h[0] = 0u; h[1] = 0u;
  if (hipHostGetDevicePointer((void**)&ctx->sf_host_dev, const_cast<unsigned int*>(h), 0) != hipSuccess) { hipHostFree((void*)h); ctx->err = "hipHostGetDevicePointer"; return EMAP_ERR_HIP; }
  ctx->sf_host = h;
  return EMAP_OK;
}

static GateArgs gate_args(emap_ctx* ctx, double position_noise, double orientation_noise) {
  const emap_params& p = ctx->prm;
  GateArgs g; memset(&g, 0, sizeof g);
  g.enable = p.enable_drift_compensation; g.min_cnt = p.min_height_drift_cnt; g.max_drift = p.max_drift; g.alpha = (float)p.drift_compensation_alpha;
  g.noise_ok = (position_noise > p.position_noise_thresh) || (orientation_noise > p.orientation_noise_thresh);
  g.use_override = ctx->use_override ? 1 : 0; g.sum_override = ctx->sum_override; g.cnt_override = ctx->cnt_override;
  g.n_points = (unsigned int)ctx->n_pts;
  return g;
}

int emap_count(emap_ctx* ctx, const float R[9], const float t[3]) {
  CKARG(ctx && R && t, "null argument"); SF_CHECK(); NEED_POINTS();
  // a cloud bucketed for this strip holds the points of its rows under ONE pose and ONE row origin: the common entry point of every
  // frame (whole, sharded or staged) checks both -- a row shift in between moves the strip's logical rows under the kept points
  if (ctx->pts_bucketed) {
    CKARG(memcmp(ctx->bucket_R, R, sizeof ctx->bucket_R) == 0 && memcmp(ctx->bucket_t, t, sizeof ctx->bucket_t) == 0, "the bound cloud was bucketed for another pose (emap_upload_points_strip)");
    CKARG(ctx->bucket_org_r == ctx->kp.org_r, "the map's rows shifted since the bound cloud was bucketed for this strip (emap_upload_points_strip): upload it again");
  }
  if (!ctx->in_update) { ctx->gate_possible = true; ctx->carry_want = false; }          // the staged API always gathers the statistics and sorts plain 16-byte records
  ctx->bin_rs = 1;
  CK(hipSetDevice(ctx->device));
  // small clouds: two launches with global atomics win; large clouds: counting sort by tile + LDS reduction
  // (a cloud bucketed for a strip decides by the size of the WHOLE cloud: every rank of a sharded frame then takes the same path)
  const bool binned = ctx->n_pts > 0 && (ctx->scatter_mode == 2 || (ctx->scatter_mode == 0 && bins_possible(ctx) && ctx->n_pts_all >= 131072));   // measured crossover on MI355X: 60-135 k points for 202^2 .. 1024^2 maps
  ctx->frame_binned = binned;
  if (binned) {
    // the ray-only sort bin exists when the parameters enable the visibility pass (a staged emap_rays call on a context without it
    // marches the cloud in its own order instead of the sorted records)
    int rc = ensure_bins(ctx, ctx->prm.enable_visibility_cleanup != 0); if (rc) return rc;
    const bool tm = ctx->stage_timing && ctx->in_update;
    const Pose pose = make_pose(ctx, R, t);
    BinStg* own = ctx->bin_strip ? ctx->bin_own : nullptr;
    if (tm) CK(hipEventRecord(ctx->ev[ST_HIST], ctx->stream));
    launch_bin_hist(ctx->stream, ctx->kp, pose, ctx->bg, ctx->pts, ctx->n_pts, ctx->stride, ctx->bin_hist, own, ctx->bin_own_cnt);
    if (tm) CK(hipEventRecord(ctx->ev[ST_SCAN], ctx->stream));
    // heavy tiles are split only when k_tile_count runs in this frame (it leaves the per-cell counts k_tile_fuse's parts need)
    static const bool split_off = getenv("EMAP_SPLIT") && atoi(getenv("EMAP_SPLIT")) == 0;      // A/B and test hook
    ctx->split.on = ctx->gate_possible && !split_off ? 1 : 0;
    {   // extra workgroups of this frame's tile kernels: what the most recent finished scan asked for, + 25 %, and never fewer than a
        // STANDING POOL (EMAP_SPLIT_POOL, default 0 = none).  A heavy tile takes as many parts as it finds room for (k_bin_scan's
        // tail), so a pool would split the heavy tiles of a scene the host has not heard of yet.  Measured with a pool of 16 (round 5,
        // same box A/B): launches with extras are the SPLIT instantiations of the tile kernels, and those cost the uniform benchmarks
        // more than the pool saves -- cfg2 70.7 -> 73.9 us per frame (k_tile_count + 1.8, k_tile_fuse + 1.4 us), cfg5 3.39 -> 4.06 ms
        // (k_tile_semantic<true> alone + 0.55 ms) -- while the first terrain frame after uniform frames only went from 0.484 to 0.468
        // ms (steady 0.309: the pool's 16 parts go to whichever heavy tiles reserve first, not to the one that holds a third of the
        // cloud).  A sensor's near field is heavy in every frame, so the lag is one frame per scene change: the pool stays off.
      static const int cap_forced = getenv("EMAP_SPLIT_CAP") ? atoi(getenv("EMAP_SPLIT_CAP")) : -1;      // test hook
      static const int pool = getenv("EMAP_SPLIT_POOL") ? atoi(getenv("EMAP_SPLIT_POOL")) : EMAP_SPLIT_POOL_DEFAULT;      // A/B knob
      const unsigned int need = *ctx->split_need;
      long cap = need ? (long)need + need / 4 + 8 : 0;
      if (ctx->split.on && cap < pool) cap = pool;
      if (cap_forced >= 0) cap = cap_forced;
      if (cap > (long)SPLIT_MAX_EXTRA) cap = SPLIT_MAX_EXTRA;
      ctx->split.cap = (int)((cap + 7) & ~7L);
    }
    // a count stage whose fuse stage never came (staged API, an error in between): its slots are still filled.  The flag is only
    // cleared by what really empties them -- this memset or a fuse stage (fuse_impl) -- never by a frame that splits nothing: such a
    // frame in between must not make the NEXT split frame add its counts on top of stale ones.
    if (ctx->split.on && ctx->split_dirty) {
      CK(hipMemsetAsync(ctx->split_mem, 0, (size_t)SPLIT_MAX_SLOTS * SPLIT_CELLS * 40 + 4 * SPLIT_MAX_SLOTS, ctx->stream));
      ctx->split_dirty = false;
    }
    ctx->split_dirty = ctx->split_dirty || (ctx->split.on != 0 && ctx->split.cap > 0);      // (slots are only used by the SPLIT instantiations: extra workgroups in the launch)
    launch_bin_scan(ctx->stream, ctx->bg, ctx->bin_hist, ctx->bin_tile_total, ctx->bin_tile_start, ctx->bin_sync, ctx->split);
    if (tm) CK(hipEventRecord(ctx->ev[ST_SCATTER], ctx->stream));
    launch_bin_scatter(ctx->stream, ctx->kp, pose, ctx->bg, ctx->pts, ctx->n_pts, ctx->stride, ctx->bin_hist, ctx->bin_tile_start, ctx->bin_recs, own, ctx->bin_own_cnt,
                       ctx->chan, ctx->carry);
    if (tm) CK(hipEventRecord(ctx->ev[ST_GATE], ctx->stream));        // the "gate" stage = per-tile error sums + k_gate
    if (ctx->gate_possible) launch_tile_count(ctx->stream, ctx->kp, ctx->bg, ctx->bin_recs, ctx->bin_rs, ctx->bin_tile_start, ctx->cells, ctx->slots, ctx->split, ctx->n_pts);
  } else {
    ctx->carry.on = 0;
    if (ctx->stage_timing && ctx->in_update)
      for (int e = ST_HIST; e <= ST_SCATTER; ++e) CK(hipEventRecord(ctx->ev[e], ctx->stream));
    // whole frames (emap_update) on this path: the drift gate rides in k_count's last workgroup, one launch less in a chain of six
    static const bool fold_off = getenv("EMAP_GATE_FOLD") && atoi(getenv("EMAP_GATE_FOLD")) == 0;      // A/B and test hook
    GateArgs ga;
    const bool fold = ctx->fold_gate && !fold_off && ctx->cnt_sync;
    if (fold) { ctx->use_override = false; ga = gate_args(ctx, ctx->pos_noise, ctx->ori_noise); }
    ctx->gate_folded = launch_count(ctx->stream, ctx->kp, make_pose(ctx, R, t), ctx->pts, ctx->n_pts, ctx->stride, ctx->cells, ctx->acc, ctx->slots,
                                    fold ? &ga : nullptr, ctx->frame, ctx->cnt_sync);
    if (ctx->stage_timing && ctx->in_update) CK(hipEventRecord(ctx->ev[ST_GATE], ctx->stream));
  }
  CK(hipGetLastError());
  return EMAP_OK;
}

static int gate_impl(emap_ctx* ctx, double position_noise, double orientation_noise, int reduce_only, double* dev_out,
                     const double* dev_totals) {
  CK(hipSetDevice(ctx->device));
  launch_gate(ctx->stream, gate_args(ctx, position_noise, orientation_noise), ctx->slots, ctx->frame, reduce_only, dev_out, dev_totals);
  CK(hipGetLastError());
  ctx->committed = false;
  return EMAP_OK;
}

int emap_set_drift_inputs(emap_ctx* ctx, double position_noise, double orientation_noise, const double* err_sum_override,
                          const uint32_t* err_cnt_override) {
  CKARG(ctx, "null ctx"); SF_CHECK();
  ctx->pos_noise = position_noise; ctx->ori_noise = orientation_noise;
  ctx->use_override = err_sum_override && err_cnt_override;
  if (ctx->use_override) { ctx->sum_override = *err_sum_override; ctx->cnt_override = *err_cnt_override; }
  return gate_impl(ctx, position_noise, orientation_noise, 0, nullptr, nullptr);
}

int emap_drift_sums_to_device(emap_ctx* ctx, double* dev_out2) {
  CKARG(ctx && dev_out2, "null argument"); SF_CHECK();
  ctx->use_override = false;
  return gate_impl(ctx, 0.0, 0.0, 1, dev_out2, nullptr);
}

int emap_set_drift_inputs_device(emap_ctx* ctx, double position_noise, double orientation_noise, const double* dev_totals2) {
  CKARG(ctx && dev_totals2, "null argument"); SF_CHECK();
  ctx->use_override = false;
  return gate_impl(ctx, position_noise, orientation_noise, 0, nullptr, dev_totals2);
}

int emap_local_drift_sums(emap_ctx* ctx, double* err_sum, uint32_t* err_cnt) {
  CKARG(ctx && err_sum && err_cnt, "null argument"); SF_CHECK();
  ctx->use_override = false;
  int rc = gate_impl(ctx, 0.0, 0.0, 1, nullptr, nullptr);
  if (rc) return rc;
  emap_stats st;
  rc = emap_get_stats(ctx, &st);
  if (rc) return rc;
  *err_sum = st.err_sum; *err_cnt = st.err_cnt;
  return EMAP_OK;
}

static int fuse_impl(emap_ctx* ctx, const float R[9], const float t[3], bool fuse_average = false, bool rays = false) {
  NEED_POINTS();
  CK(hipSetDevice(ctx->device));
  if (ctx->frame_binned) {
    if (fuse_average && rays && !ctx->inl_plane) {        // each buffer on its own: a failed second allocation must not leave the first one "done"
      unsigned int* pl = nullptr;
      CK(hipMalloc((void**)&pl, sizeof(unsigned int) * ctx->ncells_alloc));
      if (hipMemsetAsync(pl, 0, sizeof(unsigned int) * ctx->ncells_alloc, ctx->stream) != hipSuccess) { hipFree(pl); ctx->err = "hipMemsetAsync(inl_plane)"; return EMAP_ERR_HIP; }
      ctx->inl_plane = pl;
    }
    if (fuse_average && rays && !ctx->ray_thr)
      CK(hipMalloc((void**)&ctx->ray_thr, sizeof(float) * (size_t)((ctx->strip.row_count + 7) / 8 + 2) * ((ctx->prm.cell_n + 7) / 8)));
    if (fuse_average && rays && ((ctx->kp.org_c | ctx->prm.cell_n) & 63) != 0 && !ctx->inert_zero)     // unaligned columns: the tile kernel ORs its ballots into the logical bitmap
      CK(hipMemsetAsync(ctx->inert, 0, sizeof(unsigned long long) * ((size_t)ctx->strip.row_count * ((ctx->prm.cell_n + 63) / 64)), ctx->stream));
    if (fuse_average && rays) ctx->inert_zero = false;
    // a carrying frame fuses its semantic channels inside the tile kernel (emap_binned.hip: k_tile_fuse<.., SEM>) unless the launch
    // holds heavy-tile parts: then the stand-alone kernel follows (frame_sem_finish), reading the channels from the records all the same
    SemMini sm; memset(&sm, 0, sizeof sm);
    const bool merged = ctx->fsem_set && ctx->carry.on && bin_fuse_takes_semantics(ctx->split, fuse_average, rays, ctx->bin_rs);
    if (merged) {
      const SemSpec& S = ctx->fsem;
      sm.n_sum = S.n_sum; sm.n_col = S.n_col; sm.alpha = S.alpha; sm.sem = ctx->sem; sm.plane = ctx->ncells_alloc;
      for (int q = 0; q < S.n_sum; ++q) { sm.slot[q] = S.sum_chan[q] - ctx->carry.c0; sm.layer[q] = S.sum_layer[q]; sm.kind[q] = S.sum_kind[q]; }
      if (S.n_col) { sm.col_slot = S.col_chan[0] - ctx->carry.c0; sm.col_layer = S.col_layer[0]; }
    }
    ctx->fsem_merged = merged;
    if (ctx->fsem_set) ctx->update_path |= merged ? 4 : (ctx->carry.on ? 8 : 0);      // (emap_last_update_path)
    launch_bin_fuse(ctx->stream, ctx->kp, ctx->bg, ctx->bin_recs, ctx->bin_rs, ctx->bin_tile_start, ctx->cells, ctx->acc, ctx->frame, fuse_average, rays,
                    (merged && !ctx->fsem_keep_counts) ? nullptr : ctx->cnt_plane, ctx->inert, ctx->inl_plane, ctx->ray_thr, ctx->ov_args, ctx->gate_fold, ctx->split, ctx->n_pts,
                    merged ? &sm : nullptr);
    if (ctx->split.on) ctx->split_dirty = false;        // k_tile_fuse's parts cleared their slots (a frame that splits nothing leaves an older count stage's slots as they are)
    ctx->gate_fold.mode = 0;
    if (fuse_average) ctx->kp.mv.n = 0;   // every owned cell rewritten: pending map shifts are in memory now
    CK(hipGetLastError());
    return EMAP_OK;
  }
  launch_fuse(ctx->stream, ctx->kp, make_pose(ctx, R, t), ctx->pts, ctx->n_pts, ctx->stride, ctx->cells, ctx->acc, ctx->frame);
  CK(hipGetLastError());
  return EMAP_OK;
}
int emap_fuse(emap_ctx* ctx, const float R[9], const float t[3]) { CKARG(ctx && R && t, "null argument"); SF_CHECK(); return fuse_impl(ctx, R, t); }

// fuse + commit + average_map in one call for frames without a visibility pass (one tile kernel on the binned path)
int emap_fuse_average(emap_ctx* ctx, const float R[9], const float t[3]) {
  CKARG(ctx && R && t, "null argument"); SF_CHECK();
  const bool fused = ctx->frame_binned;
  int rc = fuse_impl(ctx, R, t, fused);
  if (rc) return rc;
  if (!fused) { launch_average(ctx->stream, ctx->kp, ctx->cells, ctx->acc, ctx->accr, ctx->frame, ctx->committed, false, ctx->cnt_plane, OverlapArgs{}); ctx->kp.mv.n = 0; }
  ctx->committed = false;
  CK(hipGetLastError());
  return EMAP_OK;
}

int emap_commit(emap_ctx* ctx) {
  CKARG(ctx, "null ctx"); SF_CHECK();
  CK(hipSetDevice(ctx->device));
  if (!ctx->committed) {
    launch_commit(ctx->stream, ctx->kp, ctx->cells, ctx->acc, ctx->frame, ctx->inert);
    ctx->committed = true; ctx->inert_zero = false;
    ctx->kp.mv.n = 0;                 // k_commit rewrote every owned cell: pending map shifts are in memory now
  }
  CK(hipGetLastError());
  return EMAP_OK;
}

int emap_rays(emap_ctx* ctx, const float R[9], const float t[3]) {
  CKARG(ctx && R && t, "null argument"); SF_CHECK(); NEED_POINTS();
  CK(hipSetDevice(ctx->device));
  CKARG(ctx->committed || ctx->rays_fused, "emap_rays needs emap_commit first (rays read snapshot S1)");
  CKARG(!ctx->pts_bucketed, "a bucketed cloud (emap_upload_points_strip) cannot march its rays by row: every valid point marches a ray through every strip");
  // newmap[3]: the tile kernel's dense plane, or the high halves of AccF::pts_inl (5 x u64 records) on the staged / atomic path
  const unsigned int* inl = ctx->rays_fused ? ctx->inl_plane : reinterpret_cast<const unsigned int*>(ctx->acc) + 1;
  const bool sorted = ctx->frame_binned && ctx->bg.raybin;      // the sorted records hold EVERY valid point only with the ray-only bin
  char* const ab = reinterpret_cast<char*>(ctx->accr);
  const AccRView av = {ab + offsetof(AccR, dec), ab + offsetof(AccR, hits), ab + offsetof(AccR, upper_key), (int)sizeof(AccR), (int)sizeof(AccR), (int)sizeof(AccR), 0};
  KP kr = ctx->kp;
  kr.nlag = ctx->nlag_ready ? 1 : 0;                     // sharded frame after a row shift: the row-aligned copy of the normal planes (normal_exchange)
  kr.ray_pref = ctx->split_need ? (int)ctx->split_need[1] : 0;      // (host-mapped word written by k_ray_apply: FrameDev::ray_class)
  launch_rays(ctx->stream, kr, make_pose(ctx, R, t), ctx->rt, ctx->pts, ctx->n_pts, ctx->stride, ctx->cells, av,
              ctx->nlag_ready ? ctx->nlag_buf : ctx->normal, ctx->nlag_ready ? (long)ctx->strip.row_count * ctx->prm.cell_n : ctx->ncells_alloc,
              ctx->frame, ctx->want_ray_stats, ctx->inert, inl, ctx->rays_fused ? 1 : (int)(sizeof(AccF) / 4),
              ctx->rays_fused ? ctx->ray_thr : nullptr,
              sorted ? reinterpret_cast<const unsigned int*>(ctx->bin_recs) : nullptr,        // march in tile-sorted order
              sorted ? ctx->bin_tile_start + ctx->bg.TB : nullptr);
  CK(hipGetLastError());
  return EMAP_OK;
}

int emap_average(emap_ctx* ctx) {
  CKARG(ctx, "null ctx"); SF_CHECK();
  CK(hipSetDevice(ctx->device));
  launch_average(ctx->stream, ctx->kp, ctx->cells, ctx->acc, ctx->accr, ctx->frame, ctx->committed, true, ctx->cnt_plane, OverlapArgs{});
  ctx->committed = false; ctx->kp.mv.n = 0;       // (uncommitted: k_average wrote the pending map shifts out itself)
  CK(hipGetLastError());
  return EMAP_OK;
}

static OverlapArgs overlap_args(const emap_ctx* ctx, float t_z, bool on) {
  const emap_params& p = ctx->prm;
  int cell_range = (int)(p.overlap_clear_range_xy / p.resolution);     // elevation_mapping.py:88-91
  if (cell_range < 0) cell_range = 0; if (cell_range > p.cell_n) cell_range = p.cell_n;
  OverlapArgs o; memset(&o, 0, sizeof o);
  o.on = on ? 1 : 0;
  o.cmin = p.cell_n / 2 - cell_range / 2; o.cmax = p.cell_n / 2 + cell_range / 2;
  o.hmin = t_z - (float)p.overlap_clear_range_z; o.hmax = t_z + (float)p.overlap_clear_range_z;
  return o;
}

static int overlap_clear_impl(emap_ctx* ctx, float t_z);
int emap_overlap_clear(emap_ctx* ctx, float t_z) { CKARG(ctx, "null ctx"); SF_CHECK(); return overlap_clear_impl(ctx, t_z); }
static int overlap_clear_impl(emap_ctx* ctx, float t_z) {      // (inside a frame: no SF_CHECK -- a small frame just issued must not be waited for)
  CK(hipSetDevice(ctx->device));
  FLUSH();
  const OverlapArgs o = overlap_args(ctx, t_z, true);
  launch_overlap(ctx->stream, ctx->kp, ctx->cells, o.cmin, o.cmax, o.hmin, o.hmax);
  CK(hipGetLastError());
  return EMAP_OK;
}

// Stencil stages.  The kernels work on LOGICAL rows; a strip owns the physical rows [row_begin, row_begin + row_count), i.e. the
// logical rows ls + j (mod cell_n), j = 0 .. row_count-1, with ls = (row_begin - org_r) mod cell_n: one or two logical intervals.
static void post_rows(emap_ctx* ctx, int nj, const int* j0, const int* j1, int stage, int tile_rows = 0) {      // outputs for the strip's rows j0[k] <= j < j1[k]
  const int C = ctx->prm.cell_n;
  const int ls = ((ctx->strip.row_begin - ctx->kp.org_r) % C + C) % C;
  int sb[4], se[4], n = 0;
  if (ctx->strip.row_count == C && nj == 1 && j0[0] == 0 && j1[0] == C) {      // the whole map: one interval, tiles aligned to logical row 0
    sb[0] = 0; se[0] = C; n = 1; nj = 0;
  }
  for (int k = 0; k < nj; ++k) {
    if (j1[k] <= j0[k]) continue;
    const int b = ls + j0[k], e = ls + j1[k];                             // unwrapped logical rows
    if (e <= C) { sb[n] = b; se[n] = e; n++; }
    else if (b >= C) { sb[n] = b - C; se[n] = e - C; n++; }
    else { sb[n] = b; se[n] = C; n++; sb[n] = 0; se[n] = e - C; n++; }    // the circular seam lies inside: never inside one tile
  }
  launch_post(ctx->stream, ctx->kp, ctx->prm.w1, ctx->prm.w2, ctx->prm.w3, ctx->prm.w_out, ctx->cells, ctx->trav_in, ctx->normal,
              ctx->ncells_alloc, ctx->prm.dilation_size, n, sb, se, stage, tile_rows);
}

int emap_dilate(emap_ctx* ctx) {          // dilation_filter_kernel alone: traversability_input (k_post, stage 1)
  CKARG(ctx, "null ctx"); SF_CHECK();
  CK(hipSetDevice(ctx->device));
  FLUSH();
  { const int j0 = 0, j1 = ctx->strip.row_count; post_rows(ctx, 1, &j0, &j1, 1); }
  ctx->torg_r = ctx->kp.org_r; ctx->torg_c = ctx->kp.org_c;
  CK(hipGetLastError());
  return EMAP_OK;
}

int emap_traversability_normals(emap_ctx* ctx) { return emap_post_part(ctx, 0); }     // (recomputes the dilation it consumes: same values)

// dilation + traversability + normals in one launch.  part: 0 = whole strip,
// 1 = only the rows that do not depend on halo rows (can run while the halo exchange is in flight),
// 2 = the remaining (boundary) rows.
static int post_part_impl(emap_ctx* ctx, int32_t part);
int emap_post_part(emap_ctx* ctx, int32_t part) { CKARG(ctx && part >= 0 && part <= 2, "bad argument"); SF_CHECK(); return post_part_impl(ctx, part); }
static int post_part_impl(emap_ctx* ctx, int32_t part) {      // (inside a frame: no SF_CHECK, see overlap_clear_impl)
  CK(hipSetDevice(ctx->device));
  FLUSH();
  const int n = ctx->strip.row_count;
  const int reach = ctx->prm.dilation_size + 4;                  // rows a stencil looks beyond its own (dilation + filter + row wrap)
  const int lo = reach < n ? reach : n, hi = n - reach > lo ? n - reach : lo;      // interior rows [lo, hi)
  const int jb0[2] = {0, hi}, je0[2] = {lo, n}, z = 0;
  if (part == 0) post_rows(ctx, 1, &z, &n, 0);
  else if (part == 1) post_rows(ctx, 1, &lo, &hi, 0);
  else { int tr = 4; while (tr < reach && tr < 32) tr *= 2;      // the boundary bands are `reach` rows high: tiles of that height, not the strip's 32-row tiles
         post_rows(ctx, 2, jb0, je0, 0, tr); }          // at most 3 logical intervals: the seam lies in one of the two boundary bands
  if (part != 1) { ctx->kp.norg_r = ctx->torg_r = ctx->kp.org_r; ctx->kp.norg_c = ctx->torg_c = ctx->kp.org_c; }   // outputs carry the current origin
  CK(hipGetLastError());
  return EMAP_OK;
}
int emap_post(emap_ctx* ctx) { return emap_post_part(ctx, 0); }

int emap_update_variance(emap_ctx* ctx) { CKARG(ctx, "null ctx"); SF_CHECK(); CK(hipSetDevice(ctx->device)); launch_var_time(ctx->stream, ctx->kp, ctx->cells, 1, 0); ctx->kp.mv.n = 0; CK(hipGetLastError()); return EMAP_OK; }
int emap_update_time(emap_ctx* ctx) { CKARG(ctx, "null ctx"); SF_CHECK(); CK(hipSetDevice(ctx->device)); launch_var_time(ctx->stream, ctx->kp, ctx->cells, 0, 1); ctx->kp.mv.n = 0; CK(hipGetLastError()); return EMAP_OK; }

int emap_get_stats(emap_ctx* ctx, emap_stats* out) {
  CKARG(ctx && out, "null argument"); SF_CHECK();
  CK(hipSetDevice(ctx->device));
  FrameDev f;
  CK(hipMemcpyAsync(&f, ctx->frame, sizeof f, hipMemcpyDeviceToHost, ctx->stream));
  CK(hipStreamSynchronize(ctx->stream));
  out->err_sum = (double)f.err_sum_fix / EM_SCALE_E; out->err_cnt = (uint32_t)f.err_cnt; out->gate_fired = f.gate_fired;
  out->mean_error = f.mean_error; out->additive_mean_error = f.additive_mean_error; out->shift = f.shift;
  out->n_points = f.n_points; out->ray_visits = f.ray_visits;
  return EMAP_OK;
}

int emap_update(emap_ctx* ctx, const float R[9], const float t[3], double position_noise, double orientation_noise, emap_stats* stats) {
  const int rc = update_impl(ctx, R, t, position_noise, orientation_noise, stats);
  if (ctx) { ctx->fsem_set = false; ctx->carry_want = false; }      // the declared semantic fusion belongs to ONE frame, whatever became of it
  return rc;
}
static int update_impl(emap_ctx* ctx, const float R[9], const float t[3], double position_noise, double orientation_noise, emap_stats* stats) {
  CKARG(ctx && R && t, "null argument"); NEED_POINTS();
  CK(hipSetDevice(ctx->device));
  const emap_params& p = ctx->prm;
  const bool tm = ctx->stage_timing;
  int rc;
#define STAGE(i) do { if (tm) CK(hipEventRecord(ctx->ev[i], ctx->stream)); } while (0)
  const bool rays_on = p.enable_visibility_cleanup != 0;
  // robot scale (small clouds on small maps: the atomic path): count, gate, fuse and commit / average in ONE launch
  static const bool sf_env_off = getenv("EMAP_SMALL_FRAME") && atoi(getenv("EMAP_SMALL_FRAME")) == 0;      // A/B and test hook
  const bool atomic_path = !(ctx->scatter_mode == 2 || (ctx->scatter_mode == 0 && bins_possible(ctx) && ctx->n_pts_all >= 131072));      // (emap_count's choice)
  // (not with a visibility pass or a declared semantic fusion behind it: an ABORTED launch must leave nothing for the rest of the frame to
  // act on -- the stencil launch that follows is a pure function of the map -- until sf_recover has re-run the frame)
  const int sf_grid = (!sf_env_off && !ctx->sf_off && !ctx->sf_redo && atomic_path && ctx->cnt_sync && !ctx->pts_bucketed && !rays_on && !ctx->fsem_set) ? small_frame_grid(ctx->kp, ctx->n_pts) : 0;
  if (sf_grid == 0) SF_CHECK();                 // (a frame on any other path: the small frames in flight are settled first)
  if ((rc = frame_sem_begin(ctx, rays_on))) return rc;
  // clear_overlap_map rides on the kernel that rewrites the cells last (tile kernel / k_average, or k_ray_apply after a visibility pass)
  ctx->ov_args = overlap_args(ctx, t[2], p.enable_overlap_clearance != 0);
  const bool ov_folded = ctx->ov_args.on != 0;
  // drift gate of elevation_mapping.py:346-349: with compensation off or both noises below their thresholds it cannot fire
  ctx->gate_possible = p.enable_drift_compensation && (position_noise > p.position_noise_thresh || orientation_noise > p.orientation_noise_thresh);
  ctx->pos_noise = position_noise; ctx->ori_noise = orientation_noise;
  bool fused_small = false;
  if (sf_grid > 0) {
    if ((rc = ensure_barrier_word(ctx))) { ctx->ov_args.on = 0; return rc; }
    sf_forget_applied(ctx);
    if (ctx->sf_count == emap_ctx::SF_RING || ctx->sf_epoch >= 0x7ffffff0u) {      // no room to remember another frame (or the epochs wrap): settle the ones in flight
      if ((rc = sf_settle(ctx))) { ctx->ov_args.on = 0; return rc; }
      if (ctx->sf_epoch >= 0x7ffffff0u) { CK(hipStreamSynchronize(ctx->stream)); ctx->sf_epoch = 0u; ctx->sf_host[0] = 0u; }
    }
    for (int e = ST_HIST; e <= ST_SCATTER; ++e) STAGE(e);
    ctx->frame_binned = false; ctx->use_override = false;
    ++ctx->sf_epoch;
    {   // what the launch stands for, should a barrier be aborted (sf_recover)
      emap_ctx::SfFrame& f = ctx->sf_ring[(ctx->sf_head + ctx->sf_count) % emap_ctx::SF_RING];
      f.epoch = ctx->sf_epoch; memcpy(f.R, R, sizeof f.R); memcpy(f.t, t, sizeof f.t); f.pn = position_noise; f.on = orientation_noise; f.mv = ctx->kp.mv;
      f.pts = ctx->pts; f.n_pts = ctx->n_pts; f.n_pts_all = ctx->n_pts_all; f.stride = ctx->stride; f.chan = ctx->chan; f.n_cols = ctx->n_cols;
      ++ctx->sf_count;
    }
    unsigned int spin = SF_SPIN_DEFAULT; int test_abort = 0;      // test hooks (read per frame: a test toggles them)
    if (const char* e = getenv("EMAP_SF_SPIN_LIMIT")) { const long v = atol(e); if (v >= 1 && v <= (1L << 24)) spin = (unsigned int)v; }
    if (const char* e = getenv("EMAP_SF_TEST_ABORT")) test_abort = atoi(e);
    launch_small_frame(ctx->stream, sf_grid, ctx->kp, make_pose(ctx, R, t), ctx->pts, ctx->n_pts, ctx->stride, ctx->cells, ctx->acc,
                       ctx->cnt_plane, ctx->ov_args, gate_args(ctx, position_noise, orientation_noise), ctx->frame, ctx->frame_save, ctx->slots,
                       ctx->cnt_sync, ctx->cnt_sync + 2048, ctx->sf_host_dev, ctx->sf_poison, ctx->sf_epoch, spin, test_abort);
    CK(hipGetLastError());
    fused_small = true;
    STAGE(ST_GATE);
  } else {
    ctx->in_update = true;
    ctx->fold_gate = true; ctx->gate_folded = false;
    rc = emap_count(ctx, R, t);                 // records ST_HIST / ST_SCAN / ST_SCATTER itself
    ctx->in_update = false; ctx->fold_gate = false;
    if (rc) { ctx->ov_args.on = 0; return rc; }            // (emap_count also recorded ST_GATE: the stage starts with the per-tile error sums)
    if (ctx->gate_folded) { ctx->committed = false; ctx->gate_folded = false; }      // (small clouds: k_count's last workgroup was the gate)
    else if ((rc = emap_set_drift_inputs(ctx, position_noise, orientation_noise, nullptr, nullptr))) { ctx->ov_args.on = 0; return rc; }
  }
  ctx->update_path = fused_small ? 2 : (ctx->frame_binned ? 1 : 0);
  STAGE(ST_FUSE);
  // binned scatter: fusion, commit and averaging happen in ONE tile kernel; with the visibility pass it also writes the inert
  // bitmap and the inlier plane, and the ray effects are applied by k_ray_apply ("average" stage) afterwards
  const bool fused_avg = ctx->frame_binned;
  rc = fused_small ? EMAP_OK : fuse_impl(ctx, R, t, fused_avg, rays_on);
  const OverlapArgs ov = ctx->ov_args; ctx->ov_args.on = 0;
  if (rc) return rc;
  STAGE(ST_COMMIT);
  ctx->rays_fused = fused_avg && rays_on;
  if (rays_on) {
    if (!fused_avg && (rc = emap_commit(ctx))) return rc;
    STAGE(ST_RAYS);
    rc = emap_rays(ctx, R, t);
    if (rc) { ctx->rays_fused = false; return rc; }
  } else STAGE(ST_RAYS);
  STAGE(ST_AVERAGE);
  if (fused_small) ctx->kp.mv.n = 0;                        // (k_small_frame committed and averaged: every cell rewritten)
  else if (!fused_avg) { launch_average(ctx->stream, ctx->kp, ctx->cells, ctx->acc, ctx->accr, ctx->frame, ctx->committed, rays_on, ctx->cnt_plane, ov); ctx->kp.mv.n = 0; }
  else if (rays_on) { launch_ray_apply(ctx->stream, ctx->kp, ctx->cells, ctx->accr, ctx->inert, ov, ctx->frame, ctx->split.need_host ? ctx->split.need_host + 1 : nullptr, ctx->ray_par ^= 1); ctx->inert_zero = true; }
  ctx->committed = false; ctx->rays_fused = false;
  CK(hipGetLastError());
  if ((rc = frame_sem_finish(ctx, R, t))) return rc;      // semantic_map.update_layers_pointcloud (elevation_mapping.py:368) -- unless the tile kernel fused the channels itself
  STAGE(ST_OVERLAP);
  if (p.enable_overlap_clearance && !ov_folded && (rc = overlap_clear_impl(ctx, t[2]))) return rc;
  STAGE(ST_POST);
  if ((rc = post_part_impl(ctx, 0))) return rc;
  STAGE(ST_N);
#undef STAGE
  if (tm) {
    CK(hipEventSynchronize(ctx->ev[ST_N]));
    for (int i = 0; i < ST_N; ++i) CK(hipEventElapsedTime(&ctx->stage_ms[i], ctx->ev[i], ctx->ev[i + 1]));
  }
  if (stats) return emap_get_stats(ctx, stats);
  return EMAP_OK;
}

// ---- state access -------------------------------------------------------------------------------------------
// Views are (row_count, cell_n) host arrays in LOGICAL order (k_plane_view / k_get_plane): row j = logical row j of a full map,
// or the j-th logical row of a strip (emap_strip_logical_begin), columns logical.
static int plane_view(emap_ctx* ctx, int plane, float* host, bool to_device) { SF_CHECK();
  const size_t bytes = sizeof(float) * (size_t)ctx->strip.row_count * ctx->prm.cell_n;
  FLUSH();
  if (to_device) CK(hipMemcpyAsync(ctx->scratch, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  if (plane < 7) {
    if (to_device) launch_set_plane(ctx->stream, ctx->kp, ctx->cells, plane, ctx->scratch);
    else launch_get_plane(ctx->stream, ctx->kp, ctx->cells, plane, ctx->scratch);
  } else if (plane == EMAP_PLANE_TRAV_INPUT) launch_plane_view(ctx->stream, ctx->kp, ctx->torg_r, ctx->torg_c, ctx->trav_in, ctx->scratch, to_device);
  else launch_plane_view(ctx->stream, ctx->kp, ctx->kp.norg_r, ctx->kp.norg_c, ctx->normal + (long)(plane - EMAP_PLANE_NORMAL_X) * ctx->ncells_alloc,
                         ctx->scratch, to_device);
  CK(hipGetLastError());
  if (!to_device) CK(hipMemcpyAsync(host, ctx->scratch, bytes, hipMemcpyDeviceToHost, ctx->stream));
  CK(hipStreamSynchronize(ctx->stream));
  return EMAP_OK;
}

int emap_get_layer(emap_ctx* ctx, int plane, float* host_out) {
  CKARG(ctx && host_out && plane >= 0 && plane < EMAP_PLANE_COUNT, "bad argument");
  CK(hipSetDevice(ctx->device));
  return plane_view(ctx, plane, host_out, false);
}

int emap_publish_layer(emap_ctx* ctx, int32_t kind, float center_z, int32_t use_only_above_for_upper_bound, float* host_out) {
  CKARG(ctx && host_out && kind >= 0 && kind <= 8, "bad argument"); SF_CHECK();
  CKARG(ctx->strip.halo_rows == 0 && ctx->strip.row_count == ctx->prm.cell_n, "emap_publish_layer: single-strip contexts only");
  CK(hipSetDevice(ctx->device));
  FLUSH();
  const long M = ctx->prm.cell_n - 2;
  launch_publish(ctx->stream, ctx->kp, ctx->cells, ctx->normal, ctx->ncells_alloc, kind, center_z, use_only_above_for_upper_bound, ctx->scratch);
  CK(hipGetLastError());
  CK(hipMemcpyAsync(host_out, ctx->scratch, sizeof(float) * (size_t)(M * M), hipMemcpyDeviceToHost, ctx->stream));
  CK(hipStreamSynchronize(ctx->stream));
  return EMAP_OK;
}

int emap_set_layer(emap_ctx* ctx, int plane, const float* host_in) {
  CKARG(ctx && host_in && plane >= 0 && plane < EMAP_PLANE_COUNT, "bad argument");
  CK(hipSetDevice(ctx->device));
  return plane_view(ctx, plane, const_cast<float*>(host_in), true);
}

int emap_strip_logical_begin(emap_ctx* ctx, int32_t* logical_row) {
  CKARG(ctx && logical_row, "null argument"); SF_CHECK();
  const int C = ctx->prm.cell_n;
  *logical_row = ctx->strip.row_count == C ? 0 : ((ctx->strip.row_begin - ctx->kp.org_r) % C + C) % C;
  return EMAP_OK;
}

// ElevationMap.shift_map_xy / shift_map_z (EM/elevation_mapping.py:200-226) WITHOUT moving data: the roll becomes a rotation of the
// circular origin, the reset of the entering band and the z offset become a pending entry that the frame kernels replay (cell_now)
// and the next full rewrite of the cells writes out; only the semantic layers clear their entering band here (O(border) bytes).
// The normal planes and traversability_input keep their own origin: the reference does not shift them.
int emap_shift(emap_ctx* ctx, int32_t shift_rows, int32_t shift_cols, float dz) {
  CKARG(ctx, "null ctx"); SF_CHECK();
  CK(hipSetDevice(ctx->device));
  const int C = ctx->prm.cell_n;
  CKARG(shift_rows > -(1 << 20) && shift_rows < (1 << 20) && shift_cols > -(1 << 20) && shift_cols < (1 << 20), "absurd shift");   // (a shift of cell_n or more resets every cell: the replay predicate covers it)
  if (shift_rows == 0 && shift_cols == 0 && dz == 0.f) return EMAP_OK;
  if (ctx->kp.mv.n == EM_MAX_MOVES) FLUSH();       // more moves than the replay list holds before the next frame: one full pass
  KP& k = ctx->kp;
  k.org_r = ((k.org_r - shift_rows) % C + C) % C;   // roll: new logical r holds old logical r - shift  =>  physical = r - shift + org
  k.org_c = ((k.org_c - shift_cols) % C + C) % C;
  Moves& mv = k.mv;
  mv.org_r[mv.n] = k.org_r; mv.org_c[mv.n] = k.org_c; mv.sr[mv.n] = shift_rows; mv.sc[mv.n] = shift_cols; mv.dz[mv.n] = dz;
  mv.n++;
  if (ctx->sem_layers > 0 && (shift_rows != 0 || shift_cols != 0)) {   // SemanticMap.shift_map_xy (semantic_map.py:127-136): roll + zero pad
    launch_band_clear(ctx->stream, k, ctx->sem, ctx->sem_layers, ctx->ncells_alloc, shift_rows, shift_cols);
    if (ctx->sem_alpha) launch_band_clear(ctx->stream, k, ctx->sem_alpha, ctx->sem_layers, ctx->ncells_alloc, shift_rows, shift_cols);   // new_map too (:135-136)
    CK(hipGetLastError());
  }
  return EMAP_OK;
}

// ---- RGB / semantic layers (EM/semantic_map.py, EM/fusion/pointcloud_{average,class_average,color}.py) -------------
int emap_semantic_configure(emap_ctx* ctx, int32_t n_layers) {
  CKARG(ctx && n_layers >= 0 && n_layers <= 64, "bad layer count"); SF_CHECK();
  CK(hipSetDevice(ctx->device));
  const long n = ctx->ncells_alloc;
  if (!ctx->cnt_plane) {
    CK(hipMalloc((void**)&ctx->cnt_plane, sizeof(unsigned int) * n));
    CK(hipMemsetAsync(ctx->cnt_plane, 0, sizeof(unsigned int) * n, ctx->stream));
    CK(hipMalloc((void**)&ctx->sem_col, sizeof(unsigned int) * n * 13));
    CK(hipMemsetAsync(ctx->sem_col, 0, sizeof(unsigned int) * n * 13, ctx->stream));
  }
  if (n_layers > ctx->sem_layers) {      // grow, keeping existing layers (SemanticMap.add_layer, semantic_map.py:80-97)
    float* ns = nullptr; double* nq = nullptr;
    CK(hipMalloc((void**)&ns, sizeof(float) * n * n_layers));
    CK(hipMalloc((void**)&nq, sizeof(double) * n * n_layers));
    CK(hipMemsetAsync(ns, 0, sizeof(float) * n * n_layers, ctx->stream));
    CK(hipMemsetAsync(nq, 0, sizeof(double) * n * n_layers, ctx->stream));
    if (ctx->sem_layers > 0) CK(hipMemcpyAsync(ns, ctx->sem, sizeof(float) * n * ctx->sem_layers, hipMemcpyDeviceToDevice, ctx->stream));
    if (ctx->sem_alpha) {
      float* na = nullptr;
      CK(hipMalloc((void**)&na, sizeof(float) * n * n_layers));
      CK(hipMemsetAsync(na, 0, sizeof(float) * n * n_layers, ctx->stream));
      CK(hipMemcpyAsync(na, ctx->sem_alpha, sizeof(float) * n * ctx->sem_layers, hipMemcpyDeviceToDevice, ctx->stream));
      CK(hipStreamSynchronize(ctx->stream));
      CK(hipFree(ctx->sem_alpha)); ctx->sem_alpha = na;
    }
    CK(hipStreamSynchronize(ctx->stream));
    if (ctx->sem) CK(hipFree(ctx->sem));
    if (ctx->sem_sums) CK(hipFree(ctx->sem_sums));
    ctx->sem = ns; ctx->sem_sums = nq; ctx->sem_layers = n_layers;
  }
  return EMAP_OK;
}

static int ensure_alpha(emap_ctx* ctx) {
  if (ctx->sem_alpha || ctx->sem_layers == 0) return EMAP_OK;
  CK(hipMalloc((void**)&ctx->sem_alpha, sizeof(float) * ctx->ncells_alloc * ctx->sem_layers));
  CK(hipMemsetAsync(ctx->sem_alpha, 0, sizeof(float) * ctx->ncells_alloc * ctx->sem_layers, ctx->stream));
  return EMAP_OK;
}

// emap_sem_spec -> SemSpec, checked against the bound cloud and the configured layers
static int sem_spec_checked(emap_ctx* ctx, const emap_sem_spec* spec, SemSpec* out) {
  CKARG(spec->n_sum >= 0 && spec->n_sum <= SEM_MAX_CH && spec->n_col >= 0 && spec->n_col <= 4, "too many channels");
  CKARG(ctx->cnt_plane, "emap_semantic_configure must be called before the frame (the average pass records the counts)");
  for (int k = 0; k < spec->n_sum; ++k)
    CKARG(spec->sum_layer[k] >= 0 && spec->sum_layer[k] < ctx->sem_layers && spec->sum_chan[k] >= 3 && spec->sum_chan[k] < ctx->n_cols, "bad channel/layer index");
  for (int k = 0; k < spec->n_col; ++k)
    CKARG(spec->col_layer[k] >= 0 && spec->col_layer[k] < ctx->sem_layers && spec->col_chan[k] >= 3 && spec->col_chan[k] < ctx->n_cols, "bad colour channel/layer index");
  SemSpec& S = *out; memset(&S, 0, sizeof S); memcpy(&S, spec, sizeof *spec);
  int nk[4] = {0, 0, 0, 0};
  for (int k = 0; k < S.n_sum; ++k) { CKARG(S.sum_kind[k] >= 0 && S.sum_kind[k] <= 3, "bad fusion kind"); nk[S.sum_kind[k]]++; }
  int seen[4] = {0, 0, 0, 0};
  for (int k = 0; k < S.n_sum; ++k) {
    const int kind = S.sum_kind[k];
    S.sum_K[k] = kind >= 2 ? nk[kind] : 1;
    S.sum_q[k] = kind >= 2 ? seen[kind]++ : 0;
  }
  S.any_bayes = nk[2] > 0;
  // class_bayesian / bayesian_inference reproduce the reference's launch decode (element exists while id * K + q < N): the GLOBAL point
  // index and cloud size -- a cloud bucketed for a strip renumbers its points
  CKARG(!(ctx->pts_bucketed && (nk[2] + nk[3]) > 0), "a bucketed cloud (emap_upload_points_strip) cannot feed class_bayesian / bayesian_inference fusions: they decode the global point index");
  return EMAP_OK;
}
static int semantic_update_impl(emap_ctx* ctx, const float R[9], const float t[3], const SemSpec& S);

int emap_frame_semantics(emap_ctx* ctx, const emap_sem_spec* spec_or_null, int32_t keep_counts) {
  CKARG(ctx, "null ctx");
  ctx->fsem_set = false; ctx->fsem_keep_counts = keep_counts != 0;
  if (!spec_or_null || spec_or_null->n_sum + spec_or_null->n_col == 0) return EMAP_OK;
  CKARG(spec_or_null->n_sum >= 0 && spec_or_null->n_sum <= SEM_MAX_CH && spec_or_null->n_col >= 0 && spec_or_null->n_col <= 4, "too many channels");
  memset(&ctx->fsem, 0, sizeof ctx->fsem); memcpy(&ctx->fsem, spec_or_null, sizeof *spec_or_null);      // (checked against the cloud bound when the frame starts)
  ctx->fsem_set = true;
  return EMAP_OK;
}
// Start of a whole frame: check the declared fusion against the bound cloud and decide whether the frame's sort may CARRY the channels
// (32-byte records): kinds average / class_average and at most one colour channel, at most four channel columns, all within four
// consecutive columns of the cloud; no visibility pass (k_rays walks 16-byte records).  emap_count then settles it (tile-binned
// path, plain point passes: ensure_bins).  EMAP_SEM_CARRY=0: never (A/B and test hook).
static int frame_sem_begin(emap_ctx* ctx, bool rays_on) {
  ctx->carry_want = false; ctx->fsem_merged = false; memset(&ctx->carry, 0, sizeof ctx->carry);
  if (!ctx->fsem_set) return EMAP_OK;
  emap_sem_spec raw; memcpy(&raw, &ctx->fsem, sizeof raw);
  SemSpec S;
  int rc = sem_spec_checked(ctx, &raw, &S);
  if (rc) { ctx->fsem_set = false; return rc; }
  ctx->fsem = S;
  if (S.any_bayes && (rc = ensure_alpha(ctx))) { ctx->fsem_set = false; return rc; }
  static const bool off = getenv("EMAP_SEM_CARRY") && atoi(getenv("EMAP_SEM_CARRY")) == 0;
  if (off || rays_on || S.n_sum > 4 || S.n_col > 1 || S.n_sum + S.n_col == 0) return EMAP_OK;
  int cmin = 1 << 30, cmax = -1;
  for (int k = 0; k < S.n_sum; ++k) { if (S.sum_kind[k] > 1) return EMAP_OK; cmin = std::min(cmin, S.sum_chan[k]); cmax = std::max(cmax, S.sum_chan[k]); }
  for (int k = 0; k < S.n_col; ++k) { cmin = std::min(cmin, S.col_chan[k]); cmax = std::max(cmax, S.col_chan[k]); }
  if (cmax - cmin >= 4) return EMAP_OK;
  ctx->carry_want = true;
  ctx->carry.c0 = cmin; ctx->carry.ncols = ctx->n_cols;
  // the de-interleaved (N, 4) channel matrix of an uploaded cloud: one aligned 16-byte load per point
  ctx->carry.on = (ctx->chan.stride == 4 && ctx->chan.col0 == cmin && ((uintptr_t)ctx->chan.p & 15) == 0) ? 2 : 1;
  return EMAP_OK;
}
// End of the frame's fusion stages: whatever the tile kernel did not fuse itself runs now (the frame's counts are in cnt_plane)
static int frame_sem_finish(emap_ctx* ctx, const float R[9], const float t[3]) {
  if (!ctx->fsem_set) return EMAP_OK;
  ctx->fsem_set = false; ctx->carry_want = false;
  if (ctx->fsem_merged) { ctx->fsem_merged = false; return EMAP_OK; }
  return semantic_update_impl(ctx, R, t, ctx->fsem);
}

int emap_semantic_update(emap_ctx* ctx, const float R[9], const float t[3], const emap_sem_spec* spec) {
  CKARG(ctx && R && t && spec, "null argument"); SF_CHECK(); NEED_POINTS();
  SemSpec S;
  int rc = sem_spec_checked(ctx, spec, &S);
  if (rc) return rc;
  return semantic_update_impl(ctx, R, t, S);
}
static int semantic_update_impl(emap_ctx* ctx, const float R[9], const float t[3], const SemSpec& S) {
  CK(hipSetDevice(ctx->device));
  if (S.any_bayes) { int rc = ensure_alpha(ctx); if (rc) return rc; }
  if (ctx->frame_binned) {   // the frame's tile-sorted records are still valid: reduce in LDS, no global atomics
    if (ctx->split.on && ctx->split.cap > 0 && !ctx->sem_split_mem && sem_split_possible(S)) {      // the frame listed heavy tiles: their semantic sums are shared too
      void* m = nullptr;
      CK(hipMalloc(&m, sem_split_bytes(SEM_SPLIT_SLOTS)));
      if (hipMemsetAsync(m, 0, sem_split_bytes(SEM_SPLIT_SLOTS), ctx->stream) != hipSuccess) { hipFree(m); ctx->err = "hipMemsetAsync(semantic split scratch)"; return EMAP_ERR_HIP; }
      ctx->sem_split_mem = m;
    }
    launch_tile_semantic(ctx->stream, ctx->kp, ctx->bg, S, ctx->bin_recs, ctx->bin_rs, ctx->carry.on ? ctx->carry.c0 : -1, ctx->bin_tile_start, ctx->chan, ctx->n_pts,
                         ctx->cnt_plane, ctx->sem, ctx->sem_alpha, ctx->ncells_alloc, ctx->split, ctx->sem_split_mem, SEM_SPLIT_SLOTS);
    CK(hipGetLastError());
    return EMAP_OK;
  }
  launch_sem_points(ctx->stream, ctx->kp, make_pose(ctx, R, t), S, ctx->pts, ctx->n_pts, ctx->stride, ctx->chan, ctx->sem_sums, ctx->sem_col, ctx->ncells_alloc);
  launch_sem_finalize(ctx->stream, ctx->kp, S, ctx->cnt_plane, ctx->sem_sums, ctx->sem_col, ctx->sem, ctx->sem_alpha, ctx->ncells_alloc);
  CK(hipGetLastError());
  return EMAP_OK;
}

// ---- the reference's semantic kernel factories on caller arrays (EM/kernels/custom_semantic_kernels.py) ------------------------
// Host arrays in, host arrays out (the factories of the compat package hand NumPy arrays over); device buffers live for the call.
namespace {
struct DevBuf {           // a device copy of a host array for the duration of a call
  void* d = nullptr; size_t bytes = 0; void* back = nullptr;
  hipError_t put(const void* host, size_t n, hipStream_t s, void* write_back) {
    bytes = n; back = write_back;
    if (!n) return hipSuccess;
    hipError_t e = hipMalloc(&d, n);
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(d, host, n, hipMemcpyHostToDevice, s);
  }
  hipError_t get(hipStream_t s) { return (back && bytes) ? hipMemcpyAsync(back, d, bytes, hipMemcpyDeviceToHost, s) : hipSuccess; }
  ~DevBuf() { if (d) hipFree(d); }
};
}  // namespace

int emap_semantic_accumulate(emap_ctx* ctx, int32_t op, const float* points, int64_t n_rows, int32_t stride, const int32_t* pcl_chan,
                             const int32_t* map_lay, int32_t n_ch, int64_t size, int64_t cells, void* newmap_inout, int32_t newmap_layers,
                             const float* max_pt, const int32_t* max_id, int32_t n_max) {
  CKARG(ctx && points && newmap_inout && op >= 0 && op <= 4, "bad argument");
  CKARG(n_rows >= 0 && stride >= 3 && n_ch >= 1 && n_ch <= 64 && size >= 0 && cells > 0 && newmap_layers >= 1, "bad shape");
  CKARG(op == 2 ? (max_pt && max_id && n_max >= 1 && size <= n_rows) : (pcl_chan && map_lay && size <= n_rows * (int64_t)n_ch), "bad channel description / size");
  // The index CONTENTS decide which plane / column a thread touches: check them on the host before anything is
  // launched -- a bad index from a caller of the kernel factories must not write into other device memory of the shared context.
  if (op == 2) {
    for (int64_t k = 0; k < size * (int64_t)n_max; ++k) CKARG(max_id[k] >= 0 && max_id[k] < newmap_layers, "sum_max: class id outside the planes of newmap");
  } else {
    for (int k = 0; k < n_ch; ++k) {
      CKARG(pcl_chan[k] >= 0 && pcl_chan[k] < stride, "pcl channel index outside the point rows");
      CKARG(op == 1 || op == 4 || (map_lay[k] >= 0 && map_lay[k] < newmap_layers), "map layer index outside the planes of newmap");
    }
    CKARG(op != 1 || n_ch <= newmap_layers, "sum_compact: newmap needs one plane per channel");
    CKARG(op != 4 || 3 * n_ch + 1 <= newmap_layers, "add_color: the colour map needs 3 n_ch + 1 planes");
  }
  CK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  DevBuf P, PC, ML, MP, MI, NM;
  CK(P.put(points, sizeof(float) * (size_t)n_rows * stride, st, nullptr));
  if (pcl_chan) { CK(PC.put(pcl_chan, sizeof(int) * n_ch, st, nullptr)); CK(ML.put(map_lay, sizeof(int) * n_ch, st, nullptr)); }
  if (op == 2) { CK(MP.put(max_pt, sizeof(float) * (size_t)n_rows * n_max, st, nullptr)); CK(MI.put(max_id, sizeof(int) * (size_t)n_rows * n_max, st, nullptr)); }
  CK(NM.put(newmap_inout, 4 * (size_t)newmap_layers * cells, st, newmap_inout));
  SemRaw A; memset(&A, 0, sizeof A);
  A.op = op; A.stride = stride; A.K = n_ch; A.n_max = n_max; A.size = size; A.cells = cells;
  launch_semraw_acc(st, A, (const float*)P.d, (const int*)PC.d, (const int*)ML.d, (const float*)MP.d, (const int*)MI.d, (float*)NM.d, (unsigned int*)NM.d);
  CK(hipGetLastError());
  CK(NM.get(st));
  CK(hipStreamSynchronize(st));
  return EMAP_OK;
}

int emap_semantic_finalize(emap_ctx* ctx, int32_t op, void* newmap_inout, int32_t newmap_layers, const int32_t* map_lay, int32_t n_ch, int64_t size,
                           int64_t cells, const float* new_elmap3, const float* sum_mean, int32_t sum_layers, float* map_inout, int32_t map_layers, double alpha) {
  CKARG(ctx && newmap_inout && map_lay && map_inout && op >= 0 && op <= 3, "bad argument");
  CKARG(n_ch >= 1 && n_ch <= 64 && size >= 0 && size <= cells * (int64_t)n_ch && cells > 0 && newmap_layers >= 1 && map_layers >= 1, "bad shape");
  CKARG(op == 3 || new_elmap3, "the accepted-point counts (new_elmap plane 2) are needed");
  CKARG(op != 2 || (sum_mean && sum_layers >= n_ch), "bayesian_inference needs sum_mean");
  for (int k = 0; k < n_ch; ++k)      // (colour: newmap is the colour map, indexed by channel; every other op indexes newmap AND map with map_lay)
    CKARG(map_lay[k] >= 0 && map_lay[k] < map_layers && (op == 3 || map_lay[k] < newmap_layers), "map layer index outside the planes of map / newmap");
  CKARG(op != 3 || 3 * n_ch + 1 <= newmap_layers, "color_average: the colour map needs 3 n_ch + 1 planes");
  CK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  DevBuf NM, ML, EL, SM, MP;
  CK(NM.put(newmap_inout, 4 * (size_t)newmap_layers * cells, st, op == 2 ? newmap_inout : nullptr));
  CK(ML.put(map_lay, sizeof(int) * n_ch, st, nullptr));
  if (new_elmap3) CK(EL.put(new_elmap3, sizeof(float) * 3 * (size_t)cells, st, nullptr));
  if (sum_mean) CK(SM.put(sum_mean, sizeof(float) * (size_t)sum_layers * cells, st, nullptr));
  CK(MP.put(map_inout, sizeof(float) * (size_t)map_layers * cells, st, map_inout));
  SemRaw A; memset(&A, 0, sizeof A);
  A.op = op; A.K = n_ch; A.size = size; A.cells = cells; A.alpha = alpha;
  launch_semraw_fin(st, A, (float*)NM.d, (const unsigned int*)NM.d, (const int*)ML.d, (const float*)EL.d, (const float*)SM.d, (float*)MP.d);
  CK(hipGetLastError());
  CK(NM.get(st)); CK(MP.get(st));
  CK(hipStreamSynchronize(st));
  return EMAP_OK;
}

// ---- pointcloud_class_max (EM/fusion/pointcloud_class_max.py:80-126; kernels: emap_semantic.hip) ------------------------------
// The class-id planes (the reference's elements_to_shift["id_max"]) live in the layers' persistent planes (sem_alpha): they move
// with the map and read back through emap_semantic_get_alpha as uint32 bit patterns.  `prev_unique` is the fusion's unique_id array
// of the previous frame ([0] before the first: :59).  The reference gathers unique_id[id_max] (:85) -- the planes hold class VALUES
// and the table is indexed with them; positions beyond the table wrap around (CuPy's integer-array indexing), reproduced here and
// pinned by the reference's own statements executed from its file (tests/golden/class_max_ref66.npz).
int emap_semantic_class_max(emap_ctx* ctx, const float R[9], const float t[3], int32_t n_ch, const int32_t* chan, const int32_t* layer,
                            const uint32_t* prev_unique, int32_t n_prev, uint32_t* unique_out, int32_t unique_cap, int32_t* n_unique_out) {
  CKARG(ctx && R && t && chan && layer && unique_out && n_unique_out && n_ch >= 1 && n_ch <= 8 && n_prev >= 0 && (n_prev == 0 || prev_unique), "bad argument"); SF_CHECK();
  // the id set and the planes zeroed between two layers are properties of the WHOLE map: a row strip would take them from its own rows
  CKARG(ctx->strip.halo_rows == 0 && ctx->strip.row_count == ctx->prm.cell_n, "class_max: single-strip contexts only");
  NEED_POINTS();
  for (int k = 0; k < n_ch; ++k)
    CKARG(layer[k] >= 0 && layer[k] < ctx->sem_layers && chan[k] >= 3 && chan[k] < ctx->n_cols, "bad channel/layer index");
  CK(hipSetDevice(ctx->device));
  { int rc = ensure_alpha(ctx); if (rc) return rc; }
  hipStream_t st = ctx->stream;
  CmaxSpec S; memset(&S, 0, sizeof S);
  S.n = n_ch; for (int k = 0; k < n_ch; ++k) { S.chan[k] = chan[k]; S.layer[k] = layer[k]; }
  const long plane = ctx->ncells_alloc;
  // (1) the ids of this frame and of the map
  unsigned char* d_seen = nullptr;
  CK(hipMalloc((void**)&d_seen, 2 * 65536));
  struct Free { void* p; ~Free() { if (p) hipFree(p); } } f_seen{d_seen};
  CK(hipMemsetAsync(d_seen, 0, 2 * 65536, st));
  launch_cmax_ids(st, ctx->kp, S, ctx->chan, ctx->n_pts, ctx->sem_alpha, plane, d_seen, d_seen + 65536);
  CK(hipGetLastError());
  std::vector<unsigned char> seen(2 * 65536);
  CK(hipMemcpyAsync(seen.data(), d_seen, 2 * 65536, hipMemcpyDeviceToHost, st));
  CK(hipStreamSynchronize(st));
  const uint32_t zero_id = 0;
  if (n_prev == 0) { prev_unique = &zero_id; n_prev = 1; }
  std::vector<unsigned char> in_set(65536, 0);
  for (int v = 0; v < 65536; ++v) {
    if (seen[v]) in_set[v] = 1;                                                            // unique(pt_id) (:84)
    if (seen[65536 + v] && prev_unique[v % n_prev] < 65536u) in_set[prev_unique[v % n_prev]] = 1;      // unique(unique_id[id_max]) (:85): positions beyond the table wrap around, like CuPy's integer-array gather
  }
  std::vector<uint32_t> uniq; std::vector<int> pos(65536, 0);
  for (int v = 0; v < 65536; ++v) if (in_set[v]) { pos[v] = (int)uniq.size(); uniq.push_back((uint32_t)v); }
  const int U = (int)uniq.size();
  CKARG(U <= unique_cap, "unique_out too small for the class ids of this frame");
  const size_t sum_bytes = sizeof(long long) * (size_t)U * plane;
  CKARG(sum_bytes <= ((size_t)16 << 30), "class_max: (classes x cells) probability sums beyond 16 GB");
  // (2) probability sums per (class, cell)
  long long* d_sum = nullptr; int* d_pos = nullptr; unsigned int* d_uniq = nullptr; unsigned char* d_flags = nullptr; float* d_new = nullptr;
  CK(hipMalloc((void**)&d_sum, sum_bytes)); Free f_sum{d_sum};
  CK(hipMalloc((void**)&d_pos, sizeof(int) * 65536)); Free f_pos{d_pos};
  CK(hipMalloc((void**)&d_uniq, sizeof(unsigned int) * U)); Free f_uniq{d_uniq};
  CK(hipMalloc((void**)&d_flags, 2 * (size_t)U)); Free f_flags{d_flags};
  CK(hipMalloc((void**)&d_new, sizeof(float) * (size_t)n_ch * plane)); Free f_new{d_new};
  CK(hipMemsetAsync(d_sum, 0, sum_bytes, st));
  CK(hipMemsetAsync(d_flags, 0, 2 * (size_t)U, st));
  CK(hipMemsetAsync(d_new, 0, sizeof(float) * (size_t)n_ch * plane, st));
  CK(hipMemcpyAsync(d_pos, pos.data(), sizeof(int) * 65536, hipMemcpyHostToDevice, st));
  CK(hipMemcpyAsync(d_uniq, uniq.data(), sizeof(unsigned int) * U, hipMemcpyHostToDevice, st));
  launch_cmax_sum(st, ctx->kp, make_pose(ctx, R, t), S, ctx->pts, ctx->n_pts, ctx->stride, ctx->chan, d_pos, d_sum, plane);
  // (3) + (4): per layer the maximum and its class, the winners' planes zeroed in between; then the normalisation
  launch_cmax_select(st, ctx->kp, S, U, d_sum, plane, d_flags, d_flags + U, d_uniq, d_new, ctx->sem_alpha, ctx->sem);
  CK(hipGetLastError());
  CK(hipStreamSynchronize(st));              // the host vectors and the temporaries end with the call
  memcpy(unique_out, uniq.data(), sizeof(uint32_t) * U);
  *n_unique_out = U;
  return EMAP_OK;
}

static int sem_view(emap_ctx* ctx, float* planes, int32_t layer, float* host, bool to_device) { SF_CHECK();   // semantic layers share the map's origin
  const size_t bytes = sizeof(float) * (size_t)ctx->strip.row_count * ctx->prm.cell_n;
  if (to_device) CK(hipMemcpyAsync(ctx->scratch, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  launch_plane_view(ctx->stream, ctx->kp, ctx->kp.org_r, ctx->kp.org_c, planes + (long)layer * ctx->ncells_alloc, ctx->scratch, to_device);
  CK(hipGetLastError());
  if (!to_device) CK(hipMemcpyAsync(host, ctx->scratch, bytes, hipMemcpyDeviceToHost, ctx->stream));
  CK(hipStreamSynchronize(ctx->stream));
  return EMAP_OK;
}
int emap_semantic_get_alpha(emap_ctx* ctx, int32_t layer, float* host_out) {
  CKARG(ctx && host_out && layer >= 0 && layer < ctx->sem_layers, "bad argument");
  CK(hipSetDevice(ctx->device));
  int rc = ensure_alpha(ctx); if (rc) return rc;
  return sem_view(ctx, ctx->sem_alpha, layer, host_out, false);
}
int emap_semantic_set_alpha(emap_ctx* ctx, int32_t layer, const float* host_in) {
  CKARG(ctx && host_in && layer >= 0 && layer < ctx->sem_layers, "bad argument");
  CK(hipSetDevice(ctx->device));
  int rc = ensure_alpha(ctx); if (rc) return rc;
  return sem_view(ctx, ctx->sem_alpha, layer, const_cast<float*>(host_in), true);
}

int emap_semantic_get_layer(emap_ctx* ctx, int32_t layer, float* host_out) {
  CKARG(ctx && host_out && layer >= 0 && layer < ctx->sem_layers, "bad argument");
  CK(hipSetDevice(ctx->device));
  return sem_view(ctx, ctx->sem, layer, host_out, false);
}
int emap_semantic_set_layer(emap_ctx* ctx, int32_t layer, const float* host_in) {
  CKARG(ctx && host_in && layer >= 0 && layer < ctx->sem_layers, "bad argument");
  CK(hipSetDevice(ctx->device));
  return sem_view(ctx, ctx->sem, layer, const_cast<float*>(host_in), true);
}
int emap_semantic_clear(emap_ctx* ctx) {
  CKARG(ctx, "null ctx"); SF_CHECK();
  CK(hipSetDevice(ctx->device));
  if (ctx->sem_layers > 0) CK(hipMemsetAsync(ctx->sem, 0, sizeof(float) * ctx->ncells_alloc * ctx->sem_layers, ctx->stream));
  return EMAP_OK;
}

// ---- MinFilter plugin (EM/plugins/min_filter.py:84-118) on caller-provided planes ------------------------------------
static int minmax_filter(emap_ctx* ctx, const float* host_elevation, const float* host_valid, int32_t dilation_size, int32_t iteration_n,
                         float* host_out, int32_t* sweeps_run, bool is_max) {
  CKARG(ctx && host_out && ((host_elevation && host_valid) || (!host_elevation && !host_valid)), "null argument");
  CKARG(dilation_size >= 0 && dilation_size <= 32 && iteration_n >= 0 && iteration_n <= 4096, "bad filter size / iteration count");
  CKARG(ctx->strip.halo_rows == 0 && ctx->strip.row_count == ctx->prm.cell_n, "emap_min_filter: single-strip contexts only");
  CK(hipSetDevice(ctx->device));
  const int C = ctx->prm.cell_n; const size_t L = (size_t)C * C, bytes = L * sizeof(float);
  int rc = plugin_scratch(ctx, 5, iteration_n + 1); if (rc) return rc;
  float* buf = ctx->plug_buf; unsigned int* cnt = ctx->plug_cnt;
  float *orig = buf, *v0 = buf + L, *m0 = buf + 2 * L, *v1 = buf + 3 * L, *m1 = buf + 4 * L;
  if (host_elevation) {
    CK(hipMemcpyAsync(orig, host_valid, bytes, hipMemcpyHostToDevice, ctx->stream));
    CK(hipMemcpyAsync(v0, host_elevation, bytes, hipMemcpyHostToDevice, ctx->stream));
  } else {                         // the map's own planes, de-interleaved on the device (no PCIe round trip of the inputs)
    FLUSH();
    launch_get_plane(ctx->stream, ctx->kp, ctx->cells, 2, orig);
    launch_get_plane(ctx->stream, ctx->kp, ctx->cells, 0, v0);
  }
  CK(hipMemcpyAsync(m0, orig, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  CK(hipMemsetAsync(cnt, 0, sizeof(unsigned int) * (iteration_n + 1), ctx->stream));
  for (int k = 0; k < iteration_n; ++k) {
    launch_min_sweep(ctx->stream, C, dilation_size, orig, (k & 1) ? v1 : v0, (k & 1) ? m1 : m0, (k & 1) ? v0 : v1, (k & 1) ? m0 : m1,
                     k > 0 ? cnt + (k - 1) : nullptr, cnt + k, is_max);
    CK(hipGetLastError());
  }
  const float* fv = (iteration_n & 1) ? v1 : v0; const float* fm = (iteration_n & 1) ? m1 : m0;
  std::vector<float> mask(L);
  std::vector<unsigned int> hc(iteration_n + 1);
  CK(hipMemcpyAsync(host_out, fv, bytes, hipMemcpyDeviceToHost, ctx->stream));
  CK(hipMemcpyAsync(mask.data(), fm, bytes, hipMemcpyDeviceToHost, ctx->stream));
  CK(hipMemcpyAsync(hc.data(), cnt, sizeof(unsigned int) * (iteration_n + 1), hipMemcpyDeviceToHost, ctx->stream));
  CK(hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < L; ++i) if (!(mask[i] > 0.5f)) host_out[i] = NAN;     // cp.where(mask > 0.5, filtered, nan), :116
  if (sweeps_run) { int n = 0; for (int k = 0; k < iteration_n; ++k) { ++n; if (hc[k] == 0) break; } *sweeps_run = n; }
  return EMAP_OK;
}

int emap_min_filter(emap_ctx* ctx, const float* host_elevation, const float* host_valid, int32_t dilation_size, int32_t iteration_n,
                    float* host_out, int32_t* sweeps_run) {
  return minmax_filter(ctx, host_elevation, host_valid, dilation_size, iteration_n, host_out, sweeps_run, false);
}
int emap_max_filter(emap_ctx* ctx, const float* host_elevation, const float* host_valid, int32_t dilation_size, int32_t iteration_n,
                    float* host_out, int32_t* sweeps_run) {
  return minmax_filter(ctx, host_elevation, host_valid, dilation_size, iteration_n, host_out, sweeps_run, true);
}

// ---- SmoothFilter plugin (EM/plugins/smooth_filter.py:56-58): `passes` x uniform_filter(size=3) on a host plane ---------------
int emap_smooth_filter(emap_ctx* ctx, const float* host_in, int32_t passes, float* host_out) {
  CKARG(ctx && host_in && host_out && passes >= 1 && passes <= 64, "bad argument");
  CK(hipSetDevice(ctx->device));
  const int C = ctx->prm.cell_n; const size_t L = (size_t)C * C, bytes = L * sizeof(float);
  float* buf = nullptr;
  CK(hipMalloc((void**)&buf, bytes * 2));
  float *a = buf, *b = buf + L;
  hipError_t e = hipMemcpyAsync(a, host_in, bytes, hipMemcpyHostToDevice, ctx->stream);
  for (int k = 0; k < passes && e == hipSuccess; ++k) { launch_box3(ctx->stream, C, a, b); e = hipGetLastError(); float* t = a; a = b; b = t; }
  if (e == hipSuccess) e = hipMemcpyAsync(host_out, a, bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(buf);
  if (e != hipSuccess) { ctx->err = std::string("emap_smooth_filter: ") + hipGetErrorString(e); return EMAP_ERR_HIP; }
  return EMAP_OK;
}

// ---- Erosion plugin (EM/plugins/erosion.py:96-104): cv2.erode with a k x k rectangle, `iterations` times, on a host plane ------
int emap_erode(emap_ctx* ctx, const float* host_in, int32_t kernel_size, int32_t iterations, float* host_out) {
  CKARG(ctx && host_in && host_out && kernel_size >= 1 && kernel_size <= 63 && iterations >= 0 && iterations <= 256, "bad argument");
  CK(hipSetDevice(ctx->device));
  const int C = ctx->prm.cell_n; const size_t L = (size_t)C * C, bytes = L * sizeof(float);
  float* buf = nullptr;
  CK(hipMalloc((void**)&buf, bytes * 2));
  float *a = buf, *b = buf + L;
  hipError_t e = hipMemcpyAsync(a, host_in, bytes, hipMemcpyHostToDevice, ctx->stream);
  for (int k = 0; k < iterations && e == hipSuccess; ++k) { launch_erode(ctx->stream, C, kernel_size, a, b); e = hipGetLastError(); float* t = a; a = b; b = t; }
  if (e == hipSuccess) e = hipMemcpyAsync(host_out, a, bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(buf);
  if (e != hipSuccess) { ctx->err = std::string("emap_erode: ") + hipGetErrorString(e); return EMAP_ERR_HIP; }
  return EMAP_OK;
}

// ---- Inpainting plugin substitute (see emap_semantic.hip): fill the pixels with known == 0 of an 8-bit image ---------
int emap_inpaint_u8(emap_ctx* ctx, const float* host_image, const float* host_known, int32_t max_sweeps, float* host_out,
                    int32_t* sweeps_run) {
  CKARG(ctx && host_image && host_known && host_out && max_sweeps >= 0 && max_sweeps <= 65536, "bad argument");
  CK(hipSetDevice(ctx->device));
  const int C = ctx->prm.cell_n; const size_t L = (size_t)C * C, bytes = L * sizeof(float);
  const int BATCH = 16;                      // sweeps between two looks at the unfilled counter
  int rc = plugin_scratch(ctx, 4, BATCH + 1); if (rc) return rc;
  float* buf = ctx->plug_buf; unsigned int* cnt = ctx->plug_cnt;
  float *v0 = buf, *m0 = buf + L, *v1 = buf + 2 * L, *m1 = buf + 3 * L;
  CK(hipMemcpyAsync(v0, host_image, bytes, hipMemcpyHostToDevice, ctx->stream));
  CK(hipMemcpyAsync(m0, host_known, bytes, hipMemcpyHostToDevice, ctx->stream));
  int done = 0; bool filled = false;
  std::vector<unsigned int> hc(BATCH + 1);
  while (done < max_sweeps && !filled) {      // the front usually closes after a few sweeps: stop launching once nothing is left
    const int nb = max_sweeps - done < BATCH ? max_sweeps - done : BATCH;
    CK(hipMemsetAsync(cnt, 0, sizeof(unsigned int) * (BATCH + 1), ctx->stream));
    for (int k = 0; k < nb; ++k) {
      const int g = done + k;
      launch_inpaint_sweep(ctx->stream, C, (g & 1) ? v1 : v0, (g & 1) ? m1 : m0, (g & 1) ? v0 : v1, (g & 1) ? m0 : m1,
                           k > 0 ? cnt + (k - 1) : nullptr, cnt + k);
      CK(hipGetLastError());
    }
    CK(hipMemcpyAsync(hc.data(), cnt, sizeof(unsigned int) * (BATCH + 1), hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    int used = nb;
    for (int k = 0; k < nb; ++k) if (hc[k] == 0) { used = k + 1; filled = true; break; }
    done += nb;                               // sweeps after the closing one are copies: the result is the latest buffer either way
    if (sweeps_run) *sweeps_run = done - nb + used;
  }
  if (sweeps_run && max_sweeps == 0) *sweeps_run = 0;
  CK(hipMemcpyAsync(host_out, (done & 1) ? v1 : v0, bytes, hipMemcpyDeviceToHost, ctx->stream));
  CK(hipStreamSynchronize(ctx->stream));
  return EMAP_OK;
}

// ---- camera path (EM/elevation_mapping.py:468-562, EM/kernels/custom_image_kernels.py) -------------------------------
int emap_image_correspondence(emap_ctx* ctx, float x1, float y1, float z1, const float P[12], const float K[9], const float D[5],
                              float image_height, float image_width, const float center[3]) {
  CKARG(ctx && P && K && D && center, "null argument"); SF_CHECK();
  CKARG(ctx->strip.halo_rows == 0 && ctx->strip.row_count == ctx->prm.cell_n, "camera path: single-strip contexts only");
  CK(hipSetDevice(ctx->device));
  FLUSH();
  const size_t L = (size_t)ctx->prm.cell_n * ctx->prm.cell_n;
  if (!ctx->img_uv) { CK(hipMalloc((void**)&ctx->img_uv, sizeof(float) * 2 * L)); CK(hipMalloc((void**)&ctx->img_valid, L)); }
  CamArgs A;
  memcpy(A.P, P, sizeof A.P); memcpy(A.K, K, sizeof A.K); memcpy(A.D, D, sizeof A.D); memcpy(A.center, center, sizeof A.center);
  A.x1 = x1; A.y1 = y1; A.z1 = z1; A.ih = image_height; A.iw = image_width; A.tol = ctx->img_tol_set ? ctx->img_tol : 0.10;
  launch_image_corr(ctx->stream, ctx->kp, A, ctx->cells, ctx->img_uv, ctx->img_valid);
  CK(hipGetLastError());
  return EMAP_OK;
}
int emap_image_get_correspondence(emap_ctx* ctx, float* uv_host, uint8_t* valid_host) {
  CKARG(ctx && uv_host && valid_host && ctx->img_uv, "no correspondence computed yet"); SF_CHECK();
  CK(hipSetDevice(ctx->device));
  const size_t L = (size_t)ctx->prm.cell_n * ctx->prm.cell_n;
  CK(hipMemcpyAsync(uv_host, ctx->img_uv, sizeof(float) * 2 * L, hipMemcpyDeviceToHost, ctx->stream));
  CK(hipMemcpyAsync(valid_host, ctx->img_valid, L, hipMemcpyDeviceToHost, ctx->stream));
  CK(hipStreamSynchronize(ctx->stream));
  return EMAP_OK;
}
int emap_image_fuse(emap_ctx* ctx, int32_t kind, int32_t layer, const float* host_image, int32_t n_planes, int32_t height, int32_t width,
                    double alpha) {
  CKARG(ctx && host_image && kind >= 0 && kind <= 2 && layer >= 0 && layer < ctx->sem_layers, "bad argument"); SF_CHECK();
  CKARG(ctx->img_uv, "emap_image_correspondence must run first");
  CKARG(n_planes >= (kind == 1 ? 3 : 1) && height > 0 && width > 0, "bad image shape");
  CK(hipSetDevice(ctx->device));
  const size_t need = (size_t)n_planes * height * width;
  if (need > ctx->img_cap) {
    CK(hipStreamSynchronize(ctx->stream));
    if (ctx->img_buf) CK(hipFree(ctx->img_buf));
    ctx->img_buf = nullptr; ctx->img_cap = 0;
    CK(hipMalloc((void**)&ctx->img_buf, sizeof(float) * need));
    ctx->img_cap = need;
  }
  CK(hipMemcpyAsync(ctx->img_buf, host_image, sizeof(float) * need, hipMemcpyHostToDevice, ctx->stream));
  launch_image_fuse(ctx->stream, ctx->kp, kind, ctx->sem + (size_t)layer * ctx->ncells_alloc, ctx->img_buf, ctx->img_uv, ctx->img_valid,
                    (float)height, (float)width, alpha);
  CK(hipGetLastError());
  CK(hipStreamSynchronize(ctx->stream));   // host image is only borrowed for the call
  return EMAP_OK;
}

int emap_image_set_tolerance(emap_ctx* ctx, double tolerance_z_collision) {
  CKARG(ctx && tolerance_z_collision == tolerance_z_collision, "bad argument");
  ctx->img_tol = tolerance_z_collision; ctx->img_tol_set = true;
  return EMAP_OK;
}
// the three *_correspondences_to_map kernels on caller arrays (the factories of the compat package): one (cell_n, cell_n) plane in,
// one out; cells without a valid correspondence keep their value (the reference's else branch copies sem_map to new_sem_map)
int emap_image_fuse_arrays(emap_ctx* ctx, int32_t kind, const float* sem_plane, const float* host_image, int32_t n_planes, int32_t height,
                           int32_t width, const float* uv, const uint8_t* valid, double alpha, float* out_plane) {
  CKARG(ctx && sem_plane && host_image && uv && valid && out_plane && kind >= 0 && kind <= 2, "bad argument"); SF_CHECK();
  CKARG(n_planes >= (kind == 1 ? 3 : 1) && height > 0 && width > 0, "bad image shape");
  CK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t L = (size_t)ctx->prm.cell_n * ctx->prm.cell_n;
  DevBuf S, I, U, V;
  CK(S.put(sem_plane, sizeof(float) * L, st, out_plane));
  CK(I.put(host_image, sizeof(float) * (size_t)n_planes * height * width, st, nullptr));
  CK(U.put(uv, sizeof(float) * 2 * L, st, nullptr));
  CK(V.put(valid, L, st, nullptr));
  KP kp = ctx->kp; kp.org_r = kp.org_c = 0;               // caller arrays are logical: no circular origin
  launch_image_fuse(st, kp, kind, (float*)S.d, (const float*)I.d, (const float*)U.d, (const unsigned char*)V.d, (float)height, (float)width, alpha);
  CK(hipGetLastError());
  CK(S.get(st));
  CK(hipStreamSynchronize(st));
  return EMAP_OK;
}

// ---- safety polygon (reference elevation_mapping.py:837-889, polygon_mask_kernel custom_kernels.py:509-651) -------------
// get_idx of the polygon kernel (:587-603): float16 helper parameters, FLOAT resolution / width constants (unlike the map
// kernels), index clamped through float16.
static int polygon_cell(const emap_params& p, float x, float y, float cx, float cy, int* ix, int* iy) {
  const bool h = p.mode == EMAP_MODE_REFERENCE_FP16;
  auto Q = [&](float v) { return h ? q16(v) : v; };
  auto axis = [&](float v, float c) {
    const float q = (Q(v) - Q(c)) / (float)p.resolution;
    const double val = (double)q + 0.5 * (double)(float)p.cell_n;
    int i = (val != val) ? 0 : (int)fmin(fmax(val, -2147483648.0), 2147483647.0);
    float fi = Q((float)i);
    fi = fmaxf(fminf(fi, Q((float)(p.cell_n - 1))), Q(0.0f));
    return (int)fi;
  };
  const int idx = p.cell_n * axis(x, cx) + axis(y, cy);
  *ix = idx / p.cell_n; *iy = idx % p.cell_n;
  return idx;
}
int emap_polygon_mask(emap_ctx* ctx, const float* polygon_xy, int32_t n_vertices, float center_x, float center_y, float* host_mask) {
  CKARG(ctx && polygon_xy && host_mask && n_vertices >= 1 && n_vertices <= 4096, "bad polygon");
  CKARG(ctx->strip.halo_rows == 0 && ctx->strip.row_count == ctx->prm.cell_n, "emap_polygon_mask: single-strip contexts only");
  CK(hipSetDevice(ctx->device));
  const int C = ctx->prm.cell_n;
  std::vector<int> v(2 * (size_t)n_vertices);
  float mn[2] = {polygon_xy[0], polygon_xy[1]}, mx[2] = {polygon_xy[0], polygon_xy[1]};
  for (int j = 0; j < n_vertices; ++j) {
    polygon_cell(ctx->prm, polygon_xy[2 * j], polygon_xy[2 * j + 1], center_x, center_y, &v[j], &v[n_vertices + j]);
    for (int a = 0; a < 2; ++a) { mn[a] = fminf(mn[a], polygon_xy[2 * j + a]); mx[a] = fmaxf(mx[a], polygon_xy[2 * j + a]); }
  }
  int bbox[4];
  polygon_cell(ctx->prm, mn[0], mn[1], center_x, center_y, &bbox[0], &bbox[1]);
  polygon_cell(ctx->prm, mx[0], mx[1], center_x, center_y, &bbox[2], &bbox[3]);
  int* dv = nullptr;
  CK(hipMalloc((void**)&dv, sizeof(int) * v.size()));
  hipError_t e = hipMemcpyAsync(dv, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    launch_polygon_mask(ctx->stream, C, dv, dv + n_vertices, n_vertices, bbox, ctx->scratch);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(host_mask, ctx->scratch, sizeof(float) * (size_t)C * C, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(dv);
  if (e != hipSuccess) { ctx->err = std::string("emap_polygon_mask: ") + hipGetErrorString(e); return EMAP_ERR_HIP; }
  return EMAP_OK;
}

// ---- dilation of caller planes: ElevationMap.initialize_map (reference elevation_mapping.py:899-923) ---------------------
int emap_dilate_planes(emap_ctx* ctx, const float* host_plane, const float* host_mask, int32_t dilation_size, int32_t iterations,
                       float* host_out, float* host_out_mask) {
  CKARG(ctx && host_plane && host_mask && host_out && host_out_mask, "null argument");
  CKARG(dilation_size >= 0 && dilation_size <= 64 && iterations >= 1 && iterations <= 64, "bad dilation size / iteration count");
  CK(hipSetDevice(ctx->device));
  const int C = ctx->prm.cell_n; const size_t L = (size_t)C * C, bytes = L * sizeof(float);
  float* buf = nullptr;
  CK(hipMalloc((void**)&buf, bytes * 4));
  float *p0 = buf, *m0 = buf + L, *p1 = buf + 2 * L, *m1 = buf + 3 * L;
  hipError_t e = hipMemcpyAsync(p0, host_plane, bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(m0, host_mask, bytes, hipMemcpyHostToDevice, ctx->stream);
  for (int it = 0; it < iterations && e == hipSuccess; ++it) {
    launch_dilate_planes(ctx->stream, C, dilation_size, p0, m0, p1, m1);
    e = hipGetLastError();
    float* t = p0; p0 = p1; p1 = t; t = m0; m0 = m1; m1 = t;
  }
  if (e == hipSuccess) e = hipMemcpyAsync(host_out, p0, bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(host_out_mask, m0, bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(buf);
  if (e != hipSuccess) { ctx->err = std::string("emap_dilate_planes: ") + hipGetErrorString(e); return EMAP_ERR_HIP; }
  return EMAP_OK;
}

// ---- halos ----------------------------------------------------------------------------------------------------
int emap_halo_bytes(emap_ctx* ctx, int64_t* bytes_per_side) {
  CKARG(ctx && bytes_per_side, "null argument");
  *bytes_per_side = (int64_t)ctx->strip.halo_rows * ctx->prm.cell_n * (int64_t)sizeof(float4);      // the cold half-cell plane only (see halo_exchange_start)
  return EMAP_OK;
}
int emap_halo_pack(emap_ctx* ctx, int side, float* dev_buf) {
  CKARG(ctx && dev_buf && (side == 0 || side == 1), "bad argument"); SF_CHECK();
  CK(hipSetDevice(ctx->device));
  const long H = ctx->strip.halo_rows, C = ctx->prm.cell_n, n = ctx->strip.row_count;
  if (H == 0) return EMAP_OK;
  CKARG(n >= H, "strip thinner than its halo");
  const long off = side == 0 ? H * C : (H + n - H) * C;                   // first / last H owned rows of the COLD half-cell plane
  CK(hipMemcpyAsync(dev_buf, ctx->cells.cold + off, sizeof(float4) * H * C, hipMemcpyDeviceToDevice, ctx->stream));
  return EMAP_OK;
}
int emap_halo_unpack(emap_ctx* ctx, int side, const float* dev_buf) {
  CKARG(ctx && dev_buf && (side == 0 || side == 1), "bad argument"); SF_CHECK();
  CK(hipSetDevice(ctx->device));
  const long H = ctx->strip.halo_rows, C = ctx->prm.cell_n, n = ctx->strip.row_count;
  if (H == 0) return EMAP_OK;
  const long off = side == 0 ? 0 : (H + n) * C;
  CK(hipMemcpyAsync(ctx->cells.cold + off, dev_buf, sizeof(float4) * H * C, hipMemcpyDeviceToDevice, ctx->stream));
  return EMAP_OK;
}

// boundary rows of the three normal planes (3 x halo_rows x cell_n floats per side) for the exchange after a row shift
// (see normal_exchange); emap_normal_row_lag tells the caller whether it is due
int emap_normal_row_lag(emap_ctx* ctx, int32_t* lag) { CKARG(ctx && lag, "null argument"); *lag = normal_row_lag(ctx); return EMAP_OK; }
int emap_normal_halo_pack(emap_ctx* ctx, int side, float* dev_buf) {
  CKARG(ctx && dev_buf && (side == 0 || side == 1), "bad argument"); SF_CHECK();
  CK(hipSetDevice(ctx->device));
  const long H = ctx->strip.halo_rows, C = ctx->prm.cell_n, n = ctx->strip.row_count;
  if (H == 0) return EMAP_OK;
  for (int k = 0; k < 3; ++k)
    CK(hipMemcpyAsync(dev_buf + (size_t)k * H * C, ctx->normal + (size_t)k * ctx->ncells_alloc + (side == 0 ? H * C : n * C), sizeof(float) * H * C,
                      hipMemcpyDeviceToDevice, ctx->stream));
  return EMAP_OK;
}
int emap_normal_halo_unpack(emap_ctx* ctx, int side, const float* dev_buf) {
  CKARG(ctx && dev_buf && (side == 0 || side == 1), "bad argument"); SF_CHECK();
  CK(hipSetDevice(ctx->device));
  const long H = ctx->strip.halo_rows, C = ctx->prm.cell_n, n = ctx->strip.row_count;
  if (H == 0) return EMAP_OK;
  for (int k = 0; k < 3; ++k)
    CK(hipMemcpyAsync(ctx->normal + (size_t)k * ctx->ncells_alloc + (side == 0 ? 0 : (H + n) * C), dev_buf + (size_t)k * H * C, sizeof(float) * H * C,
                      hipMemcpyDeviceToDevice, ctx->stream));
  return EMAP_OK;
}

// ---- multi-GPU: row strips, one process per GPU, RCCL over xGMI ---------------------------------------------------
// The two exchange steps of the path: an all-reduce of the drift sums (2 x f64) between the count and fuse
// stages, and the neighbour exchange of halo rows before the stencils.  Halo rows are contiguous in the cold half-cell plane (the
// only one the stencils read), so RCCL sends the first / last owned rows and receives into the halo rows IN PLACE (no pack / unpack copies); the
// exchange runs on its own stream while the stencil tiles that do not depend on halo rows run on the main stream.
static RcclApi* rccl_open(const char* path, std::string* why) {
  void* h = dlopen(path && *path ? path : "librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) { *why = std::string("dlopen: ") + dlerror(); return nullptr; }
  RcclApi* a = new RcclApi(); a->handle = h;
  bool ok = true;
#define SYM(field, name) do { a->field = (decltype(a->field))dlsym(h, name); if (!a->field) { ok = false; *why = std::string("missing symbol ") + name; } } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllReduce, "ncclAllReduce"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString"); SYM(CommCount, "ncclCommCount");
#undef SYM
  if (!ok) { delete a; return nullptr; }   // the handle stays open: unloading a GPU runtime library is not safe
  return a;
}
#define CKN(call)                                                                                                       \
  do { ncclResult_t r_ = (call);                                                                                        \
       if (r_ != ncclSuccess) { ctx->err = std::string(#call) + ": " + ctx->rccl->GetErrorString(r_); return EMAP_ERR_COMM; } } while (0)

int emap_comm_unique_id(const char* rccl_path, uint8_t id_out[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (!id_out) return EMAP_ERR_INVALID;
  std::string why;
  RcclApi* a = rccl_open(rccl_path, &why);
  if (!a) { fprintf(stderr, "emap_comm_unique_id: %s\n", why.c_str()); return EMAP_ERR_COMM; }
  ncclUniqueId id;
  ncclResult_t r = a->GetUniqueId(&id);
  if (r != ncclSuccess) { fprintf(stderr, "emap_comm_unique_id: %s\n", a->GetErrorString(r)); delete a; return EMAP_ERR_COMM; }
  memcpy(id_out, &id, 128);
  delete a;
  return EMAP_OK;
}

static long window_cap(const emap_ctx* ctx);
static int alloc_window(emap_ctx* ctx, long cap);
int emap_comm_init(emap_ctx* ctx, const char* rccl_path, const uint8_t id[128], int32_t rank, int32_t world) {
  CKARG(ctx && id && world >= 1 && rank >= 0 && rank < world, "bad rank / world");
  CKARG(world <= 16, "at most 16 ranks");                    // (before anything collective: every rank sees the same `world`)
  CKARG(!ctx->rccl, "communicator already initialised");
  CKARG(world == 1 || ctx->strip.halo_rows > 0, "a strip of a multi-rank map needs halo rows");
  CK(hipSetDevice(ctx->device));
  std::string why;
  RcclApi* a = rccl_open(rccl_path, &why);
  if (!a) { ctx->err = why; return EMAP_ERR_COMM; }
  ctx->rccl = a;
  ncclUniqueId uid; memcpy(&uid, id, 128);
  ncclResult_t r = a->CommInitRank(&ctx->comm, world, uid, rank);
  if (r != ncclSuccess) { ctx->err = std::string("ncclCommInitRank: ") + a->GetErrorString(r); delete a; ctx->rccl = nullptr; ctx->comm = nullptr; return EMAP_ERR_COMM; }
  ctx->comm_rank = rank; ctx->comm_world = world;
  // From here on the call is COLLECTIVE: a rank that fails locally must not leave the others waiting in an all-reduce.  The local set-up
  // therefore only collects a status; the first all-reduce carries it (max over the ranks) together with the ABI check, every rank
  // returns the same verdict, and a failed init leaves no half-built communicator behind.
  int local_rc = EMAP_OK;
  auto step = [&](hipError_t e, const char* what) { if (local_rc == EMAP_OK && e != hipSuccess) { local_rc = EMAP_ERR_HIP; ctx->err = std::string(what) + ": " + hipGetErrorString(e); } };
  step(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking), "hipStreamCreate(comm)");
  step(hipEventCreateWithFlags(&ctx->ev_ready, hipEventDisableTiming), "hipEventCreate");
  step(hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming), "hipEventCreate");
  step(hipMalloc((void**)&ctx->comm_sums, sizeof(double) * 36), "hipMalloc(comm_sums)");
  if (local_rc == EMAP_OK) step(hipMemsetAsync(ctx->comm_sums, 0, sizeof(double) * 36, ctx->stream), "hipMemsetAsync(comm_sums)");
  if (local_rc == EMAP_OK && world > 1 && ctx->prm.enable_visibility_cleanup && ctx->win_cap < window_cap(ctx))      // rays by ray: see ensure_window
    local_rc = alloc_window(ctx, window_cap(ctx));
  auto fail = [&](int rc) { const std::string keep = ctx->err; emap_comm_destroy(ctx); ctx->err = keep; return rc; };
  if (world > 1 && !ctx->comm_sums) return fail(local_rc ? local_rc : EMAP_ERR_HIP);      // (nothing to reduce through: the peers time out in RCCL -- an out-of-memory device at start-up)
  // every rank's owned physical rows (the strips need not be equally high): who holds which normal rows after a row shift
  ctx->cut_begin.assign(world, 0); ctx->cut_count.assign(world, 0);
  ctx->cut_begin[rank] = ctx->strip.row_begin; ctx->cut_count[rank] = ctx->strip.row_count;
  if (world > 1) {
    // Every rank must speak the same ABI: the halo rows are raw 16-byte cold half cells and the ray window raw 32-byte records, so a
    // peer built against another layout would exchange misaligned bytes silently.  max(v) and max(-v) over the ranks: equal and
    // opposite iff all ranks agree; the third word is the worst local status.
    const double mine[3] = {(double)EMAP_ABI_VERSION, -(double)EMAP_ABI_VERSION, (double)(local_rc != EMAP_OK)};
    double got[3] = {0.0, 0.0, 0.0};
    CK(hipMemcpyAsync(ctx->comm_sums + 16, mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream));
    CKN(a->AllReduce(ctx->comm_sums + 16, ctx->comm_sums + 20, 3, ncclFloat64, ncclMax, ctx->comm, ctx->stream));
    CK(hipMemcpyAsync(got, ctx->comm_sums + 20, sizeof got, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    if (got[2] != 0.0) {
      if (local_rc == EMAP_OK) ctx->err = "emap_comm_init: another rank could not set up its communicator state";
      return fail(local_rc ? local_rc : EMAP_ERR_COMM);
    }
    if (got[0] != mine[0] || got[1] != mine[1]) {
      ctx->err = "emap_comm_init: the ranks were built against different EMAP_ABI_VERSIONs (" + std::to_string((int)-got[1]) + " .. " + std::to_string((int)got[0]) + ", this rank: " + std::to_string(EMAP_ABI_VERSION) + ")";
      return fail(EMAP_ERR_COMM);
    }
    double cuts[32], all[32];
    for (int k = 0; k < 32; ++k) cuts[k] = 0.0;
    cuts[2 * rank] = ctx->strip.row_begin; cuts[2 * rank + 1] = ctx->strip.row_count;
    CK(hipMemcpyAsync(ctx->comm_sums + 4, cuts, sizeof(double) * 2 * 16, hipMemcpyHostToDevice, ctx->stream));      // ([4..36): the host all-reduce's slots)
    CKN(a->AllReduce(ctx->comm_sums + 4, ctx->comm_sums + 4, 2 * (size_t)world, ncclFloat64, ncclSum, ctx->comm, ctx->stream));
    CK(hipMemcpyAsync(all, ctx->comm_sums + 4, sizeof(double) * 2 * world, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    long total = 0;
    for (int q = 0; q < world; ++q) { ctx->cut_begin[q] = (int)all[2 * q]; ctx->cut_count[q] = (int)all[2 * q + 1]; total += ctx->cut_count[q]; }
    ctx->cuts_ok = total == ctx->prm.cell_n;               // (checked where the boundaries are needed: normal_exchange)
  } else if (local_rc != EMAP_OK) return fail(local_rc);
  return EMAP_OK;
}

int emap_comm_destroy(emap_ctx* ctx) {
  if (!ctx || !ctx->rccl) return EMAP_OK;
  hipSetDevice(ctx->device);
  if (ctx->comm_stream) hipStreamSynchronize(ctx->comm_stream);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  if (ctx->comm) ctx->rccl->CommDestroy(ctx->comm);
  if (ctx->comm_stream) hipStreamDestroy(ctx->comm_stream);
  if (ctx->ev_ready) hipEventDestroy(ctx->ev_ready);
  if (ctx->ev_done) hipEventDestroy(ctx->ev_done);
  hipFree(ctx->comm_sums); hipFree(ctx->gather_buf); ctx->gather_buf = nullptr;
  hipFree(ctx->nlag_buf); ctx->nlag_buf = nullptr; ctx->nlag_cap = 0; ctx->nlag_ready = false;
  delete ctx->rccl;
  ctx->rccl = nullptr; ctx->comm = nullptr; ctx->comm_stream = nullptr; ctx->ev_ready = ctx->ev_done = nullptr; ctx->comm_sums = nullptr;
  return EMAP_OK;
}

// halo exchange with the strip neighbours, in place, on the communication stream; main-stream work issued after this call and
// before the wait on ev_done overlaps it.  The strips are PHYSICAL row ranges of a circular map, so the neighbours form a ring:
// rank r sends its first H owned rows to r-1 and its last H owned rows to r+1 (mod world).  The logical seam of the map lies
// wherever the circular origin put it; rows across it are received like any others and masked by the stencils (is_inside).
// Posting order (sends: low, high; receives: upper halo, lower halo) keeps the pairs apart when both neighbours are the same rank.
static int ring_exchange(emap_ctx* ctx, char* base, size_t row_bytes, hipStream_t st) {
  const long H = ctx->strip.halo_rows, n = ctx->strip.row_count;
  const size_t bytes = row_bytes * (size_t)H;
  const RcclApi* a = ctx->rccl;
  const int W = ctx->comm_world, prev = (ctx->comm_rank + W - 1) % W, next = (ctx->comm_rank + 1) % W;
  CKN(a->GroupStart());
  CKN(a->Send(base + row_bytes * H, bytes, ncclChar, prev, ctx->comm, st));                  // first H owned rows
  CKN(a->Send(base + row_bytes * n, bytes, ncclChar, next, ctx->comm, st));                  // rows [n-H, n) of the strip
  CKN(a->Recv(base + row_bytes * (H + n), bytes, ncclChar, next, ctx->comm, st));            // upper halo
  CKN(a->Recv(base, bytes, ncclChar, prev, ctx->comm, st));                                  // lower halo
  CKN(a->GroupEnd());
  return EMAP_OK;
}
static int halo_exchange_start(emap_ctx* ctx) {
  CK(hipEventRecord(ctx->ev_ready, ctx->stream));
  CK(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_ready, 0));
  // Only the COLD half-cell plane travels: the halo rows have exactly one reader, the stencil kernel's staging loop, and it reads
  // (upper_bound, is_upper_bound, valid') from cells.cold alone (emap_kernels.hip: k_post / k_post_dma) -- the point passes, the ray
  // pass and every per-cell pass address owned rows only.  (Until round 3 both planes were sent: twice the xGMI bytes for nothing.)
  int rc = ring_exchange(ctx, reinterpret_cast<char*>(ctx->cells.cold), sizeof(float4) * (size_t)ctx->prm.cell_n, ctx->comm_stream);
  if (rc) return rc;
  CK(hipEventRecord(ctx->ev_done, ctx->comm_stream));
  return EMAP_OK;
}
// After a row shift the un-shifted normal planes (the reference does not roll normal_map, elevation_mapping.py:200-214) sit `lag`
// rows away from the cells they belong to: the normal of the cell in PHYSICAL row p lives in physical row (p + lag) mod C of the
// planes -- in this strip, with a neighbour, or (a robot that moved further than a strip is high) with a rank further away.  Before
// the visibility pass every rank therefore fetches the rows [(row_begin + lag) mod C, + row_count) into a ROW-ALIGNED copy (row j =
// the normals of owned row j; columns stay at the planes' own origin: normal_index in emap_device.h) from whoever owns them: local
// pieces by device copies, the others by ONE grouped send / receive.  Every rank derives the same list of pieces from the strips'
// boundaries (gathered by emap_comm_init) and the lag (a property of the shared origin), walks it in the same order, and so posts its
// sends and receives in matching order.  Any lag, any strip heights: a strip never reads zeros where the single context reads stale
// normals (until round 4 only lags up to halo_rows were served, from the planes' halo rows).
struct LagPiece { int q, r, src, dst, rows; };      // rank q needs physical rows [src, src + rows), owned by rank r, as rows [dst, dst + rows) of its copy
static void lag_pieces(int C, int W, const int* cut_begin, const int* cut_count, int lag, std::vector<LagPiece>& out) {
  lag = ((lag % C) + C) % C;
  for (int q = 0; q < W; ++q) {                               // rank q needs the physical rows [(b_q + lag) mod C, + n_q): <= 2 linear pieces
    const int b = (cut_begin[q] + lag) % C, nq = cut_count[q];
    for (int piece = 0; piece < 2; ++piece) {
      const int p0 = piece == 0 ? b : 0, p1 = piece == 0 ? std::min(C, b + nq) : b + nq - C;      // physical rows [p0, p1)
      const int d0 = piece == 0 ? 0 : C - b;                                                       // row of q's copy where the piece starts
      if (p1 <= p0) continue;
      for (int r = 0; r < W; ++r) {                           // ... cut by the owners of those rows
        const int o0 = std::max(p0, cut_begin[r]), o1 = std::min(p1, cut_begin[r] + cut_count[r]);
        if (o1 > o0) out.push_back(LagPiece{q, r, o0, d0 + o0 - p0, o1 - o0});
      }
    }
  }
}
// the plan as data (CPU property tests: every row of every rank's copy is written exactly once, from the row it belongs to)
int emap_normal_lag_plan(int32_t cell_n, int32_t world, const int32_t* cut_begin, const int32_t* cut_count, int32_t lag, int32_t* pieces5, int32_t max_pieces, int32_t* n_pieces) {
  if (!cut_begin || !cut_count || !pieces5 || !n_pieces || cell_n < 1 || world < 1 || max_pieces < 0) return EMAP_ERR_INVALID;
  std::vector<LagPiece> v;
  lag_pieces(cell_n, world, cut_begin, cut_count, lag, v);
  *n_pieces = (int32_t)v.size();
  if ((long)v.size() > max_pieces) return EMAP_ERR_INVALID;
  for (size_t k = 0; k < v.size(); ++k) { pieces5[5 * k] = v[k].q; pieces5[5 * k + 1] = v[k].r; pieces5[5 * k + 2] = v[k].src; pieces5[5 * k + 3] = v[k].dst; pieces5[5 * k + 4] = v[k].rows; }
  return EMAP_OK;
}
static int normal_exchange(emap_ctx* ctx) {
  const int C = ctx->prm.cell_n, W = ctx->comm_world, me = ctx->comm_rank;
  const long n = ctx->strip.row_count, H = ctx->strip.halo_rows;
  if (!ctx->cuts_ok) { ctx->err = "normal_exchange: the strips the ranks reported to emap_comm_init do not tile the map"; return EMAP_ERR_COMM; }
  if ((long)n * C > ctx->nlag_cap) {
    CK(hipStreamSynchronize(ctx->stream));
    hipFree(ctx->nlag_buf); ctx->nlag_buf = nullptr; ctx->nlag_cap = 0;
    CK(hipMalloc((void**)&ctx->nlag_buf, sizeof(float) * 3 * (size_t)n * C));
    ctx->nlag_cap = n * C;
  }
  std::vector<LagPiece> plan;
  lag_pieces(C, W, ctx->cut_begin.data(), ctx->cut_count.data(), normal_row_lag(ctx), plan);
  const RcclApi* a = ctx->rccl;
  const size_t rowb = sizeof(float) * (size_t)C;
  bool grouped = false;
  for (const LagPiece& pc : plan) {
    if (pc.q != me && pc.r != me) continue;
    const size_t bytes = rowb * (size_t)pc.rows;
    for (int k = 0; k < 3; ++k) {
      float* dst = ctx->nlag_buf + (size_t)k * n * C + (size_t)pc.dst * C;                                                  // (q == me)
      const float* src = ctx->normal + (size_t)k * ctx->ncells_alloc + (size_t)(H + pc.src - ctx->strip.row_begin) * C;     // (r == me)
      if (pc.q == me && pc.r == me) CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
      else {
        if (!grouped) { CKN(a->GroupStart()); grouped = true; }
        if (pc.q == me) CKN(a->Recv(dst, bytes, ncclChar, pc.r, ctx->comm, ctx->stream));
        else CKN(a->Send(src, bytes, ncclChar, pc.q, ctx->comm, ctx->stream));
      }
    }
  }
  if (grouped) CKN(a->GroupEnd());
  ctx->nlag_ready = true;
  return EMAP_OK;
}

// ---- rays by ray (emap_kernels.hip: k_win_pack / k_win_prepare / k_win_unpack) ---------------------------------------------------
// Decided from values every rank shares (parameters, world size, cloud size): all ranks take the same branch of the collective code.
static bool frame_binned_everywhere(const emap_ctx* ctx) {      // emap_count's choice, for a cloud of this size
  return ctx->n_pts_all > 0 && (ctx->scatter_mode == 2 || (ctx->scatter_mode == 0 && bins_possible(ctx) && ctx->n_pts_all >= 131072));
}
static bool rays_by_ray(const emap_ctx* ctx) {
  if (!ctx->rccl || ctx->comm_world <= 1 || !ctx->prm.enable_visibility_cleanup || ctx->ray_mode == 1) return false;
  if (!frame_binned_everywhere(ctx)) return false;                // the window march walks the frame's tile-sorted records
  // by row below 2048^2 cells: the window exchange (three all-reduces) costs about what the whole pass costs there
  return ctx->ray_mode == 2 || ctx->prm.cell_n >= 2048;
}
// the window of this frame: everything within reach of a ray of at most max_ray_length that starts at the sensor (t is map-centre
// relative); false: no cell of the map is within reach
static bool ray_window(const emap_ctx* ctx, const float t[3], Win* w) {
  const emap_params& p = ctx->prm;
  const int C = p.cell_n;
  const double reach = p.max_ray_length * 1.01 + 4.0 * p.resolution;     // (the float16 ray direction is a unit vector to 2^-10; the sample positions are rounded to half)
  auto lo = [&](double x) { double v = std::floor((x - reach) / p.resolution + 0.5 * C) - 2.0; return v < 0 ? 0 : (v > C ? C : (int)v); };
  auto hi = [&](double x) { double v = std::ceil((x + reach) / p.resolution + 0.5 * C) + 3.0; return v < 0 ? 0 : (v > C ? C : (int)v); };
  const int r_lo = lo(t[0]), r_hi = hi(t[0]), c_lo = lo(t[1]), c_hi = hi(t[1]);
  if (r_hi <= r_lo || c_hi <= c_lo) return false;
  w->r0 = r_lo & ~7; w->nr = ((r_hi - w->r0) + 7) & ~7;
  w->c0 = c_lo & ~63; w->nc = ((c_hi - w->c0) + 63) & ~63;
  return true;
}
// The largest window ray_window can return for the context's parameters (reach on both sides + its alignment slack, clipped to the map).
static long window_cap(const emap_ctx* ctx) {
  const emap_params& p = ctx->prm;
  const long C = p.cell_n;
  const double reach = p.max_ray_length * 1.01 + 4.0 * p.resolution;
  const long W = 2 * (long)std::ceil(reach / p.resolution) + 8;
  const long nr = std::min<long>(((W + 7) & ~7L) + 8, (C + 7) & ~7L), nc = std::min<long>(((W + 63) & ~63L) + 64, (C + 63) & ~63L);
  return nr * nc;
}
static int alloc_window(emap_ctx* ctx, long cap) {
  CK(hipStreamSynchronize(ctx->stream));
  hipFree(ctx->win_state); hipFree(ctx->win_rec); hipFree(ctx->win_bits); hipFree(ctx->win_thr); hipFree(ctx->win_dh); hipFree(ctx->win_key); hipFree(ctx->win_red_dh); hipFree(ctx->win_red_key);
  ctx->win_state = nullptr; ctx->win_rec = nullptr; ctx->win_bits = nullptr; ctx->win_thr = nullptr; ctx->win_dh = nullptr; ctx->win_key = nullptr; ctx->win_red_dh = nullptr; ctx->win_red_key = nullptr; ctx->win_cap = 0;
  CK(hipMalloc((void**)&ctx->win_state, sizeof(unsigned int) * 12 * (size_t)cap));      // hot 4 + cold 4 + normals 3 + wall flag 1 words per cell (local expansion)
  CK(hipMalloc((void**)&ctx->win_rec, sizeof(unsigned int) * 8 * (size_t)cap));         // the 32-byte records that travel
  CK(hipMalloc((void**)&ctx->win_bits, sizeof(unsigned long long) * ((size_t)cap / 64 + 2)));
  CK(hipMalloc((void**)&ctx->win_thr, sizeof(float) * ((size_t)cap / 64 + 1)));
  CK(hipMalloc((void**)&ctx->win_dh, sizeof(long long) * 2 * (size_t)cap));
  CK(hipMalloc((void**)&ctx->win_key, sizeof(unsigned int) * (size_t)cap));
  if (ctx->comm_world > 1) {       // what the other ranks send an owner: their {dec, hits} pairs and keys for its rows (at most the whole window, from every other rank)
    CK(hipMalloc((void**)&ctx->win_red_dh, sizeof(long long) * 2 * (size_t)cap * (size_t)(ctx->comm_world - 1)));
    CK(hipMalloc((void**)&ctx->win_red_key, sizeof(unsigned int) * (size_t)cap * (size_t)(ctx->comm_world - 1)));
  }
  ctx->win_cap = cap;
  return EMAP_OK;
}
// The window buffers are allocated COLLECTIVELY, in emap_comm_init (every rank, for the largest window its parameters allow; the
// caller's agreement step follows): an allocation that fails on one rank in the middle of a frame would leave the other ranks alone
// in the three all-reduces below.  The growth path here only runs when the parameters changed after emap_comm_init.
static int ensure_window(emap_ctx* ctx, Win* w) {
  const long n = (long)w->nr * w->nc;
  if (n > ctx->win_cap) { int rc = alloc_window(ctx, std::max(n + n / 8, window_cap(ctx))); if (rc) return rc; }
  w->hot = reinterpret_cast<float4*>(ctx->win_state); w->cold = w->hot + n;
  w->normal = reinterpret_cast<float*>(w->cold + n); w->inl = ctx->win_state + 11 * n;
  w->bits = ctx->win_bits; w->thr = ctx->win_thr; w->dh = ctx->win_dh; w->key = ctx->win_key; w->rec = ctx->win_rec;
  return EMAP_OK;
}
// The visibility pass of a sharded frame, by ray: called where emap_rays would be, between the tile kernel and k_ray_apply.
static int rays_by_ray_pass(emap_ctx* ctx, const float R[9], const float t[3]) {
  Win w; memset(&w, 0, sizeof w);
  if (!ray_window(ctx, t, &w)) return EMAP_OK;                    // the same decision on every rank: no collective is skipped one-sidedly
  int rc = ensure_window(ctx, &w); if (rc) return rc;
  const RcclApi* a = ctx->rccl;
  const long n = (long)w.nr * w.nc;
  hipStream_t st = ctx->stream;
  // (1) the window's cells, normals and wall flags as 32-byte records: owners fill their rows, an exact integer all-reduce replicates them
  CK(hipMemsetAsync(ctx->win_rec, 0, sizeof(unsigned int) * 8 * (size_t)n, st));
  KP kpk = ctx->kp;
  kpk.nlag = ctx->nlag_ready ? 1 : 0;
  launch_win_pack(st, kpk, w, ctx->cells, ctx->nlag_ready ? ctx->nlag_buf : ctx->normal, ctx->nlag_ready ? (long)ctx->strip.row_count * ctx->prm.cell_n : ctx->ncells_alloc,
                  ctx->inl_plane, ctx->inert, ctx->rt.f_wall);
  CK(hipGetLastError());
  // Who owns which rows of the window: runs of consecutive window rows per owner, from the strips' physical rows (gathered by
  // emap_comm_init) and the map's row origin -- the same list on every rank.  With it the window is replicated by BROADCASTS from its
  // two or three owners and the effects return by REDUCTIONS to them (round 6): grouped sends / receives, every rank receives each window
  // byte once and only the owners receive effects -- a ring all-reduce moves 2 (W - 1) / W of the whole window through every rank, twice.
  // EMAP_BYRAY_ALLREDUCE=1 (or strips that do not tile the map): the three all-reduces of round 4 / 5.
  struct Run { int q, a, rows; };
  std::vector<Run> runs;
  static const bool force_allreduce = getenv("EMAP_BYRAY_ALLREDUCE") && atoi(getenv("EMAP_BYRAY_ALLREDUCE")) != 0;
  bool owners = ctx->comm_world > 1 && ctx->cuts_ok && !force_allreduce && ctx->win_red_dh;
  if (owners) {
    const int C = ctx->prm.cell_n;
    for (int wr = 0; wr < w.nr && w.r0 + wr < C && owners; ++wr) {
      const int prow = (w.r0 + wr + ctx->kp.org_r) % C;
      int q = -1;
      for (int r = 0; r < ctx->comm_world; ++r) if (prow >= ctx->cut_begin[r] && prow < ctx->cut_begin[r] + ctx->cut_count[r]) { q = r; break; }
      if (q < 0) { owners = false; break; }
      if (!runs.empty() && runs.back().q == q && runs.back().a + runs.back().rows == wr) runs.back().rows++;
      else runs.push_back(Run{q, wr, 1});
    }
  }
  const int me = ctx->comm_rank, W = ctx->comm_world;
  size_t moved = 0;                                          // bytes this rank sends + receives in the frame's three exchange steps
  if (owners) {
    CKN(a->GroupStart());
    for (const Run& r : runs) {
      char* slab = reinterpret_cast<char*>(ctx->win_rec) + (size_t)r.a * w.nc * 32;
      const size_t bytes = (size_t)r.rows * w.nc * 32;
      if (r.q == me) { for (int p = 0; p < W; ++p) if (p != me) { CKN(a->Send(slab, bytes, ncclChar, p, ctx->comm, st)); moved += bytes; } }
      else { CKN(a->Recv(slab, bytes, ncclChar, r.q, ctx->comm, st)); moved += bytes; }
    }
    CKN(a->GroupEnd());
  } else {
    CKN(a->AllReduce(ctx->win_rec, ctx->win_rec, 8 * (size_t)n, ncclUint32, ncclSum, ctx->comm, st));
    moved = 32 * (size_t)n + 16 * (size_t)n + 4 * (size_t)n;      // per rank and frame: the window records + the effects ({dec, hits} sums, key maxima) that come back
  }
  // (2) bitmap + block thresholds of the window, accumulators re-armed
  launch_win_prepare(st, w, ctx->prm.cell_n);
  CK(hipMemsetAsync(w.bits + n / 64, 0xff, sizeof(unsigned long long), st));          // the all-ones word behind the last row (k_rays' passive lanes)
  CK(hipMemsetAsync(w.dh, 0, sizeof(long long) * 2 * (size_t)n, st));
  CK(hipMemsetAsync(w.key, 0, sizeof(unsigned int) * (size_t)n, st));
  // (3) the march: this rank's sorted records = the valid points of its rows, over the window
  KP kw = ctx->kp;
  kw.ray_pref = ctx->split_need ? (int)ctx->split_need[1] : 0;
  kw.org_r = kw.org_c = kw.norg_r = kw.norg_c = 0; kw.mv.n = 0;
  kw.row0 = w.r0; kw.nrows = w.nr; kw.halo = 0; kw.col0 = w.c0; kw.ncols = w.nc; kw.pitch = w.nc; kw.wmode = 1;
  Cells wc; wc.hot = w.hot; wc.cold = w.cold;
  char* const db = reinterpret_cast<char*>(w.dh);
  const AccRView av = {db, db + 8, reinterpret_cast<char*>(w.key), 16, 16, 4, 0};
  launch_rays(st, kw, make_pose(ctx, R, t), ctx->rt, ctx->pts, ctx->n_pts, ctx->stride, wc, av, w.normal, n, ctx->frame, ctx->want_ray_stats,
              w.bits, w.inl, 1, w.thr, reinterpret_cast<const unsigned int*>(ctx->bin_recs), ctx->bin_tile_start + ctx->bg.TB);
  CK(hipGetLastError());
  // (4) effects back to the owners: sums of {dec, hits}, maxima of the upper-bound keys
  if (owners) {
    // every rank sends its pairs and keys for a run to the run's owner; an owner receives W - 1 parts per run, packed run after run
    // (part j = the j-th other rank), and folds them into its own slab
    CKN(a->GroupStart());
    long off = 0;
    for (const Run& r : runs) {
      const size_t cells = (size_t)r.rows * w.nc;
      if (r.q == me) {
        for (int p = 0, j = 0; p < W; ++p) if (p != me) {
          CKN(a->Recv(reinterpret_cast<char*>(ctx->win_red_dh + 2 * ((size_t)j * ctx->win_cap + off)), cells * 16, ncclChar, p, ctx->comm, st));
          CKN(a->Recv(reinterpret_cast<char*>(ctx->win_red_key + ((size_t)j * ctx->win_cap + off)), cells * 4, ncclChar, p, ctx->comm, st));
          moved += cells * 20; ++j;
        }
        off += (long)cells;
      } else {
        CKN(a->Send(reinterpret_cast<char*>(w.dh + 2 * (size_t)r.a * w.nc), cells * 16, ncclChar, r.q, ctx->comm, st));
        CKN(a->Send(reinterpret_cast<char*>(w.key + (size_t)r.a * w.nc), cells * 4, ncclChar, r.q, ctx->comm, st));
        moved += cells * 20;
      }
    }
    CKN(a->GroupEnd());
    off = 0;
    for (const Run& r : runs) if (r.q == me) {
      const long cells = (long)r.rows * w.nc;
      launch_win_reduce(st, w.dh + 2 * (size_t)r.a * w.nc, w.key + (size_t)r.a * w.nc, ctx->win_red_dh, ctx->win_red_key, W - 1, ctx->win_cap, off, cells);
      off += cells;
    }
  } else {
    CKN(a->AllReduce(w.dh, w.dh, 2 * (size_t)n, ncclInt64, ncclSum, ctx->comm, st));
    CKN(a->AllReduce(w.key, w.key, (size_t)n, ncclUint32, ncclMax, ctx->comm, st));
  }
  ctx->wire_bytes = moved;
  launch_win_unpack(st, ctx->kp, w, ctx->accr);
  CK(hipGetLastError());
  return EMAP_OK;
}

// One frame of the strip: the stage order of ShardedElevationMap.update (sharded.py) with both exchange steps issued from
// here -- no Python, no host synchronisation between the stages.
static int update_sharded_impl(emap_ctx* ctx, const float R[9], const float t[3], double position_noise, double orientation_noise, emap_stats* stats);
int emap_update_sharded(emap_ctx* ctx, const float R[9], const float t[3], double position_noise, double orientation_noise, emap_stats* stats) {
  const int rc = update_sharded_impl(ctx, R, t, position_noise, orientation_noise, stats);
  if (ctx) { ctx->fsem_set = false; ctx->carry_want = false; ctx->in_update = false; }
  return rc;
}
static int update_sharded_impl(emap_ctx* ctx, const float R[9], const float t[3], double position_noise, double orientation_noise, emap_stats* stats) {
  CKARG(ctx && R && t, "null argument"); SF_CHECK(); NEED_POINTS();
  CKARG(ctx->rccl && ctx->comm, "emap_comm_init has not been called");
  CK(hipSetDevice(ctx->device));
  const emap_params& p = ctx->prm;
  const RcclApi* a = ctx->rccl;
  const bool tm = ctx->stage_timing;
  int rc;
#define STAGE(i) do { if (tm) CK(hipEventRecord(ctx->ev[i], ctx->stream)); } while (0)
  if ((rc = frame_sem_begin(ctx, p.enable_visibility_cleanup != 0))) return rc;
  ctx->in_update = true;
  ctx->gate_possible = p.enable_drift_compensation && (position_noise > p.position_noise_thresh || orientation_noise > p.orientation_noise_thresh);
  ctx->byray_frame = rays_by_ray(ctx);        // (before the sort: a by-ray frame sorts only the points of the strip's rows)
  ctx->wire_bytes = 0;
  if (ctx->pts_bucketed) {                    // emap_upload_points_strip: the cloud only holds the points of this strip's rows, for ONE pose
    if (memcmp(ctx->bucket_R, R, sizeof ctx->bucket_R) != 0 || memcmp(ctx->bucket_t, t, sizeof ctx->bucket_t) != 0) {
      ctx->in_update = false; ctx->byray_frame = false; ctx->err = "the bound cloud was bucketed for another pose (emap_upload_points_strip)"; return EMAP_ERR_INVALID; }
    if (p.enable_visibility_cleanup && !ctx->byray_frame) {
      ctx->in_update = false; ctx->err = "a bucketed cloud cannot feed a visibility pass that marches BY ROW (every valid point marches a ray through every strip): upload the whole cloud, or march by ray (emap_set_ray_mode)"; return EMAP_ERR_INVALID; }
  }
  rc = emap_count(ctx, R, t);                 // records ST_HIST / ST_SCAN / ST_SCATTER itself
  ctx->in_update = false;
  if (rc) { ctx->byray_frame = false; return rc; }          // "gate" (recorded by emap_count) = per-tile error sums + local sums + all-reduce + gate
  if (ctx->byray_frame && !ctx->frame_binned) { ctx->byray_frame = false; ctx->err = "rays by ray: this rank could not take the tile-binned path the other ranks take"; return EMAP_ERR_INVALID; }
  ctx->update_path = ctx->frame_binned ? 1 : 0;
  ctx->use_override = false;
  if ((rc = gate_impl(ctx, 0.0, 0.0, 1, ctx->comm_sums, nullptr))) return rc;                       // local sums -> device
  CKN(a->AllReduce(ctx->comm_sums, ctx->comm_sums + 2, 2, ncclFloat64, ncclSum, ctx->comm, ctx->stream));   // exchange step 1
  if (ctx->frame_binned) {        // the decision on the all-reduced totals rides in the head of the tile kernel
    memset(&ctx->gate_fold, 0, sizeof ctx->gate_fold);
    ctx->gate_fold.mode = 1; ctx->gate_fold.dev_totals = ctx->comm_sums + 2; ctx->gate_fold.A = gate_args(ctx, position_noise, orientation_noise);
    ctx->committed = false;
  } else if ((rc = gate_impl(ctx, position_noise, orientation_noise, 0, nullptr, ctx->comm_sums + 2))) return rc;
  STAGE(ST_FUSE);
  const bool fused_avg = ctx->frame_binned;
  const bool rays_on = p.enable_visibility_cleanup != 0;
  // clear_overlap_map rides on the kernel that rewrites the cells last (tile kernel, or k_ray_apply after a visibility pass)
  ctx->ov_args = overlap_args(ctx, t[2], p.enable_overlap_clearance != 0);
  const bool ov_folded = ctx->ov_args.on != 0;
  rc = fuse_impl(ctx, R, t, fused_avg, rays_on);
  const OverlapArgs ov = ctx->ov_args; ctx->ov_args.on = 0;
  if (rc) return rc;
  STAGE(ST_COMMIT);
  ctx->rays_fused = fused_avg && rays_on;
  if (rays_on) {
    if (!fused_avg && (rc = emap_commit(ctx))) return rc;
    STAGE(ST_RAYS);
    if (ctx->comm_world > 1 && normal_row_lag(ctx) != 0 && (rc = normal_exchange(ctx))) { ctx->rays_fused = false; ctx->byray_frame = false; ctx->nlag_ready = false; return rc; }
    rc = ctx->byray_frame ? rays_by_ray_pass(ctx, R, t) : emap_rays(ctx, R, t);
    ctx->byray_frame = false; ctx->nlag_ready = false;
    if (rc) { ctx->rays_fused = false; return rc; }
  } else STAGE(ST_RAYS);
  STAGE(ST_AVERAGE);
  if (!fused_avg) { launch_average(ctx->stream, ctx->kp, ctx->cells, ctx->acc, ctx->accr, ctx->frame, ctx->committed, rays_on, ctx->cnt_plane, ov); ctx->kp.mv.n = 0; }
  else if (rays_on) { launch_ray_apply(ctx->stream, ctx->kp, ctx->cells, ctx->accr, ctx->inert, ov, ctx->frame, ctx->split.need_host ? ctx->split.need_host + 1 : nullptr, ctx->ray_par ^= 1); ctx->inert_zero = true; }
  ctx->committed = false; ctx->rays_fused = false;
  CK(hipGetLastError());
  if ((rc = frame_sem_finish(ctx, R, t))) return rc;      // (as in emap_update)
  STAGE(ST_OVERLAP);
  if (p.enable_overlap_clearance && !ov_folded && (rc = emap_overlap_clear(ctx, t[2]))) return rc;
  STAGE(ST_POST);                             // "post" = halo exchange + stencils
  if (ctx->comm_world > 1) {
    if ((rc = halo_exchange_start(ctx))) return rc;                                                 // exchange step 2 ...
    if ((rc = emap_post_part(ctx, 1))) return rc;                                                   // ... overlapped with the interior tiles
    CK(hipStreamWaitEvent(ctx->stream, ctx->ev_done, 0));
    if ((rc = emap_post_part(ctx, 2))) return rc;
  } else if ((rc = emap_post_part(ctx, 0))) return rc;
  STAGE(ST_N);
#undef STAGE
  if (tm) {
    CK(hipEventSynchronize(ctx->ev[ST_N]));
    for (int i = 0; i < ST_N; ++i) CK(hipEventElapsedTime(&ctx->stage_ms[i], ctx->ev[i], ctx->ev[i + 1]));
  }
  if (stats) return emap_get_stats(ctx, stats);
  return EMAP_OK;
}

int emap_comm_wire_bytes(emap_ctx* ctx, uint64_t* bytes) {
  CKARG(ctx && bytes, "null argument");
  *bytes = (uint64_t)ctx->wire_bytes;
  return EMAP_OK;
}

int emap_comm_count(emap_ctx* ctx, int32_t* ranks) {
  CKARG(ctx && ranks && ctx->rccl && ctx->comm, "emap_comm_init has not been called");
  int n = 0;
  CKN(ctx->rccl->CommCount(ctx->comm, &n));
  *ranks = n;
  return EMAP_OK;
}

// One plane of the FULL map on every rank, assembled from the strips: every rank writes its rows (logical order) into a zeroed
// cell_n x cell_n device plane and the planes are all-reduced (x + 0 + ... + 0 is exact for every x; only -0.0 comes back as +0.0).
// Read-back for publishing on a sharded map; not on the per-frame path.
int emap_comm_gather_layer(emap_ctx* ctx, int32_t plane, float* host_full_out) {
  CKARG(ctx && ctx->rccl && ctx->comm, "emap_comm_init has not been called"); SF_CHECK();
  CKARG(host_full_out && plane >= 0 && plane < EMAP_PLANE_COUNT, "bad argument");
  CK(hipSetDevice(ctx->device));
  FLUSH();
  const long C = ctx->prm.cell_n, rows = ctx->strip.row_count;
  if (!ctx->gather_buf) CK(hipMalloc((void**)&ctx->gather_buf, sizeof(float) * (size_t)C * C));
  if (plane < 7) launch_get_plane(ctx->stream, ctx->kp, ctx->cells, plane, ctx->scratch);
  else if (plane == EMAP_PLANE_TRAV_INPUT) launch_plane_view(ctx->stream, ctx->kp, ctx->torg_r, ctx->torg_c, ctx->trav_in, ctx->scratch, 0);
  else launch_plane_view(ctx->stream, ctx->kp, ctx->kp.norg_r, ctx->kp.norg_c, ctx->normal + (long)(plane - EMAP_PLANE_NORMAL_X) * ctx->ncells_alloc, ctx->scratch, 0);
  CK(hipGetLastError());
  CK(hipMemsetAsync(ctx->gather_buf, 0, sizeof(float) * (size_t)C * C, ctx->stream));
  int32_t b = 0; emap_strip_logical_begin(ctx, &b);                       // view row j = logical row (b + j) mod cell_n
  const long first = rows < C - b ? rows : C - b;
  CK(hipMemcpyAsync(ctx->gather_buf + (size_t)b * C, ctx->scratch, sizeof(float) * (size_t)first * C, hipMemcpyDeviceToDevice, ctx->stream));
  if (first < rows) CK(hipMemcpyAsync(ctx->gather_buf, ctx->scratch + (size_t)first * C, sizeof(float) * (size_t)(rows - first) * C, hipMemcpyDeviceToDevice, ctx->stream));
  CKN(ctx->rccl->AllReduce(ctx->gather_buf, ctx->gather_buf, (size_t)C * C, ncclFloat32, ncclSum, ctx->comm, ctx->stream));
  CK(hipMemcpyAsync(host_full_out, ctx->gather_buf, sizeof(float) * (size_t)C * C, hipMemcpyDeviceToHost, ctx->stream));
  CK(hipStreamSynchronize(ctx->stream));
  return EMAP_OK;
}

int emap_comm_allreduce_host(emap_ctx* ctx, double* inout, int32_t n, int32_t op) {
  CKARG(ctx && ctx->rccl && ctx->comm, "emap_comm_init has not been called");
  CKARG(inout && n >= 1 && n <= 16 && (op == 0 || op == 1), "bad argument");
  CK(hipSetDevice(ctx->device));
  double* buf = ctx->comm_sums + 4;     // [4..20) send, [20..36) receive
  CK(hipMemcpyAsync(buf, inout, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  CKN(ctx->rccl->AllReduce(buf, buf + 16, (size_t)n, ncclFloat64, op == 0 ? ncclSum : ncclMax, ctx->comm, ctx->stream));
  CK(hipMemcpyAsync(inout, buf + 16, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  CK(hipStreamSynchronize(ctx->stream));
  return EMAP_OK;
}

// Hardware self-test of the communicator (any world size): all-reduce of known values and one halo round trip.  With a single
// rank the halo rows are exchanged with the rank itself (send to / receive from rank 0 inside one group).
int emap_comm_selftest(emap_ctx* ctx) {
  CKARG(ctx && ctx->rccl && ctx->comm, "emap_comm_init has not been called");
  CK(hipSetDevice(ctx->device));
  const RcclApi* a = ctx->rccl;
  const double mine[2] = {1.0 + ctx->comm_rank, 0.5};
  CK(hipMemcpyAsync(ctx->comm_sums, mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream));
  CKN(a->AllReduce(ctx->comm_sums, ctx->comm_sums + 2, 2, ncclFloat64, ncclSum, ctx->comm, ctx->stream));
  double got[2];
  CK(hipMemcpyAsync(got, ctx->comm_sums + 2, sizeof got, hipMemcpyDeviceToHost, ctx->stream));
  CK(hipStreamSynchronize(ctx->stream));
  const int W = ctx->comm_world;
  if (got[0] != W * (W + 1) / 2.0 || got[1] != 0.5 * W) { ctx->err = "all-reduce self-test: wrong sum"; return EMAP_ERR_COMM; }
  if (W == 1) {   // self send / recv of one row of cells through the scratch plane
    const size_t bytes = sizeof(float) * (size_t)ctx->prm.cell_n;
    float* src = ctx->scratch; float* dst = ctx->scratch + ctx->prm.cell_n;
    if ((size_t)ctx->ncells_alloc < 2 * (size_t)ctx->prm.cell_n) return EMAP_OK;
    std::string pat(bytes, 0); for (size_t i = 0; i < bytes; ++i) pat[i] = (char)(i * 7 + 3);
    CK(hipMemcpyAsync(src, pat.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
    CK(hipMemsetAsync(dst, 0, bytes, ctx->stream));
    CK(hipEventRecord(ctx->ev_ready, ctx->stream));
    CK(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_ready, 0));
    CKN(a->GroupStart());
    CKN(a->Send(src, bytes, ncclChar, 0, ctx->comm, ctx->comm_stream));
    CKN(a->Recv(dst, bytes, ncclChar, 0, ctx->comm, ctx->comm_stream));
    CKN(a->GroupEnd());
    CK(hipEventRecord(ctx->ev_done, ctx->comm_stream));
    CK(hipStreamWaitEvent(ctx->stream, ctx->ev_done, 0));
    std::string back(bytes, 0);
    CK(hipMemcpyAsync(&back[0], dst, bytes, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    if (back != pat) { ctx->err = "send/recv self-test: payload mismatch"; return EMAP_ERR_COMM; }
  } else if (ctx->strip.halo_rows > 0) {
    int rc = halo_exchange_start(ctx); if (rc) return rc;
    CK(hipStreamWaitEvent(ctx->stream, ctx->ev_done, 0));
    CK(hipStreamSynchronize(ctx->stream));
  }
  return EMAP_OK;
}

// ---- timing -----------------------------------------------------------------------------------------------------
int emap_timer_begin(emap_ctx* ctx) { CKARG(ctx, "null ctx"); CK(hipSetDevice(ctx->device)); CK(hipEventRecord(ctx->t0, ctx->stream)); return EMAP_OK; }
int emap_timer_end(emap_ctx* ctx, float* ms) {
  CKARG(ctx && ms, "null argument");
  CK(hipSetDevice(ctx->device));
  CK(hipEventRecord(ctx->t1, ctx->stream)); CK(hipEventSynchronize(ctx->t1)); CK(hipEventElapsedTime(ms, ctx->t0, ctx->t1));
  return EMAP_OK;
}
int emap_enable_stage_timing(emap_ctx* ctx, int enable) { CKARG(ctx, "null ctx"); ctx->stage_timing = enable != 0; ctx->want_ray_stats = enable > 1; return EMAP_OK; }
int emap_small_frame_aborts(emap_ctx* ctx, uint32_t* frames) { CKARG(ctx && frames, "null argument"); SF_CHECK(); *frames = ctx->sf_aborts; return EMAP_OK; }
int emap_last_update_path(emap_ctx* ctx, int32_t* path) { CKARG(ctx && path, "null argument"); *path = ctx->update_path; return EMAP_OK; }
int emap_get_stage_times(emap_ctx* ctx, float ms_out[10]) { CKARG(ctx && ms_out, "null argument"); for (int i = 0; i < ST_N; ++i) ms_out[i] = ctx->stage_ms[i]; return EMAP_OK; }

}  // extern "C"
