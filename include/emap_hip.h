/* emap_hip.h -- C ABI of the MI355X (gfx950) elevation-map fusion core.
 *
 * The reference (leggedrobotics/elevation_mapping_cupy) has no C FFI: its numeric core is a set of CuPy
 * ElementwiseKernels JIT-compiled from strings and driven by the Python class ElevationMap.  This header is
 * the drop-in boundary a maintainer binds instead (ctypes stub in INTEGRATION.md); every entry point cites
 * the reference interface it replaces.  `EM/` = elevation_mapping_cupy/script/elevation_mapping_cupy/.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * emap_status; nothing throws; no global state; one HIP stream per context; a context is
 * thread-compatible (callers serialise per context, as the reference does with map_lock).
 * Host pointers are borrowed for the duration of the call; the context owns all device memory.
 */
#ifndef EMAP_HIP_H_
#define EMAP_HIP_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4/5): emap_halo_pack / emap_halo_unpack / emap_halo_bytes move 16-byte COLD half cells (halo * cell_n * 4 floats) instead of
 * 32-byte cells; emap_upload_points de-interleaves the extra channels; emap_comm_init makes the ranks agree on this number. */
#define EMAP_ABI_VERSION 3

typedef enum {
  EMAP_OK = 0,
  EMAP_ERR_INVALID = -1,   /* bad argument */
  EMAP_ERR_HIP = -2,       /* HIP runtime error, see emap_last_error */
  EMAP_ERR_NO_POINTS = -3, /* stage needs a point cloud but none is bound */
  EMAP_ERR_UNSUPPORTED = -4,
  EMAP_ERR_COMM = -5       /* RCCL could not be loaded or a collective failed, see emap_last_error */
} emap_status;

/* index/rounding mode: 0 reproduces the reference's CuPy float16 helper-parameter rounding
 * bit for bit (valid for cell_n <= 2049); 1 = the same source with float16 := float (any cell_n), the cell index evaluated
 * in float: trunc(x * float(1/resolution) + cell_n/2), two roundings, clamped to [0, cell_n-1]. */
enum { EMAP_MODE_REFERENCE_FP16 = 0, EMAP_MODE_FP32 = 1 };

/* Mirrors the scalar fields of the reference's Parameter dataclass (EM/parameter.py:137-216) that the hot
 * path bakes into its kernels (EM/elevation_mapping.py:228-282).  Doubles hold the Python floats exactly. */
typedef struct emap_params {
  int32_t cell_n;                    /* map side incl. the 1-cell border (parameter.py:287) */
  int32_t mode;                      /* EMAP_MODE_* */
  int32_t enable_edge_sharpen, enable_visibility_cleanup, enable_drift_compensation, enable_overlap_clearance;
  int32_t dilation_size, pad_;
  double resolution, sensor_noise_factor, mahalanobis_thresh, outlier_variance;
  double drift_compensation_variance_inlier, traversability_inlier, wall_num_thresh, max_ray_length;
  double cleanup_step, cleanup_cos_thresh, min_valid_distance, max_height_range;
  double ramped_height_range_a, ramped_height_range_b, ramped_height_range_c, max_variance;
  double initial_variance, time_variance, time_interval, min_height_drift_cnt;
  double max_drift, drift_compensation_alpha, position_noise_thresh, orientation_noise_thresh;
  double overlap_clear_range_xy, overlap_clear_range_z, ray_step, reserved_;
  float w1[36], w2[36], w3[36], w_out[12]; /* traversability filter weights (EM/traversability_filter.py:15-24) */
} emap_params;

/* Row-strip placement of this context inside the global map (single GPU: row_begin 0, row_count cell_n,
 * halo_rows 0).  Cell index = cell_n * ix + iy (custom_kernels.py:45-49), so a strip is a range of ix. */
typedef struct emap_strip {
  int32_t row_begin, row_count, halo_rows, pad_;
} emap_strip;

typedef struct emap_stats {
  double err_sum;          /* sum of z - map_h over drift inliers (custom_kernels.py:331-334), local strip */
  uint32_t err_cnt;        /* number of drift inliers, local strip */
  int32_t gate_fired;      /* drift gate of elevation_mapping.py:346-354 fired this frame */
  float mean_error;        /* elevation_mapping.py:355 */
  float additive_mean_error; /* elevation_mapping.py:356 / get_additive_mean_error :412 */
  float shift;             /* amount added to the elevation plane (:357), 0 if none */
  uint32_t n_points;       /* points bound for the frame */
  uint64_t ray_visits;     /* inside-map cells visited by the visibility pass (0 unless stats requested).  Row strips marching BY ROW: the
                              visits inside this strip's rows; marching BY RAY (emap_set_ray_mode): the visits of THIS rank's rays over the
                              whole ray window -- either way the sum over the ranks is the single context's count */
} emap_stats;

typedef struct emap_ctx emap_ctx;

/* plane ids for emap_get_layer / emap_set_layer: 0..6 = the reference's elevation_map planes in its order
 * (elevation, variance, is_valid, traversability, time, upper_bound, is_upper_bound; elevation_mapping.py:68-77),
 * 7..9 = normal_map x,y,z (:81), 10 = traversability_input (dilated upper bound, :377-383). */
enum { EMAP_PLANE_ELEVATION = 0, EMAP_PLANE_VARIANCE, EMAP_PLANE_IS_VALID, EMAP_PLANE_TRAVERSABILITY, EMAP_PLANE_TIME,
       EMAP_PLANE_UPPER_BOUND, EMAP_PLANE_IS_UPPER_BOUND, EMAP_PLANE_NORMAL_X, EMAP_PLANE_NORMAL_Y,
       EMAP_PLANE_NORMAL_Z, EMAP_PLANE_TRAV_INPUT, EMAP_PLANE_COUNT };

/* ---- lifetime ------------------------------------------------------------------------------------- */
int emap_abi_version(void);
/* Replaces ElevationMap.__init__ allocation + compile_kernels (EM/elevation_mapping.py:52-117, 228-282).
 * `stream` = an existing hipStream_t to enqueue on (e.g. torch's current stream) or NULL for a private one. */
int emap_create(const emap_params* params, const emap_strip* strip_or_null, int device, void* stream, emap_ctx** out);
int emap_destroy(emap_ctx* ctx);
int emap_set_params(emap_ctx* ctx, const emap_params* params); /* same cell_n; parameters are kernargs, no JIT */
const char* emap_last_error(const emap_ctx* ctx);
int emap_sync(emap_ctx* ctx);
/* ElevationMap.clear (EM/elevation_mapping.py:119-128) */
int emap_clear(emap_ctx* ctx);

/* ---- point cloud ------------------------------------------------------------------------------------ */
/* ElevationMap.input_pointcloud's H2D + cast (EM/elevation_mapping.py:456): rows of `stride` elements,
 * xyz first; dtype 0 = float32, 1 = float64.  Rows with NaN in xyz are skipped inside the kernels (:458).
 * A cloud with extra channels (stride > 3) is de-interleaved while it is converted: on the device it is an (n, 3) xyz matrix
 * and an (n, stride - 3) channel matrix (the reference keeps the interleaved rows and strides over them in every kernel). */
int emap_upload_points(emap_ctx* ctx, const void* host, int64_t n, int64_t stride, int dtype);
/* Row strips (multi-GPU): every rank is handed the SAME sensor cloud, but only the points of its rows reach its tile kernels.  Converts,
 * uploads and binds only the points that can land in this strip's rows under the pose (R, t map-centre relative, as emap_update takes
 * them) -- a conservative superset decided on the host, order preserved, so the strip's maps stay bit-identical -- i.e. 1 / G of the
 * cloud crosses PCIe and is streamed per rank (reference side: input_pointcloud uploads the whole cloud, EM/elevation_mapping.py:434-466;
 * SURVEY 8(e) "host bucket by strip").  The frame that follows must use the same pose and must not march its rays BY ROW
 * (emap_update_sharded checks both); class_bayesian / bayesian_inference fusions need the whole cloud (global point index).  On a
 * whole-map context it is emap_upload_points.  n_kept (may be NULL): points bound. */
int emap_upload_points_strip(emap_ctx* ctx, const void* host, int64_t n, int64_t stride, int dtype, const float R[9], const float t[3], int64_t* n_kept);
/* the same predicate as data: keep[i] = 1 iff emap_upload_points_strip would upload point i (for callers that keep their clouds on
 * the device and bucket them themselves; they then declare the bound cloud with emap_declare_points_bucketed) */
int emap_strip_point_mask(emap_ctx* ctx, const void* host, int64_t n, int64_t stride, int dtype, const float R[9], const float t[3], uint8_t* keep);
int emap_declare_points_bucketed(emap_ctx* ctx, const float R[9], const float t[3], int64_t n_all);   /* n_all: size of the whole cloud */
/* bind a device-resident float32 cloud without copying (update_map_with_kernel takes device arrays, :316) */
int emap_set_points_device(emap_ctx* ctx, const float* dev, int64_t n, int64_t stride);
/* the same for a cloud that is already de-interleaved on the device (what emap_upload_points produces): xyz (n, 3) and the
 * extra channels (n, n_chan), both row-major float32; channel column c of the caller's (n, 3 + n_chan) numbering is column
 * c - 3 of chan_dev */
int emap_set_points_device_split(emap_ctx* ctx, const float* xyz_dev, const float* chan_dev, int64_t n, int64_t n_chan);
/* tail of add_points_kernel (custom_kernels.py:260-262): per point cell idx, is_valid, is_inside */
int emap_point_index(emap_ctx* ctx, const float R[9], const float t[3], int32_t* idx, uint8_t* valid, uint8_t* inside);

/* ---- one frame = update_map_with_kernel (EM/elevation_mapping.py:316-391); t already centre-relative -- */
int emap_update(emap_ctx* ctx, const float R[9], const float t[3], double position_noise, double orientation_noise,
                emap_stats* stats_or_null);
/* individually callable stages (same order inside emap_update) */
int emap_count(emap_ctx* ctx, const float R[9], const float t[3]);          /* error_counting_kernel :334-345 */
int emap_set_drift_inputs(emap_ctx* ctx, double position_noise, double orientation_noise, const double* err_sum_override,
                          const uint32_t* err_cnt_override); /* gate inputs (:346-354); overrides = all-reduced totals */
/* row strips: publish the strip-local (err_sum, err_cnt) as 2 doubles in DEVICE memory (async, no host sync) so
 * the caller can all-reduce them (RCCL), then gate on the reduced totals read from device memory. The _local_ form
 * is the blocking host variant. */
int emap_drift_sums_to_device(emap_ctx* ctx, double* dev_out2);
int emap_set_drift_inputs_device(emap_ctx* ctx, double position_noise, double orientation_noise, const double* dev_totals2);
int emap_local_drift_sums(emap_ctx* ctx, double* err_sum, uint32_t* err_cnt);
/* how the count/fuse passes scatter into the map: 0 = auto (tile-binned LDS reduction for clouds >= 131072 points, else global
 * atomics), 1 = global atomics, 2 = tile-binned. Results are bit-identical. Bins are 16x64-cell tiles; maps with more than
 * 16384 tiles stack 2, 4, ... tiles per bin. Test hook: bits 8..15 of `mode` force a minimum stacking factor (power of two). */
int emap_set_scatter_mode(emap_ctx* ctx, int32_t mode);
int emap_fuse(emap_ctx* ctx, const float R[9], const float t[3]);           /* add_points_kernel fusion part */
int emap_fuse_average(emap_ctx* ctx, const float R[9], const float t[3]);   /* fuse+commit+average when no ray pass follows */
int emap_commit(emap_ctx* ctx);                                             /* side effects of :174,:189-192 -> S1 */
int emap_rays(emap_ctx* ctx, const float R[9], const float t[3]);           /* add_points_kernel visibility part */
int emap_average(emap_ctx* ctx);                                            /* average_map_kernel :369 */
int emap_overlap_clear(emap_ctx* ctx, float t_z);                           /* clear_overlap_map :393-410 */
int emap_dilate(emap_ctx* ctx);                                             /* dilation_filter_kernel :376-383 */
int emap_traversability_normals(emap_ctx* ctx);                             /* :385-391 (filter + update_normal) */
int emap_post(emap_ctx* ctx);                                               /* the two stages above fused (used by emap_update) */
/* row strips: part 1 = tile rows independent of the halo (overlaps the halo exchange), part 2 = boundary tile rows */
int emap_post_part(emap_ctx* ctx, int32_t part);
int emap_update_variance(emap_ctx* ctx);                                    /* :420-422 */
int emap_update_time(emap_ctx* ctx);                                        /* :424-426 */
int emap_get_stats(emap_ctx* ctx, emap_stats* out);                         /* blocking D2H of the frame scalars */

/* ---- state access (elevation_map attribute / get_map_with_name_ref's raw planes; test state injection) */
int emap_get_layer(emap_ctx* ctx, int plane, float* host_out /* (row_count, cell_n) */);
int emap_set_layer(emap_ctx* ctx, int plane, const float* host_in);
/* ElevationMap.get_map_with_name_ref for the built-in layers (EM/elevation_mapping.py:579-775): fills a C-order
 * (cell_n-2, cell_n-2) buffer: border stripped, both axes flipped, NaN for unknown cells, +center_z for heights.
 * kind: 0 elevation, 1 variance, 2 traversability, 3 time, 4 upper_bound, 5 is_upper_bound, 6..8 normal_x/y/z */
int emap_publish_layer(emap_ctx* ctx, int32_t kind, float center_z, int32_t use_only_above_for_upper_bound, float* host_out);
/* ElevationMap.shift_map_xy / shift_map_z (EM/elevation_mapping.py:200-226): roll by (dx rows, dy cols) with
 * padding (0; variance plane initial_variance), planes 0 and 5 += dz -- as a rotation of the map's circular origin plus a pending
 * entry that later kernels replay: no cell is moved, only the entering band of the semantic layers is cleared.  Works on strips
 * (every rank calls it with the same arguments; a strip keeps its PHYSICAL rows, the logical rows it holds change). */
int emap_shift(emap_ctx* ctx, int32_t shift_rows, int32_t shift_cols, float dz);
/* first LOGICAL map row of the (row_count, cell_n) views emap_get_layer / emap_set_layer exchange (0 for a full map; a strip holds
 * the logical rows begin, begin + 1, ... modulo cell_n, which change with every row shift) */
int emap_strip_logical_begin(emap_ctx* ctx, int32_t* logical_row);

/* ---- RGB / semantic point-cloud fusion (EM/semantic_map.py:223-259; kernels EM/kernels/custom_semantic_kernels.py:
 * sum :9-51 + average :167-194 (kind 0) or class_average :233-267 (kind 1); add_color :270-317 + color_average :320-375;
 * kind 2 = class_bayesian (EM/fusion/pointcloud_class_bayesian.py:12-75: alpha kernel + renormalisation over the kind-2 layers
 * of the call, pseudo-counts persist in the "alpha" planes = the reference's new_map layers, semantic_map.py:54-56);
 * kind 3 = bayesian_inference (EM/fusion/pointcloud_bayesian_inference.py:12-122, restated literally).
 * Channel indices are column indices of the bound cloud (>= 3), layer indices address the semantic layer store. */
typedef struct emap_sem_spec {
  int32_t n_sum, sum_chan[16], sum_layer[16], sum_kind[16];
  int32_t n_col, col_chan[4], col_layer[4];
  double alpha; /* Parameter.average_weight */
} emap_sem_spec;
int emap_semantic_configure(emap_ctx* ctx, int32_t n_layers);                 /* SemanticMap.add_layer */
int emap_semantic_update(emap_ctx* ctx, const float R[9], const float t[3], const emap_sem_spec* spec); /* after emap_update */
/* semantic_map.update_layers_pointcloud as the reference runs it -- INSIDE update_map_with_kernel, between average_map_kernel and
 * clear_overlap_map (EM/elevation_mapping.py:366-368; EM/semantic_map.py:223-259): declares the fusion of the extra channels of
 * the bound cloud for the NEXT emap_update / emap_update_sharded call, which then fuses them before it returns (the declaration is
 * consumed by that one frame, also when the frame fails; spec_or_null == NULL or an empty spec withdraws it).  The layers hold what
 * an emap_semantic_update call after the frame would have left.  What the frame gains: a tile-binned frame without a visibility pass
 * whose fusions are average / class_average / at most one colour channel over at most four consecutive channel columns sorts 32-byte
 * records that carry those columns and fuses them in the tile pass that fuses the heights (no second pass over the records, no gather
 * of channel rows by point index, no count plane).  keep_counts != 0: the frame also leaves the per-cell accepted counts (new_map[2])
 * for a later emap_semantic_update / emap_semantic_class_max call on the same frame.  The spec is checked against the cloud that is
 * bound when the frame starts (EMAP_ERR_INVALID from that emap_update). */
int emap_frame_semantics(emap_ctx* ctx, const emap_sem_spec* spec_or_null, int32_t keep_counts);
int emap_semantic_get_layer(emap_ctx* ctx, int32_t layer, float* host_out);
int emap_semantic_set_layer(emap_ctx* ctx, int32_t layer, const float* host_in);
int emap_semantic_clear(emap_ctx* ctx);                                        /* SemanticMap.clear (layers only, :47-49) */

/* pointcloud_class_max (EM/fusion/pointcloud_class_max.py:80-126) on the bound cloud: channels chan[k] carry (half probability |
 * class id << 16) packed in a float, layer[k] the k-th of the fusion's layers (n_ch <= 8).  Per frame: sorted union of the ids in the
 * cloud and in the map's id planes; per (class, cell) the EXACT sum of the probabilities of the valid, inside points (64-bit integers
 * in units of 2^-24); per layer in turn the per-cell maximum over the classes and its id, the planes of all winning classes set to
 * zero before the next layer; the layers of a cell normalised by their sum.  The id planes are the layers' persistent planes
 * (emap_semantic_get_alpha returns them as uint32 bit patterns; they move with the map).  prev_unique / unique_out: the fusion's
 * unique_id array of the previous / this frame (n_prev = 0 before the first frame). */
int emap_semantic_class_max(emap_ctx* ctx, const float R[9], const float t[3], int32_t n_ch, const int32_t* chan, const int32_t* layer,
                            const uint32_t* prev_unique, int32_t n_prev, uint32_t* unique_out, int32_t unique_cap, int32_t* n_unique_out);
/* ---- the reference's semantic KERNEL FACTORIES on caller arrays (EM/kernels/custom_semantic_kernels.py) -------------------
 * Raw-array elementwise kernels over `size` elements; the points carry (cell index, valid, inside) in their first three columns
 * (what add_points_kernel leaves there, custom_kernels.py:260-262).  Host arrays in and out, for the staged / test surface
 * (compat/elevation_mapping_cupy/kernels); the per-frame path is emap_semantic_update.
 *   accumulate op: 0 sum_kernel (:9-51), 1 sum_compact_kernel (:54-86), 2 sum_max_kernel (:89-123), 3 alpha_kernel (:126-164),
 *                  4 add_color_kernel (:270-318; newmap_inout is the uint32 colour map of 3 n_ch + 1 planes)
 *   finalize op:   0 average_kernel (:167-194), 1 class_average_kernel (:233-267), 2 bayesian_inference_kernel (:197-230; newmap
 *                  = the variance planes, updated in place), 3 color_average_kernel (:320-375; newmap = the uint32 colour map)
 * new_elmap3: the (3, cells) planes the reference hands over, of which plane 2 (accepted points per cell) is read. */
int emap_semantic_accumulate(emap_ctx* ctx, int32_t op, const float* points, int64_t n_rows, int32_t stride, const int32_t* pcl_chan,
                             const int32_t* map_lay, int32_t n_ch, int64_t size, int64_t cells, void* newmap_inout, int32_t newmap_layers,
                             const float* max_pt, const int32_t* max_id, int32_t n_max);
int emap_semantic_finalize(emap_ctx* ctx, int32_t op, void* newmap_inout, int32_t newmap_layers, const int32_t* map_lay, int32_t n_ch, int64_t size,
                           int64_t cells, const float* new_elmap3, const float* sum_mean, int32_t sum_layers, float* map_inout, int32_t map_layers,
                           double alpha);
int emap_semantic_get_alpha(emap_ctx* ctx, int32_t layer, float* host_out);    /* SemanticMap.new_map[layer] of a class_bayesian layer */
int emap_semantic_set_alpha(emap_ctx* ctx, int32_t layer, const float* host_in);

/* MinFilter plugin (EM/plugins/min_filter.py:84-118): fills cells with valid < 0.5 by the window minimum of already
 * filled values, up to iteration_n sweeps, stops after the sweep that filled everything; NaN where still unfilled.
 * Planes are (cell_n, cell_n) host buffers handed to the plugin (PluginBase.__call__ convention); with both inputs NULL the
 * map's own elevation / is_valid planes are taken on the device (the built-in plugin: only the result crosses PCIe). */
int emap_min_filter(emap_ctx* ctx, const float* host_elevation, const float* host_valid, int32_t dilation_size,
                    int32_t iteration_n, float* host_out, int32_t* sweeps_run_or_null);

/* MaxFilter plugin (EM/plugins/max_filter.py:36-112): same sweep structure with the window maximum; a cell is filled while its
 * running mask is < 0.5 and the reference itself runs this one out of place. */
int emap_max_filter(emap_ctx* ctx, const float* host_elevation, const float* host_valid, int32_t dilation_size,
                    int32_t iteration_n, float* host_out, int32_t* sweeps_run_or_null);
/* SmoothFilter plugin (EM/plugins/smooth_filter.py:56-58): `passes` applications of a 3x3 uniform filter with the semantics of
 * scipy.ndimage.uniform_filter(size=3) ('reflect' borders, separable, float32 intermediate); the plugin uses passes = 2. */
int emap_smooth_filter(emap_ctx* ctx, const float* host_in, int32_t passes, float* host_out);

/* Erosion plugin (EM/plugins/erosion.py:96-104): what its cv2.erode(img, ones((k, k)), iterations=n) call computes -- the k x k
 * window minimum (anchor k/2, pixels outside the image ignored), n times.  OpenCV is third-party and absent: parity unpinned. */
int emap_erode(emap_ctx* ctx, const float* host_in, int32_t kernel_size, int32_t iterations, float* host_out);

/* Inpainting plugin (EM/plugins/inpainting.py:53-61) -- DOCUMENTED SUBSTITUTE for the OpenCV Telea call it makes: fills
 * the pixels with known == 0 of a (cell_n, cell_n) image holding 8-bit values, front by front, with the distance-weighted
 * mean of the known 8-neighbours. Values stay in [0, 255], integers. */
int emap_inpaint_u8(emap_ctx* ctx, const float* host_image, const float* host_known, int32_t max_sweeps, float* host_out,
                    int32_t* sweeps_run_or_null);
/* Inpainting plugin, method "telea" (EM/plugins/inpainting.py:59: cv2.inpaint(h, mask, 1, cv2.INPAINT_TELEA) on the host): Telea's
 * fast-marching fill of the pixels with mask != 0 in an 8-bit image, HOST arrays in and out, no context needed -- a serial
 * priority-queue algorithm that the reference also runs on the CPU.  A restatement of the published algorithm (OpenCV is absent
 * offline: parity with its values is not pinned).  Images of fewer than 2 x 2 pixels are rejected (EMAP_ERR_INVALID), as by
 * emap_inpaint_ns_u8: the clamped neighbour rows / columns of the image-gradient term need two of each. */
int emap_inpaint_telea_u8(const uint8_t* image, const uint8_t* mask, int32_t rows, int32_t cols, int32_t radius, uint8_t* out);
/* Inpainting plugin, method "ns" (EM/plugins/inpainting.py:33-38,59: cv2.inpaint(h, mask, 1, cv2.INPAINT_NS) on the host): the
   Navier-Stokes based fill in its fast-marching form -- same march as above, a pixel = mean of the known pixels within `radius`
   weighted along the isophote direction (csrc/emap_inpaint_ns.cpp).  HOST code, no context, same arguments and errors as
   emap_inpaint_telea_u8; images of fewer than 2 x 2 pixels are rejected (EMAP_ERR_INVALID).  Parity with OpenCV's values is NOT pinned
   (OpenCV absent offline); pinned against oracle/ns_inpaint.py. */
int emap_inpaint_ns_u8(const uint8_t* image, const uint8_t* mask, int32_t rows, int32_t cols, int32_t radius, uint8_t* out);

/* ---- camera path: ElevationMap.input_image (EM/elevation_mapping.py:468-562).
 * emap_image_correspondence = image_to_map_correspondence_kernel (EM/kernels/custom_image_kernels.py:9-157): x1, y1 = camera
 * cell (uint32 valued), z1 = camera height above the map centre, P = K [R|t] row major, D = 5 radtan coefficients (all 0 =
 * none).  emap_image_fuse = exponential_ (kind 0, alpha 0.7 in the reference) / color_ (kind 1, planes 0..2 = r,g,b) / average_
 * (kind 2: the sample replaces the value, :160-192) correspondences_to_map_kernel applied to semantic layer `layer` with a host image
 * (n_planes, H, W) float32.  emap_image_set_tolerance = the factory parameter tolerance_z_collision of the occlusion walk (:9; the
 * reference's only call passes 0.10, the default).  emap_image_fuse_arrays = the same three kernels on caller arrays (one
 * (cell_n, cell_n) semantic plane in, one out, uv (2, cell_n, cell_n), valid (cell_n, cell_n) bytes): what the kernel factories bind. */
int emap_image_correspondence(emap_ctx* ctx, float x1, float y1, float z1, const float P[12], const float K[9], const float D[5],
                              float image_height, float image_width, const float center[3]);
int emap_image_get_correspondence(emap_ctx* ctx, float* uv_host /* (2, cell_n, cell_n) */, uint8_t* valid_host);
int emap_image_fuse(emap_ctx* ctx, int32_t kind, int32_t layer, const float* host_image, int32_t n_planes, int32_t height,
                    int32_t width, double alpha);
int emap_image_set_tolerance(emap_ctx* ctx, double tolerance_z_collision);
int emap_image_fuse_arrays(emap_ctx* ctx, int32_t kind, const float* sem_plane, const float* host_image, int32_t n_planes, int32_t height,
                           int32_t width, const float* uv, const uint8_t* valid, double alpha, float* out_plane);

/* ---- safety-polygon service: polygon_mask_kernel (EM/kernels/custom_kernels.py:509-651) as launched by
 * ElevationMap.get_polygon_traversability (EM/elevation_mapping.py:837-889).  `polygon_xy` = (n, 2) float32 world
 * coordinates already clipped to the map (:851-855); writes the (cell_n, cell_n) 0/1 mask to host memory. */
int emap_polygon_mask(emap_ctx* ctx, const float* polygon_xy, int32_t n_vertices, float center_x, float center_y, float* host_mask);

/* dilation_filter_kernel (EM/kernels/custom_kernels.py:392-449) on host planes (cell_n x cell_n), `iterations` out-of-place
 * passes -- the two passes of ElevationMap.initialize_map (EM/elevation_mapping.py:914-921; in place, i.e. racy, there). */
int emap_dilate_planes(emap_ctx* ctx, const float* host_plane, const float* host_mask, int32_t dilation_size, int32_t iterations,
                       float* host_out, float* host_out_mask);

/* ---- row-strip halos (multi-GPU; exchange itself is done by the caller, e.g. torch.distributed/RCCL) ---- */
/* pack `halo_rows` owned boundary rows next to the lower (side 0) / upper (side 1) neighbour into a device buffer; unpack a
 * neighbour's rows into the halo.  Only what the stencils read of a halo row travels: the 16-byte cold half cells (time,
 * upper_bound, is_upper_bound, is_valid).  Buffers: halo_rows*cell_n*4 floats (emap_halo_bytes). */
int emap_halo_bytes(emap_ctx* ctx, int64_t* bytes_per_side);
int emap_halo_pack(emap_ctx* ctx, int side, float* dev_buf);
int emap_halo_unpack(emap_ctx* ctx, int side, const float* dev_buf);
/* The strips are physical row ranges of a circular map: neighbours form a RING (rank 0's lower neighbour is the last rank).
 * After a row shift the normal planes (not shifted in the reference) lag the cells by emap_normal_row_lag rows; when that is not
 * 0 and the visibility pass is on, the callers exchange the planes' boundary rows too (3 x halo_rows x cell_n floats per side)
 * before emap_rays.  emap_update_sharded does both by itself. */
int emap_normal_row_lag(emap_ctx* ctx, int32_t* lag);
int emap_normal_halo_pack(emap_ctx* ctx, int side, float* dev_buf);
int emap_normal_halo_unpack(emap_ctx* ctx, int side, const float* dev_buf);
/* Row strips after a ROW shift of more than halo_rows: which rank's normal rows each rank's cells belong to (emap_update_sharded
 * fetches them itself, emap_api.hip: normal_exchange; EM/elevation_mapping.py:200-214 -- normal_map is not shifted with the map).  Pure
 * host arithmetic, exposed for tests and for callers that drive the exchange themselves: rank q needs physical rows [src, src + rows),
 * owned by rank r, as rows [dst, dst + rows) of its row-aligned copy; pieces5 = {q, r, src, dst, rows} per piece, in posting order. */
int emap_normal_lag_plan(int32_t cell_n, int32_t world, const int32_t* cut_begin, const int32_t* cut_count, int32_t lag,
                         int32_t* pieces5, int32_t max_pieces, int32_t* n_pieces);

/* ---- row-strip communicator: one process per GPU, RCCL over xGMI issued from the library itself -------------
 * Nothing like it exists in the reference (single GPU).  `rccl_path` names the RCCL shared object to dlopen (NULL =
 * "librccl.so.1" from the loader path); rank 0 creates the 128-byte ncclUniqueId, the caller distributes it (any
 * bootstrap channel) and every rank calls emap_comm_init -- a collective call.  emap_update_sharded is emap_update for a
 * strip: count -> all-reduce(2 x f64 drift sums) -> gate -> fuse [-> commit -> rays] -> average -> overlap clearance ->
 * halo exchange of the boundary rows (in place, on a second stream) overlapped with the interior stencil tiles ->
 * boundary stencil tiles.  emap_comm_selftest checks an all-reduce and a send/recv round trip on the hardware.
 * emap_comm_init (world <= 16) also gathers every rank's rows (who holds which normal rows after a map shift), allocates the ray window
 * of a frame that marches its rays by ray -- so that no rank can fail alone in the middle of a frame's collectives -- and makes the
 * ranks agree on EMAP_ABI_VERSION. */
int emap_comm_unique_id(const char* rccl_path, uint8_t id_out[128]);
int emap_comm_init(emap_ctx* ctx, const char* rccl_path, const uint8_t id[128], int32_t rank, int32_t world);
int emap_comm_destroy(emap_ctx* ctx);
int emap_comm_selftest(emap_ctx* ctx);
/* number of ranks RCCL itself reports for the communicator (ncclCommCount): what a launcher prints as evidence that the
 * strips really talk through one RCCL communicator of that size */
int emap_comm_count(emap_ctx* ctx, int32_t* ranks);
/* bytes THIS rank sent + received in the exchange steps of the last frame's visibility pass when it marched BY RAY; 0 for a frame that
 * marched by row.  Round 6: the window records (32 B per window cell) are broadcast by the owners of its rows and the effects (20 B
 * per cell) are reduced to them -- grouped ncclSend / ncclRecv, an owner then folds the parts it received (k_win_reduce); an owner of
 * a share f of the window moves (W - 1) f (32 + 20) + (1 - f) (32 + 20) bytes per cell, a rank that owns none of it 52.  With
 * EMAP_BYRAY_ALLREDUCE=1, or strips that do not tile the map: the three all-reduces of rounds 4 / 5 (52 B per cell of payload on
 * every rank, moved 2 (W - 1) / W times by a ring). */
int emap_comm_wire_bytes(emap_ctx* ctx, uint64_t* bytes);
/* out-of-band reductions over the ranks through the communicator itself (barriers and timing reductions of a launcher:
 * no second bootstrap channel needed once RCCL is up): all-reduce of n <= 16 host doubles, op 0 = sum, 1 = max, in place;
 * blocks until the result is back on the host, i.e. it is also a barrier behind all work enqueued on the strip's stream. */
int emap_comm_allreduce_host(emap_ctx* ctx, double* inout, int32_t n, int32_t op);
/* one plane (EMAP_PLANE_*) of the FULL map on every rank: the strips' rows placed at their logical rows in a zeroed cell_n x cell_n
 * plane, all-reduced (exact: x + 0; -0.0 returns as +0.0).  Collective; for read-back / publishing, not on the per-frame path.
 * The reference has one map object and reads it directly (get_map_with_name_ref, elevation_mapping.py:720-775). */
int emap_comm_gather_layer(emap_ctx* ctx, int32_t plane, float* host_full_out);
int emap_update_sharded(emap_ctx* ctx, const float R[9], const float t[3], double position_noise, double orientation_noise,
                        emap_stats* stats /* may be NULL: no host synchronisation */);
/* How a sharded frame runs the visibility pass (the ray part of add_points_kernel, EM/kernels/custom_kernels.py:198-259; the
 * reference is single GPU).  Every ray starts at the sensor, so with rays marched BY ROW (each rank marches every ray through its
 * own rows) the strips around the sensor do what the whole map does.  BY RAY: every rank marches the rays of the points of its rows
 * over a replicated copy of the cells a ray can reach (one exact integer all-reduce of the window around the sensor), the effects
 * are all-reduced (sum of the validity decrements and hit counts, max of the upper-bound keys) and the owners apply their rows:
 * the same visits, bit-identical maps.  mode 0 = automatic (by ray from 2048 x 2048 cells on, frames on the tile-binned path),
 * 1 = always by row, 2 = by ray whenever the frame is on the tile-binned path.  Every rank must use the same mode. */
int emap_set_ray_mode(emap_ctx* ctx, int32_t mode);

/* ---- timing on the context's stream (hipEvents; bench.py's roofline leg) -------------------------- */
int emap_timer_begin(emap_ctx* ctx);
int emap_timer_end(emap_ctx* ctx, float* elapsed_ms); /* records, synchronises, returns elapsed */
/* per-stage device times of the last emap_update when profiling is enabled (order: hist, scan, scatter, gate, fuse,
 * commit, rays, average, overlap, post; hist/scan are 0 and "scatter" is the count kernel on the global-atomic path);
 * enabling inserts hipEvents between the stages (each costs a few microseconds of its own) */
int emap_enable_stage_timing(emap_ctx* ctx, int enable);
int emap_get_stage_times(emap_ctx* ctx, float ms_out[10]);
/* which kernels the last emap_update ran for phases count .. commit / average (results are bit-identical on all three):
 * 0 = chain of launches with global atomics (k_count, k_fuse, k_commit / k_average), 1 = tile-binned (sort front-end + tile kernels),
 * 2 = ONE launch, k_small_frame: small clouds on maps of up to 512^2 cells -- the robot-scale configuration the reference ships
 * (EM/parameter.py:137,165 -> 202^2 cells; EM/elevation_mapping.py:316-391 is a chain of ~12 dependent launches there).
 * EMAP_SMALL_FRAME=0 in the environment keeps such frames on path 0.
 * Bits 2-3 (the value & 3 is the path above): how the frame fused the channels declared by emap_frame_semantics -- 4 = inside the tile
 * kernel that fused the heights (32-byte sorted records), 8 = 32-byte records, stand-alone semantic kernel (a launch with heavy-tile
 * parts), neither = stand-alone kernels on 16-byte records / the atomic path.  EMAP_SEM_CARRY=0 keeps every frame on the last form. */
int emap_last_update_path(emap_ctx* ctx, int32_t* path);
/* k_small_frame synchronises its own grid (two barriers inside ONE launch).  If foreign work on the device keeps part of that grid from
 * starting for ~0.1 s, the launch aborts -- one compare-and-swap decides for the whole grid -- and leaves the map, the frame
 * accumulators and the drift record bit for bit as it found them; small frames already queued behind it do nothing.  The library re-runs
 * those frames, in order, on the chain of launches (path 0) before the next call on the context reads or changes anything: a frame
 * always completes, as the reference's does (EM/elevation_mapping.py:316-391), only later.  Precondition inherited from
 * emap_set_points_device: a device cloud stays valid until the frame that reads it has completed (emap_sync).  *frames = how many
 * frames have been re-run this way since emap_create (normally 0).  Test hooks: EMAP_SF_TEST_ABORT=1|2 (workgroup 0 gives up at once
 * at that barrier), EMAP_SF_SPIN_LIMIT=<polls> (a waiter's patience). */
int emap_small_frame_aborts(emap_ctx* ctx, uint32_t* frames);

#ifdef __cplusplus
}
#endif
#endif
