/* loamx.h — C ABI of libloamx.so, the MI355X (gfx950) implementation of the two hot paths of
 * DanMcGann/loam: loam::extractFeatures and loam::registerFeatures.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types. The C++ header
 * shim (the headers under include/loam/), the pybind11 module and bench.py all call through it. Each entry point
 * cites the reference interface it replaces (paths relative to the reference repo root).
 *
 * There is NO CPU fallback: every compute entry point returns LOAMX_ERR_NO_DEVICE / LOAMX_ERR_HIP
 * when no gfx950 device is usable.
 *
 * Conventions
 *   - points: row-major N x 3 FP64 (x,y,z), the layout the reference's Accessors read one by one
 *     (loam/include/loam/common.h:55-78).
 *   - pose: double[7] = {qx, qy, qz, qw, tx, ty, tz}  (Eigen coefficient order, geometry.h:27-31).
 *   - "host" entry points take host pointers and return host results (one scan / one pair);
 *     "_dev" entry points take device pointers, run on the context's stream and do not synchronise.
 */
#ifndef LOAMX_H_
#define LOAMX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct loamx_ctx loamx_ctx;

enum {
  LOAMX_OK = 0,
  LOAMX_ERR_SCAN_SIZE = 1,   /* scan size != scan_lines * points_per_line (common.h:104-113) */
  LOAMX_ERR_BAD_PARAM = 2,
  LOAMX_ERR_HIP = 3,
  LOAMX_ERR_CAPACITY = 4,    /* caller-provided output capacity too small */
  LOAMX_ERR_UNSUPPORTED = 5, /* parameter combination outside what the kernels implement */
  LOAMX_ERR_NO_DEVICE = 6,
  LOAMX_ERR_COMM = 7         /* RCCL error in the multi-GPU gather (loamx_last_error has ncclGetErrorString) */
};

/* loam::LidarParams (loam/include/loam/common.h:29-41) */
typedef struct {
  uint64_t scan_lines;
  uint64_t points_per_line;
  double min_range;
  double max_range;
} loamx_lidar_params;

/* loam::FeatureExtractionParams (loam/include/loam/features.h:37-66): same order, same defaults */
typedef struct {
  uint64_t neighbor_points;             /* 3 */
  uint64_t number_sectors;              /* 6 */
  uint64_t max_edge_feats_per_sector;   /* 10 */
  uint64_t max_planar_feats_per_sector; /* 50 */
  double edge_feat_threshold;           /* 100.0 */
  double planar_feat_threshold;         /* 1.0 */
  double occlusion_thresh;              /* 0.5 */
  double parallel_thresh;               /* 1.0 */
} loamx_fe_params;

/* loam::RegistrationParams (loam/include/loam/registration.h:40-75): same order, same defaults */
typedef struct {
  uint64_t num_edge_neighbors;        /* 5 */
  double max_edge_neighbor_dist;      /* 1.0 */
  uint64_t min_line_fit_points;       /* 3 */
  double min_line_condition_number;   /* 10 */
  uint64_t num_plane_neighbors;       /* 5 */
  double max_plane_neighbor_dist;     /* 2.0 */
  uint64_t min_plane_fit_points;      /* 4 */
  double max_avg_point_plane_dist;    /* 0.1 */
  uint64_t max_iterations;            /* 10 */
  double rotation_convergence_thresh; /* 1e-3 */
  double position_convergence_thresh; /* 1e-2 */
  uint64_t min_associations;          /* 100 */
} loamx_reg_params;

/* loam::RegistrationDetail::TerminationType (registration.h:83) */
enum { LOAMX_CONVERGED = 0, LOAMX_MAX_ITER = 1, LOAMX_INSUFFICIENT_ASSOCIATIONS = 2 };

/* Result of one registration: the returned Pose3d (registration.h:128-131) plus the termination
 * type and the number of ICF iterations executed. 64 bytes; this is the record gathered across
 * ranks in multi-GPU batch mode. */
typedef struct {
  double pose[7];
  uint32_t termination;
  uint32_t iterations;
} loamx_reg_result;

/* One RegistrationDetail::IterationInfo (registration.h:86-104) without the pair lists. */
typedef struct {
  double target_T_source_init[7];
  double estimate_update[7];
  uint32_t n_edge_associations;
  uint32_t n_plane_associations;
} loamx_iter_info;

/* Optional detail capture for loamx_register_features (RegistrationDetail, registration.h:79-109).
 * iter_info: capacity >= params.max_iterations entries. Association pair lists [source idx, nearest
 * target idx] of iteration i are written at edge_pairs + i * 2 * pairs_cap_edge (uint32 pairs) with
 * their count in n_edge_pairs[i] (same for planes); pass NULL pair pointers to skip them. */
typedef struct {
  loamx_iter_info* iter_info;
  uint32_t n_iter_info;   /* out */
  uint32_t* edge_pairs;   /* max_iterations x pairs_cap_edge x 2 */
  size_t pairs_cap_edge;  /* >= number of source edge points */
  uint32_t* n_edge_pairs; /* out, max_iterations entries */
  uint32_t* plane_pairs;
  size_t pairs_cap_plane;
  uint32_t* n_plane_pairs;
} loamx_reg_detail;

void loamx_default_fe_params(loamx_fe_params* p);
void loamx_default_reg_params(loamx_reg_params* p);
const char* loamx_status_string(int status);
const char* loamx_last_error(const loamx_ctx* ctx);

/* Context = one device + one stream + grow-on-demand device workspace. Thread-safe per context. */
int loamx_ctx_create(int device, loamx_ctx** out);
void loamx_ctx_destroy(loamx_ctx* ctx);
/* Use an external hipStream_t; NULL (handle 0) selects the context's OWN private hipStreamNonBlocking stream.
 * Careful with frameworks: torch's default stream has handle 0, i.e. it is NOT adopted — work enqueued by a "_dev"
 * entry point is then ordered only against this context's stream; call loamx_ctx_synchronize (or pass the
 * non-zero handle of an explicit side stream) before another stream or library reads the results. */
int loamx_ctx_set_stream(loamx_ctx* ctx, void* hip_stream);
int loamx_ctx_synchronize(loamx_ctx* ctx);
/* Cumulative counters of the two rare paths of the extraction kernels (synchronises the stream):
 *   tie_replays    scan lines on which two candidates of EQUAL curvature could decide a pick or the output order and
 *                  which were therefore replayed in the order libstdc++'s std::sort gives the reference (features-inl.h:38)
 *   scan_fallbacks calls in which a scan line gave up its (bounded) wait for the lines before it and the features were
 *                  gathered by the fallback kernel instead — the results are the same */
int loamx_ctx_extract_counters(loamx_ctx* ctx, uint64_t* tie_replays, uint64_t* scan_fallbacks);
/* Debug / measurement switches of ONE context (no reference counterpart). loamx_ctx_create reads the environment
 * variables LOAMX_<NAME> once as the defaults (set = 1); no entry point looks at the environment afterwards, and a
 * switch only ever affects the context it was set on. None changes a result beyond the order in which a pair's residual
 * terms are summed: NO_MOMENTS / NO_REF_MOMENTS (records streamed instead of taken through the moment matrix) and NO_SMALL_SETS
 * (the source edge features are then fed in Morton order instead of the given order: low bits of the pose, asserted by
 * tests/test_gpu_multi.py) — and CHECK_FINITE, which only adds a refusal. For the same reason the default results of two
 * RELEASES may differ in the low bits (far below the 1e-5 parity bar): round 5 began to feed the source edge features of a
 * pair whose target edge set has at most 512 points in their given order. Names — the complete list; DESIGN.md section 5:
 *   extraction:    FORCE_TIE_REPLAY, FORCE_SCAN_GIVEUP, CURV_V1, NO_FUSED_COMPACT, NO_MIS_SELECT, NO_ROW_SELECT, FUSED_EXTRACT,
 *                  FUSED_ROWS, NO_SPLIT_CURV, STAGE_ALWAYS
 *   registration:  NO_MOMENTS, NO_REF_MOMENTS, NO_PACKED_GRID, NO_BIG_GRID, NO_GRID_SIDE, NO_EXTRACT_BOXES, NO_SMALL_SETS, DEBUG_POISON,
 *                  QUEUE_TWO_STAGE, QUEUE_ONE_STAGE, NO_COOP_LEFT, NO_MIXED_ASSOC, MAP_CELLS_LOG2 (a number: 0 = default)
 *   host streaming: STREAM_CHUNK_PAIRS (a number: pairs per uploaded chunk of loamx_register_scan_pairs; 0 = default, 128)
 *   multi-GPU:     FORCE_RCCL (a one-rank communicator really enqueues the RCCL collectives)
 *   input checks:  CHECK_FINITE (see "Non-finite input" below)
 * Unknown name: LOAMX_ERR_BAD_PARAM.
 *
 * Non-finite input. The reference is undefined on NaN / Inf coordinates (loam/include/loam/features-inl.h:38 sorts on
 * curvatures computed from them; a NaN range passes every comparison of loam/src/features.cpp:30-68; nanoflann and Ceres
 * receive them as they are). Here: every HOST entry point (loamx_compute_curvature / _valid_points, loamx_extract_features,
 * loamx_register_features / _indexed, loamx_register_scan_pairs, loamx_associate, loamx_fit_lines / _planes, loamx_knn_search, loamx_target_index_create /
 * _insert, and their _f32 forms) refuses such input with LOAMX_ERR_BAD_PARAM: its uploaded copy is looked at by one small
 * kernel before anything else is launched (a 4-byte read-back, one extra stream synchronisation; an index is left as it was). The "_dev" entry points (loamx_extract_features_batch_dev, loamx_register_features_batch_dev,
 * loamx_register_scan_pairs_dev, and their _f32 forms) do not look unless the context option CHECK_FINITE is set: then one
 * small kernel and a 4-byte read-back precede the call (it synchronises) and non-finite input is refused the same way.
 * Without it their result on such input is unspecified, as the reference's. */
int loamx_ctx_set_option(loamx_ctx* ctx, const char* name, int value);
int loamx_ctx_get_option(loamx_ctx* ctx, const char* name, int* value);

/* ---- host entry points (one scan / one pair; H2D, kernels, D2H, synchronous) ------------------ */

/* loam::computeCurvature (features.h:119-122, features-inl.h:53-87): curvature_out[n_points] */
int loamx_compute_curvature(loamx_ctx* ctx, const double* xyz, size_t n_points, const loamx_lidar_params* lidar,
                            const loamx_fe_params* fe, double* curvature_out);
/* loam::computeValidPoints (features.h:166-169, features-inl.h:90-124): mask_out[n_points] in {0,1} */
int loamx_compute_valid_points(loamx_ctx* ctx, const double* xyz, size_t n_points,
                               const loamx_lidar_params* lidar, const loamx_fe_params* fe, uint8_t* mask_out);
/* loam::extractFeatures (features.h:108-111, features-inl.h:11-50). Writes the indices of the edge
 * and planar feature points in the reference's output order; the caller copies the points
 * (the C++ shim does, reproducing LoamFeatures). Capacity needed: scan_lines * number_sectors *
 * (max_*_feats_per_sector + 1). */
int loamx_extract_features(loamx_ctx* ctx, const double* xyz, size_t n_points, const loamx_lidar_params* lidar,
                           const loamx_fe_params* fe, uint32_t* edge_idx, size_t edge_cap, size_t* n_edge,
                           uint32_t* planar_idx, size_t planar_cap, size_t* n_planar);
/* FP32-input twins: xyz as n_points x 3 floats, e.g. PCL points. The reference's FieldAccessor
 * (common.h:55-60) widens every coordinate to double before any arithmetic; the kernels do the same on load,
 * so the results are those of the FP64 entry points on the widened values, bit for bit, at half the input bytes. */
int loamx_compute_curvature_f32(loamx_ctx* ctx, const float* xyz, size_t n_points, const loamx_lidar_params* lidar,
                                const loamx_fe_params* fe, double* curvature_out);
int loamx_compute_valid_points_f32(loamx_ctx* ctx, const float* xyz, size_t n_points,
                                   const loamx_lidar_params* lidar, const loamx_fe_params* fe, uint8_t* mask_out);
int loamx_extract_features_f32(loamx_ctx* ctx, const float* xyz, size_t n_points, const loamx_lidar_params* lidar,
                               const loamx_fe_params* fe, uint32_t* edge_idx, size_t edge_cap, size_t* n_edge,
                               uint32_t* planar_idx, size_t planar_cap, size_t* n_planar);
/* loam::registerFeatures (registration.h:128-131, registration-inl.h:11-78). detail may be NULL. */
int loamx_register_features(loamx_ctx* ctx, const double* src_edge, size_t n_src_edge, const double* src_planar,
                            size_t n_src_planar, const double* tgt_edge, size_t n_tgt_edge,
                            const double* tgt_planar, size_t n_tgt_planar, const double init_pose[7],
                            const loamx_reg_params* reg, loamx_reg_result* result, loamx_reg_detail* detail);

/* ---- persistent target index (scan-to-map; the reference rebuilds both KD-trees on every
 * call, registration-inl.h:20-23). Build the spatial index of a target feature set (e.g. a local map)
 * once, keep it resident on the device and register any number of source scans against it.
 * The index depends on max_edge_neighbor_dist / max_plane_neighbor_dist of `reg` (cell size);
 * loamx_register_features_indexed fails with LOAMX_ERR_BAD_PARAM if they differ. */
typedef struct loamx_target_index loamx_target_index;
int loamx_target_index_create(loamx_ctx* ctx, const double* tgt_edge, size_t n_tgt_edge, const double* tgt_planar,
                              size_t n_tgt_planar, const loamx_reg_params* reg, loamx_target_index** out);
void loamx_target_index_destroy(loamx_ctx* ctx, loamx_target_index* index);
/* Appends points to the index (a map that grows scan by scan). The new points take the indices that follow the
 * existing ones, and every search and registration against the index afterwards returns what it would against an
 * index created over the concatenated sets, bit for bit (the searches are exact: their result does not depend on
 * the cell structure). Cost: one upload of the NEW points, then per feature kind either
 *   - a merge into the existing grid (map-sized kinds: counts and scatter of the new points, one table scan and one
 *     streaming copy of the cell-sorted arrays into their twin buffers: ~0.1 ms per million resident points), or
 *   - a rebuild of that kind (scan-sized kinds: one workgroup; any kind when a new point lies outside its grid, when
 *     the kind has doubled since its grid was chosen — the cell edge follows the density — or when its buffers grew).
 * Amortised over a map grown scan by scan the rebuilds are O(log n) events. */
int loamx_target_index_insert(loamx_ctx* ctx, loamx_target_index* index, const double* edge, size_t n_edge,
                              const double* planar, size_t n_planar);
/* how the index has been maintained so far: full (re)builds of a feature kind's grid (counted per kind: creating an index
 * with both kinds counts two) / merges into an existing one. A kind that receives no points in an insert is left alone. */
int loamx_target_index_stats(const loamx_target_index* index, uint64_t* full_builds, uint64_t* merges);
/* number of edge / planar points in the index (either pointer may be NULL) */
int loamx_target_index_size(const loamx_target_index* index, size_t* n_edge, size_t* n_planar);
/* same contract as loamx_register_features, target taken from the index */
int loamx_register_features_indexed(loamx_ctx* ctx, const loamx_target_index* index, const double* src_edge,
                                    size_t n_src_edge, const double* src_planar, size_t n_src_planar,
                                    const double init_pose[7], const loamx_reg_params* reg, loamx_reg_result* result,
                                    loamx_reg_detail* detail);

/* ---- rows a16-a19 one by one (round 3): the reference's internal functions behind registerFeatures, callable from the
 * host. Same device functions the association kernels run; used by the header shim's geometry_internal / kdtree_internal
 * namespaces and by the parity tests that compare neighbour lists and fits with the oracle directly. ------------------ */

/* geometry_internal::fitLine (geometry.h:102, geometry.cpp:42-59) for n_sets point sets of k points each
 * (points: n_sets x k x 3 doubles, 2 <= k <= 32). lines_out: n_sets x 6 = {a, b} with a = centre + 0.1 dir,
 * b = centre - 0.1 dir; cond_out (may be NULL): the condition number the reference returns, i.e. DBL_MAX always
 * (geometry.cpp:55-56 computes the ratio and drops it). */
int loamx_fit_lines(loamx_ctx* ctx, const double* points, size_t n_sets, size_t k, double* lines_out, double* cond_out);
/* geometry_internal::fitPlane (geometry.h:123, geometry.cpp:62-73), 3 <= k <= 32. planes_out: n_sets x 4 = {normal, d};
 * avg_dist_out (may be NULL): the signed mean of P n - d. */
int loamx_fit_planes(loamx_ctx* ctx, const double* points, size_t n_sets, size_t k, double* planes_out, double* avg_dist_out);
/* kdtree_internal::knnSearch (kdtree.h:49, kdtree.cpp:10-28) for n_queries points against one set of a target index
 * (which_set: 0 = its edge points, 1 = its planar points; the index plays the role of the reference's KDTree):
 * exact k nearest (k <= 16), ascending, then the strict radius filter (max_dist <= 0: none). indices_out:
 * n_queries x k indices into the array the index was built from (0xFFFFFFFF past the count); counts_out: n_queries. */
int loamx_knn_search(loamx_ctx* ctx, const loamx_target_index* index, int which_set, const double* queries, size_t n_queries,
                     size_t k, double max_dist, uint32_t* indices_out, uint32_t* counts_out);
/* registration_internal::associateEdges / associatePlanes (registration.h:205-222, registration.cpp:23-103) at a given
 * estimate: ONE association pass of the registration kernels themselves (index builds, round-1 k-NN, queue chain,
 * fits), read out per source feature instead of being handed to the solver. Any pointer may be NULL (skipped).
 *   *_nn_count  n_src          neighbours that passed the radius filter (kdtree.cpp:25)
 *   *_nn_idx    n_src x k      their indices in the target array, ascending distance (0xFFFFFFFF past the count)
 *   *_valid     n_src          1: the reference would add a residual block for this point (all guards passed)
 *   *_moved     n_src x 3      pose.act(source point) (registration.cpp:34 / :75)
 *   edge_lines  n_src x 6      fitted line {a, b};  plane_planes  n_src x 4  {normal, d}   (when >= min_*_fit_points) */
typedef struct {
  uint32_t* edge_nn_count;
  uint32_t* edge_nn_idx;
  uint8_t* edge_valid;
  double* edge_moved;
  double* edge_lines;
  uint32_t* plane_nn_count;
  uint32_t* plane_nn_idx;
  uint8_t* plane_valid;
  double* plane_moved;
  double* plane_planes;
  uint32_t* queue_lengths; /* 4: queries the first k-NN pass handed on {edge, plane}, then those its second pass handed on */
} loamx_assoc_dump;
int loamx_associate(loamx_ctx* ctx, const double* src_edge, size_t n_src_edge, const double* src_planar, size_t n_src_planar,
                    const double* tgt_edge, size_t n_tgt_edge, const double* tgt_planar, size_t n_tgt_planar,
                    const double pose[7], const loamx_reg_params* reg, loamx_assoc_dump* out);

/* ---- device-resident batch entry points (asynchronous on the context stream) ------------------ */

/* Feature buffers of scan s live at base + s * stride with
 *   edge stride   = loamx_edge_capacity(lidar, fe)   entries,
 *   planar stride = loamx_planar_capacity(lidar, fe) entries. */
size_t loamx_edge_capacity(const loamx_lidar_params* lidar, const loamx_fe_params* fe);
size_t loamx_planar_capacity(const loamx_lidar_params* lidar, const loamx_fe_params* fe);

/* extractFeatures over n_scans scans stored back to back (d_xyz: n_scans x N x 3 doubles).
 * d_*_idx: uint32 indices into the scan; d_*_xyz: copies of the points (may be NULL);
 * d_n_edge / d_n_planar: one uint32 count per scan. */
int loamx_extract_features_batch_dev(loamx_ctx* ctx, const double* d_xyz, size_t n_scans,
                                     const loamx_lidar_params* lidar, const loamx_fe_params* fe,
                                     uint32_t* d_edge_idx, uint32_t* d_n_edge, double* d_edge_xyz,
                                     uint32_t* d_planar_idx, uint32_t* d_n_planar, double* d_planar_xyz);

/* the same over float scans (d_xyz: n_scans x N x 3 floats); the point copies are FP64 (widened) */
int loamx_extract_features_batch_dev_f32(loamx_ctx* ctx, const float* d_xyz, size_t n_scans,
                                         const loamx_lidar_params* lidar, const loamx_fe_params* fe,
                                         uint32_t* d_edge_idx, uint32_t* d_n_edge, double* d_edge_xyz,
                                         uint32_t* d_planar_idx, uint32_t* d_n_planar, double* d_planar_xyz);

/* registerFeatures over n_pairs independent pairs. Feature set f of pair p: points at
 * d_*[p * stride * 3], count d_n_*[p]. d_init: n_pairs x 7 doubles or NULL (identity). */
int loamx_register_features_batch_dev(loamx_ctx* ctx, size_t n_pairs, const double* d_src_edge,
                                      const uint32_t* d_n_src_edge, const double* d_src_planar,
                                      const uint32_t* d_n_src_planar, const double* d_tgt_edge,
                                      const uint32_t* d_n_tgt_edge, const double* d_tgt_planar,
                                      const uint32_t* d_n_tgt_planar, size_t edge_stride, size_t planar_stride,
                                      const double* d_init, const loamx_reg_params* reg,
                                      loamx_reg_result* d_results);

/* The north-star unit: one scan-pair registration = extractFeatures(target scan), extractFeatures(
 * source scan), registerFeatures(source, target, identity). d_xyz holds n_pairs x 2 scans, target
 * scan first. Results are device resident. */
int loamx_register_scan_pairs_dev(loamx_ctx* ctx, const double* d_xyz, size_t n_pairs,
                                  const loamx_lidar_params* lidar, const loamx_fe_params* fe,
                                  const loamx_reg_params* reg, loamx_reg_result* d_results);

/* The same unit from HOST memory to host memory — the reference's own unit of use (README.md:44-60: a scan pair in host
 * memory in, a pose out). The batch is cut into chunks of `STREAM_CHUNK_PAIRS` pairs (context option, a number; 0 = the
 * default, 128); chunk k + 1 travels to the device on a copy stream (hipMemcpyAsync into the second of two staging buffers)
 * while chunk k is registered through loamx_register_scan_pairs_dev's own path; the 64-byte results come back in one copy at
 * the end. Results are bit-identical to the "_dev" entry point's. For the upload to overlap the kernels `xyz` must be pinned
 * (hipHostMalloc / hipHostRegister); pageable memory works, one chunk at a time. Throughput is PCIe's: 3.1 MB of FP64 scans
 * per pair (1.6 MB as floats) against ~10 us of kernels. Non-finite input is refused (LOAMX_ERR_BAD_PARAM) chunk by chunk. */
int loamx_register_scan_pairs(loamx_ctx* ctx, const double* xyz, size_t n_pairs, const loamx_lidar_params* lidar,
                              const loamx_fe_params* fe, const loamx_reg_params* reg, loamx_reg_result* results);
int loamx_register_scan_pairs_f32(loamx_ctx* ctx, const float* xyz, size_t n_pairs, const loamx_lidar_params* lidar,
                                  const loamx_fe_params* fe, const loamx_reg_params* reg, loamx_reg_result* results);

/* the same over float scans */
int loamx_register_scan_pairs_dev_f32(loamx_ctx* ctx, const float* d_xyz, size_t n_pairs,
                                      const loamx_lidar_params* lidar, const loamx_fe_params* fe,
                                      const loamx_reg_params* reg, loamx_reg_result* d_results);

/* ---- multi-GPU batch mode (BASELINE configs[3]) ------------------------------------------------
 * The reference has no counterpart (registration-inl.h:11-78 takes everything by value / const-ref: scan pairs are
 * independent units). One process per GPU owns a contiguous block of pair ids and runs the single-GPU entry points
 * on it; the only communication is the gather of the 64-byte loamx_reg_result records, done here with RCCL
 * (ncclAllGather, or grouped ncclBroadcast when the shards are uneven) on the context's stream, asynchronously —
 * it is ordered after the registration kernels that produce the records by the stream itself. */
#define LOAMX_COMM_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */
typedef struct loamx_comm loamx_comm;
/* contiguous block [first, first + count) of rank `rank`; block sizes differ by at most one */
void loamx_shard_range(size_t total_pairs, int world_size, int rank, size_t* first, size_t* count);
/* rank 0 calls this once and hands the 128 bytes to every rank out of band (file, socket, launcher) */
int loamx_comm_get_unique_id(unsigned char id_out[LOAMX_COMM_ID_BYTES]);
/* collective over all ranks: ncclCommInitRank on the context's device */
int loamx_comm_create(loamx_ctx* ctx, const unsigned char id[LOAMX_COMM_ID_BYTES], int world_size, int rank,
                      loamx_comm** out);
/* adopts a communicator the host already owns (an ncclComm_t); loamx_comm_destroy then leaves it alone */
int loamx_comm_wrap(loamx_ctx* ctx, void* nccl_comm, loamx_comm** out);
void loamx_comm_destroy(loamx_comm* comm);
/* what RCCL reports for the communicator: ncclCommCount / ncclCommUserRank / ncclCommCuDevice */
int loamx_comm_info(const loamx_comm* comm, int* world_size, int* rank, int* device);
/* d_local: this rank's n_local records (device); d_all: total_pairs records (device), filled in pair-id order on
 * every rank. n_local must equal this rank's loamx_shard_range count. Asynchronous on the context's stream. */
int loamx_gather_results_dev(loamx_ctx* ctx, loamx_comm* comm, const loamx_reg_result* d_local, size_t n_local,
                             size_t total_pairs, loamx_reg_result* d_all);
/* all ranks wait for each other (1-element all-reduce + stream synchronisation); *max_value, if given, is replaced by
 * the maximum over ranks (the bench's max-over-ranks timing without a second communication library) */
int loamx_comm_barrier(loamx_ctx* ctx, loamx_comm* comm, double* max_value);
/* What the two entry points above have really enqueued on this communicator so far, by kind. A one-rank communicator
 * takes the device-copy shortcut (LOAMX_COMM_STAT_MEMCPY) unless the context's option FORCE_RCCL is set: then the
 * gather enqueues ncclAllGather AND the grouped ncclBroadcast form (in place, same bytes), the barrier ncclAllReduce —
 * the pre-flight of a box without a second GPU (tools/multi_gpu_selfcheck.py, tests/test_gpu_multi.py).
 * Environment LOAMX_RCCL_LIB names the RCCL library to dlopen instead of librccl.so.1 (a missing file: LOAMX_ERR_COMM). */
enum {
  LOAMX_COMM_STAT_ALL_GATHER = 0,
  LOAMX_COMM_STAT_BROADCAST = 1, /* members of the grouped broadcast (one per non-empty shard) */
  LOAMX_COMM_STAT_ALL_REDUCE = 2,
  LOAMX_COMM_STAT_MEMCPY = 3, /* one-rank shortcuts: no collective was enqueued */
  LOAMX_COMM_STAT_COUNT = 4
};
int loamx_comm_stats(const loamx_comm* comm, uint64_t counts[LOAMX_COMM_STAT_COUNT]);

/* ---- per-kernel timing (hipEvents on the context stream), for bench.py's roofline object ------- */
enum {
  LOAMX_K_CURVATURE = 0, /* curvature + validity, 33 B/point algorithmic */
  LOAMX_K_SELECT = 1,    /* per-sector greedy selection */
  LOAMX_K_COMPACT = 2,   /* feature gather */
  LOAMX_K_GRID = 3,      /* spatial index builds (targets: searched; sources: ordered): 24 B read + 32 B written per point,
                            + 12 B of float copies and 4 B per grid cell for the target sets that are searched by cells */
  LOAMX_K_ASSOC = 4,     /* kNN + line/plane fit */
  LOAMX_K_SWEEP = 5,     /* residual / Jacobian / normal equations, 56 B per plane + 72 B per edge slot streamed */
  LOAMX_K_LM = 6,        /* per-pair trust-region bookkeeping */
  LOAMX_K_MOMENT = 7,    /* plane moment pass (Gram matrix of the plane coefficients), 56 B per plane slot */
  LOAMX_K_KNN_PLANE = 8, /* sub-scope of LOAMX_K_ASSOC: the round-1 k-NN kernel of the plane features alone
                            (instruction-bound: bench.py prices it against the vector-issue roofline) */
  LOAMX_K_EXTRACT_FUSED = 9, /* curvature + validity + selection + compaction in one pass over the scan: 24 B/point read
                                (12 with float input) + (4 + 24) B per feature written. LOAMX_K_CURVATURE / _SELECT count
                                the separate kernels, which run when the parameters rule the fused one out */
  LOAMX_K_COUNT = 10
};
typedef struct {
  uint64_t launches;
  double total_ms;
  double algorithmic_bytes; /* summed over launches */
} loamx_kernel_stat;
int loamx_ctx_enable_kernel_timing(loamx_ctx* ctx, int enable);
int loamx_ctx_reset_kernel_stats(loamx_ctx* ctx);
/* synchronises the stream, resolves pending events, fills stats[LOAMX_K_COUNT] */
int loamx_ctx_get_kernel_stats(loamx_ctx* ctx, loamx_kernel_stat* stats);
const char* loamx_kernel_name(int kernel_id);

/* ---- synthetic workload generator (tests and bench only) ------------------------- */
/* ground-truth target_T_source of pair `pair_id` */
void loamx_synth_pair_pose(uint64_t seed, uint64_t pair_id, double pose_out[7]);
/* host generation of one scan (which: 0 = target scan A, 1 = source scan B) */
void loamx_synth_scan_host(uint64_t seed, uint64_t pair_id, uint32_t which, uint32_t scan_lines,
                           uint32_t points_per_line, double sigma, double* xyz_out);
/* device generation of n_pairs x 2 scans (target first), bit-identical to the host generator */
int loamx_synth_scan_pairs_dev(loamx_ctx* ctx, uint64_t seed, uint64_t first_pair, size_t n_pairs,
                               uint32_t scan_lines, uint32_t points_per_line, double sigma, double* d_xyz);

/* raw device memory helpers so that hosts without torch can drive the _dev entry points */
int loamx_dev_alloc(loamx_ctx* ctx, size_t bytes, void** d_ptr);
int loamx_dev_free(loamx_ctx* ctx, void* d_ptr);
int loamx_copy_to_device(loamx_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int loamx_copy_to_host(loamx_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* LOAMX_H_ */
