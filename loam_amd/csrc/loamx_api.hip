// loamx_api.hip — the C ABI of libloamx.so (include/loamx.h): context, device workspace,
// orchestration of the extraction and registration kernels, per-kernel hipEvent timing.
// No CPU compute path exists here: without a usable HIP device every entry point fails.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <utility>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library itself is opened on first use (struct Rccl below)

#include "loamx_internal.h"
#include "synth.h"

using namespace loamx;

namespace {

enum WsId {
  WS_XYZ = 0, WS_CURV, WS_MASK, WS_EDGE_STAGE, WS_PLANAR_STAGE, WS_EDGE_CNT, WS_PLANAR_CNT,
  WS_EDGE_IDX, WS_PLANAR_IDX, WS_N_EDGE, WS_N_PLANAR, WS_EDGE_XYZ, WS_PLANAR_XYZ,
  WS_GRID_DESC_E, WS_GRID_DESC_P, WS_CELLS_E, WS_CELLS_P, WS_SORTED_E, WS_SORTED_P, WS_REL_E, WS_REL_P,
  WS_SGRID_DESC_E, WS_SGRID_DESC_P, WS_SCELLS_E, WS_SCELLS_P, WS_SSORTED_E, WS_SSORTED_P, WS_SORT_SCRATCH, WS_SORT_SCRATCH_SRC, WS_ASSOC_E, WS_ASSOC_P, WS_NN_E, WS_NN_P, WS_RNN_E, WS_RNN_P, WS_NEAREST_E, WS_NEAREST_P, WS_REST_E, WS_REST_P, WS_EXACT_E, WS_EXACT_P, WS_NASSOC, WS_STATE, WS_PARTIALS, WS_MOM_PARTIALS, WS_MOMENTS, WS_FLAGGED_LIST, WS_FLAGGED_COUNT, WS_LINE_TOT, WS_EXTRACT_EVENTS, WS_BOX, WS_FINITE_FLAG,
  WS_COUNTERS, WS_ITERINFO, WS_STREAM_IN0, WS_STREAM_IN1, WS_STREAM_RES, WS_DUMP_E, WS_DUMP_P, WS_FIT_IN, WS_FIT_OUT, WS_SRC_E, WS_SRC_P, WS_TGT_E, WS_TGT_P, WS_FCOUNTS, WS_RESULTS, WS_INIT,
  WS_COUNT
};

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
};

struct PendingEvent {
  int kernel;
  hipEvent_t e0, e1;
  double bytes;
  bool own_e0;  // false: e0 is the previous scope's e1 (back-to-back scopes share the event between them)
};

}  // namespace

struct loamx_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  bool max_counts_clean = false;     // write_results_kernel has reset RegBatch::max_counts for the next call
  hipStream_t aux_stream = nullptr;  // edge association chain, forked from / joined into `stream` with the two events
  hipEvent_t ev_fork = nullptr, ev_mid = nullptr, ev_join = nullptr;
  hipEvent_t ev_counts = nullptr;     // marks the read-back of the largest source set sizes (register_dev)
  hipStream_t aux2_stream = nullptr;  // the plane queue chain (so that it does not wait behind the edge chain)
  hipEvent_t ev_join2 = nullptr;
  std::string last_error;
  Buf ws[WS_COUNT];
  uint32_t* h_pinned = nullptr;  // small pinned readback area
  bool timing = false;
  std::vector<PendingEvent> pending;
  std::vector<hipEvent_t> event_pool;
  hipEvent_t tail_event = nullptr;  // end event of the last timed scope ...
  bool tail_fresh = false;          // ... and nothing has been enqueued on the stream since
  loamx_kernel_stat stats[LOAMX_K_COUNT] = {};

  // debug / measurement switches (loamx_ctx_set_option; defaults from LOAMX_<NAME>, read once at creation)
  uint32_t extract_flags = 0;  // kFlag* of extract_math.h
  uint32_t reg_flags = 0;      // kRegFlag* of loamx_internal.h
  int map_cells_log2 = 0;      // cell table of a map-sized persistent index (0: kGridMapCellsCap)
  int stream_chunk_pairs = 0;  // pairs per uploaded chunk of loamx_register_scan_pairs (0: kStreamChunkPairs)
  hipStream_t copy_stream = nullptr;  // uploads of loamx_register_scan_pairs (created on first use)
  hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};

  unsigned long long sweep_slots_base[2] = {0, 0};
  unsigned long long features_base = 0;  // events[2] at the last loamx_ctx_reset_kernel_stats
  std::mutex mu;
};

struct loamx_target_index {
  GridDesc* desc[2] = {nullptr, nullptr};       // [edge, plane], one GridDesc each
  uint32_t* cells[2] = {nullptr, nullptr};      // kGridCellsCap + 1 entries each
  GridPoint* sorted[2] = {nullptr, nullptr};    // n + kGridPad entries each
  float* rel[2] = {nullptr, nullptr};           // 3 x (n + kGridPad) single-precision offsets (FP32 pre-selection)
  double* pts[2] = {nullptr, nullptr};          // the points in insertion order (index = `orig` of the sorted copy)
  size_t n[2] = {0, 0};
  size_t cap[2] = {0, 0};                       // points the buffers above hold without growing
  double radius[2] = {0, 0};
  uint32_t* counts = nullptr;                   // device copy of n[] for the build kernels
  void* scratch = nullptr;                      // box keys + cursors of the multi-workgroup build
  uint32_t cells_cap[2] = {0, 0};               // 0: kGridCellsCap; map-sized sets own a larger cell table
  size_t cells_alloc[2] = {0, 0}, scratch_alloc = 0;
  // incremental insert (index_merge): twin buffers of sorted / rel, the set size at the last full build of the kind
  // (a kind that has doubled since is rebuilt: its cell edge is chosen for the density it had then), event counters
  GridPoint* sorted2[2] = {nullptr, nullptr};
  float* rel2[2] = {nullptr, nullptr};
  size_t n_at_build[2] = {0, 0};
  bool grid_valid[2] = {false, false};  // the kind's cell-sorted arrays + table describe idx->pts[k][0 .. n[k]) (full build or merges since)
  uint64_t full_builds = 0, merges = 0;  // per kind: a call that rebuilds both kinds counts two
};

namespace loamx {
thread_local LaunchScope* g_launch_scope = nullptr;
int g_debug_sync = getenv("LOAMX_DEBUG_SYNC") ? 1 : 0;
}

namespace {

const char* kKernelNames[LOAMX_K_COUNT] = {"curvature_valid_kernel", "select_kernel", "compact_kernel",
                                           "grid_build_kernel",      "associate_kernel", "sweep_kernel",
                                           "lm_kernels",             "moment_kernel",    "knn_plane_kernel",
                                           "extract_fused_kernel"};

struct OptionName {
  const char* name;
  bool extract;  // bit of loamx_ctx::extract_flags, else of reg_flags
  uint32_t bit;
};
const OptionName kOptionNames[] = {
    {"FORCE_TIE_REPLAY", true, kFlagForceReplay}, {"FORCE_SCAN_GIVEUP", true, kFlagForceGiveUp}, {"CURV_V1", true, kFlagCurvV1},
    {"NO_FUSED_COMPACT", true, kFlagNoFusedCompact}, {"NO_MIS_SELECT", true, kFlagNoMisSelect}, {"FUSED_EXTRACT", true, kFlagFusedExtract},
    {"NO_ROW_SELECT", true, kFlagNoRowSelect}, {"FUSED_ROWS", true, kFlagFusedRows}, {"NO_SPLIT_CURV", true, kFlagNoSplitCurv}, {"STAGE_ALWAYS", true, kFlagStageAlways},
    {"NO_MOMENTS", false, kRegFlagNoMoments}, {"NO_PACKED_GRID", false, kRegFlagNoPackedGrid}, {"NO_BIG_GRID", false, kRegFlagNoBigGrid},
    {"NO_GRID_SIDE", false, kRegFlagNoGridSide}, {"DEBUG_POISON", false, kRegFlagPoison},
    {"QUEUE_TWO_STAGE", false, kRegFlagQueueTwoStage}, {"QUEUE_ONE_STAGE", false, kRegFlagQueueOneStage},
    {"NO_MIXED_ASSOC", false, kRegFlagNoMixedAssoc}, {"FORCE_RCCL", false, kRegFlagForceRccl},
    {"NO_COOP_LEFT", false, kRegFlagNoCoopLeft}, {"NO_REF_MOMENTS", false, kRegFlagNoRefMoments},
    {"NO_EXTRACT_BOXES", false, kRegFlagNoExtractBoxes}, {"CHECK_FINITE", false, kRegFlagCheckFinite},
    {"NO_SMALL_SETS", false, kRegFlagNoSmallSets}};

int fail(loamx_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->last_error = msg;
  return code;
}

#define HIP_TRY(ctx, expr)                                                                          \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      return fail(ctx, LOAMX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
  } while (0)

int ensure(loamx_ctx* ctx, int id, size_t bytes) {
  Buf& b = ctx->ws[id];
  if (bytes == 0) bytes = 8;
  if (b.cap >= bytes) return LOAMX_OK;
  if (b.p) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(b.p));
    b.p = nullptr, b.cap = 0;
  }
  HIP_TRY(ctx, hipMalloc(&b.p, bytes));
  b.cap = bytes;
  return LOAMX_OK;
}
#define ENSURE(ctx, id, bytes)                       \
  do {                                               \
    int rc_ = ensure(ctx, id, bytes);                \
    if (rc_ != LOAMX_OK) return rc_;                 \
  } while (0)

// ---- non-finite input (loamx.h: "Non-finite input") -------------------------------------------------------------------
// The reference is undefined on NaN / Inf coordinates (features-inl.h:38 sorts on curvatures computed from them; nanoflann and
// Ceres receive them as they are). Host entry points refuse such input; "_dev" entry points look only when the context option
// CHECK_FINITE is set (one small kernel + a 4-byte read-back).
const char* const kNonFiniteMsg = "non-finite coordinate in the input (the reference's behaviour is undefined there)";
bool host_all_finite(const void* p, bool f32, size_t n_scalars) {
  if (!p) return true;
  if (f32) {
    const float* v = static_cast<const float*>(p);
    float acc = 0.0f;
    for (size_t i = 0; i < n_scalars; i++) acc += v[i] * 0.0f;  // 0 unless some v[i] is NaN or infinite
    return acc == 0.0f;
  }
  const double* v = static_cast<const double*>(p);
  double acc = 0.0;
  for (size_t i = 0; i < n_scalars; i++) acc += v[i] * 0.0;
  return acc == 0.0;
}
template <typename T>
T* wsp(loamx_ctx* ctx, int id) {
  return reinterpret_cast<T*>(ctx->ws[id].p);
}

// timing helpers ------------------------------------------------------------------------------------
hipEvent_t take_event(loamx_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

struct TimedScope {
  loamx_ctx* ctx;
  PendingEvent pe;
  bool on, attach;
  LaunchScope ls;
  // attach: the events ride on the scope's own kernels (single-stream scopes); otherwise marker events around it
  TimedScope(loamx_ctx* c, int kernel, double bytes, bool attach_ = false) : ctx(c), on(c->timing), attach(attach_) {
    if (on) {
      pe.kernel = kernel, pe.bytes = bytes;
      if (attach) {
        pe.own_e0 = true;
        pe.e0 = take_event(ctx), pe.e1 = take_event(ctx);
        ls = LaunchScope{pe.e0, pe.e1, true};
        g_launch_scope = &ls;
        return;
      }
      // back-to-back scopes share one event: half the event packets between the kernels
      pe.own_e0 = !(ctx->tail_fresh && ctx->tail_event);
      pe.e0 = pe.own_e0 ? take_event(ctx) : ctx->tail_event;
      pe.e1 = take_event(ctx);
      if (pe.own_e0) (void)hipEventRecord(pe.e0, ctx->stream);
    }
  }
  ~TimedScope() {
    if (on) {
      if (attach) {
        g_launch_scope = nullptr;
        if (ls.first) {  // no kernel was launched inside: nothing to time
          ctx->event_pool.push_back(pe.e0), ctx->event_pool.push_back(pe.e1);
        } else {
          ctx->pending.push_back(pe);
        }
        ctx->tail_fresh = false;
        return;
      }
      (void)hipEventRecord(pe.e1, ctx->stream);
      ctx->pending.push_back(pe);
      ctx->tail_event = pe.e1, ctx->tail_fresh = true;
    }
  }
};
// call before enqueueing anything outside a TimedScope: the next scope must record its own start
inline void untimed(loamx_ctx* ctx) { ctx->tail_fresh = false; }

// The check itself runs on the device in every case (a CPU loop over a 128 x 2048 scan costs more than its upload): zero the
// flag word, one finite_kernel launch per array, a 4-byte read-back. Host entry points do it on their uploaded copies before
// they launch anything else (one extra stream synchronisation, ~30 us); "_dev" entry points only under CHECK_FINITE.
int finite_begin(loamx_ctx* ctx) {
  ENSURE(ctx, WS_FINITE_FLAG, 16);
  untimed(ctx);
  HIP_TRY(ctx, hipMemsetAsync(ctx->ws[WS_FINITE_FLAG].p, 0, 16, ctx->stream));
  return LOAMX_OK;
}
// d_n == nullptr: `stride` points per set
void finite_add(loamx_ctx* ctx, const void* d_pts, bool f32, const uint32_t* d_n, size_t n_sets, size_t stride, uint32_t pitch) {
  if (d_pts && n_sets && stride) launch_check_finite(d_pts, f32, d_n, n_sets, stride, pitch, static_cast<uint32_t*>(ctx->ws[WS_FINITE_FLAG].p), ctx->stream);
}
int finite_end(loamx_ctx* ctx) {
  uint32_t bad = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&bad, ctx->ws[WS_FINITE_FLAG].p, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return bad ? fail(ctx, LOAMX_ERR_BAD_PARAM, kNonFiniteMsg) : LOAMX_OK;
}
int dev_check_finite(loamx_ctx* ctx, const void* d_pts, bool f32, const uint32_t* d_n, size_t n_sets, size_t stride, uint32_t pitch, bool force = false) {
  if ((!force && !(ctx->reg_flags & kRegFlagCheckFinite)) || !d_pts || n_sets == 0 || stride == 0) return LOAMX_OK;
  int rc = finite_begin(ctx);
  if (rc != LOAMX_OK) return rc;
  finite_add(ctx, d_pts, f32, d_n, n_sets, stride, pitch);
  return finite_end(ctx);
}


int check_launch(loamx_ctx* ctx, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ctx, LOAMX_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
  return LOAMX_OK;
}
#define CHECK_LAUNCH(ctx, what)                      \
  do {                                               \
    int rc_ = check_launch(ctx, what);               \
    if (rc_ != LOAMX_OK) return rc_;                 \
  } while (0)

// parameter translation -------------------------------------------------------------------------------
int make_extract_params(loamx_ctx* ctx, const loamx_lidar_params* lidar, const loamx_fe_params* fe, ExtractParams& P) {
  if (!lidar || !fe) return fail(ctx, LOAMX_ERR_BAD_PARAM, "null parameter struct");
  if (fe->number_sectors == 0) return fail(ctx, LOAMX_ERR_BAD_PARAM, "number_sectors must be >= 1");
  if (fe->neighbor_points == 0)
    return fail(ctx, LOAMX_ERR_BAD_PARAM, "neighbor_points must be >= 1 (the reference reads point idx-1)");
  if (fe->neighbor_points > (uint64_t)kMaxNeighborPoints)
    return fail(ctx, LOAMX_ERR_UNSUPPORTED, "neighbor_points > 16 not supported by the kernels");
  if (lidar->points_per_line > (uint64_t)kMaxLineWidth)
    return fail(ctx, LOAMX_ERR_UNSUPPORTED, "points_per_line > 4096 not supported by the kernels");
  if (lidar->scan_lines * lidar->points_per_line > 0xFFFFFFFFull / 4)
    return fail(ctx, LOAMX_ERR_UNSUPPORTED, "scan too large for 32-bit point indices");
  if (fe->number_sectors > 65535) return fail(ctx, LOAMX_ERR_UNSUPPORTED, "number_sectors > 65535");
  P = ExtractParams{};
  P.H = (uint32_t)lidar->scan_lines, P.W = (uint32_t)lidar->points_per_line;
  P.np = (uint32_t)fe->neighbor_points, P.S = (uint32_t)fe->number_sectors;
  P.pps = P.W / P.S;
  const uint64_t longest = (uint64_t)P.W - (uint64_t)(P.S - 1) * P.pps;  // last sector takes the remainder
  const uint64_t lim = 0x7FFFFFFFull;
  P.max_edge = (uint32_t)(fe->max_edge_feats_per_sector < lim ? fe->max_edge_feats_per_sector : lim);
  P.max_planar = (uint32_t)(fe->max_planar_feats_per_sector < lim ? fe->max_planar_feats_per_sector : lim);
  P.cap_edge = (uint32_t)((uint64_t)P.max_edge + 1 < longest ? (uint64_t)P.max_edge + 1 : longest);
  P.cap_planar = (uint32_t)((uint64_t)P.max_planar + 1 < longest ? (uint64_t)P.max_planar + 1 : longest);
  if (P.cap_edge == 0) P.cap_edge = 1;
  if (P.cap_planar == 0) P.cap_planar = 1;
  P.min_range = lidar->min_range, P.max_range = lidar->max_range;
  P.edge_thr = fe->edge_feat_threshold, P.planar_thr = fe->planar_feat_threshold;
  P.occ_thr = fe->occlusion_thresh, P.par_thr = fe->parallel_thresh;
  P.flags = ctx ? ctx->extract_flags : 0u;  // the context's switches: results never depend on them
  return LOAMX_OK;
}

int make_reg_config(loamx_ctx* ctx, const loamx_reg_params* r, RegConfig& C) {
  if (!r) return fail(ctx, LOAMX_ERR_BAD_PARAM, "null registration params");
  if (r->num_edge_neighbors > (uint64_t)kMaxK || r->num_plane_neighbors > (uint64_t)kMaxK)
    return fail(ctx, LOAMX_ERR_UNSUPPORTED, "num_*_neighbors > 16 not supported by the kernels");
  if (r->min_plane_fit_points < 3 && r->num_plane_neighbors > 0)
    return fail(ctx, LOAMX_ERR_BAD_PARAM, "min_plane_fit_points must be >= 3");
  if (r->min_line_fit_points < 2 && r->num_edge_neighbors > 0)
    return fail(ctx, LOAMX_ERR_BAD_PARAM, "min_line_fit_points must be >= 2");
  if (r->max_iterations > 1000) return fail(ctx, LOAMX_ERR_UNSUPPORTED, "max_iterations > 1000");
  memset(&C, 0, sizeof(C));
  C.k_edge = (int)r->num_edge_neighbors, C.k_plane = (int)r->num_plane_neighbors;
  C.min_line_pts = (int)(r->min_line_fit_points > 64 ? 64 : r->min_line_fit_points);
  C.min_plane_pts = (int)(r->min_plane_fit_points > 64 ? 64 : r->min_plane_fit_points);
  C.r_edge = r->max_edge_neighbor_dist, C.r_plane = r->max_plane_neighbor_dist;
  C.pass_edge = knn_radius_pass_max(C.r_edge), C.pass_plane = knn_radius_pass_max(C.r_plane);
  C.min_line_cond = r->min_line_condition_number, C.max_avg_plane_dist = r->max_avg_point_plane_dist;
  C.max_iterations = (uint32_t)r->max_iterations;
  C.rot_thresh = r->rotation_convergence_thresh, C.pos_thresh = r->position_convergence_thresh;
  C.min_associations = (uint32_t)(r->min_associations > 0xFFFFFFFFull ? 0xFFFFFFFFull : r->min_associations);
  C.flags = ctx ? ctx->reg_flags : 0u;
  return LOAMX_OK;
}

size_t edge_capacity(const ExtractParams& P) { return (size_t)P.H * P.S * P.cap_edge; }
size_t planar_capacity(const ExtractParams& P) { return (size_t)P.H * P.S * P.cap_planar; }

// Bounding boxes of the feature sets of every scan, taken by the selection's copy phase (select_rows_kernel with the fused
// compaction): min / max[scan][kind][axis] as ordered keys; valid while *bad == 0 (a tied or given-up line sends its scan through
// compact_kernel, which takes no boxes). All nullptr when the extraction went another way.
struct ExtractBoxes {
  const unsigned long long *min = nullptr, *max = nullptr;
  const uint32_t* bad = nullptr;
};

// extraction over device-resident scans ------------------------------------------------------------------
// d_xyz: double, or float when f32 (FP32-input path)
int extract_dev(loamx_ctx* ctx, const void* d_xyz, bool f32, size_t n_scans, const ExtractParams& P, uint32_t* d_edge_idx,
                uint32_t* d_n_edge, double* d_edge_xyz, uint32_t* d_planar_idx, uint32_t* d_n_planar,
                double* d_planar_xyz, bool only_curvature_mask, ExtractBoxes* boxes = nullptr) {
  const size_t N = (size_t)P.H * P.W;
  if (boxes) *boxes = ExtractBoxes{};
  if (n_scans == 0) return LOAMX_OK;
  untimed(ctx);
  if (N == 0) {
    if (!only_curvature_mask) {
      untimed(ctx);
      HIP_TRY(ctx, hipMemsetAsync(d_n_edge, 0, n_scans * sizeof(uint32_t), ctx->stream));
      HIP_TRY(ctx, hipMemsetAsync(d_n_planar, 0, n_scans * sizeof(uint32_t), ctx->stream));
    }
    return LOAMX_OK;
  }
  ENSURE(ctx, WS_CURV, n_scans * N * sizeof(double));
  ENSURE(ctx, WS_MASK, n_scans * N);
  if (only_curvature_mask) {  // loamx_compute_curvature / loamx_compute_valid_points: the standalone kernel
    {
      TimedScope t(ctx, LOAMX_K_CURVATURE, (double)n_scans * (double)N * (f32 ? 21.0 : 33.0), true);
      launch_curvature_valid(d_xyz, f32, n_scans, P, wsp<double>(ctx, WS_CURV), wsp<uint8_t>(ctx, WS_MASK), ctx->stream);
    }
    CHECK_LAUNCH(ctx, "curvature_valid_kernel");
    return LOAMX_OK;
  }
  const size_t groups = n_scans * P.H * P.S;
  ENSURE(ctx, WS_EDGE_STAGE, groups * P.cap_edge * sizeof(uint32_t));
  ENSURE(ctx, WS_PLANAR_STAGE, groups * P.cap_planar * sizeof(uint32_t));
  ENSURE(ctx, WS_EDGE_CNT, groups * sizeof(uint32_t));
  ENSURE(ctx, WS_PLANAR_CNT, groups * sizeof(uint32_t));
  ExtractStage st{wsp<uint32_t>(ctx, WS_EDGE_STAGE), wsp<uint32_t>(ctx, WS_PLANAR_STAGE),
                  wsp<uint32_t>(ctx, WS_EDGE_CNT), wsp<uint32_t>(ctx, WS_PLANAR_CNT)};
  // the selection writes the final feature arrays itself when it can (launch_select); its chained scan over the
  // lines of a scan needs the per-line slots zeroed
  // ... and one more word behind them: the give-up flag of that chained scan (zeroed by the same memset)
  const size_t n_lines = n_scans * P.H;
  ENSURE(ctx, WS_LINE_TOT, (n_lines + 1) * sizeof(unsigned long long));
  {
    const bool fresh = ctx->ws[WS_EXTRACT_EVENTS].cap == 0;  // cumulative counters: zeroed once
    ENSURE(ctx, WS_EXTRACT_EVENTS, 4 * sizeof(unsigned long long));
    untimed(ctx);
    if (fresh) HIP_TRY(ctx, hipMemsetAsync(ctx->ws[WS_EXTRACT_EVENTS].p, 0, 4 * sizeof(unsigned long long), ctx->stream));
  }
  untimed(ctx);
  uint32_t* d_gave_up = reinterpret_cast<uint32_t*>(wsp<unsigned long long>(ctx, WS_LINE_TOT) + n_lines);
  unsigned long long* d_events = wsp<unsigned long long>(ctx, WS_EXTRACT_EVENTS);
  ExtractFused fz{wsp<unsigned long long>(ctx, WS_LINE_TOT), 0u, d_xyz, f32 ? 1u : 0u, d_edge_idx, d_n_edge, d_edge_xyz,
                  edge_capacity(P), d_planar_idx, d_n_planar, d_planar_xyz, planar_capacity(P), d_gave_up, d_events, nullptr, nullptr};
  if (boxes && d_edge_xyz && d_planar_xyz && launch_select_takes_boxes(P)) {
    ENSURE(ctx, WS_BOX, 2 * n_scans * 6 * sizeof(unsigned long long));
    fz.box_min = wsp<unsigned long long>(ctx, WS_BOX), fz.box_max = fz.box_min + n_scans * 6;
  }
  untimed(ctx);
  launch_extract_init(fz.line_tot, n_lines + 1, fz.box_min, fz.box_max, fz.box_min ? n_scans * 6 : 0, ctx->stream);
  CHECK_LAUNCH(ctx, "extract_init_kernel");
  {
    // rows a5-a10 in one pass over the scan — opt-in (context option FUSED_EXTRACT; the two kernels below are the default,
    // see launch_extract_fused) and only where the parameters allow: 24 B/point read + (4 + 24) B per feature written;
    // the features are counted on the device (events[2]) for the roofline figure
    TimedScope t(ctx, LOAMX_K_EXTRACT_FUSED, (double)n_scans * (double)N * (f32 ? 12.0 : 24.0), true);
    if (launch_extract_fused(d_xyz, f32, n_scans, P, st, fz, wsp<double>(ctx, WS_CURV), wsp<uint8_t>(ctx, WS_MASK), ctx->stream)) {
      launch_replay(wsp<double>(ctx, WS_CURV), wsp<uint8_t>(ctx, WS_MASK), n_scans, P, st, fz, ctx->stream);
      launch_compact(d_xyz, f32, n_scans, P, st, d_edge_idx, d_n_edge, d_edge_xyz, edge_capacity(P), d_planar_idx, d_n_planar,
                     d_planar_xyz, planar_capacity(P), ctx->stream, d_gave_up, d_events + 1);
      return check_launch(ctx, "extract_fused_kernel");
    }
  }
  {
    // rows a5-a10 in one kernel, four scan lines per wavefront (round 5; select_rows.h) — opt-in (context option FUSED_ROWS):
    // bit-identical, but measured slower than the two kernels (EXPERIMENTS.md round 5)
    TimedScope t(ctx, LOAMX_K_EXTRACT_FUSED, (double)n_scans * (double)N * (f32 ? 12.0 : 24.0), true);
    if (launch_extract_rows_fused(d_xyz, f32, n_scans, P, st, fz, wsp<double>(ctx, WS_CURV), wsp<uint8_t>(ctx, WS_MASK), ctx->stream)) {
      const bool fused_compact = P.S <= 64 && !(P.flags & kFlagNoFusedCompact);
      launch_replay(wsp<double>(ctx, WS_CURV), wsp<uint8_t>(ctx, WS_MASK), n_scans, P, st, fz, ctx->stream);
      launch_compact(d_xyz, f32, n_scans, P, st, d_edge_idx, d_n_edge, d_edge_xyz, edge_capacity(P), d_planar_idx, d_n_planar,
                     d_planar_xyz, planar_capacity(P), ctx->stream, fused_compact ? d_gave_up : nullptr, fused_compact ? d_events + 1 : nullptr);
      return check_launch(ctx, "select_rows_kernel (fused)");
    }
  }
  // Round 5: between these two kernels the curvature travels as hi words | lo words with the validity in the sign bit where
  // both know that form (kFlagSplitCurv; the selection then reads 4 instead of 9 bytes per point)
  ExtractParams Pk = P;
  if (launch_extract_split_ok(P, n_scans)) Pk.flags |= kFlagSplitCurv;
  const bool split = (Pk.flags & kFlagSplitCurv) != 0u;
  {
    TimedScope t(ctx, LOAMX_K_CURVATURE, (double)n_scans * (double)N * ((f32 ? 21.0 : 33.0) - (split ? 1.0 : 0.0)), true);
    launch_curvature_valid(d_xyz, f32, n_scans, Pk, wsp<double>(ctx, WS_CURV), wsp<uint8_t>(ctx, WS_MASK), ctx->stream);
  }
  CHECK_LAUNCH(ctx, "curvature_valid_kernel");
  bool fused = false, rows_ran = false;
  {
    TimedScope t(ctx, LOAMX_K_SELECT, (double)n_scans * (double)N * (split ? 4.0 : 9.0), true);
    fused = launch_select(wsp<double>(ctx, WS_CURV), wsp<uint8_t>(ctx, WS_MASK), n_scans, Pk, st, &fz, ctx->stream, &rows_ran);
    // scan lines on which a curvature tie can decide something: again, in the reference's std::sort order (a no-op without)
    launch_replay(wsp<double>(ctx, WS_CURV), wsp<uint8_t>(ctx, WS_MASK), n_scans, Pk, st, fz, ctx->stream);
    // A scan line whose wavefront gave up waiting for the lines before it (bounded wait: unusual scheduling) left its
    // features in the stage arrays; this launch then gathers the batch from them and is a no-op otherwise (every
    // workgroup reads the flag and leaves): the call stays asynchronous and never fails for that reason.
    if (fused)
      launch_compact(d_xyz, f32, n_scans, P, st, d_edge_idx, d_n_edge, d_edge_xyz, edge_capacity(P), d_planar_idx, d_n_planar,
                     d_planar_xyz, planar_capacity(P), ctx->stream, d_gave_up, d_events + 1);
  }
  CHECK_LAUNCH(ctx, "select_kernel");
  // (the split curvature form is understood by select_rows_kernel alone: had launch_select refused it on a condition
  // launch_extract_split_ok does not share, the other selection kernels would have read hi / lo words as doubles)
  if (split && !rows_ran) return fail(ctx, LOAMX_ERR_UNSUPPORTED, "internal: split curvature form without the row selection");
  if (fused && rows_ran && boxes && fz.box_min) boxes->min = fz.box_min, boxes->max = fz.box_max, boxes->bad = d_gave_up;
  if (fused) return LOAMX_OK;
  {
    TimedScope t(ctx, LOAMX_K_COMPACT, 0.0, true);
    launch_compact(d_xyz, f32, n_scans, P, st, d_edge_idx, d_n_edge, d_edge_xyz, edge_capacity(P), d_planar_idx,
                   d_n_planar, d_planar_xyz, planar_capacity(P), ctx->stream);
  }
  CHECK_LAUNCH(ctx, "compact_kernel");
  return LOAMX_OK;
}

// registration over device-resident feature sets ------------------------------------------------------
struct RegInputs {
  size_t n_pairs, edge_stride, planar_stride;
  uint32_t in_pitch;
  const double *src_edge, *src_planar, *tgt_edge, *tgt_planar;
  const uint32_t *n_src_edge, *n_src_planar, *n_tgt_edge, *n_tgt_planar;
  const double* init;
  ExtractBoxes boxes;  // (optional) bounding boxes of the sets, left by the extraction that produced them
};

// host-side hook called after the association kernels of iteration `it` (detail capture)
typedef int (*AfterAssocHook)(loamx_ctx*, const RegBatch&, uint32_t it, void* user);

// the launch sequence of one ICF iteration on ctx->stream (+ the auxiliary streams); `it` only selects between the
// first iteration's sequence (no moment pass) and the later ones'
// the association kernels of one ICF iteration (all feature kinds, all chains), timed as LOAMX_K_ASSOC
int enqueue_association(loamx_ctx* ctx, const RegBatch& B, const RegConfig& C, uint32_t what = kAssocEdges | kAssocPlanes) {
  hipStream_t s = ctx->stream;
    {
      TimedScope t(ctx, LOAMX_K_ASSOC, 0.0);
      // (a single scan-sized pair cannot fill the chip twice over: the forks and joins only add latency there —
      // one 64 x 1024 pair 1.15 ms with the auxiliary streams, 1.10 ms without; one 128 x 2048 scan against a
      // 1 M-point map the other way round: 5.9 vs 6.8 ms)
      const bool side = (size_t)B.n_pairs * B.assoc_blocks_plane >= 128;
      // sub-scope: the plane round-1 k-NN kernel alone (events attached to its own dispatch)
      PendingEvent knn{};
      LaunchScope knn_ls{nullptr, nullptr, true};
      if (ctx->timing) {
        knn.kernel = LOAMX_K_KNN_PLANE, knn.bytes = 0.0, knn.own_e0 = true;
        knn.e0 = take_event(ctx), knn.e1 = take_event(ctx);
        knn_ls = LaunchScope{knn.e0, knn.e1, true};
      }
      launch_associate(B, C, s, side ? ctx->aux_stream : nullptr, side ? ctx->aux2_stream : nullptr, ctx->ev_fork, ctx->ev_mid, ctx->ev_join,
                       ctx->ev_join2, ctx->timing ? &knn_ls : nullptr, what);
      if (ctx->timing) {
        if (knn_ls.first) ctx->event_pool.push_back(knn.e0), ctx->event_pool.push_back(knn.e1);  // (kernel not launched)
        else ctx->pending.push_back(knn);
      }
    }
    CHECK_LAUNCH(ctx, "associate_kernel");
  return LOAMX_OK;
}

int enqueue_icf_iteration(loamx_ctx* ctx, const RegBatch& B, const RegConfig& C, uint32_t it, AfterAssocHook hook, void* hook_user) {
  hipStream_t s = ctx->stream;
    {
      const int rc_assoc = enqueue_association(ctx, B, C);
      if (rc_assoc != LOAMX_OK) return rc_assoc;
    }
#ifdef LOAMX_NN_SAME_STATS
    debug_nn_same(B, it, s);
#endif
    if (hook) {
      untimed(ctx);
      int rc = hook(ctx, B, it, hook_user);
      if (rc != LOAMX_OK) return rc;
    }
    {
      TimedScope t(ctx, LOAMX_K_LM, 0.0, true);
      launch_lm_begin(B, C, s);
    }
    // The first ICF iteration (state_init: stream_planes = 1, use_moments = 0): ONE sweep of the records at the identity update
    // and its bookkeeping step, which fixes the first candidate; the moments are then taken at that candidate and the solve
    // goes on as in the later iterations (round 4; until then five sweeps + five steps: 1.07 of the step's 11.3 ms).
    const bool ref_first = it == 0 && B.ref_moments != 0u;
    if (ref_first) {
      {
        TimedScope t(ctx, LOAMX_K_SWEEP, 0.0, true);
        launch_sweep(B, s);
      }
      TimedScope t(ctx, LOAMX_K_LM, 0.0, true);
      launch_lm_step(B, s);
    }
    if (it > 0 || ref_first) {
      TimedScope t(ctx, LOAMX_K_MOMENT, 0.0, true);
      launch_moments(B, s);
    }
    if (it > 0 || ref_first) {
      // one workgroup per pair runs the whole solve off the moments (and streams by itself whatever they cannot cover)
      TimedScope t(ctx, LOAMX_K_LM, 0.0, true);
      launch_lm_pair_loop(B, C, s);  // (ends with the pair's outer update)
    } else {
      for (int k = 0; k < 5; k++) {  // iteration-0 evaluation + max_num_iterations = 4 candidates
        {
          TimedScope t(ctx, LOAMX_K_SWEEP, 0.0, true);
          launch_sweep(B, s);
        }
        {
          TimedScope t(ctx, LOAMX_K_LM, 0.0, true);
          launch_lm_step(B, s);
        }
      }
    }
    CHECK_LAUNCH(ctx, "sweep/lm kernels");
    if (it == 0 && !ref_first) {  // (otherwise: inside lm_pair_loop_kernel; the active-pair counter is reset by lm_begin_kernel)
      TimedScope t(ctx, LOAMX_K_LM, 0.0, true);
      launch_outer_update(B, C, s);
    }
    CHECK_LAUNCH(ctx, "outer_update_kernel");
  return LOAMX_OK;
}

// dump != nullptr (loamx_associate): index builds + ONE association pass at the initial estimate, read out into the
// host arrays of *dump (one pair); no solve, d_results untouched
int register_dev(loamx_ctx* ctx, const RegInputs& in, const RegConfig& C_in, loamx_reg_result* d_results, bool want_iter_info,
                 AfterAssocHook hook, void* hook_user, const loamx_target_index* prebuilt = nullptr, const loamx_assoc_dump* dump = nullptr,
                 size_t dump_n_se = 0, size_t dump_n_sp = 0) {
  if (in.n_pairs == 0) return LOAMX_OK;
  RegConfig C = C_in;
  // The reference's associateEdges / associatePlanes (registration.cpp:23-103) do not know max_iterations; here a pair with
  // max_iterations == 0 is never active (state_init_kernel), so its association kernels would return at once and the dump
  // would read workspace nobody wrote. One association pass needs one iteration's worth of "active".
  if (dump && C.max_iterations == 0) C.max_iterations = 1;
  untimed(ctx);
  if (in.n_pairs > 0x7FFFFFFFull / 128) return fail(ctx, LOAMX_ERR_UNSUPPORTED, "too many pairs in one call");
  const size_t np = in.n_pairs, es = in.edge_stride ? in.edge_stride : 1, ps = in.planar_stride ? in.planar_stride : 1;
  if (es > 0x3FFFFFFFull || ps > 0x3FFFFFFFull) return fail(ctx, LOAMX_ERR_UNSUPPORTED, "feature set too large");  // (queue entries: 30-bit query index + 2 flags)
  RegBatch B{};
  B.n_pairs = np, B.edge_stride = es, B.planar_stride = ps, B.in_pitch = in.in_pitch;
  B.src_edge = in.src_edge, B.n_src_edge = in.n_src_edge, B.src_planar = in.src_planar, B.n_src_planar = in.n_src_planar;
  B.tgt_edge = in.tgt_edge, B.n_tgt_edge = in.n_tgt_edge, B.tgt_planar = in.tgt_planar, B.n_tgt_planar = in.n_tgt_planar;
  B.init = in.init;
  if (!(C.flags & kRegFlagNoExtractBoxes)) B.box_min = in.boxes.min, B.box_max = in.boxes.max, B.box_bad = in.boxes.bad;
  ENSURE(ctx, WS_GRID_DESC_E, np * sizeof(GridDesc));
  ENSURE(ctx, WS_GRID_DESC_P, np * sizeof(GridDesc));
  ENSURE(ctx, WS_CELLS_E, (np * (size_t)(kGridCellsCap + 1) + 4) * sizeof(uint32_t));  // (+4: the search reads four entries at a time)
  ENSURE(ctx, WS_CELLS_P, (np * (size_t)(kGridCellsCap + 1) + 4) * sizeof(uint32_t));
  ENSURE(ctx, WS_SORTED_E, np * (es + kGridPad) * sizeof(GridPoint));
  ENSURE(ctx, WS_SORTED_P, np * (ps + kGridPad) * sizeof(GridPoint));
  ENSURE(ctx, WS_REL_E, np * 3 * (es + kGridPad) * sizeof(float));
  ENSURE(ctx, WS_REL_P, np * 3 * (ps + kGridPad) * sizeof(float));
  ENSURE(ctx, WS_SGRID_DESC_E, np * sizeof(GridDesc));
  ENSURE(ctx, WS_SGRID_DESC_P, np * sizeof(GridDesc));
  ENSURE(ctx, WS_SCELLS_E, np * (size_t)(kGridCellsCap + 1) * sizeof(uint32_t));
  ENSURE(ctx, WS_SCELLS_P, np * (size_t)(kGridCellsCap + 1) * sizeof(uint32_t));
  ENSURE(ctx, WS_SSORTED_E, np * es * sizeof(GridPoint));
  ENSURE(ctx, WS_SSORTED_P, np * ps * sizeof(GridPoint));
  ENSURE(ctx, WS_SORT_SCRATCH, np * (es > ps ? es : ps) * sizeof(GridPoint));
  // (sets the packed build takes are ordered in LDS and need no scratch: the predicate is the launcher's own)
  ENSURE(ctx, WS_SORT_SCRATCH_SRC, grid_small(es > ps ? es : ps, C.flags) ? sizeof(GridPoint) : np * (es > ps ? es : ps) * sizeof(GridPoint));
  ENSURE(ctx, WS_ASSOC_E, 9 * np * es * sizeof(double));
  ENSURE(ctx, WS_ASSOC_P, 7 * np * ps * sizeof(double));
  // (neighbour lists: the count + as many positions as the kernel variant in use keeps: 5, 8 or 16 — launch_associate)
  const size_t km_e = C.k_edge <= 5 ? 5 : (C.k_edge <= 8 ? 8 : kMaxK), km_p = C.k_plane <= 5 ? 5 : (C.k_plane <= 8 ? 8 : kMaxK);
  ENSURE(ctx, WS_NN_E, (1 + km_e) * np * es * sizeof(uint32_t));
  ENSURE(ctx, WS_NN_P, (1 + km_p) * np * ps * sizeof(uint32_t));
  ENSURE(ctx, WS_RNN_E, (1 + km_e) * np * es * sizeof(uint32_t));
  ENSURE(ctx, WS_RNN_P, (1 + km_p) * np * ps * sizeof(uint32_t));
  ENSURE(ctx, WS_NEAREST_E, np * es * sizeof(uint32_t));
  ENSURE(ctx, WS_NEAREST_P, np * ps * sizeof(uint32_t));
  ENSURE(ctx, WS_REST_E, np * es * sizeof(uint32_t));
  ENSURE(ctx, WS_REST_P, np * ps * sizeof(uint32_t));
  ENSURE(ctx, WS_EXACT_E, np * es * sizeof(uint32_t));
  ENSURE(ctx, WS_EXACT_P, np * ps * sizeof(uint32_t));
  ENSURE(ctx, WS_NASSOC, np * 8 * sizeof(uint32_t));
  ENSURE(ctx, WS_STATE, np * sizeof(PairState));
  B.blocks_per_pair = (uint32_t)((es + ps + kSweepChunk - 1) / kSweepChunk);
  ENSURE(ctx, WS_PARTIALS, np * B.blocks_per_pair * kAccSize * sizeof(double));
  B.mom_blocks_per_pair = (uint32_t)((ps + kSweepChunk - 1) / kSweepChunk);
  ENSURE(ctx, WS_MOM_PARTIALS, np * B.mom_blocks_per_pair * 4 * (size_t)(kMomSize + 2) * sizeof(double));
  ENSURE(ctx, WS_MOMENTS, np * (size_t)(kMomSize + 2) * sizeof(double));
  ENSURE(ctx, WS_FLAGGED_LIST, np * (size_t)B.mom_blocks_per_pair * kSweepChunk * sizeof(uint32_t));
  ENSURE(ctx, WS_FLAGGED_COUNT, np * (size_t)B.mom_blocks_per_pair * 4 * sizeof(uint32_t));
  {
    const bool fresh = ctx->ws[WS_COUNTERS].cap == 0;
    ENSURE(ctx, WS_COUNTERS, 128);
    untimed(ctx);
    if (fresh) HIP_TRY(ctx, hipMemsetAsync(ctx->ws[WS_COUNTERS].p, 0, 128, ctx->stream));
  }
  if (C.flags & kRegFlagPoison) {  // debugging: every scratch buffer of the registration starts as 0xFF bytes
    static const int kScratch[] = {WS_GRID_DESC_E, WS_GRID_DESC_P, WS_CELLS_E, WS_CELLS_P, WS_SORTED_E, WS_SORTED_P, WS_REL_E, WS_REL_P,
                                   WS_SGRID_DESC_E, WS_SGRID_DESC_P, WS_SCELLS_E, WS_SCELLS_P, WS_SSORTED_E, WS_SSORTED_P, WS_SORT_SCRATCH, WS_SORT_SCRATCH_SRC,
                                   WS_ASSOC_E, WS_ASSOC_P, WS_NN_E, WS_NN_P, WS_RNN_E, WS_RNN_P, WS_NEAREST_E, WS_NEAREST_P, WS_REST_E,
                                   WS_REST_P, WS_EXACT_E, WS_EXACT_P, WS_NASSOC, WS_STATE, WS_PARTIALS, WS_MOM_PARTIALS, WS_MOMENTS,
                                   WS_FLAGGED_LIST, WS_FLAGGED_COUNT};
    for (int id : kScratch)
      if (ctx->ws[id].p) HIP_TRY(ctx, hipMemsetAsync(ctx->ws[id].p, 0xFF, ctx->ws[id].cap, ctx->stream));
  }
  if (want_iter_info) ENSURE(ctx, WS_ITERINFO, np * (size_t)(C.max_iterations ? C.max_iterations : 1) * sizeof(loamx_iter_info));
  B.grid_edge = GridSet{wsp<GridDesc>(ctx, WS_GRID_DESC_E), wsp<uint32_t>(ctx, WS_CELLS_E), wsp<GridPoint>(ctx, WS_SORTED_E), es + kGridPad,
                        wsp<float>(ctx, WS_REL_E)};
  B.grid_plane = GridSet{wsp<GridDesc>(ctx, WS_GRID_DESC_P), wsp<uint32_t>(ctx, WS_CELLS_P), wsp<GridPoint>(ctx, WS_SORTED_P), ps + kGridPad,
                         wsp<float>(ctx, WS_REL_P)};
  B.src_grid_edge = GridSet{wsp<GridDesc>(ctx, WS_SGRID_DESC_E), wsp<uint32_t>(ctx, WS_SCELLS_E), wsp<GridPoint>(ctx, WS_SSORTED_E), es, nullptr};
  B.src_grid_plane = GridSet{wsp<GridDesc>(ctx, WS_SGRID_DESC_P), wsp<uint32_t>(ctx, WS_SCELLS_P), wsp<GridPoint>(ctx, WS_SSORTED_P), ps, nullptr};
  B.sort_scratch = wsp<GridPoint>(ctx, WS_SORT_SCRATCH);
  B.sort_scratch_src = wsp<GridPoint>(ctx, WS_SORT_SCRATCH_SRC);
  B.assoc = AssocBuffers{wsp<double>(ctx, WS_ASSOC_E), wsp<double>(ctx, WS_ASSOC_P), wsp<uint32_t>(ctx, WS_NN_E),
                         wsp<uint32_t>(ctx, WS_NN_P), wsp<uint32_t>(ctx, WS_RNN_E), wsp<uint32_t>(ctx, WS_RNN_P),
                         wsp<uint32_t>(ctx, WS_NEAREST_E),
                         wsp<uint32_t>(ctx, WS_NEAREST_P), wsp<uint32_t>(ctx, WS_REST_E), wsp<uint32_t>(ctx, WS_REST_P),
                         wsp<uint32_t>(ctx, WS_EXACT_E), wsp<uint32_t>(ctx, WS_EXACT_P),
                         wsp<uint32_t>(ctx, WS_NASSOC)};
  B.state = wsp<PairState>(ctx, WS_STATE);
  B.partials = wsp<double>(ctx, WS_PARTIALS);
  B.mom_partials = wsp<double>(ctx, WS_MOM_PARTIALS);
  B.moments = wsp<double>(ctx, WS_MOMENTS);
  B.flagged_list = wsp<uint32_t>(ctx, WS_FLAGGED_LIST);
  B.flagged_count = wsp<uint32_t>(ctx, WS_FLAGGED_COUNT);
  // counters: [0] n_active (u32), [8..48) sweep / association / moment slot counters (5 x u64), [48..56) index-build bytes
  B.n_active = wsp<uint32_t>(ctx, WS_COUNTERS);
  B.sweep_slots = reinterpret_cast<unsigned long long*>(wsp<unsigned char>(ctx, WS_COUNTERS) + 8);
  B.assoc_slots = B.sweep_slots + 2;
  B.grid_bytes = B.sweep_slots + 5;
  B.iter_info = want_iter_info ? wsp<loamx_iter_info>(ctx, WS_ITERINFO) : nullptr;
  B.want_nearest = hook ? 1u : 0u;
  B.ref_moments = (C.flags & (kRegFlagNoMoments | kRegFlagNoRefMoments)) ? 0u : 1u;
  B.max_counts = wsp<uint32_t>(ctx, WS_COUNTERS) + 16;  // bytes 64..88
  B.assoc_blocks_edge = B.assoc_blocks_plane = 0xFFFFFFFFu;
  B.knn_mode_edge = B.knn_mode_plane = 0u;
  B.small_edge_sets = (C.flags & kRegFlagNoSmallSets) ? 0u : (!prebuilt ? 1u : (prebuilt->n[0] <= kBruteMax ? 2u : 0u));
  hipStream_t s = ctx->stream;
  // The per-pair state first: its kernel also finds the largest source sets, which come back to the host while
  // the index builds run (an event right behind the copy: the builds are already queued when the host waits).
  untimed(ctx);
  if (!ctx->max_counts_clean) {  // (write_results_kernel leaves them ready for the next call: no copy kernel in front of state_init_kernel then)
    const uint32_t init[6] = {0u, 0u, 0u, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu};
    memcpy(&ctx->h_pinned[8], init, sizeof(init));
    HIP_TRY(ctx, hipMemcpyAsync(B.max_counts, &ctx->h_pinned[8], sizeof(init), hipMemcpyHostToDevice, s));
  }
  ctx->max_counts_clean = false;
  launch_state_init(B, C, s);
  CHECK_LAUNCH(ctx, "state_init_kernel");
  if (!ctx->ev_counts) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_counts, hipEventDisableTiming));
  HIP_TRY(ctx, hipMemcpyAsync(&ctx->h_pinned[16], B.max_counts, 6 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipEventRecord(ctx->ev_counts, s));

  if (prebuilt) {  // persistent target index: only the source sets are (re)ordered
    B.grid_edge = GridSet{prebuilt->desc[0], prebuilt->cells[0], prebuilt->sorted[0], prebuilt->cap[0] + kGridPad, prebuilt->rel[0], prebuilt->cells_cap[0]};
    B.grid_plane = GridSet{prebuilt->desc[1], prebuilt->cells[1], prebuilt->sorted[1], prebuilt->cap[1] + kGridPad, prebuilt->rel[1], prebuilt->cells_cap[1]};
  }
  {
    TimedScope t(ctx, LOAMX_K_GRID, 0.0);
    // the source builds next to the target builds on the auxiliary stream: every workgroup of one build kernel is in
    // the same phase at the same time (reads, then writes), two different kernels side by side even the HBM demand
    // out (measured 1.19 -> 1.11 ms per step). For a single pair too since round 3 (the two planar builds are one
    // workgroup of 45 and 83 us each: side by side 100 instead of 150 us; one pair 0.82 -> 0.80 ms).
    // (Round 3: the first ICF iteration's edge chains right behind the edge builds on the auxiliary stream, i.e. next to
    // the planar index builds instead of next to the plane k-NN kernel: 11.60 / 11.72 vs 11.65 / 11.53 ms per step —
    // what the k-NN kernel gains the HBM-bound builds lose. Not kept.)
    // (the fork waits on the event already recorded behind the counts' copy: one marker packet fewer in front of the builds)
    const bool side = !(C.flags & kRegFlagNoGridSide) && ctx->aux_stream && !prebuilt && hipStreamWaitEvent(ctx->aux_stream, ctx->ev_counts, 0) == hipSuccess;
    if (!prebuilt) launch_grid_build_targets(B, C, s);
    launch_grid_build_sources(B, C, side ? ctx->aux_stream : s);
    if (side && hipEventRecord(ctx->ev_join, ctx->aux_stream) == hipSuccess) (void)hipStreamWaitEvent(s, ctx->ev_join, 0);
  }
  CHECK_LAUNCH(ctx, "grid_build_kernel");
  HIP_TRY(ctx, hipEventSynchronize(ctx->ev_counts));
  B.assoc_blocks_edge = (ctx->h_pinned[16] + 255u) / 256u;   // kAssocThreads queries per workgroup
  B.assoc_blocks_plane = (ctx->h_pinned[17] + 255u) / 256u;
  {  // which of the two k-NN kernels of a feature kind has work at all (target sizes relative to kBruteMax)
    uint32_t tmax[2] = {ctx->h_pinned[18], ctx->h_pinned[19]}, tmin[2] = {ctx->h_pinned[20], ctx->h_pinned[21]};
    if (prebuilt) tmax[0] = tmin[0] = (uint32_t)prebuilt->n[0], tmax[1] = tmin[1] = (uint32_t)prebuilt->n[1];
    uint32_t* mode[2] = {&B.knn_mode_edge, &B.knn_mode_plane};
    for (int k = 0; k < 2; k++) *mode[k] = tmin[k] > tmax[k] ? 0u : (tmin[k] > kBruteMax ? 1u : (tmax[k] <= kBruteMax ? 2u : 0u));
  }
  if (dump) {
    if (np != 1) return fail(ctx, LOAMX_ERR_BAD_PARAM, "association dump: one pair at a time");
    int rc_a = enqueue_association(ctx, B, C);
    if (rc_a != LOAMX_OK) return rc_a;
    untimed(ctx);
    const size_t n[2] = {dump_n_se, dump_n_sp}, kq[2] = {(size_t)C.k_edge, (size_t)C.k_plane}, pw[2] = {6, 4};
    AssocDumpSet D[2] = {};
    for (int kind = 0; kind < 2; kind++) {
      const size_t bytes = n[kind] * (24 + 8 * pw[kind] + 4 + 4 * kq[kind] + 1) + 64;
      ENSURE(ctx, kind ? WS_DUMP_P : WS_DUMP_E, bytes);
      unsigned char* base = wsp<unsigned char>(ctx, kind ? WS_DUMP_P : WS_DUMP_E);
      HIP_TRY(ctx, hipMemsetAsync(base, 0, bytes, s));  // (features beyond the capacity, if any, read as "no neighbours")
      D[kind].moved = reinterpret_cast<double*>(base);
      D[kind].prim = D[kind].moved + 3 * n[kind];
      D[kind].nn_count = reinterpret_cast<uint32_t*>(D[kind].prim + pw[kind] * n[kind]);
      D[kind].nn_idx = D[kind].nn_count + n[kind];
      D[kind].valid = reinterpret_cast<uint8_t*>(D[kind].nn_idx + kq[kind] * n[kind]);
      if (n[kind] == 0) D[kind].nn_count = nullptr;
    }
    launch_assoc_dump(B, C, D[0], D[1], s);
    CHECK_LAUNCH(ctx, "assoc_dump_kernel");
    struct Out { void* host; const void* dev; size_t bytes; };
    const Out outs[10] = {
        {dump->edge_nn_count, D[0].nn_count, 4 * n[0]}, {dump->edge_nn_idx, D[0].nn_idx, 4 * kq[0] * n[0]}, {dump->edge_valid, D[0].valid, n[0]},
        {dump->edge_moved, D[0].moved, 24 * n[0]}, {dump->edge_lines, D[0].prim, 48 * n[0]},
        {dump->plane_nn_count, D[1].nn_count, 4 * n[1]}, {dump->plane_nn_idx, D[1].nn_idx, 4 * kq[1] * n[1]}, {dump->plane_valid, D[1].valid, n[1]},
        {dump->plane_moved, D[1].moved, 24 * n[1]}, {dump->plane_planes, D[1].prim, 32 * n[1]}};
    for (const Out& o : outs)
      if (o.host && o.bytes && o.dev) HIP_TRY(ctx, hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, s));
    if (dump->queue_lengths) HIP_TRY(ctx, hipMemcpyAsync(dump->queue_lengths, B.assoc.n_assoc + 2, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return LOAMX_OK;
  }
  // (Replaying an ICF iteration as a hipGraph was measured in round 2 — captured once, cached, ~110 kernel nodes over
  // three streams: 13.6 vs 13.1 ms per 1 024-pair step and 1.06 vs 1.07 ms for one pair. The GPU-side turnaround of
  // dependent kernels bounds both, not the host's launch rate, and the graph loses the stream priorities; removed.)
  for (uint32_t it = 0; it < C.max_iterations; it++) {
    {
      const int rc_body = enqueue_icf_iteration(ctx, B, C, it, hook, hook_user);
      if (rc_body != LOAMX_OK) return rc_body;
    }
    if (it + 1 < C.max_iterations && it > (B.n_pairs >= 256 ? 1u : 0u)) {
      // one 4-byte readback per outer iteration: stop as soon as every pair has terminated. Not after the first
      // iteration: a registration that starts more than the convergence thresholds away from its answer cannot
      // converge there, so the second iteration is enqueued without waiting (if every pair did stop — too few
      // associations everywhere — its kernels find no active pair and return). A large batch does not wait after
      // the second iteration either: that hundreds of pairs all stop there is as good as excluded, and the
      // synchronisation idles the GPU for ~35 us (a single pair does wait: a third iteration would cost it 0.3 ms)
      // The results go out BEFORE the host waits: when this was the last iteration (the usual case where the host looks at
      // all) they are written while the read-back travels, instead of ~26 us of idle GPU later; otherwise they are written
      // again at the end.
      untimed(ctx);
      HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned, B.n_active, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      launch_write_results(B, d_results, s);
      CHECK_LAUNCH(ctx, "write_results_kernel");
      HIP_TRY(ctx, hipStreamSynchronize(s));
      if (ctx->h_pinned[0] == 0) {
        ctx->max_counts_clean = true;
        return LOAMX_OK;
      }
    }
  }
  untimed(ctx);
  launch_write_results(B, d_results, s);
  CHECK_LAUNCH(ctx, "write_results_kernel");
  ctx->max_counts_clean = true;
  return LOAMX_OK;
}

int resolve_events(loamx_ctx* ctx) {
  if (ctx->pending.empty()) return LOAMX_OK;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (PendingEvent& pe : ctx->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pe.e0, pe.e1) == hipSuccess) {
      ctx->stats[pe.kernel].launches++;
      ctx->stats[pe.kernel].total_ms += (double)ms;
      ctx->stats[pe.kernel].algorithmic_bytes += pe.bytes;
    }
    if (pe.own_e0) ctx->event_pool.push_back(pe.e0);
    ctx->event_pool.push_back(pe.e1);
  }
  ctx->pending.clear();
  ctx->tail_fresh = false;
  return LOAMX_OK;
}

}  // namespace

/* ================================================================================================== */
extern "C" {

void loamx_default_fe_params(loamx_fe_params* p) { *p = loamx_fe_params{3, 6, 10, 50, 100.0, 1.0, 0.5, 1.0}; }
void loamx_default_reg_params(loamx_reg_params* p) {
  *p = loamx_reg_params{5, 1.0, 3, 10.0, 5, 2.0, 4, 0.1, 10, 1e-3, 1e-2, 100};
}

const char* loamx_status_string(int status) {
  switch (status) {
    case LOAMX_OK: return "ok";
    case LOAMX_ERR_SCAN_SIZE: return "LOAM: provided lidar scan size does not match provided lidar parameters";
    case LOAMX_ERR_BAD_PARAM: return "bad parameter";
    case LOAMX_ERR_HIP: return "HIP error";
    case LOAMX_ERR_CAPACITY: return "output capacity too small";
    case LOAMX_ERR_UNSUPPORTED: return "unsupported parameter combination";
    case LOAMX_ERR_NO_DEVICE: return "no usable HIP device";
    case LOAMX_ERR_COMM: return "RCCL error";
    default: return "unknown status";
  }
}

const char* loamx_last_error(const loamx_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }
const char* loamx_kernel_name(int id) { return (id >= 0 && id < LOAMX_K_COUNT) ? kKernelNames[id] : "?"; }

int loamx_ctx_create(int device, loamx_ctx** out) {
  if (!out) return LOAMX_ERR_BAD_PARAM;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return LOAMX_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return LOAMX_ERR_NO_DEVICE;
  loamx_ctx* ctx = new loamx_ctx;
  ctx->device = device;
  if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pinned), 256, hipHostMallocDefault) != hipSuccess) {
    delete ctx;
    return LOAMX_ERR_HIP;
  }
  ctx->stream = ctx->own_stream;
  // the switches' defaults: the environment, read here and nowhere else
  for (const OptionName& o : kOptionNames)
    if (getenv((std::string("LOAMX_") + o.name).c_str())) (o.extract ? ctx->extract_flags : ctx->reg_flags) |= o.bit;
  if (const char* mc = getenv("LOAMX_MAP_CELLS_LOG2")) ctx->map_cells_log2 = atoi(mc);
  if (const char* sc = getenv("LOAMX_STREAM_CHUNK_PAIRS")) ctx->stream_chunk_pairs = atoi(sc) > 0 ? atoi(sc) : 0;
  // optional: without the auxiliary stream the edge and plane association chains simply run in sequence
  // The auxiliary streams carry the small, latency-bound kernels next to the big ones of the main stream. Round 1 gave
  // them the highest priority (their workgroups dispatched as soon as they are ready: association 2.61 -> 2.57 ms then);
  // with the round-2 kernels it is the other way round (2.12 -> 2.09 ms at the LOWEST priority — HIP's range here is
  // least = 1, greatest = -1, the context stream runs at 0: the plane k-NN kernel is the critical path and loses less to
  // its neighbours). LOAMX_AUX_HIGH_PRIO=1 restores the old arrangement. (Round 3: priority 1 / 0 / -1 for the queue
  // chain's stream: 2.077 / 2.084 / 2.090 ms.)
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  if (!getenv("LOAMX_NO_AUX_STREAM") &&
      hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, getenv("LOAMX_AUX_HIGH_PRIO") ? prio_greatest : prio_least) == hipSuccess) {
    if (hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_mid, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
      (void)hipStreamDestroy(ctx->aux_stream);
      ctx->aux_stream = nullptr;
    } else if (!getenv("LOAMX_NO_AUX2_STREAM") &&
               hipStreamCreateWithPriority(&ctx->aux2_stream, hipStreamNonBlocking, (getenv("LOAMX_AUX_HIGH_PRIO") || getenv("LOAMX_AUX2_HIGH_PRIO")) ? prio_greatest : prio_least) == hipSuccess) {
      if (hipEventCreateWithFlags(&ctx->ev_join2, hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamDestroy(ctx->aux2_stream);
        ctx->aux2_stream = nullptr;
      }
    }
  }
  *out = ctx;
  return LOAMX_OK;
}

void loamx_ctx_destroy(loamx_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (PendingEvent& pe : ctx->pending) {
    if (pe.own_e0) (void)hipEventDestroy(pe.e0);
    (void)hipEventDestroy(pe.e1);
  }
  for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
  for (Buf& b : ctx->ws)
    if (b.p) (void)hipFree(b.p);
  if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
  if (ctx->aux_stream) {
    (void)hipStreamSynchronize(ctx->aux_stream);
    (void)hipStreamDestroy(ctx->aux_stream);
  }
  if (ctx->aux2_stream) {
    (void)hipStreamSynchronize(ctx->aux2_stream);
    (void)hipStreamDestroy(ctx->aux2_stream);
  }
  if (ctx->copy_stream) {
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamDestroy(ctx->copy_stream);
  }
  for (int b = 0; b < 2; b++) {
    if (ctx->ev_up[b]) (void)hipEventDestroy(ctx->ev_up[b]);
    if (ctx->ev_free[b]) (void)hipEventDestroy(ctx->ev_free[b]);
  }
  if (ctx->ev_join2) (void)hipEventDestroy(ctx->ev_join2);
  if (ctx->ev_counts) (void)hipEventDestroy(ctx->ev_counts);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->ev_mid) (void)hipEventDestroy(ctx->ev_mid);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

int loamx_ctx_set_stream(loamx_ctx* ctx, void* hip_stream) {
  if (!ctx) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return LOAMX_OK;
}

int loamx_ctx_synchronize(loamx_ctx* ctx) {
  if (!ctx) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return LOAMX_OK;
}

int loamx_ctx_set_option(loamx_ctx* ctx, const char* name, int value) {
  if (!ctx || !name) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (!strcmp(name, "MAP_CELLS_LOG2")) {
    ctx->map_cells_log2 = value;
    return LOAMX_OK;
  }
  if (!strcmp(name, "STREAM_CHUNK_PAIRS")) {
    if (value < 0) return fail(ctx, LOAMX_ERR_BAD_PARAM, "STREAM_CHUNK_PAIRS: a number of pairs (0 = default)");
    ctx->stream_chunk_pairs = value;
    return LOAMX_OK;
  }
  for (const OptionName& o : kOptionNames)
    if (!strcmp(name, o.name)) {
      uint32_t& w = o.extract ? ctx->extract_flags : ctx->reg_flags;
      w = value ? (w | o.bit) : (w & ~o.bit);
      return LOAMX_OK;
    }
  return fail(ctx, LOAMX_ERR_BAD_PARAM, std::string("unknown option ") + name);
}

int loamx_ctx_get_option(loamx_ctx* ctx, const char* name, int* value) {
  if (!ctx || !name || !value) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (!strcmp(name, "MAP_CELLS_LOG2")) {
    *value = ctx->map_cells_log2;
    return LOAMX_OK;
  }
  if (!strcmp(name, "STREAM_CHUNK_PAIRS")) {
    *value = ctx->stream_chunk_pairs;
    return LOAMX_OK;
  }
  for (const OptionName& o : kOptionNames)
    if (!strcmp(name, o.name)) {
      *value = ((o.extract ? ctx->extract_flags : ctx->reg_flags) & o.bit) ? 1 : 0;
      return LOAMX_OK;
    }
  return fail(ctx, LOAMX_ERR_BAD_PARAM, std::string("unknown option ") + name);
}

int loamx_ctx_extract_counters(loamx_ctx* ctx, uint64_t* tie_replays, uint64_t* scan_fallbacks) {
  if (!ctx) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  unsigned long long ev[2] = {0, 0};
  if (ctx->ws[WS_EXTRACT_EVENTS].p) {
    HIP_TRY(ctx, hipMemcpyAsync(ev, ctx->ws[WS_EXTRACT_EVENTS].p, sizeof(ev), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (tie_replays) *tie_replays = ev[0];
  if (scan_fallbacks) *scan_fallbacks = ev[1];
  return LOAMX_OK;
}

size_t loamx_edge_capacity(const loamx_lidar_params* lidar, const loamx_fe_params* fe) {
  ExtractParams P;
  if (make_extract_params(nullptr, lidar, fe, P) != LOAMX_OK) return 0;
  return edge_capacity(P);
}
size_t loamx_planar_capacity(const loamx_lidar_params* lidar, const loamx_fe_params* fe) {
  ExtractParams P;
  if (make_extract_params(nullptr, lidar, fe, P) != LOAMX_OK) return 0;
  return planar_capacity(P);
}

/* ---- host entry points ---------------------------------------------------------------------------- */
static int host_curv_mask(loamx_ctx* ctx, const void* xyz, bool f32, size_t n_points, const loamx_lidar_params* lidar,
                          const loamx_fe_params* fe, double* curvature_out, uint8_t* mask_out) {
  if (!ctx) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!lidar || !fe) return fail(ctx, LOAMX_ERR_BAD_PARAM, "null parameter struct");
  if (n_points != lidar->scan_lines * lidar->points_per_line) {  // common.h:104-113
    char msg[256];
    snprintf(msg, sizeof(msg), "LOAM: provided lidar scan size ( %zu)  does not match provided lidar parameters (%llu x %llu)",
             n_points, (unsigned long long)lidar->scan_lines, (unsigned long long)lidar->points_per_line);
    return fail(ctx, LOAMX_ERR_SCAN_SIZE, msg);
  }
  if (n_points == 0) return LOAMX_OK;
  ExtractParams P;
  int rc = make_extract_params(ctx, lidar, fe, P);
  if (rc != LOAMX_OK) return rc;
  if (!xyz) return fail(ctx, LOAMX_ERR_BAD_PARAM, "null argument");
  const size_t scalar = f32 ? sizeof(float) : sizeof(double);
  ENSURE(ctx, WS_XYZ, n_points * 3 * scalar);
  HIP_TRY(ctx, hipMemcpyAsync(ctx->ws[WS_XYZ].p, xyz, n_points * 3 * scalar, hipMemcpyHostToDevice, ctx->stream));
  rc = dev_check_finite(ctx, ctx->ws[WS_XYZ].p, f32, nullptr, 1, n_points, 1, true);
  if (rc != LOAMX_OK) return rc;
  rc = extract_dev(ctx, ctx->ws[WS_XYZ].p, f32, 1, P, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, true);
  if (rc != LOAMX_OK) return rc;
  if (curvature_out)
    HIP_TRY(ctx, hipMemcpyAsync(curvature_out, ctx->ws[WS_CURV].p, n_points * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (mask_out)
    HIP_TRY(ctx, hipMemcpyAsync(mask_out, ctx->ws[WS_MASK].p, n_points, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return LOAMX_OK;
}

int loamx_compute_curvature(loamx_ctx* ctx, const double* xyz, size_t n_points, const loamx_lidar_params* lidar,
                            const loamx_fe_params* fe, double* curvature_out) {
  return host_curv_mask(ctx, xyz, false, n_points, lidar, fe, curvature_out, nullptr);
}

int loamx_compute_valid_points(loamx_ctx* ctx, const double* xyz, size_t n_points, const loamx_lidar_params* lidar,
                               const loamx_fe_params* fe, uint8_t* mask_out) {
  return host_curv_mask(ctx, xyz, false, n_points, lidar, fe, nullptr, mask_out);
}

int loamx_compute_curvature_f32(loamx_ctx* ctx, const float* xyz, size_t n_points, const loamx_lidar_params* lidar,
                                const loamx_fe_params* fe, double* curvature_out) {
  return host_curv_mask(ctx, xyz, true, n_points, lidar, fe, curvature_out, nullptr);
}

int loamx_compute_valid_points_f32(loamx_ctx* ctx, const float* xyz, size_t n_points, const loamx_lidar_params* lidar,
                                   const loamx_fe_params* fe, uint8_t* mask_out) {
  return host_curv_mask(ctx, xyz, true, n_points, lidar, fe, nullptr, mask_out);
}

static int host_extract(loamx_ctx* ctx, const void* xyz, bool f32, size_t n_points, const loamx_lidar_params* lidar,
                        const loamx_fe_params* fe, uint32_t* edge_idx, size_t edge_cap, size_t* n_edge,
                        uint32_t* planar_idx, size_t planar_cap, size_t* n_planar) {
  if (!ctx) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!lidar || !fe || !n_edge || !n_planar) return fail(ctx, LOAMX_ERR_BAD_PARAM, "null argument");
  if (n_points != lidar->scan_lines * lidar->points_per_line) {
    char msg[256];
    snprintf(msg, sizeof(msg), "LOAM: provided lidar scan size ( %zu)  does not match provided lidar parameters (%llu x %llu)",
             n_points, (unsigned long long)lidar->scan_lines, (unsigned long long)lidar->points_per_line);
    return fail(ctx, LOAMX_ERR_SCAN_SIZE, msg);
  }
  *n_edge = 0, *n_planar = 0;
  if (n_points == 0) return LOAMX_OK;
  ExtractParams P;
  int rc = make_extract_params(ctx, lidar, fe, P);
  if (rc != LOAMX_OK) return rc;
  const size_t ecap = edge_capacity(P), pcap = planar_capacity(P);
  const size_t scalar = f32 ? sizeof(float) : sizeof(double);
  if (!xyz) return fail(ctx, LOAMX_ERR_BAD_PARAM, "null argument");
  ENSURE(ctx, WS_XYZ, n_points * 3 * scalar);
  ENSURE(ctx, WS_EDGE_IDX, ecap * sizeof(uint32_t));
  ENSURE(ctx, WS_PLANAR_IDX, pcap * sizeof(uint32_t));
  ENSURE(ctx, WS_N_EDGE, sizeof(uint32_t));
  ENSURE(ctx, WS_N_PLANAR, sizeof(uint32_t));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->ws[WS_XYZ].p, xyz, n_points * 3 * scalar, hipMemcpyHostToDevice, ctx->stream));
  rc = dev_check_finite(ctx, ctx->ws[WS_XYZ].p, f32, nullptr, 1, n_points, 1, true);
  if (rc != LOAMX_OK) return rc;
  rc = extract_dev(ctx, ctx->ws[WS_XYZ].p, f32, 1, P, wsp<uint32_t>(ctx, WS_EDGE_IDX), wsp<uint32_t>(ctx, WS_N_EDGE), nullptr,
                   wsp<uint32_t>(ctx, WS_PLANAR_IDX), wsp<uint32_t>(ctx, WS_N_PLANAR), nullptr, false);
  if (rc != LOAMX_OK) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(&ctx->h_pinned[0], ctx->ws[WS_N_EDGE].p, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(&ctx->h_pinned[1], ctx->ws[WS_N_PLANAR].p, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  const size_t ne = ctx->h_pinned[0], npl = ctx->h_pinned[1];
  *n_edge = ne, *n_planar = npl;
  if (ne > edge_cap || npl > planar_cap) return fail(ctx, LOAMX_ERR_CAPACITY, "feature index capacity too small");
  if (ne) HIP_TRY(ctx, hipMemcpy(edge_idx, ctx->ws[WS_EDGE_IDX].p, ne * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (npl) HIP_TRY(ctx, hipMemcpy(planar_idx, ctx->ws[WS_PLANAR_IDX].p, npl * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return LOAMX_OK;
}

int loamx_extract_features(loamx_ctx* ctx, const double* xyz, size_t n_points, const loamx_lidar_params* lidar,
                           const loamx_fe_params* fe, uint32_t* edge_idx, size_t edge_cap, size_t* n_edge,
                           uint32_t* planar_idx, size_t planar_cap, size_t* n_planar) {
  return host_extract(ctx, xyz, false, n_points, lidar, fe, edge_idx, edge_cap, n_edge, planar_idx, planar_cap, n_planar);
}

int loamx_extract_features_f32(loamx_ctx* ctx, const float* xyz, size_t n_points, const loamx_lidar_params* lidar,
                               const loamx_fe_params* fe, uint32_t* edge_idx, size_t edge_cap, size_t* n_edge,
                               uint32_t* planar_idx, size_t planar_cap, size_t* n_planar) {
  return host_extract(ctx, xyz, true, n_points, lidar, fe, edge_idx, edge_cap, n_edge, planar_idx, planar_cap, n_planar);
}

namespace {
struct DetailHook {
  loamx_reg_detail* detail;
  size_t n_se, n_sp;
};
int detail_hook(loamx_ctx* ctx, const RegBatch& B, uint32_t it, void* user) {
  DetailHook* h = static_cast<DetailHook*>(user);
  loamx_reg_detail* d = h->detail;
  if (!d) return LOAMX_OK;
  for (int kind = 0; kind < 2; kind++) {
    const size_t n = kind ? h->n_sp : h->n_se;
    uint32_t* base = kind ? d->plane_pairs : d->edge_pairs;
    const size_t cap = kind ? d->pairs_cap_plane : d->pairs_cap_edge;
    uint32_t* out_n = kind ? d->n_plane_pairs : d->n_edge_pairs;
    if (!base || !out_n) continue;
    out_n[it] = 0;
    if (cap == 0 || n == 0) continue;
    uint32_t* out = base + (size_t)it * 2 * cap;
    std::vector<uint32_t> nearest(n);
    HIP_TRY(ctx, hipMemcpyAsync(nearest.data(), kind ? B.assoc.nearest_plane : B.assoc.nearest_edge, n * sizeof(uint32_t),
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
      if (nearest[i] == 0xFFFFFFFFu) continue;
      if (m < cap) out[2 * m] = (uint32_t)i, out[2 * m + 1] = nearest[i];
      m++;
    }
    out_n[it] = (uint32_t)(m < cap ? m : cap);
  }
  return LOAMX_OK;
}
}  // namespace

static int register_features_impl(loamx_ctx* ctx, const loamx_target_index* index, const double* src_edge, size_t n_se,
                                  const double* src_planar, size_t n_sp, const double* tgt_edge, size_t n_te,
                                  const double* tgt_planar, size_t n_tp, const double init_pose[7],
                                  const loamx_reg_params* reg, loamx_reg_result* result, loamx_reg_detail* detail,
                                  const loamx_assoc_dump* dump = nullptr) {
  if (!ctx) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if ((!result && !dump) || !init_pose) return fail(ctx, LOAMX_ERR_BAD_PARAM, "null argument");
  RegConfig C;
  int rc = make_reg_config(ctx, reg, C);
  if (rc != LOAMX_OK) return rc;
  if (index) {
    if (index->radius[0] != C.r_edge || index->radius[1] != C.r_plane)
      return fail(ctx, LOAMX_ERR_BAD_PARAM, "registration params do not match the ones the target index was built with");
    // (an insert that failed between growing a set and rebuilding its grid leaves n ahead of the grid)
    for (int k = 0; k < 2; k++)
      if (index->n[k] != 0 && !index->grid_valid[k])
        return fail(ctx, LOAMX_ERR_BAD_PARAM, "the target index is inconsistent (an insert into it failed): insert again or build a new one");
    n_te = 0, n_tp = 0;  // the target lives in the index
  }
  const size_t es = n_se > n_te ? n_se : n_te, ps = n_sp > n_tp ? n_sp : n_tp;
  if (es > 0x3FFFFFFFull || ps > 0x3FFFFFFFull) return fail(ctx, LOAMX_ERR_UNSUPPORTED, "feature set too large");  // (queue entries: 30-bit query index + 2 flags)
  if (!host_all_finite(init_pose, false, 7)) return fail(ctx, LOAMX_ERR_BAD_PARAM, kNonFiniteMsg);
  hipStream_t s = ctx->stream;
  const size_t esz = (es ? es : 1) * 3 * sizeof(double), psz = (ps ? ps : 1) * 3 * sizeof(double);
  ENSURE(ctx, WS_SRC_E, esz);
  ENSURE(ctx, WS_TGT_E, esz);
  ENSURE(ctx, WS_SRC_P, psz);
  ENSURE(ctx, WS_TGT_P, psz);
  ENSURE(ctx, WS_FCOUNTS, 4 * sizeof(uint32_t));
  ENSURE(ctx, WS_RESULTS, sizeof(loamx_reg_result));
  ENSURE(ctx, WS_INIT, 7 * sizeof(double));
  if (n_se) HIP_TRY(ctx, hipMemcpyAsync(ctx->ws[WS_SRC_E].p, src_edge, n_se * 24, hipMemcpyHostToDevice, s));
  if (n_te) HIP_TRY(ctx, hipMemcpyAsync(ctx->ws[WS_TGT_E].p, tgt_edge, n_te * 24, hipMemcpyHostToDevice, s));
  if (n_sp) HIP_TRY(ctx, hipMemcpyAsync(ctx->ws[WS_SRC_P].p, src_planar, n_sp * 24, hipMemcpyHostToDevice, s));
  if (n_tp) HIP_TRY(ctx, hipMemcpyAsync(ctx->ws[WS_TGT_P].p, tgt_planar, n_tp * 24, hipMemcpyHostToDevice, s));
  const uint32_t counts[4] = {(uint32_t)n_se, (uint32_t)n_sp, (uint32_t)n_te, (uint32_t)n_tp};
  {  // non-finite coordinates are refused before anything else is launched (loamx.h: "Non-finite input")
    rc = finite_begin(ctx);
    if (rc != LOAMX_OK) return rc;
    finite_add(ctx, ctx->ws[WS_SRC_E].p, false, nullptr, 1, n_se, 1), finite_add(ctx, ctx->ws[WS_TGT_E].p, false, nullptr, 1, n_te, 1);
    finite_add(ctx, ctx->ws[WS_SRC_P].p, false, nullptr, 1, n_sp, 1), finite_add(ctx, ctx->ws[WS_TGT_P].p, false, nullptr, 1, n_tp, 1);
    rc = finite_end(ctx);
    if (rc != LOAMX_OK) return rc;
  }
  HIP_TRY(ctx, hipMemcpyAsync(ctx->ws[WS_FCOUNTS].p, counts, sizeof(counts), hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->ws[WS_INIT].p, init_pose, 7 * sizeof(double), hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));  // `counts` is a stack array
  RegInputs in{};
  in.n_pairs = 1, in.edge_stride = es, in.planar_stride = ps, in.in_pitch = 1;
  in.src_edge = wsp<double>(ctx, WS_SRC_E), in.src_planar = wsp<double>(ctx, WS_SRC_P);
  in.tgt_edge = wsp<double>(ctx, WS_TGT_E), in.tgt_planar = wsp<double>(ctx, WS_TGT_P);
  const uint32_t* fc = wsp<uint32_t>(ctx, WS_FCOUNTS);
  in.n_src_edge = fc, in.n_src_planar = fc + 1, in.n_tgt_edge = fc + 2, in.n_tgt_planar = fc + 3;
  in.init = wsp<double>(ctx, WS_INIT);
  DetailHook hook{detail, n_se, n_sp};
  if (detail) detail->n_iter_info = 0;
  rc = register_dev(ctx, in, C, wsp<loamx_reg_result>(ctx, WS_RESULTS), detail && detail->iter_info,
                    detail ? detail_hook : nullptr, &hook, index, dump, n_se, n_sp);
  if (rc != LOAMX_OK || dump) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(result, ctx->ws[WS_RESULTS].p, sizeof(loamx_reg_result), hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  if (detail && detail->iter_info && result->iterations) {
    HIP_TRY(ctx, hipMemcpy(detail->iter_info, ctx->ws[WS_ITERINFO].p, result->iterations * sizeof(loamx_iter_info),
                           hipMemcpyDeviceToHost));
    detail->n_iter_info = result->iterations;
  }
  return LOAMX_OK;
}

int loamx_register_features(loamx_ctx* ctx, const double* src_edge, size_t n_se, const double* src_planar, size_t n_sp,
                            const double* tgt_edge, size_t n_te, const double* tgt_planar, size_t n_tp,
                            const double init_pose[7], const loamx_reg_params* reg, loamx_reg_result* result,
                            loamx_reg_detail* detail) {
  return register_features_impl(ctx, nullptr, src_edge, n_se, src_planar, n_sp, tgt_edge, n_te, tgt_planar, n_tp, init_pose,
                                reg, result, detail);
}

int loamx_register_features_indexed(loamx_ctx* ctx, const loamx_target_index* index, const double* src_edge, size_t n_se,
                                    const double* src_planar, size_t n_sp, const double init_pose[7],
                                    const loamx_reg_params* reg, loamx_reg_result* result, loamx_reg_detail* detail) {
  if (!index) return LOAMX_ERR_BAD_PARAM;
  return register_features_impl(ctx, index, src_edge, n_se, src_planar, n_sp, nullptr, 0, nullptr, 0, init_pose, reg, result,
                                detail);
}

/* ---- rows a16-a19 one by one ------------------------------------------------------------------------------ */
int loamx_associate(loamx_ctx* ctx, const double* src_edge, size_t n_se, const double* src_planar, size_t n_sp, const double* tgt_edge,
                    size_t n_te, const double* tgt_planar, size_t n_tp, const double pose[7], const loamx_reg_params* reg,
                    loamx_assoc_dump* out) {
  if (!out) return LOAMX_ERR_BAD_PARAM;
  return register_features_impl(ctx, nullptr, src_edge, n_se, src_planar, n_sp, tgt_edge, n_te, tgt_planar, n_tp, pose, reg, nullptr,
                                nullptr, out);
}

static int fit_sets(loamx_ctx* ctx, bool plane, const double* points, size_t n_sets, size_t k, double* prim_out, double* aux_out) {
  if (!ctx) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (n_sets == 0) return LOAMX_OK;
  if (!points || !prim_out) return fail(ctx, LOAMX_ERR_BAD_PARAM, "null argument");
  if (k < (plane ? 3u : 2u)) return fail(ctx, LOAMX_ERR_BAD_PARAM, plane ? "fitPlane needs k >= 3 points" : "fitLine needs k >= 2 points");
  if (k > (size_t)kFitMaxK) return fail(ctx, LOAMX_ERR_UNSUPPORTED, "point sets of more than 32 points are not supported by the fit kernels");
  if (n_sets > 0x7FFFFFFFull / 64) return fail(ctx, LOAMX_ERR_UNSUPPORTED, "too many point sets in one call");
  const size_t pw = plane ? 4 : 6;
  untimed(ctx);
  ENSURE(ctx, WS_FIT_IN, n_sets * k * 24);
  ENSURE(ctx, WS_FIT_OUT, n_sets * (pw + 1) * sizeof(double));
  double* d_prim = wsp<double>(ctx, WS_FIT_OUT);
  double* d_aux = d_prim + n_sets * pw;
  hipStream_t s = ctx->stream;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->ws[WS_FIT_IN].p, points, n_sets * k * 24, hipMemcpyHostToDevice, s));
  {
    int rc = dev_check_finite(ctx, ctx->ws[WS_FIT_IN].p, false, nullptr, 1, n_sets * k, 1, true);
    if (rc != LOAMX_OK) return rc;
  }
  launch_fit_sets(plane, wsp<double>(ctx, WS_FIT_IN), n_sets, (int)k, d_prim, d_aux, s);
  CHECK_LAUNCH(ctx, "fit_sets_kernel");
  HIP_TRY(ctx, hipMemcpyAsync(prim_out, d_prim, n_sets * pw * sizeof(double), hipMemcpyDeviceToHost, s));
  if (aux_out) HIP_TRY(ctx, hipMemcpyAsync(aux_out, d_aux, n_sets * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return LOAMX_OK;
}

int loamx_fit_lines(loamx_ctx* ctx, const double* points, size_t n_sets, size_t k, double* lines_out, double* cond_out) {
  return fit_sets(ctx, false, points, n_sets, k, lines_out, cond_out);
}
int loamx_fit_planes(loamx_ctx* ctx, const double* points, size_t n_sets, size_t k, double* planes_out, double* avg_dist_out) {
  return fit_sets(ctx, true, points, n_sets, k, planes_out, avg_dist_out);
}

int loamx_knn_search(loamx_ctx* ctx, const loamx_target_index* index, int which_set, const double* queries, size_t n_queries, size_t k,
                     double max_dist, uint32_t* indices_out, uint32_t* counts_out) {
  if (!ctx || !index) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (which_set != 0 && which_set != 1) return fail(ctx, LOAMX_ERR_BAD_PARAM, "which_set: 0 = edge points, 1 = planar points");
  if (index->n[which_set] != 0 && !index->grid_valid[which_set])
    return fail(ctx, LOAMX_ERR_BAD_PARAM, "the target index is inconsistent (an insert into it failed): insert again or build a new one");
  if (n_queries == 0) return LOAMX_OK;
  if (!queries || !indices_out || !counts_out) return fail(ctx, LOAMX_ERR_BAD_PARAM, "null argument");
  if (k == 0) {
    memset(counts_out, 0, n_queries * sizeof(uint32_t));
    return LOAMX_OK;
  }
  if (k > (size_t)kMaxK) return fail(ctx, LOAMX_ERR_UNSUPPORTED, "k > 16 neighbours not supported by the search kernels");
  if (n_queries > 0x7FFFFFFFull / 64) return fail(ctx, LOAMX_ERR_UNSUPPORTED, "too many queries in one call");
  untimed(ctx);
  ENSURE(ctx, WS_FIT_IN, n_queries * 24);
  ENSURE(ctx, WS_FIT_OUT, n_queries * (k + 1) * sizeof(uint32_t));
  uint32_t* d_idx = wsp<uint32_t>(ctx, WS_FIT_OUT);
  uint32_t* d_cnt = d_idx + n_queries * k;
  hipStream_t s = ctx->stream;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->ws[WS_FIT_IN].p, queries, n_queries * 24, hipMemcpyHostToDevice, s));
  {
    int rc = dev_check_finite(ctx, ctx->ws[WS_FIT_IN].p, false, nullptr, 1, n_queries, 1, true);
    if (rc != LOAMX_OK) return rc;
  }
  const int w = which_set;
  const GridSet gs{index->desc[w], index->cells[w], index->sorted[w], index->cap[w] + kGridPad, index->rel[w], index->cells_cap[w]};
  launch_knn_queries(gs, wsp<double>(ctx, WS_FIT_IN), n_queries, (int)k, max_dist, d_idx, d_cnt, s);
  CHECK_LAUNCH(ctx, "knn_queries_kernel");
  HIP_TRY(ctx, hipMemcpyAsync(indices_out, d_idx, n_queries * k * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(counts_out, d_cnt, n_queries * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return LOAMX_OK;
}

/* ---- persistent target index ---------------------------------------------------------------------------- */
namespace {
void index_free(loamx_target_index* idx) {
  for (int k = 0; k < 2; k++) {
    if (idx->desc[k]) (void)hipFree(idx->desc[k]);
    if (idx->cells[k]) (void)hipFree(idx->cells[k]);
    if (idx->sorted[k]) (void)hipFree(idx->sorted[k]);
    if (idx->rel[k]) (void)hipFree(idx->rel[k]);
    if (idx->pts[k]) (void)hipFree(idx->pts[k]);
    if (idx->sorted2[k]) (void)hipFree(idx->sorted2[k]);
    if (idx->rel2[k]) (void)hipFree(idx->rel2[k]);
  }
  if (idx->counts) (void)hipFree(idx->counts);
  if (idx->scratch) (void)hipFree(idx->scratch);
  delete idx;
}

// Makes room for `extra` more points of kind k (amortised doubling; the old points are copied over on the device).
int index_reserve(loamx_ctx* ctx, loamx_target_index* idx, int k, size_t extra) {
  const size_t need = idx->n[k] + extra;
  if (need <= idx->cap[k] && idx->pts[k]) return LOAMX_OK;
  size_t cap = idx->cap[k] ? idx->cap[k] : 1;
  while (cap < need) cap *= 2;
  if (idx->cap[k] == 0) cap = need ? need : 1;  // the first build is sized exactly
  double* pts = nullptr;
  GridPoint* sorted = nullptr;
  float* rel = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&pts), cap * 24) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&sorted), (cap + kGridPad) * sizeof(GridPoint)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&rel), 3 * (cap + kGridPad) * sizeof(float)) != hipSuccess) {
    if (pts) (void)hipFree(pts);
    if (sorted) (void)hipFree(sorted);
    if (rel) (void)hipFree(rel);
    return fail(ctx, LOAMX_ERR_HIP, "hipMalloc failed for the target index");
  }
  if (idx->pts[k] && idx->n[k])
    HIP_TRY(ctx, hipMemcpyAsync(pts, idx->pts[k], idx->n[k] * 24, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (idx->pts[k]) (void)hipFree(idx->pts[k]);
  if (idx->sorted[k]) (void)hipFree(idx->sorted[k]);
  if (idx->rel[k]) (void)hipFree(idx->rel[k]);
  if (idx->sorted2[k]) (void)hipFree(idx->sorted2[k]);  // (the twins are made again, at the new capacity, by the next merge)
  if (idx->rel2[k]) (void)hipFree(idx->rel2[k]);
  idx->sorted2[k] = nullptr, idx->rel2[k] = nullptr;
  idx->pts[k] = pts, idx->sorted[k] = sorted, idx->rel[k] = rel, idx->cap[k] = cap;
  idx->n_at_build[k] = 0, idx->grid_valid[k] = false;  // (the cell-sorted arrays are gone: this kind is rebuilt)
  return LOAMX_OK;
}

// (Re)builds the grids of the kinds in `kinds` (bit 0 edge, bit 1 planar) from idx->pts: cell edge, dimensions and order
// chosen afresh for the set as it is now.
int index_build(loamx_ctx* ctx, loamx_target_index* idx, unsigned kinds = 3u) {
  hipStream_t s = ctx->stream;
  const uint32_t counts[2] = {(uint32_t)idx->n[0], (uint32_t)idx->n[1]};
  HIP_TRY(ctx, hipMemcpyAsync(idx->counts, counts, sizeof(counts), hipMemcpyHostToDevice, s));
  RegConfig C{};
  C.r_edge = idx->radius[0], C.r_plane = idx->radius[1];
  C.flags = ctx->reg_flags;
  RegBatch B{};
  B.n_pairs = 1, B.in_pitch = 1;
  B.edge_stride = idx->cap[0], B.planar_stride = idx->cap[1];
  B.tgt_edge = idx->pts[0], B.tgt_planar = idx->pts[1];
  B.n_tgt_edge = idx->counts, B.n_tgt_planar = idx->counts + 1;
  // map-sized sets get a cell table of up to kGridMapCellsCap entries (the scan-sized table of 65 536 would put
  // hundreds of points of a million-point map into every cell); tables and cursors grow with the set
  size_t scratch_need = kGridBigScratchBytes;
  for (int k = 0; k < 2; k++) {
    if (!(kinds & (1u << k))) continue;
    const int mc = ctx->map_cells_log2;  // experiment knob (16 = the scan-sized table)
    // default: about one table entry per two points, between 2^18 and 2^22 (measured on a 1.02 M-point map, registration of
    // a 128 x 2048 scan: 2^17 6.22, 2^18 6.10, 2^19 5.23, 2^20 6.04, 2^21 6.88 ms — finer cells shorten the candidate
    // streams of the dense cells near the sensor, coarser ones keep k points inside the 3x3x3 block)
    uint32_t dflt = kGridMapCellsCap;
    while (dflt < (1u << 22) && (size_t)dflt * 2 < idx->n[k]) dflt <<= 1;
    idx->cells_cap[k] = idx->n[k] > 200000 ? (mc >= 8 && mc <= 24 ? (1u << mc) : dflt) : 0u;
    if (idx->cells_cap[k] <= kGridCellsCap) idx->cells_cap[k] = 0u;
    const size_t cap = idx->cells_cap[k] ? idx->cells_cap[k] : kGridCellsCap;
    const size_t need = (cap + 1 + 4) * sizeof(uint32_t);  // (+4: the search reads four entries at a time)
    if (idx->cells_alloc[k] < need) {
      HIP_TRY(ctx, hipStreamSynchronize(s));
      if (idx->cells[k]) (void)hipFree(idx->cells[k]);
      idx->cells[k] = nullptr, idx->cells_alloc[k] = 0;
      HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&idx->cells[k]), need));
      idx->cells_alloc[k] = need;
    }
    if (64 + (cap + cap / 4096 + 8) * sizeof(uint32_t) > scratch_need) scratch_need = 64 + (cap + cap / 4096 + 8) * sizeof(uint32_t);  // (box keys, cursors, tile sums)
  }
  if (idx->scratch_alloc < scratch_need) {
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (idx->scratch) (void)hipFree(idx->scratch);
    idx->scratch = nullptr, idx->scratch_alloc = 0;
    HIP_TRY(ctx, hipMalloc(&idx->scratch, scratch_need));
    idx->scratch_alloc = scratch_need;
  }
  B.sort_scratch = static_cast<GridPoint*>(idx->scratch);
  B.grid_edge = GridSet{idx->desc[0], idx->cells[0], idx->sorted[0], idx->cap[0] + kGridPad, idx->rel[0], idx->cells_cap[0]};
  B.grid_plane = GridSet{idx->desc[1], idx->cells[1], idx->sorted[1], idx->cap[1] + kGridPad, idx->rel[1], idx->cells_cap[1]};
  untimed(ctx);
  {
    TimedScope t(ctx, LOAMX_K_GRID, 0.0);
    for (int k = 0; k < 2; k++)
      if (kinds & (1u << k)) launch_grid_build_target(B, C, k == 1, s);
  }
  int rc = check_launch(ctx, "grid_build_kernel");
  if (rc != LOAMX_OK) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(s));
  for (int k = 0; k < 2; k++)
    if (kinds & (1u << k)) idx->n_at_build[k] = idx->n[k], idx->grid_valid[k] = true, idx->full_builds++;
  return LOAMX_OK;
}

// May kind k take `add` more points by a merge into its existing grid (index_insert_* kernels)? Map-sized sets only (a
// scan-sized set is rebuilt by one workgroup in ~0.1 ms), while the set has not doubled since its grid was chosen (the
// cell edge follows the density) and stays on the same side of the map-table threshold.
bool index_can_merge(const loamx_target_index* idx, int k, size_t add) {
  const size_t n_old = idx->n[k], n_new = n_old + add;
  return idx->n_at_build[k] > kGridSmallCap && n_old >= idx->n_at_build[k] && n_new <= 2 * idx->n_at_build[k] && n_new <= idx->cap[k] &&
         (n_new > 200000) == (idx->cells_cap[k] != 0u);
}

int index_append(loamx_ctx* ctx, loamx_target_index* idx, const double* edge, size_t n_e, const double* planar, size_t n_p) {
  const double* host[2] = {edge, planar};
  const size_t add[2] = {n_e, n_p};
  for (int k = 0; k < 2; k++) {
    if (idx->n[k] + add[k] > 0x0FFFFFFFull) return fail(ctx, LOAMX_ERR_UNSUPPORTED, "feature set too large");
    if (add[k] && !host[k]) return fail(ctx, LOAMX_ERR_BAD_PARAM, "null point array");
  }
  hipStream_t s = ctx->stream;
  // The new points go to a staging buffer first and are looked at there (loamx.h: "Non-finite input"): a refused insert must
  // leave the index as it was, and making room (index_reserve) already re-allocates its arrays.
  ENSURE(ctx, WS_FIT_IN, (add[0] + add[1] ? add[0] + add[1] : 1) * 24);
  double* stage[2] = {wsp<double>(ctx, WS_FIT_IN), wsp<double>(ctx, WS_FIT_IN) + add[0] * 3};
  {
    for (int k = 0; k < 2; k++)
      if (add[k]) HIP_TRY(ctx, hipMemcpyAsync(stage[k], host[k], add[k] * 24, hipMemcpyHostToDevice, s));
    int rc = finite_begin(ctx);
    if (rc != LOAMX_OK) return rc;
    for (int k = 0; k < 2; k++) finite_add(ctx, stage[k], false, nullptr, 1, add[k], 1);
    rc = finite_end(ctx);
    if (rc != LOAMX_OK) return rc;
  }
  for (int k = 0; k < 2; k++) {  // all the room first: a failed allocation leaves the index as it was
    int rc = index_reserve(ctx, idx, k, add[k]);
    if (rc != LOAMX_OK) return rc;
  }
  const size_t n_old[2] = {idx->n[0], idx->n[1]};
  unsigned rebuild = 0u;
  bool merge[2] = {false, false};
  for (int k = 0; k < 2; k++)  // the new points behind the sets
    if (add[k]) HIP_TRY(ctx, hipMemcpyAsync(idx->pts[k] + idx->n[k] * 3, stage[k], add[k] * 24, hipMemcpyDeviceToDevice, s));
  for (int k = 0; k < 2; k++) {
    merge[k] = add[k] != 0 && index_can_merge(idx, k, add[k]);
    // a kind that receives nothing keeps its grid — also one grown by merges since its last full build (the size
    // at the last full build was compared here, and an edge-only insert re-sorted a million-point planar map)
    if (!merge[k] && (add[k] != 0 || !idx->grid_valid[k])) rebuild |= 1u << k;
    // (from here on the grid no longer describes the set — n has grown — until the build or the merge below has
    // succeeded; a failure in between must leave the flag down, so that the next insert rebuilds)
    if (add[k] != 0) idx->grid_valid[k] = false;
    idx->n[k] += add[k];
  }
  // ---- merges: count the new points per cell (and learn whether they all lie inside the grid), then move + scatter
  uint32_t* ws[2] = {nullptr, nullptr};
  for (int k = 0; k < 2; k++) {
    if (!merge[k]) continue;
    const size_t cells = idx->cells_cap[k] ? idx->cells_cap[k] : kGridCellsCap;
    const size_t need = index_insert_ws_bytes(cells, add[k]);
    // (the two kinds share idx->scratch: the planar kind's part sits behind the edge kind's)
    const size_t off = k == 1 && merge[0] ? ((index_insert_ws_bytes(idx->cells_cap[0] ? idx->cells_cap[0] : kGridCellsCap, add[0]) + 255) & ~(size_t)255) : 0;
    if (idx->scratch_alloc < off + need) {
      HIP_TRY(ctx, hipStreamSynchronize(s));
      void* bigger = nullptr;
      HIP_TRY(ctx, hipMalloc(&bigger, 2 * (off + need)));
      if (idx->scratch) (void)hipFree(idx->scratch);
      idx->scratch = bigger, idx->scratch_alloc = 2 * (off + need);
      if (k == 1 && merge[0]) {  // (the edge kind's counts were in the old block: once more)
        const GridSet gs0{idx->desc[0], idx->cells[0], idx->sorted[0], idx->cap[0] + kGridPad, idx->rel[0], idx->cells_cap[0]};
        ws[0] = static_cast<uint32_t*>(idx->scratch);
        launch_index_insert_count(gs0, idx->cells_cap[0] ? idx->cells_cap[0] : kGridCellsCap, idx->pts[0] + n_old[0] * 3, (uint32_t)add[0], ws[0], s);
      }
    }
    ws[k] = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(idx->scratch) + off);
    const GridSet gs{idx->desc[k], idx->cells[k], idx->sorted[k], idx->cap[k] + kGridPad, idx->rel[k], idx->cells_cap[k]};
    untimed(ctx);
    launch_index_insert_count(gs, cells, idx->pts[k] + n_old[k] * 3, (uint32_t)add[k], ws[k], s);
  }
  for (int k = 0; k < 2; k++) {
    if (!merge[k]) continue;
    HIP_TRY(ctx, hipMemcpyAsync(&ctx->h_pinned[32 + k], ws[k], sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  }
  if (merge[0] || merge[1]) HIP_TRY(ctx, hipStreamSynchronize(s));
  for (int k = 0; k < 2; k++) {
    if (!merge[k]) continue;
    if (ctx->h_pinned[32 + k] != 0u) {  // a new point outside the grid: this kind is rebuilt around the larger set
      merge[k] = false, rebuild |= 1u << k, idx->grid_valid[k] = false;
      continue;
    }
    if (!idx->sorted2[k]) {
      if (hipMalloc(reinterpret_cast<void**>(&idx->sorted2[k]), (idx->cap[k] + kGridPad) * sizeof(GridPoint)) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&idx->rel2[k]), 3 * (idx->cap[k] + kGridPad) * sizeof(float)) != hipSuccess) {
        if (idx->sorted2[k]) (void)hipFree(idx->sorted2[k]);
        idx->sorted2[k] = nullptr, idx->rel2[k] = nullptr;
        merge[k] = false, rebuild |= 1u << k, idx->grid_valid[k] = false;  // (no room for the twins: the rebuild needs none)
        continue;
      }
    }
    const size_t cells = idx->cells_cap[k] ? idx->cells_cap[k] : kGridCellsCap;
    const GridSet gs{idx->desc[k], idx->cells[k], idx->sorted[k], idx->cap[k] + kGridPad, idx->rel[k], idx->cells_cap[k]};
    launch_index_insert_merge(gs, cells, (uint32_t)n_old[k], idx->pts[k] + n_old[k] * 3, (uint32_t)add[k], ws[k], idx->sorted2[k], idx->rel2[k], s);
    int rc = check_launch(ctx, "index_insert kernels");
    if (rc != LOAMX_OK) return rc;
    std::swap(idx->sorted[k], idx->sorted2[k]);
    std::swap(idx->rel[k], idx->rel2[k]);
    idx->grid_valid[k] = true;
    idx->merges++;
  }
  if (rebuild) return index_build(ctx, idx, rebuild);
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return LOAMX_OK;
}
}  // namespace

void loamx_target_index_destroy(loamx_ctx* ctx, loamx_target_index* index) {
  if (!index) return;
  if (ctx) {
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
  }
  index_free(index);
}

int loamx_target_index_create(loamx_ctx* ctx, const double* tgt_edge, size_t n_te, const double* tgt_planar, size_t n_tp,
                              const loamx_reg_params* reg, loamx_target_index** out) {
  if (!ctx || !out) return LOAMX_ERR_BAD_PARAM;
  *out = nullptr;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  RegConfig C;
  int rc = make_reg_config(ctx, reg, C);
  if (rc != LOAMX_OK) return rc;
  loamx_target_index* idx = new loamx_target_index;
  idx->radius[0] = C.r_edge, idx->radius[1] = C.r_plane;
  bool ok = hipMalloc(reinterpret_cast<void**>(&idx->counts), 2 * sizeof(uint32_t)) == hipSuccess;
  for (int k = 0; k < 2 && ok; k++) ok = hipMalloc(reinterpret_cast<void**>(&idx->desc[k]), sizeof(GridDesc)) == hipSuccess;  // (tables: index_build)
  rc = ok ? index_append(ctx, idx, tgt_edge, n_te, tgt_planar, n_tp) : fail(ctx, LOAMX_ERR_HIP, "hipMalloc failed for the target index");
  if (rc != LOAMX_OK) {
    index_free(idx);
    return rc;
  }
  *out = idx;
  return LOAMX_OK;
}

int loamx_target_index_insert(loamx_ctx* ctx, loamx_target_index* index, const double* edge, size_t n_edge, const double* planar,
                              size_t n_planar) {
  if (!ctx || !index) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (n_edge == 0 && n_planar == 0) return LOAMX_OK;
  return index_append(ctx, index, edge, n_edge, planar, n_planar);
}

int loamx_target_index_stats(const loamx_target_index* index, uint64_t* full_builds, uint64_t* merges) {
  if (!index) return LOAMX_ERR_BAD_PARAM;
  if (full_builds) *full_builds = index->full_builds;
  if (merges) *merges = index->merges;
  return LOAMX_OK;
}

int loamx_target_index_size(const loamx_target_index* index, size_t* n_edge, size_t* n_planar) {
  if (!index) return LOAMX_ERR_BAD_PARAM;
  if (n_edge) *n_edge = index->n[0];
  if (n_planar) *n_planar = index->n[1];
  return LOAMX_OK;
}

/* ---- device-resident batch entry points ----------------------------------------------------------- */
static int extract_batch_dev(loamx_ctx* ctx, const void* d_xyz, bool f32, size_t n_scans, const loamx_lidar_params* lidar,
                             const loamx_fe_params* fe, uint32_t* d_edge_idx, uint32_t* d_n_edge, double* d_edge_xyz,
                             uint32_t* d_planar_idx, uint32_t* d_n_planar, double* d_planar_xyz) {
  if (!ctx) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  ExtractParams P;
  int rc = make_extract_params(ctx, lidar, fe, P);
  if (rc != LOAMX_OK) return rc;
  rc = dev_check_finite(ctx, d_xyz, f32, nullptr, n_scans, (size_t)P.H * P.W, 1);
  if (rc != LOAMX_OK) return rc;
  return extract_dev(ctx, d_xyz, f32, n_scans, P, d_edge_idx, d_n_edge, d_edge_xyz, d_planar_idx, d_n_planar, d_planar_xyz, false);
}

int loamx_extract_features_batch_dev(loamx_ctx* ctx, const double* d_xyz, size_t n_scans, const loamx_lidar_params* lidar,
                                     const loamx_fe_params* fe, uint32_t* d_edge_idx, uint32_t* d_n_edge,
                                     double* d_edge_xyz, uint32_t* d_planar_idx, uint32_t* d_n_planar,
                                     double* d_planar_xyz) {
  return extract_batch_dev(ctx, d_xyz, false, n_scans, lidar, fe, d_edge_idx, d_n_edge, d_edge_xyz, d_planar_idx, d_n_planar, d_planar_xyz);
}

int loamx_extract_features_batch_dev_f32(loamx_ctx* ctx, const float* d_xyz, size_t n_scans, const loamx_lidar_params* lidar,
                                         const loamx_fe_params* fe, uint32_t* d_edge_idx, uint32_t* d_n_edge,
                                         double* d_edge_xyz, uint32_t* d_planar_idx, uint32_t* d_n_planar,
                                         double* d_planar_xyz) {
  return extract_batch_dev(ctx, d_xyz, true, n_scans, lidar, fe, d_edge_idx, d_n_edge, d_edge_xyz, d_planar_idx, d_n_planar, d_planar_xyz);
}

int loamx_register_features_batch_dev(loamx_ctx* ctx, size_t n_pairs, const double* d_src_edge,
                                      const uint32_t* d_n_src_edge, const double* d_src_planar,
                                      const uint32_t* d_n_src_planar, const double* d_tgt_edge,
                                      const uint32_t* d_n_tgt_edge, const double* d_tgt_planar,
                                      const uint32_t* d_n_tgt_planar, size_t edge_stride, size_t planar_stride,
                                      const double* d_init, const loamx_reg_params* reg, loamx_reg_result* d_results) {
  if (!ctx) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  RegConfig C;
  int rc = make_reg_config(ctx, reg, C);
  if (rc != LOAMX_OK) return rc;
  RegInputs in{n_pairs, edge_stride, planar_stride, 1, d_src_edge, d_src_planar, d_tgt_edge, d_tgt_planar,
               d_n_src_edge, d_n_src_planar, d_n_tgt_edge, d_n_tgt_planar, d_init};
  if (ctx->reg_flags & kRegFlagCheckFinite) {
    const struct { const double* p; const uint32_t* n; size_t stride; } sets[4] = {{d_src_edge, d_n_src_edge, edge_stride}, {d_src_planar, d_n_src_planar, planar_stride},
                                                                                   {d_tgt_edge, d_n_tgt_edge, edge_stride}, {d_tgt_planar, d_n_tgt_planar, planar_stride}};
    for (const auto& st : sets) {
      rc = dev_check_finite(ctx, st.p, false, st.n, n_pairs, st.stride, 1);
      if (rc != LOAMX_OK) return rc;
    }
    if (d_init && n_pairs) {  // (7 doubles per pair, counted as scalars: the buffer need not end on a whole point)
      rc = finite_begin(ctx);
      if (rc != LOAMX_OK) return rc;
      launch_check_finite_scalars(d_init, n_pairs * 7, static_cast<uint32_t*>(ctx->ws[WS_FINITE_FLAG].p), ctx->stream);
      rc = finite_end(ctx);
      if (rc != LOAMX_OK) return rc;
    }
  }
  return register_dev(ctx, in, C, d_results, false, nullptr, nullptr);
}

// (the caller holds ctx->mu and has selected the device; `look` = refuse non-finite input whatever CHECK_FINITE says)
static int register_scan_pairs_locked(loamx_ctx* ctx, const void* d_xyz, bool f32, size_t n_pairs, const loamx_lidar_params* lidar,
                                      const loamx_fe_params* fe, const loamx_reg_params* reg, loamx_reg_result* d_results, bool look = false) {
  ExtractParams P;
  int rc = make_extract_params(ctx, lidar, fe, P);
  if (rc != LOAMX_OK) return rc;
  RegConfig C;
  rc = make_reg_config(ctx, reg, C);
  if (rc != LOAMX_OK) return rc;
  if (n_pairs == 0) return LOAMX_OK;
  const size_t n_scans = 2 * n_pairs, ecap = edge_capacity(P), pcap = planar_capacity(P);
  rc = dev_check_finite(ctx, d_xyz, f32, nullptr, n_scans, (size_t)P.H * P.W, 1, look);
  if (rc != LOAMX_OK) return rc;
  ENSURE(ctx, WS_N_EDGE, n_scans * sizeof(uint32_t));
  ENSURE(ctx, WS_N_PLANAR, n_scans * sizeof(uint32_t));
  ENSURE(ctx, WS_EDGE_XYZ, n_scans * ecap * 3 * sizeof(double));
  ENSURE(ctx, WS_PLANAR_XYZ, n_scans * pcap * 3 * sizeof(double));
  ExtractBoxes boxes;
  // (no index arrays: the registration reads the features' points, and 4 bytes per feature are 0.15 GB per 1 024-pair step)
  rc = extract_dev(ctx, d_xyz, f32, n_scans, P, nullptr, wsp<uint32_t>(ctx, WS_N_EDGE), wsp<double>(ctx, WS_EDGE_XYZ), nullptr,
                   wsp<uint32_t>(ctx, WS_N_PLANAR), wsp<double>(ctx, WS_PLANAR_XYZ), false, &boxes);
  if (rc != LOAMX_OK) return rc;
  // scan 2p = target, scan 2p+1 = source (interleaved => in_pitch 2)
  RegInputs in{};
  in.boxes = boxes;
  in.n_pairs = n_pairs, in.edge_stride = ecap, in.planar_stride = pcap, in.in_pitch = 2;
  in.tgt_edge = wsp<double>(ctx, WS_EDGE_XYZ), in.src_edge = in.tgt_edge + ecap * 3;
  in.tgt_planar = wsp<double>(ctx, WS_PLANAR_XYZ), in.src_planar = in.tgt_planar + pcap * 3;
  in.n_tgt_edge = wsp<uint32_t>(ctx, WS_N_EDGE), in.n_src_edge = in.n_tgt_edge + 1;
  in.n_tgt_planar = wsp<uint32_t>(ctx, WS_N_PLANAR), in.n_src_planar = in.n_tgt_planar + 1;
  in.init = nullptr;
  return register_dev(ctx, in, C, d_results, false, nullptr, nullptr);
}
static int register_scan_pairs(loamx_ctx* ctx, const void* d_xyz, bool f32, size_t n_pairs, const loamx_lidar_params* lidar,
                               const loamx_fe_params* fe, const loamx_reg_params* reg, loamx_reg_result* d_results) {
  if (!ctx) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return register_scan_pairs_locked(ctx, d_xyz, f32, n_pairs, lidar, fe, reg, d_results);
}

// Host memory in, host memory out (loamx.h: loamx_register_scan_pairs): chunk k + 1 is uploaded on the copy stream into the
// other staging buffer while chunk k goes through register_scan_pairs_locked — the host blocks inside that call (its two
// read-backs), so the next upload is enqueued BEFORE it; a buffer is refilled once the chunk that read it has finished.
constexpr size_t kStreamChunkPairs = 128;
static int register_scan_pairs_host(loamx_ctx* ctx, const void* xyz, bool f32, size_t n_pairs, const loamx_lidar_params* lidar,
                                    const loamx_fe_params* fe, const loamx_reg_params* reg, loamx_reg_result* results) {
  if (!ctx) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  {  // (parameter errors before anything moves)
    ExtractParams P;
    int rc = make_extract_params(ctx, lidar, fe, P);
    if (rc != LOAMX_OK) return rc;
    RegConfig C;
    rc = make_reg_config(ctx, reg, C);
    if (rc != LOAMX_OK) return rc;
  }
  if (n_pairs == 0) return LOAMX_OK;
  if (!xyz || !results) return fail(ctx, LOAMX_ERR_BAD_PARAM, "null argument");
  const size_t pair_bytes = 2 * (size_t)lidar->scan_lines * lidar->points_per_line * 3 * (f32 ? sizeof(float) : sizeof(double));
  size_t chunk = ctx->stream_chunk_pairs > 0 ? (size_t)ctx->stream_chunk_pairs : kStreamChunkPairs;
  chunk = chunk < n_pairs ? chunk : n_pairs;
  const size_t n_chunks = (n_pairs + chunk - 1) / chunk;
  untimed(ctx);
  ENSURE(ctx, WS_STREAM_IN0, chunk * pair_bytes);
  if (n_chunks > 1) ENSURE(ctx, WS_STREAM_IN1, chunk * pair_bytes);
  ENSURE(ctx, WS_STREAM_RES, n_pairs * sizeof(loamx_reg_result));
  if (!ctx->copy_stream) {
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; b++) {
      HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_up[b], hipEventDisableTiming));
      HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_free[b], hipEventDisableTiming));
    }
  }
  unsigned char* in[2] = {wsp<unsigned char>(ctx, WS_STREAM_IN0), n_chunks > 1 ? wsp<unsigned char>(ctx, WS_STREAM_IN1) : nullptr};
  loamx_reg_result* d_res = wsp<loamx_reg_result>(ctx, WS_STREAM_RES);
  const unsigned char* host = static_cast<const unsigned char*>(xyz);
  auto pairs_of = [&](size_t k) { return k + 1 < n_chunks ? chunk : n_pairs - k * chunk; };
  auto upload = [&](size_t k) -> hipError_t {
    const int b = (int)(k & 1);
    hipError_t e = hipSuccess;
    if (k >= 2) e = hipStreamWaitEvent(ctx->copy_stream, ctx->ev_free[b], 0);  // (the chunk that read this buffer is done)
    if (e == hipSuccess) e = hipMemcpyAsync(in[b], host + k * chunk * pair_bytes, pairs_of(k) * pair_bytes, hipMemcpyHostToDevice, ctx->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(ctx->ev_up[b], ctx->copy_stream);
    return e;
  };
  // (whatever earlier calls left on the context's stream may still read the staging buffers' neighbours: nothing to wait for,
  // the buffers are this entry point's own — but a previous call of THIS entry point has synchronised before it returned)
  // (whatever fails below: uploads in flight must not outlive the caller's buffer, nor the staging buffers a later call may grow)
  auto drain = [&](int code) {
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamSynchronize(ctx->stream);
    return code;
  };
#define STREAM_TRY(expr)                                                                                  \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) return drain(fail(ctx, LOAMX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_))); \
  } while (0)
  STREAM_TRY(upload(0));
  int rc = LOAMX_OK;
  for (size_t k = 0; k < n_chunks && rc == LOAMX_OK; k++) {
    const int b = (int)(k & 1);
    if (k + 1 < n_chunks) STREAM_TRY(upload(k + 1));
    STREAM_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_up[b], 0));
    rc = register_scan_pairs_locked(ctx, in[b], f32, pairs_of(k), lidar, fe, reg, d_res + k * chunk, true);
    untimed(ctx);
    if (rc == LOAMX_OK) STREAM_TRY(hipEventRecord(ctx->ev_free[b], ctx->stream));
  }
  if (rc != LOAMX_OK) return drain(rc);
  STREAM_TRY(hipMemcpyAsync(results, d_res, n_pairs * sizeof(loamx_reg_result), hipMemcpyDeviceToHost, ctx->stream));
  STREAM_TRY(hipStreamSynchronize(ctx->stream));
  return LOAMX_OK;
#undef STREAM_TRY
}

int loamx_register_scan_pairs(loamx_ctx* ctx, const double* xyz, size_t n_pairs, const loamx_lidar_params* lidar,
                              const loamx_fe_params* fe, const loamx_reg_params* reg, loamx_reg_result* results) {
  return register_scan_pairs_host(ctx, xyz, false, n_pairs, lidar, fe, reg, results);
}
int loamx_register_scan_pairs_f32(loamx_ctx* ctx, const float* xyz, size_t n_pairs, const loamx_lidar_params* lidar,
                                  const loamx_fe_params* fe, const loamx_reg_params* reg, loamx_reg_result* results) {
  return register_scan_pairs_host(ctx, xyz, true, n_pairs, lidar, fe, reg, results);
}

int loamx_register_scan_pairs_dev(loamx_ctx* ctx, const double* d_xyz, size_t n_pairs, const loamx_lidar_params* lidar,
                                  const loamx_fe_params* fe, const loamx_reg_params* reg, loamx_reg_result* d_results) {
  return register_scan_pairs(ctx, d_xyz, false, n_pairs, lidar, fe, reg, d_results);
}

int loamx_register_scan_pairs_dev_f32(loamx_ctx* ctx, const float* d_xyz, size_t n_pairs, const loamx_lidar_params* lidar,
                                      const loamx_fe_params* fe, const loamx_reg_params* reg, loamx_reg_result* d_results) {
  return register_scan_pairs(ctx, d_xyz, true, n_pairs, lidar, fe, reg, d_results);
}

/* ---- kernel timing ------------------------------------------------------------------------------------ */
int loamx_ctx_enable_kernel_timing(loamx_ctx* ctx, int enable) {
  if (!ctx) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  ctx->timing = enable != 0;
  return LOAMX_OK;
}

static int read_sweep_slots(loamx_ctx* ctx, unsigned long long out[6]) {
  out[0] = out[1] = out[2] = out[3] = out[4] = out[5] = 0;
  if (!ctx->ws[WS_COUNTERS].p) return LOAMX_OK;
  HIP_TRY(ctx, hipMemcpy(out, wsp<unsigned char>(ctx, WS_COUNTERS) + 8, 48, hipMemcpyDeviceToHost));
  return LOAMX_OK;
}

int loamx_ctx_reset_kernel_stats(loamx_ctx* ctx) {
  if (!ctx) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = resolve_events(ctx);
  if (rc != LOAMX_OK) return rc;
  memset(ctx->stats, 0, sizeof(ctx->stats));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->ws[WS_COUNTERS].p) HIP_TRY(ctx, hipMemset(wsp<unsigned char>(ctx, WS_COUNTERS) + 8, 0, 48));
  if (ctx->ws[WS_EXTRACT_EVENTS].p) {
    unsigned long long ev[4] = {0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpy(ev, ctx->ws[WS_EXTRACT_EVENTS].p, sizeof(ev), hipMemcpyDeviceToHost));
    ctx->features_base = ev[2];
  }
  return LOAMX_OK;
}

int loamx_ctx_get_kernel_stats(loamx_ctx* ctx, loamx_kernel_stat* stats) {
  if (!ctx || !stats) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = resolve_events(ctx);
  if (rc != LOAMX_OK) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  unsigned long long slots[6];
  rc = read_sweep_slots(ctx, slots);
  if (rc != LOAMX_OK) return rc;
  memcpy(stats, ctx->stats, sizeof(ctx->stats));
  // sweep: one association record per slot (edge 72 B, plane 56 B)
  stats[LOAMX_K_SWEEP].algorithmic_bytes = 72.0 * (double)slots[0] + 56.0 * (double)slots[1];
  // associate: read the 24 B source point, write the record + the 4 B nearest index
  stats[LOAMX_K_ASSOC].algorithmic_bytes = 100.0 * (double)slots[2] + 84.0 * (double)slots[3];
  stats[LOAMX_K_MOMENT].algorithmic_bytes = 56.0 * (double)slots[4];  // moment pass: every plane record once
  stats[LOAMX_K_GRID].algorithmic_bytes = (double)slots[5];            // index builds: counted by the build kernels
  if (ctx->ws[WS_EXTRACT_EVENTS].p) {  // fused extraction: + (4 + 24) B per feature written (counted by the kernel since the reset)
    unsigned long long ev[4] = {0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpy(ev, ctx->ws[WS_EXTRACT_EVENTS].p, sizeof(ev), hipMemcpyDeviceToHost));
    stats[LOAMX_K_EXTRACT_FUSED].algorithmic_bytes += 28.0 * (double)(ev[2] - ctx->features_base);
    // the selection with its fused compaction copies every feature: 24 B read + 24 B written per feature on top of the
    // curvature words it was priced with at launch (without them the scope read as 0.07 of HBM while the
    // kernel moves 4.9 TB/s); the features are counted by the kernel itself (events[2])
    if (stats[LOAMX_K_SELECT].launches != 0 && stats[LOAMX_K_EXTRACT_FUSED].launches == 0)
      stats[LOAMX_K_SELECT].algorithmic_bytes += 48.0 * (double)(ev[2] - ctx->features_base);
  }
  return LOAMX_OK;
}

/* ---- synthetic workload ------------------------------------------------------------------------------- */
void loamx_synth_pair_pose(uint64_t seed, uint64_t pair_id, double pose_out[7]) {
  const loamx_synth::Pose7 p = loamx_synth::pair_pose(seed, pair_id);
  for (int i = 0; i < 4; i++) pose_out[i] = p.q[i];
  for (int i = 0; i < 3; i++) pose_out[4 + i] = p.t[i];
}

void loamx_synth_scan_host(uint64_t seed, uint64_t pair_id, uint32_t which, uint32_t H, uint32_t W, double sigma,
                           double* xyz_out) {
  const loamx_synth::Pose7 pose = loamx_synth::pair_pose(seed, pair_id);
  for (uint32_t l = 0; l < H; l++)
    for (uint32_t c = 0; c < W; c++)
      loamx_synth::scan_point(seed, pair_id, which, pose, l, c, H, W, sigma, xyz_out + 3 * ((size_t)l * W + c));
}

int loamx_synth_scan_pairs_dev(loamx_ctx* ctx, uint64_t seed, uint64_t first_pair, size_t n_pairs, uint32_t H, uint32_t W,
                               double sigma, double* d_xyz) {
  if (!ctx) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  launch_synth_pairs(seed, first_pair, n_pairs, H, W, sigma, d_xyz, ctx->stream);
  CHECK_LAUNCH(ctx, "synth_kernel");
  return LOAMX_OK;
}

/* ---- raw device memory helpers ------------------------------------------------------------------------ */
int loamx_dev_alloc(loamx_ctx* ctx, size_t bytes, void** d_ptr) {
  if (!ctx || !d_ptr) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMalloc(d_ptr, bytes ? bytes : 8));
  return LOAMX_OK;
}
int loamx_dev_free(loamx_ctx* ctx, void* d_ptr) {
  if (!ctx) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipFree(d_ptr));
  return LOAMX_OK;
}
int loamx_copy_to_device(loamx_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
  if (!ctx) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return LOAMX_OK;
}
int loamx_copy_to_host(loamx_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
  if (!ctx) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return LOAMX_OK;
}

/* ---- multi-GPU batch mode: RCCL gather of the result records ---------------------------------------------- */
}  // extern "C"

struct loamx_comm {
  ncclComm_t comm = nullptr;
  bool owned = false;
  int world = 1, rank = 0, device = 0;
  double* d_scalar = nullptr;  // barrier / max-reduce scratch (device)
  uint64_t enqueued[LOAMX_COMM_STAT_COUNT] = {};  // what loamx_gather_results_dev / loamx_comm_barrier really enqueued
};

namespace {
// RCCL is opened when the first loamx_comm_* entry point runs: single-GPU users and the host entry points load
// libloamx.so on a machine without librccl. (In a process that imported torch first, "librccl.so.1" resolves to the copy
// torch already mapped: same SONAME.)
struct Rccl {
#define LOAMX_RCCL_FN(name) decltype(&::nccl##name) name = nullptr;
  LOAMX_RCCL_FN(CommCount) LOAMX_RCCL_FN(CommUserRank) LOAMX_RCCL_FN(CommCuDevice) LOAMX_RCCL_FN(GetUniqueId)
  LOAMX_RCCL_FN(CommInitRank) LOAMX_RCCL_FN(GetErrorString) LOAMX_RCCL_FN(CommDestroy) LOAMX_RCCL_FN(AllGather)
  LOAMX_RCCL_FN(GroupStart) LOAMX_RCCL_FN(GroupEnd) LOAMX_RCCL_FN(Broadcast) LOAMX_RCCL_FN(AllReduce)
#undef LOAMX_RCCL_FN
  std::string error;
  bool ok = false;
  Rccl() {
    void* h = nullptr;
    // LOAMX_RCCL_LIB names the one library to open (a site-specific build; the tests point it at a missing file)
    const char* forced = getenv("LOAMX_RCCL_LIB");
    std::string why;
    for (const char* n : {forced ? forced : "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
      // dlerror() clears the message it returns: read it ONCE per failed attempt (the first attempt's message is kept)
      const char* e = dlerror();
      if (why.empty()) why = std::string(n) + ": " + (e ? e : "?");
      if (forced) break;
    }
    if (!h) {
      error = "librccl not found: " + why;
      return;
    }
    bool all = true;
#define LOAMX_RCCL_FN(name) all &= (name = reinterpret_cast<decltype(name)>(dlsym(h, "nccl" #name))) != nullptr;
    LOAMX_RCCL_FN(CommCount) LOAMX_RCCL_FN(CommUserRank) LOAMX_RCCL_FN(CommCuDevice) LOAMX_RCCL_FN(GetUniqueId)
    LOAMX_RCCL_FN(CommInitRank) LOAMX_RCCL_FN(GetErrorString) LOAMX_RCCL_FN(CommDestroy) LOAMX_RCCL_FN(AllGather)
    LOAMX_RCCL_FN(GroupStart) LOAMX_RCCL_FN(GroupEnd) LOAMX_RCCL_FN(Broadcast) LOAMX_RCCL_FN(AllReduce)
#undef LOAMX_RCCL_FN
    ok = all;
    if (!ok) error = "librccl lacks an expected ncclXxx symbol";
  }
};
const Rccl& rccl() {
  static const Rccl r;
  return r;
}
#define RCCL_NEED(ctx)                                                          \
  do {                                                                          \
    if (!rccl().ok) return fail(ctx, LOAMX_ERR_COMM, rccl().error);             \
  } while (0)
#define NCCL_TRY(ctx, expr)                                                                        \
  do {                                                                                             \
    ncclResult_t r_ = (expr);                                                                      \
    if (r_ != ncclSuccess) return fail(ctx, LOAMX_ERR_COMM, std::string(#expr) + ": " + rccl().GetErrorString(r_)); \
  } while (0)

int comm_finish_init(loamx_ctx* ctx, loamx_comm* c) {
  NCCL_TRY(ctx, rccl().CommCount(c->comm, &c->world));
  NCCL_TRY(ctx, rccl().CommUserRank(c->comm, &c->rank));
  NCCL_TRY(ctx, rccl().CommCuDevice(c->comm, &c->device));
  if (c->device != ctx->device) return fail(ctx, LOAMX_ERR_BAD_PARAM, "communicator and context live on different devices");
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&c->d_scalar), 2 * sizeof(double)));
  return LOAMX_OK;
}
}  // namespace

extern "C" {

void loamx_shard_range(size_t total_pairs, int world_size, int rank, size_t* first, size_t* count) {
  const size_t w = world_size > 0 ? (size_t)world_size : 1, r = rank > 0 ? (size_t)rank : 0;
  const size_t base = total_pairs / w, rem = total_pairs % w;
  if (first) *first = r * base + (r < rem ? r : rem);
  if (count) *count = r < w ? base + (r < rem ? 1 : 0) : 0;
}

int loamx_comm_get_unique_id(unsigned char id_out[LOAMX_COMM_ID_BYTES]) {
  static_assert(LOAMX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  if (!id_out) return LOAMX_ERR_BAD_PARAM;
  if (!rccl().ok) return LOAMX_ERR_COMM;
  ncclUniqueId id;
  if (rccl().GetUniqueId(&id) != ncclSuccess) return LOAMX_ERR_COMM;
  memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return LOAMX_OK;
}

int loamx_comm_create(loamx_ctx* ctx, const unsigned char id[LOAMX_COMM_ID_BYTES], int world_size, int rank, loamx_comm** out) {
  if (!ctx || !id || !out) return LOAMX_ERR_BAD_PARAM;
  *out = nullptr;
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (world_size < 1 || rank < 0 || rank >= world_size) return fail(ctx, LOAMX_ERR_BAD_PARAM, "bad world size / rank");
  RCCL_NEED(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  ncclUniqueId uid;
  memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
  loamx_comm* c = new loamx_comm;
  c->owned = true;
  ncclResult_t r = rccl().CommInitRank(&c->comm, world_size, uid, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(ctx, LOAMX_ERR_COMM, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
  }
  int rc = comm_finish_init(ctx, c);
  if (rc != LOAMX_OK) {
    loamx_comm_destroy(c);
    return rc;
  }
  *out = c;
  return LOAMX_OK;
}

int loamx_comm_wrap(loamx_ctx* ctx, void* nccl_comm, loamx_comm** out) {
  if (!ctx || !nccl_comm || !out) return LOAMX_ERR_BAD_PARAM;
  *out = nullptr;
  std::lock_guard<std::mutex> lock(ctx->mu);
  RCCL_NEED(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  loamx_comm* c = new loamx_comm;
  c->comm = static_cast<ncclComm_t>(nccl_comm), c->owned = false;
  int rc = comm_finish_init(ctx, c);
  if (rc != LOAMX_OK) {
    loamx_comm_destroy(c);
    return rc;
  }
  *out = c;
  return LOAMX_OK;
}

void loamx_comm_destroy(loamx_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->d_scalar) (void)hipFree(c->d_scalar);
  if (c->owned && c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
  delete c;
}

int loamx_comm_info(const loamx_comm* c, int* world_size, int* rank, int* device) {
  if (!c) return LOAMX_ERR_BAD_PARAM;
  if (world_size) *world_size = c->world;
  if (rank) *rank = c->rank;
  if (device) *device = c->device;
  return LOAMX_OK;
}

int loamx_gather_results_dev(loamx_ctx* ctx, loamx_comm* c, const loamx_reg_result* d_local, size_t n_local, size_t total_pairs,
                             loamx_reg_result* d_all) {
  if (!ctx || !c || !d_all || (n_local && !d_local)) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  size_t first = 0, count = 0;
  loamx_shard_range(total_pairs, c->world, c->rank, &first, &count);
  if (count != n_local) return fail(ctx, LOAMX_ERR_BAD_PARAM, "n_local is not this rank's shard of total_pairs (loamx_shard_range)");
  if (total_pairs == 0) return LOAMX_OK;
  untimed(ctx);
  hipStream_t s = ctx->stream;
  static_assert(sizeof(loamx_reg_result) == 64, "record size");
  // One rank: a device copy — unless option FORCE_RCCL is set, which sends the one-rank communicator through BOTH collective
  // forms below (the all-gather, then the grouped broadcast in place: same bytes), so that they have executed before
  // the first multi-GPU node runs them.
  const bool forced = c->world == 1 && (ctx->reg_flags & kRegFlagForceRccl) != 0;
  if (c->world == 1 && !forced) {
    if (d_all != d_local) HIP_TRY(ctx, hipMemcpyAsync(d_all, d_local, n_local * sizeof(loamx_reg_result), hipMemcpyDeviceToDevice, s));
    c->enqueued[LOAMX_COMM_STAT_MEMCPY]++;
    return LOAMX_OK;
  }
  if (total_pairs % (size_t)c->world == 0) {  // equal shards: one all-gather of n_local * 64 bytes per rank
    NCCL_TRY(ctx, rccl().AllGather(d_local, d_all, n_local * sizeof(loamx_reg_result), ncclChar, c->comm, s));
    c->enqueued[LOAMX_COMM_STAT_ALL_GATHER]++;
    if (!forced) return LOAMX_OK;
    d_local = d_all;  // (forced: the broadcast form runs in place on what the all-gather delivered)
  }
  // uneven shards (sizes differ by one): every rank broadcasts its block to its place, as one grouped operation
  NCCL_TRY(ctx, rccl().GroupStart());
  for (int r = 0; r < c->world; r++) {
    size_t f = 0, n = 0;
    loamx_shard_range(total_pairs, c->world, r, &f, &n);
    if (n == 0) continue;
    ncclResult_t rr = rccl().Broadcast(r == c->rank ? static_cast<const void*>(d_local) : static_cast<const void*>(d_all + f), d_all + f,
                                    n * sizeof(loamx_reg_result), ncclChar, r, c->comm, s);
    if (rr != ncclSuccess) {
      (void)rccl().GroupEnd();
      return fail(ctx, LOAMX_ERR_COMM, std::string("ncclBroadcast: ") + rccl().GetErrorString(rr));
    }
    c->enqueued[LOAMX_COMM_STAT_BROADCAST]++;
  }
  NCCL_TRY(ctx, rccl().GroupEnd());
  return LOAMX_OK;
}

int loamx_comm_stats(const loamx_comm* c, uint64_t counts[LOAMX_COMM_STAT_COUNT]) {
  if (!c || !counts) return LOAMX_ERR_BAD_PARAM;
  for (int i = 0; i < LOAMX_COMM_STAT_COUNT; i++) counts[i] = c->enqueued[i];
  return LOAMX_OK;
}

int loamx_comm_barrier(loamx_ctx* ctx, loamx_comm* c, double* max_value) {
  if (!ctx || !c) return LOAMX_ERR_BAD_PARAM;
  std::lock_guard<std::mutex> lock(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  untimed(ctx);
  hipStream_t s = ctx->stream;
  const double v = max_value ? *max_value : 0.0;
  HIP_TRY(ctx, hipMemcpyAsync(c->d_scalar, &v, sizeof(double), hipMemcpyHostToDevice, s));
  if (c->world > 1 || (ctx->reg_flags & kRegFlagForceRccl)) {
    NCCL_TRY(ctx, rccl().AllReduce(c->d_scalar, c->d_scalar + 1, 1, ncclDouble, ncclMax, c->comm, s));
    c->enqueued[LOAMX_COMM_STAT_ALL_REDUCE]++;
  } else {
    HIP_TRY(ctx, hipMemcpyAsync(c->d_scalar + 1, c->d_scalar, sizeof(double), hipMemcpyDeviceToDevice, s));
    c->enqueued[LOAMX_COMM_STAT_MEMCPY]++;
  }
  double o = 0.0;
  HIP_TRY(ctx, hipMemcpyAsync(&o, c->d_scalar + 1, sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  if (max_value) *max_value = o;
  return LOAMX_OK;
}

}  // extern "C"
