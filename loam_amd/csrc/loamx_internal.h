// loamx_internal.h — declarations shared by the .hip translation units of libloamx.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/loamx.h"
#include "extract_math.h"
#include "reg_math.h"

namespace loamx {

/* ---- kernel launches with timing attached (loamx_api.hip: TimedScope) ------------------------------------
 * While a timing scope of the calling thread is in "attach" mode, the first kernel launched records the scope's
 * start event with its own begin and every kernel re-records the stop event with its end (the last one
 * stands): hipExtLaunchKernelGGL takes the timestamps from the dispatch itself, so no marker packets sit between
 * the kernels of the timed region (barrier-style hipEventRecord cost 2.5 % of the step). */
struct LaunchScope {
  hipEvent_t start, stop;
  bool first;
};
extern thread_local LaunchScope* g_launch_scope;
// LOAMX_DEBUG_SYNC=1 (read once): every launch is followed by a stream synchronisation and a line on stderr with the
// kernel's name and the status — the last line before a GPU fault names the kernel that faulted (debugging only)
extern int g_debug_sync;
template <typename F, typename... Args>
inline void launch_kernel_named(const char* name, F kernel, const dim3& grid, const dim3& block, size_t shmem, hipStream_t s, Args... args) {
  LaunchScope* sc = g_launch_scope;
  if (g_debug_sync) fprintf(stderr, "[loamx] launch %s grid %u x %u block %u\n", name, grid.x, grid.y, block.x), fflush(stderr);
  if (sc) {
    hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)shmem, s, sc->first ? sc->start : nullptr, sc->stop, 0, args...);
    sc->first = false;
  } else {
    hipLaunchKernelGGL(kernel, grid, block, shmem, s, args...);
  }
  if (g_debug_sync) {
    const hipError_t e = hipStreamSynchronize(s);
    fprintf(stderr, "[loamx]   done %s: %s\n", name, hipGetErrorString(e)), fflush(stderr);
  }
}
#define launch_kernel(kernel, ...) launch_kernel_named(#kernel, kernel, __VA_ARGS__)

/* ---- extraction (extract_kernels.hip) ---------------------------------------------------------- */
constexpr int kMaxNeighborPoints = 16;  // LDS halo bound of curvature_valid_kernel
constexpr int kMaxLineWidth = 4096;     // select_kernel keeps one line's curvature + mask in LDS

// staging layout written by select_kernel: per (scan, line, sector) a fixed slot of cap entries
struct ExtractStage {
  uint32_t* edge_stage;    // [n_scans][H][S][cap_edge]
  uint32_t* planar_stage;  // [n_scans][H][S][cap_planar]
  uint32_t* edge_cnt;      // [n_scans][H][S]
  uint32_t* planar_cnt;    // [n_scans][H][S]
};

// d_xyz: n_scans x H x W x 3 scalars, double or (f32) float
// Outputs select_mis_kernel writes itself when the compaction is fused into the selection (see the kernel).
struct ExtractFused {
  unsigned long long* line_tot;  // [n_lines] published totals of every scan line (bit 63), or the tie mark (bit 62); zeroed before the launch
  uint32_t fuse;                 // 1: select_mis_kernel writes the final arrays itself (chained scan over line_tot)
  const void* xyz;               // the scans (double, or float when f32)
  uint32_t f32;
  uint32_t* edge_idx;            // outputs as launch_compact
  uint32_t* n_edge;
  double* edge_xyz;
  size_t edge_stride;
  uint32_t* planar_idx;
  uint32_t* n_planar;
  double* planar_xyz;
  size_t planar_stride;
  uint32_t* error;  // flag word (zeroed before every launch, next to line_tot). Bit 0: a wavefront gave up waiting for the
                    // lines before it (or they are tied): the fallback compaction gathers the batch from the stage arrays;
                    // bit 1: some line is tied (replay_kernel has work)
  unsigned long long* events;  // [3] cumulative: scan lines replayed in the reference's tie order, give-up fallbacks taken, features written by the fused path
  // optional (select_rows_kernel with the fused compaction only): bounding boxes of every scan's edge / planar features as ordered
  // keys (dbl_key), box_min / box_max[scan][kind][axis], kind 0 = edge, 1 = planar; the caller presets them (all ones / zero).
  // Complete iff the flag word *error stays zero (no line gave up or was tied: those scans went through compact_kernel).
  unsigned long long* box_min;
  unsigned long long* box_max;
  // select_rows_kernel only (round 5). no_stage: the stage arrays and per-sector counts — the fallback compaction's input — are
  // not written; only_if: the launch leaves at once unless this word is non-zero. launch_select runs the fused selection
  // without the stage arrays (4 B per feature saved) and, behind it, the plain one under only_if = error: when a line gave up
  // or was tied, that second launch produces what replay_kernel / compact_kernel read; otherwise it costs its launch.
  uint32_t no_stage;
  const uint32_t* only_if;
};
// doubles as unsigned keys with the same order (for atomicMin / atomicMax)
__device__ __forceinline__ unsigned long long dbl_key(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_dbl(unsigned long long k) {
  return __longlong_as_double((long long)((k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k));
}

void launch_curvature_valid(const void* d_xyz, bool f32, size_t n_scans, const ExtractParams& P, double* d_curv,
                            uint8_t* d_mask, hipStream_t s);
// fz != nullptr: the selection may also write the final feature arrays (returns true if it did: no
// launch_compact needed); otherwise, or when it returns false, stage + counts only.
bool launch_select(const double* d_curv, const uint8_t* d_mask, size_t n_scans, const ExtractParams& P,
                   const ExtractStage& st, const ExtractFused* fz, hipStream_t s, bool* rows_ran = nullptr);
// will launch_select run the kernel that fills ExtractFused::box_* (select_rows_kernel with the fused compaction)?
bool launch_select_takes_boxes(const ExtractParams& P);
// edge_stride / planar_stride: entries per scan in the output arrays
// rows a5-a10 in one kernel (extract_fused_kernel); false: not applicable to these parameters, nothing was launched
bool launch_extract_fused(const void* d_xyz, bool f32, size_t n_scans, const ExtractParams& P, const ExtractStage& st,
                          const ExtractFused& fz, double* d_curv, uint8_t* d_mask, hipStream_t s);
// rows a5-a10 in one kernel, four scan lines per wavefront (opt-in: context option FUSED_ROWS). false: not requested or not
// applicable to these parameters, nothing was launched
bool launch_extract_rows_fused(const void* d_xyz, bool f32, size_t n_scans, const ExtractParams& P, const ExtractStage& st,
                               const ExtractFused& fz, double* d_curv, uint8_t* d_mask, hipStream_t s);
// the tie path: scan lines the selection kernel marked are redone in the reference's std::sort order (stage + counts)
bool launch_extract_split_ok(const ExtractParams& P, size_t n_scans);
void launch_extract_init(unsigned long long* line_tot, size_t n_tot, unsigned long long* box_min, unsigned long long* box_max, size_t n_box,
                         hipStream_t s);
void launch_replay(const double* d_curv, const uint8_t* d_mask, size_t n_scans, const ExtractParams& P, const ExtractStage& st,
                   const ExtractFused& fz, hipStream_t s);
// only_if != nullptr: every workgroup first reads *only_if and leaves if it is 0 (the fused compaction's fallback)
void launch_compact(const void* d_xyz, bool f32, size_t n_scans, const ExtractParams& P, const ExtractStage& st,
                    uint32_t* d_edge_idx, uint32_t* d_n_edge, double* d_edge_xyz, size_t edge_stride,
                    uint32_t* d_planar_idx, uint32_t* d_n_planar, double* d_planar_xyz, size_t planar_stride,
                    hipStream_t s, const uint32_t* only_if = nullptr, unsigned long long* fallback_counter = nullptr);

/* ---- registration (register_kernels.hip) ------------------------------------------------------- */
struct RegConfig {
  int k_edge, k_plane;
  int min_line_pts, min_plane_pts;
  double r_edge, r_plane;
  double pass_edge, pass_plane;  // knn_radius_pass_max(r_*): largest squared distance that passes the strict radius filter
  double min_line_cond, max_avg_plane_dist;
  uint32_t max_iterations;
  double rot_thresh, pos_thresh;
  uint32_t min_associations;
  uint32_t flags;  // the context's switches (loamx_ctx_set_option): kRegFlag*
};
constexpr uint32_t kRegFlagNoMoments = 1u;     // never use the moments: every evaluation streams the records
constexpr uint32_t kRegFlagNoPackedGrid = 2u;  // scan-sized sets through the single-workgroup index build with the 32-bit table
constexpr uint32_t kRegFlagNoBigGrid = 4u;     // map-sized sets through the single-workgroup build as well
constexpr uint32_t kRegFlagNoGridSide = 8u;    // source index builds behind the target builds on the context stream
constexpr uint32_t kRegFlagPoison = 16u;       // registration scratch starts as 0xFF bytes
constexpr uint32_t kRegFlagQueueTwoStage = 32u;  // queue chain: always lean 5x5x5 search + listed leftovers (launch_associate)
constexpr uint32_t kRegFlagQueueOneStage = 64u;  // queue chain: always the FP64 search over all rounds in one kernel
constexpr uint32_t kRegFlagForceRccl = 256u;     // a one-rank communicator really enqueues ncclAllGather / ncclBroadcast / ncclAllReduce (host side only)
constexpr uint32_t kRegFlagNoCoopLeft = 512u;    // listed queue leftovers one lane per query (associate_knn_left_kernel, round 3), not one wavefront per query
constexpr uint32_t kRegFlagNoRefMoments = 1024u;  // first ICF iteration as in round 3: five sweeps of the records, no moments
constexpr uint32_t kRegFlagNoSmallSets = 8192u;     // edge-sized sets through grid_build_kernel like every other set
constexpr uint32_t kRegFlagCheckFinite = 4096u;     // the "_dev" entry points look for non-finite input coordinates first (host side only)
constexpr uint32_t kRegFlagNoExtractBoxes = 2048u;  // the index builds take their bounding boxes themselves even when the extraction left them
constexpr uint32_t kRegFlagNoMixedAssoc = 128u;  // edge and plane first kernels as separate launches on two streams (launch_associate)

// One target feature set's spatial index (device pointers into the workspace)
struct GridSet {
  GridDesc* desc;        // [n_pairs]
  uint32_t* cell_start;  // [n_pairs][kGridCellsCap + 1]
  GridPoint* sorted;     // [n_pairs][stride] points re-ordered cell by cell (32 B each: xyz + original index)
  size_t stride;
  float* rel;            // [n_pairs][3][stride] single-precision offsets of the sorted points from the grid origin
                         // (x plane, y plane, z plane) for the FP32 pre-selection; nullptr for source sets
  uint32_t cells_cap;    // 0: kGridCellsCap. A persistent index of a map-sized set (one "pair") may own a larger
                         // cell table: a million-point map at the scan's cell edge puts hundreds of points in a cell
};

// Association records, structure-of-arrays over the whole batch (field-major) so the residual
// sweep streams each field with fully coalesced 8-byte loads.
//   edges : 9 fields (moved point xyz, line a xyz, line b xyz) x n_pairs x edge_stride
//   planes: 7 fields (moved point xyz, normal xyz, d)          x n_pairs x planar_stride
// An invalid slot has NaN in field 0.
struct AssocBuffers {
  double* edge;       // [9][n_pairs * edge_stride]
  double* plane;      // [7][n_pairs * planar_stride]
  uint32_t* nn_edge;        // [1 + kMaxK][n_pairs * edge_stride]   neighbour count, then positions in the sorted target
  uint32_t* nn_plane;       // [1 + kMaxK][n_pairs * planar_stride]
  uint32_t* rnn_edge;       // as nn_*, for the queued queries, indexed by queue position
  uint32_t* rnn_plane;
  uint32_t* nearest_edge;   // [n_pairs * edge_stride]   nearest target index (detail capture)
  uint32_t* nearest_plane;  // [n_pairs * planar_stride]
  uint32_t* rest_edge;      // [n_pairs * edge_stride]   queue of the queries round 1 of the k-NN did not finish
  uint32_t* rest_plane;     // [n_pairs * planar_stride]
  uint32_t* exact_edge;     // queues of the queries the keyed collector could not decide (exact collector re-runs them)
  uint32_t* exact_plane;
  uint32_t* n_assoc;  // [n_pairs][8]: valid edge / plane associations of the current iteration [0,1], lengths of the
                      // queues rest_* [2,3] and exact_* [4,5]
};

struct PairState {
  double est[7];
  LmState lm;
  uint32_t active;       // outer loop still running
  uint32_t termination;
  uint32_t iterations;
  uint32_t first_sweep;  // next sweep is the iteration-0 evaluation
  uint32_t stream_planes;  // 1: the next sweep streams this pair's plane records (its moments cannot stand in for them)
  uint32_t use_moments;    // this ICF iteration runs the moment pass for the pair
  // First ICF iteration (round 4): after its first evaluation (a sweep at the identity update) the moments are taken
  // relative to the first candidate instead of the identity: mom_ref_on = 1, mom_ref = that candidate (lm_step_pair)
  uint32_t mom_ref_on;
  double mom_ref[7];
};

constexpr int kSweepThreads = 256;
#ifndef LOAMX_SWEEP_ITEMS
#define LOAMX_SWEEP_ITEMS 16  // measured per launch: 4 -> 0.268 ms, 8 -> 0.179, 16 -> 0.148 (56 % of HBM peak), 32 -> 0.150, 64 -> 0.206
#endif
constexpr int kSweepItems = LOAMX_SWEEP_ITEMS;  // association slots per thread
constexpr int kSweepChunk = kSweepThreads * kSweepItems;

struct RegBatch {
  size_t n_pairs;
  size_t edge_stride, planar_stride;  // capacity (points) of one feature set; also the slot pitch of assoc/grid arrays
  // Input feature sets of pair p start at base + p * in_pitch * stride * 3 and their counts at
  // n[p * in_pitch]: in_pitch = 1 for separate arrays, 2 when source and target scans are interleaved.
  uint32_t in_pitch;
  const double* src_edge;
  const uint32_t* n_src_edge;
  const double* src_planar;
  const uint32_t* n_src_planar;
  const double* tgt_edge;
  const uint32_t* n_tgt_edge;
  const double* tgt_planar;
  const uint32_t* n_tgt_planar;
  const double* init;  // may be null
  GridSet grid_edge, grid_plane;          // target sets: searched
  GridSet src_grid_edge, src_grid_plane;  // source sets: only their cell-sorted order is used (coherent queries)
  GridPoint* sort_scratch;      // [n_pairs][max(edge_stride, planar_stride)] scratch of the multi-workgroup target builds
  GridPoint* sort_scratch_src;  // the same for the ordered source builds (they run next to the target builds on another stream)
  AssocBuffers assoc;
  PairState* state;      // [n_pairs]
  double* partials;      // [n_pairs][blocks_per_pair][kAccSize]
  double* mom_partials;  // [n_pairs][mom_blocks_per_pair][4][kMomSize + 2] wavefront tiles of the moment pass
  double* moments;       // [n_pairs][kMomSize + 2] plane moment matrix (16x16), max |s0|, max |v|^2 (reg_math.h)
  uint32_t mom_blocks_per_pair;
  uint32_t* flagged_list;   // [n_pairs][mom_blocks_per_pair][4 wavefronts][kSweepChunk / 4] plane slots the moments leave out, in slot order
  uint32_t* flagged_count;  // [n_pairs][mom_blocks_per_pair][4]
  uint32_t blocks_per_pair;
  uint32_t* n_active;    // device counter read back by the host after every outer iteration
  unsigned long long* sweep_slots;  // [0,1] edge / plane association slots streamed by sweep_kernel (roofline bytes);
                                    // [2,3] assoc_slots; [4] plane slots streamed by the moment pass
  unsigned long long* assoc_slots;  // [2] edge / plane source features processed by associate_kernel
  unsigned long long* grid_bytes;   // roofline bytes of the single-workgroup index builds (may be null)
  loamx_iter_info* iter_info;  // optional [n_pairs][max_iterations]
  uint32_t* max_counts;   // [6] over the active pairs (state_init_kernel; read back by the host): largest source edge / planar
                          // count, largest target edge / planar count, smallest target edge / planar count
  uint32_t knn_mode_edge, knn_mode_plane;  // 0: grid and brute-force k-NN kernels both launched (target sizes on both
                                           // sides of kBruteMax, or unknown); 1: grid only; 2: brute force only
  uint32_t assoc_blocks_edge, assoc_blocks_plane;  // workgroups per pair of the association kernels; 0xFFFFFFFF = by capacity
  uint32_t want_nearest;  // 1: a detail hook will read nearest_* (RegistrationDetail pairs); 0: the fit kernels skip that write
  uint32_t ref_moments;   // 1: the first ICF iteration takes its moments at its first candidate after ONE sweep (enqueue_icf_iteration)
  // optional: bounding boxes of the input feature sets, computed by the extraction that produced them (ExtractFused::box_*):
  // [scan][kind][axis] keys, scan = pair * in_pitch (+ 1 for the source scan of interleaved pairs). Used by the index builds
  // in place of their own read pass while *box_bad == 0; nullptr: the builds compute the boxes.
  const unsigned long long* box_min;
  const unsigned long long* box_max;
  const uint32_t* box_bad;
  uint32_t small_edge_sets;  // 1: pairs whose TARGET edge set holds at most kBruteMax points get both edge sets from small_sets_build_kernel;
                             // 2: the target is a persistent index whose edge set is that small: every source edge set in its given order
};

// scratch the multi-workgroup build of a map-sized target set needs per pair (at B.sort_scratch + pair * stride points)
// (box keys, one cursor per cell, the tile sums of the table scan)
constexpr size_t kGridBigScratchBytes = 64 + ((size_t)kGridCellsCap + kGridCellsCap / 4096 + 8) * sizeof(uint32_t);
// cell table of a map-sized persistent index. Measured on a 1.02 M-point map (config 5; index build / registration
// of a 39 k-feature scan): 2^16 cells 1.04 / 6.71 ms, 2^17 1.07 / 5.97, 2^18 1.38 / 5.88, 2^19 1.77 / 5.71,
// 2^20 2.49 / 5.74, 2^21 4.75 / 8.19 (cells too small for the 5th neighbour: more second rounds)
constexpr uint32_t kGridSmallCap = 20480;  // sets up to this size are indexed by the packed single-workgroup build (no scratch)
constexpr uint32_t kBruteMax = 512;  // target sets up to this size are searched by associate_knn_brute_kernel
constexpr uint32_t kGridMapCellsCap = 1u << 18;
// scan-sized sets (the capacity bounds the count) take the packed cell table + LDS lists and need no scratch copy;
// the workspace is sized with the same predicate the launcher routes by
inline bool grid_small(size_t stride, uint32_t reg_flags) { return stride <= kGridSmallCap && !(reg_flags & kRegFlagNoPackedGrid); }
void launch_grid_build_targets(const RegBatch& B, const RegConfig& C, hipStream_t s);
void launch_grid_build_sources(const RegBatch& B, const RegConfig& C, hipStream_t s);
void launch_grid_build_target(const RegBatch& B, const RegConfig& C, bool plane, hipStream_t s);  // one feature kind
void launch_grid_build_source(const RegBatch& B, const RegConfig& C, bool plane, hipStream_t s);
void launch_state_init(const RegBatch& B, const RegConfig& C, hipStream_t s);
// aux == nullptr: everything on s; aux2 == nullptr: the plane queue chain follows the edge chain on aux
// knn_scope != nullptr: the plane round-1 k-NN kernel is launched with that timing scope attached
constexpr uint32_t kAssocEdges = 1u, kAssocPlanes = 2u;
void launch_associate(const RegBatch& B, const RegConfig& C, hipStream_t s, hipStream_t aux, hipStream_t aux2, hipEvent_t ev_fork,
                      hipEvent_t ev_mid, hipEvent_t ev_join, hipEvent_t ev_join2, LaunchScope* knn_scope = nullptr,
                      uint32_t what = kAssocEdges | kAssocPlanes);
#ifdef LOAMX_NN_SAME_STATS
void debug_nn_same(const RegBatch& B, uint32_t it, hipStream_t s);
#endif
void launch_lm_begin(const RegBatch& B, const RegConfig& C, hipStream_t s);
void launch_sweep(const RegBatch& B, hipStream_t s);
void launch_lm_pair_loop(const RegBatch& B, const RegConfig& C, hipStream_t s);
void launch_lm_step(const RegBatch& B, hipStream_t s);
void launch_moments(const RegBatch& B, hipStream_t s);
void launch_outer_update(const RegBatch& B, const RegConfig& C, hipStream_t s);
void launch_write_results(const RegBatch& B, loamx_reg_result* d_results, hipStream_t s);

// incremental insert into a persistent index whose grid stays (register_kernels.hip); `cells` = entries of the cell table
size_t index_insert_ws_bytes(size_t cells, size_t n_add);
void launch_index_insert_count(const GridSet& gs, size_t cells, const double* d_add, uint32_t n_add, void* ws, hipStream_t s);
void launch_index_insert_merge(const GridSet& gs, size_t cells, uint32_t n_old, const double* d_add, uint32_t n_add, void* ws,
                               GridPoint* sorted2, float* rel2, hipStream_t s);

// direct read-outs of rows a16-a19 (loamx_fit_lines / _planes, loamx_knn_search, loamx_associate)
constexpr int kFitMaxK = 32;  // point sets of the host-callable fits (the association kernels keep <= kMaxK neighbours)
struct AssocDumpSet {  // device arrays of one feature kind, indexed by the caller's source index
  uint32_t* nn_count;  // [n_src]
  uint32_t* nn_idx;    // [n_src][k]
  uint8_t* valid;      // [n_src]
  double* moved;       // [n_src][3]
  double* prim;        // [n_src][6] line a, b | [n_src][4] plane normal, d
};
void launch_fit_sets(bool plane, const double* d_pts, size_t n_sets, int k, double* d_prim, double* d_aux, hipStream_t s);
void launch_knn_queries(const GridSet& gs, const double* d_q, size_t n_q, int k, double max_dist, uint32_t* d_idx, uint32_t* d_count,
                        hipStream_t s);
void launch_assoc_dump(const RegBatch& B, const RegConfig& C, const AssocDumpSet& edge, const AssocDumpSet& plane, hipStream_t s);

/* ---- non-finite input check of the "_dev" entry points (context option CHECK_FINITE; synth_kernels.hip) --------- */
// sets of `stride` points, `pitch` sets apart; d_n: points held by set i at d_n[i * pitch], or nullptr = all `stride`
void launch_check_finite(const void* d_pts, bool f32, const uint32_t* d_n, size_t n_sets, size_t stride, uint32_t pitch, uint32_t* d_flag,
                         hipStream_t s);
void launch_check_finite_scalars(const double* d_v, size_t n_scalars, uint32_t* d_flag, hipStream_t s);

/* ---- synthetic generator (synth_kernels.hip) --------------------------------------------------- */
void launch_synth_pairs(uint64_t seed, uint64_t first_pair, size_t n_pairs, uint32_t H, uint32_t W, double sigma,
                        double* d_xyz, hipStream_t s);

}  // namespace loamx
