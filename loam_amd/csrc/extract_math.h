// extract_math.h — per-point math of the feature-extraction kernels (curvature, range, validity).
// __host__ __device__ so tests/hostcheck can check the exact kernel arithmetic bit for bit against
// the oracle on the CPU; the product only calls it from HIP kernels (extract_kernels.hip).
// Must be compiled with -ffp-contract=off: the reference's x86-64 build has no FMA.
//
// Reference: loam/include/loam/features-inl.h:53-124, loam/src/features.cpp:20-68,
//            loam/include/loam/common.h:81-86.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LOAMX_EHD __host__ __device__ __forceinline__
#else
#include <math.h>
#define LOAMX_EHD inline
#endif

namespace loamx {

struct ExtractParams {
  uint32_t H, W;         // scan_lines, points_per_line
  uint32_t np;           // neighbor_points
  uint32_t S;            // number_sectors
  uint32_t pps;          // points_per_sector = W / S (features-inl.h:18)
  uint32_t max_edge;     // max_edge_feats_per_sector (up to max+1 are taken)
  uint32_t max_planar;
  uint32_t cap_edge;     // per-sector staging capacity = min(max+1, ceil(sector_len))
  uint32_t cap_planar;
  double min_range, max_range;
  double edge_thr, planar_thr, occ_thr, par_thr;
  uint32_t flags;        // the context's test / measurement switches (loamx_ctx_set_option), never a parameter of the path
};
enum : uint32_t {
  kFlagForceReplay = 1u,     // every scan line takes the tie path (std::sort replay) of the selection kernels
  kFlagForceGiveUp = 2u,     // the fused compaction's chained scan gives up at once (exercises the fallback)
  // host-side routing only (the kernels never look at these)
  kFlagCurvV1 = 4u,          // one-column curvature kernel instead of curvature_valid2_kernel<3>
  kFlagNoFusedCompact = 8u,  // selection writes stage + counts only, compact_kernel gathers
  kFlagNoMisSelect = 16u,    // arg-max fallback select_kernel for every scan
  kFlagFusedExtract = 32u,   // one-pass extract_fused_kernel where the parameters allow
  kFlagNoRowSelect = 64u,    // one scan line per wavefront (select_mis_kernel) instead of four (select_rows_kernel)
  kFlagFusedRows = 128u,     // opt-in: the fused form of select_rows_kernel (curvature + validity inside it) instead of the two kernels
  kFlagStageAlways = 512u,   // the fused selection writes the stage arrays too (one launch; otherwise a second, conditional one does)
  kFlagNoSplitCurv = 256u,   // curvature as doubles + validity bytes between the two kernels even where the split form applies
  kFlagSplitCurv = 1u << 30  // (set by extract_dev, not an option) this call's kernels use the split form: see curvature_valid2_kernel
};

// features-inl.h:66-67 / features.cpp:22: col < np || col >= W - np in size_t arithmetic.
// (For W < np the reference's unsigned W - np wraps and only col < np remains, which is then true
// for every column; col + np >= W gives the same answer without wrapping.)
LOAMX_EHD bool is_line_end(uint32_t col, uint32_t W, uint32_t np) { return col < np || col + np >= W; }

// common.h:81-86: sqrt(x*x + y*y + z*z), left to right
LOAMX_EHD double point_range(double x, double y, double z) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __dsqrt_rn(x * x + y * y + z * z);
#else
  return sqrt(x * x + y * y + z * z);
#endif
}

// features-inl.h:73-82. p points at the xyz of the line's local point 0; i is the local index.
LOAMX_EHD double curvature_at(const double* p, int i, uint32_t np) {
  double dx = -(2.0 * np) * p[3 * i];
  double dy = -(2.0 * np) * p[3 * i + 1];
  double dz = -(2.0 * np) * p[3 * i + 2];
  for (int n = 1; n <= (int)np; n++) {
    dx = dx + p[3 * (i - n)] + p[3 * (i + n)];
    dy = dy + p[3 * (i - n) + 1] + p[3 * (i + n) + 1];
    dz = dz + p[3 * (i - n) + 2] + p[3 * (i + n) + 2];
  }
  return dx * dx + dy * dy + dz * dz;
}

// Which invalidation an interior point triggers (features-inl.h:103-119 with the `continue`
// short-circuit): 0 none, 1 out of range, 2 occlusion case 1, 3 occlusion case 2, 4 parallel.
enum : uint8_t { kCodeNone = 0, kCodeRange = 1, kCodeOcc1 = 2, kCodeOcc2 = 3, kCodeParallel = 4 };
LOAMX_EHD uint8_t point_code(double r_prev, double r, double r_next, const ExtractParams& P) {
  if (r < P.min_range || r > P.max_range) return kCodeRange;      // features.cpp:32
  if (r_next - r > P.occ_thr) return kCodeOcc1;                   // features.cpp:46
  if (r - r_next > P.occ_thr) return kCodeOcc2;                   // features.cpp:49
  const double diff_next = fabs(r_prev - r);                      // features.cpp:60-61
  const double diff_prev = fabs(r_next - r);
  if (diff_next > P.par_thr * r && diff_prev > P.par_thr * r) return kCodeParallel;
  return kCodeNone;
}

// The reference scatters invalidations (mask[idx +- n] = false); none of them reads the mask, so
// the final mask is the AND over all writes and can be gathered per point. code points at the
// line's local code 0; i is the local index of the column `col`. Codes of line-end columns are 0.
//   code 1 at j clears j-np..j+np     (features.cpp:33-37)
//   code 2 at j clears j+1..j+np      (features.cpp:47)
//   code 3 at j clears j-(np-1)..j    (features.cpp:50)
//   code 4 at j clears j              (features.cpp:62)
LOAMX_EHD bool valid_from_codes(const uint8_t* code, int i, uint32_t col, uint32_t W, uint32_t np) {
  if (is_line_end(col, W, np)) return false;
  bool ok = code[i] != kCodeParallel;
  for (int n = -(int)np; n <= (int)np; n++) {
    const int c = (int)col + n;
    if (c < 0 || c >= (int)W) continue;
    const uint8_t k = code[i + n];
    if (k == kCodeRange) ok = false;
    if (k == kCodeOcc1 && n <= -1) ok = false;           // j = col + n, col in j+1..j+np
    if (k == kCodeOcc2 && n >= 0 && n <= (int)np - 1) ok = false;  // col in j-(np-1)..j
  }
  return ok;
}

// Selection order (features-inl.h:38, :138-180): the sector is sorted ascending by curvature; the
// edge pass walks it from the top, the planar pass from the bottom. Ties are broken as a stable
// ascending sort would (lower index first) — identical to std::sort on tie-free input.
//   edge  : a before b  <=>  (c_a, i_a) > (c_b, i_b)
//   planar: a before b  <=>  (c_a, i_a) < (c_b, i_b)
LOAMX_EHD bool edge_before(double ca, int32_t ia, double cb, int32_t ib) { return ca > cb || (ca == cb && ia > ib); }
LOAMX_EHD bool planar_before(double ca, int32_t ia, double cb, int32_t ib) { return ca < cb || (ca == cb && ia < ib); }


/* ------------------------------------------------------------------------------------------------
 * The reference's order on ties: features-inl.h:38 sorts a sector with std::sort, which is not
 * stable, so the walk order among points of EQUAL curvature is whatever libstdc++'s introsort leaves. That order is
 * a deterministic function of the comparison outcomes, so it can be replayed: below is libstdc++'s std::sort
 * (bits/stl_algo.h, bits/stl_heap.h as shipped with GCC 5 - 14: __introsort_loop with depth limit 2 lg n,
 * median-of-three to the front, unguarded partition, heap sort when the depth runs out, threshold 16, final insertion
 * sort) restated on an array of point indices with the comparator curvature(a) < curvature(b). tests/hostcheck runs
 * it against the real std::sort. The kernels call it only for scan lines on which a tie can decide something.
 * `less(a, b)` compares two array ELEMENTS (indices); moves of elements are moves of indices.
 * ---------------------------------------------------------------------------------------------- */
template <typename I, typename Less>
LOAMX_EHD void stl_unguarded_linear_insert(I* a, int last, Less less) {
  const I val = a[last];
  int next = last - 1;
  while (less(val, a[next])) {
    a[last] = a[next];
    last = next;
    --next;
  }
  a[last] = val;
}
template <typename I, typename Less>
LOAMX_EHD void stl_insertion_sort(I* a, int first, int last, Less less) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (less(a[i], a[first])) {
      const I val = a[i];
      for (int j = i; j > first; --j) a[j] = a[j - 1];  // move_backward(first, i, i + 1)
      a[first] = val;
    } else {
      stl_unguarded_linear_insert(a, i, less);
    }
  }
}
// bits/stl_heap.h: __adjust_heap (with the __push_heap at its end); `first` is the heap's base position in a
template <typename I, typename Less>
LOAMX_EHD void stl_adjust_heap(I* a, int first, int hole, int len, I value, Less less) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (less(a[first + child], a[first + child - 1])) child--;
    a[first + hole] = a[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    a[first + hole] = a[first + child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && less(a[first + parent], value)) {
    a[first + hole] = a[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a[first + hole] = value;
}
// __partial_sort(first, last, last) = __heap_select (make_heap only: middle == last) + __sort_heap
template <typename I, typename Less>
LOAMX_EHD void stl_heap_sort(I* a, int first, int last, Less less) {
  const int len = last - first;
  if (len >= 2) {  // __make_heap
    int parent = (len - 2) / 2;
    for (;;) {
      const I value = a[first + parent];
      stl_adjust_heap(a, first, parent, len, value, less);
      if (parent == 0) break;
      parent--;
    }
  }
  while (last - first > 1) {  // __sort_heap: __pop_heap(first, last - 1, last - 1)
    --last;
    const I value = a[last];
    a[last] = a[first];
    stl_adjust_heap(a, first, 0, last - first, value, less);
  }
}
template <typename I, typename Less>
LOAMX_EHD void stl_sort(I* a, int first, int last, Less less) {
  if (first == last) return;
  constexpr int kThreshold = 16;
  int lg = 0;  // std::__lg(n) = floor(log2 n)
  for (int n = last - first; n > 1; n >>= 1) lg++;
  // __introsort_loop: the recursion on [cut, last) becomes an explicit stack (disjoint ranges: the order in which
  // they are finished does not change the result); the stack never holds more than the depth limit
  int stk_first[40], stk_last[40], stk_depth[40], sp = 0;
  stk_first[0] = first, stk_last[0] = last, stk_depth[0] = 2 * lg, sp = 1;
  while (sp > 0) {
    --sp;
    int f = stk_first[sp], l = stk_last[sp], depth = stk_depth[sp];
    while (l - f > kThreshold) {
      if (depth == 0) {
#if defined(LOAMX_STL_SORT_STATS)
        g_heap_sorts++;
#endif
        stl_heap_sort(a, f, l, less);
        break;
      }
      --depth;
      // __unguarded_partition_pivot: __move_median_to_first(f, f + 1, mid, l - 1), then __unguarded_partition(f + 1, l, f)
      const int mid = f + (l - f) / 2, ia = f + 1, ib = mid, ic = l - 1;
      int m;
      if (less(a[ia], a[ib])) {
        if (less(a[ib], a[ic])) m = ib;
        else if (less(a[ia], a[ic])) m = ic;
        else m = ia;
      } else if (less(a[ia], a[ic])) m = ia;
      else if (less(a[ib], a[ic])) m = ic;
      else m = ib;
      {
        const I t = a[f];
        a[f] = a[m], a[m] = t;
      }
      int lo = f + 1, hi = l;
      for (;;) {
        while (less(a[lo], a[f])) ++lo;
        --hi;
        while (less(a[f], a[hi])) --hi;
        if (!(lo < hi)) break;
        const I t = a[lo];
        a[lo] = a[hi], a[hi] = t;
        ++lo;
      }
      if (sp < 40) stk_first[sp] = lo, stk_last[sp] = l, stk_depth[sp] = depth, sp++;  // __introsort_loop(cut, last, depth)
      l = lo;
    }
  }
  // __final_insertion_sort
  if (last - first > kThreshold) {
    stl_insertion_sort(a, first, first + kThreshold, less);
    for (int i = first + kThreshold; i != last; ++i) stl_unguarded_linear_insert(a, i, less);
  } else {
    stl_insertion_sort(a, first, last, less);
  }
}

/* ------------------------------------------------------------------------------------------------
 * Selection as a lexicographically-first maximal independent set on lane bitmasks.
 *
 * The reference's greedy walk over the sorted sector (features-inl.h:138-180) picks a candidate iff
 * no higher-priority candidate within +-(np-1) points was picked before it. Without the per-sector
 * cap that is the lexicographically-first MIS of the "within R = np-1 points" graph, which the
 * classic parallel rule computes exactly: in every round each remaining candidate that beats all of
 * its remaining neighbours is picked and its neighbours are removed. The cap (max+1 picks) then
 * keeps the first max+1 picks in priority order, because a pick never depends on lower priorities.
 *
 * One wavefront owns a scan line; lane l owns the CH = ceil(W/64) consecutive points
 * [l*CH, (l+1)*CH) as bits of a 64-bit mask. A lane's "window" adds R halo bits from each
 * neighbouring lane: window bit t <-> point l*CH - R + t, t in [0, CH + 2R). Needs R <= CH,
 * CH + 2R <= 64.
 * ---------------------------------------------------------------------------------------------- */
LOAMX_EHD uint64_t low_mask(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }

template <int R>
LOAMX_EHD uint64_t mis_window(uint64_t own, uint64_t prev, uint64_t next, int CH) {
  const uint64_t mR = low_mask(R);
  return ((prev >> (CH - R)) & mR) | (own << R) | ((next & mR) << (R + CH));
}

// gt[d-1] bit t: curvature(window point t) > curvature(window point t+d).
// Returns the window bits that beat every remaining neighbour (only the own bits are meaningful).
template <int R, bool EDGE>
LOAMX_EHD uint64_t mis_winners(uint64_t Uw, const uint64_t gt[R]) {
  uint64_t blocked = 0;
#pragma unroll
  for (int d = 1; d <= R; d++) {
    const uint64_t g = gt[d - 1];
    const uint64_t gs = g << d;                      // bit t: c(t-d) > c(t)
    const uint64_t beats_up = EDGE ? g : ~g;         // t is before t+d in the walk order
    const uint64_t beats_dn = EDGE ? ~gs : gs;       // t is before t-d
    blocked |= (Uw >> d) & ~beats_up;
    blocked |= (Uw << d) & ~beats_dn;
  }
  return Uw & ~blocked;
}

// every window bit within R of a set bit of Ww
template <int R>
LOAMX_EHD uint64_t mis_spread(uint64_t Ww) {
  uint64_t r = Ww;
#pragma unroll
  for (int d = 1; d <= R; d++) r |= (Ww << d) | (Ww >> d);
  return r;
}

}  // namespace loamx
