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
