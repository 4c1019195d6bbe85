// reg_math.h — per-thread math of the registration kernels: pose algebra, exact k-NN over the
// uniform grid index, line / plane fits, residual + Jacobian rows, and the per-pair trust-region
// (Levenberg-Marquardt) state machine. Everything is FP64.
//
// The functions are __host__ __device__ so that tests/hostcheck can run the very same code on the
// CPU (serial loops in place of kernels) and compare it with the oracle without a GPU. The product
// only ever calls them from HIP kernels (register_kernels.hip); there is no CPU execution path in
// libloamx.so.
//
// Reference behaviour restated here (paths relative to the reference repo):
//   loam/src/geometry.cpp:10-29, :42-73        Pose3d algebra, fitLine, fitPlane
//   loam/include/loam/geometry-inl.h:21-33     point-to-line / point-to-plane distance
//   loam/src/kdtree.cpp:10-28                  k-NN contract (exact, ascending, strict radius)
//   loam/src/registration.cpp:23-103           association guards
//   loam/include/loam/registration-inl.h:28-77 outer ICF loop, Ceres solve, compose, convergence
// Ceres 2.2.0 / Eigen / nanoflann semantics follow SURVEY.md appendices A-C.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#include <math.h>
#define LOAMX_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define LOAMX_HD inline
#endif

namespace loamx {

constexpr int kMaxK = 16;         // upper bound on num_*_neighbors the kernels keep in registers (instantiated for 5, 8 and 16)
constexpr double kDblMin = 2.2250738585072014e-308;
constexpr double kDblMax = 1.7976931348623157e308;
constexpr double kDblEps = 2.220446049250313e-16;

struct Vec3 {
  double x, y, z;
};
LOAMX_HD Vec3 v3(double x, double y, double z) { return Vec3{x, y, z}; }
LOAMX_HD Vec3 vadd(Vec3 a, Vec3 b) { return Vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
LOAMX_HD Vec3 vsub(Vec3 a, Vec3 b) { return Vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
LOAMX_HD Vec3 vscale(double s, Vec3 a) { return Vec3{s * a.x, s * a.y, s * a.z}; }
LOAMX_HD Vec3 vcross(Vec3 a, Vec3 b) {
  return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
LOAMX_HD double vdot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
LOAMX_HD double vnorm(Vec3 a) { return sqrt(vdot(a, a)); }

// q = (x,y,z,w). Eigen: v + w*(2 u x v) + u x (2 u x v)
LOAMX_HD Vec3 quat_rotate(const double q[4], Vec3 v) {
  const Vec3 u = v3(q[0], q[1], q[2]);
  Vec3 uv = vcross(u, v);
  uv = vadd(uv, uv);
  return vadd(vadd(v, vscale(q[3], uv)), vcross(u, uv));
}
LOAMX_HD void quat_mul(const double a[4], const double b[4], double o[4]) {
  const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  const double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  const double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x, o[1] = y, o[2] = z, o[3] = w;
}
// Pose3d::act (geometry.cpp:21)
LOAMX_HD Vec3 pose_act(const double P[7], Vec3 p) { return vadd(quat_rotate(P, p), v3(P[4], P[5], P[6])); }
// Pose3d::compose (geometry.cpp:16-18): out = a (+) b
LOAMX_HD void pose_compose(const double a[7], const double b[7], double out[7]) {
  double q[4];
  quat_mul(a, b, q);
  const Vec3 t = vadd(v3(a[4], a[5], a[6]), quat_rotate(a, v3(b[4], b[5], b[6])));
  out[0] = q[0], out[1] = q[1], out[2] = q[2], out[3] = q[3];
  out[4] = t.x, out[5] = t.y, out[6] = t.z;
}
// rotation.angularDistance(Identity) (registration-inl.h:68): 2*atan2(|vec|, |w|)
LOAMX_HD double quat_angle_to_identity(const double q[4]) {
  return 2.0 * atan2(sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]), fabs(q[3]));
}

/* ------------------------------------------------------------------------------------------------
 * Uniform grid index over one target feature set (replaces the nanoflann KD-tree of kdtree.h:24-41;
 * same contract: exact k-NN). Points are stored cell by cell (x fastest), cell_start has
 * nx*ny*nz + 1 entries.
 * ---------------------------------------------------------------------------------------------- */
struct GridDesc {
  double ox, oy, oz;  // origin = bbox min
  double h, inv_h;    // cell edge
  int32_t nx, ny, nz;
  uint32_t n_points;
};

LOAMX_HD int32_t grid_cell_coord(double v, double origin, double inv_h) {
  double c = floor((v - origin) * inv_h);
  if (c < -1048576.0) c = -1048576.0;
  if (c > 1048576.0) c = 1048576.0;
  return (int32_t)c;
}
LOAMX_HD int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
// cell_start[i] through a 32-bit byte offset from the (wave-uniform) table base
LOAMX_HD uint32_t cell_start_at(const uint32_t* __restrict__ cell_start, uint32_t i) {
  return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(cell_start) + (uint32_t)(i << 2));
}
struct alignas(4) CellStart4 {
  uint32_t v[4];
};
LOAMX_HD CellStart4 cell_start4_at(const uint32_t* __restrict__ cell_start, uint32_t i) {
  return *reinterpret_cast<const CellStart4*>(reinterpret_cast<const char*>(cell_start) + (uint32_t)(i << 2));
}
// cell of a target point at build time (always inside the grid)
LOAMX_HD uint32_t grid_cell_of_point(const GridDesc& g, Vec3 p) {
  const int32_t ix = clampi(grid_cell_coord(p.x, g.ox, g.inv_h), 0, g.nx - 1);
  const int32_t iy = clampi(grid_cell_coord(p.y, g.oy, g.inv_h), 0, g.ny - 1);
  const int32_t iz = clampi(grid_cell_coord(p.z, g.oz, g.inv_h), 0, g.nz - 1);
  return (uint32_t)((iz * g.ny + iy) * g.nx + ix);
}

#ifndef LOAMX_GRID_LDS_CELLS
#define LOAMX_GRID_LDS_CELLS 32768
#endif
constexpr uint32_t kGridLdsCells = LOAMX_GRID_LDS_CELLS;  // cells counted per pass of grid_build_kernel (LDS table)
constexpr uint32_t kGridCellsCap = 65536;  // cells of one target grid (two build passes)
constexpr uint32_t kGridPad = 4;           // spare GridPoint entries after every sorted set (unclamped 4-wide candidate loads)
constexpr float kRelPad = 3.0e38f;         // their single-precision offsets: farther than any point (knn_scan_batch_f32<.., false>)

// Chooses the cell edge and grid dimensions for a target set with bounding box [lo, hi].
// Cell edge: a quarter of the search radius (measured on 64x1024 feature sets: R/4 -> 48 candidates
// + 3.6 row look-ups per query, R/6 -> 30 + 6.9; same kernel time, R/6 needs a second build pass),
// shrunk for dense sets (aim <= ~8 points per occupied
// cell assuming surface-like data), then grown until the table fits cells_cap.
LOAMX_HD void grid_choose(GridDesc& g, Vec3 lo, Vec3 hi, uint32_t n, double max_dist, uint32_t cells_cap) {
  g.n_points = n;
  if (n == 0) {
    g.ox = g.oy = g.oz = 0.0;
    g.h = 1.0, g.inv_h = 1.0;
    g.nx = g.ny = g.nz = 1;
    return;
  }
  const double ex = hi.x - lo.x, ey = hi.y - lo.y, ez = hi.z - lo.z;
  const double area = 2.0 * (ex * ey + ey * ez + ex * ez);
#ifndef LOAMX_GRID_DIV
#define LOAMX_GRID_DIV 4.0
#endif
  double h = max_dist > 0.0 ? max_dist / LOAMX_GRID_DIV : 0.0;
  const double h_dense = sqrt(8.0 * area / (double)n);
  if (h_dense > 0.0 && (h <= 0.0 || h_dense < h)) h = h_dense;
  if (!(h > 0.0)) h = 1.0;
  if (max_dist > 0.0 && h < 1e-3 * max_dist) h = 1e-3 * max_dist;
  // sparse sets: no more than ~16 cells per point (a table far larger than the set costs build time and
  // sends most queries through several rounds of empty cells)
  {
    const double sparse_cap = 16.0 * (double)n > 4096.0 ? 16.0 * (double)n : 4096.0;
    if (sparse_cap < (double)cells_cap) cells_cap = (uint32_t)sparse_cap;
  }
  int32_t nx = 1, ny = 1, nz = 1;
  for (int it = 0; it < 400; it++) {
    const double fx = floor(ex / h), fy = floor(ey / h), fz = floor(ez / h);
    if (fx < 1e6 && fy < 1e6 && fz < 1e6) {
      nx = (int32_t)fx + 1, ny = (int32_t)fy + 1, nz = (int32_t)fz + 1;
      if ((double)nx * (double)ny * (double)nz <= (double)cells_cap) break;
    }
    h *= 1.1;
  }
  g.ox = lo.x, g.oy = lo.y, g.oz = lo.z;
  g.h = h, g.inv_h = 1.0 / h;
  g.nx = nx, g.ny = ny, g.nz = nz;
}

// Source feature sets are only *ordered* by a grid (never searched): a 2^L x 2^L x 2^L grid over the
// bounding box, cells numbered along the Morton (Z-order) curve, so that any run of consecutive
// points is spatially compact at every scale. L = 5 (15-bit codes fill the 32 768-entry LDS table
// exactly) except for small sets, where fewer cells keep the table work proportional to the set.
LOAMX_HD void grid_choose_morton(GridDesc& g, Vec3 lo, Vec3 hi, uint32_t n) {
  g.n_points = n;
  const int level = n <= 128u ? 2 : (n <= 1024u ? 3 : (n <= 4096u ? 4 : 5));
  const int32_t dim = 1 << level;
  double ext = hi.x - lo.x;
  if (hi.y - lo.y > ext) ext = hi.y - lo.y;
  if (hi.z - lo.z > ext) ext = hi.z - lo.z;
  double h = ext / (double)dim * (1.0 + 1e-9);
  if (!(h > 0.0) || n == 0) h = 1.0;
  g.ox = n ? lo.x : 0.0, g.oy = n ? lo.y : 0.0, g.oz = n ? lo.z : 0.0;
  g.h = h, g.inv_h = 1.0 / h;
  g.nx = g.ny = g.nz = dim;
}
LOAMX_HD uint32_t morton_spread5(uint32_t v) {  // abcde -> a00b00c00d00e
  v &= 31u;
  v = (v | (v << 8)) & 0x100Fu;
  v = (v | (v << 4)) & 0x10C3u;
  v = (v | (v << 2)) & 0x1249u;
  return v;
}
LOAMX_HD uint32_t grid_morton_of_point(const GridDesc& g, Vec3 p) {
  const uint32_t ix = (uint32_t)clampi(grid_cell_coord(p.x, g.ox, g.inv_h), 0, g.nx - 1);
  const uint32_t iy = (uint32_t)clampi(grid_cell_coord(p.y, g.oy, g.inv_h), 0, g.nx - 1);
  const uint32_t iz = (uint32_t)clampi(grid_cell_coord(p.z, g.oz, g.inv_h), 0, g.nx - 1);
  return morton_spread5(ix) | (morton_spread5(iy) << 1) | (morton_spread5(iz) << 2);
}

// One indexed target point: 32 bytes so a candidate is two 16-byte loads.
struct alignas(32) GridPoint {
  double x, y, z;
  uint32_t orig;  // index in the caller's array
  uint32_t pad;
};

// k best neighbours, ascending by (squared distance, original index): a strict total order, so
// the result does not depend on the storage order inside a cell.
template <int KM>
struct KnnResult {
  double d2[KM];
  uint32_t pos[KM];   // position in the cell-sorted array
  uint32_t orig[KM];  // index in the caller's target array
  double worst;          // d2 of the k-th best once k are held, else DBL_MAX
  int count;
};

// Branch-free insertion: the list is sorted ascending by (d2, orig) and padded with
// (DBL_MAX, 0xFFFFFFFF); the newcomer is pushed down a chain of compare-exchanges. `count` is not
// maintained here (knn_finish counts the non-padding entries).
template <int KM>
LOAMX_HD void knn_insert(KnnResult<KM>& r, int k, double d2, uint32_t pos, uint32_t orig) {
#pragma unroll
  for (int j = 0; j < KM; j++) {
    const bool lt = d2 < r.d2[j] || (d2 == r.d2[j] && orig < r.orig[j]);
    const double td = r.d2[j];
    const uint32_t tp = r.pos[j], to = r.orig[j];
    r.d2[j] = lt ? d2 : td;
    r.pos[j] = lt ? pos : tp;
    r.orig[j] = lt ? orig : to;
    d2 = lt ? td : d2;
    pos = lt ? tp : pos;
    orig = lt ? to : orig;
  }
  double w = r.d2[KM - 1];
#pragma unroll
  for (int j = 0; j < KM - 1; j++)
    if (j == k - 1) w = r.d2[j];
  r.worst = w;  // DBL_MAX until k points have been seen
}

// upper bound on the k-th smallest squared distance seen so far (DBL_MAX until k were seen)
template <int KM>
LOAMX_HD double knn_bound(const KnnResult<KM>& r, int k) {
  (void)k;
  return r.worst;
}

// four candidates (the first min(n, 4) are real). Measured: four straight-line insert sites beat one
// rolled insert loop (7.3 vs 9.1 ms per launch).
template <int KM>
LOAMX_HD void knn_offer4(KnnResult<KM>& r, int k, double d0, double d1, double d2, double d3, uint32_t p, uint32_t n,
                         uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3) {
  if (d0 <= r.worst) knn_insert(r, k, d0, p, o0);
  if (n > 1 && d1 <= r.worst) knn_insert(r, k, d1, p + 1, o1);
  if (n > 2 && d2 <= r.worst) knn_insert(r, k, d2, p + 2, o2);
  if (n > 3 && d3 <= r.worst) knn_insert(r, k, d3, p + 3, o3);
}

/* Keyed collector — the fast path of the search.
 *
 * A candidate is folded into ONE double: the bit pattern of its squared distance with the low
 * `bits` mantissa bits replaced by its position in the cell-sorted array (bits = just enough for
 * the set). For non-negative doubles the numeric order equals the order of the bit patterns, so
 * keys order candidates by (truncated d2, position) and the KM+1 smallest keys are maintained with a
 * chain of v_min_f64 / v_max_f64 pairs: 2 instructions per slot instead of the ~11 of the exact
 * (d2, orig) compare-exchange above.
 *
 * Truncation is monotone, so trunc(a) < trunc(b) implies a < b. The result is therefore exact
 * whenever the truncated distances of slots 0..k (k neighbours plus the best rejected candidate) are
 * pairwise different: then slots 0..k-1 are the k nearest in exactly the (d2, orig) order. Otherwise
 * (two candidates within ~2^-(52-bits) relative of each other: exact ties in structured data, or
 * once in ~10^9 queries on noisy data), or when the radius filter cannot be decided from a truncated
 * distance, knn_keys_finish reports "undecided" and the caller repeats the query with the exact
 * collector. Results are thus always those of the exact search. */
template <int KM>
struct KnnKeys {
  double key[KM + 1];  // ascending: KM-k sentinels (-1), then the k+1 smallest keys; +inf = empty
  uint32_t mask;       // low-word mask of the position bits
};

LOAMX_HD double knn_key_empty() { return __builtin_huge_val(); }
LOAMX_HD uint32_t knn_key_lo(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__double2loint(v);
#else
  uint64_t b;
  __builtin_memcpy(&b, &v, 8);
  return (uint32_t)b;
#endif
}
LOAMX_HD uint32_t knn_key_hi(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__double2hiint(v);
#else
  uint64_t b;
  __builtin_memcpy(&b, &v, 8);
  return (uint32_t)(b >> 32);
#endif
}
LOAMX_HD double knn_key_join(uint32_t hi, uint32_t lo) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __hiloint2double((int)hi, (int)lo);
#else
  const uint64_t b = ((uint64_t)hi << 32) | lo;
  double v;
  __builtin_memcpy(&v, &b, 8);
  return v;
#endif
}
// candidates whose distance is not a finite number (NaN / inf coordinates) never enter the list,
// exactly as `d <= worst` keeps them out of the exact collector
LOAMX_HD double knn_key_pack(double d2, uint32_t pos, uint32_t mask, bool real) {
  const double key = knn_key_join(knn_key_hi(d2), (knn_key_lo(d2) & ~mask) | pos);
  return (real && d2 <= kDblMax) ? key : knn_key_empty();
}
LOAMX_HD uint32_t knn_key_mask(uint32_t n_points) {
  if (n_points <= 2u) return 1u;
  return 0xFFFFFFFFu >> __builtin_clz(n_points - 1u);
}
// min / max of two keys. Keys are never NaN, so the plain machine instructions are exact; inline
// asm keeps the compiler from wrapping every operand in a canonicalising v_max_f64 x, x (which the
// IEEE-mode lowering of fmin / fmax requires for possible signalling NaNs).
LOAMX_HD double knn_key_min(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return a < b ? a : b;
#endif
}
LOAMX_HD double knn_key_max(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return a < b ? b : a;
#endif
}
template <int KM>
LOAMX_HD void knn_key_insert(KnnKeys<KM>& c, double key) {
#pragma unroll
  for (int j = 0; j < KM; j++) {
    const double s = c.key[j];
    c.key[j] = knn_key_min(s, key);
    key = knn_key_max(s, key);
  }
  c.key[KM] = knn_key_min(c.key[KM], key);
}
// The k-th smallest real key sits in slot KM-1 and the best rejected one in slot KM, whatever k is:
// the first KM-k slots hold the sentinel -1 (smaller than every key), see knn_init.
template <int KM>
LOAMX_HD double knn_bound(const KnnKeys<KM>& c, int k) {
  (void)k;
  const double kth = c.key[KM - 1];
  // the true d2 of every one of the k smallest keys is <= its key with all position bits set
  return kth < knn_key_empty() ? knn_key_join(knn_key_hi(kth), knn_key_lo(kth) | c.mask) : kDblMax;
}
template <int KM>
LOAMX_HD void knn_offer4(KnnKeys<KM>& c, int k, double d0, double d1, double d2, double d3, uint32_t p, uint32_t n,
                         uint32_t, uint32_t, uint32_t, uint32_t) {
  (void)k;
  knn_key_insert(c, knn_key_pack(d0, p, c.mask, true));
  knn_key_insert(c, knn_key_pack(d1, p + 1, c.mask, n > 1));
  knn_key_insert(c, knn_key_pack(d2, p + 2, c.mask, n > 2));
  knn_key_insert(c, knn_key_pack(d3, p + 3, c.mask, n > 3));
}

// Candidates are fetched four at a time (four independent 32-byte loads in flight per lane).
// The array is allocated with kGridPad spare entries, so the four loads never need clamping;
// entries at or beyond the range end belong to the next cell (or are padding) and are masked by n
// (= entries left in the range, may exceed 4).
template <class Coll>
LOAMX_HD void knn_scan_batch(Coll& r, int k, Vec3 q, const GridPoint* __restrict__ sp, uint32_t p, uint32_t n) {
  // 32-bit byte offset from the (wave-uniform) set base: one VGPR of address per lane, the +32 / +64
  // / +96 of the neighbours go into the instruction's immediate offset. Sets stay below 2^26 points.
  const char* __restrict__ q0 = reinterpret_cast<const char*>(sp) + (uint32_t)(p << 5);
  const GridPoint t0 = *reinterpret_cast<const GridPoint*>(q0);
  const GridPoint t1 = *reinterpret_cast<const GridPoint*>(q0 + 32);
  const GridPoint t2 = *reinterpret_cast<const GridPoint*>(q0 + 64);
  const GridPoint t3 = *reinterpret_cast<const GridPoint*>(q0 + 96);
#if defined(LOAMX_KNN_STATS)
  g_cand += n < 4u ? n : 4u;
#endif
  // nanoflann L2_Simple: ((dx^2 + dy^2) + dz^2)
  double dx = q.x - t0.x, dy = q.y - t0.y, dz = q.z - t0.z;
  const double d0 = dx * dx + dy * dy + dz * dz;
  dx = q.x - t1.x, dy = q.y - t1.y, dz = q.z - t1.z;
  const double d1 = dx * dx + dy * dy + dz * dz;
  dx = q.x - t2.x, dy = q.y - t2.y, dz = q.z - t2.z;
  const double d2 = dx * dx + dy * dy + dz * dz;
  dx = q.x - t3.x, dy = q.y - t3.y, dz = q.z - t3.z;
  const double d3 = dx * dx + dy * dy + dz * dz;
  knn_offer4(r, k, d0, d1, d2, d3, p, n, t0.orig, t1.orig, t2.orig, t3.orig);
}

template <class Coll>
LOAMX_HD void knn_scan_range(Coll& r, int k, Vec3 q, const GridPoint* __restrict__ sp, uint32_t begin, uint32_t end) {
  for (uint32_t p = begin; p < end; p += 4) knn_scan_batch(r, k, q, sp, p, end - p);
}

// distance from coordinate v to the slab of cell index c along one axis, shrunk by a safety margin
// that covers the rounding of grid_cell_coord (never over-estimates the true distance)
LOAMX_HD double slab_dist(double v, double origin, double h, int32_t c) {
  const double lo = origin + (double)c * h, hi = lo + h;
  double d = v < lo ? lo - v : (v > hi ? v - hi : 0.0);
  d -= 1e-9 * h;
  return d > 0.0 ? d : 0.0;
}

template <int KM>
LOAMX_HD void knn_init(KnnResult<KM>& r) {
  r.count = 0;
  r.worst = kDblMax;
#pragma unroll
  for (int j = 0; j < KM; j++) {
    r.d2[j] = kDblMax;
    r.pos[j] = 0;
    r.orig[j] = 0xFFFFFFFFu;
  }
}
template <int KM>
LOAMX_HD void knn_init(KnnKeys<KM>& c, int k, uint32_t n_points) {
  c.mask = knn_key_mask(n_points);
#pragma unroll
  for (int j = 0; j <= KM; j++) c.key[j] = j < KM - k ? -1.0 : knn_key_empty();
}

// Chebyshev distance (in cells) from the query cell to the grid box
LOAMX_HD int32_t grid_outside_distance(const GridDesc& g, int32_t cx, int32_t cy, int32_t cz) {
  const int32_t ex = cx < 0 ? -cx : (cx > g.nx - 1 ? cx - (g.nx - 1) : 0);
  const int32_t ey = cy < 0 ? -cy : (cy > g.ny - 1 ? cy - (g.ny - 1) : 0);
  const int32_t ez = cz < 0 ? -cz : (cz > g.nz - 1 ? cz - (g.nz - 1) : 0);
  int32_t out = ex > ey ? ex : ey;
  return out > ez ? out : ez;
}

// The strict radius filter of kdtree.cpp:25 is sqrt(d2) < max_dist with a correctly rounded sqrt,
// which is monotone: there is a largest double d2 that passes. The host computes it once per call
// (RegConfig) so that the keyed collector decides the filter with two compares and no sqrt.
#if defined(__HIPCC__)
__host__
#endif
inline double knn_radius_pass_max(double max_dist) {
  if (!(max_dist > 0.0)) return kDblMax;  // filter disabled
  double x = max_dist * max_dist;
  if (!(x <= kDblMax)) return kDblMax;
  while (x > 0.0 && !(sqrt(x) < max_dist)) x = nextafter(x, 0.0);
  while (x < kDblMax && sqrt(nextafter(x, kDblMax)) < max_dist) x = nextafter(x, kDblMax);
  return sqrt(x) < max_dist ? x : -1.0;  // -1: nothing passes (max_dist below sqrt of the smallest double)
}

// radius^2 bound used for pruning: points at distance >= max_dist are dropped by the radius filter anyway
LOAMX_HD double knn_radius_bound(double max_dist) { return max_dist > 0.0 ? max_dist * max_dist * (1.0 + 1e-12) : kDblMax; }

// The rounds of the search. Cubes of cells of growing half-width w around the query cell are
// visited; after round w every unvisited point is farther than w*h along some axis.

// The per-thread list of the non-empty rows of the 3x3x3 block around cell (cx, cy, cz), in the order
// centre, faces, corners: entry = {begin | row code << 28, end}. The 18 cell_start entries are fetched
// up front (independent loads instead of nine dependent round trips). Returns the number of entries.
LOAMX_HD int knn_round1_list(const GridDesc& g, const uint32_t* __restrict__ cell_start, int32_t cx, int32_t cy, int32_t cz,
                             uint32_t* row_scratch, int row_stride, uint32_t* population = nullptr) {
  uint32_t rb[9], re[9];
  const int32_t xa = cx - 1 < 0 ? 0 : cx - 1, xb = cx + 1 > g.nx - 1 ? g.nx - 1 : cx + 1;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    const int32_t iy = cy + (j % 3) - 1, iz = cz + (j / 3) - 1;
    const bool ok = xa <= xb && iy >= 0 && iy <= g.ny - 1 && iz >= 0 && iz <= g.nz - 1;
    const uint32_t row = ok ? (uint32_t)((iz * g.ny + iy) * g.nx) : 0u;
    rb[j] = ok ? cell_start_at(cell_start, row + (uint32_t)xa) : 0u;
    re[j] = ok ? cell_start_at(cell_start, row + (uint32_t)xb + 1u) : 0u;
  }
  int nrow = 0;
  uint32_t pop = 0;
#pragma unroll
  for (int o = 0; o < 9; o++) {
    constexpr int kOrder[9] = {4, 1, 3, 5, 7, 0, 2, 6, 8};
    const int j = kOrder[o];
    if (rb[j] < re[j]) {
      row_scratch[(2 * nrow) * row_stride] = rb[j] | ((uint32_t)j << 28);  // set sizes stay below 2^28
      row_scratch[(2 * nrow + 1) * row_stride] = re[j];
      nrow++;
      pop += re[j] - rb[j];
    }
  }
  if (population) *population = pop;
  return nrow;
}

// Round 1 when it starts at the query's own cell: the 3x3x3 block, as one flattened candidate stream.
template <class Coll>
LOAMX_HD void knn_round1(const GridDesc& g, const uint32_t* __restrict__ cell_start, const GridPoint* __restrict__ sp, Vec3 q,
                         int k, double r2, int32_t cx, int32_t cy, int32_t cz, Coll& r, uint32_t* row_scratch,
                         int row_stride) {
  // Common case: the 3x3x3 block. The cell_start entries of its nine rows are fetched up front
  // (18 independent loads instead of nine dependent round trips). The non-empty rows are written
  // to a small per-thread list (LDS in the kernels) in the order centre, faces, corners, and the
  // lane then walks ONE flattened stream of 4-wide candidate batches over that list, skipping a
  // row when its slab is already farther than the current bound. Flattening matters on the GPU:
  // a wavefront then runs for max_lanes(sum of batches) instead of sum_rows(max_lanes(batches))
  // (measured on the 64x1024 workload: 20.4 vs 37.0 batch steps per wavefront).
  const int nrow = knn_round1_list(g, cell_start, cx, cy, cz, row_scratch, row_stride);
  // squared slab distances to the neighbouring rows (the query's own row is at distance 0)
  double sy2m = slab_dist(q.y, g.oy, g.h, cy - 1), sy2p = slab_dist(q.y, g.oy, g.h, cy + 1);
  double sz2m = slab_dist(q.z, g.oz, g.h, cz - 1), sz2p = slab_dist(q.z, g.oz, g.h, cz + 1);
  sy2m *= sy2m, sy2p *= sy2p, sz2m *= sz2m, sz2p *= sz2p;
  uint32_t p = 0, e = 0;
  int ri = 0;
  for (;;) {
    while (p >= e && ri < nrow) {  // next admissible row
      const uint32_t bj = row_scratch[(2 * ri) * row_stride], e2 = row_scratch[(2 * ri + 1) * row_stride];
      ri++;
      const int j = (int)(bj >> 28), jy = j % 3, jz = j / 3;
      const double sy2 = jy == 0 ? sy2m : (jy == 1 ? 0.0 : sy2p), sz2 = jz == 0 ? sz2m : (jz == 1 ? 0.0 : sz2p);
      const double worst = knn_bound(r, k);
      const double bound = worst < r2 ? worst : r2;
      if (sy2 + sz2 <= bound) {
#if defined(LOAMX_KNN_STATS)
        g_rows++;
#endif
        p = bj & 0x0FFFFFFFu, e = e2;
      }
    }
    if (p >= e) break;
    knn_scan_batch(r, k, q, sp, p, e - p);
    p += 4;
  }
}

// Round w > 1 (or a first round that starts away from the query's cell): cells at Chebyshev distance
// <= w (first) or == w (later rounds). The row pieces are taken nine at a time: their extents are
// computed from the current bound, the eighteen cell_start entries are fetched together, and the
// candidates of all nine run through one flattened stream — three memory round trips per nine pieces
// instead of two per piece.
#ifndef LOAMX_ROUND_CHUNK
#define LOAMX_ROUND_CHUNK 9
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define LOAMX_WAVE_ANY(x) (__any((int)(x)) != 0)
#else
#define LOAMX_WAVE_ANY(x) (x)
#endif
constexpr int kRoundChunk = LOAMX_ROUND_CHUNK;  // <= 9 (row_scratch holds 18 words per thread)
template <class Coll>
LOAMX_HD void knn_general_round(const GridDesc& g, const uint32_t* __restrict__ cell_start,
                                const GridPoint* __restrict__ sp, Vec3 q, int k, double r2, int32_t cx, int32_t cy,
                                int32_t cz, Coll& r, int32_t w, bool first, uint32_t* row_scratch, int row_stride) {
#if defined(LOAMX_KNN_STATS)
  g_general++;
#endif
  // rows of the cube that exist: a query far outside the grid (w = thousands of cells) must not walk (2 w + 1)^2 row
  // positions to find the handful inside it
  const int32_t dy_lo = -w > -cy ? -w : -cy, dy_hi = w < g.ny - 1 - cy ? w : g.ny - 1 - cy;
  const int32_t dz_lo = -w > -cz ? -w : -cz, dz_hi = w < g.nz - 1 - cz ? w : g.nz - 1 - cz;
  int32_t dz = dz_lo, dy = dy_lo;
  int sg = 0;
  bool more = dy_lo <= dy_hi && dz_lo <= dz_hi;
  // squared slab distances of the current row (dy, dz) and of the two end cells cx -+ w: recomputed only
  // when the cursor moves to another row, so that a row out of reach costs a handful of instructions
  double sz2 = slab_dist(q.z, g.oz, g.h, cz + dz), sy2 = slab_dist(q.y, g.oy, g.h, cy + dy);
  sz2 *= sz2, sy2 *= sy2;
  double sx2m = slab_dist(q.x, g.ox, g.h, cx - w), sx2p = slab_dist(q.x, g.ox, g.h, cx + w);
  sx2m *= sx2m, sx2p *= sx2p;
  while (more) {
    // ---- collect up to kRoundChunk admissible pieces
    int n = 0;
    const double worst = knn_bound(r, k);
    const double bound = worst < r2 ? worst : r2;
    // (the cursor advances in lock-step for the whole wavefront: the chunk ends as soon as any lane's
    // list is full, so dy / dz / sg stay wave-uniform and no lane waits for another's gaps)
    while (more && !LOAMX_WAVE_ANY(n >= kRoundChunk)) {
      const int32_t ady = dy < 0 ? -dy : dy, adz = dz < 0 ? -dz : dz;
      const int nseg = (first || ady == w || adz == w) ? 1 : 2;  // full row, or just its two end cells
      const int32_t iy = cy + dy, iz = cz + dz;
      const double rowmin2 = sy2 + sz2;
      if (iy >= 0 && iy <= g.ny - 1 && iz >= 0 && iz <= g.nz - 1 && rowmin2 <= bound) {
        int32_t xlo = (nseg == 2 && sg == 1) ? cx + w : cx - w;
        int32_t xhi = (nseg == 2 && sg == 0) ? cx - w : cx + w;
        bool ok = true;
        if (nseg == 2) {
          ok = rowmin2 + (sg == 0 ? sx2m : sx2p) <= bound && xlo >= 0 && xlo <= g.nx - 1;
        } else {
          if (xlo < 0) xlo = 0;
          if (xhi > g.nx - 1) xhi = g.nx - 1;
          if (bound < kDblMax) {
            // cells whose x slab is farther than sqrt(bound - rowmin2) cannot contribute
            const double reach = sqrt(bound - rowmin2) + 1e-9 * g.h;
            const int32_t xl = grid_cell_coord(q.x - reach, g.ox, g.inv_h), xh = grid_cell_coord(q.x + reach, g.ox, g.inv_h);
            if (xl > xlo) xlo = xl;
            if (xh < xhi) xhi = xh;
          }
          ok = xlo <= xhi;
        }
        if (ok) {
          const uint32_t row = (uint32_t)((iz * g.ny + iy) * g.nx);
          row_scratch[(2 * n) * row_stride] = row + (uint32_t)xlo;
          row_scratch[(2 * n + 1) * row_stride] = row + (uint32_t)xhi + 1u;
          n++;
        }
      }
      if (++sg >= nseg) {
        sg = 0;
        if (++dy > dy_hi) {
          dy = dy_lo;
          if (++dz > dz_hi) more = false;
          sz2 = slab_dist(q.z, g.oz, g.h, cz + dz);
          sz2 *= sz2;
        }
        sy2 = slab_dist(q.y, g.oy, g.h, cy + dy);
        sy2 *= sy2;
      }
    }
    // ---- their cell_start entries, all in flight together
    uint32_t pb[kRoundChunk], pe[kRoundChunk];
#pragma unroll
    for (int j = 0; j < kRoundChunk; j++) {
      const bool use = j < n;
      const uint32_t a = use ? row_scratch[(2 * j) * row_stride] : 0u, b = use ? row_scratch[(2 * j + 1) * row_stride] : 0u;
      pb[j] = use ? cell_start_at(cell_start, a) : 0u;
      pe[j] = use ? cell_start_at(cell_start, b) : 0u;
    }
#pragma unroll
    for (int j = 0; j < kRoundChunk; j++) {
      row_scratch[(2 * j) * row_stride] = pb[j];
      row_scratch[(2 * j + 1) * row_stride] = pe[j];
    }
    // ---- one flattened candidate stream over the pieces
    uint32_t p = 0, e = 0;
    int ri = 0;
    for (;;) {
      while (p >= e && ri < n) {
        p = row_scratch[(2 * ri) * row_stride], e = row_scratch[(2 * ri + 1) * row_stride];
        ri++;
#if defined(LOAMX_KNN_STATS)
        if (p < e) g_rows++;
#endif
      }
      if (p >= e) break;
      knn_scan_batch(r, k, q, sp, p, e - p);
      p += 4;
    }
  }
}

// Is the search over after the cube of half-width w?
template <class Coll>
LOAMX_HD bool knn_done(const GridDesc& g, Vec3 q, int k, double max_dist, int32_t cx, int32_t cy, int32_t cz, const Coll& r,
                       int32_t w) {
  // Every unvisited point lies beyond one of the six faces of the visited block of cells
  // [c-w, c+w]^3, so it is at least as far as the nearest face that still has grid cells behind it
  // (>= w*h; on average ~1.25*h for w = 1, which ends the search after the first round more often
  // than the plain w*h bound). A face on the grid boundary has nothing behind it.
  double guard = kDblMax;
  {
    const double m = 1e-9 * g.h;
    const int32_t c3[3] = {cx, cy, cz}, n3[3] = {g.nx, g.ny, g.nz};
    const double q3[3] = {q.x, q.y, q.z}, o3[3] = {g.ox, g.oy, g.oz};
#pragma unroll
    for (int a = 0; a < 3; a++) {
      if (c3[a] - w > 0) {
        const double d = (q3[a] - (o3[a] + (double)(c3[a] - w) * g.h)) * (1.0 - 1e-9) - m;
        guard = d < guard ? d : guard;
      }
      if (c3[a] + w < n3[a] - 1) {
        const double d = ((o3[a] + (double)(c3[a] + w + 1) * g.h) - q3[a]) * (1.0 - 1e-9) - m;
        guard = d < guard ? d : guard;
      }
    }
    if (guard < 0.0) guard = 0.0;
  }
  if (guard == kDblMax) return true;                       // the block covers the whole grid
  if (knn_bound(r, k) < guard * guard) return true;        // k found (bound is DBL_MAX otherwise), all closer than anything unvisited
  if (max_dist > 0.0 && guard >= max_dist) return true;    // anything unvisited fails the radius filter
  return false;
}

template <class Coll>
LOAMX_HD void knn_rounds(const GridDesc& g, const uint32_t* __restrict__ cell_start, const GridPoint* __restrict__ sp, Vec3 q,
                         int k, double max_dist, int32_t cx, int32_t cy, int32_t cz, Coll& r, int32_t w,
                         uint32_t* row_scratch, int row_stride) {
  const double r2 = knn_radius_bound(max_dist);
  bool first = true;
  for (;;) {
    if (first && w == 1) knn_round1(g, cell_start, sp, q, k, r2, cx, cy, cz, r, row_scratch, row_stride);
    else knn_general_round(g, cell_start, sp, q, k, r2, cx, cy, cz, r, w, first, row_scratch, row_stride);
    first = false;
    if (knn_done(g, q, k, max_dist, cx, cy, cz, r, w)) break;
    w++;
  }
}

// pos[(KM - k) + j] = src[j] for j < k (the shifted layout of the keyed collectors), zero elsewhere — with STATIC register
// indices only. The obvious form, `if (j < k) pos[(KM - k) + j] = src[j]` in an unrolled loop, is compiled by hipcc 7.2
// into an UNCONDITIONAL indexed register write (s_set_gpr_idx_on ... v_mov) followed by a select: in the iterations the
// guard excludes, the index lies beyond the array and the write lands in whatever registers follow it. Found in round 3
// when KM = 16, k = 13 overwrote the thread's own query index (a GPU memory fault); with KM = 5 / 8 the registers behind
// the array happened to be dead.
template <int KM>
LOAMX_HD void knn_shift_positions(const uint32_t src[KM], int k, uint32_t pos[KM]) {
#pragma unroll
  for (int u = 0; u < KM; u++) {
    uint32_t v = 0u;
#pragma unroll
    for (int t = 0; t < KM; t++)
      if (t + (KM - k) == u) v = src[t];
    pos[u] = v;
  }
}

// strict radius filter of kdtree.cpp:25 (max_dist <= 0 disables it): number of neighbours kept (prefix of r)
template <int KM>
LOAMX_HD int knn_finish(KnnResult<KM>& r, int k, double max_dist) {
  int count = 0, kept = 0;
#pragma unroll
  for (int j = 0; j < KM; j++) {
    if (j < k && r.orig[j] != 0xFFFFFFFFu) count = j + 1;
  }
  r.count = count;
#pragma unroll
  for (int j = 0; j < KM; j++) {
    if (j < count && kept == j && (max_dist <= 0.0 || sqrt(r.d2[j]) < max_dist)) kept = j + 1;
  }
  return kept;
}

// Keyed collector: number of neighbours kept (a prefix of the real slots, ascending), or -1 =
// undecided (see the KnnKeys comment): two of the k+1 real slots share a truncated distance, or a
// truncated distance straddles the radius. pos[i] is the position held by slot i; neighbour j of the
// result is slot (KM - k) + j.
template <int KM>
LOAMX_HD int knn_keys_finish(const KnnKeys<KM>& c, int k, double pass_max, uint32_t pos[KM]) {
  int kept = 0;
  bool undecided = false, open = true;
#pragma unroll
  for (int i = 0; i < KM; i++) {
    const double a = c.key[i], b = c.key[i + 1];
    const bool real = i >= KM - k && a < knn_key_empty();
    pos[i] = knn_key_lo(a) & c.mask;
    if (real && b < knn_key_empty() && knn_key_hi(a) == knn_key_hi(b) && ((knn_key_lo(a) ^ knn_key_lo(b)) & ~c.mask) == 0u)
      undecided = true;
    if (real && open) {
      const double lo = knn_key_join(knn_key_hi(a), knn_key_lo(a) & ~c.mask);  // lo <= d2 <= hi
      const double hi = knn_key_join(knn_key_hi(a), knn_key_lo(a) | c.mask);
      if (hi <= pass_max) {  // pass_max = knn_radius_pass_max(max_dist): d2 <= pass_max <=> sqrt(d2) < max_dist
        kept++;
      } else {
        open = false;
        if (lo <= pass_max) undecided = true;
      }
    }
  }
  return undecided ? -1 : kept;
}

// Exact k-NN of q among the indexed points, then the strict radius filter of kdtree.cpp:25.
// Returns the number of neighbours kept (prefix of r).
template <int KM>
LOAMX_HD int knn_search(const GridDesc& g, const uint32_t* __restrict__ cell_start, const GridPoint* __restrict__ sp,
                        Vec3 q, int k, double max_dist, KnnResult<KM>& r, uint32_t* row_scratch, int row_stride) {
  knn_init(r);
  if (g.n_points == 0 || k <= 0) return 0;
  const int32_t cx = grid_cell_coord(q.x, g.ox, g.inv_h);
  const int32_t cy = grid_cell_coord(q.y, g.oy, g.inv_h);
  const int32_t cz = grid_cell_coord(q.z, g.oz, g.inv_h);
  const int32_t out = grid_outside_distance(g, cx, cy, cz);
  // every point is at least (out-1)*h away: nothing can pass the radius filter
  if (max_dist > 0.0 && out >= 1 && (double)(out - 1) * g.h >= max_dist) return 0;
  knn_rounds(g, cell_start, sp, q, k, max_dist, cx, cy, cz, r, out > 1 ? out : 1, row_scratch, row_stride);
  return knn_finish(r, k, max_dist);
}

// The keyed search on its own: number of neighbours kept, or -1 when the keys cannot decide the
// query (the caller then runs knn_search). Neighbour j (ascending) is pos[(KM - k) + j] — the keyed
// collector keeps its k-th key in a fixed register, which shifts the list by KM - k.
// pass_max = knn_radius_pass_max(max_dist), computed once on the host.
template <int KM>
LOAMX_HD int knn_search_keyed(const GridDesc& g, const uint32_t* __restrict__ cell_start, const GridPoint* __restrict__ sp,
                              Vec3 q, int k, double max_dist, double pass_max, uint32_t pos[KM], uint32_t* row_scratch,
                              int row_stride) {
#pragma unroll
  for (int j = 0; j < KM; j++) pos[j] = 0;
  if (g.n_points == 0 || k <= 0) return 0;
  if (k > KM) k = KM;
  const int32_t cx = grid_cell_coord(q.x, g.ox, g.inv_h);
  const int32_t cy = grid_cell_coord(q.y, g.oy, g.inv_h);
  const int32_t cz = grid_cell_coord(q.z, g.oz, g.inv_h);
  const int32_t out = grid_outside_distance(g, cx, cy, cz);
  if (max_dist > 0.0 && out >= 1 && (double)(out - 1) * g.h >= max_dist) return 0;
  KnnKeys<KM> c;
  knn_init(c, k, g.n_points);
  knn_rounds(g, cell_start, sp, q, k, max_dist, cx, cy, cz, c, out > 1 ? out : 1, row_scratch, row_stride);
  return knn_keys_finish(c, k, pass_max, pos);
}

/* ------------------------------------------------------------------------------------------------
 * FP32 pre-selection (round 1 of the fast kernel).
 *
 * The candidate loop of the keyed collector above is VALU-bound: 8 FP64 operations for the distance
 * and 11 for the insertion, per candidate. Here the loop runs on single-precision copies of the
 * target coordinates (relative to the grid origin, SoA: x[], y[], z[]): ~3 packed instructions for
 * the distance, and a 32-bit key — the bits of the float d2 with the low 8 bits replaced by the
 * candidate's running number in this query's stream — kept in a sorted list of k+1 by v_med3_u32
 * (new slot j = median(old slot j-1, old slot j, key): one instruction per slot, no carry chain).
 *
 * Exactness is restored afterwards, per query: the k selected candidates are fetched in FP64, their
 * exact d2 (bit-identical to what the FP64 collectors compute) must be strictly ascending, and the
 * best rejected key T6 must exceed the exact k-th distance D5 by more than the rigorous error of
 * the FP32 evaluation at that distance:
 *     |d2_32 - d2| <= 2*sqrt(3)*a*d + 3*a^2 + 4*u*d2,   a = 3*u*L,  u = 2^-24,
 * L = largest coordinate offset from the grid origin that can occur (grid extent + 2 cells): the
 * stored offsets and the query's are rounded once (u*L each), their difference once more. Every
 * rejected candidate has d2_32 >= T6, hence d2 >= T6 - err(d); if it were <= D5 its error would be
 * <= err(sqrt(D5)), so T6 - err(sqrt(D5)) > D5 rules that out. Whatever fails a check (ties, an
 * inversion, a near-tie at the boundary, > 63 batches, non-finite values) is "undecided" and goes
 * to the queue like a query that needs more rounds: the result is always that of the exact search.
 * Pruning uses the same inequality the other way round (an upper bound on the true k-th d2).
 * ---------------------------------------------------------------------------------------------- */
template <int KM>
struct KnnKeys32 {
  uint32_t key[KM + 1];  // ascending: KM-k sentinels (0), then the k+1 smallest keys; 0xFFFFFFFF = empty
};

LOAMX_HD uint32_t knn_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
LOAMX_HD uint32_t knn_umed3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
#else
  const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
  return c < lo ? lo : (c > hi ? hi : c);
#endif
}
LOAMX_HD uint32_t knn_f32_bits(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __float_as_uint(v);
#else
  uint32_t b;
  __builtin_memcpy(&b, &v, 4);
  return b;
#endif
}
LOAMX_HD float knn_bits_f32(uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(b);
#else
  float v;
  __builtin_memcpy(&v, &b, 4);
  return v;
#endif
}
template <int KM>
LOAMX_HD void knn_init(KnnKeys32<KM>& c, int k) {
#pragma unroll
  for (int j = 0; j <= KM; j++) c.key[j] = j < KM - k ? 0u : 0xFFFFFFFFu;
}
template <int KM>
LOAMX_HD void knn_key32_insert(KnnKeys32<KM>& c, uint32_t x) {
  uint32_t nk[KM + 1];
  nk[0] = knn_umin(c.key[0], x);
#pragma unroll
  for (int j = 1; j <= KM; j++) nk[j] = knn_umed3(c.key[j - 1], c.key[j], x);
#pragma unroll
  for (int j = 0; j <= KM; j++) c.key[j] = nk[j];
}
// error unit a = 3 u L of the FP32 evaluation on grid g
LOAMX_HD double knn_f32_err_unit(const GridDesc& g) {
  int32_t m = g.nx > g.ny ? g.nx : g.ny;
  m = m > g.nz ? m : g.nz;
  return 3.0 * 5.9604644775390625e-8 * ((double)(m + 2) * g.h);
}
// rigorous upper bound on the true k-th squared distance (DBL_MAX until k keys are held):
// d2 <= d2_32 + 2 sqrt(3) a d + 3 a^2 + 4 u d2 and 2 sqrt(3) a d <= 1e-3 d2 + 3000 a^2
template <int KM>
LOAMX_HD double knn_bound32(const KnnKeys32<KM>& c, double a, uint32_t imask) {
  const uint32_t kth = c.key[KM - 1];
  if (kth >= 0x7F800000u) return kDblMax;  // empty or not finite
  const double hi = (double)knn_bits_f32(kth | imask);
  return (hi + 3003.0 * a * a) * 1.002;
}

struct alignas(4) KnnF4 {
  float v[4];
};
// four candidates p..p+3 of the SoA copy (planes x, y, z of `plane` floats each); lidx = 4 * batch number
// MASK = false: the candidates past the range's end are NOT masked. They are real points of the following cells (or the
// pad entries behind the set, whose offsets are 3e38), so the search stays exact; a point offered twice (as such an
// extra and again in its own row) shows up as two equal distances among the keys and is caught by the verification's
// strictly-ascending test, which sends the query to the queue.
template <int KM, bool MASK = true>
LOAMX_HD void knn_scan_batch_f32(KnnKeys32<KM>& c, float qx, float qy, float qz, const float* __restrict__ rel,
                                 uint32_t plane, uint32_t p, uint32_t n, uint32_t lidx, uint32_t imask) {
  const char* __restrict__ base = reinterpret_cast<const char*>(rel) + (uint32_t)(p << 2);
  const KnnF4 x = *reinterpret_cast<const KnnF4*>(base);
  const KnnF4 y = *reinterpret_cast<const KnnF4*>(base + (size_t)plane * 4);
  const KnnF4 z = *reinterpret_cast<const KnnF4*>(base + (size_t)plane * 8);
#if defined(LOAMX_KNN_STATS)
  g_cand += n < 4u ? n : 4u;
#endif
  uint32_t key[4];
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 mqx = {-qx, -qx}, mqy = {-qy, -qy}, mqz = {-qz, -qz};
  const f2 x01 = {x.v[0], x.v[1]}, x23 = {x.v[2], x.v[3]}, y01 = {y.v[0], y.v[1]}, y23 = {y.v[2], y.v[3]};
  const f2 z01 = {z.v[0], z.v[1]}, z23 = {z.v[2], z.v[3]};
  const f2 dx01 = x01 + mqx, dx23 = x23 + mqx, dy01 = y01 + mqy, dy23 = y23 + mqy, dz01 = z01 + mqz, dz23 = z23 + mqz;
  f2 s01 = dx01 * dx01, s23 = dx23 * dx23;
  s01 = __builtin_elementwise_fma(dy01, dy01, s01), s23 = __builtin_elementwise_fma(dy23, dy23, s23);
  s01 = __builtin_elementwise_fma(dz01, dz01, s01), s23 = __builtin_elementwise_fma(dz23, dz23, s23);
  const float d[4] = {s01.x, s01.y, s23.x, s23.y};
#else
  float d[4];
  for (int i = 0; i < 4; i++) {
    const float dx = x.v[i] - qx, dy = y.v[i] - qy, dz = z.v[i] - qz;
    d[i] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  }
#endif
#pragma unroll
  for (int i = 0; i < 4; i++) {
    key[i] = (knn_f32_bits(d[i]) & ~imask) | (lidx + (uint32_t)i);
    if (MASK && (uint32_t)i >= n) key[i] = 0xFFFFFFFFu;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) knn_key32_insert(c, key[i]);
}

// Four loaded candidates (offsets from the grid origin) through the collector: 8-bit running numbers lidx..lidx+3, tail
// masked by n (valid candidates, may exceed 4).
template <int KM>
LOAMX_HD void knn_collect_batch_f32(KnnKeys32<KM>& c, float qx, float qy, float qz, const KnnF4& x, const KnnF4& y,
                                    const KnnF4& z, uint32_t n, uint32_t lidx, uint32_t keep) {
#if defined(LOAMX_KNN_STATS)
  g_cand += n < 4u ? n : 4u;
#endif
  uint32_t key[4];
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 mqx = {-qx, -qx}, mqy = {-qy, -qy}, mqz = {-qz, -qz};
  const f2 x01 = {x.v[0], x.v[1]}, x23 = {x.v[2], x.v[3]}, y01 = {y.v[0], y.v[1]}, y23 = {y.v[2], y.v[3]};
  const f2 z01 = {z.v[0], z.v[1]}, z23 = {z.v[2], z.v[3]};
  const f2 dx01 = x01 + mqx, dx23 = x23 + mqx, dy01 = y01 + mqy, dy23 = y23 + mqy, dz01 = z01 + mqz, dz23 = z23 + mqz;
  f2 s01 = dx01 * dx01, s23 = dx23 * dx23;
  s01 = __builtin_elementwise_fma(dy01, dy01, s01), s23 = __builtin_elementwise_fma(dy23, dy23, s23);
  s01 = __builtin_elementwise_fma(dz01, dz01, s01), s23 = __builtin_elementwise_fma(dz23, dz23, s23);
  const float d[4] = {s01.x, s01.y, s23.x, s23.y};
#else
  float d[4];
  for (int i = 0; i < 4; i++) {
    const float dx = x.v[i] - qx, dy = y.v[i] - qy, dz = z.v[i] - qz;
    d[i] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  }
#endif
#pragma unroll
  for (int i = 0; i < 4; i++) {
#if defined(__HIP_DEVICE_COMPILE__)
    // (one v_and_or_b32 with `keep` in a vector and the wave-uniform running number in a scalar register)
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key[i]) : "v"(knn_f32_bits(d[i])), "v"(keep), "s"(lidx + (uint32_t)i));
#else
    key[i] = (knn_f32_bits(d[i]) & keep) | (lidx + (uint32_t)i);
#endif
    if ((uint32_t)i >= n) key[i] = 0xFFFFFFFFu;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) knn_key32_insert(c, key[i]);
}
// The batch at position p of the three coordinate planes, given as separate (wave-uniform) base pointers: three loads
// off scalar bases with one 32-bit offset, no per-lane 64-bit address arithmetic.
template <int KM>
LOAMX_HD void knn_scan_batch_f32_soa(KnnKeys32<KM>& c, float qx, float qy, float qz, const float* __restrict__ rel_x,
                                     const float* __restrict__ rel_y, const float* __restrict__ rel_z, uint32_t p,
                                     uint32_t n, uint32_t lidx, uint32_t keep) {
  const uint32_t off = p << 2;
  const KnnF4 x = *reinterpret_cast<const KnnF4*>(reinterpret_cast<const char*>(rel_x) + off);
  const KnnF4 y = *reinterpret_cast<const KnnF4*>(reinterpret_cast<const char*>(rel_y) + off);
  const KnnF4 z = *reinterpret_cast<const KnnF4*>(reinterpret_cast<const char*>(rel_z) + off);
  knn_collect_batch_f32<KM>(c, qx, qy, qz, x, y, z, n, lidx, keep);
}

// Round 1 with the FP32 collector. Returns the number of neighbours kept, or -1 (undecided / needs
// more rounds: queue it). pos as in knn_search_keyed (neighbour j in pos[(KM - k) + j]).
// `rel` = SoA single-precision offsets of the sorted target points from the grid origin, planes of
// `plane` floats.
// Walk, termination test and exact verification of knn_search_f32_round1. WIDE = the running number takes
// 10 or 12 bits (retry of a query that ran out of the 8-bit numbers); the 8-bit variant keeps its masks as
// instruction constants. Returns kept >= 0, -1 (undecided / more rounds) or -2 (running number exhausted).
template <int KM, bool WIDE>
LOAMX_HD int knn_f32_round1_body(const GridDesc& g, const GridPoint* __restrict__ sp, const float* __restrict__ rel,
                                 uint32_t plane, Vec3 q, int k, double max_dist, double pass_max, uint32_t pos[KM],
                                 uint32_t* row_scratch, int row_stride, int32_t cx, int32_t cy, int32_t cz, int nrow,
                                 uint32_t slots) {
  const double a = knn_f32_err_unit(g);
  const double r2 = knn_radius_bound(max_dist);
  const float qx = (float)(q.x - g.ox), qy = (float)(q.y - g.oy), qz = (float)(q.z - g.oz);
  KnnKeys32<KM> c;
  knn_init(c, k);
  // row pruning in single precision, conservatively: slab distances rounded down, the bound up
  const float kDown = 0.99999f, kUp = 1.00001f;
  double sy2m = slab_dist(q.y, g.oy, g.h, cy - 1), sy2p = slab_dist(q.y, g.oy, g.h, cy + 1);
  double sz2m = slab_dist(q.z, g.oz, g.h, cz - 1), sz2p = slab_dist(q.z, g.oz, g.h, cz + 1);
  const float fy2m = (float)(sy2m * sy2m) * kDown, fy2p = (float)(sy2p * sy2p) * kDown;
  const float fz2m = (float)(sz2m * sz2m) * kDown, fz2p = (float)(sz2p * sz2p) * kDown;
  const float fr2 = r2 < 1e37 ? (float)r2 * kUp : 3.0e38f;
  const float fa2 = (float)(3003.0 * a * a) * kUp;
  const uint32_t imask = WIDE ? (slots <= 1024u ? 0x3FFu : 0xFFFu) : 0xFFu;  // (a compile-time constant when !WIDE)
  const uint32_t tmax = (imask + 1u) >> 2;
  uint32_t p = 0, e = 0, t = 0;  // t = batches consumed so far
  int ri = 0;
  for (;;) {
    while (p >= e && ri < nrow) {  // next admissible row
      const uint32_t bj = row_scratch[(2 * ri) * row_stride], e2 = row_scratch[(2 * ri + 1) * row_stride];
      const int j = (int)(bj >> 28), jy = j % 3, jz = j / 3;
      const float sy2 = jy == 0 ? fy2m : (jy == 1 ? 0.0f : fy2p), sz2 = jz == 0 ? fz2m : (jz == 1 ? 0.0f : fz2p);
      // upper bound on the true k-th d2 (see knn_bound32), evaluated in float with upward slack
      const uint32_t kth = c.key[KM - 1];
      const float worst = kth >= 0x7F800000u ? 3.0e38f : (knn_bits_f32(kth | imask) + fa2) * (1.002f * kUp);
      const float bound = worst < fr2 ? worst : fr2;
      if (sy2 + sz2 <= bound) {
#if defined(LOAMX_KNN_STATS)
        g_rows++;
#endif
        p = bj & 0x0FFFFFFFu, e = e2;
        row_scratch[(2 * ri + 1) * row_stride] = 0x80000000u | t;  // visited: remember where its batches start
      }
      ri++;
    }
    if (p >= e || t >= tmax) break;
    knn_scan_batch_f32(c, qx, qy, qz, rel, plane, p, e - p, t << 2, imask);
    p += 4, t++;
  }
  if (p < e) return -2;  // the running number is exhausted (more than 63 / 255 / 1 023 batches)
  // ---- is the search over after the 3x3x3 block? (same test as knn_done, with the rigorous bound)
  {
    double guard = kDblMax;
    const double m = 1e-9 * g.h;
    const int32_t c3[3] = {cx, cy, cz}, n3[3] = {g.nx, g.ny, g.nz};
    const double q3[3] = {q.x, q.y, q.z}, o3[3] = {g.ox, g.oy, g.oz};
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
      if (c3[ax] - 1 > 0) {
        const double d = (q3[ax] - (o3[ax] + (double)(c3[ax] - 1) * g.h)) * (1.0 - 1e-9) - m;
        guard = d < guard ? d : guard;
      }
      if (c3[ax] + 1 < n3[ax] - 1) {
        const double d = ((o3[ax] + (double)(c3[ax] + 2) * g.h) - q3[ax]) * (1.0 - 1e-9) - m;
        guard = d < guard ? d : guard;
      }
    }
    if (guard < 0.0) guard = 0.0;
    const bool done = guard == kDblMax || knn_bound32(c, a, imask) < guard * guard || (max_dist > 0.0 && guard >= max_dist);
    if (!done) return -1;
  }
  // ---- exact verification of the k selected candidates
  // running number -> position: the visited row with the largest first batch <= the key's batch
  uint32_t wbegin[KM], wts[KM];
#pragma unroll
  for (int i = 0; i < KM; i++) wbegin[i] = 0u, wts[i] = 0u;
  for (int rr = 0; rr < nrow; rr++) {
    const uint32_t w0 = row_scratch[(2 * rr) * row_stride] & 0x0FFFFFFFu, w1 = row_scratch[(2 * rr + 1) * row_stride];
    const uint32_t rts = w1 & 0x7FFFFFFFu;
    const bool visited = (w1 & 0x80000000u) != 0u;
#pragma unroll
    for (int i = 0; i < KM; i++) {
      const uint32_t tb = (c.key[i] & imask) >> 2;
      const bool take = visited && rts <= tb && rts >= wts[i];
      wbegin[i] = take ? w0 : wbegin[i];
      wts[i] = take ? rts : wts[i];
    }
  }
  int count = 0, kept = 0;
  bool undecided = false, open = true;
  double prev = -1.0, d5 = 0.0;
#pragma unroll
  for (int i = 0; i < KM; i++) {
    const uint32_t key = c.key[i];
    const bool real = i >= KM - k && key != 0xFFFFFFFFu;
    if (real) {
      if (key >= 0x7F800000u) undecided = true;
      const uint32_t tb = (key & imask) >> 2, ii = key & 3u;
      const uint32_t pp = wbegin[i] + (tb - wts[i]) * 4u + ii;
      pos[i] = pp;
      const GridPoint tp = sp[pp];
      const double dx = q.x - tp.x, dy = q.y - tp.y, dz = q.z - tp.z;
      const double d2 = dx * dx + dy * dy + dz * dz;  // as knn_scan_batch
      if (!(d2 > prev)) undecided = true;             // a tie or an inversion: the exact order is not this one
      if (!(d2 <= kDblMax)) undecided = true;
      prev = d2, d5 = d2;
      count++;
      if (open) {
        if (d2 <= pass_max) kept++;
        else open = false;
      }
    }
  }
  const uint32_t k6 = c.key[KM];
  if (count == k && k6 != 0xFFFFFFFFu) {
    const double t6 = (double)knn_bits_f32(k6 & ~imask);
    const double err = 2.0 * (3.4641016151377544 * a * sqrt(d5) + 3.0 * a * a + 2.384185791015625e-7 * d5);  // x2 safety
    if (!(t6 > d5 + err)) undecided = true;
  }
  return undecided ? -1 : kept;
}


/* ------------------------------------------------------------------------------------------------
 * Round 1 with the FP32 collector, lean form (round 2) — same search, same exactness argument and the same
 * return contract as knn_f32_round1_body<KM, false>, restructured for instruction count: the PMC pass of round 1
 * showed 2 440 vector instructions per wavefront of which the candidate loop was 1 100; the rest was set-up,
 * divergent list handling and a 9-row x k-key scan that mapped running numbers back to positions.
 *   - the non-empty rows of the 3x3x3 block go to the per-thread list as begin | end << 16 (16-bit positions: sets
 *     below 65 536 points) in visiting order, with their squared slab distance as a float next to them;
 *   - the walk takes ONE row step per trip, predicated instead of branched (a lane whose range is used up looks at
 *     its next row and takes it or not; every lane that holds a range scans one batch): no inner loop and none of
 *     the exec-mask bookkeeping the nested divergent loops cost (measured: 1.58 -> 1.45 ms per launch);
 *   - every visited row appends (begin | first batch << 16) to a compact list and sets bit `first batch` in a
 *     64-bit mask: the row of a key with batch number tb is entry popcount(mask & ((2 << tb) - 1)) - 1 — five
 *     instructions per key instead of a scan over the rows.
 * Per-thread list: kLeanRowWords words ([word][thread] in LDS): 9 row ranges, 9 thresholds; the visited list overwrites
 * the row ranges from the front (entry nv <= the index of the row being taken, whose range is in a register by then).
 * ---------------------------------------------------------------------------------------------- */
#if defined(LOAMX_KNN_STATS)
#define LOAMX_LEAN_REASON(r) (g_lean_reason = (r))
#else
#define LOAMX_LEAN_REASON(r) ((void)0)
#endif
constexpr int kLeanRowWords = 18;
#ifndef LOAMX_LEAN_TMAX
#define LOAMX_LEAN_TMAX 64
#endif
constexpr uint32_t kLeanMaxPoints = 65535u;  // positions are packed as 16-bit halves

LOAMX_HD int knn_popcount64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popcll(v);
#else
  return __builtin_popcountll(v);
#endif
}

// What follows the walk of the 3x3x3 block, shared by the lean forms: is the search over (rigorous bound against the
// distance to the block's faces), then the exact verification of the k selected candidates. The visited list (begin |
// first batch << 16 per visited range) starts at word `vis_off` of the per-thread list.
template <int KM, int W = 1>
LOAMX_HD int knn_lean_finish(const GridDesc& g, const GridPoint* __restrict__ sp, Vec3 q, int k, double max_dist, double pass_max,
                             double a, int32_t cx, int32_t cy, int32_t cz, const KnnKeys32<KM>& c, uint64_t started,
                             const uint32_t* row_scratch, int row_stride, int vis_off, uint32_t pos[KM]) {
  constexpr uint32_t imask = 0xFFu;
  constexpr int NR = (2 * W + 1) * (2 * W + 1);  // rows of the block (W = 1: the 3x3x3 block, W = 2: 5x5x5)
  // ---- is the search over after the block of half-width W? (same test as knn_done, with the rigorous bound)
  {
    double guard = kDblMax;
    const double m = 1e-9 * g.h;
    const int32_t c3[3] = {cx, cy, cz}, n3[3] = {g.nx, g.ny, g.nz};
    const double q3[3] = {q.x, q.y, q.z}, o3[3] = {g.ox, g.oy, g.oz};
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
      if (c3[ax] - W > 0) {
        const double d = (q3[ax] - (o3[ax] + (double)(c3[ax] - W) * g.h)) * (1.0 - 1e-9) - m;
        guard = d < guard ? d : guard;
      }
      if (c3[ax] + W < n3[ax] - 1) {
        const double d = ((o3[ax] + (double)(c3[ax] + W + 1) * g.h) - q3[ax]) * (1.0 - 1e-9) - m;
        guard = d < guard ? d : guard;
      }
    }
    if (guard < 0.0) guard = 0.0;
    const bool done = guard == kDblMax || knn_bound32(c, a, imask) < guard * guard || (max_dist > 0.0 && guard >= max_dist);
    if (!done) { LOAMX_LEAN_REASON(1); return -1; }
  }
  // ---- exact verification of the k selected candidates. Three passes, each unconditional over the KM slots, so that
  // the list reads, then the 2 * KM point loads, are all in flight together (as five branches they were five round trips)
  int count = 0, kept = 0;
  bool undecided = false, open = true;
  double prev = -1.0, d5 = 0.0;
  uint32_t vrow[KM];
#pragma unroll
  for (int i = 0; i < KM; i++) {
    const uint32_t tb = (c.key[i] & imask) >> 2;
    const int ord = knn_popcount64(started & ((2ull << tb) - 1ull)) - 1;  // the visited row this batch belongs to
    vrow[i] = row_scratch[(vis_off + (ord < 0 ? 0 : (ord > NR - 1 ? NR - 1 : ord))) * row_stride];
  }
  GridPoint tp[KM];
#pragma unroll
  for (int i = 0; i < KM; i++) {
    const uint32_t key = c.key[i];
    const bool real = i >= KM - k && key != 0xFFFFFFFFu;
    const uint32_t tb = (key & imask) >> 2, ii = key & 3u;
    const uint32_t pp = real ? (vrow[i] & 0xFFFFu) + (tb - (vrow[i] >> 16)) * 4u + ii : 0u;
    pos[i] = pp;
    tp[i] = sp[pp < g.n_points ? pp : 0u];
  }
#pragma unroll
  for (int i = 0; i < KM; i++) {
    const uint32_t key = c.key[i];
    const bool real = i >= KM - k && key != 0xFFFFFFFFu;
    if (real) {
      if (key >= 0x7F800000u) undecided = true;
      const double dx = q.x - tp[i].x, dy = q.y - tp[i].y, dz = q.z - tp[i].z;
      const double d2 = dx * dx + dy * dy + dz * dz;  // as knn_scan_batch
      if (!(d2 > prev)) undecided = true;             // a tie or an inversion: the exact order is not this one
      if (!(d2 <= kDblMax)) undecided = true;
      prev = d2, d5 = d2;
      count++;
      if (open) {
        if (d2 <= pass_max) kept++;
        else open = false;
      }
    }
  }
  const uint32_t k6 = c.key[KM];
  if (count == k && k6 != 0xFFFFFFFFu) {
    const double t6 = (double)knn_bits_f32(k6 & ~imask);
    // an upper bound of sqrt(d5) is enough (float square root, rounded up generously)
#if defined(__HIP_DEVICE_COMPILE__)
    const double sd5 = (double)(__fsqrt_rn((float)d5) * 1.000001f) + 1e-18;
#else
    const double sd5 = (double)(sqrtf((float)d5) * 1.000001f) + 1e-18;
#endif
    const double err = 2.0 * (3.4641016151377544 * a * sd5 + 3.0 * a * a + 2.384185791015625e-7 * d5);  // x2 safety
    if (!(t6 > d5 + err)) undecided = true;
  }
  if (undecided) LOAMX_LEAN_REASON(2);
  return undecided ? -3 : kept;  // (-3: the keys cannot decide — ties; -1 above: the block does not reach far enough)
}

// one batch of four candidates in registers (n = 0: none)
struct LeanBatch {
  KnnF4 x, y, z;
  uint32_t n;
};

// Half of the pipelined loop's body (trip `tu`): step to the next row if the range is used up, issue the loads of the next
// batch into `nxt`, run `cur` (loaded one trip ago) through the collector. Returns whether this lane has anything left.
template <int KM, int NR = 9>
LOAMX_HD bool knn_lean_half_trip(KnnKeys32<KM>& c, LeanBatch& cur, LeanBatch& nxt, uint32_t tu, uint32_t& p, uint32_t& e, int& ri,
                                 int& nv, uint64_t& started, uint32_t& rw, uint32_t& rthr, int nrow, uint32_t* row_scratch,
                                 int row_stride, float qx, float qy, float qz, const float* __restrict__ rel_x,
                                 const float* __restrict__ rel_y, const float* __restrict__ rel_z, uint32_t keep) {
  const bool step = p >= e && ri < nrow;
  const bool take = step && c.key[KM - 1] >= rthr;
  if (take) {
    row_scratch[nv * row_stride] = (rw & 0xFFFFu) | ((tu + 1u) << 16);
    nv++;
    started |= 1ull << ((tu + 1u) & 63u);
  }
  p = take ? (rw & 0xFFFFu) : p, e = take ? (rw >> 16) : e;
  ri += step ? 1 : 0;
  {  // the entry of the row that is next now (used one trip later at the earliest)
    const int rr = ri < NR - 1 ? ri : NR - 1;
    rw = row_scratch[rr * row_stride], rthr = row_scratch[(NR + rr) * row_stride];
  }
  {  // (unconditionally as well — a lane without a range reads position 0 — so that the wait for `cur` below can be
     // "all but the three loads just issued" on every path)
    const bool work = p < e;
    nxt.n = work ? e - p : 0u;
    const uint32_t off = (work ? p : 0u) << 2;
    nxt.x = *reinterpret_cast<const KnnF4*>(reinterpret_cast<const char*>(rel_x) + off);
    nxt.y = *reinterpret_cast<const KnnF4*>(reinterpret_cast<const char*>(rel_y) + off);
    nxt.z = *reinterpret_cast<const KnnF4*>(reinterpret_cast<const char*>(rel_z) + off);
    p += work ? 4u : 0u;
  }
  // (unconditionally: with n = 0 every key is masked. Inside a branch the wait for `cur`'s loads would be conditional too,
  // and the compiler would then drain the load queue before it issues the next loads)
  knn_collect_batch_f32<KM>(c, qx, qy, qz, cur.x, cur.y, cur.z, cur.n, tu << 2, keep);
  return nxt.n != 0u || ri < nrow;
}

// requery: the query point once more, for the verification behind the walk. The round-1 kernel passes a functor that reloads
// and moves the source point again, so that the point's three doubles and its cell are not live through the candidate loop
// (9 registers: what the kernel needs to fit a sixth wavefront per SIMD); KnnSameQuery hands the argument back.
struct KnnSameQuery {
  LOAMX_HD Vec3 operator()(const Vec3& q) const { return q; }
};
template <int KM, typename Requery = KnnSameQuery>
LOAMX_HD int knn_lean_round1(const GridDesc& g, const uint32_t* __restrict__ cell_start, const GridPoint* __restrict__ sp,
                             const float* __restrict__ rel, uint32_t plane, Vec3 q, int k, double max_dist, double pass_max,
                             uint32_t pos[KM], uint32_t* row_scratch, int row_stride, Requery requery = Requery()) {
#pragma unroll
  for (int j = 0; j < KM; j++) pos[j] = 0;
  if (g.n_points == 0 || k <= 0) return 0;
  if (k > KM) k = KM;
  const int32_t cx = grid_cell_coord(q.x, g.ox, g.inv_h);
  const int32_t cy = grid_cell_coord(q.y, g.oy, g.inv_h);
  const int32_t cz = grid_cell_coord(q.z, g.oz, g.inv_h);
  const int32_t out = grid_outside_distance(g, cx, cy, cz);
  if (max_dist > 0.0 && out >= 1 && (double)(out - 1) * g.h >= max_dist) return 0;
  if (out > 1) { LOAMX_LEAN_REASON(3); return -1; }
  const double a = knn_f32_err_unit(g);
  const double r2 = knn_radius_bound(max_dist);
  const float qx = (float)(q.x - g.ox), qy = (float)(q.y - g.oy), qz = (float)(q.z - g.oz);
  // row pruning in single precision, conservatively: slab distances rounded down, the bound up
  const float kDown = 0.99999f, kUp = 1.00001f;
  const double sy2m = slab_dist(q.y, g.oy, g.h, cy - 1), sy2p = slab_dist(q.y, g.oy, g.h, cy + 1);
  const double sz2m = slab_dist(q.z, g.oz, g.h, cz - 1), sz2p = slab_dist(q.z, g.oz, g.h, cz + 1);
  const float fy2[3] = {(float)(sy2m * sy2m) * kDown, 0.0f, (float)(sy2p * sy2p) * kDown};
  const float fz2[3] = {(float)(sz2m * sz2m) * kDown, 0.0f, (float)(sz2p * sz2p) * kDown};
  const float fr2 = r2 < 1e37 ? (float)r2 * kUp : 3.0e38f;
  const float fa2 = (float)(3003.0 * a * a) * kUp;
  int nrow = 0;
  {  // the non-empty rows of the nine within the radius, centre first, then faces, then corners
    const int32_t xa = cx - 1 < 0 ? 0 : cx - 1, xb = cx + 1 > g.nx - 1 ? g.nx - 1 : cx + 1;
    uint32_t rb[9], re[9];
#pragma unroll
    for (int o = 0; o < 9; o++) {
      constexpr int kOrder[9] = {4, 1, 3, 5, 7, 0, 2, 6, 8};
      const int j = kOrder[o];
      const int32_t iy = cy + (j % 3) - 1, iz = cz + (j / 3) - 1;
      const bool ok = xa <= xb && iy >= 0 && iy <= g.ny - 1 && iz >= 0 && iz <= g.nz - 1;
      const uint32_t row = ok ? (uint32_t)((iz * g.ny + iy) * g.nx) : 0u;
      // both ends of the row's range with ONE 16-byte load of entries xa .. xa + 3 (the range ends at xb + 1 <= xa + 3;
      // the table carries spare entries behind its last cell), issued unconditionally: 9 loads instead of 18 behind
      // branches (the kernel is co-limited by the texture addresser: measured 1.24 -> 1.21 ms)
      const CellStart4 c4 = cell_start4_at(cell_start, ok ? row + (uint32_t)xa : 0u);
      const int span = xb + 1 - xa;  // 1 .. 3
      rb[o] = ok ? c4.v[0] : 0u;
      re[o] = ok ? (span == 1 ? c4.v[1] : (span == 2 ? c4.v[2] : c4.v[3])) : 0u;
    }
#pragma unroll
    for (int o = 0; o < 9; o++) {
      constexpr int kOrder[9] = {4, 1, 3, 5, 7, 0, 2, 6, 8};
      const int j = kOrder[o];
      // A row is admissible while its squared slab distance s2 is not above the bound on the k-th distance,
      // (float(kth | imask) + fa2) * 1.002 * kUp, nor above the radius. As a test on the KEY: kth >= thr with
      // thr = bits(x) & ~imask for any x <= s2 / (1.002 kUp) - fa2 (float bits of positive values are monotone, and
      // (kth | imask) >= bits(x) is a comparison of the upper 24 bits) — one integer compare per row step instead of
      // rebuilding the bound from the key every trip. x is taken low: 0.99799 < 1 / (1.002 * 1.00001) by 4e-6.
      const float s2 = fy2[j % 3] + fz2[j / 3];
      const float x = (s2 * 0.99799f - fa2) * kDown;
      const uint32_t thr = x > 0.0f ? (knn_f32_bits(x) & ~0xFFu) : 0u;
      if (rb[o] < re[o] && s2 <= fr2) {
        row_scratch[nrow * row_stride] = rb[o] | (re[o] << 16);
        row_scratch[(9 + nrow) * row_stride] = thr;
        nrow++;
      }
    }
  }
  KnnKeys32<KM> c;
  knn_init(c, k);
#ifndef LOAMX_LEAN_TMAX
#define LOAMX_LEAN_TMAX 64
#endif
  constexpr uint32_t imask = 0xFFu, tmax = LOAMX_LEAN_TMAX;  // trips a lane may take before it hands its query to the queue (<= 64)
  static_assert(imask == 0xFFu, "the row thresholds above clear the same bits");
  const float* __restrict__ rel_x = rel;
  const float* __restrict__ rel_y = rel + (size_t)plane;
  const float* __restrict__ rel_z = rel + 2 * (size_t)plane;
  uint32_t p = 0, e = 0;
  int ri = 0, nv = 0;
  uint64_t started = 0;  // bit t: a visited row starts with batch t
  // One row step per trip, predicated instead of branched: a lane whose range is used up looks at its next row (taking
  // it or not), every lane that holds a range scans one batch; no inner loop, no exec-mask bookkeeping. Every lane
  // still in the loop has made the same number of trips, so the trip counter lives in a scalar register and so do the
  // running numbers derived from it.
  // ~imask held in a vector register the compiler cannot fold back into a literal: (bits & keep) | number is then one
  // v_and_or_b32 per key (a 32-bit literal is not encodable in that instruction; with it the compiler emits two)
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t keep;
  asm volatile("v_mov_b32 %0, 0xffffff00" : "=v"(keep));
#else
  const uint32_t keep = ~imask;
#endif
  uint32_t t = 0;
#if !defined(LOAMX_LEAN_UNPIPELINED)
  // Software-pipelined form: the batch scanned in trip t was loaded in trip t - 1, so the loads of the next batch are in
  // flight while this one goes through the collector. The row step for the next batch is therefore decided BEFORE the
  // current batch is inserted (a bound that is one batch stale: still an upper bound, the search stays exact), and a
  // visited row's first batch number is t + 1. Two register sets alternate (the loop is unrolled by two), so nothing is
  // copied. The list entry of row `ri` is fetched from LDS one trip ahead as well.
  LeanBatch b0 = {}, b1 = {};
  uint32_t rw = row_scratch[0], rthr = row_scratch[9 * row_stride];
  bool more = true;
  for (; t < tmax; t += 2) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t tu = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
#else
    const uint32_t tu = t;
#endif
    // (no exit test between the halves: the compiler would sink the first half's loads behind it, next to their use)
    knn_lean_half_trip<KM>(c, b0, b1, tu, p, e, ri, nv, started, rw, rthr, nrow, row_scratch, row_stride, qx, qy, qz, rel_x, rel_y,
                           rel_z, keep);
    more = knn_lean_half_trip<KM>(c, b1, b0, tu + 1u, p, e, ri, nv, started, rw, rthr, nrow, row_scratch, row_stride, qx, qy, qz,
                                  rel_x, rel_y, rel_z, keep);
    if (!more) break;
  }
  if (more) return -2;  // the trip budget is used up with work left: the queue's business
#else
  for (; t < tmax; t++) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t tu = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
#else
    const uint32_t tu = t;
#endif
    const bool have = ri < nrow, step = p >= e && have;
    const int rr = have ? ri : 0;
    const uint32_t w = row_scratch[rr * row_stride], thr = row_scratch[(9 + rr) * row_stride];
    const bool take = step && c.key[KM - 1] >= thr;
    ri += step ? 1 : 0;
    p = take ? (w & 0xFFFFu) : p, e = take ? (w >> 16) : e;
    if (take) {
      row_scratch[nv * row_stride] = (w & 0xFFFFu) | (tu << 16);
      nv++;
      started |= 1ull << tu;
    }
    if (p >= e && ri >= nrow) break;  // (lanes leave for good)
    // (masked tails: without the mask — 6 of 62 instructions per batch, measured 1.45 -> 1.435 ms — a batch that runs past
    // its row's end offers points of the following cells; in a scene without a far wall those are the next row's first
    // points, i.e. candidates of this very query, and every such duplicate sends the query to the queue)
    if (p < e) {
      knn_scan_batch_f32_soa<KM>(c, qx, qy, qz, rel_x, rel_y, rel_z, p, e - p, tu << 2, keep);
      p += 4u;
    }
  }
#endif
#if defined(LOAMX_KNN_STATS)
  g_lean_trips = t, g_lean_nrow = (uint32_t)nrow, g_lean_taken = (uint32_t)nv;
#endif
#if defined(LOAMX_LEAN_UNPIPELINED)
  if (p < e || ri < nrow) return -2;  // the trip budget is used up with rows still to look at: the queue's business
#endif
  {
    const Vec3 q2 = requery(q);
    const int32_t cx2 = grid_cell_coord(q2.x, g.ox, g.inv_h), cy2 = grid_cell_coord(q2.y, g.oy, g.inv_h), cz2 = grid_cell_coord(q2.z, g.oz, g.inv_h);
    return knn_lean_finish<KM>(g, sp, q2, k, max_dist, pass_max, a, cx2, cy2, cz2, c, started, row_scratch, row_stride, 0, pos);
  }
}


/* ------------------------------------------------------------------------------------------------
 * The same lean search over a larger block (round 3), half-width W cells: W = 2 (5x5x5) is the second chance of a query
 * whose 3x3x3 block did not hold k points closer than the block's faces — sparse neighbourhoods: ~3 % of the plane
 * queries of the bench workload, which the FP64 search over all rounds (knn_search_keyed) served at 19 times the cost
 * per query of the first round; W = 4 (9x9x9) reaches the search radius itself when the cell edge is a quarter of it
 * (analysed on the host only — tests/hostcheck, EXPERIMENTS.md: it finishes half of the isolated queries, the other half
 * run out of the 63 batches — the kernels use W = 2).
 * (2W+1)^2 rows of up to 2W+1 cells, in the order of their squared slab distance, 2W+1 rows' table entries in flight
 * at a time; same collector, same exactness argument (verification against the FP64 points, best rejected key,
 * distance to the faces of the block that WAS searched, or the radius), same return contract: >= 0 done, -1 the block
 * does not reach far enough, -2 more than 63 batches, -3 tied keys. Per-thread list: 2 (2W+1)^2 words.
 * ---------------------------------------------------------------------------------------------- */
template <int W>
struct LeanRowOrder {  // the rows (dy, dz) of the block, entry = (dy + W) + (2W+1) (dz + W), by dy^2 + dz^2 (stable)
  static constexpr int D = 2 * W + 1, NR = D * D;
  int v[NR];
  static constexpr int key(int j) { return (j % D - W) * (j % D - W) + (j / D - W) * (j / D - W); }
  constexpr LeanRowOrder() : v() {
    for (int i = 0; i < NR; i++) v[i] = i;
    for (int i = 1; i < NR; i++) {
      const int x = v[i];
      int j = i - 1;
      while (j >= 0 && key(v[j]) > key(x)) { v[j + 1] = v[j]; j--; }
      v[j + 1] = x;
    }
  }
};
constexpr int kLean2Rows = 25, kLean2RowWords = 2 * kLean2Rows, kLean4Rows = 81, kLean4RowWords = 2 * kLean4Rows;
template <int KM, int W>
LOAMX_HD int knn_lean_block(const GridDesc& g, const uint32_t* __restrict__ cell_start, const GridPoint* __restrict__ sp,
                            const float* __restrict__ rel, uint32_t plane, Vec3 q, int k, double max_dist, double pass_max,
                            uint32_t pos[KM], uint32_t* row_scratch, int row_stride) {
  constexpr int D = 2 * W + 1, NR = D * D;
#pragma unroll
  for (int j = 0; j < KM; j++) pos[j] = 0;
  if (g.n_points == 0 || k <= 0) return 0;
  if (k > KM) k = KM;
  const int32_t cx = grid_cell_coord(q.x, g.ox, g.inv_h);
  const int32_t cy = grid_cell_coord(q.y, g.oy, g.inv_h);
  const int32_t cz = grid_cell_coord(q.z, g.oz, g.inv_h);
  const int32_t out = grid_outside_distance(g, cx, cy, cz);
  if (max_dist > 0.0 && out >= 1 && (double)(out - 1) * g.h >= max_dist) return 0;
  if (out > W) return -1;
  const double a = knn_f32_err_unit(g);
  const double r2 = knn_radius_bound(max_dist);
  const float qx = (float)(q.x - g.ox), qy = (float)(q.y - g.oy), qz = (float)(q.z - g.oz);
  const float kDown = 0.99999f, kUp = 1.00001f;
  float fy2[D], fz2[D];
#pragma unroll
  for (int d = 0; d < D; d++) {
    const double sy = d == W ? 0.0 : slab_dist(q.y, g.oy, g.h, cy + d - W), sz = d == W ? 0.0 : slab_dist(q.z, g.oz, g.h, cz + d - W);
    fy2[d] = (float)(sy * sy) * kDown, fz2[d] = (float)(sz * sz) * kDown;
  }
  const float fr2 = r2 < 1e37 ? (float)r2 * kUp : 3.0e38f;
  const float fa2 = (float)(3003.0 * a * a) * kUp;
  int nrow = 0;
  {
    const int32_t xa = cx - W < 0 ? 0 : cx - W, xb = cx + W > g.nx - 1 ? g.nx - 1 : cx + W;
    constexpr LeanRowOrder<W> order{};  // W = 2: 12, 7, 11, 13, 17, 6, 8, 16, 18, 2, 10, 14, 22, 1, 3, 5, 9, ...
#pragma unroll
    for (int o0 = 0; o0 < NR; o0 += D) {  // D rows' table entries in flight at a time
      uint32_t rb[D], re[D];
#pragma unroll
      for (int u = 0; u < D; u++) {
        const int j = order.v[o0 + u];
        const int32_t iy = cy + (j % D) - W, iz = cz + (j / D) - W;
        const bool ok = xa <= xb && iy >= 0 && iy <= g.ny - 1 && iz >= 0 && iz <= g.nz - 1;
        const uint32_t row = ok ? (uint32_t)((iz * g.ny + iy) * g.nx) : 0u;
        rb[u] = cell_start_at(cell_start, ok ? row + (uint32_t)xa : 0u);
        re[u] = cell_start_at(cell_start, ok ? row + (uint32_t)xb + 1u : 0u);
        if (!ok) rb[u] = re[u] = 0u;
      }
#pragma unroll
      for (int u = 0; u < D; u++) {
        const int j = order.v[o0 + u];
        const float s2 = fy2[j % D] + fz2[j / D];
        const float x = (s2 * 0.99799f - fa2) * kDown;  // (as knn_lean_round1: the row's admissibility as a threshold on the key)
        const uint32_t thr = x > 0.0f ? (knn_f32_bits(x) & ~0xFFu) : 0u;
        if (rb[u] < re[u] && s2 <= fr2) {
          row_scratch[nrow * row_stride] = rb[u] | (re[u] << 16);
          row_scratch[(NR + nrow) * row_stride] = thr;
          nrow++;
        }
      }
    }
  }
  KnnKeys32<KM> c;
  knn_init(c, k);
  constexpr uint32_t tmax = 64;
  const float* __restrict__ rel_x = rel;
  const float* __restrict__ rel_y = rel + (size_t)plane;
  const float* __restrict__ rel_z = rel + 2 * (size_t)plane;
  uint32_t p = 0, e = 0;
  int ri = 0, nv = 0;
  uint64_t started = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t keep;
  asm volatile("v_mov_b32 %0, 0xffffff00" : "=v"(keep));
#else
  const uint32_t keep = ~0xFFu;
#endif
  LeanBatch b0 = {}, b1 = {};
  uint32_t rw = row_scratch[0], rthr = row_scratch[NR * row_stride];
  bool more = true;
  uint32_t t = 0;
  for (; t < tmax; t += 2) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t tu = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
#else
    const uint32_t tu = t;
#endif
    knn_lean_half_trip<KM, NR>(c, b0, b1, tu, p, e, ri, nv, started, rw, rthr, nrow, row_scratch, row_stride, qx, qy, qz, rel_x, rel_y,
                               rel_z, keep);
    more = knn_lean_half_trip<KM, NR>(c, b1, b0, tu + 1u, p, e, ri, nv, started, rw, rthr, nrow, row_scratch, row_stride, qx, qy, qz,
                                      rel_x, rel_y, rel_z, keep);
    if (!more) break;
  }
  if (more) return -2;  // more than 63 batches: the FP64 search's business
  return knn_lean_finish<KM, W>(g, sp, q, k, max_dist, pass_max, a, cx, cy, cz, c, started, row_scratch, row_stride, 0, pos);
}
template <int KM>
LOAMX_HD int knn_lean_round2(const GridDesc& g, const uint32_t* __restrict__ cell_start, const GridPoint* __restrict__ sp,
                             const float* __restrict__ rel, uint32_t plane, Vec3 q, int k, double max_dist, double pass_max,
                             uint32_t pos[KM], uint32_t* row_scratch, int row_stride) {
  return knn_lean_block<KM, 2>(g, cell_start, sp, rel, plane, q, k, max_dist, pass_max, pos, row_scratch, row_stride);
}

// WIDE = false: 8-bit running numbers (the fast kernel; a query with more than 63 batches is queued);
// WIDE = true: 10 or 12 bits (the queue kernel tries this before the FP64 search: dense local maps).
template <int KM, bool WIDE = false, typename Requery = KnnSameQuery>
LOAMX_HD int knn_search_f32_round1(const GridDesc& g, const uint32_t* __restrict__ cell_start,
                                   const GridPoint* __restrict__ sp, const float* __restrict__ rel, uint32_t plane, Vec3 q,
                                   int k, double max_dist, double pass_max, uint32_t pos[KM], uint32_t* row_scratch,
                                   int row_stride, Requery requery = Requery()) {
#if !defined(LOAMX_NO_LEAN_KNN)
  if (!WIDE && g.n_points <= kLeanMaxPoints)  // (wave-uniform: a property of the target set)
    return knn_lean_round1<KM>(g, cell_start, sp, rel, plane, q, k, max_dist, pass_max, pos, row_scratch, row_stride, requery);
#endif
#pragma unroll
  for (int j = 0; j < KM; j++) pos[j] = 0;
  if (g.n_points == 0 || k <= 0) return 0;
  if (k > KM) k = KM;
  const int32_t cx = grid_cell_coord(q.x, g.ox, g.inv_h);
  const int32_t cy = grid_cell_coord(q.y, g.oy, g.inv_h);
  const int32_t cz = grid_cell_coord(q.z, g.oz, g.inv_h);
  const int32_t out = grid_outside_distance(g, cx, cy, cz);
  if (max_dist > 0.0 && out >= 1 && (double)(out - 1) * g.h >= max_dist) return 0;
  if (out > 1) return -1;
  uint32_t pop = 0;
  const int nrow = knn_round1_list(g, cell_start, cx, cy, cz, row_scratch, row_stride, &pop);
  // the key carries the candidate's running number 4 t + i in its low bits: 8 bits (63 batches) cover a
  // scan-sized target; a denser block (a local map) takes more bits at the price of a coarser pre-selection
  const uint32_t slots = pop + 4u * (uint32_t)nrow + 4u;  // every row may end in a partly filled batch
  const int kept = knn_f32_round1_body<KM, WIDE>(g, sp, rel, plane, q, k, max_dist, pass_max, pos, row_scratch, row_stride, cx,
                                                 cy, cz, nrow, slots);
  return kept;  // >= 0, -1 (undecided / needs more rounds) or -2 (ran out of running numbers)
}

// The complete search: keyed collector over all rounds, then the exact collector for a query whose
// keys are undecided (associate_knn_rest_kernel; queries queued by the round-1 kernel). Same shifted
// layout of pos.
// `fallbacks`, when given, counts the re-runs.
template <int KM>
LOAMX_HD int knn_search_positions(const GridDesc& g, const uint32_t* __restrict__ cell_start,
                                  const GridPoint* __restrict__ sp, Vec3 q, int k, double max_dist, double pass_max,
                                  uint32_t pos[KM], uint32_t* row_scratch, int row_stride, uint32_t* fallbacks = nullptr) {
  int kept = knn_search_keyed<KM>(g, cell_start, sp, q, k, max_dist, pass_max, pos, row_scratch, row_stride);
  if (kept < 0) {
    if (fallbacks) (*fallbacks)++;
    if (k > KM) k = KM;
    KnnResult<KM> r;
    kept = knn_search(g, cell_start, sp, q, k, max_dist, r, row_scratch, row_stride);
    knn_shift_positions<KM>(r.pos, k, pos);
  }
  return kept;
}

/* ------------------------------------------------------------------------------------------------
 * fitLine (geometry.cpp:42-59): PCA direction through the centroid; the condition number the
 * reference returns is always DBL_MAX (dead guard), so it is not computed.
 * ---------------------------------------------------------------------------------------------- */
LOAMX_HD void jacobi_rotate(double& app, double& aqq, double& apq, double& arp, double& arq, double& vp0, double& vp1,
                            double& vp2, double& vq0, double& vq1, double& vq2) {
  // annihilate apq; r is the third index. (app,aqq,apq) 2x2 block, (arp,arq) couplings.
  if (apq == 0.0) return;
  const double theta = (aqq - app) / (2.0 * apq);
  const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
  const double app_n = app - t * apq, aqq_n = aqq + t * apq;
  const double arp_n = c * arp - s * arq, arq_n = s * arp + c * arq;
  app = app_n, aqq = aqq_n, apq = 0.0, arp = arp_n, arq = arq_n;
  double a, b;
  a = vp0, b = vq0, vp0 = c * a - s * b, vq0 = s * a + c * b;
  a = vp1, b = vq1, vp1 = c * a - s * b, vq1 = s * a + c * b;
  a = vp2, b = vq2, vp2 = c * a - s * b, vq2 = s * a + c * b;
}

template <int KM>
LOAMX_HD void fit_line(const Vec3 pts[KM], int K, Vec3& a, Vec3& b) {
  Vec3 sum = v3(0, 0, 0);
#pragma unroll
  for (int i = 0; i < KM; i++)
    if (i < K) sum = vadd(sum, pts[i]);
  const Vec3 center = v3(sum.x / (double)K, sum.y / (double)K, sum.z / (double)K);
  double a00 = 0, a11 = 0, a22 = 0, a01 = 0, a02 = 0, a12 = 0;
#pragma unroll
  for (int i = 0; i < KM; i++) {
    if (i < K) {
      const Vec3 d = vsub(pts[i], center);
      a00 += d.x * d.x, a11 += d.y * d.y, a22 += d.z * d.z;
      a01 += d.x * d.y, a02 += d.x * d.z, a12 += d.y * d.z;
    }
  }
  // cyclic Jacobi; columns of V are eigenvectors
  double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
  for (int sweep = 0; sweep < 32; sweep++) {
    if (a01 == 0.0 && a02 == 0.0 && a12 == 0.0) break;
    jacobi_rotate(a00, a11, a01, a02, a12, v00, v10, v20, v01, v11, v21);  // (p,q,r) = (0,1,2)
    jacobi_rotate(a00, a22, a02, a01, a12, v00, v10, v20, v02, v12, v22);  // (0,2,1)
    jacobi_rotate(a11, a22, a12, a01, a02, v01, v11, v21, v02, v12, v22);  // (1,2,0)
  }
  // "The eigenvector of the largest eigenvalue" (geometry.cpp:51) as Eigen orders equal eigenvalues [RECALLED]: its
  // selection sort takes the FIRST minimum of the remaining values, so for a diagonal covariance (neighbours on a
  // lattice) (0, s, s) -> column 2, (s, s, 0) -> column 0, (s, 0, s) -> column 2, (s, s, s) -> column 2.
  int c0 = 0, c1 = 1, c2 = 2;
  double e0 = a00, e1 = a11, e2 = a22;
  {  // i = 0: first minimum of (e0, e1, e2) goes to the front
    int k = 0;
    if (e1 < e0) k = 1;
    if (e2 < (k == 1 ? e1 : e0)) k = 2;
    if (k == 1) { double t = e0; e0 = e1; e1 = t; int ti = c0; c0 = c1; c1 = ti; }
    if (k == 2) { double t = e0; e0 = e2; e2 = t; int ti = c0; c0 = c2; c2 = ti; }
  }
  if (e2 < e1) { int ti = c1; c1 = c2; c2 = ti; }  // i = 1
  (void)c0, (void)c1, (void)e0;
  const Vec3 dir = c2 == 0 ? v3(v00, v10, v20) : (c2 == 1 ? v3(v01, v11, v21) : v3(v02, v12, v22));
  a = vadd(center, vscale(0.1, dir));  // geometry.cpp:53
  b = vsub(center, vscale(0.1, dir));
}

/* ------------------------------------------------------------------------------------------------
 * Square roots and divisions of fit_plane. The compiler's FP64 sequences are correctly rounded and long: a division
 * is v_div_scale x2, v_rcp (quarter rate) + two Newton steps, quotient, remainder, v_div_fmas, v_div_fixup = 11
 * instructions; a square root 17, of which 7 only rescale tiny arguments and patch 0 / inf. They were 55 % of the
 * issue slots of associate_fit_kernel. FAST = the very same sequences with
 *   - the steps that depend on the divisor alone (refined reciprocal) done once for the divisions that share it,
 *   - the scaling / special-case steps left out, which is the identity when every operand is a normal number of
 *     moderate exponent: that is CHECKED per operand (`ok`), and a fit that fails one check anywhere is redone with
 *     the plain operators (fit_plane below). Results are bit-identical to the plain operators either way
 *     (tests/test_gpu_direct.py compares the kernels with the g++ build of this header bit for bit).
 * Host builds only have the plain operators.
 * ---------------------------------------------------------------------------------------------- */
struct SharedDivisor {
  double y, r;  // divisor, refined reciprocal (FAST) — or just the divisor
};
#if defined(__HIP_DEVICE_COMPILE__)
LOAMX_HD bool fast_range(double x) {  // 2^-300 < |x| < 2^300: v_div_scale would leave both operands alone, no step under- / overflows
  const double a = fabs(x);
  return a > 4.909093465297727e-91 && a < 2.037035976334486e+90;
}
template <bool FAST>
LOAMX_HD double fit_sqrt(double x, bool& ok) {
  if (!FAST) return sqrt(x);
  ok = ok && x > 1.9010815379079637e-211 && x < 1.0715086071862673e+301;  // (2^-700, 2^1000): no rescaling, not 0 / inf / nan
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = y * 0.5;
  const double e = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, e, g);
  h = __builtin_fma(h, e, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}
template <bool FAST>
LOAMX_HD SharedDivisor fit_divisor(double y, bool& ok) {
  if (!FAST) return SharedDivisor{y, 0.0};
  ok = ok && fast_range(y);
  double r = __builtin_amdgcn_rcp(y);
  double e = __builtin_fma(-y, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-y, r, 1.0);
  r = __builtin_fma(r, e, r);
  return SharedDivisor{y, r};
}
template <bool FAST>
LOAMX_HD double fit_div(double x, const SharedDivisor& s, bool& ok) {
  if (!FAST) return x / s.y;
  ok = ok && fast_range(x);  // (a zero numerator too: the remainder step would lose the sign of a -0 quotient)
  const double q = x * s.r;
  const double rem = __builtin_fma(-s.y, q, x);
  return __builtin_fma(rem, s.r, q);
}
#else
template <bool FAST>
LOAMX_HD double fit_sqrt(double x, bool&) { return sqrt(x); }
template <bool FAST>
LOAMX_HD SharedDivisor fit_divisor(double y, bool&) { return SharedDivisor{y, 0.0}; }
template <bool FAST>
LOAMX_HD double fit_div(double x, const SharedDivisor& s, bool&) { return x / s.y; }
#endif

/* ------------------------------------------------------------------------------------------------
 * fitPlane (geometry.cpp:62-73): least squares P * abc = 1 by column-pivoted Householder QR
 * (Eigen ColPivHouseholderQR semantics incl. its near-zero pivot cut-off), n = abc/|abc|,
 * d = 1/|abc|, returns the signed mean of P n - d.
 * ---------------------------------------------------------------------------------------------- */
template <int KM, bool FAST>
LOAMX_HD double fit_plane_impl(const Vec3 pts[KM], int K, Vec3& normal, double& d_out, bool& ok) {
  double c0[KM], c1[KM], c2[KM];  // columns of the K x 3 matrix
#pragma unroll
  for (int r = 0; r < KM; r++) {
    c0[r] = r < K ? pts[r].x : 0.0;
    c1[r] = r < K ? pts[r].y : 0.0;
    c2[r] = r < K ? pts[r].z : 0.0;
  }
  double nu0 = 0, nu1 = 0, nu2 = 0;
#pragma unroll
  for (int r = 0; r < KM; r++) nu0 += c0[r] * c0[r], nu1 += c1[r] * c1[r], nu2 += c2[r] * c2[r];
  nu0 = fit_sqrt<FAST>(nu0, ok), nu1 = fit_sqrt<FAST>(nu1, ok), nu2 = fit_sqrt<FAST>(nu2, ok);
  double nd0 = nu0, nd1 = nu1, nd2 = nu2;
  double maxn = nu0 > nu1 ? nu0 : nu1;
  maxn = maxn > nu2 ? maxn : nu2;
  const double threshold_helper = (maxn * kDblEps) * (maxn * kDblEps) / (double)K;
  const double downdate_thr = 1.4901161193847656e-08;  // sqrt(eps)
  int p0 = 0, p1 = 1, p2 = 2;  // column permutation: physical column j holds original column pj
  int nonzero_pivots = 3;
  double tau0 = 0, tau1 = 0, tau2 = 0;
  SharedDivisor piv0 = {1.0, 1.0}, piv1 = {1.0, 1.0}, piv2 = {1.0, 1.0};  // the pivots beta_k divide tau_k here and y_k in the back substitution

#define LOAMX_SWAP(a_, b_) { double t_ = a_; a_ = b_; b_ = t_; }
#define LOAMX_SWAPCOL(ca, cb) { _Pragma("unroll") for (int r_ = 0; r_ < KM; r_++) LOAMX_SWAP(ca[r_], cb[r_]) }
  // ---- k = 0
  {
    int big = 0;
    double bn = nu0;
    if (nu1 > bn) big = 1, bn = nu1;
    if (nu2 > bn) big = 2, bn = nu2;
    if (nonzero_pivots == 3 && bn * bn < threshold_helper * (double)(K - 0)) nonzero_pivots = 0;
    if (big == 1) { LOAMX_SWAPCOL(c0, c1) LOAMX_SWAP(nu0, nu1) LOAMX_SWAP(nd0, nd1) int t = p0; p0 = p1; p1 = t; }
    if (big == 2) { LOAMX_SWAPCOL(c0, c2) LOAMX_SWAP(nu0, nu2) LOAMX_SWAP(nd0, nd2) int t = p0; p0 = p2; p2 = t; }
    double tail = 0;
#pragma unroll
    for (int r = 1; r < KM; r++) tail += c0[r] * c0[r];
    const double x0 = c0[0];
    double beta;
    if (tail <= kDblMin) {
      tau0 = 0, beta = x0;
#pragma unroll
      for (int r = 1; r < KM; r++) c0[r] = 0;
    } else {
      beta = fit_sqrt<FAST>(x0 * x0 + tail, ok);
      if (x0 >= 0) beta = -beta;
      const SharedDivisor dv = fit_divisor<FAST>(x0 - beta, ok);
#pragma unroll
      for (int r = 1; r < KM; r++) c0[r] = fit_div<FAST>(c0[r], dv, ok);
    }
    piv0 = fit_divisor<FAST>(beta, ok);
    if (!(tail <= kDblMin)) tau0 = fit_div<FAST>(beta - x0, piv0, ok);
    c0[0] = beta;
    if (tau0 != 0) {
      double t1 = c1[0], t2 = c2[0];
#pragma unroll
      for (int r = 1; r < KM; r++) t1 += c0[r] * c1[r], t2 += c0[r] * c2[r];
      c1[0] -= tau0 * t1, c2[0] -= tau0 * t2;
#pragma unroll
      for (int r = 1; r < KM; r++) c1[r] -= tau0 * c0[r] * t1, c2[r] -= tau0 * c0[r] * t2;
    }
    // norm down-dating for columns 1, 2
    if (nu1 != 0) {
      double temp = fabs(c1[0]) / nu1;
      temp = (1.0 + temp) * (1.0 - temp);
      temp = temp < 0 ? 0 : temp;
      const double ratio = nu1 / nd1;
      // (one square root for both outcomes: the compiler evaluated both sides)
      const bool redo = temp * ratio * ratio <= downdate_thr;
      double s = 0;
#pragma unroll
      for (int r = 1; r < KM; r++) s += c1[r] * c1[r];
      const double sq = fit_sqrt<FAST>(redo ? s : temp, ok);
      if (redo) nd1 = sq, nu1 = sq;
      else nu1 *= sq;
    }
    if (nu2 != 0) {
      double temp = fabs(c2[0]) / nu2;
      temp = (1.0 + temp) * (1.0 - temp);
      temp = temp < 0 ? 0 : temp;
      const double ratio = nu2 / nd2;
      // (one square root for both outcomes: the compiler evaluated both sides)
      const bool redo = temp * ratio * ratio <= downdate_thr;
      double s = 0;
#pragma unroll
      for (int r = 1; r < KM; r++) s += c2[r] * c2[r];
      const double sq = fit_sqrt<FAST>(redo ? s : temp, ok);
      if (redo) nd2 = sq, nu2 = sq;
      else nu2 *= sq;
    }
  }
  // ---- k = 1
  {
    int big = 1;
    double bn = nu1;
    if (nu2 > bn) big = 2, bn = nu2;
    if (nonzero_pivots == 3 && bn * bn < threshold_helper * (double)(K - 1)) nonzero_pivots = 1;
    if (big == 2) { LOAMX_SWAPCOL(c1, c2) LOAMX_SWAP(nu1, nu2) LOAMX_SWAP(nd1, nd2) int t = p1; p1 = p2; p2 = t; }
    double tail = 0;
#pragma unroll
    for (int r = 2; r < KM; r++) tail += c1[r] * c1[r];
    const double x0 = c1[1];
    double beta;
    if (tail <= kDblMin) {
      tau1 = 0, beta = x0;
#pragma unroll
      for (int r = 2; r < KM; r++) c1[r] = 0;
    } else {
      beta = fit_sqrt<FAST>(x0 * x0 + tail, ok);
      if (x0 >= 0) beta = -beta;
      const SharedDivisor dv = fit_divisor<FAST>(x0 - beta, ok);
#pragma unroll
      for (int r = 2; r < KM; r++) c1[r] = fit_div<FAST>(c1[r], dv, ok);
    }
    piv1 = fit_divisor<FAST>(beta, ok);
    if (!(tail <= kDblMin)) tau1 = fit_div<FAST>(beta - x0, piv1, ok);
    c1[1] = beta;
    if (tau1 != 0) {
      double t2 = c2[1];
#pragma unroll
      for (int r = 2; r < KM; r++) t2 += c1[r] * c2[r];
      c2[1] -= tau1 * t2;
#pragma unroll
      for (int r = 2; r < KM; r++) c2[r] -= tau1 * c1[r] * t2;
    }
    if (nu2 != 0) {
      double temp = fabs(c2[1]) / nu2;
      temp = (1.0 + temp) * (1.0 - temp);
      temp = temp < 0 ? 0 : temp;
      const double ratio = nu2 / nd2;
      // (one square root for both outcomes: the compiler evaluated both sides)
      const bool redo = temp * ratio * ratio <= downdate_thr;
      double s = 0;
#pragma unroll
      for (int r = 2; r < KM; r++) s += c2[r] * c2[r];
      const double sq = fit_sqrt<FAST>(redo ? s : temp, ok);
      if (redo) nd2 = sq, nu2 = sq;
      else nu2 *= sq;
    }
  }
  // ---- k = 2
  {
    if (nonzero_pivots == 3 && nu2 * nu2 < threshold_helper * (double)(K - 2)) nonzero_pivots = 2;
    double tail = 0;
#pragma unroll
    for (int r = 3; r < KM; r++) tail += c2[r] * c2[r];
    const double x0 = c2[2];
    double beta;
    if (tail <= kDblMin) {
      tau2 = 0, beta = x0;
#pragma unroll
      for (int r = 3; r < KM; r++) c2[r] = 0;
    } else {
      beta = fit_sqrt<FAST>(x0 * x0 + tail, ok);
      if (x0 >= 0) beta = -beta;
      const SharedDivisor dv = fit_divisor<FAST>(x0 - beta, ok);
#pragma unroll
      for (int r = 3; r < KM; r++) c2[r] = fit_div<FAST>(c2[r], dv, ok);
    }
    piv2 = fit_divisor<FAST>(beta, ok);
    if (!(tail <= kDblMin)) tau2 = fit_div<FAST>(beta - x0, piv2, ok);
    c2[2] = beta;
  }
#undef LOAMX_SWAPCOL
#undef LOAMX_SWAP
  // ---- solve: c = Q^T * ones (first nonzero_pivots reflectors), back substitution, un-permute.
  // Rows >= K of the zero-padded columns are zero, so they never contribute.
  double rhs[KM];
#pragma unroll
  for (int r = 0; r < KM; r++) rhs[r] = r < K ? 1.0 : 0.0;
  if (nonzero_pivots > 0 && tau0 != 0) {
    double t = rhs[0];
#pragma unroll
    for (int r = 1; r < KM; r++) t += c0[r] * rhs[r];
    rhs[0] -= tau0 * t;
#pragma unroll
    for (int r = 1; r < KM; r++) rhs[r] -= tau0 * c0[r] * t;
  }
  if (nonzero_pivots > 1 && tau1 != 0) {
    double t = rhs[1];
#pragma unroll
    for (int r = 2; r < KM; r++) t += c1[r] * rhs[r];
    rhs[1] -= tau1 * t;
#pragma unroll
    for (int r = 2; r < KM; r++) rhs[r] -= tau1 * c1[r] * t;
  }
  if (nonzero_pivots > 2) {
    if (K - 2 == 1) {
      rhs[2] *= (1.0 - tau2);
    } else if (tau2 != 0) {
      double t = rhs[2];
#pragma unroll
      for (int r = 3; r < KM; r++) t += c2[r] * rhs[r];
      rhs[2] -= tau2 * t;
    }
  }
  // R = [[c0[0], c1[0], c2[0]], [0, c1[1], c2[1]], [0, 0, c2[2]]]
  double y0 = 0, y1 = 0, y2 = 0;
  if (nonzero_pivots > 2) y2 = fit_div<FAST>(rhs[2], piv2, ok);  // (c2[2] == beta of k = 2, etc.)
  if (nonzero_pivots > 1) y1 = fit_div<FAST>(rhs[1] - c2[1] * y2, piv1, ok);
  if (nonzero_pivots > 0) y0 = fit_div<FAST>(rhs[0] - c1[0] * y1 - c2[0] * y2, piv0, ok);
  double abc[3] = {0, 0, 0};
#pragma unroll
  for (int j = 0; j < 3; j++) {
    if (p0 == j) abc[j] = y0;
    if (p1 == j) abc[j] = y1;
    if (p2 == j) abc[j] = y2;
  }
  const double n = fit_sqrt<FAST>(abc[0] * abc[0] + abc[1] * abc[1] + abc[2] * abc[2], ok);
  const SharedDivisor dn = fit_divisor<FAST>(n, ok);
  normal = v3(fit_div<FAST>(abc[0], dn, ok), fit_div<FAST>(abc[1], dn, ok), fit_div<FAST>(abc[2], dn, ok));
  d_out = fit_div<FAST>(1.0, dn, ok);
  double sum = 0;
#pragma unroll
  for (int r = 0; r < KM; r++)
    if (r < K) sum += (pts[r].x * normal.x + pts[r].y * normal.y + pts[r].z * normal.z) - d_out;
  return sum / (double)K;
}

template <int KM>
LOAMX_HD double fit_plane(const Vec3 pts[KM], int K, Vec3& normal, double& d_out) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (K == KM) {  // (zero-padded rows are zero numerators: those sets take the plain operators directly)
    bool ok = true;
    const double avg = fit_plane_impl<KM, true>(pts, K, normal, d_out, ok);
    if (ok) return avg;
  }
#endif
  bool unused = true;
  return fit_plane_impl<KM, false>(pts, K, normal, d_out, unused);
}

/* ------------------------------------------------------------------------------------------------
 * One residual block: value, tangent Jacobian row (1x6), Huber(1.0) corrected, accumulated into the
 * normal equations. acc[0..20] = upper triangle of J^T J (row-major: 00 01 .. 05 11 12 ..),
 * acc[21..26] = J^T f, acc[27] = cost, acc[28] = number of non-finite evaluations.
 * x = ambient update (qx,qy,qz,qw,tx,ty,tz); p = the already moved source point.
 * Edge: prim = a(3), b(3) (registration-inl.h:92-103); plane: prim = n(3), d (registration-inl.h:106-117).
 * ---------------------------------------------------------------------------------------------- */
constexpr int kAccSize = 29;

LOAMX_HD void residual_accumulate(bool is_plane, Vec3 v, const double prim[6], const double x[7], double acc[kAccSize]) {
  const Vec3 u = v3(x[0], x[1], x[2]);
  const double w = x[3];
  Vec3 uv = vcross(u, v);
  uv = vadd(uv, uv);
  const Vec3 pp = vadd(vadd(vadd(v, vscale(w, uv)), vcross(u, uv)), v3(x[4], x[5], x[6]));
  Vec3 g;
  double r;
  if (is_plane) {
    const Vec3 n = v3(prim[0], prim[1], prim[2]);
    const double s = vdot(n, pp) - prim[3];
    r = fabs(s);
    g = vscale(copysign(1.0, s), n);  // Jet abs: copysign(1, s)
  } else {
    const Vec3 a = v3(prim[0], prim[1], prim[2]), b = v3(prim[3], prim[4], prim[5]);
    const Vec3 c = vcross(vsub(pp, a), vsub(pp, b));
    const double cn = vnorm(c);
    const Vec3 ab = vsub(a, b);
    const double den = vnorm(ab);
    r = cn / den;
    g = vscale(1.0 / (cn * den), vcross(ab, c));
  }
  // ambient Jacobian wrt (ux,uy,uz,w), then Ceres QuaternionManifold::PlusJacobian applied to the
  // Eigen-ordered storage: rows indexed by storage slot, read as (W,X,Y,Z)
  const double udv = vdot(u, v);
  // M = -2w[v]x + 2(u.v)I + 2 u v^T - 4 v u^T ; amb_j = g^T M[:,j]
  const double gu = vdot(g, u), gv = vdot(g, v);
  const Vec3 vxg = vcross(v, g);  // g^T [v]x = (g x v)^T  =>  g^T (-2w [v]x) = 2w (v x g)^T
  const double amb0 = 2.0 * w * vxg.x + 2.0 * udv * g.x + 2.0 * gu * v.x - 4.0 * gv * u.x;
  const double amb1 = 2.0 * w * vxg.y + 2.0 * udv * g.y + 2.0 * gu * v.y - 4.0 * gv * u.y;
  const double amb2 = 2.0 * w * vxg.z + 2.0 * udv * g.z + 2.0 * gu * v.z - 4.0 * gv * u.z;
  const double amb3 = 2.0 * vdot(g, vcross(u, v));
  const double W = x[0], X = x[1], Y = x[2], Z = x[3];
  double J[6];
  J[0] = amb0 * (-X) + amb1 * W + amb2 * (-Z) + amb3 * Y;
  J[1] = amb0 * (-Y) + amb1 * Z + amb2 * W + amb3 * (-X);
  J[2] = amb0 * (-Z) + amb1 * (-Y) + amb2 * X + amb3 * W;
  J[3] = g.x, J[4] = g.y, J[5] = g.z;
  bool finite = (r - r == 0.0);
#pragma unroll
  for (int j = 0; j < 6; j++) finite = finite && (J[j] - J[j] == 0.0);
  if (!finite) {
    acc[28] += 1.0;
    return;
  }
  // HuberLoss(1.0) + Corrector (rho'' <= 0 => scale residual and Jacobian by sqrt(rho'))
  const double s2 = r * r;
  double rho0 = s2, rho1 = 1.0;
  if (s2 > 1.0) {
    const double sr = sqrt(s2);
    rho0 = 2.0 * sr - 1.0;
    rho1 = 1.0 / sr;
    if (rho1 < kDblMin) rho1 = kDblMin;
  }
  const double sc = sqrt(rho1);
  const double f = r * sc;
#pragma unroll
  for (int j = 0; j < 6; j++) J[j] *= sc;
  int t = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) {
#pragma unroll
    for (int j = i; j < 6; j++) acc[t++] += J[i] * J[j];
  }
#pragma unroll
  for (int j = 0; j < 6; j++) acc[21 + j] += J[j] * f;
  acc[27] += 0.5 * rho0;
}

/* ------------------------------------------------------------------------------------------------
 * Plane residuals through moments.
 *
 * The signed point-to-plane residual is a polynomial in the ambient update x = (u, w, t) whose
 * coefficients depend on the association record (moved point v, normal n, offset d) only:
 *     s(x) = n.(v + 2w (u x v) + 2 u x (u x v) + t) - d  =  c . phi(x),
 *     c   = [n.v - d,  v x n (3),  q_xx q_yy q_zz q_xy q_xz q_yz,  n (3)]        (13 numbers per record)
 *     phi = [1, 2w u (3), ux^2 uy^2 uz^2 ux*uy ux*uz uy*uz, t (3)],
 *     q_aa = 2 n_a v_a - 2 n.v,  q_ab = 2 (n_a v_b + n_b v_a).
 * The residual block is |s| with Jacobian sign(s) ds/dx, so J^T J, J^T f and the cost only see s.
 * While every |s| stays below the Huber threshold (1.0: the corrector is the identity there) the sums
 * over all plane records of one pair are therefore
 *     cost = phi^T M phi / 2,   J^T f = Z^T M phi,   J^T J = Z^T M Z,     M = sum c c^T  (13x13),
 * Z(x) = d phi / d tangent (13x6, with the same QuaternionManifold convention as residual_accumulate).
 * M is a Gram matrix: accumulated once per ICF iteration by streaming the records through FP64 MFMA
 * (moment_kernel); the evaluations of the trust-region solve then cost O(1) per pair instead of a pass
 * over the records each. The guarantee |s_i(x)| <= max|s_i(0)| + |pp_i(x) - v_i| < 1 is checked per pair
 * and candidate from the two maxima the moment pass also delivers; a pair / candidate that fails it
 * streams its plane records as before (the first ICF iteration usually does: its update is large).
 * ---------------------------------------------------------------------------------------------- */
constexpr int kMomDim = 13;
constexpr int kMomStride = 16;                       // the moment matrix is kept as 16x16 (MFMA tile), rows/cols 13..15 zero
constexpr int kMomSize = kMomStride * kMomStride;    // doubles per pair, + 2 maxima
constexpr double kMomInlier = 0.5;  // plane records with |s0| above this stay out of the moments (evaluated one by one)
LOAMX_HD void plane_coeffs(Vec3 v, Vec3 n, double d, double c[kMomDim]) {
  const double nv = vdot(n, v);
  c[0] = nv - d;
  const Vec3 m = vcross(v, n);
  c[1] = m.x, c[2] = m.y, c[3] = m.z;
  c[4] = 2.0 * n.x * v.x - 2.0 * nv, c[5] = 2.0 * n.y * v.y - 2.0 * nv, c[6] = 2.0 * n.z * v.z - 2.0 * nv;
  c[7] = 2.0 * (n.x * v.y + n.y * v.x), c[8] = 2.0 * (n.x * v.z + n.z * v.x), c[9] = 2.0 * (n.y * v.z + n.z * v.y);
  c[10] = n.x, c[11] = n.y, c[12] = n.z;
}
// adds the plane terms of one pair at x to acc (layout of residual_accumulate); M = 16x16 row-major
LOAMX_HD void plane_eval_from_moments(const double* M, const double x[7], double acc[29]) {
  const double ux = x[0], uy = x[1], uz = x[2], w = x[3];
  double phi[kMomDim] = {1.0, 2.0 * w * ux, 2.0 * w * uy, 2.0 * w * uz, ux * ux, uy * uy, uz * uz,
                         ux * uy, ux * uz, uy * uz, x[4], x[5], x[6]};
  // ambient gradient of phi wrt (ux, uy, uz, w), then the tangent map of residual_accumulate
  double A[kMomDim][4];
#pragma unroll
  for (int j = 0; j < kMomDim; j++) A[j][0] = A[j][1] = A[j][2] = A[j][3] = 0.0;
  A[1][0] = 2.0 * w, A[1][3] = 2.0 * ux;
  A[2][1] = 2.0 * w, A[2][3] = 2.0 * uy;
  A[3][2] = 2.0 * w, A[3][3] = 2.0 * uz;
  A[4][0] = 2.0 * ux, A[5][1] = 2.0 * uy, A[6][2] = 2.0 * uz;
  A[7][0] = uy, A[7][1] = ux;
  A[8][0] = uz, A[8][2] = ux;
  A[9][1] = uz, A[9][2] = uy;
  const double W = x[0], X = x[1], Y = x[2], Zq = x[3];
  double Z[kMomDim][6];
#pragma unroll
  for (int j = 0; j < kMomDim; j++) {
    Z[j][0] = A[j][0] * (-X) + A[j][1] * W + A[j][2] * (-Zq) + A[j][3] * Y;
    Z[j][1] = A[j][0] * (-Y) + A[j][1] * Zq + A[j][2] * W + A[j][3] * (-X);
    Z[j][2] = A[j][0] * (-Zq) + A[j][1] * (-Y) + A[j][2] * X + A[j][3] * W;
    Z[j][3] = j == 10 ? 1.0 : 0.0, Z[j][4] = j == 11 ? 1.0 : 0.0, Z[j][5] = j == 12 ? 1.0 : 0.0;
  }
  double y[kMomDim], T[kMomDim][6];
#pragma unroll
  for (int i = 0; i < kMomDim; i++) {
    double yi = 0.0, ti[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < kMomDim; j++) {
      const double mij = M[i * kMomStride + j];
      yi += mij * phi[j];
#pragma unroll
      for (int a = 0; a < 6; a++) ti[a] += mij * Z[j][a];
    }
    y[i] = yi;
#pragma unroll
    for (int a = 0; a < 6; a++) T[i][a] = ti[a];
  }
  int t = 0;
#pragma unroll
  for (int a = 0; a < 6; a++) {
#pragma unroll
    for (int b = a; b < 6; b++) {
      double sum = 0.0;
#pragma unroll
      for (int i = 0; i < kMomDim; i++) sum += Z[i][a] * T[i][b];
      acc[t++] += sum;
    }
  }
  double cost = 0.0;
#pragma unroll
  for (int i = 0; i < kMomDim; i++) cost += phi[i] * y[i];
#pragma unroll
  for (int a = 0; a < 6; a++) {
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < kMomDim; i++) sum += Z[i][a] * y[i];
    acc[21 + a] += sum;
  }
  acc[27] += 0.5 * cost;
  if (!(cost - cost == 0.0)) acc[28] += 1.0;  // non-finite moments: as a failed evaluation
}
// May the moments stand in for the plane records at x? Yes if every |s_i(x)| is provably below the
// Huber threshold: |s_i(x)| <= |s_i(0)| + |pp_i(x) - v_i|, and for the quaternion formula (also for
// non-unit quaternions) |pp - v - t| = |2w u x v + 2 (u (u.v) - v (u.u))| <= (2 |w| |u| + 4 |u|^2) |v|.
LOAMX_HD bool plane_moments_valid_at(double s0max, double v2max, const double x[7]) {
  const double uu = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
  const double rot = 2.0 * fabs(x[3]) * sqrt(uu) + 4.0 * uu;
  const double trans = sqrt(x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
  return s0max + rot * sqrt(v2max) + trans < 0.999;  // (NaN compares false)
}

// The same relative to a reference candidate r at which every |s_i(r)| <= sref_max is known (first ICF iteration: the
// moments are taken after the first step, whose motion alone exceeds the bound above):
// |s_i(x)| <= |s_i(r)| + |pp_i(x) - pp_i(r)|, pp(x) = v + A(x) v + t with A(x) = 2 w [u]x + 2 (u u^T - (u.u) I), hence
// |pp_i(x) - pp_i(r)| <= |A(x) - A(r)| |v_i| + |t - t_r| and
// |A(x) - A(r)| <= 2 (|w - w_r| |u| + |w_r| |u - u_r|) + 4 |u - u_r| (|u| + |u_r|)      (|[a]x| = |a|, |a b^T| = |a| |b|).
LOAMX_HD bool plane_moments_valid_rel(double sref_max, double v2max, const double x[7], const double r[7]) {
  const double du = sqrt((x[0] - r[0]) * (x[0] - r[0]) + (x[1] - r[1]) * (x[1] - r[1]) + (x[2] - r[2]) * (x[2] - r[2]));
  const double nu = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]), nr = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  const double rot = 2.0 * (fabs(x[3] - r[3]) * nu + fabs(r[3]) * du) + 4.0 * du * (nu + nr);
  const double trans = sqrt((x[4] - r[4]) * (x[4] - r[4]) + (x[5] - r[5]) * (x[5] - r[5]) + (x[6] - r[6]) * (x[6] - r[6]));
  return sref_max + rot * sqrt(v2max) + trans < 0.999;  // (NaN compares false)
}
// phi(x) of the comment above: s_i(x) = c_i . phi(x)
LOAMX_HD void plane_phi(const double x[7], double phi[kMomDim]) {
  const double ux = x[0], uy = x[1], uz = x[2], w = x[3];
  phi[0] = 1.0, phi[1] = 2.0 * w * ux, phi[2] = 2.0 * w * uy, phi[3] = 2.0 * w * uz;
  phi[4] = ux * ux, phi[5] = uy * uy, phi[6] = uz * uz, phi[7] = ux * uy, phi[8] = ux * uz, phi[9] = uy * uz;
  phi[10] = x[4], phi[11] = x[5], phi[12] = x[6];
}

/* ------------------------------------------------------------------------------------------------
 * Ceres manifold Plus for the 7 ambient doubles (QuaternionManifold on raw storage read as
 * (W,X,Y,Z), EuclideanManifold<3>)
 * ---------------------------------------------------------------------------------------------- */
LOAMX_HD void manifold_plus(const double x[7], const double delta[6], double out[7]) {
  const double nd = sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
  if (nd == 0.0) {
    out[0] = x[0], out[1] = x[1], out[2] = x[2], out[3] = x[3];
  } else {
    const double sbd = sin(nd) / nd;
    const double z0 = cos(nd), z1 = sbd * delta[0], z2 = sbd * delta[1], z3 = sbd * delta[2];
    out[0] = z0 * x[0] - z1 * x[1] - z2 * x[2] - z3 * x[3];
    out[1] = z0 * x[1] + z1 * x[0] + z2 * x[3] - z3 * x[2];
    out[2] = z0 * x[2] - z1 * x[3] + z2 * x[0] + z3 * x[1];
    out[3] = z0 * x[3] + z1 * x[2] - z2 * x[1] + z3 * x[0];
  }
  out[4] = x[4] + delta[3], out[5] = x[5] + delta[4], out[6] = x[6] + delta[5];
}

/* ------------------------------------------------------------------------------------------------
 * Per-pair trust-region state (Ceres 2.2.0 TrustRegionMinimizer + LevenbergMarquardtStrategy with
 * DENSE_QR replaced by the algebraically identical damped normal equations on the 6x6 system).
 * One sweep = evaluation of cost / J^T J / J^T f at `xeval`; lm_begin / lm_step consume a sweep
 * and either publish the next point to evaluate or finish.
 * ---------------------------------------------------------------------------------------------- */
struct LmState {
  double x[7];       // current accepted point
  double x_user[7];  // what the caller's parameter blocks hold (updated by successful steps)
  double xeval[7];   // point the next sweep evaluates
  double H[21], g[6];
  double x_cost, minimum_cost, x_norm;
  double scaling[6], diagonal[6];
  double radius, decrease_factor;
  double model_cost_change;
  int32_t iteration;
  int32_t num_invalid;
  int32_t reuse_diagonal;
  int32_t active;  // 1 while more sweeps are needed
};

LOAMX_HD double sym_at(const double H[21], int i, int j) {
  if (i > j) {
    const int t = i;
    i = j;
    j = t;
  }
  // offset of row i in the packed upper triangle: i*6 - i*(i-1)/2
  return H[i * 6 - (i * (i - 1)) / 2 + (j - i)];
}

// Solve (S H S + D^2) y = S g by Cholesky; returns false if not positive definite / non-finite.
LOAMX_HD bool lm_solve6(const double H[21], const double g[6], const double scaling[6], const double d2[6], double y[6]) {
  double A[6][6], b[6];
  #pragma unroll
  for (int i = 0; i < 6; i++) {
    #pragma unroll
    for (int j = 0; j < 6; j++) A[i][j] = scaling[i] * sym_at(H, i, j) * scaling[j];
    A[i][i] += d2[i];
    b[i] = scaling[i] * g[i];
  }
  #pragma unroll
  for (int j = 0; j < 6; j++) {
    double s = A[j][j];
    #pragma unroll
    for (int k = 0; k < j; k++) s -= A[j][k] * A[j][k];
    if (!(s > 0.0)) return false;
    const double l = sqrt(s);
    A[j][j] = l;
    #pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double t = A[i][j];
      #pragma unroll
      for (int k = 0; k < j; k++) t -= A[i][k] * A[j][k];
      A[i][j] = t / l;
    }
  }
  #pragma unroll
  for (int i = 0; i < 6; i++) {
    double t = b[i];
    #pragma unroll
    for (int k = 0; k < i; k++) t -= A[i][k] * y[k];
    y[i] = t / A[i][i];
  }
  #pragma unroll
  for (int i = 5; i >= 0; i--) {
    double t = y[i];
    #pragma unroll
    for (int k = i + 1; k < 6; k++) t -= A[k][i] * y[k];
    y[i] = t / A[i][i];
  }
  #pragma unroll
  for (int i = 0; i < 6; i++)
    if (!(y[i] - y[i] == 0.0)) return false;
  return true;
}

// Runs trust-region iterations that need no new evaluation (invalid steps) until it can publish a
// candidate in st.xeval (returns true) or the solve is over (returns false).
LOAMX_HD bool lm_propose(LmState& st) {
  const int max_num_iterations = 4;
  for (;;) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue (the successful-step bookkeeping is done by
    // the caller before it gets here)
    if (st.iteration >= max_num_iterations) return false;
    if (st.radius <= 1e-32) return false;
    st.iteration++;
    // LevenbergMarquardtStrategy::ComputeStep on the column-scaled Jacobian
    if (!st.reuse_diagonal) {
      #pragma unroll
      for (int j = 0; j < 6; j++) {
        double d = st.scaling[j] * sym_at(st.H, j, j) * st.scaling[j];
        d = d < 1e-6 ? 1e-6 : d;
        d = d > 1e32 ? 1e32 : d;
        st.diagonal[j] = d;
      }
    }
    double d2[6], y[6];
    #pragma unroll
    for (int j = 0; j < 6; j++) d2[j] = st.diagonal[j] / st.radius;  // lm_diagonal^2
    const bool solved = lm_solve6(st.H, st.g, st.scaling, d2, y);
    st.reuse_diagonal = 1;
    bool valid = false;
    double step[6];
    if (solved) {
      #pragma unroll
      for (int j = 0; j < 6; j++) step[j] = -y[j];
      // model_cost_change = -(J step)^T (f + J step / 2) = -step^T S g - step^T S H S step / 2
      double lin = 0, quad = 0;
      #pragma unroll
      for (int i = 0; i < 6; i++) {
        lin += step[i] * st.scaling[i] * st.g[i];
        double row = 0;
        #pragma unroll
        for (int j = 0; j < 6; j++) row += st.scaling[i] * sym_at(st.H, i, j) * st.scaling[j] * step[j];
        quad += step[i] * row;
      }
      st.model_cost_change = -lin - 0.5 * quad;
      valid = st.model_cost_change > 0.0;
    }
    if (!valid) {
      if (++st.num_invalid >= 5) return false;
      st.radius *= 0.5;  // LM StepIsInvalid
      st.reuse_diagonal = 0;
      continue;
    }
    st.num_invalid = 0;
    double delta[6];
    #pragma unroll
    for (int j = 0; j < 6; j++) delta[j] = step[j] * st.scaling[j];
    manifold_plus(st.x, delta, st.xeval);
    return true;
  }
}

LOAMX_HD void lm_init(LmState& st) {
  #pragma unroll
  for (int i = 0; i < 7; i++) st.x[i] = st.x_user[i] = st.xeval[i] = (i == 3) ? 1.0 : 0.0;  // Pose3d() identity
  st.radius = 1e4, st.decrease_factor = 2.0;
  st.iteration = 0, st.num_invalid = 0, st.reuse_diagonal = 0, st.active = 1;
  st.x_cost = 0, st.minimum_cost = 0, st.x_norm = 1.0, st.model_cost_change = 0;
  #pragma unroll
  for (int j = 0; j < 6; j++) st.scaling[j] = 1.0, st.diagonal[j] = 0.0, st.g[j] = 0.0;
  #pragma unroll
  for (int j = 0; j < 21; j++) st.H[j] = 0.0;
}

LOAMX_HD double gradient_max_norm(const LmState& st) {
  double neg[6], xp[7];
  #pragma unroll
  for (int j = 0; j < 6; j++) neg[j] = -st.g[j];
  manifold_plus(st.x, neg, xp);
  double m = 0;
  #pragma unroll
  for (int i = 0; i < 7; i++) {
    const double a = fabs(st.x[i] - xp[i]);
    m = a > m ? a : m;
  }
  return m;
}

// Consume the sweep at xeval. `first` = this was the iteration-0 evaluation at the identity.
// acc layout as residual_accumulate. Sets st.active = 0 when the solve is over.
LOAMX_HD void lm_consume(LmState& st, const double acc[kAccSize], bool first) {
  const bool eval_ok = (acc[28] == 0.0);
  if (first) {
    if (!eval_ok) {  // "Initial residual and Jacobian evaluation failed": parameters untouched
      st.active = 0;
      return;
    }
    #pragma unroll
    for (int j = 0; j < 21; j++) st.H[j] = acc[j];
    #pragma unroll
    for (int j = 0; j < 6; j++) st.g[j] = acc[21 + j];
    st.x_cost = acc[27];
    st.minimum_cost = st.x_cost;
    #pragma unroll
    for (int j = 0; j < 6; j++) st.scaling[j] = 1.0 / (1.0 + sqrt(sym_at(st.H, j, j)));  // jacobi scaling, once
    st.active = lm_propose(st) ? 1 : 0;
    return;
  }
  const double cand_cost = eval_ok ? acc[27] : kDblMax;
  // ParameterToleranceReached: candidate discarded
  double step_norm = 0;
  #pragma unroll
  for (int i = 0; i < 7; i++) step_norm += (st.x[i] - st.xeval[i]) * (st.x[i] - st.xeval[i]);
  step_norm = sqrt(step_norm);
  if (step_norm <= 1e-8 * (st.x_norm + 1e-8)) {
    st.active = 0;
    return;
  }
  // FunctionToleranceReached: candidate discarded
  if (fabs(st.x_cost - cand_cost) <= 1e-6 * st.x_cost) {
    st.active = 0;
    return;
  }
  const double relative_decrease = (st.x_cost - cand_cost) / st.model_cost_change;
  if (relative_decrease > 1e-3) {
    // HandleSuccessfulStep: the sweep already holds J^T J, J^T f at the candidate
    #pragma unroll
    for (int i = 0; i < 7; i++) st.x[i] = st.xeval[i];
    double n2 = 0;
    #pragma unroll
    for (int i = 0; i < 7; i++) n2 += st.x[i] * st.x[i];
    st.x_norm = sqrt(n2);
    // (a non-finite Jacobian at an accepted candidate would have made cand_cost = DBL_MAX above)
    #pragma unroll
    for (int j = 0; j < 21; j++) st.H[j] = acc[j];
    #pragma unroll
    for (int j = 0; j < 6; j++) st.g[j] = acc[21 + j];
    st.x_cost = acc[27];
    double q = 2.0 * relative_decrease - 1.0;
    q = 1.0 - q * q * q;
    st.radius = st.radius / (q > 1.0 / 3.0 ? q : 1.0 / 3.0);
    st.radius = st.radius < 1e16 ? st.radius : 1e16;
    st.decrease_factor = 2.0;
    st.reuse_diagonal = 0;
    // Finalize: publish to the user's parameter blocks
    if (st.x_cost < st.minimum_cost) {
      st.minimum_cost = st.x_cost;
      #pragma unroll
      for (int i = 0; i < 7; i++) st.x_user[i] = st.x[i];
    }
    if (st.iteration < 4 && gradient_max_norm(st) <= 1e-10) {
      st.active = 0;
      return;
    }
  } else {
    st.radius = st.radius / st.decrease_factor;  // StepRejected
    st.decrease_factor *= 2.0;
    st.reuse_diagonal = 1;
  }
  st.active = lm_propose(st) ? 1 : 0;
}

// registration-inl.h:63-73: est <- update (+) est; converged iff the update is small.
LOAMX_HD bool outer_update(double est[7], const double update[7], double rot_thresh, double pos_thresh) {
  double next[7];
  pose_compose(update, est, next);
  for (int i = 0; i < 7; i++) est[i] = next[i];
  const double angle_change = quat_angle_to_identity(update);
  const double position_change = sqrt(update[4] * update[4] + update[5] * update[5] + update[6] * update[6]);
  return angle_change < rot_thresh && position_change < pos_thresh;
}

}  // namespace loamx
