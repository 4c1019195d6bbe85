// register_kernels.hip — loam::registerFeatures on gfx950 for a batch of independent pairs
// (reference: loam/include/loam/registration-inl.h:11-78, loam/src/registration.cpp:23-103).
//
// Per outer ICF iteration the host enqueues, for all pairs at once:
//   associate_knn_* / associate_fit_* kernels (edge, plane)   rows a16-a19: move the source point by the
//                           current estimate, exact k-NN in the target's uniform grid (replaces the
//                           nanoflann KD-tree), fitLine / fitPlane in registers, guards, write the
//                           association record (structure of arrays). See launch_associate.
//   lm_begin_kernel         min_associations check, Ceres problem/solver state reset.
//   first ICF iteration: 5 x { sweep_kernel, lm_step_kernel }
//                           rows a20-a22: sweep = residual + Jacobian row + Huber + 6x6 normal
//                           equations over all association slots of a pair, streamed from HBM
//                           (72 B per edge slot, 56 B per plane slot), wavefront shuffle + LDS
//                           reduction to one 29-double partial per workgroup; lm_step = fixed-order
//                           reduction of the partials and the trust-region bookkeeping of one
//                           Levenberg-Marquardt iteration (6x6 Cholesky) per pair.
//   later ICF iterations: moment_kernel, lm_pair_loop_kernel
//                           the plane records of a pair summarised by a 13x13 moment matrix (FP64 MFMA), then
//                           ONE wavefront per pair runs the whole Levenberg-Marquardt solve off it.
//   outer_update_kernel     row a23: est <- update (+) est, convergence test, termination type.
// grid_build_kernel (once per call) builds the per-target-set grid entirely in LDS.
#include "loamx_internal.h"

namespace loamx {

namespace {

/* ------------------------------------------------------------------------------------------------ */
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmin(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  return v;
}

#ifndef LOAMX_BUILD_THREADS
#define LOAMX_BUILD_THREADS 1024
#endif
constexpr int kBuildThreads = LOAMX_BUILD_THREADS;

// One workgroup builds the whole index of one target set: bounding box, cell size, counting sort by
// cell with the cell table in LDS (128 KiB), exclusive scan, scatter. Deterministic in everything
// but the order of points inside a cell, which the search does not depend on.
// PACKED (sets of at most kGridSmallCap points: every scan-sized set): two 16-bit cell counters per LDS word (the
// table is 64 KiB; counts and offsets stay below 65 536, so a half never carries into its neighbour), and the
// rest of the LDS holds the cell order as a list of 16-bit point indices. The scatter then is an LDS write
// (s_inv[position] = index) and the points leave the workgroup in one COALESCED pass over the finished order
// (gather of 24 bytes from the L2-resident input, 32-byte GridPoint + the float copies written in sequence).
// Measured per workgroup before: the scattered 2 x 16-byte global stores per point took 52-68 us of the
// build's 110-135 us and the float-copy pass another 14-30 us. ORDERED sets get their reproducible order in LDS
// as well (rank among the cell mates -> second list), which replaces the scratch copy and grid_rank_kernel.
template <bool ORDERED, bool PACKED>
__global__ __launch_bounds__(kBuildThreads) void grid_build_kernel(const double* __restrict__ pts_base,
                                                                   const uint32_t* __restrict__ n_pts, size_t stride,
                                                                   uint32_t in_pitch, double max_dist, GridSet gs, GridPoint* __restrict__ scratch,
                                                                   unsigned long long* __restrict__ bytes, const unsigned long long* __restrict__ box_min,
                                                                   const unsigned long long* __restrict__ box_max, const uint32_t* __restrict__ box_bad,
                                                                   const uint32_t* __restrict__ small_if) {
  // (pairs whose target set of this kind is brute-force sized: small_sets_build_kernel builds both of their sets)
  if (small_if && small_if[blockIdx.x * in_pitch] <= kBruteMax) return;
  __shared__ uint32_t s_cells[PACKED ? kGridLdsCells / 2 : kGridLdsCells];
  auto cell_get = [&](uint32_t c) -> uint32_t { return PACKED ? (s_cells[c >> 1] >> ((c & 1u) * 16u)) & 0xFFFFu : s_cells[c]; };
  auto cell_add = [&](uint32_t c) -> uint32_t {  // returns the value before the increment
    if (PACKED) {
      const uint32_t sh = (c & 1u) * 16u;
      return (atomicAdd(&s_cells[c >> 1], 1u << sh) >> sh) & 0xFFFFu;
    }
    return atomicAdd(&s_cells[c], 1u);
  };
  __shared__ uint16_t s_inv[PACKED ? kGridSmallCap : 1];
  __shared__ uint16_t s_cellof[PACKED ? kGridSmallCap : 1];  // cell of every point (count + scatter phases) ...
  uint16_t* s_inv2 = s_cellof;                               // ... then the ranked list of ORDERED sets
  __shared__ double s_red[6][kBuildThreads / 64];
  __shared__ uint32_t s_wave_sum[kBuildThreads / 64];
  __shared__ GridDesc s_g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t pair = blockIdx.x;
  const uint32_t n_raw = n_pts[pair * in_pitch];
  const uint32_t n = n_raw < stride ? n_raw : (uint32_t)stride;
  const double* __restrict__ pts = pts_base + pair * in_pitch * stride * 3;
#ifdef LOAMX_BUILD_PROFILE
  unsigned long long stamp[12];
  int ns = 0;
#define STAMP() do { __syncthreads(); stamp[ns++] = wall_clock64(); } while (0)
#else
#define STAMP() do {} while (0)
#endif
  STAMP();

  // The bounding box: handed over by the extraction that produced the set (round 5: select_rows_kernel's copy phase takes
  // the minima / maxima of the points it copies — the same six numbers, min and max are exact), or one read pass here.
  const bool have_box = box_min != nullptr && __hip_atomic_load(box_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;  // uniform
  if (!have_box) {
    double lx = kDblMax, ly = kDblMax, lz = kDblMax, hx = -kDblMax, hy = -kDblMax, hz = -kDblMax;
#pragma unroll 4
    for (uint32_t i = tid; i < n; i += kBuildThreads) {
      const double x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
      lx = fmin(lx, x), ly = fmin(ly, y), lz = fmin(lz, z);
      hx = fmax(hx, x), hy = fmax(hy, y), hz = fmax(hz, z);
    }
    lx = wave_min(lx), ly = wave_min(ly), lz = wave_min(lz);
    hx = wave_max(hx), hy = wave_max(hy), hz = wave_max(hz);
    if (lane == 0) {
      s_red[0][wave] = lx, s_red[1][wave] = ly, s_red[2][wave] = lz;
      s_red[3][wave] = hx, s_red[4][wave] = hy, s_red[5][wave] = hz;
    }
  }
  STAMP();  // bbox loads
  __syncthreads();
  if (tid == 0) {
    double a[6];
    if (have_box) {
      for (int k = 0; k < 3; k++) {
        a[k] = n ? key_dbl(box_min[(pair * in_pitch) * 6 + k]) : kDblMax;
        a[3 + k] = n ? key_dbl(box_max[(pair * in_pitch) * 6 + k]) : -kDblMax;
      }
    } else {
      for (int k = 0; k < 6; k++) {
        a[k] = s_red[k][0];
        for (int w = 1; w < kBuildThreads / 64; w++) a[k] = k < 3 ? fmin(a[k], s_red[k][w]) : fmax(a[k], s_red[k][w]);
      }
    }
    GridDesc g;
    if (ORDERED) grid_choose_morton(g, v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), n);
    else grid_choose(g, v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), n, max_dist, kGridCellsCap);
    s_g = g;
    gs.desc[pair] = g;
  }
  __syncthreads();
  const GridDesc g = s_g;
  STAMP();  // grid choice
  const uint32_t ncell = (uint32_t)(g.nx * g.ny * g.nz);
  uint32_t* __restrict__ cs = gs.cell_start + pair * (size_t)(kGridCellsCap + 1);
  GridPoint* __restrict__ sp = gs.sorted + pair * gs.stride;
  // ORDERED: scatter into the scratch copy first, then place every point at (cell begin + number of
  // cell mates with a smaller original index): the final layout does not depend on the order in
  // which the atomics were served, so every later summation order is reproducible run to run.
  GridPoint* __restrict__ dst = ORDERED ? scratch + pair * gs.stride : sp;
  // The LDS table holds kGridLdsCells cells; larger grids are built in passes over cell ranges
  // (every pass re-reads the points; the running total carries the exclusive scan across passes).
  uint32_t carry = 0;
  if (PACKED) {  // the cell of every point, once (ncell <= 65 536): the count and scatter phases then run from LDS
#pragma unroll 4
    for (uint32_t i = tid; i < n; i += kBuildThreads) {
      const Vec3 pt = v3(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]);
      s_cellof[i] = (uint16_t)(ORDERED ? grid_morton_of_point(g, pt) : grid_cell_of_point(g, pt));
    }
  }
  for (uint32_t c_lo = 0; c_lo < ncell; c_lo += kGridLdsCells) {
    const uint32_t nc = ncell - c_lo < kGridLdsCells ? ncell - c_lo : kGridLdsCells;
    for (uint32_t c = tid; c < (PACKED ? (nc + 1) / 2 : nc); c += kBuildThreads) s_cells[c] = 0;
    __syncthreads();
    if (PACKED) {
      for (uint32_t i = tid; i < n; i += kBuildThreads) {
        const uint32_t cell = (uint32_t)s_cellof[i] - c_lo;
        if (cell < nc) (void)cell_add(cell);
      }
    } else {
#pragma unroll 4
      for (uint32_t i = tid; i < n; i += kBuildThreads) {
        const Vec3 pt = v3(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]);
        const uint32_t cell = (ORDERED ? grid_morton_of_point(g, pt) : grid_cell_of_point(g, pt)) - c_lo;
        if (cell < nc) (void)cell_add(cell);
      }
    }
    __syncthreads();
    STAMP();  // clear + count
    // exclusive scan of s_cells[0..nc): contiguous chunk per thread + block scan of chunk sums.
    // The chunk length is odd, so the 64 lanes of a wavefront walk 64 different LDS banks (an even
    // length such as 32 would put every lane on the same bank).
    // (PACKED: an odd number of WORDS per chunk, i.e. chunks start on word boundaries and never share a word)
    const uint32_t per = PACKED ? 2u * (((nc + 2 * kBuildThreads - 1) / (2 * kBuildThreads)) | 1u) : ((nc + kBuildThreads - 1) / kBuildThreads) | 1u;
    const uint32_t c0 = tid * per < nc ? tid * per : nc, c1 = c0 + per < nc ? c0 + per : nc;
    uint32_t local = 0;
    if (PACKED) {
      // (c0 < c1: a thread past the end has c0 == c1 == nc and must not touch the last word, which an odd nc leaves
      // half used — it would count that word again and then overwrite its offsets)
      for (uint32_t w = c0 >> 1; c0 < c1 && 2 * w < c1; w++) local += (s_cells[w] & 0xFFFFu) + (s_cells[w] >> 16);  // (a half past nc is 0)
    } else {
      for (uint32_t c = c0; c < c1; c++) local += s_cells[c];
    }
    uint32_t incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave_sum[wave] = incl;
    __syncthreads();
    uint32_t wave_off = carry;
    for (int w = 0; w < wave; w++) wave_off += s_wave_sum[w];
    uint32_t pass_total = 0;
    for (int w = 0; w < kBuildThreads / 64; w++) pass_total += s_wave_sum[w];
    uint32_t run = wave_off + incl - local;
    if (PACKED) {
      for (uint32_t w = c0 >> 1; c0 < c1 && 2 * w < c1; w++) {
        const uint32_t v = s_cells[w], lo = v & 0xFFFFu, hi = v >> 16;
        s_cells[w] = run | ((run + lo) << 16);
        run += lo + hi;
      }
    } else {
      for (uint32_t c = c0; c < c1; c++) {
        const uint32_t cnt = s_cells[c];
        s_cells[c] = run;
        run += cnt;
      }
    }
    __syncthreads();
    STAMP();  // scan
    // coalesced copy of the scan — for the sets whose table somebody reads: not the source sets that are ranked in LDS
    // (they are only ordered) and not the target sets small enough for associate_knn_brute_kernel, whose 300 points
    // would otherwise leave a table of up to 65 536 entries behind (0.5 GB of writes per 1 024-pair step)
    if (!(ORDERED && PACKED) && (ORDERED || n > kBruteMax))
      for (uint32_t c = tid; c < nc; c += kBuildThreads) cs[c_lo + c] = cell_get(c);
    __syncthreads();
    STAMP();  // table write
    if (PACKED) {
      for (uint32_t i = tid; i < n; i += kBuildThreads) {
        const uint32_t cell = (uint32_t)s_cellof[i] - c_lo;
        if (cell < nc) s_inv[cell_add(cell)] = (uint16_t)i;
      }
    } else {
#pragma unroll 4
      for (uint32_t i = tid; i < n; i += kBuildThreads) {
        const double x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
        const uint32_t cell = (ORDERED ? grid_morton_of_point(g, v3(x, y, z)) : grid_cell_of_point(g, v3(x, y, z))) - c_lo;
        if (cell < nc) dst[cell_add(cell)] = GridPoint{x, y, z, i, 0u};
      }
    }
    carry += pass_total;
    __syncthreads();
    STAMP();  // scatter
  }
  if (tid == 0) cs[ncell] = n;
  // roofline bytes of the index builds (SURVEY 8d style: what has to move at least): 24 B read per point, the 32-byte cell-
  // ordered copy written, + 12 B of float copies and the cell table (4 B per cell) for a target set that is searched by cells
  if (tid == 0 && bytes)
    atomicAdd(bytes, (unsigned long long)n * (ORDERED ? 56ull : (gs.rel ? 68ull : 56ull)) +
                         ((!(ORDERED && PACKED) && (ORDERED || n > kBruteMax)) ? 4ull * (ncell + 1) : 0ull));
  if (PACKED && ORDERED) {
    // Reproducible order inside a cell: position = cell begin + number of cell mates with a smaller index (the cursors hold the
    // cell ends now). Round 5: the point's cell comes from the LDS list of the count phase (it was recomputed from the point,
    // a second gather of the set), and the point leaves for its final place at once — a thread's entry p lies in the cell
    // segment [b, en) it is ranked in, so neighbouring threads still write neighbouring 32-byte points — instead of through
    // a second LDS list and a pass of its own: one gather and one phase fewer (rank + output 38 + 55 us -> see EXPERIMENTS.md).
#pragma unroll 2
    for (uint32_t p = tid; p < n; p += kBuildThreads) {
      const uint32_t i = s_inv[p];
      const uint32_t cell = s_cellof[i];
      const uint32_t b = cell ? cell_get(cell - 1) : 0u, en = cell_get(cell);
      const double x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
      uint32_t rank = 0;
      for (uint32_t j = b; j < en; j++) rank += s_inv[j] < i ? 1u : 0u;
      sp[b + rank] = GridPoint{x, y, z, i, 0u};
    }
  } else if (PACKED) {
    const uint16_t* order = s_inv;
    STAMP();  // rank
    float* __restrict__ rel = gs.rel ? gs.rel + pair * 3 * gs.stride : nullptr;
#pragma unroll 4
    for (uint32_t p = tid; p < n; p += kBuildThreads) {
      const uint32_t i = order[p];
      const double x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
      sp[p] = GridPoint{x, y, z, i, 0u};
      if (!ORDERED && rel) rel[p] = (float)(x - g.ox), rel[gs.stride + p] = (float)(y - g.oy), rel[2 * gs.stride + p] = (float)(z - g.oz);
    }
    if (!ORDERED && rel && (uint32_t)tid < kGridPad) rel[n + tid] = kRelPad, rel[gs.stride + n + tid] = kRelPad, rel[2 * gs.stride + n + tid] = kRelPad;
  } else if (!ORDERED && gs.rel) {
    // single-precision offsets from the grid origin, SoA (FP32 pre-selection of the k-NN): one coalesced
    // pass over the finished cell order instead of three scattered 4-byte writes per point
    __syncthreads();
    __threadfence_block();
    float* __restrict__ rel = gs.rel + pair * 3 * gs.stride;
    for (uint32_t p = tid; p < n; p += kBuildThreads) {
      const GridPoint e = sp[p];
      rel[p] = (float)(e.x - g.ox), rel[gs.stride + p] = (float)(e.y - g.oy), rel[2 * gs.stride + p] = (float)(e.z - g.oz);
    }
    if ((uint32_t)tid < kGridPad) rel[n + tid] = kRelPad, rel[gs.stride + n + tid] = kRelPad, rel[2 * gs.stride + n + tid] = kRelPad;
  }
#ifdef LOAMX_BUILD_PROFILE
  STAMP();
  if (tid == 0 && pair == 700 && n > 2000 && ns >= 8)
    printf("build ORDERED=%d n=%u ncell=%u ns=%d: bbox %llu choice %llu count %llu scan %llu table %llu scatter %llu rank %llu out %llu (x10 ns)\n", (int)ORDERED, n, ncell, ns,
           stamp[1] - stamp[0], stamp[2] - stamp[1], stamp[3] - stamp[2], stamp[4] - stamp[3], stamp[5] - stamp[4], stamp[6] - stamp[5], stamp[ns - 2] - stamp[ns - 3], stamp[ns - 1] - stamp[ns - 2]);
#endif
#undef STAMP
}

/* ------------------------------------------------------------------------------------------------
 * Brute-force sized sets (round 5). A target set of at most kBruteMax points — the edge features of a scan — is searched
 * by associate_knn_brute_kernel: every query scans all of it, its cell table is never read, and the order of the source
 * queries does not matter for it either. grid_build_kernel gave each such set a 1 024-thread workgroup with 144 KB of LDS,
 * i.e. a whole compute unit that a planar build could not share (two edge builds: ~120 us of the builds' 1.0 ms per step).
 * This kernel builds BOTH sets of such a pair with 256 threads and a few hundred bytes of LDS, next to the planar builds:
 * target = the same GridDesc as grid_build_kernel (grid_choose over the same box), the points in their given order with
 * their float offsets; source = the points in their given order.
 * ---------------------------------------------------------------------------------------------- */
constexpr int kSmallThreads = 256;
__global__ __launch_bounds__(kSmallThreads) void small_sets_build_kernel(const double* __restrict__ tgt_base, const uint32_t* __restrict__ n_tgt,
                                                                         const double* __restrict__ src_base, const uint32_t* __restrict__ n_src,
                                                                         size_t stride, uint32_t in_pitch, double max_dist, GridSet tgt_gs, GridSet src_gs,
                                                                         unsigned long long* __restrict__ bytes, const unsigned long long* __restrict__ box_min,
                                                                         const unsigned long long* __restrict__ box_max, const uint32_t* __restrict__ box_bad) {
  __shared__ double s_red[6][kSmallThreads / 64];
  __shared__ GridDesc s_g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t pair = blockIdx.x;
  if (n_tgt == nullptr) {  // source sets only (the target lives in a persistent index whose set of this kind is brute-force sized)
    const uint32_t ns_raw = n_src[pair * in_pitch];
    const uint32_t ns = ns_raw < stride ? ns_raw : (uint32_t)stride;
    const double* __restrict__ qp = src_base + pair * in_pitch * stride * 3;
    GridPoint* __restrict__ ssp = src_gs.sorted + pair * src_gs.stride;
    for (uint32_t i = tid; i < ns; i += kSmallThreads) ssp[i] = GridPoint{qp[3 * (size_t)i], qp[3 * (size_t)i + 1], qp[3 * (size_t)i + 2], i, 0u};
    if (tid == 0 && bytes) atomicAdd(bytes, (unsigned long long)ns * 56ull);
    return;
  }
  const uint32_t nt_raw = n_tgt[pair * in_pitch];
  if (nt_raw > kBruteMax) return;  // uniform: grid_build_kernel takes this pair's sets
  const uint32_t nt = nt_raw < stride ? nt_raw : (uint32_t)stride;
  const double* __restrict__ tp = tgt_base + pair * in_pitch * stride * 3;
  const bool have_box = box_min != nullptr && __hip_atomic_load(box_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;  // uniform
  if (!have_box) {
    double lx = kDblMax, ly = kDblMax, lz = kDblMax, hx = -kDblMax, hy = -kDblMax, hz = -kDblMax;
    for (uint32_t i = tid; i < nt; i += kSmallThreads) {
      const double x = tp[3 * (size_t)i], y = tp[3 * (size_t)i + 1], z = tp[3 * (size_t)i + 2];
      lx = fmin(lx, x), ly = fmin(ly, y), lz = fmin(lz, z);
      hx = fmax(hx, x), hy = fmax(hy, y), hz = fmax(hz, z);
    }
    lx = wave_min(lx), ly = wave_min(ly), lz = wave_min(lz);
    hx = wave_max(hx), hy = wave_max(hy), hz = wave_max(hz);
    if (lane == 0) {
      s_red[0][wave] = lx, s_red[1][wave] = ly, s_red[2][wave] = lz;
      s_red[3][wave] = hx, s_red[4][wave] = hy, s_red[5][wave] = hz;
    }
  }
  __syncthreads();
  if (tid == 0) {
    double a[6];
    if (have_box) {
      for (int k = 0; k < 3; k++) {
        a[k] = nt ? key_dbl(box_min[(pair * in_pitch) * 6 + k]) : kDblMax;
        a[3 + k] = nt ? key_dbl(box_max[(pair * in_pitch) * 6 + k]) : -kDblMax;
      }
    } else {
      for (int k = 0; k < 6; k++) {
        a[k] = s_red[k][0];
        for (int w = 1; w < kSmallThreads / 64; w++) a[k] = k < 3 ? fmin(a[k], s_red[k][w]) : fmax(a[k], s_red[k][w]);
      }
    }
    GridDesc g;
    grid_choose(g, v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), nt, max_dist, kGridCellsCap);
    s_g = g;
    tgt_gs.desc[pair] = g;
  }
  __syncthreads();
  const GridDesc g = s_g;
  GridPoint* __restrict__ sp = tgt_gs.sorted + pair * tgt_gs.stride;
  float* __restrict__ rel = tgt_gs.rel ? tgt_gs.rel + pair * 3 * tgt_gs.stride : nullptr;
  for (uint32_t p = tid; p < nt; p += kSmallThreads) {
    const double x = tp[3 * (size_t)p], y = tp[3 * (size_t)p + 1], z = tp[3 * (size_t)p + 2];
    sp[p] = GridPoint{x, y, z, p, 0u};
    if (rel) rel[p] = (float)(x - g.ox), rel[tgt_gs.stride + p] = (float)(y - g.oy), rel[2 * tgt_gs.stride + p] = (float)(z - g.oz);
  }
  if (rel && (uint32_t)tid < kGridPad) rel[nt + tid] = kRelPad, rel[tgt_gs.stride + nt + tid] = kRelPad, rel[2 * tgt_gs.stride + nt + tid] = kRelPad;
  // the source set: its points in their given order
  const uint32_t ns_raw = n_src[pair * in_pitch];
  const uint32_t ns = ns_raw < stride ? ns_raw : (uint32_t)stride;
  const double* __restrict__ qp = src_base + pair * in_pitch * stride * 3;
  GridPoint* __restrict__ ssp = src_gs.sorted + pair * src_gs.stride;
  for (uint32_t i = tid; i < ns; i += kSmallThreads) ssp[i] = GridPoint{qp[3 * (size_t)i], qp[3 * (size_t)i + 1], qp[3 * (size_t)i + 2], i, 0u};
  if (tid == 0 && bytes) atomicAdd(bytes, (unsigned long long)nt * (rel ? 68ull : 56ull) + (unsigned long long)ns * 56ull);
}

/* ------------------------------------------------------------------------------------------------
 * Map-sized target sets (more than kGridSmallCap points: a local map of 10^5 - 10^6 points, BASELINE config 5,
 * or the planar features of a 128 x 2048 scan): the same counting sort spread over the whole chip — the
 * single-workgroup kernel above spent 4.4 ms on a 1 M-point map. Bounding box and cell counts go through global
 * atomics (order-independent integers), the 65 536-entry scan is one workgroup, the scatter takes its position
 * from a global per-cell cursor. Per pair, the scratch area holds the six box keys and the cursors.
 * ---------------------------------------------------------------------------------------------- */
constexpr int kBigThreads = 256, kBigItems = 16, kBigChunk = kBigThreads * kBigItems;
constexpr size_t kBigScratchBytes = kGridBigScratchBytes;  // per pair: box keys, cursors (loamx_internal.h)

__device__ __forceinline__ unsigned char* big_scratch(GridPoint* scratch, size_t pair, size_t stride) {
  return reinterpret_cast<unsigned char*>(scratch + pair * stride);
}

__global__ void gridbig_init_kernel(GridPoint* scratch, size_t stride) {
  unsigned long long* box = reinterpret_cast<unsigned long long*>(big_scratch(scratch, blockIdx.x, stride));
  if (threadIdx.x < 6) box[threadIdx.x] = threadIdx.x < 3 ? ~0ull : 0ull;  // minima, maxima
}

__global__ __launch_bounds__(kBigThreads) void gridbig_bbox_kernel(const double* __restrict__ pts_base, const uint32_t* __restrict__ n_pts,
                                                                   size_t stride, uint32_t in_pitch, GridPoint* scratch) {
  const size_t pair = blockIdx.y;
  const uint32_t n_raw = n_pts[pair * in_pitch], n = n_raw < stride ? n_raw : (uint32_t)stride;
  const uint32_t base = blockIdx.x * kBigChunk;
  if (base >= n) return;  // uniform
  const double* __restrict__ pts = pts_base + pair * in_pitch * stride * 3;
  double lx = kDblMax, ly = kDblMax, lz = kDblMax, hx = -kDblMax, hy = -kDblMax, hz = -kDblMax;
#pragma unroll 4
  for (int k = 0; k < kBigItems; k++) {
    const uint32_t i = base + k * kBigThreads + threadIdx.x;
    if (i < n) {
      const double x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
      lx = fmin(lx, x), ly = fmin(ly, y), lz = fmin(lz, z);
      hx = fmax(hx, x), hy = fmax(hy, y), hz = fmax(hz, z);
    }
  }
  lx = wave_min(lx), ly = wave_min(ly), lz = wave_min(lz);
  hx = wave_max(hx), hy = wave_max(hy), hz = wave_max(hz);
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* box = reinterpret_cast<unsigned long long*>(big_scratch(scratch, pair, stride));
    atomicMin(&box[0], dbl_key(lx)), atomicMin(&box[1], dbl_key(ly)), atomicMin(&box[2], dbl_key(lz));
    atomicMax(&box[3], dbl_key(hx)), atomicMax(&box[4], dbl_key(hy)), atomicMax(&box[5], dbl_key(hz));
  }
}

__global__ void gridbig_choose_kernel(const uint32_t* __restrict__ n_pts, size_t stride, uint32_t in_pitch, double max_dist,
                                      GridSet gs, GridPoint* scratch, size_t n_pairs) {
  const size_t pair = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= n_pairs) return;
  const uint32_t n_raw = n_pts[pair * in_pitch], n = n_raw < stride ? n_raw : (uint32_t)stride;
  const unsigned long long* box = reinterpret_cast<const unsigned long long*>(big_scratch(scratch, pair, stride));
  GridDesc g;
  grid_choose(g, v3(key_dbl(box[0]), key_dbl(box[1]), key_dbl(box[2])), v3(key_dbl(box[3]), key_dbl(box[4]), key_dbl(box[5])), n, max_dist,
              gs.cells_cap ? gs.cells_cap : kGridCellsCap);
  gs.desc[pair] = g;
}

template <bool SCATTER>
__global__ __launch_bounds__(kBigThreads) void gridbig_pass_kernel(const double* __restrict__ pts_base, const uint32_t* __restrict__ n_pts,
                                                                   size_t stride, uint32_t in_pitch, GridSet gs, GridPoint* scratch) {
  const size_t pair = blockIdx.y;
  const uint32_t n_raw = n_pts[pair * in_pitch], n = n_raw < stride ? n_raw : (uint32_t)stride;
  const uint32_t base = blockIdx.x * kBigChunk;
  if (base >= n) return;  // uniform
  const double* __restrict__ pts = pts_base + pair * in_pitch * stride * 3;
  const GridDesc g = gs.desc[pair];
  // count: into the cell table itself; scatter: positions from the cursors (a copy of the scanned table)
  // (a table larger than kGridCellsCap belongs to a single-pair index: the pitch is not used then)
  uint32_t* __restrict__ table = SCATTER ? reinterpret_cast<uint32_t*>(big_scratch(scratch, pair, stride) + 64)
                                         : gs.cell_start + pair * (size_t)(kGridCellsCap + 1);
  GridPoint* __restrict__ sp = gs.sorted + pair * gs.stride;
#pragma unroll 4
  for (int k = 0; k < kBigItems; k++) {
    const uint32_t i = base + k * kBigThreads + threadIdx.x;
    if (i < n) {
      const double x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
      const uint32_t pos = atomicAdd(&table[grid_cell_of_point(g, v3(x, y, z))], 1u);
      if (SCATTER) sp[pos] = GridPoint{x, y, z, i, 0u};
    }
  }
}

// Exclusive scan of the cell counts of every pair in three small launches over tiles of kScanTile entries (tile sums,
// scan of the tile sums, scan inside the tiles; blockIdx.y = pair). One workgroup per pair walking its whole table —
// the first form — took 0.58 ms for the 2^18 cells of a map-sized index: 64 serial loads per thread and tile.
constexpr uint32_t kScanTile = 4096, kScanThreads = 256, kScanPer = kScanTile / kScanThreads;
__device__ __forceinline__ uint32_t block_incl_scan_256(uint32_t v, uint32_t* s_wave, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  uint32_t before = 0, all = 0;
  for (int w = 0; w < (int)(kScanThreads / 64); w++) {
    if (w < wave) before += s_wave[w];
    all += s_wave[w];
  }
  __syncthreads();
  if (total) *total = all;
  return incl + before;
}
// (the tile sums of a pair live behind its cursors in the scratch area: kGridBigScratchBytes has the room)
__device__ __forceinline__ uint32_t* big_tile_sums(GridPoint* scratch, size_t pair, size_t stride, const GridSet& gs) {
  return reinterpret_cast<uint32_t*>(big_scratch(scratch, pair, stride) + 64) + (gs.cells_cap ? gs.cells_cap : kGridCellsCap);
}
__global__ __launch_bounds__(kScanThreads) void gridbig_tile_sum_kernel(size_t stride, GridSet gs, GridPoint* scratch) {
  __shared__ uint32_t s_wave[kScanThreads / 64];
  const size_t pair = blockIdx.y;
  const GridDesc g = gs.desc[pair];
  const uint32_t ncell = (uint32_t)(g.nx * g.ny * g.nz);
  if (blockIdx.x * kScanTile >= ncell) return;  // uniform
  const uint32_t* __restrict__ cs = gs.cell_start + pair * (size_t)(kGridCellsCap + 1);
  const uint32_t c0 = blockIdx.x * kScanTile + threadIdx.x * kScanPer;
  uint32_t local = 0;
#pragma unroll
  for (uint32_t u = 0; u < kScanPer; u++)
    if (c0 + u < ncell) local += cs[c0 + u];
  uint32_t total = 0;
  (void)block_incl_scan_256(local, s_wave, &total);
  if (threadIdx.x == 0) big_tile_sums(scratch, pair, stride, gs)[blockIdx.x] = total;
}
__global__ __launch_bounds__(kScanThreads) void gridbig_tile_scan_kernel(size_t stride, GridSet gs, GridPoint* scratch) {  // one workgroup per pair
  __shared__ uint32_t s_wave[kScanThreads / 64];
  const size_t pair = blockIdx.x;
  const GridDesc g = gs.desc[pair];
  const uint32_t ncell = (uint32_t)(g.nx * g.ny * g.nz), n_tiles = (ncell + kScanTile - 1) / kScanTile;
  uint32_t* __restrict__ ts = big_tile_sums(scratch, pair, stride, gs);
  uint32_t carry = 0;
  for (uint32_t t0 = 0; t0 < n_tiles; t0 += kScanThreads) {
    const uint32_t t = t0 + threadIdx.x;
    const uint32_t v = t < n_tiles ? ts[t] : 0u;
    uint32_t total = 0;
    const uint32_t incl = block_incl_scan_256(v, s_wave, &total);
    if (t < n_tiles) ts[t] = carry + incl - v;
    carry += total;
  }
}
__global__ __launch_bounds__(kScanThreads) void gridbig_scan_kernel(const uint32_t* __restrict__ n_pts, size_t stride, uint32_t in_pitch, GridSet gs,
                                                                    GridPoint* scratch) {
  __shared__ uint32_t s_wave[kScanThreads / 64];
  const size_t pair = blockIdx.y;
  const uint32_t n_raw = n_pts[pair * in_pitch], n = n_raw < stride ? n_raw : (uint32_t)stride;
  const GridDesc g = gs.desc[pair];
  const uint32_t ncell = (uint32_t)(g.nx * g.ny * g.nz);
  if (blockIdx.x * kScanTile > ncell) return;  // uniform (the tile that holds entry ncell writes the total)
  uint32_t* __restrict__ cs = gs.cell_start + pair * (size_t)(kGridCellsCap + 1);
  uint32_t* __restrict__ cursor = reinterpret_cast<uint32_t*>(big_scratch(scratch, pair, stride) + 64);
  const uint32_t c0 = blockIdx.x * kScanTile + threadIdx.x * kScanPer;
  uint32_t cnt[kScanPer], local = 0;
#pragma unroll
  for (uint32_t u = 0; u < kScanPer; u++) cnt[u] = c0 + u < ncell ? cs[c0 + u] : 0u, local += cnt[u];
  const uint32_t tile_base = blockIdx.x * kScanTile < ncell ? big_tile_sums(scratch, pair, stride, gs)[blockIdx.x] : n;
  uint32_t run = tile_base + block_incl_scan_256(local, s_wave, nullptr) - local;
#pragma unroll
  for (uint32_t u = 0; u < kScanPer; u++) {
    const uint32_t c = c0 + u;
    if (c < ncell) cs[c] = run, cursor[c] = run;
    else if (c == ncell) cs[c] = n;
    run += cnt[u];
  }
}

__global__ __launch_bounds__(kBigThreads) void gridbig_rel_kernel(const uint32_t* __restrict__ n_pts, size_t stride, uint32_t in_pitch,
                                                                  GridSet gs) {
  const size_t pair = blockIdx.y;
  const uint32_t n_raw = n_pts[pair * in_pitch], n = n_raw < stride ? n_raw : (uint32_t)stride;
  const uint32_t p = blockIdx.x * kBigThreads + threadIdx.x;
  float* __restrict__ rel = gs.rel + pair * 3 * gs.stride;
  if (p >= n) {
    if (p < n + kGridPad) rel[p] = kRelPad, rel[gs.stride + p] = kRelPad, rel[2 * gs.stride + p] = kRelPad;
    return;
  }
  const GridDesc g = gs.desc[pair];
  const GridPoint e = gs.sorted[pair * gs.stride + p];
  rel[p] = (float)(e.x - g.ox), rel[gs.stride + p] = (float)(e.y - g.oy), rel[2 * gs.stride + p] = (float)(e.z - g.oz);
}

/* ------------------------------------------------------------------------------------------------
 * Incremental insert into a map-sized persistent index (round 3): the grid (origin, cell edge, dimensions)
 * stays, the new points are merged into the cell-sorted arrays. Work: O(new points) for the counts and the scatter, one
 * table scan, and ONE streaming copy of the old arrays into their twin buffers (44 bytes per old point each way:
 * ~30 us per million points) — no bounding-box pass, no atomics over the old points, no re-sort. The result is an
 * exactly packed index (no slack: the searches scan what a fresh build of the same grid would give them); only the
 * order of the new points inside a cell depends on the order the atomics were served, which no search result depends
 * on. A new point outside the grid raises a flag and the host rebuilds from scratch (index_build).
 *   ws = [0] flag, [16 ...) add / shift table (ncell + 1), cursor table (ncell), cell of every new point
 * ---------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(256) void index_insert_count_kernel(const double* __restrict__ add, uint32_t n_add, const GridDesc* __restrict__ desc,
                                                                 uint32_t* __restrict__ flag, uint32_t* __restrict__ add_count,
                                                                 uint32_t* __restrict__ cell_of) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_add) return;
  const GridDesc g = *desc;
  const Vec3 p = v3(add[3 * (size_t)i], add[3 * (size_t)i + 1], add[3 * (size_t)i + 2]);
  const int32_t cx = grid_cell_coord(p.x, g.ox, g.inv_h), cy = grid_cell_coord(p.y, g.oy, g.inv_h), cz = grid_cell_coord(p.z, g.oz, g.inv_h);
  // (a point outside the grid, or not finite, cannot be filed under a boundary cell: the searches prune by the cells' slabs)
  const bool inside = cx >= 0 && cx < g.nx && cy >= 0 && cy < g.ny && cz >= 0 && cz < g.nz && p.x - p.x == 0.0 && p.y - p.y == 0.0 && p.z - p.z == 0.0;
  if (!inside) {
    atomicOr(flag, 1u);
    cell_of[i] = 0u;
    return;
  }
  const uint32_t cell = (uint32_t)((cz * g.ny + cy) * g.nx + cx);
  cell_of[i] = cell;
  atomicAdd(&add_count[cell], 1u);
}

// shift[c] = number of new points in the cells before c (exclusive scan of the counts, in place) and the cursor of every
// cell's new points in the merged array: new begin + old population. Tiled like gridbig_scan_kernel.
__global__ __launch_bounds__(kScanThreads) void index_insert_tile_sum_kernel(const GridDesc* __restrict__ desc, const uint32_t* __restrict__ shift,
                                                                             uint32_t* __restrict__ tile_sum) {
  __shared__ uint32_t s_wave[kScanThreads / 64];
  const uint32_t ncell = (uint32_t)(desc->nx * desc->ny * desc->nz);
  const uint32_t c0 = blockIdx.x * kScanTile + threadIdx.x * kScanPer;
  uint32_t local = 0;
#pragma unroll
  for (uint32_t u = 0; u < kScanPer; u++)
    if (c0 + u < ncell) local += shift[c0 + u];
  uint32_t total = 0;
  (void)block_incl_scan_256(local, s_wave, &total);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}
__global__ __launch_bounds__(kScanThreads) void index_insert_tile_scan_kernel(uint32_t* __restrict__ tile_sum, uint32_t n_tiles) {  // one workgroup
  __shared__ uint32_t s_wave[kScanThreads / 64];
  uint32_t carry = 0;
  for (uint32_t t0 = 0; t0 < n_tiles; t0 += kScanThreads) {
    const uint32_t t = t0 + threadIdx.x;
    const uint32_t v = t < n_tiles ? tile_sum[t] : 0u;
    uint32_t total = 0;
    const uint32_t incl = block_incl_scan_256(v, s_wave, &total);
    if (t < n_tiles) tile_sum[t] = carry + incl - v;
    carry += total;
  }
}
__global__ __launch_bounds__(kScanThreads) void index_insert_scan_kernel(const GridDesc* __restrict__ desc, const uint32_t* __restrict__ cell_start,
                                                                         const uint32_t* __restrict__ tile_sum, uint32_t* __restrict__ shift,
                                                                         uint32_t* __restrict__ cursor) {
  __shared__ uint32_t s_wave[kScanThreads / 64];
  const uint32_t ncell = (uint32_t)(desc->nx * desc->ny * desc->nz);
  const uint32_t c0 = blockIdx.x * kScanTile + threadIdx.x * kScanPer;
  uint32_t cnt[kScanPer], local = 0;
#pragma unroll
  for (uint32_t u = 0; u < kScanPer; u++) cnt[u] = c0 + u < ncell ? shift[c0 + u] : 0u, local += cnt[u];
  uint32_t run = tile_sum[blockIdx.x] + block_incl_scan_256(local, s_wave, nullptr) - local;
#pragma unroll
  for (uint32_t u = 0; u < kScanPer; u++) {
    const uint32_t c = c0 + u;
    if (c < ncell) {
      shift[c] = run;
      cursor[c] = cell_start[c + 1] + run;  // old end of the cell, moved by the new points of the cells before it
    } else if (c == ncell) {
      shift[c] = run;  // the total
    }
    run += cnt[u];
  }
}

// every old point to its place in the twin arrays: old position + new points filed before its cell
__global__ __launch_bounds__(256) void index_insert_move_kernel(const GridDesc* __restrict__ desc, uint32_t n_old, size_t stride,
                                                                const GridPoint* __restrict__ sorted, const float* __restrict__ rel,
                                                                const uint32_t* __restrict__ shift, GridPoint* __restrict__ sorted2,
                                                                float* __restrict__ rel2) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n_old) return;
  const GridDesc g = *desc;
  const GridPoint e = sorted[p];
  const uint32_t np = p + shift[grid_cell_of_point(g, v3(e.x, e.y, e.z))];
  sorted2[np] = e;
  rel2[np] = rel[p], rel2[stride + np] = rel[stride + p], rel2[2 * stride + np] = rel[2 * stride + p];
}

__global__ __launch_bounds__(256) void index_insert_scatter_kernel(const double* __restrict__ add, uint32_t n_add, uint32_t first_index,
                                                                   const GridDesc* __restrict__ desc, size_t stride,
                                                                   const uint32_t* __restrict__ cell_of, uint32_t* __restrict__ cursor,
                                                                   GridPoint* __restrict__ sorted2, float* __restrict__ rel2) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_add) return;
  const GridDesc g = *desc;
  const double x = add[3 * (size_t)i], y = add[3 * (size_t)i + 1], z = add[3 * (size_t)i + 2];
  const uint32_t pos = atomicAdd(&cursor[cell_of[i]], 1u);
  sorted2[pos] = GridPoint{x, y, z, first_index + i, 0u};
  rel2[pos] = (float)(x - g.ox), rel2[stride + pos] = (float)(y - g.oy), rel2[2 * stride + pos] = (float)(z - g.oz);
}

// the cell table moves with the points; the point count and the float pads behind the set follow
__global__ __launch_bounds__(256) void index_insert_table_kernel(GridDesc* __restrict__ desc, uint32_t* __restrict__ cell_start,
                                                                 const uint32_t* __restrict__ shift, uint32_t n_new, size_t stride,
                                                                 float* __restrict__ rel2) {
  const uint32_t ncell = (uint32_t)(desc->nx * desc->ny * desc->nz);
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c <= ncell) cell_start[c] += shift[c];  // (entry ncell: the total)
  if (c < kGridPad) rel2[n_new + c] = kRelPad, rel2[stride + n_new + c] = kRelPad, rel2[2 * stride + n_new + c] = kRelPad;
  if (c == 0) desc->n_points = n_new;  // (read by nobody in this launch: the kernels before it took the grid by value)
}

// Second half of the ORDERED build (source sets): every point of the scratch copy is placed at
// (cell begin + number of cell mates with a smaller original index). A kernel of its own because the
// pair-wise comparison inside a cell is a latency-bound gather that wants far more waves in flight
// than the one 128 KiB-LDS workgroup per CU of grid_build_kernel can offer.
constexpr int kRankThreads = 256;
__global__ __launch_bounds__(kRankThreads) void grid_rank_kernel(const uint32_t* __restrict__ n_pts, size_t stride,
                                                                 uint32_t in_pitch, GridSet gs,
                                                                 const GridPoint* __restrict__ scratch, const uint32_t* __restrict__ small_if) {
  const size_t pair = blockIdx.y;
  if (small_if && small_if[pair * in_pitch] <= kBruteMax) return;  // (uniform: small_sets_build_kernel built this pair's set; nothing was scattered)
  const uint32_t n_raw = n_pts[pair * in_pitch];
  const uint32_t n = n_raw < stride ? n_raw : (uint32_t)stride;
  const uint32_t p = blockIdx.x * kRankThreads + threadIdx.x;
  if (p >= n) return;
  const GridDesc g = gs.desc[pair];
  const uint32_t* __restrict__ cs = gs.cell_start + pair * (size_t)(kGridCellsCap + 1);
  const GridPoint* __restrict__ src = scratch + pair * gs.stride;
  GridPoint* __restrict__ sp = gs.sorted + pair * gs.stride;
  const GridPoint e = src[p];
  const uint32_t cell = grid_morton_of_point(g, v3(e.x, e.y, e.z));
  const uint32_t b = cs[cell], en = cs[cell + 1];
  uint32_t rank = 0;
#pragma unroll 4
  for (uint32_t j = b; j < en; j++) rank += src[j].orig < e.orig ? 1u : 0u;
  sp[b + rank] = e;
}

/* ------------------------------------------------------------------------------------------------ */
__global__ void state_init_kernel(RegBatch B, RegConfig C) {
  const size_t pair = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= B.n_pairs) return;
  PairState& S = B.state[pair];
  for (int i = 0; i < 7; i++) S.est[i] = B.init ? B.init[pair * 7 + i] : (i == 3 ? 1.0 : 0.0);
  S.active = C.max_iterations > 0 ? 1u : 0u;
  S.termination = LOAMX_MAX_ITER;  // registration-inl.h:27
  S.iterations = 0;
  S.first_sweep = 0;
  S.stream_planes = 1;
  S.use_moments = 0;
  S.mom_ref_on = 0;
  S.lm.active = 0;
  for (int c = 0; c < 6; c++) B.assoc.n_assoc[8 * pair + c] = 0;
  if (B.max_counts) {  // the host sizes the association grids by the largest set instead of by the capacity
    const uint32_t ne = B.n_src_edge[pair * B.in_pitch], np = B.n_src_planar[pair * B.in_pitch];
    atomicMax(&B.max_counts[0], ne < B.edge_stride ? ne : (uint32_t)B.edge_stride);
    atomicMax(&B.max_counts[1], np < B.planar_stride ? np : (uint32_t)B.planar_stride);
    if (B.n_tgt_edge && B.n_tgt_planar) {  // (a persistent index has its sizes on the host)
      uint32_t te = B.n_tgt_edge[pair * B.in_pitch], tp = B.n_tgt_planar[pair * B.in_pitch];
      te = te < B.edge_stride ? te : (uint32_t)B.edge_stride, tp = tp < B.planar_stride ? tp : (uint32_t)B.planar_stride;
      atomicMax(&B.max_counts[2], te), atomicMax(&B.max_counts[3], tp);
      atomicMin(&B.max_counts[4], te), atomicMin(&B.max_counts[5], tp);
    }
  }
}

constexpr int kAssocThreads = 256;
#ifndef LOAMX_ASSOC_WAVES
#define LOAMX_ASSOC_WAVES 4  // measured: 2 -> 12.3 ms, 3 -> 9.3, 4 -> 8.1 (320 B scratch), 5+ spills badly
#endif
#ifndef LOAMX_KNN_REQUERY
#define LOAMX_KNN_REQUERY 1
#endif
#ifndef LOAMX_ASSOC_WAVES5
#define LOAMX_ASSOC_WAVES5 (LOAMX_KNN_REQUERY ? 6 : 4)  // the round-1 kernels of k <= 5 (knn_round1_body: 80 registers with the requery)
#endif

// Workgroup -> (pair, chunk) mapping: workgroups are dealt round-robin over the 8 XCDs, so all
// chunks of one pair are given ids with the same id % 8 and share one XCD's L2 (the pair's target
// index + points are ~0.5 MB). With fewer than 8 pairs (single registrations, scan-to-map) that would
// leave XCDs idle, so the chunks are spread over all of them instead. Placement only affects speed,
// never results. (Grids are sized ceil(n_pairs / 8) * 8 * blocks_per_pair for both mappings.)
__device__ __forceinline__ bool xcd_pair_map(uint32_t block, uint32_t blocks_per_pair, size_t n_pairs, size_t& pair,
                                             uint32_t& chunk) {
  if (n_pairs < 8) {
    pair = block / blocks_per_pair;
    chunk = block % blocks_per_pair;
  } else {
    const uint32_t xcd = block & 7u, slot = block >> 3;
    pair = (size_t)xcd + 8u * (size_t)(slot / blocks_per_pair);
    chunk = slot % blocks_per_pair;
  }
  return pair < n_pairs;
}

// Small target sets (a few hundred edge features per scan) are searched exhaustively: the whole set
// streams through LDS in tiles and every lane scans all of it — no divergence, no dependent loads,
// and none of the empty-cell rounds a grid search spends on sparse sets. Same collectors, same result.
constexpr int kBruteTile = 256;
// LDS of a brute-force workgroup: one tile of 32-byte points (FP64 passes) and behind it three rows of floats (FP32 pass)
constexpr size_t kBruteLdsBytes = (kBruteTile + kGridPad) * sizeof(GridPoint) + ((3 * (kBruteTile + kGridPad) * sizeof(float) + 31) / 32) * 32;

template <class Coll>
__device__ __forceinline__ void brute_scan_tile(Coll& c, int k, Vec3 q, const GridPoint* s_tile, uint32_t tile, uint32_t tn) {
  for (uint32_t u = 0; u < tn; u += 4) {
    const GridPoint t0 = s_tile[u], t1 = s_tile[u + 1], t2 = s_tile[u + 2], t3 = s_tile[u + 3];
    // nanoflann L2_Simple: ((dx^2 + dy^2) + dz^2), as knn_scan_batch
    double dx = q.x - t0.x, dy = q.y - t0.y, dz = q.z - t0.z;
    const double d0 = dx * dx + dy * dy + dz * dz;
    dx = q.x - t1.x, dy = q.y - t1.y, dz = q.z - t1.z;
    const double d1 = dx * dx + dy * dy + dz * dz;
    dx = q.x - t2.x, dy = q.y - t2.y, dz = q.z - t2.z;
    const double d2 = dx * dx + dy * dy + dz * dz;
    dx = q.x - t3.x, dy = q.y - t3.y, dz = q.z - t3.z;
    const double d3 = dx * dx + dy * dy + dz * dz;
    knn_offer4(c, k, d0, d1, d2, d3, tile + u, tn - u, t0.orig, t1.orig, t2.orig, t3.orig);
  }
}

// (bodies of the k-NN kernels as functions of a workgroup number: associate_knn_mixed_kernel runs both kinds of workgroup
// in one launch)
template <bool PLANE, int KM>
__device__ __forceinline__ void knn_brute_body(const RegBatch& B, const RegConfig& C, uint32_t blocks_per_pair, uint32_t block,
                                               GridPoint* s_tile) {
  size_t pair;
  uint32_t chunk;
  if (!xcd_pair_map(block, blocks_per_pair, B.n_pairs, pair, chunk)) return;
  const PairState& S = B.state[pair];
  if (!S.active) return;  // uniform per workgroup
  const GridSet& gs = PLANE ? B.grid_plane : B.grid_edge;
  const uint32_t n_tgt = gs.desc[pair].n_points;
  if (n_tgt > kBruteMax) return;  // uniform: the grid kernels take this pair
  const size_t stride = PLANE ? B.planar_stride : B.edge_stride;
  const uint32_t n_src = PLANE ? B.n_src_planar[pair * B.in_pitch] : B.n_src_edge[pair * B.in_pitch];
  if ((PLANE ? B.knn_mode_plane : B.knn_mode_edge) == 2u && chunk == 0 && threadIdx.x == 0 && B.assoc_slots)  // (the grid kernel was not launched)
    atomicAdd(&B.assoc_slots[PLANE ? 1 : 0], (unsigned long long)(n_src < stride ? n_src : stride));
  if (chunk * kAssocThreads >= n_src || chunk * kAssocThreads >= stride) return;  // uniform: no query here
  const uint32_t i = chunk * kAssocThreads + threadIdx.x;
  const bool has = i < n_src && i < stride;
  const GridSet& src_gs = PLANE ? B.src_grid_plane : B.src_grid_edge;
  const GridPoint* __restrict__ sp = gs.sorted + pair * gs.stride;
  Vec3 p = v3(0, 0, 0);
  if (has) {
    const GridPoint sq = src_gs.sorted[pair * src_gs.stride + i];
    p = pose_act(S.est, v3(sq.x, sq.y, sq.z));  // registration.cpp:34 / :75
  }
  int k = PLANE ? C.k_plane : C.k_edge;
  k = k < KM ? k : KM;
  const double max_dist = PLANE ? C.r_plane : C.r_edge;
  uint32_t pos[KM];
#pragma unroll
  for (int j = 0; j < KM; j++) pos[j] = 0;
  int kept = 0;
  if (k > 0 && n_tgt > 0) {
    const double pass_max = PLANE ? C.pass_plane : C.pass_edge;
    // ---- FP32 pre-selection, as in the grid kernel: 32-bit keys (float bits of d2, the low 9 bits = the candidate's
    // position: the set has at most 512 points) through the v_med3_u32 collector, 10 instructions per candidate instead
    // of 25; then the k selected are fetched in FP64 and the usual three checks (strictly ascending exact distances, best
    // rejected key beyond the FP32 error, finite) decide whether the answer stands. Otherwise: the FP64 passes below.
    kept = has ? -1 : 0;
    const GridDesc g = gs.desc[pair];
    if (gs.rel != nullptr && g.h > 0.0) {
      float* s_f = reinterpret_cast<float*>(s_tile + kBruteTile + kGridPad);  // three rows of kBruteTile + kGridPad floats
      constexpr int kRow = kBruteTile + kGridPad;
      const float* __restrict__ rel = gs.rel + pair * 3 * gs.stride;
      const double a = knn_f32_err_unit(g);
      const bool inside = has && grid_outside_distance(g, grid_cell_coord(p.x, g.ox, g.inv_h), grid_cell_coord(p.y, g.oy, g.inv_h),
                                                       grid_cell_coord(p.z, g.oz, g.inv_h)) <= 1;
      const float qx = (float)(p.x - g.ox), qy = (float)(p.y - g.oy), qz = (float)(p.z - g.oz);
      constexpr uint32_t imask = 0x1FFu;  // (positions 0 .. kBruteMax - 1)
      static_assert(kBruteMax <= 512, "nine position bits");
      KnnKeys32<KM> c;
      knn_init(c, k);
      for (uint32_t tile = 0; tile < n_tgt; tile += kBruteTile) {
        const uint32_t tn = n_tgt - tile < (uint32_t)kBruteTile ? n_tgt - tile : (uint32_t)kBruteTile;
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < tn + kGridPad; t += kAssocThreads) {
          s_f[t] = rel[tile + t], s_f[kRow + t] = rel[gs.stride + tile + t], s_f[2 * kRow + t] = rel[2 * gs.stride + tile + t];
        }
        __syncthreads();
        if (inside)
          for (uint32_t u = 0; u < tn; u += 4)
            knn_collect_batch_f32<KM>(c, qx, qy, qz, *reinterpret_cast<const KnnF4*>(s_f + u), *reinterpret_cast<const KnnF4*>(s_f + kRow + u),
                                      *reinterpret_cast<const KnnF4*>(s_f + 2 * kRow + u), tn - u, tile + u, ~imask);
      }
      if (inside) {
        int count = 0, kk = 0;
        bool undecided = false, open = true;
        double prev = -1.0, d5 = 0.0;
        GridPoint tp[KM];
#pragma unroll
        for (int j = 0; j < KM; j++) {
          const uint32_t pp = c.key[j] & imask;
          pos[j] = j >= KM - k && c.key[j] != 0xFFFFFFFFu ? pp : 0u;
          tp[j] = sp[pp < n_tgt ? pp : 0u];
        }
#pragma unroll
        for (int j = 0; j < KM; j++) {
          const uint32_t key = c.key[j];
          if (j >= KM - k && key != 0xFFFFFFFFu) {
            if (key >= 0x7F800000u) undecided = true;
            const double dx = p.x - tp[j].x, dy = p.y - tp[j].y, dz = p.z - tp[j].z;
            const double d2 = dx * dx + dy * dy + dz * dz;  // as knn_scan_batch
            if (!(d2 > prev) || !(d2 <= kDblMax)) undecided = true;  // a tie, an inversion, not finite
            prev = d2, d5 = d2;
            count++;
            if (open) {
              if (d2 <= pass_max) kk++;
              else open = false;
            }
          }
        }
        const uint32_t k6 = c.key[KM];
        if (count == k && k6 != 0xFFFFFFFFu) {
          const double t6 = (double)knn_bits_f32(k6 & ~imask);
          const double err = 2.0 * (3.4641016151377544 * a * sqrt(d5) + 3.0 * a * a + 2.384185791015625e-7 * d5);  // x2 safety (as knn_lean_finish; the key's position bits only lower t6)
          if (!(t6 > d5 + err)) undecided = true;
        }
        if (!undecided) kept = kk;
      }
    }
    if (__syncthreads_or(kept < 0)) {  // FP64 keyed collector for the lanes the pre-selection could not finish
      KnnKeys<KM> c;
      knn_init(c, k, n_tgt);
      for (uint32_t tile = 0; tile < n_tgt; tile += kBruteTile) {
        const uint32_t tn = n_tgt - tile < (uint32_t)kBruteTile ? n_tgt - tile : (uint32_t)kBruteTile;
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < tn + kGridPad; t += kAssocThreads) s_tile[t] = sp[tile + t];  // (the set has kGridPad spare entries)
        __syncthreads();
        if (kept < 0) brute_scan_tile(c, k, p, s_tile, tile, tn);
      }
      if (kept < 0) kept = knn_keys_finish(c, k, pass_max, pos);
    }
    if (__syncthreads_or(kept < 0)) {  // undecided keys somewhere in the workgroup: exact collector for those lanes
      KnnResult<KM> r;
      knn_init(r);
      for (uint32_t tile = 0; tile < n_tgt; tile += kBruteTile) {
        const uint32_t tn = n_tgt - tile < (uint32_t)kBruteTile ? n_tgt - tile : (uint32_t)kBruteTile;
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < tn + kGridPad; t += kAssocThreads) s_tile[t] = sp[tile + t];
        __syncthreads();
        if (kept < 0) brute_scan_tile(r, k, p, s_tile, tile, tn);
      }
      if (kept < 0) {
        kept = knn_finish(r, k, max_dist);
        knn_shift_positions<KM>(r.pos, k, pos);
      }
    }
  }
  if (!has) return;
  const size_t field = B.n_pairs * stride, slot = pair * stride + i;
  uint32_t* __restrict__ nn = PLANE ? B.assoc.nn_plane : B.assoc.nn_edge;
  nn[slot] = (uint32_t)kept;
#pragma unroll
  for (int j = 0; j < KM; j++) nn[(1 + j) * field + slot] = pos[j];
}
template <bool PLANE, int KM>
__global__ __launch_bounds__(kAssocThreads) void associate_knn_brute_kernel(RegBatch B, RegConfig C, uint32_t blocks_per_pair) {
  __shared__ GridPoint s_tile[kBruteLdsBytes / sizeof(GridPoint)];
  knn_brute_body<PLANE, KM>(B, C, blocks_per_pair, blockIdx.x, s_tile);
}

// entries of the queues rest_*: a query index and why round 1 queued it
constexpr uint32_t kQueueIndex = 0x3FFFFFFFu, kQueueWide = 0x80000000u, kQueueTied = 0x40000000u;

// Association is split in two kernels so that each runs at its own register budget:
//   associate_knn_kernel : the latency-bound grid walk; writes the neighbour count and the positions
//                          (in the cell-sorted target array) of the k nearest, ascending.
//   associate_fit_kernel : pure FP64 arithmetic — gathers the neighbours, fitLine / fitPlane, guards,
//                          writes the association record.
template <bool PLANE, int KM>
__device__ __forceinline__ void knn_round1_body(const RegBatch& B, const RegConfig& C, uint32_t blocks_per_pair, uint32_t block,
                                                uint32_t* s_rows) {
  size_t pair;
  uint32_t chunk;
  if (!xcd_pair_map(block, blocks_per_pair, B.n_pairs, pair, chunk)) return;
  const uint32_t i = chunk * kAssocThreads + threadIdx.x;
  const PairState& S = B.state[pair];
  if (!S.active) return;  // uniform per workgroup
  const size_t stride = PLANE ? B.planar_stride : B.edge_stride;
  const uint32_t n_src = PLANE ? B.n_src_planar[pair * B.in_pitch] : B.n_src_edge[pair * B.in_pitch];
  const GridSet& gs = PLANE ? B.grid_plane : B.grid_edge;
  const GridSet& src_gs = PLANE ? B.src_grid_plane : B.src_grid_edge;
  if (chunk == 0 && threadIdx.x == 0 && B.assoc_slots)
    atomicAdd(&B.assoc_slots[PLANE ? 1 : 0], (unsigned long long)(n_src < stride ? n_src : stride));
  const GridDesc g = gs.desc[pair];
  if (g.n_points <= kBruteMax) return;  // uniform: small target sets belong to associate_knn_brute_kernel
  if (i >= n_src || i >= stride) return;
  // queries are taken in the source set's own cell order: neighbouring lanes look at neighbouring
  // target cells (shared cache lines, similar trip counts)
  const GridPoint sq = src_gs.sorted[pair * src_gs.stride + i];
  const Vec3 p = pose_act(S.est, v3(sq.x, sq.y, sq.z));  // registration.cpp:34 / :75
  const uint32_t* __restrict__ cs = gs.cell_start + pair * (size_t)(kGridCellsCap + 1);
  const GridPoint* __restrict__ sp = gs.sorted + pair * gs.stride;
  uint32_t pos[KM];  // (s_rows: per-thread row lists of knn_lean_round1, [word][thread], conflict free)
  const float* __restrict__ rel = gs.rel + pair * 3 * gs.stride;
#if LOAMX_KNN_REQUERY
  // (round 5) the verification behind the walk takes the query point from a second load + transform instead of from nine
  // registers held through the candidate loop: 89 -> 80 registers, a sixth wavefront per SIMD without scratch (the recomputation
  // alone, at five: 1.155 -> 1.195 ms; with the sixth wavefront: 1.139 ms)
  const GridPoint* __restrict__ src_pts = src_gs.sorted + pair * src_gs.stride;
  auto requery = [&](const Vec3&) -> Vec3 {
    uint32_t ii = i;
    asm volatile("" : "+v"(ii));  // (opaque: a second load, not the first one's registers kept)
    const GridPoint s2 = src_pts[ii];
    return pose_act(S.est, v3(s2.x, s2.y, s2.z));
  };
  const int kept = knn_search_f32_round1<KM, false, decltype(requery)>(g, cs, sp, rel, (uint32_t)gs.stride, p, PLANE ? C.k_plane : C.k_edge,
                                             PLANE ? C.r_plane : C.r_edge, PLANE ? C.pass_plane : C.pass_edge, pos,
                                             s_rows + threadIdx.x, kAssocThreads, requery);
#else
  const int kept = knn_search_f32_round1<KM>(g, cs, sp, rel, (uint32_t)gs.stride, p, PLANE ? C.k_plane : C.k_edge,
                                             PLANE ? C.r_plane : C.r_edge, PLANE ? C.pass_plane : C.pass_edge, pos,
                                             s_rows + threadIdx.x, kAssocThreads);
#endif
  const size_t field = B.n_pairs * stride, slot = pair * stride + i;
  uint32_t* __restrict__ nn = PLANE ? B.assoc.nn_plane : B.assoc.nn_edge;  // [1 + KM][n_pairs * stride]
  nn[slot] = kept < 0 ? 0xFFFFFFFFu : (uint32_t)kept;
  if (kept < 0) {  // not finished by round 1: queued for associate_knn_rest_kernel
    const uint32_t at = atomicAdd(&B.assoc.n_assoc[8 * pair + (PLANE ? 3 : 2)], 1u);
    // (bit 31: the query only ran out of 8-bit running numbers — a dense block: the queue kernel retries the FP32
    // pre-selection with wide numbers; bit 30: its FP32 keys were tied — an ordinary neighbourhood the FP64 keys decide)
    (PLANE ? B.assoc.rest_plane : B.assoc.rest_edge)[pair * stride + at] = i | (kept == -2 ? kQueueWide : kept == -3 ? kQueueTied : 0u);
  }
#pragma unroll
  for (int j = 0; j < KM; j++) nn[(1 + j) * field + slot] = pos[j];  // neighbour j is slot (KM - k) + j
}
template <bool PLANE, int KM>
__global__ __launch_bounds__(kAssocThreads, KM <= 5 ? LOAMX_ASSOC_WAVES5 : LOAMX_ASSOC_WAVES) void associate_knn_kernel(RegBatch B, RegConfig C,
                                                                                         uint32_t blocks_per_pair) {
  __shared__ uint32_t s_rows[kLeanRowWords * kAssocThreads];
  knn_round1_body<PLANE, KM>(B, C, blocks_per_pair, blockIdx.x, s_rows);
}
// The edge sets of a scan (<= 512 points: brute force) and its planar sets (grid search) in ONE launch: the first
// `edge_blocks` workgroups are the brute-force kernel's, the rest the round-1 kernel's. As two kernels on two streams the
// edge chain cost the plane kernel about half of its own duration (1.07 -> 1.22 ms); as extra workgroups of the same
// dispatch it costs its instructions. One LDS buffer serves both kinds (the row lists are the larger).
template <int KME, int KMP>
__global__ __launch_bounds__(kAssocThreads, (KME <= 5 && KMP <= 5) ? LOAMX_ASSOC_WAVES5 : LOAMX_ASSOC_WAVES) void associate_knn_mixed_kernel(RegBatch B, RegConfig C, uint32_t blocks_edge,
                                                                                               uint32_t blocks_plane, uint32_t edge_blocks) {
  constexpr size_t kRowBytes = sizeof(uint32_t) * kLeanRowWords * kAssocThreads, kTileBytes = kBruteLdsBytes;
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[kRowBytes > kTileBytes ? kRowBytes : kTileBytes];
  if (blockIdx.x < edge_blocks) knn_brute_body<false, KME>(B, C, blocks_edge, blockIdx.x, reinterpret_cast<GridPoint*>(s_raw));
  else knn_round1_body<true, KMP>(B, C, blocks_plane, blockIdx.x - edge_blocks, reinterpret_cast<uint32_t*>(s_raw));
}

// The queue chain. Round 1 queues 1-7 % of the plane queries, nearly all of them because their 3x3x3 block does not hold
// k points closer than its faces (sparse neighbourhoods). Three kernels of 64-thread workgroups and at most ~100
// registers, so that their wavefronts find room next to the plane fit kernel, which runs at the same time:
//   associate_knn_rest_kernel  : the lean FP32 search again, over the 5x5x5 block (or with wide running numbers for the
//                                queries of dense blocks): finishes 93-98 % of the queue on dense wavefronts. What is
//                                left — isolated queries, tied FP32 keys, every entry of a map-sized set — is listed.
//   associate_knn_left_kernel  : the listed ones, compacted (a few hundred wavefronts for a 1 024-pair batch): the FP64
//                                keyed collector over all rounds. Long walks, few wavefronts: latency the fit kernel hides.
//   associate_fit_queued_kernel: exact collector for what even the FP64 keys leave undecided, then the fit.
// (Until round 3 the second kernel's search ran on every wavefront of the first: 50-70 % of them held one or two
// isolated queries and walked every shell with 27 % of their lanes — 0.3-0.45 ms next to the fit kernel, which it slowed
// from 0.63 to 0.96 ms.)
// Workgroups of pairs with an empty queue leave after one scalar load.
#ifndef LOAMX_REST_THREADS
#define LOAMX_REST_THREADS 64
#endif
constexpr int kRestThreads = LOAMX_REST_THREADS;  // small workgroups: the queues are short and uneven
// The listed leftovers of a queue: one WAVEFRONT per entry (associate_knn_coop_kernel) or one lane per entry
// (associate_knn_left_kernel)? The cooperative search walks shells up to the one the radius closes; without a radius it falls
// back to lane 0 alone, 63 lanes idle — such searches take the one-lane kernel's dense packing instead.
__host__ __device__ inline bool queue_leftovers_coop(const RegConfig& C, bool plane) {
  return !(C.flags & kRegFlagNoCoopLeft) && (plane ? C.r_plane : C.r_edge) > 0.0;
}
#ifndef LOAMX_COOP_WAVES
#define LOAMX_COOP_WAVES 4
#endif
#ifndef LOAMX_REST_WAVES
// (round 2 measured 4 -> 2.12 ms, 5 -> 2.11, 6 -> 2.14 of association scope on the one-stage form; since the two-stage chain every
// instance settles at 3 wavefronts per SIMD (k <= 8) or 2 (k = 16) whatever is asked: the attribute now says what the compiler does)
#define LOAMX_REST_WAVES(KM) ((KM) <= 8 ? 3 : 2)
#endif
template <bool PLANE, int KM, bool ONE_STAGE>
__global__ __launch_bounds__(kRestThreads, LOAMX_REST_WAVES(KM)) void associate_knn_rest_kernel(RegBatch B, RegConfig C,
                                                                                              uint32_t blocks_per_pair) {
  constexpr bool one_stage = ONE_STAGE;  // (two instantiations: each at its own register budget)
  size_t pair;
  uint32_t chunk0;
  if (!xcd_pair_map(blockIdx.x, blocks_per_pair, B.n_pairs, pair, chunk0)) return;
  const PairState& S = B.state[pair];
  if (!S.active) return;                                              // uniform per workgroup
  const uint32_t queued = B.assoc.n_assoc[8 * pair + (PLANE ? 3 : 2)];
  if (queued == 0u) return;                                           // uniform per workgroup
  __shared__ uint32_t s_rows[(ONE_STAGE ? kLeanRowWords : kLean2RowWords) * kRestThreads];  // (the largest per-thread row list used below)
  const size_t stride = PLANE ? B.planar_stride : B.edge_stride;
  const size_t field = B.n_pairs * stride;
  uint32_t* __restrict__ rnn = PLANE ? B.assoc.rnn_plane : B.assoc.rnn_edge;  // results by queue position
  const uint32_t* __restrict__ rest = (PLANE ? B.assoc.rest_plane : B.assoc.rest_edge) + pair * stride;
  uint32_t* __restrict__ left = (PLANE ? B.assoc.exact_plane : B.assoc.exact_edge) + pair * stride;
  const GridSet& gs = PLANE ? B.grid_plane : B.grid_edge;
  const GridSet& src_gs = PLANE ? B.src_grid_plane : B.src_grid_edge;
  const GridDesc g = gs.desc[pair];
  const uint32_t* __restrict__ cs = gs.cell_start + pair * (size_t)(kGridCellsCap + 1);
  const GridPoint* __restrict__ sp = gs.sorted + pair * gs.stride;
  // Which queue entries this wavefront takes. A batch wants dense wavefronts (64 consecutive entries: the kernel runs next
  // to the plane fit and should cost as few issue slots as possible); a handful of pairs leaves the chip empty, and then
  // the latency of the slowest wavefront is the kernel's time: the entries are dealt out across ALL workgroups, one or
  // two per wavefront (one 64 x 1024 pair: 480 entries on 276 wavefronts instead of 8 — 71 -> 25 us per launch).
  const bool spread = B.n_pairs < 8 && (size_t)queued * 4 <= (size_t)blocks_per_pair * kRestThreads;
  const uint32_t t0 = spread ? chunk0 + blocks_per_pair * threadIdx.x : chunk0 * kRestThreads + threadIdx.x;
  const uint32_t dt = spread ? 0xFFFFFFFFu - t0 : blocks_per_pair * kRestThreads;  // (spread: one entry per lane)
  const bool lean_set = g.n_points <= kLeanMaxPoints;  // (uniform: a property of the set; positions travel as 16-bit halves)
  const int kq = PLANE ? C.k_plane : C.k_edge;
  const double max_dist = PLANE ? C.r_plane : C.r_edge, pass_max = PLANE ? C.pass_plane : C.pass_edge;
  for (uint32_t t = t0; t < queued; t += dt) {
    const uint32_t entry = rest[t], i = entry & kQueueIndex;
    const size_t slot = pair * stride + t;  // (queue position, not query index)
    uint32_t pos[KM];
#pragma unroll
    for (int j = 0; j < KM; j++) pos[j] = 0;
    int kept = -1;
    const bool coop = !one_stage && queue_leftovers_coop(C, PLANE);  // (wide entries: a dense block is the cooperative kernel's business)
    if (one_stage || (!(entry & kQueueTied) && ((lean_set && !(entry & kQueueWide)) || ((entry & kQueueWide) && !coop)))) {
      const GridPoint sq = src_gs.sorted[pair * src_gs.stride + i];
      const Vec3 p = pose_act(S.est, v3(sq.x, sq.y, sq.z));
      if (entry & kQueueWide)
        kept = knn_search_f32_round1<KM, true>(g, cs, sp, gs.rel + pair * 3 * gs.stride, (uint32_t)gs.stride, p, kq, max_dist, pass_max, pos,
                                                s_rows + threadIdx.x, kRestThreads);
      else if (!one_stage)
        kept = knn_lean_round2<KM>(g, cs, sp, gs.rel + pair * 3 * gs.stride, (uint32_t)gs.stride, p, kq, max_dist, pass_max, pos,
                                   s_rows + threadIdx.x, kRestThreads);
      // one stage (small batches, see launch_associate): the FP64 search over all rounds here and now
      if (one_stage && kept < 0)
        kept = knn_search_keyed<KM>(g, cs, sp, p, kq, max_dist, pass_max, pos, s_rows + threadIdx.x, kRestThreads);
    }
    rnn[slot] = (uint32_t)kept;
    if (kept < 0) left[atomicAdd(&B.assoc.n_assoc[8 * pair + (PLANE ? 5 : 4)], 1u)] = t;  // (one stage: counted only)
#pragma unroll
    for (int j = 0; j < KM; j++) rnn[(1 + j) * field + slot] = pos[j];
  }
}

// The listed queue entries (see above): the FP64 keyed collector — each candidate ONE double, the bits of d2 with the
// low mantissa bits replaced by the position — over all rounds. What its keys cannot decide keeps its negative count.
template <bool PLANE, int KM>
__global__ __launch_bounds__(kRestThreads, 5) void associate_knn_left_kernel(RegBatch B, RegConfig C,
                                                                                              uint32_t blocks_per_pair) {
  size_t pair;
  uint32_t chunk0;
  if (!xcd_pair_map(blockIdx.x, blocks_per_pair, B.n_pairs, pair, chunk0)) return;
  const PairState& S = B.state[pair];
  if (!S.active) return;                                              // uniform per workgroup
  const uint32_t listed = B.assoc.n_assoc[8 * pair + (PLANE ? 5 : 4)];
  if (listed == 0u) return;                                           // uniform per workgroup
  __shared__ uint32_t s_rows[kLeanRowWords * kRestThreads];
  const size_t stride = PLANE ? B.planar_stride : B.edge_stride;
  const size_t field = B.n_pairs * stride;
  uint32_t* __restrict__ rnn = PLANE ? B.assoc.rnn_plane : B.assoc.rnn_edge;
  const uint32_t* __restrict__ rest = (PLANE ? B.assoc.rest_plane : B.assoc.rest_edge) + pair * stride;
  const uint32_t* __restrict__ left = (PLANE ? B.assoc.exact_plane : B.assoc.exact_edge) + pair * stride;
  const GridSet& gs = PLANE ? B.grid_plane : B.grid_edge;
  const GridSet& src_gs = PLANE ? B.src_grid_plane : B.src_grid_edge;
  const GridDesc g = gs.desc[pair];
  const uint32_t* __restrict__ cs = gs.cell_start + pair * (size_t)(kGridCellsCap + 1);
  const GridPoint* __restrict__ sp = gs.sorted + pair * gs.stride;
  const bool spread = B.n_pairs < 8 && (size_t)listed * 4 <= (size_t)blocks_per_pair * kRestThreads;  // (as above)
  const uint32_t t0 = spread ? chunk0 + blocks_per_pair * threadIdx.x : chunk0 * kRestThreads + threadIdx.x;
  const uint32_t dt = spread ? 0xFFFFFFFFu - t0 : blocks_per_pair * kRestThreads;
  for (uint32_t t = t0; t < listed; t += dt) {
    const uint32_t qpos = left[t], i = rest[qpos] & kQueueIndex;
    const size_t slot = pair * stride + qpos;
    const GridPoint sq = src_gs.sorted[pair * src_gs.stride + i];
    const Vec3 p = pose_act(S.est, v3(sq.x, sq.y, sq.z));
    uint32_t pos[KM];
    const int kept = knn_search_keyed<KM>(g, cs, sp, p, PLANE ? C.k_plane : C.k_edge, PLANE ? C.r_plane : C.r_edge,
                                          PLANE ? C.pass_plane : C.pass_edge, pos, s_rows + threadIdx.x, kRestThreads);
    rnn[slot] = (uint32_t)kept;
#pragma unroll
    for (int j = 0; j < KM; j++) rnn[(1 + j) * field + slot] = pos[j];
  }
}

/* ------------------------------------------------------------------------------------------------
 * Wave-cooperative search (round 4): the 64 lanes of a wavefront share ONE query. The listed leftovers of the queue are
 * the searches one lane cannot do well: isolated queries (the whole radius cube: 81 rows at a quarter-radius cell, nearly
 * all empty — ~100 dependent round trips for one lane) and, against a map-sized set, queries whose 3x3x3 block holds
 * thousands of points (one lane walking a thousand batches: associate_knn_rest_kernel took 4 x 1 ms of BASELINE config 5).
 * Here the rows of the block are dealt out over the lanes (both table entries of up to 64 rows in flight at once), and
 * every row's candidates are taken 64 at a time, one per lane, coalesced. Each lane keeps its own FP64 keyed collector
 * (KnnKeys: one double per candidate, distance bits + position); the k + 1 smallest keys of the wavefront are then drawn
 * from the lanes' lists by k + 1 wave-wide minima. That is the collector's own contract — the k + 1 smallest keys of all
 * the candidates examined — so knn_done / knn_keys_finish decide exactly as they do for one lane, and the result is the
 * exact search's (what the keys cannot decide keeps its negative count for the exact collector, as before).
 * The block grows shell by shell as in the one-lane search (knn_rounds): after shell w the wavefront's exact k-th key is
 * tested against the faces of the visited block (knn_done); the last shell is the cube whose faces lie beyond the radius.
 * Pieces farther than the bound (the pooled k-th key; within a shell also any lane's own k-th key) or than the radius are
 * skipped; inside a row only the cells the bound reaches.
 * Returns kept >= 0, -1 undecided, -2 not applicable (more than kCoopMaxShell shells could be needed): the caller falls
 * back to the one-lane search.
 * ---------------------------------------------------------------------------------------------- */
constexpr int32_t kCoopMaxShell = 64;  // shells a cooperative search may walk (a radius of 62 cells; without a radius: grids up to that size)

__device__ __forceinline__ double wave_min_all(double v) {  // (every lane gets the minimum)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double o = __shfl_xor(v, off);
    v = knn_key_min(v, o);
  }
  return v;
}

template <int KM>
__device__ __forceinline__ int knn_coop_search(const GridDesc& g, const uint32_t* __restrict__ cell_start, const GridPoint* __restrict__ sp,
                                               Vec3 q, int k, double max_dist, double pass_max, uint32_t pos[KM]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < KM; j++) pos[j] = 0;
  if (g.n_points == 0 || k <= 0) return 0;
  if (k > KM) k = KM;
  const int32_t cx = grid_cell_coord(q.x, g.ox, g.inv_h), cy = grid_cell_coord(q.y, g.oy, g.inv_h), cz = grid_cell_coord(q.z, g.oz, g.inv_h);
  const int32_t out = grid_outside_distance(g, cx, cy, cz);
  if (max_dist > 0.0 && out >= 1 && (double)(out - 1) * g.h >= max_dist) return 0;
  // Shells of growing half-width w, as the one-lane search visits them (knn_rounds): shell 1 = the 3x3x3 block, shell w > 1 =
  // the cells at Chebyshev distance w. The last shell a radius can need: the cube whose faces lie strictly beyond it
  // (+ 1: whatever the query's place in its cell; + out: a query outside the grid sits `out` cells off the nearest grid cell).
  int32_t W = 0x7FFFFFFF;
  if (max_dist > 0.0) {
    const double wf = ceil(max_dist * g.inv_h) + 1.0 + (double)out;
    if (wf < 1.0e6) W = (int32_t)wf;
  }
  if (W > kCoopMaxShell) {  // no radius (or a huge one in cells): only while the grid itself is that small
    const int32_t span = (g.nx > g.ny ? (g.nx > g.nz ? g.nx : g.nz) : (g.ny > g.nz ? g.ny : g.nz)) + out;
    if (span > kCoopMaxShell) return -2;
    W = span;
  }
  const double r2 = knn_radius_bound(max_dist);
  double bound = r2;  // wave-uniform: nothing farther than this can be among the k kept
  KnnKeys<KM> c;
  knn_init(c, k, g.n_points);
  bool done = false;
  double kth = knn_key_empty();  // the k-th smallest key of the wavefront after the last shell that offered a candidate
  bool fresh = false;            // (wave-uniform) candidates were offered since kth was drawn
  for (int32_t w = 1; w <= W && !done; w++) {
    // pieces of shell w: the rows at Chebyshev distance w over x in [cx - w, cx + w] (shell 1: all nine rows of the block,
    // centre first), then for w > 1 the two end cells cx -+ w of every row inside
    const int ring_rows = w == 1 ? 9 : 8 * w;
    const int inner_rows = w == 1 ? 0 : (2 * w - 1) * (2 * w - 1);
    const int pieces = ring_rows + 2 * inner_rows;
    for (int u0 = 0; u0 < pieces; u0 += 64) {
      // ---- lane l owns one piece: its distance and, if the bound reaches it, its range
      uint32_t rb = 0, re = 0;
      double s2 = kDblMax;
      {
        const int u = u0 + lane;
        int32_t dy = 0, dz = 0, xlo = cx - w, xhi = cx + w;
        bool cap = false;
        if (w == 1) {
          constexpr int kOrder[9] = {4, 1, 3, 5, 7, 0, 2, 6, 8};
          const int j = kOrder[u < 9 ? u : 0];
          dy = j % 3 - 1, dz = j / 3 - 1;
        } else if (u < ring_rows) {
          const int side = 2 * w + 1, inner = 2 * w - 1;
          if (u < side) dy = u - w, dz = -w;
          else if (u < 2 * side) dy = u - side - w, dz = w;
          else if (u < 2 * side + inner) dy = -w, dz = u - 2 * side - (w - 1);
          else dy = w, dz = u - 2 * side - inner - (w - 1);
        } else {
          const int v = u - ring_rows, ri = v >> 1, inner = 2 * w - 1;
          dy = ri % inner - (w - 1), dz = ri / inner - (w - 1);
          xlo = xhi = (v & 1) ? cx + w : cx - w;
          cap = true;
        }
        const int32_t iy = cy + dy, iz = cz + dz;
        if (u < pieces && iy >= 0 && iy <= g.ny - 1 && iz >= 0 && iz <= g.nz - 1) {
          const double sy = dy == 0 ? 0.0 : slab_dist(q.y, g.oy, g.h, iy), sz = dz == 0 ? 0.0 : slab_dist(q.z, g.oz, g.h, iz);
          double pmin2 = sy * sy + sz * sz;
          if (cap) {
            const double sx = slab_dist(q.x, g.ox, g.h, xlo);
            pmin2 += sx * sx;
          }
          if (pmin2 <= bound) {
            if (!cap) {
              if (xlo < 0) xlo = 0;
              if (xhi > g.nx - 1) xhi = g.nx - 1;
              if (bound < kDblMax) {  // cells whose x slab is farther than sqrt(bound - rowmin2) cannot contribute (as knn_general_round)
                const double reach = sqrt(bound - pmin2) + 1e-9 * g.h;
                const int32_t xl = grid_cell_coord(q.x - reach, g.ox, g.inv_h), xh = grid_cell_coord(q.x + reach, g.ox, g.inv_h);
                if (xl > xlo) xlo = xl;
                if (xh < xhi) xhi = xh;
              }
            }
            if (xlo <= xhi && xlo >= 0 && xhi <= g.nx - 1) {
              const uint32_t row = (uint32_t)((iz * g.ny + iy) * g.nx);
              rb = cell_start_at(cell_start, row + (uint32_t)xlo);
              re = cell_start_at(cell_start, row + (uint32_t)xhi + 1u);
              s2 = pmin2;
            }
          }
        }
      }
      // ---- the non-empty pieces of this turn, one after the other, 64 candidates at a time
      unsigned long long rows = __ballot(rb < re);
      while (rows) {
        const int b = __ffsll(rows) - 1;
        rows &= rows - 1;
        const uint32_t begin = (uint32_t)__builtin_amdgcn_readlane((int)rb, b), end = (uint32_t)__builtin_amdgcn_readlane((int)re, b);
        const double rs2 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s2), b), __builtin_amdgcn_readlane(__double2loint(s2), b));
        if (rs2 > bound) continue;  // (the bound has moved since the piece was listed)
        fresh = true;
        for (uint32_t p0 = begin; p0 < end; p0 += 64) {
          const uint32_t p = p0 + (uint32_t)lane;
          const bool real = p < end;
          const GridPoint t = sp[real ? p : begin];
          const double dx = q.x - t.x, dy = q.y - t.y, dz = q.z - t.z;
          const double d2 = dx * dx + dy * dy + dz * dz;  // nanoflann L2_Simple, as knn_scan_batch
          knn_key_insert(c, knn_key_pack(d2, p, c.mask, real));
        }
        if (end - begin >= 256u) {  // a populous piece may have moved the bound (a lane with k keys bounds the true k-th distance)
          const double wb = wave_min_all(knn_bound(c, k));
          bound = wb < bound ? wb : bound;
        }
      }
    }
    // ---- is the search over after shell w? The k-th smallest key of the wavefront, drawn from the lanes' lists as at the
    // end (on a copy), against the faces of the visited block: knn_done's test
    {
      // (an empty shell — an isolated query walks ceil(R / h) + 1 of them — leaves the k-th key where it was: the k
      // wave-wide minima are only drawn again after a shell that offered candidates)
      if (fresh) {
      fresh = false;
      KnnKeys<KM> cc = c;
      kth = knn_key_empty();
      for (int t = 0; t < k; t++) {
        double head = knn_key_empty();
#pragma unroll
        for (int j = KM; j >= 0; j--)
          if (j >= KM - k) head = cc.key[j] < head ? cc.key[j] : head;
        const double best = wave_min_all(head);
        kth = best;
        if (head == best && best < knn_key_empty()) {
#pragma unroll
          for (int j = 0; j < KM; j++)
            if (j >= KM - k) cc.key[j] = cc.key[j + 1];
          cc.key[KM] = knn_key_empty();
        }
      }
      }
      KnnKeys<KM> probe;  // (only slot KM - 1 and the mask are read by knn_bound)
      probe.mask = c.mask;
#pragma unroll
      for (int j = 0; j <= KM; j++) probe.key[j] = kth;
      if (kth < knn_key_empty()) {
        const double ub = knn_bound(probe, k);
        bound = ub < bound ? ub : bound;
      }
      done = knn_done(g, q, k, max_dist, cx, cy, cz, probe, w);  // (wave-uniform)
    }
  }
  if (!done) return -1;
  // ---- the k + 1 smallest keys of the wavefront: k + 1 times the smallest head of the lanes' lists
  KnnKeys<KM> m;
  knn_init(m, k, g.n_points);
#pragma unroll
  for (int t = 0; t <= KM; t++) {
    if (t >= KM - k) {  // (slots below hold the sentinel)
      double head = knn_key_empty();
#pragma unroll
      for (int j = KM; j >= 0; j--)
        if (j >= KM - k) head = c.key[j] < head ? c.key[j] : head;  // the list is ascending: its first real slot — written branch-free
      const double best = wave_min_all(head);
      m.key[t] = best;
      if (head == best && best < knn_key_empty()) {  // the owner drops it (keys of different candidates differ in their position bits)
#pragma unroll
        for (int j = 0; j < KM; j++)
          if (j >= KM - k) c.key[j] = c.key[j + 1];
        c.key[KM] = knn_key_empty();
      }
    }
  }
  return knn_keys_finish(m, k, pass_max, pos);
}

// The listed leftovers, one WAVEFRONT per entry (see knn_coop_search); an entry the cooperative search does not apply to
// (no radius limit) is searched by lane 0 alone, as associate_knn_left_kernel does it.
// (one wavefront per workgroup: the entry loop steps by workgroups and s_rows is one lane's list — ADVICE r4)
static_assert(kRestThreads == 64, "associate_knn_coop_kernel is written for one wavefront per workgroup");
template <bool PLANE, int KM>
__global__ __launch_bounds__(kRestThreads, LOAMX_COOP_WAVES) void associate_knn_coop_kernel(RegBatch B, RegConfig C, uint32_t blocks_per_pair) {
  size_t pair;
  uint32_t chunk0;
  if (!xcd_pair_map(blockIdx.x, blocks_per_pair, B.n_pairs, pair, chunk0)) return;
  const PairState& S = B.state[pair];
  if (!S.active) return;                                              // uniform per workgroup
  const uint32_t listed = B.assoc.n_assoc[8 * pair + (PLANE ? 5 : 4)];
  if (listed == 0u) return;                                           // uniform per workgroup
  __shared__ uint32_t s_rows[kLeanRowWords];
  const size_t stride = PLANE ? B.planar_stride : B.edge_stride;
  const size_t field = B.n_pairs * stride;
  uint32_t* __restrict__ rnn = PLANE ? B.assoc.rnn_plane : B.assoc.rnn_edge;
  const uint32_t* __restrict__ rest = (PLANE ? B.assoc.rest_plane : B.assoc.rest_edge) + pair * stride;
  const uint32_t* __restrict__ left = (PLANE ? B.assoc.exact_plane : B.assoc.exact_edge) + pair * stride;
  const GridSet& gs = PLANE ? B.grid_plane : B.grid_edge;
  const GridSet& src_gs = PLANE ? B.src_grid_plane : B.src_grid_edge;
  const GridDesc g = gs.desc[pair];
  const uint32_t* __restrict__ cs = gs.cell_start + pair * (size_t)(kGridCellsCap + 1);
  const GridPoint* __restrict__ sp = gs.sorted + pair * gs.stride;
  const int kq = PLANE ? C.k_plane : C.k_edge;
  const double max_dist = PLANE ? C.r_plane : C.r_edge, pass_max = PLANE ? C.pass_plane : C.pass_edge;
  for (uint32_t t = chunk0; t < listed; t += blocks_per_pair) {  // (wave-uniform: one entry per wavefront and turn)
    const uint32_t qpos = left[t], i = rest[qpos] & kQueueIndex;
    const size_t slot = pair * stride + qpos;
    const GridPoint sq = src_gs.sorted[pair * src_gs.stride + i];
    const Vec3 p = pose_act(S.est, v3(sq.x, sq.y, sq.z));
    uint32_t pos[KM];
    int kept = knn_coop_search<KM>(g, cs, sp, p, kq, max_dist, pass_max, pos);
    if (kept == -2) {  // uniform
      if ((threadIdx.x & 63) == 0) kept = knn_search_keyed<KM>(g, cs, sp, p, kq, max_dist, pass_max, pos, s_rows, 1);
      kept = __builtin_amdgcn_readfirstlane(kept);
#pragma unroll
      for (int j = 0; j < KM; j++) pos[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos[j]);
    }
    if ((threadIdx.x & 63) == 0) {
      rnn[slot] = (uint32_t)kept;
#pragma unroll
      for (int j = 0; j < KM; j++) rnn[(1 + j) * field + slot] = pos[j];
    }
  }
}

// Third pass: the queries whose keys stayed undecided (exact distance ties, a truncated distance
// straddling the radius) are searched with the exact (d2, index) collector — by the thread that fits them, at the
// head of associate_fit_queued_kernel. (It was a kernel of its own between the two until round 3: its 256-thread
// workgroups found no room next to the plane fit kernel and so it ENDED with that kernel, whatever little it had to
// do — 8 us of work, 260-490 us in the trace — and the queued fit started only then.)
template <bool PLANE, int KM>
__device__ __forceinline__ void knn_exact_one(const RegBatch& B, const RegConfig& C, const PairState& S, size_t pair, uint32_t i,
                                              size_t slot, uint32_t* rnn, uint32_t* s_rows, int rows_stride) {
  const size_t stride = PLANE ? B.planar_stride : B.edge_stride;
  const size_t field = B.n_pairs * stride;
  const GridSet& gs = PLANE ? B.grid_plane : B.grid_edge;
  const GridSet& src_gs = PLANE ? B.src_grid_plane : B.src_grid_edge;
  const GridDesc g = gs.desc[pair];
  const uint32_t* __restrict__ cs = gs.cell_start + pair * (size_t)(kGridCellsCap + 1);
  const GridPoint* __restrict__ sp = gs.sorted + pair * gs.stride;
  int k = PLANE ? C.k_plane : C.k_edge;
  k = k < KM ? k : KM;
  const GridPoint sq = src_gs.sorted[pair * src_gs.stride + i];
  const Vec3 p = pose_act(S.est, v3(sq.x, sq.y, sq.z));
  KnnResult<KM> r;
  const int kept = knn_search(g, cs, sp, p, k, PLANE ? C.r_plane : C.r_edge, r, s_rows, rows_stride);
  uint32_t pos[KM];
  knn_shift_positions<KM>(r.pos, k, pos);  // neighbour j is slot (KM - k) + j
  rnn[slot] = (uint32_t)kept;
#pragma unroll
  for (int j = 0; j < KM; j++) rnn[(size_t)(1 + j) * field + slot] = pos[j];
}

// fitLine / fitPlane on the neighbours of query i of `pair` and its association record; the neighbour
// count and positions are read at index nidx of the (1 + KM)-field array nnsrc. Returns "valid".
template <bool PLANE, int KM, bool QUEUED = false>
__device__ __forceinline__ bool fit_one(const RegBatch& B, const RegConfig& C, const PairState& S, size_t pair, uint32_t i,
                                        const uint32_t* __restrict__ nnsrc, size_t nidx) {
  const size_t stride = PLANE ? B.planar_stride : B.edge_stride;
  const GridSet& gs = PLANE ? B.grid_plane : B.grid_edge;
  const GridSet& src_gs = PLANE ? B.src_grid_plane : B.src_grid_edge;
  const GridPoint sq = src_gs.sorted[pair * src_gs.stride + i];
  const Vec3 p = pose_act(S.est, v3(sq.x, sq.y, sq.z));
  const GridPoint* __restrict__ sp = gs.sorted + pair * gs.stride;
  const size_t field = B.n_pairs * stride, slot = pair * stride + i;
  const int kept = (int)nnsrc[nidx];
  const int kq = PLANE ? C.k_plane : C.k_edge;
  const int shift = KM - (kq < KM ? kq : KM);  // neighbour j is slot shift + j (knn_search_positions)
  double prim[6] = {0, 0, 0, 0, 0, 0};
  uint32_t nearest = 0xFFFFFFFFu;
  bool valid = false;
  // The KM positions, then the KM points, fetched unconditionally (slots past `kept` hold position 0 or a stale but
  // valid one; clamped to the set's capacity anyway): two round trips for the lane instead of one or two per neighbour
  // behind a branch each.
  uint32_t at[KM];
#pragma unroll
  for (int j = 0; j < KM; j++) at[j] = shift + j < KM ? nnsrc[(size_t)(1 + shift + j) * field + nidx] : 0u;
  if (!QUEUED && kept == -1) return false;  // 0xFFFFFFFF: round 1 queued this query, associate_fit_queued_kernel writes its record
  GridPoint tp[KM];
#pragma unroll
  for (int j = 0; j < KM; j++) tp[j] = sp[at[j] < (uint32_t)gs.stride ? at[j] : 0u];
  if (kept >= (PLANE ? C.min_plane_pts : C.min_line_pts)) {  // registration.cpp:39 / :80
    Vec3 nb[KM];
#pragma unroll
    for (int j = 0; j < KM; j++) nb[j] = j < kept ? v3(tp[j].x, tp[j].y, tp[j].z) : v3(0, 0, 0);
    if (kept > 0) nearest = tp[0].orig;
    if (PLANE) {
      Vec3 nrm;
      double d;
      const double avg = fit_plane<KM>(nb, kept, nrm, d);
      valid = !(avg > C.max_avg_plane_dist);  // registration.cpp:90
      prim[0] = nrm.x, prim[1] = nrm.y, prim[2] = nrm.z, prim[3] = d;
    } else {
      Vec3 a, b;
      fit_line<KM>(nb, kept, a, b);
      valid = !(kDblMax < C.min_line_cond);  // registration.cpp:49 (condition number is always DBL_MAX)
      prim[0] = a.x, prim[1] = a.y, prim[2] = a.z, prim[3] = b.x, prim[4] = b.y, prim[5] = b.z;
    }
  }
  double* __restrict__ rec = PLANE ? B.assoc.plane : B.assoc.edge;
  rec[slot] = valid ? p.x : __longlong_as_double(0x7FF8000000000000ll);
  rec[field + slot] = p.y;
  rec[2 * field + slot] = p.z;
#pragma unroll
  for (int f = 0; f < (PLANE ? 4 : 6); f++) rec[(3 + f) * field + slot] = prim[f];
  // detail capture is indexed by the caller's source index
  if (B.want_nearest) (PLANE ? B.assoc.nearest_plane : B.assoc.nearest_edge)[pair * stride + sq.orig] = valid ? nearest : 0xFFFFFFFFu;
  return valid;
}

// The queries round 1 of the k-NN finished (everything but the queued ones, whose count stays 0xFFFFFFFF in nn).
#ifndef LOAMX_FIT_WAVES
#define LOAMX_FIT_WAVES 4
#endif
template <bool PLANE, int KM>
__device__ __forceinline__ void fit_body(const RegBatch& B, const RegConfig& C, uint32_t blocks_per_pair, uint32_t block) {
  size_t pair;
  uint32_t chunk;
  if (!xcd_pair_map(block, blocks_per_pair, B.n_pairs, pair, chunk)) return;
  const uint32_t i = chunk * kAssocThreads + threadIdx.x;
  const PairState& S = B.state[pair];
  if (!S.active) return;  // uniform per workgroup
  const size_t stride = PLANE ? B.planar_stride : B.edge_stride;
  const uint32_t n_src = PLANE ? B.n_src_planar[pair * B.in_pitch] : B.n_src_edge[pair * B.in_pitch];
  const uint32_t* __restrict__ nn = PLANE ? B.assoc.nn_plane : B.assoc.nn_edge;
  bool valid = false;
  if (i < n_src && i < stride) valid = fit_one<PLANE, KM>(B, C, S, pair, i, nn, pair * stride + i);  // (queued queries: skipped inside)
  // (one atomic per wavefront, no barrier: a wavefront that is done leaves)
  const unsigned long long m = __ballot(valid);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&B.assoc.n_assoc[8 * pair + (PLANE ? 1 : 0)], (uint32_t)__popcll(m));
}
template <bool PLANE, int KM>
__global__ __launch_bounds__(kAssocThreads, LOAMX_FIT_WAVES) void associate_fit_kernel(RegBatch B, RegConfig C,
                                                                                        uint32_t blocks_per_pair) {
  fit_body<PLANE, KM>(B, C, blocks_per_pair, blockIdx.x);
}
template <int KME, int KMP>  // (edge and plane fits in one launch, as associate_knn_mixed_kernel)
__global__ __launch_bounds__(kAssocThreads, LOAMX_FIT_WAVES) void associate_fit_mixed_kernel(RegBatch B, RegConfig C, uint32_t blocks_edge,
                                                                                              uint32_t blocks_plane, uint32_t edge_blocks) {
  if (blockIdx.x < edge_blocks) fit_body<false, KME>(B, C, blocks_edge, blockIdx.x);
  else fit_body<true, KMP>(B, C, blocks_plane, blockIdx.x - edge_blocks);
}

// The queued queries, after associate_knn_rest_kernel: what its keys left undecided is searched exactly first, then the
// same fit, results read by queue position.
template <bool PLANE, int KM>
__global__ __launch_bounds__(kRestThreads, 3) void associate_fit_queued_kernel(RegBatch B, RegConfig C, uint32_t blocks_per_pair) {  // (3: <= 168 registers — unbounded it took 230 and waited for two fit wavefronts to leave a SIMD at once; at 128 it spills 376 bytes)
  size_t pair;
  uint32_t chunk0;
  if (!xcd_pair_map(blockIdx.x, blocks_per_pair, B.n_pairs, pair, chunk0)) return;
  const PairState& S = B.state[pair];
  if (!S.active) return;                                              // uniform per workgroup
  const uint32_t queued = B.assoc.n_assoc[8 * pair + (PLANE ? 3 : 2)];
  if (queued == 0u) return;                                           // uniform per workgroup
  __shared__ uint32_t s_rows[18 * kRestThreads];
  const size_t stride = PLANE ? B.planar_stride : B.edge_stride;
  uint32_t* rnn = PLANE ? B.assoc.rnn_plane : B.assoc.rnn_edge;  // (written and read back here: no __restrict__)
  const uint32_t* __restrict__ rest = (PLANE ? B.assoc.rest_plane : B.assoc.rest_edge) + pair * stride;
  uint32_t count = 0;
  for (uint32_t t = chunk0 * kRestThreads + threadIdx.x; t < queued; t += blocks_per_pair * kRestThreads) {
    const uint32_t i = rest[t] & kQueueIndex;
    if ((int)rnn[pair * stride + t] < 0)  // (rare: its own stores, read back by the same thread in fit_one)
      knn_exact_one<PLANE, KM>(B, C, S, pair, i, pair * stride + t, rnn, s_rows + threadIdx.x, kRestThreads);
    count += fit_one<PLANE, KM, true>(B, C, S, pair, i, rnn, pair * stride + t) ? 1u : 0u;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) count += __shfl_xor(count, off);
  if ((threadIdx.x & 63) == 0 && count) atomicAdd(&B.assoc.n_assoc[8 * pair + (PLANE ? 1 : 0)], count);
}

// (the ICF iteration number of an active pair is S.iterations, the iterations it has completed: no kernel argument
// changes from one iteration to the next, so that the iteration can be replayed as a hipGraph)
__global__ __launch_bounds__(64) void lm_begin_kernel(RegBatch B, RegConfig C) {
  const size_t pair = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= B.n_pairs) return;
  if (pair == 0) *B.n_active = 0u;  // outer_update_pair counts the pairs that go on (read back by the host after the iteration)
  PairState& S = B.state[pair];
  if (!S.active) return;
  const uint32_t iteration = S.iterations;
  const uint32_t ne = B.assoc.n_assoc[8 * pair], np = B.assoc.n_assoc[8 * pair + 1];
  for (int c = 0; c < 6; c++) B.assoc.n_assoc[8 * pair + c] = 0;
  if ((uint64_t)ne + np < C.min_associations) {  // registration-inl.h:45-48
    S.termination = LOAMX_INSUFFICIENT_ASSOCIATIONS;
    S.active = 0;
    S.lm.active = 0;
    return;
  }
  lm_init(S.lm);
  S.first_sweep = 1;
  S.mom_ref_on = 0;  // (the moments of this ICF iteration are taken at the identity update unless lm_step_pair says otherwise)
  if (B.iter_info) {
    loamx_iter_info& I = B.iter_info[pair * C.max_iterations + iteration];
    for (int i = 0; i < 7; i++) I.target_T_source_init[i] = S.est[i];
    I.n_edge_associations = ne;
    I.n_plane_associations = np;
  }
}

/* ------------------------------------------------------------------------------------------------ */
#ifndef LOAMX_SWEEP_WAVES
#define LOAMX_SWEEP_WAVES 3  // measured: 2 waves/SIMD (178 VGPRs) 0.31 ms, 3 (166) 0.24 ms, 4 (spills) 0.44 ms
#endif
__global__ __launch_bounds__(kSweepThreads, LOAMX_SWEEP_WAVES) void sweep_kernel(RegBatch B) {
  __shared__ double s_part[kSweepThreads / 64][kAccSize];
  const size_t pair = blockIdx.x / B.blocks_per_pair;
  const uint32_t blk = blockIdx.x % B.blocks_per_pair;
  const PairState& S = B.state[pair];
  if (!S.active || !S.lm.active) return;  // uniform per workgroup
  double x[7];
#pragma unroll
  for (int i = 0; i < 7; i++) x[i] = S.lm.xeval[i];
  const uint32_t n_se_raw = B.n_src_edge[pair * B.in_pitch], n_sp_raw = B.n_src_planar[pair * B.in_pitch];
  const uint32_t n_se = n_se_raw < B.edge_stride ? n_se_raw : (uint32_t)B.edge_stride;
  const uint32_t n_sp = n_sp_raw < B.planar_stride ? n_sp_raw : (uint32_t)B.planar_stride;
  const size_t efield = B.n_pairs * B.edge_stride, pfield = B.n_pairs * B.planar_stride;
  const double* __restrict__ E = B.assoc.edge + pair * B.edge_stride;
  const double* __restrict__ Pl = B.assoc.plane + pair * B.planar_stride;
  double acc[kAccSize];
#pragma unroll
  for (int j = 0; j < kAccSize; j++) acc[j] = 0.0;
  // the slot space of a pair is [edges 0..n_se) ++ [planes 0..n_sp); chunk `blk` of it. A pair whose moment
  // matrix can stand in for its plane records at this point (reg_math.h: "Plane residuals through moments") is
  // left to lm_pair_loop_kernel.
  if (S.stream_planes == 0u) return;  // uniform (moments + listed records)
  const uint32_t total = n_se + n_sp;
  const uint32_t base = blk * kSweepChunk;
  if (base >= total) return;  // uniform per workgroup
  if (blk == 0 && threadIdx.x == 0 && B.sweep_slots) {
    atomicAdd(&B.sweep_slots[0], (unsigned long long)n_se);
    atomicAdd(&B.sweep_slots[1], (unsigned long long)n_sp);
  }
  {
  // software-pipelined: the record of item it+1 is in flight while item it is evaluated
  struct Rec {
    double f[9];
    int kind;  // 0 none / invalid, 1 edge, 2 plane
  };
  auto load_rec = [&](uint32_t v) {
    Rec R;
    R.kind = 0;
#pragma unroll
    for (int f = 0; f < 9; f++) R.f[f] = 0.0;
    if (v < total) {
      if (v < n_se) {
        R.kind = 1;
#pragma unroll
        for (int f = 0; f < 9; f++) R.f[f] = E[f * efield + v];
      } else {
        const uint32_t q = v - n_se;
        R.kind = 2;
#pragma unroll
        for (int f = 0; f < 7; f++) R.f[f] = Pl[f * pfield + q];
      }
    }
    return R;
  };
  Rec cur = load_rec(base + threadIdx.x);
#pragma unroll 1
  for (int it = 0; it < kSweepItems; it++) {
    const Rec nxt = (it + 1 < kSweepItems) ? load_rec(base + (it + 1) * kSweepThreads + threadIdx.x) : Rec{{0, 0, 0, 0, 0, 0, 0, 0, 0}, 0};
    if (cur.kind != 0 && cur.f[0] == cur.f[0]) {  // NaN in field 0 marks an invalid slot
      double prim[6];
#pragma unroll
      for (int f = 0; f < 6; f++) prim[f] = cur.f[3 + f];
      residual_accumulate(cur.kind == 2, v3(cur.f[0], cur.f[1], cur.f[2]), prim, x, acc);
    }
    cur = nxt;
  }
  }  // streaming path
  // wavefront shuffle reduction, then LDS across the 4 wavefronts, fixed order => deterministic
#pragma unroll
  for (int j = 0; j < kAccSize; j++) {
    double v = acc[j];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off);
    acc[j] = v;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < kAccSize; j++) s_part[wave][j] = acc[j];
  }
  __syncthreads();
  if (threadIdx.x < kAccSize) {
    double v = s_part[0][threadIdx.x];
    for (int w = 1; w < kSweepThreads / 64; w++) v += s_part[w][threadIdx.x];
    B.partials[(pair * B.blocks_per_pair + blk) * kAccSize + threadIdx.x] = v;
  }
}

/* The evaluation of a pair that is on moments, by ONE WAVEFRONT: its edge records and the listed (far-from-plane)
 * plane records one by one, plus the plane terms of all other records from the moment matrix (the 13x7 products
 * M [Z | phi] spread over the lanes). Lanes 0..kAccSize-1 return their entry of the pair's partial. `mom` is in LDS. */
struct LightLds {
  double T[kMomDim][7], Z[kMomDim][7];
};
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
// Kept in LDS for the whole solve of a pair (lm_pair_loop_kernel), because every one of its <= 5 evaluations reads them
// and from global memory they were dependent round trips on the one wavefront's critical path (measured per evaluation
// of a single pair: 20 counts of listed records, one after the other, 5.2 us; edge records 4.2 us of a 15 us evaluation):
// the pair's edge records and the numbers of listed plane records of its moment tiles.
constexpr uint32_t kEdgeCache = 320;   // records (9 doubles each: 23 KB); a scan yields ~290 edge features
constexpr uint32_t kListCache = 64;    // moment tiles of a pair (4 per chunk of kSweepChunk slots)
// Round 5: the LISTED plane records of a pair (the ones its moments leave out: 0 - 1 % of the slots) as one flat list in LDS for
// the whole solve, like the edge records. They were read tile by tile in every evaluation — up to 20 rounds of two dependent
// global round trips with a handful of lanes each: a third of the kernel (stamped: 22 k of an evaluation's ~60 k cycles).
constexpr uint32_t kFlatCache = 192;   // records (7 doubles each: 10.5 KB); more than that: the tile-by-tile walk
__device__ __forceinline__ double light_eval(const RegBatch& B, size_t pair, const double x[7], uint32_t n_se, uint32_t n_sp,
                                             const double* mom, LightLds& L, const double (*s_edge)[kEdgeCache], const uint32_t* s_listed,
                                             const double (*s_frec)[kFlatCache], uint32_t n_flat) {
  const int lane = threadIdx.x;  // (64-thread workgroup)
  const size_t efield = B.n_pairs * B.edge_stride, pfield = B.n_pairs * B.planar_stride;
  const double* __restrict__ E = B.assoc.edge + pair * B.edge_stride;
  const double* __restrict__ Pl = B.assoc.plane + pair * B.planar_stride;
  double acc[kAccSize];
#pragma unroll
  for (int j = 0; j < kAccSize; j++) acc[j] = 0.0;
  for (uint32_t v = lane; v < n_se; v += 64) {
    const bool cached = v < kEdgeCache;  // (the same records in the same order either way)
    const double f0 = cached ? s_edge[0][v] : E[v];
    if (f0 == f0) {  // NaN in field 0 marks an invalid slot
      double prim[6];
#pragma unroll
      for (int f = 0; f < 6; f++) prim[f] = cached ? s_edge[3 + f][v] : E[(3 + f) * efield + v];
      residual_accumulate(false, v3(f0, cached ? s_edge[1][v] : E[efield + v], cached ? s_edge[2][v] : E[2 * efield + v]), prim, x, acc);
    }
  }
  if (n_flat != 0xFFFFFFFFu) {  // (uniform) the listed records from LDS, in list order
    for (uint32_t e = lane; e < n_flat; e += 64) {
      const double prim[6] = {s_frec[3][e], s_frec[4][e], s_frec[5][e], s_frec[6][e], 0.0, 0.0};
      residual_accumulate(true, v3(s_frec[0][e], s_frec[1][e], s_frec[2][e]), prim, x, acc);
    }
  } else {
    // Tile by tile, but every record on the lane — and in the per-lane order — the flat list gives it: entry e of the pair's
    // concatenated lists belongs to lane e % 64. (Round 6: each tile used to start at lane 0 again. Which of the two walks a pair
    // takes depends on the slot capacity of the call — more than kListCache tiles when a scan is registered against a map
    // through the plain entry point, fewer through a persistent index — so the two gave sums that differed in the last bit for
    // some inputs: tools/debug_c5.py, seeds 6 and 7 of config 5; the seed of tests/test_gpu_multi.py happened to agree.)
    uint32_t off = 0;  // uniform
    for (uint32_t lb = 0; lb < B.mom_blocks_per_pair * 4 && (lb / 4) * kSweepChunk < n_sp; lb++) {
      const uint32_t cnt = lb < kListCache ? s_listed[lb] : B.flagged_count[pair * B.mom_blocks_per_pair * 4 + lb];
      if (cnt == 0u) continue;  // (uniform)
      const uint32_t* __restrict__ fl = B.flagged_list + (pair * B.mom_blocks_per_pair * 4 + lb) * (size_t)(kSweepChunk / 4);
      for (uint32_t k = ((uint32_t)lane + 64u - (off & 63u)) & 63u; k < cnt; k += 64) {
        const uint32_t q = fl[k];
        const double prim[6] = {Pl[3 * pfield + q], Pl[4 * pfield + q], Pl[5 * pfield + q], Pl[6 * pfield + q], 0.0, 0.0};
        residual_accumulate(true, v3(Pl[q], Pl[pfield + q], Pl[2 * pfield + q]), prim, x, acc);
      }
      off += cnt;
    }
  }
  // every lane ends up with the wavefront's sums (xor butterfly: the same value in every lane, fixed order)
#pragma unroll
  for (int j = 0; j < kAccSize; j++) {
    double v = acc[j];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    acc[j] = v;
  }
  // ---- plane terms from the moments: [Z | phi] (13 x 7), T = M [Z | phi], then Z^T T
  wave_lds_fence();  // (L may still be read from the previous evaluation)
  for (int idx = lane; idx < kMomDim * 7; idx += 64) {
    const int j = idx / 7, a = idx % 7;
    const double ux = x[0], uy = x[1], uz = x[2], w = x[3];
    double val;
    if (a == 6) {  // phi_j
      const double phi[kMomDim] = {1.0, 2.0 * w * ux, 2.0 * w * uy, 2.0 * w * uz, ux * ux, uy * uy, uz * uz,
                                   ux * uy, ux * uz, uy * uz, x[4], x[5], x[6]};
      val = 0.0;
#pragma unroll
      for (int jj = 0; jj < kMomDim; jj++)
        if (jj == j) val = phi[jj];
    } else {  // Z_ja: ambient gradient of phi_j, then the tangent map (as plane_eval_from_moments)
      double A[kMomDim][4];
#pragma unroll
      for (int jj = 0; jj < kMomDim; jj++) A[jj][0] = A[jj][1] = A[jj][2] = A[jj][3] = 0.0;
      A[1][0] = 2.0 * w, A[1][3] = 2.0 * ux;
      A[2][1] = 2.0 * w, A[2][3] = 2.0 * uy;
      A[3][2] = 2.0 * w, A[3][3] = 2.0 * uz;
      A[4][0] = 2.0 * ux, A[5][1] = 2.0 * uy, A[6][2] = 2.0 * uz;
      A[7][0] = uy, A[7][1] = ux;
      A[8][0] = uz, A[8][2] = ux;
      A[9][1] = uz, A[9][2] = uy;
      double g[4] = {0, 0, 0, 0};
#pragma unroll
      for (int jj = 0; jj < kMomDim; jj++)
        if (jj == j) g[0] = A[jj][0], g[1] = A[jj][1], g[2] = A[jj][2], g[3] = A[jj][3];
      const double W = x[0], X = x[1], Y = x[2], Zq = x[3];
      const double z0 = g[0] * (-X) + g[1] * W + g[2] * (-Zq) + g[3] * Y;
      const double z1 = g[0] * (-Y) + g[1] * Zq + g[2] * W + g[3] * (-X);
      const double z2 = g[0] * (-Zq) + g[1] * (-Y) + g[2] * X + g[3] * W;
      val = a == 0 ? z0 : (a == 1 ? z1 : (a == 2 ? z2 : ((j == 10 + (a - 3)) ? 1.0 : 0.0)));
    }
    L.Z[j][a] = val;
  }
  wave_lds_fence();
  for (int idx = lane; idx < kMomDim * 7; idx += 64) {
    const int i = idx / 7, a = idx % 7;
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < kMomDim; j++) t += mom[i * kMomStride + j] * L.Z[j][a];
    L.T[i][a] = t;
  }
  wave_lds_fence();
  double v = 0.0;
  if (lane < kAccSize) {
    const int o = lane;
    v = 0.0;
#pragma unroll
    for (int j = 0; j < kAccSize; j++)
      if (j == o) v = acc[j];
    // plane terms: o < 21: (Z^T T)[a][b]; 21..26: Z^T (M phi); 27: phi . (M phi) / 2; 28: non-finite flag
    if (o < 21) {
      int a = 0, rem = o;
      while (rem >= 6 - a) rem -= 6 - a, a++;
      const int b = a + rem;
      double sum = 0.0;
#pragma unroll
      for (int i = 0; i < kMomDim; i++) sum += L.Z[i][a] * L.T[i][b];
      v += sum;
    } else if (o < 27) {
      const int a = o - 21;
      double sum = 0.0;
#pragma unroll
      for (int i = 0; i < kMomDim; i++) sum += L.Z[i][a] * L.T[i][6];
      v += sum;
    } else {
      double cost = 0.0;
#pragma unroll
      for (int i = 0; i < kMomDim; i++) cost += L.Z[i][6] * L.T[i][6];
      if (o == 27) v += 0.5 * cost;
      else if (!(cost - cost == 0.0)) v += 1.0;
    }
  }
  return v;
}

/* All association records of a pair streamed by one wavefront (an evaluation its moments cannot stand in for: the
 * candidate has left their validity bound — rare). Lanes 0..kAccSize-1 return their entry of the sums. */
__device__ __forceinline__ double stream_eval_wave(const RegBatch& B, size_t pair, const double x[7], uint32_t n_se, uint32_t n_sp) {
  const int lane = threadIdx.x;
  const size_t efield = B.n_pairs * B.edge_stride, pfield = B.n_pairs * B.planar_stride;
  const double* __restrict__ E = B.assoc.edge + pair * B.edge_stride;
  const double* __restrict__ Pl = B.assoc.plane + pair * B.planar_stride;
  double acc[kAccSize];
#pragma unroll
  for (int j = 0; j < kAccSize; j++) acc[j] = 0.0;
  for (uint32_t v = lane; v < n_se; v += 64) {
    const double f0 = E[v];
    if (f0 == f0) {
      double prim[6];
#pragma unroll
      for (int f = 0; f < 6; f++) prim[f] = E[(3 + f) * efield + v];
      residual_accumulate(false, v3(f0, E[efield + v], E[2 * efield + v]), prim, x, acc);
    }
  }
  for (uint32_t q = lane; q < n_sp; q += 64) {
    const double f0 = Pl[q];
    if (f0 == f0) {
      const double prim[6] = {Pl[3 * pfield + q], Pl[4 * pfield + q], Pl[5 * pfield + q], Pl[6 * pfield + q], 0.0, 0.0};
      residual_accumulate(true, v3(f0, Pl[pfield + q], Pl[2 * pfield + q]), prim, x, acc);
    }
  }
  double mine = 0.0;
#pragma unroll
  for (int j = 0; j < kAccSize; j++) {
    double v = acc[j];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if (lane == j) mine = v;
  }
  return mine;
}

// the reduction of a pair's partials is done: one Levenberg-Marquardt bookkeeping step, and whether the moments can
// stand in for the plane records at the next candidate
__device__ __forceinline__ void lm_step_pair(const RegBatch& B, size_t pair, PairState& S, const double* acc) {
  const double* __restrict__ mom = B.moments + pair * (size_t)(kMomSize + 2);
  LmState lm = S.lm;
  const bool first = S.first_sweep != 0;
  lm_consume(lm, acc, first);
  S.lm = lm;
  S.first_sweep = 0;
  if (first && B.ref_moments && !S.use_moments && lm.active) {
    // First ICF iteration, after its evaluation at the identity update: the rest of the solve runs off moments taken AT the
    // first candidate (moment_kernel next, then lm_pair_loop_kernel). At the identity the bound of plane_moments_valid_at
    // fails for every later candidate (the first step moves far points by more than the Huber threshold); relative to the
    // first candidate the later ones are refinements.
    S.use_moments = 1u, S.mom_ref_on = 1u;
    for (int i = 0; i < 7; i++) S.mom_ref[i] = lm.xeval[i];
    S.stream_planes = 1u;  // (decided by lm_pair_loop_kernel once the moments exist)
    return;
  }
  // the next sweep: may the moments stand in for the plane records at the new candidate?
  const bool valid = S.mom_ref_on ? plane_moments_valid_rel(mom[kMomSize], mom[kMomSize + 1], lm.xeval, S.mom_ref)
                                  : plane_moments_valid_at(mom[kMomSize], mom[kMomSize + 1], lm.xeval);
  S.stream_planes = (!S.use_moments || (lm.active && !valid)) ? 1u : 0u;
}

__global__ __launch_bounds__(64) void lm_step_kernel(RegBatch B) {  // one wavefront: the 6x6 solve may use the whole register file
  const size_t pair = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= B.n_pairs) return;
  PairState& S = B.state[pair];
  if (!S.active || !S.lm.active) return;
  const uint32_t n_se_raw = B.n_src_edge[pair * B.in_pitch], n_sp_raw = B.n_src_planar[pair * B.in_pitch];
  const uint32_t n_se = n_se_raw < B.edge_stride ? n_se_raw : (uint32_t)B.edge_stride;
  const uint32_t n_sp = n_sp_raw < B.planar_stride ? n_sp_raw : (uint32_t)B.planar_stride;
  const uint32_t used = S.stream_planes ? (n_se + n_sp + kSweepChunk - 1) / kSweepChunk : 1u;  // blocks that wrote a partial
  double acc[kAccSize];
  for (int j = 0; j < kAccSize; j++) acc[j] = 0.0;
  for (uint32_t b = 0; b < used && b < B.blocks_per_pair; b++) {
    const double* __restrict__ p = B.partials + (pair * B.blocks_per_pair + b) * kAccSize;
    for (int j = 0; j < kAccSize; j++) acc[j] += p[j];
  }
  lm_step_pair(B, pair, S, acc);
}

// row a23 for one pair: est <- update (+) est, convergence test, termination type (registration-inl.h:59-76)
__device__ __forceinline__ void outer_update_pair(const RegBatch& B, const RegConfig& C, size_t pair, PairState& S) {
  const uint32_t iteration = S.iterations;
  double upd[7];
  for (int i = 0; i < 7; i++) upd[i] = S.lm.x_user[i];
  if (B.iter_info) {
    loamx_iter_info& I = B.iter_info[pair * C.max_iterations + iteration];
    for (int i = 0; i < 7; i++) I.estimate_update[i] = upd[i];
  }
  S.iterations = iteration + 1;
  double est[7];
  for (int i = 0; i < 7; i++) est[i] = S.est[i];
  const bool converged = outer_update(est, upd, C.rot_thresh, C.pos_thresh);  // registration-inl.h:65-73
  // From the second ICF iteration on the updates are small and the moment pass pays off (the first one usually
  // moves the pose by more than the validity bound of the moments allows: its sweeps stream the records).
  S.use_moments = (C.flags & kRegFlagNoMoments) ? 0u : 1u;
  for (int i = 0; i < 7; i++) S.est[i] = est[i];
  if (converged) {
    S.termination = LOAMX_CONVERGED;
    S.active = 0;
  } else if (iteration + 1 >= C.max_iterations) {
    S.active = 0;  // termination stays MAX_ITER
  } else {
    atomicAdd(B.n_active, 1u);
  }
}

/* The whole Levenberg-Marquardt solve of one ICF iteration for a pair whose plane records are summarised by moments
 * (every ICF iteration but the first), by ONE WAVEFRONT in ONE launch: the moment tiles of the pair are added up,
 * then up to five times { evaluate at the candidate, bookkeeping step }. Before, that was moment_finish + 5 x
 * { sweep_kernel (nothing to do), sweep_light_kernel, lm_step_kernel }: sixteen dependent launches of a few
 * microseconds each, 9-10 us apart (0.39 ms per ICF iteration). One wavefront per pair because the bookkeeping step
 * (a 6x6 Cholesky unrolled in registers) wants most of the register file: at one wavefront per SIMD a chip still
 * holds 1 024 pairs at once. An evaluation the moments cannot stand in for (the candidate leaves their validity bound)
 * is streamed by the same wavefront: rare, slow, self-contained. From the second ICF iteration on this kernel is the
 * only evaluator, so a pair's result does not depend on the batch it is in. */
__global__ __launch_bounds__(64) void lm_pair_loop_kernel(RegBatch B, RegConfig C) {
  __shared__ LightLds L;
  __shared__ double s_mom[kMomSize + 2];
  __shared__ double s_acc[kAccSize];
  __shared__ double s_edge[9][kEdgeCache];
  __shared__ double s_frec[7][kFlatCache];
  __shared__ uint32_t s_listed[kListCache], s_pref[kListCache];
  const int lane = threadIdx.x;
  const size_t pair = blockIdx.x;
  PairState& S_global = B.state[pair];
  if (!S_global.active) return;  // uniform
  if (!S_global.lm.active) {     // (uniform; lm_begin_kernel starts every active pair's solve, so this does not happen)
    if (lane == 0) outer_update_pair(B, C, pair, S_global);
    return;
  }
  // Round 6: the pair's state lives in LDS for the length of the solve. Every bookkeeping step read and wrote the 700-byte
  // PairState in global memory through lane 0 alone, and the other lanes fetched the next candidate from there: dependent
  // round trips on the one wavefront's critical path, five times per launch.
  __shared__ PairState s_S;
  static_assert(sizeof(PairState) % 8 == 0, "copied as 64-bit words");
  {
    const unsigned long long* __restrict__ src = reinterpret_cast<const unsigned long long*>(&S_global);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&s_S);
    for (int w = lane; w < (int)(sizeof(PairState) / 8); w += 64) dst[w] = src[w];
  }
  wave_lds_fence();
  PairState& S = s_S;
  const uint32_t n_se_raw = B.n_src_edge[pair * B.in_pitch], n_sp_raw = B.n_src_planar[pair * B.in_pitch];
  const uint32_t n_se = n_se_raw < B.edge_stride ? n_se_raw : (uint32_t)B.edge_stride;
  const uint32_t n_sp = n_sp_raw < B.planar_stride ? n_sp_raw : (uint32_t)B.planar_stride;
  {  // the pair's edge records and listed-record counts into LDS (independent loads: one round trip)
    const size_t efield = B.n_pairs * B.edge_stride;
    const double* __restrict__ E = B.assoc.edge + pair * B.edge_stride;
    const uint32_t nc = n_se < kEdgeCache ? n_se : kEdgeCache;
    for (uint32_t v = lane; v < nc; v += 64) {
#pragma unroll
      for (int f = 0; f < 9; f++) s_edge[f][v] = E[f * efield + v];
    }
    static_assert(kListCache == 64, "one lane per tile");
    s_listed[lane] = (uint32_t)lane < B.mom_blocks_per_pair * 4 ? B.flagged_count[pair * B.mom_blocks_per_pair * 4 + lane] : 0u;
  }
  // ---- the pair's moment matrix: its wavefront tiles in a fixed order
  uint32_t stream = 1u;
  if (S.use_moments) {
    uint32_t used = (n_sp + kSweepChunk - 1) / kSweepChunk;
    used = used < B.mom_blocks_per_pair ? used : B.mom_blocks_per_pair;
    const double* __restrict__ part = B.mom_partials + pair * B.mom_blocks_per_pair * 4 * (size_t)(kMomSize + 2);
    double* __restrict__ mom = B.moments + pair * (size_t)(kMomSize + 2);
    double v[kMomSize / 64];
#pragma unroll
    for (int c = 0; c < kMomSize / 64; c++) v[c] = 0.0;
    double s0max = 0.0, v2max = 0.0;
    // (four tiles' loads in flight at a time, added in tile order: the sum is the same, the wavefront waits for five round
    // trips instead of twenty)
    constexpr uint32_t kTileBatch = 4;
    for (uint32_t t0 = 0; t0 < used * 4; t0 += kTileBatch) {
      double w[kTileBatch][kMomSize / 64], m0[kTileBatch], m1[kTileBatch];
#pragma unroll
      for (uint32_t b = 0; b < kTileBatch; b++) {
        const bool in = t0 + b < used * 4;
        const size_t off = (in ? t0 + b : t0) * (size_t)(kMomSize + 2);
#pragma unroll
        for (int c = 0; c < kMomSize / 64; c++) w[b][c] = in ? part[off + c * 64 + lane] : 0.0;
        m0[b] = in ? part[off + kMomSize] : 0.0, m1[b] = in ? part[off + kMomSize + 1] : 0.0;  // (uniform loads; both maxima are >= 0)
      }
#pragma unroll
      for (uint32_t b = 0; b < kTileBatch; b++) {
#pragma unroll
        for (int c = 0; c < kMomSize / 64; c++) v[c] += w[b][c];
        s0max = fmax(s0max, m0[b]), v2max = fmax(v2max, m1[b]);
      }
    }
#pragma unroll
    for (int c = 0; c < kMomSize / 64; c++) mom[c * 64 + lane] = v[c], s_mom[c * 64 + lane] = v[c];
    if (lane == 0) mom[kMomSize] = s0max, mom[kMomSize + 1] = v2max, s_mom[kMomSize] = s0max, s_mom[kMomSize + 1] = v2max;
    const double ident[7] = {0, 0, 0, 1, 0, 0, 0};
    // (first ICF iteration: the moments were taken at the candidate this loop starts with — S.mom_ref == S.lm.xeval)
    stream = (S.mom_ref_on ? plane_moments_valid_rel(s0max, v2max, S.lm.xeval, S.mom_ref) : plane_moments_valid_at(s0max, v2max, ident)) ? 0u : 1u;
  }
  if (lane == 0) S.stream_planes = stream;
  // ---- the listed plane records, flat (tile order, list order inside a tile) — when the moments are in use and the lists fit
  uint32_t n_flat = 0xFFFFFFFFu;
  if (S.use_moments && B.mom_blocks_per_pair * 4 <= kListCache) {
    const uint32_t tiles = B.mom_blocks_per_pair * 4;
    const uint32_t mine = ((uint32_t)lane < tiles && ((uint32_t)lane / 4) * kSweepChunk < n_sp) ? s_listed[lane] : 0u;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (total <= kFlatCache) {  // uniform
      s_pref[lane] = incl - mine;  // first flat entry of tile `lane`
      wave_lds_fence();
      const size_t pfield = B.n_pairs * B.planar_stride;
      const double* __restrict__ Pl = B.assoc.plane + pair * B.planar_stride;
      for (uint32_t e = lane; e < total; e += 64) {
        uint32_t lb = 0;  // the tile of flat entry e: the last one whose first entry is <= e (tiles <= 64: a short scan of LDS words)
        for (uint32_t t = 1; t < tiles; t++) lb = s_pref[t] <= e ? t : lb;
        // (an empty tile shares its first entry with its successor: the LAST such tile is the one that holds e)
        const uint32_t k = e - s_pref[lb];
        const uint32_t q = B.flagged_list[(pair * B.mom_blocks_per_pair * 4 + lb) * (size_t)(kSweepChunk / 4) + k];
#pragma unroll
        for (int f = 0; f < 7; f++) s_frec[f][e] = Pl[f * pfield + q];
      }
      n_flat = total;
    }
  }
  uint32_t lm_active = 1u;
  double x[7];
#pragma unroll
  for (int i = 0; i < 7; i++) x[i] = S.lm.xeval[i];
  wave_lds_fence();
  for (int k = 0; k < 5 && lm_active; k++) {  // iteration-0 evaluation + max_num_iterations = 4 candidates
    const double v = stream ? stream_eval_wave(B, pair, x, n_se, n_sp) : light_eval(B, pair, x, n_se, n_sp, s_mom, L, s_edge, s_listed, s_frec, n_flat);
    if (lane < kAccSize) s_acc[lane] = 0.0 + v;
    wave_lds_fence();
    if (lane == 0) {
      double acc[kAccSize];
      for (int j = 0; j < kAccSize; j++) acc[j] = s_acc[j];
      lm_step_pair(B, pair, S, acc);
    }
    wave_lds_fence();
    // (lane 0's stores to S, read back by every lane: same wavefront, program order)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    lm_active = __builtin_amdgcn_readfirstlane((int)(lane == 0 ? S.lm.active : 0u));
    stream = __builtin_amdgcn_readfirstlane((int)(lane == 0 ? S.stream_planes : 0u));
#pragma unroll
    for (int i = 0; i < 7; i++) {
      const double xi = lane == 0 ? S.lm.xeval[i] : 0.0;
      x[i] = __longlong_as_double(((long long)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(xi) >> 32)) << 32) |
                                  (unsigned int)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(xi) & 0xFFFFFFFFll)));
    }
  }
  // ---- the outer step of the ICF iteration (what outer_update_kernel does after the first iteration's solve)
  if (lane == 0) outer_update_pair(B, C, pair, S);
  wave_lds_fence();
  {
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&s_S);
    unsigned long long* __restrict__ dst = reinterpret_cast<unsigned long long*>(&S_global);
    for (int w = lane; w < (int)(sizeof(PairState) / 8); w += 64) dst[w] = src[w];
  }
}

/* ------------------------------------------------------------------------------------------------
 * Moment pass (reg_math.h: "Plane residuals through moments"): M = sum c c^T over the plane records of a
 * pair is a Gram matrix — the one dense contraction on this path — and is accumulated with
 * v_mfma_f64_16x16x4f64: A = B^T = 4 coefficient vectors (13 numbers, padded to 16) per instruction.
 * Every lane loads one record (coalesced), computes its 13 coefficients, stages them in its wavefront's
 * LDS tile, and the wavefront then issues 16 MFMAs over the 64 staged records (lane l feeds c[l % 16]
 * of record 4 t + l / 16 as both operands). Same chunking as sweep_kernel; every wavefront writes its
 * 16x16 tile, lm_pair_loop_kernel adds the tiles of a pair in a fixed order.
 * ---------------------------------------------------------------------------------------------- */
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int kMomLdsRow = 65;  // 64 records + 1: the 16 rows of a tile start in different banks

__global__ __launch_bounds__(64) void moment_kernel(RegBatch B) {  // one wavefront per workgroup: wavefronts share nothing
  __shared__ double s_c[kMomStride][kMomLdsRow];
  const int wave = (int)(blockIdx.x & 3u), lane = (int)threadIdx.x;  // the wavefront's place in its chunk of kSweepChunk slots
  const int tix = wave * 64 + lane;
  const size_t pair = (blockIdx.x >> 2) / B.mom_blocks_per_pair;
  const uint32_t blk = (blockIdx.x >> 2) % B.mom_blocks_per_pair;
  const PairState& S = B.state[pair];
  if (!S.active || !S.lm.active || !S.use_moments) return;  // uniform per workgroup
  const uint32_t n_sp_raw = B.n_src_planar[pair * B.in_pitch];
  const uint32_t n_sp = n_sp_raw < B.planar_stride ? n_sp_raw : (uint32_t)B.planar_stride;
  const uint32_t base = blk * kSweepChunk;
  if (base >= n_sp) return;  // uniform per workgroup
  if (blk == 0 && tix == 0 && B.sweep_slots) atomicAdd(&B.sweep_slots[4], (unsigned long long)n_sp);
  const size_t pfield = B.n_pairs * B.planar_stride;
  const double* __restrict__ Pl = B.assoc.plane + pair * B.planar_stride;
  double(*tile)[kMomLdsRow] = s_c;
#pragma unroll
  for (int j = kMomDim; j < kMomStride; j++) tile[j][lane] = 0.0;  // padding rows
  v4f64 acc = {0.0, 0.0, 0.0, 0.0};
  double s0max = 0.0, v2max = 0.0;
  // the candidate the moments are taken at: the identity update (s_i = c_i[0]), or — first ICF iteration — the first candidate
  const bool ref_on = S.mom_ref_on != 0u;  // uniform
  double phi_ref[kMomDim];
  {
    double xr[7];
#pragma unroll
    for (int i = 0; i < 7; i++) xr[i] = ref_on ? S.mom_ref[i] : (i == 3 ? 1.0 : 0.0);
    plane_phi(xr, phi_ref);
  }
  // the wavefront's own list of flagged slots (a quarter of the workgroup's chunk of capacity), in slot order
  uint32_t* __restrict__ flist = B.flagged_list + ((pair * B.mom_blocks_per_pair + blk) * 4 + wave) * (size_t)(kSweepChunk / 4);
  uint32_t n_flagged = 0;  // uniform in the wavefront
  struct Rec {
    double f[7];
    bool in;
  };
  auto load_rec = [&](uint32_t q) {
    Rec R;
    R.in = q < n_sp;
#pragma unroll
    for (int f = 0; f < 7; f++) R.f[f] = R.in ? Pl[f * pfield + q] : 0.0;
    return R;
  };
  constexpr int kIters = kSweepChunk / kSweepThreads;  // sub-chunks of 256 records (64 per wavefront)
  Rec cur = load_rec(base + tix);
#pragma unroll 1
  for (int it = 0; it < kIters; it++) {
    const Rec nxt = (it + 1 < kIters) ? load_rec(base + (it + 1) * kSweepThreads + tix) : Rec{{0, 0, 0, 0, 0, 0, 0}, false};
    double c[kMomDim];
#pragma unroll
    for (int j = 0; j < kMomDim; j++) c[j] = 0.0;
    bool flagged = false;
    if (cur.in && cur.f[0] == cur.f[0]) {  // NaN in field 0 marks an invalid slot
      const Vec3 v = v3(cur.f[0], cur.f[1], cur.f[2]);
      plane_coeffs(v, v3(cur.f[3], cur.f[4], cur.f[5]), cur.f[6], c);
      double sref = c[0];
      if (ref_on) {  // s_i at the reference candidate
        sref = 0.0;
#pragma unroll
        for (int j = 0; j < kMomDim; j++) sref += c[j] * phi_ref[j];
      }
      const double a0 = fabs(sref), vv = vdot(v, v);
      // a record that starts far from its plane may reach the Huber threshold: it stays out of the moments and
      // is listed (in slot order) for the sweeps, which evaluate the listed records one by one
      flagged = !(a0 <= kMomInlier);
      if (flagged) {
#pragma unroll
        for (int j = 0; j < kMomDim; j++) c[j] = 0.0;
      } else {
        s0max = a0 > s0max ? a0 : s0max;
        v2max = vv > v2max ? vv : v2max;
      }
    }
    // (the tile and the list are private to the wavefront: no workgroup barrier in this loop, only the ordering
    // of this wavefront's own LDS traffic)
    const unsigned long long fm = __ballot(flagged);
    if (flagged) flist[n_flagged + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull))] = base + it * kSweepThreads + tix;
    n_flagged += (uint32_t)__popcll(fm);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < kMomDim; j++) tile[j][lane] = c[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int t = 0; t < 16; t++) {  // (four independent accumulator chains were measured: slower)
      const double a = tile[lane & 15][4 * t + (lane >> 4)];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
    cur = nxt;
  }
  // the wavefront's tile: lane l holds D[l / 16 + 4 r][l % 16], r = 0..3 (measured: tools/probes/mfma_f64_layout.hip)
  double* __restrict__ out = B.mom_partials + ((pair * B.mom_blocks_per_pair + blk) * 4 + wave) * (size_t)(kMomSize + 2);
#pragma unroll
  for (int r = 0; r < 4; r++) out[((lane >> 4) + 4 * r) * kMomStride + (lane & 15)] = acc[r];
  s0max = wave_max(s0max), v2max = wave_max(v2max);
  if (lane == 0) out[kMomSize] = s0max, out[kMomSize + 1] = v2max;
  if (lane == 0) B.flagged_count[(pair * B.mom_blocks_per_pair + blk) * 4 + wave] = n_flagged;
}

// one workgroup per pair: fixed-order sum of the wavefront tiles, the two maxima, and whether the first
// sweep (x = identity) may use the moments
__global__ void outer_update_kernel(RegBatch B, RegConfig C) {
  const size_t pair = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= B.n_pairs) return;
  PairState& S = B.state[pair];
  if (!S.active) return;
  outer_update_pair(B, C, pair, S);
}

__global__ void write_results_kernel(RegBatch B, loamx_reg_result* __restrict__ out) {
  const size_t pair = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= B.n_pairs) return;
  if (pair == 0 && B.max_counts) {  // ready for the next call's state_init_kernel (the host read them long ago)
    for (int c = 0; c < 4; c++) B.max_counts[c] = 0u;
    B.max_counts[4] = B.max_counts[5] = 0xFFFFFFFFu;
  }
  const PairState& S = B.state[pair];
  for (int i = 0; i < 7; i++) out[pair].pose[i] = S.est[i];
  out[pair].termination = S.termination;
  out[pair].iterations = S.iterations;
}

inline unsigned per_pair_grid(size_t n_pairs) { return (unsigned)((n_pairs + 63) / 64); }

/* ------------------------------------------------------------------------------------------------
 * Direct read-outs (round 3): the device functions of rows a16-a18 behind host-callable entry points, for the
 * internal namespaces of the header shim (geometry_internal::fitLine / fitPlane, kdtree_internal::knnSearch) and for
 * parity tests that compare neighbour lists and fits with the oracle one by one instead of through the poses.
 * ---------------------------------------------------------------------------------------------- */
// geometry.cpp:42-73 on n_sets point sets of k points each (k x 3 row-major): the fit_line / fit_plane the association
// kernels call. prim_out: 6 (a, b) or 4 (normal, d) doubles per set; aux_out: condition number (always DBL_MAX,
// SURVEY Q6) or the signed mean distance.
template <bool PLANE, int KM>
__global__ __launch_bounds__(64) void fit_sets_kernel(const double* __restrict__ pts, size_t n_sets, int k, double* __restrict__ prim_out,
                                                      double* __restrict__ aux_out) {
  const size_t s = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (s >= n_sets) return;
  Vec3 nb[KM];
#pragma unroll
  for (int j = 0; j < KM; j++) {
    const size_t at = (s * (size_t)k + (size_t)(j < k ? j : 0)) * 3;
    nb[j] = j < k ? v3(pts[at], pts[at + 1], pts[at + 2]) : v3(0, 0, 0);
  }
  if (PLANE) {
    Vec3 nrm;
    double d;
    aux_out[s] = fit_plane<KM>(nb, k, nrm, d);
    prim_out[4 * s] = nrm.x, prim_out[4 * s + 1] = nrm.y, prim_out[4 * s + 2] = nrm.z, prim_out[4 * s + 3] = d;
  } else {
    Vec3 a, b;
    fit_line<KM>(nb, k, a, b);
    aux_out[s] = kDblMax;  // geometry.cpp:55-56: the ratio is computed and dropped
    prim_out[6 * s] = a.x, prim_out[6 * s + 1] = a.y, prim_out[6 * s + 2] = a.z;
    prim_out[6 * s + 3] = b.x, prim_out[6 * s + 4] = b.y, prim_out[6 * s + 5] = b.z;
  }
}

// kdtree.cpp:10-28 for a batch of queries against one indexed set (pair 0 of gs): the complete search of the queue
// kernels (keyed collector over all rounds, exact collector for what the keys leave undecided). idx_out: k entries per
// query (indices into the caller's point array, ascending distance; 0xFFFFFFFF past the count).
template <int KM>
__global__ __launch_bounds__(64) void knn_queries_kernel(GridSet gs, const double* __restrict__ q, size_t n_q, int k, double max_dist,
                                                         double pass_max, uint32_t* __restrict__ idx_out, uint32_t* __restrict__ count_out) {
  __shared__ uint32_t s_rows[kLeanRowWords * 64];
  const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n_q) return;
  const GridDesc g = gs.desc[0];
  uint32_t pos[KM];
  const int kk = k < KM ? k : KM;
  const Vec3 p = v3(q[3 * i], q[3 * i + 1], q[3 * i + 2]);
  int kept;
  if (g.n_points <= kBruteMax) {
    // (sets this small are searched exhaustively by the association kernels too, and the index build leaves their cell
    // table unwritten: grid_build_kernel)
    KnnResult<KM> r;
    knn_init(r);
    brute_scan_tile(r, kk, p, gs.sorted, 0u, g.n_points);  // (the set has kGridPad spare entries behind it)
    kept = knn_finish(r, kk, max_dist);
    knn_shift_positions<KM>(r.pos, kk, pos);
  } else {
    kept = knn_search_positions<KM>(g, gs.cell_start, gs.sorted, p, kk, max_dist, pass_max, pos, s_rows + threadIdx.x, 64);
  }
  count_out[i] = (uint32_t)kept;
#pragma unroll
  for (int u = 0; u < KM; u++) {  // (slot u holds neighbour u - (KM - kk): static register indices, see knn_shift_positions)
    const int j = u - (KM - kk);
    if (j >= 0) idx_out[i * (size_t)k + j] = j < kept ? gs.sorted[pos[u]].orig : 0xFFFFFFFFu;
  }
  for (int j = KM; j < k; j++) idx_out[i * (size_t)k + j] = 0xFFFFFFFFu;
}

// One association pass of pair 0 as the association kernels left it (launch_associate): per source feature, by the
// CALLER's index: neighbour count and target indices in ascending order, the moved point, the fitted primitive, valid.
// Threads [0, n_src) read the round-1 results, threads [n_src, n_src + queued) the queue's.
template <bool PLANE, int KM>
__global__ __launch_bounds__(256) void assoc_dump_kernel(RegBatch B, RegConfig C, AssocDumpSet D) {
  const size_t stride = PLANE ? B.planar_stride : B.edge_stride;
  const uint32_t n_raw = PLANE ? B.n_src_planar[0] : B.n_src_edge[0];
  const uint32_t n_src = n_raw < stride ? n_raw : (uint32_t)stride;
  const uint32_t queued = B.assoc.n_assoc[PLANE ? 3 : 2];
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_src + queued) return;
  const GridSet& gs = PLANE ? B.grid_plane : B.grid_edge;
  const GridSet& src_gs = PLANE ? B.src_grid_plane : B.src_grid_edge;
  const uint32_t* __restrict__ nn = PLANE ? B.assoc.nn_plane : B.assoc.nn_edge;
  const uint32_t* __restrict__ rnn = PLANE ? B.assoc.rnn_plane : B.assoc.rnn_edge;
  const uint32_t* __restrict__ rest = PLANE ? B.assoc.rest_plane : B.assoc.rest_edge;
  uint32_t i, at;
  const uint32_t* __restrict__ src;
  if (t < n_src) {
    i = t, at = t, src = nn;
    if (nn[t] == 0xFFFFFFFFu) return;  // queued: written from the queue's results below
  } else {
    at = t - n_src, i = rest[at] & kQueueIndex, src = rnn;
  }
  const uint32_t orig = src_gs.sorted[i].orig;
  const int kq = PLANE ? C.k_plane : C.k_edge, kk = kq < KM ? kq : KM, shift = KM - kk;
  const uint32_t kept = src[at];
  D.nn_count[orig] = kept;
  for (int j = 0; j < kk; j++) {
    const uint32_t pos = src[(size_t)(1 + shift + j) * stride + at];
    D.nn_idx[(size_t)orig * (size_t)kq + j] = (uint32_t)j < kept && pos < gs.stride ? gs.sorted[pos].orig : 0xFFFFFFFFu;
  }
  const double* __restrict__ rec = PLANE ? B.assoc.plane : B.assoc.edge;
  const double f0 = rec[i];
  D.valid[orig] = f0 == f0 ? 1 : 0;
  // (an invalid slot holds NaN in field 0: the moved point's x is recomputed from the source point)
  const GridPoint sq = src_gs.sorted[i];
  const Vec3 p = pose_act(B.state[0].est, v3(sq.x, sq.y, sq.z));
  D.moved[3 * (size_t)orig] = p.x, D.moved[3 * (size_t)orig + 1] = rec[stride + i], D.moved[3 * (size_t)orig + 2] = rec[2 * stride + i];
  for (int f = 0; f < (PLANE ? 4 : 6); f++) D.prim[(size_t)orig * (PLANE ? 4 : 6) + f] = rec[(size_t)(3 + f) * stride + i];
}

}  // namespace

void launch_fit_sets(bool plane, const double* d_pts, size_t n_sets, int k, double* d_prim, double* d_aux, hipStream_t s) {
  if (n_sets == 0) return;
  const dim3 grid((unsigned)((n_sets + 63) / 64));
  if (k <= 5) {  // (the instantiation the association kernels use for the reference's default of 5 neighbours)
    if (plane) launch_kernel((fit_sets_kernel<true, 5>), grid, dim3(64), 0, s, d_pts, n_sets, k, d_prim, d_aux);
    else launch_kernel((fit_sets_kernel<false, 5>), grid, dim3(64), 0, s, d_pts, n_sets, k, d_prim, d_aux);
  } else if (k <= 8) {
    if (plane) launch_kernel((fit_sets_kernel<true, 8>), grid, dim3(64), 0, s, d_pts, n_sets, k, d_prim, d_aux);
    else launch_kernel((fit_sets_kernel<false, 8>), grid, dim3(64), 0, s, d_pts, n_sets, k, d_prim, d_aux);
  } else {
    if (plane) launch_kernel((fit_sets_kernel<true, kFitMaxK>), grid, dim3(64), 0, s, d_pts, n_sets, k, d_prim, d_aux);
    else launch_kernel((fit_sets_kernel<false, kFitMaxK>), grid, dim3(64), 0, s, d_pts, n_sets, k, d_prim, d_aux);
  }
}

void launch_knn_queries(const GridSet& gs, const double* d_q, size_t n_q, int k, double max_dist, uint32_t* d_idx, uint32_t* d_count,
                        hipStream_t s) {
  if (n_q == 0) return;
  const dim3 grid((unsigned)((n_q + 63) / 64));
  const double pass_max = knn_radius_pass_max(max_dist);
  if (k <= 5) launch_kernel((knn_queries_kernel<5>), grid, dim3(64), 0, s, gs, d_q, n_q, k, max_dist, pass_max, d_idx, d_count);
  else if (k <= 8) launch_kernel((knn_queries_kernel<8>), grid, dim3(64), 0, s, gs, d_q, n_q, k, max_dist, pass_max, d_idx, d_count);
  else launch_kernel((knn_queries_kernel<16>), grid, dim3(64), 0, s, gs, d_q, n_q, k, max_dist, pass_max, d_idx, d_count);
}

void launch_assoc_dump(const RegBatch& B, const RegConfig& C, const AssocDumpSet& edge, const AssocDumpSet& plane, hipStream_t s) {
  // (both queues can be as long as the sets: threads for n_src + queued)
  const dim3 ge((unsigned)((2 * B.edge_stride + 255) / 256)), gp((unsigned)((2 * B.planar_stride + 255) / 256));
  if (edge.nn_count) {
    if (C.k_edge <= 5) launch_kernel((assoc_dump_kernel<false, 5>), ge, dim3(256), 0, s, B, C, edge);
    else if (C.k_edge <= 8) launch_kernel((assoc_dump_kernel<false, 8>), ge, dim3(256), 0, s, B, C, edge);
    else launch_kernel((assoc_dump_kernel<false, 16>), ge, dim3(256), 0, s, B, C, edge);
  }
  if (plane.nn_count) {
    if (C.k_plane <= 5) launch_kernel((assoc_dump_kernel<true, 5>), gp, dim3(256), 0, s, B, C, plane);
    else if (C.k_plane <= 8) launch_kernel((assoc_dump_kernel<true, 8>), gp, dim3(256), 0, s, B, C, plane);
    else launch_kernel((assoc_dump_kernel<true, 16>), gp, dim3(256), 0, s, B, C, plane);
  }
}

// map-sized target sets: the multi-workgroup build (needs kBigScratchBytes of scratch per pair)
static bool grid_big(size_t stride, const GridPoint* scratch, uint32_t flags) {
  return !grid_small(stride, flags) && scratch != nullptr && stride * sizeof(GridPoint) >= kBigScratchBytes && !(flags & kRegFlagNoBigGrid);
}  // (a caller that sets GridSet::cells_cap provides 64 + 4 * (cells_cap + cells_cap / 4096 + 8) bytes of scratch and one pair)
static void launch_grid_build_big(size_t n_pairs, const double* pts, const uint32_t* n_pts, size_t stride, uint32_t in_pitch,
                                  double max_dist, const GridSet& gs, GridPoint* scratch, hipStream_t s) {
  const dim3 chunks((unsigned)((stride + kBigChunk - 1) / kBigChunk), (unsigned)n_pairs);
  launch_kernel(gridbig_init_kernel, dim3((unsigned)n_pairs), dim3(64), 0, s, scratch, stride);
  launch_kernel(gridbig_bbox_kernel, chunks, dim3(kBigThreads), 0, s, pts, n_pts, stride, in_pitch, scratch);
  launch_kernel(gridbig_choose_kernel, dim3((unsigned)((n_pairs + 63) / 64)), dim3(64), 0, s, n_pts, stride, in_pitch, max_dist, gs,
                     scratch, n_pairs);
  (void)hipMemsetAsync(gs.cell_start, 0, (gs.cells_cap ? (size_t)gs.cells_cap + 1 : n_pairs * (size_t)(kGridCellsCap + 1)) * sizeof(uint32_t), s);
  launch_kernel(gridbig_pass_kernel<false>, chunks, dim3(kBigThreads), 0, s, pts, n_pts, stride, in_pitch, gs, scratch);
  const unsigned n_tiles = (unsigned)(((gs.cells_cap ? gs.cells_cap : kGridCellsCap) + 1 + kScanTile - 1) / kScanTile);
  launch_kernel(gridbig_tile_sum_kernel, dim3(n_tiles, (unsigned)n_pairs), dim3(kScanThreads), 0, s, stride, gs, scratch);
  launch_kernel(gridbig_tile_scan_kernel, dim3((unsigned)n_pairs), dim3(kScanThreads), 0, s, stride, gs, scratch);
  launch_kernel(gridbig_scan_kernel, dim3(n_tiles, (unsigned)n_pairs), dim3(kScanThreads), 0, s, n_pts, stride, in_pitch, gs, scratch);
  launch_kernel(gridbig_pass_kernel<true>, chunks, dim3(kBigThreads), 0, s, pts, n_pts, stride, in_pitch, gs, scratch);
  if (gs.rel)
    launch_kernel(gridbig_rel_kernel, dim3((unsigned)((stride + kGridPad + kBigThreads - 1) / kBigThreads), (unsigned)n_pairs), dim3(kBigThreads), 0,
                       s, n_pts, stride, in_pitch, gs);
}

// (see the kernels: "Incremental insert into a map-sized persistent index") — ws must hold index_insert_ws_bytes()
size_t index_insert_ws_bytes(size_t cells, size_t n_add) {
  return 16 + (2 * (cells + 1) + n_add + (cells + 1 + kScanTile - 1) / kScanTile) * sizeof(uint32_t);
}
void launch_index_insert_count(const GridSet& gs, size_t cells, const double* d_add, uint32_t n_add, void* ws, hipStream_t s) {
  uint32_t* flag = static_cast<uint32_t*>(ws);
  uint32_t* shift = flag + 4;
  uint32_t* cell_of = shift + 2 * (cells + 1);
  (void)hipMemsetAsync(ws, 0, 16 + (cells + 1) * sizeof(uint32_t), s);
  launch_kernel(index_insert_count_kernel, dim3((n_add + 255) / 256), dim3(256), 0, s, d_add, n_add, gs.desc, flag, shift, cell_of);
}
void launch_index_insert_merge(const GridSet& gs, size_t cells, uint32_t n_old, const double* d_add, uint32_t n_add, void* ws,
                               GridPoint* sorted2, float* rel2, hipStream_t s) {
  uint32_t* shift = static_cast<uint32_t*>(ws) + 4;
  uint32_t* cursor = shift + (cells + 1);
  uint32_t* cell_of = shift + 2 * (cells + 1);
  // (the tile sums sit in the cursor table's spare last entry region: behind the cells of the cell_of array)
  uint32_t* tile_sum = cell_of + n_add;
  const uint32_t n_tiles = (uint32_t)((cells + 1 + kScanTile - 1) / kScanTile);
  launch_kernel(index_insert_tile_sum_kernel, dim3(n_tiles), dim3(kScanThreads), 0, s, gs.desc, shift, tile_sum);
  launch_kernel(index_insert_tile_scan_kernel, dim3(1), dim3(kScanThreads), 0, s, tile_sum, n_tiles);
  launch_kernel(index_insert_scan_kernel, dim3(n_tiles), dim3(kScanThreads), 0, s, gs.desc, gs.cell_start, tile_sum, shift, cursor);
  if (n_old) launch_kernel(index_insert_move_kernel, dim3((n_old + 255) / 256), dim3(256), 0, s, gs.desc, n_old, gs.stride, gs.sorted, gs.rel, shift, sorted2, rel2);
  launch_kernel(index_insert_scatter_kernel, dim3((n_add + 255) / 256), dim3(256), 0, s, d_add, n_add, n_old, gs.desc, gs.stride, cell_of, cursor, sorted2, rel2);
  launch_kernel(index_insert_table_kernel, dim3((unsigned)((cells + 1 + 255) / 256)), dim3(256), 0, s, gs.desc, gs.cell_start, shift, n_old + n_add, gs.stride, rel2);
}

static void debug_ptr(const char* what, const void* p, size_t need) {
  size_t size = 0;
  hipDeviceptr_t base = nullptr;
  const hipError_t e = hipMemGetAddressRange(&base, &size, const_cast<void*>(p));
  const size_t off = e == hipSuccess ? (size_t)((const char*)p - (const char*)base) : 0;
  fprintf(stderr, "[loamx]     %-10s %p need %zu: %s, allocation %p + %zu of %zu%s\n", what, p, need, hipGetErrorString(e), base, off, size,
          (e != hipSuccess || off + need > size) ? "  <-- TOO SMALL / INVALID" : "");
}

// box_min / box_max: the sets' bounding boxes as the extraction left them (keys; entry [pair * in_pitch][axis] of the arrays
// passed here, i.e. already offset to the scan role and feature kind), valid while *box_bad == 0; nullptr: none
template <bool ORDERED>
static void launch_grid_build(size_t n_pairs, const double* pts, const uint32_t* n_pts, size_t stride, uint32_t in_pitch,
                              double max_dist, const GridSet& gs, GridPoint* scratch, uint32_t flags, hipStream_t s, unsigned long long* bytes,
                              const unsigned long long* box_min = nullptr, const unsigned long long* box_max = nullptr, const uint32_t* box_bad = nullptr,
                              const uint32_t* small_if = nullptr) {
  if (g_debug_sync) {
    fprintf(stderr, "[loamx]   grid build ORDERED=%d n_pairs %zu stride %zu in_pitch %u gs.stride %zu\n", (int)ORDERED, n_pairs, stride, in_pitch, gs.stride);
    debug_ptr("pts", pts, n_pairs * in_pitch * stride * 24);
    debug_ptr("n_pts", n_pts, ((n_pairs - 1) * in_pitch + 1) * 4);
    debug_ptr("desc", gs.desc, n_pairs * sizeof(GridDesc));
    debug_ptr("cell_start", gs.cell_start, n_pairs * (size_t)(kGridCellsCap + 1) * 4);
    debug_ptr("sorted", gs.sorted, n_pairs * gs.stride * sizeof(GridPoint));
    if (gs.rel) debug_ptr("rel", gs.rel, n_pairs * 3 * gs.stride * 4);
    if (scratch) debug_ptr("scratch", scratch, n_pairs * gs.stride * sizeof(GridPoint));
  }
  if (!ORDERED && grid_big(stride, scratch, flags)) {
    launch_grid_build_big(n_pairs, pts, n_pts, stride, in_pitch, max_dist, gs, scratch, s);
    return;
  }
  if (grid_small(stride, flags))
    launch_kernel((grid_build_kernel<ORDERED, true>), dim3((unsigned)n_pairs), dim3(kBuildThreads), 0, s, pts, n_pts, stride,
                       in_pitch, max_dist, gs, scratch, bytes, box_min, box_max, box_bad, small_if);
  else
    launch_kernel((grid_build_kernel<ORDERED, false>), dim3((unsigned)n_pairs), dim3(kBuildThreads), 0, s, pts, n_pts, stride,
                       in_pitch, max_dist, gs, scratch, bytes, box_min, box_max, box_bad, small_if);
}

void launch_grid_build_target(const RegBatch& B, const RegConfig& C, bool plane, hipStream_t s) {
  if (B.n_pairs == 0) return;
  // (boxes: [scan][kind][axis]; the target scan of pair p is scan p * in_pitch of the arrays in B)
  const unsigned long long* bmin = B.box_min ? B.box_min + (plane ? 3 : 0) : nullptr;
  const unsigned long long* bmax = B.box_max ? B.box_max + (plane ? 3 : 0) : nullptr;
  if (plane) launch_grid_build<false>(B.n_pairs, B.tgt_planar, B.n_tgt_planar, B.planar_stride, B.in_pitch, C.r_plane, B.grid_plane, B.sort_scratch, C.flags, s, B.grid_bytes, bmin, bmax, B.box_bad);
  else {
    launch_grid_build<false>(B.n_pairs, B.tgt_edge, B.n_tgt_edge, B.edge_stride, B.in_pitch, C.r_edge, B.grid_edge, B.sort_scratch, C.flags, s, B.grid_bytes, bmin, bmax, B.box_bad,
                             B.small_edge_sets == 1u ? B.n_tgt_edge : nullptr);
    if (B.small_edge_sets == 1u)  // (both edge sets of the pairs whose target edge set is brute-force sized)
      launch_kernel(small_sets_build_kernel, dim3((unsigned)B.n_pairs), dim3(kSmallThreads), 0, s, B.tgt_edge, B.n_tgt_edge, B.src_edge, B.n_src_edge, B.edge_stride,
                    B.in_pitch, C.r_edge, B.grid_edge, B.src_grid_edge, B.grid_bytes, bmin, bmax, B.box_bad);
  }
}
void launch_grid_build_targets(const RegBatch& B, const RegConfig& C, hipStream_t s) {
  launch_grid_build_target(B, C, false, s);
  launch_grid_build_target(B, C, true, s);
}

// source sets: only the cell-sorted (Morton) order is used
// (the scratch copy is shared by the two source sets: build + rank of one complete before the next one's build starts — both
// on ONE stream. It is NOT the target builds' scratch: those run on another stream at the same time, and the multi-workgroup
// build of a target set above kGridSmallCap points keeps its box keys and cursors there)
void launch_grid_build_source(const RegBatch& B, const RegConfig& C, bool plane, hipStream_t s) {
  if (B.n_pairs == 0) return;
  const size_t stride = plane ? B.planar_stride : B.edge_stride;
  const uint32_t* n_src = plane ? B.n_src_planar : B.n_src_edge;
  const GridSet& gs = plane ? B.src_grid_plane : B.src_grid_edge;
  // (boxes: the source scan of an interleaved pair is the scan behind its target scan)
  if (!plane && B.small_edge_sets == 2u) {  // (persistent index with a brute-force sized edge set: the sources in their given order, as in a plain call)
    launch_kernel(small_sets_build_kernel, dim3((unsigned)B.n_pairs), dim3(kSmallThreads), 0, s, static_cast<const double*>(nullptr), static_cast<const uint32_t*>(nullptr),
                  B.src_edge, B.n_src_edge, B.edge_stride, B.in_pitch, C.r_edge, B.grid_edge, B.src_grid_edge, B.grid_bytes, static_cast<const unsigned long long*>(nullptr),
                  static_cast<const unsigned long long*>(nullptr), static_cast<const uint32_t*>(nullptr));
    return;
  }
  const unsigned long long* bmin = (B.box_min && B.in_pitch == 2) ? B.box_min + 6 + (plane ? 3 : 0) : nullptr;
  const unsigned long long* bmax = (B.box_max && B.in_pitch == 2) ? B.box_max + 6 + (plane ? 3 : 0) : nullptr;
  launch_grid_build<true>(B.n_pairs, plane ? B.src_planar : B.src_edge, n_src, stride, B.in_pitch, plane ? C.r_plane : C.r_edge, gs,
                          B.sort_scratch_src, C.flags, s, B.grid_bytes, bmin, bmax, B.box_bad, (!plane && B.small_edge_sets == 1u) ? B.n_tgt_edge : nullptr);
  if (stride && !grid_small(stride, C.flags))
    launch_kernel(grid_rank_kernel, dim3((unsigned)((stride + kRankThreads - 1) / kRankThreads), (unsigned)B.n_pairs), dim3(kRankThreads), 0, s,
                  n_src, stride, B.in_pitch, gs, B.sort_scratch_src, (!plane && B.small_edge_sets == 1u) ? B.n_tgt_edge : nullptr);
}
void launch_grid_build_sources(const RegBatch& B, const RegConfig& C, hipStream_t s) {
  launch_grid_build_source(B, C, false, s);
  launch_grid_build_source(B, C, true, s);
}

void launch_state_init(const RegBatch& B, const RegConfig& C, hipStream_t s) {
  if (B.n_pairs == 0) return;
  launch_kernel(state_init_kernel, dim3(per_pair_grid(B.n_pairs)), dim3(64), 0, s, B, C);
}

#ifndef LOAMX_REST_BLOCKS
#define LOAMX_REST_BLOCKS 32u
#endif
// workgroups (of kRestThreads) per pair for the queue kernels: a few for big batches (queues are short),
// enough to cover a whole set when there are only a few pairs (scan-to-map: one pair, 40 k queries)
// Queue chain in one stage (the FP64 search over all rounds inside associate_knn_rest_kernel) or two (lean search of the
// 5x5x5 block, then the listed leftovers on compacted wavefronts)? Two stages halve the chain's work, but their
// latencies add up — about 0.3 ms for the slowest leftover walks whatever the batch — and only a plane fit kernel that
// runs at least that long hides them. Measured (same box, 64 x 1024 scan pairs): 256 pairs 3.73 (one) vs 3.88 ms (two),
// 512 / 1 024 pairs equal, 2 048 / 4 096 pairs 22.7 / 45.2 vs 22.5 / 44.3 ms; a single pair 0.92 vs 0.98 ms.
constexpr size_t kQueueTwoStageMin = 16u << 20;  // source features in the batch (1 024 pairs of 64 x 1024: 20 M slots)
static bool queue_one_stage(const RegBatch& B, const RegConfig& C, bool plane) {
  if (C.flags & kRegFlagQueueTwoStage) return false;
  if (C.flags & kRegFlagQueueOneStage) return true;
  if (!(C.flags & kRegFlagNoCoopLeft)) return false;  // the cooperative leftover kernel is short at every batch size
  return B.n_pairs * (plane ? B.planar_stride : B.edge_stride) < kQueueTwoStageMin;
}
// wavefronts per pair of associate_knn_coop_kernel (one listed entry per wavefront and turn): a batch lists 15-35 entries per pair,
// a single scan-to-map registration up to half of its queries
static uint32_t coop_blocks(size_t n_pairs) {
  const size_t want = 32768 / (n_pairs ? n_pairs : 1);
  return (uint32_t)(want < 32 ? 32 : (want > 4096 ? 4096 : want));
}
static uint32_t rest_blocks(size_t n_pairs, uint32_t nblk) {
  const uint32_t cover = nblk * (uint32_t)(kAssocThreads / kRestThreads);  // one pass over a full queue
  uint32_t want = (uint32_t)(4096 / (n_pairs ? n_pairs : 1));
  if (want < LOAMX_REST_BLOCKS) want = LOAMX_REST_BLOCKS;
  return want < cover ? want : cover;
}
// what: bit 0 = the edge chains, bit 1 = the plane chains (the first ICF iteration's edge chains can run early, next to the
// planar index builds: launch_associate(..., kAssocEdges) on the stream that built the edge sets, then kAssocPlanes)
void launch_associate(const RegBatch& B, const RegConfig& C, hipStream_t s, hipStream_t aux, hipStream_t aux2, hipEvent_t ev_fork,
                      hipEvent_t ev_mid, hipEvent_t ev_join, hipEvent_t ev_join2, LaunchScope* knn_scope, uint32_t what) {
  if (B.n_pairs == 0) return;
  const uint32_t be = B.assoc_blocks_edge != 0xFFFFFFFFu ? B.assoc_blocks_edge : (uint32_t)((B.edge_stride + kAssocThreads - 1) / kAssocThreads);
  const uint32_t bp = B.assoc_blocks_plane != 0xFFFFFFFFu ? B.assoc_blocks_plane : (uint32_t)((B.planar_stride + kAssocThreads - 1) / kAssocThreads);
  const size_t pair_groups = (B.n_pairs + 7) / 8;  // grid covers 8 XCD lanes x pair_groups x chunks
  // register-resident neighbour lists are instantiated for K <= 5 (the reference's default), K <= 8 and K <= 16
  // Per feature kind, two chains: A = brute force (small target sets) | round-1 k-NN -> fit of the
  // finished queries; B = k-NN of the queued queries (all rounds, then exact) -> their fit. B needs
  // only the round-1 kernel of A, and its short, uneven queues leave most of the GPU idle, so it runs on
  // the auxiliary stream next to A's fit kernel. The edge chains (small sets) run on the auxiliary
  // stream next to the plane round-1 kernel.
#define LOAMX_ASSOC_A1(PL, KMV, nblk, st)                                                                         \
  do {                                                                                                            \
    const dim3 grid_((unsigned)(pair_groups * 8 * (nblk)));                                                       \
    const uint32_t mode_ = (PL) ? B.knn_mode_plane : B.knn_mode_edge; /* a kernel no pair needs is not launched */   \
    if (mode_ != 2u) {                                                                                            \
      LaunchScope* outer_ = g_launch_scope;                                                                       \
      if ((PL) && knn_scope) g_launch_scope = knn_scope;                                                          \
      launch_kernel((associate_knn_kernel<PL, KMV>), grid_, dim3(kAssocThreads), 0, (st), B, C, (nblk));     \
      g_launch_scope = outer_;                                                                                    \
    }                                                                                                             \
    if (mode_ != 1u)                                                                                              \
      launch_kernel((associate_knn_brute_kernel<PL, KMV>), grid_, dim3(kAssocThreads), 0, (st), B, C, (nblk)); \
  } while (0)
#define LOAMX_ASSOC_A2(PL, KMV, nblk, st)                                                                         \
  launch_kernel((associate_fit_kernel<PL, KMV>), dim3((unsigned)(pair_groups * 8 * (nblk))), dim3(kAssocThreads), 0, \
                     (st), B, C, (nblk))
#define LOAMX_ASSOC_B(PL, KMV, nblk, st)                                                                          \
  do {                                                                                                            \
    const uint32_t rblk_ = rest_blocks(B.n_pairs, (nblk));                                                        \
    const uint32_t one_ = queue_one_stage(B, C, (PL)) ? 1u : 0u;                                                  \
    if (one_) {                                                                                                   \
      launch_kernel((associate_knn_rest_kernel<PL, KMV, true>), dim3((unsigned)(pair_groups * 8 * rblk_)),   \
                         dim3(kRestThreads), 0, (st), B, C, rblk_);                                               \
    } else {                                                                                                      \
      launch_kernel((associate_knn_rest_kernel<PL, KMV, false>), dim3((unsigned)(pair_groups * 8 * rblk_)),  \
                         dim3(kRestThreads), 0, (st), B, C, rblk_);                                               \
      if (!queue_leftovers_coop(C, (PL))) {                                                                       \
        launch_kernel((associate_knn_left_kernel<PL, KMV>), dim3((unsigned)(pair_groups * 8 * rblk_)),       \
                           dim3(kRestThreads), 0, (st), B, C, rblk_);                                             \
      } else {                                                                                                    \
        const uint32_t cblk_ = coop_blocks(B.n_pairs);                                                            \
        launch_kernel((associate_knn_coop_kernel<PL, KMV>), dim3((unsigned)(pair_groups * 8 * cblk_)),       \
                           dim3(kRestThreads), 0, (st), B, C, cblk_);                                             \
      }                                                                                                           \
    }                                                                                                             \
    launch_kernel((associate_fit_queued_kernel<PL, KMV>), dim3((unsigned)(pair_groups * 8 * rblk_)),         \
                       dim3(kRestThreads), 0, (st), B, C, rblk_);                                                 \
  } while (0)
#define LOAMX_ASSOC_K(STEP, PL, nblk, st)                \
  do {                                                   \
    if ((PL ? C.k_plane : C.k_edge) <= 5) STEP(PL, 5, nblk, st); \
    else if ((PL ? C.k_plane : C.k_edge) <= 8) STEP(PL, 8, nblk, st); \
    else STEP(PL, 16, nblk, st);                         \
  } while (0)
  const bool edges = (what & kAssocEdges) != 0u && be != 0u, planes = (what & kAssocPlanes) != 0u && bp != 0u;
  // The usual case — every edge set small enough for brute force, every planar set searched through its grid, the
  // reference's five neighbours — runs the first kernels of both kinds as ONE launch each (associate_knn_mixed_kernel,
  // associate_fit_mixed_kernel): no edge chain on a side stream.
  if (edges && planes && B.knn_mode_edge == 2u && B.knn_mode_plane == 1u && C.k_edge <= 5 && C.k_plane <= 5 &&
      !(C.flags & kRegFlagNoMixedAssoc)) {
    const uint32_t edge_blocks = (uint32_t)(pair_groups * 8 * be);
    const dim3 grid((unsigned)(edge_blocks + pair_groups * 8 * bp));
    LaunchScope* outer = g_launch_scope;
    if (knn_scope) g_launch_scope = knn_scope;
    launch_kernel((associate_knn_mixed_kernel<5, 5>), grid, dim3(kAssocThreads), 0, s, B, C, be, bp, edge_blocks);
    g_launch_scope = outer;
    hipStream_t sb = aux2 ? aux2 : aux;
    // (Round 5: the fork / join events as the kernels' own completion events — hipExtLaunchKernelGGL's stop event instead of
    // hipEventRecord — shorten the gap in front of the fit from 6.4 to 5.3 us, but the host's wait at the end of the step
    // then takes 23 us longer: not kept.)
    const bool fork2 = sb != nullptr && hipEventRecord(ev_mid, s) == hipSuccess && hipStreamWaitEvent(sb, ev_mid, 0) == hipSuccess;
    launch_kernel((associate_fit_mixed_kernel<5, 5>), grid, dim3(kAssocThreads), 0, s, B, C, be, bp, edge_blocks);
    LOAMX_ASSOC_B(true, 5, bp, (fork2 ? sb : s));
    if (fork2 && hipEventRecord(aux2 ? ev_join2 : ev_join, sb) == hipSuccess) (void)hipStreamWaitEvent(s, aux2 ? ev_join2 : ev_join, 0);
    return;
  }
  // (edges alone: on the caller's stream, nothing to run them next to)
  const bool use_aux = aux != nullptr && planes;  // the side streams only make sense next to the plane chain
  const bool fork = use_aux && (edges || !aux2) && hipEventRecord(ev_fork, s) == hipSuccess &&
                    hipStreamWaitEvent(aux, ev_fork, 0) == hipSuccess;
  hipStream_t sa = fork ? aux : s;
  if (edges) {  // edge chains, whole
    LOAMX_ASSOC_K(LOAMX_ASSOC_A1, false, be, sa);
    LOAMX_ASSOC_K(LOAMX_ASSOC_A2, false, be, sa);
    // (the brute-force kernel finishes every query itself: with no grid search in the batch nothing is ever queued)
    if (B.knn_mode_edge != 2u) LOAMX_ASSOC_K(LOAMX_ASSOC_B, false, be, sa);
  }
  if (planes) {
    LOAMX_ASSOC_K(LOAMX_ASSOC_A1, true, bp, s);
    // the plane queue chain starts when the plane round-1 kernel is done: on its own stream when there is one
    // (behind the edge chain on aux it started only after the plane fit: the edge kernels' many empty workgroups
    // wait for slots next to the plane kernels)
    hipStream_t sb = aux2 ? aux2 : aux;
    const bool fork2 = use_aux && (aux2 || fork) && hipEventRecord(ev_mid, s) == hipSuccess && hipStreamWaitEvent(sb, ev_mid, 0) == hipSuccess;
    LOAMX_ASSOC_K(LOAMX_ASSOC_A2, true, bp, s);
    if (B.knn_mode_plane != 2u) LOAMX_ASSOC_K(LOAMX_ASSOC_B, true, bp, (fork2 ? sb : s));
    if (fork2 && aux2 && hipEventRecord(ev_join2, aux2) == hipSuccess) (void)hipStreamWaitEvent(s, ev_join2, 0);
  }
  if (fork && hipEventRecord(ev_join, aux) == hipSuccess) (void)hipStreamWaitEvent(s, ev_join, 0);
#undef LOAMX_ASSOC_A1
#undef LOAMX_ASSOC_A2
#undef LOAMX_ASSOC_B
#undef LOAMX_ASSOC_K
}

#ifdef LOAMX_NN_SAME_STATS
// debug aid: how many plane queries keep their neighbour list (count + positions, in order) from one ICF iteration to the next?
__global__ void nn_same_kernel(RegBatch B, uint32_t* prev, unsigned long long* stats, uint32_t it) {
  const size_t pair = blockIdx.y;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = B.n_src_planar[pair * B.in_pitch];
  if (i >= n || i >= B.planar_stride || !B.state[pair].active) return;
  const size_t field = B.n_pairs * B.planar_stride, slot = pair * B.planar_stride + i;
  bool same = it > 0, queued = B.assoc.nn_plane[slot] == 0xFFFFFFFFu;
  for (int f = 0; f < 6; f++) {
    const uint32_t v = B.assoc.nn_plane[f * field + slot];
    if (prev[f * field + slot] != v) same = false;
    prev[f * field + slot] = v;
  }
  atomicAdd(&stats[0], 1ull);
  if (same && !queued) atomicAdd(&stats[1], 1ull);
  if (queued) atomicAdd(&stats[2], 1ull);
}
void debug_nn_same(const RegBatch& B, uint32_t it, hipStream_t s) {
  static uint32_t* prev = nullptr;
  static unsigned long long* stats = nullptr;
  const size_t words = 6 * B.n_pairs * B.planar_stride;
  if (!prev) { (void)hipMalloc(&prev, words * 4); (void)hipMalloc(&stats, 24); }
  (void)hipMemsetAsync(stats, 0, 24, s);
  launch_kernel(nn_same_kernel, dim3((unsigned)((B.planar_stride + 255) / 256), (unsigned)B.n_pairs), dim3(256), 0, s, B, prev, stats, it);
  unsigned long long h[3];
  (void)hipMemcpyAsync(h, stats, 24, hipMemcpyDeviceToHost, s);
  (void)hipStreamSynchronize(s);
  printf("nn_same it=%u: %llu active plane queries, %llu keep their list (%.1f %%), %llu queued (%.1f %%)\n", it, h[0], h[1],
         100.0 * (double)h[1] / (double)(h[0] ? h[0] : 1), h[2], 100.0 * (double)h[2] / (double)(h[0] ? h[0] : 1));
}
#endif

void launch_sweep(const RegBatch& B, hipStream_t s) {
  if (B.n_pairs == 0 || B.blocks_per_pair == 0) return;
  launch_kernel(sweep_kernel, dim3((unsigned)(B.n_pairs * B.blocks_per_pair)), dim3(kSweepThreads), 0, s, B);
}

// the same evaluation for the pairs that are on moments (timed with the LM kernels: it streams next to nothing)
void launch_lm_pair_loop(const RegBatch& B, const RegConfig& C, hipStream_t s) {
  if (B.n_pairs == 0) return;
  launch_kernel(lm_pair_loop_kernel, dim3((unsigned)B.n_pairs), dim3(64), 0, s, B, C);
}

void launch_moments(const RegBatch& B, hipStream_t s) {
  if (B.n_pairs == 0 || B.mom_blocks_per_pair == 0) return;
  launch_kernel(moment_kernel, dim3((unsigned)(B.n_pairs * B.mom_blocks_per_pair * 4)), dim3(64), 0, s, B);  // (tiles added up by lm_pair_loop_kernel)
}

void launch_lm_step(const RegBatch& B, hipStream_t s) {
  if (B.n_pairs == 0) return;
  launch_kernel(lm_step_kernel, dim3(per_pair_grid(B.n_pairs)), dim3(64), 0, s, B);
}

void launch_lm_begin(const RegBatch& B, const RegConfig& C, hipStream_t s) {
  if (B.n_pairs == 0) return;
  launch_kernel(lm_begin_kernel, dim3(per_pair_grid(B.n_pairs)), dim3(64), 0, s, B, C);
}

void launch_outer_update(const RegBatch& B, const RegConfig& C, hipStream_t s) {
  if (B.n_pairs == 0) return;
  launch_kernel(outer_update_kernel, dim3(per_pair_grid(B.n_pairs)), dim3(64), 0, s, B, C);
}

void launch_write_results(const RegBatch& B, loamx_reg_result* d_results, hipStream_t s) {
  if (B.n_pairs == 0) return;
  launch_kernel(write_results_kernel, dim3(per_pair_grid(B.n_pairs)), dim3(64), 0, s, B, d_results);
}

}  // namespace loamx
