// extract_kernels.hip — loam::extractFeatures on gfx950 (reference: loam/include/loam/features-inl.h,
// loam/src/features.cpp). Three kernels:
//
//   curvature_valid_kernel  rows a5+a6: un-normalised curvature (FP64, contraction off) and the
//                           validity mask, one 256-thread workgroup per 1024-point tile of a scan
//                           line. HBM-bound: reads 24 B/point, writes 8 B + 1 B  => 33 B/point.
//   select_kernel           rows a7-a9: the per-sector "sort + greedy walk" restated as repeated
//                           arg-max / arg-min under the sort's total order, one wavefront per scan
//                           line (sectors of a line are sequentially dependent; lines
//                           are independent), curvature + mask staged in LDS, 64-lane butterfly
//                           reductions. Compute/latency-bound.
//   compact_kernel          row a10: prefix over the per-sector counts and gather of the feature
//                           indices / point copies into the reference's output order.
#include "loamx_internal.h"

namespace loamx {

namespace {

#ifndef LOAMX_CURV_TILE
#define LOAMX_CURV_TILE 512  // measured on 2048 scans of 64x1024: 1024 -> 1.75 ms (31 % of 8 TB/s), 512 -> 1.22 ms (46 %), 256 -> 1.38 ms
#endif
constexpr int kTile = LOAMX_CURV_TILE;
#ifndef LOAMX_CURV_THREADS
#define LOAMX_CURV_THREADS 256
#endif
constexpr int kCurvThreads = LOAMX_CURV_THREADS;
constexpr int kHaloMax = kMaxNeighborPoints + 1;
constexpr int kLocalMax = kTile + 2 * kHaloMax;

constexpr int kCodeWords = (kLocalMax + 63) / 64 + 1;  // 64 local columns per word, one more for the window reads

// any flag set among the `len` (<= 64) local columns from `start` on?
__device__ __forceinline__ bool code_window(const unsigned long long* bits, int start, int len) {
  const int w = start >> 6, b = start & 63;
  const unsigned long long lo = bits[w], hi = bits[w + 1];
  const unsigned long long v = (lo >> b) | (b ? hi << (64 - b) : 0ull);
  const unsigned long long m = len >= 64 ? ~0ull : (1ull << len) - 1ull;
  return (v & m) != 0ull;
}

// NP: neighbor_points when it is known at compile time (the curvature sum unrolls), 0 = read it from P.
// The invalidation codes of a tile are kept as four flag words per 64 local columns (wavefront ballots):
// valid_from_codes (extract_math.h) then is three window tests on them instead of a loop over the neighbours.
// T: scalar type of the scan in HBM (double, or float for the FP32-input path: widened on load, which is
// what the reference's Accessor does for PCL points, so every later bit is the same as for FP64 input).
template <int NP, typename T>
__global__ __launch_bounds__(kCurvThreads) void curvature_valid_kernel(const T* __restrict__ xyz, ExtractParams P,
                                                              double* __restrict__ curv_out,
                                                              uint8_t* __restrict__ mask_out) {
  __shared__ double s_xyz[kLocalMax * 3];
  __shared__ double s_r[kLocalMax];
  __shared__ unsigned long long s_bits[4][kCodeWords];  // range / occlusion 1 / occlusion 2 / parallel
  const int tid = threadIdx.x;
  const size_t line = blockIdx.x;  // scan * H + line
  const int W = (int)P.W, np = NP ? NP : (int)P.np;
  const int halo = np + 1;
  const int t0 = (int)blockIdx.y * kTile;
  const int base = t0 - halo;  // column of local index 0
  const int lo = t0 - halo > 0 ? t0 - halo : 0;
  const int hi = t0 + kTile + halo < W ? t0 + kTile + halo : W;
  const T* __restrict__ g = xyz + line * (size_t)W * 3;

  // coalesced copy of the tile (+halo) of ring-ordered points into LDS
  const int nd = (hi - lo) * 3;
  for (int k = tid; k < nd; k += kCurvThreads) s_xyz[(lo - base) * 3 + k] = (double)g[(size_t)lo * 3 + k];
  __syncthreads();
  for (int c = lo + tid; c < hi; c += kCurvThreads) {
    const int li = c - base;
    s_r[li] = point_range(s_xyz[3 * li], s_xyz[3 * li + 1], s_xyz[3 * li + 2]);
  }
  __syncthreads();
  {
    const int clo = t0 - np > 0 ? t0 - np : 0;
    const int chi = t0 + kTile + np < W ? t0 + kTile + np : W;
    for (int l0 = 0; l0 < kCodeWords * 64; l0 += kCurvThreads) {  // whole wavefronts: the ballots below need every lane
      const int li = l0 + tid, c = base + li;
      uint8_t code = kCodeNone;
      if (c >= clo && c < chi && !is_line_end((uint32_t)c, P.W, (uint32_t)np)) code = point_code(s_r[li - 1], s_r[li], s_r[li + 1], P);
      const unsigned long long b1 = __ballot(code == kCodeRange), b2 = __ballot(code == kCodeOcc1);
      const unsigned long long b3 = __ballot(code == kCodeOcc2), b4 = __ballot(code == kCodeParallel);
      const int w = li >> 6;
      if ((tid & 63) == 0 && w < kCodeWords) s_bits[0][w] = b1, s_bits[1][w] = b2, s_bits[2][w] = b3, s_bits[3][w] = b4;
    }
  }
  __syncthreads();
  for (int k = tid; k < kTile; k += kCurvThreads) {
    const int c = t0 + k;
    if (c >= W) break;
    const int li = c - base;
    const bool end = is_line_end((uint32_t)c, P.W, (uint32_t)np);
    const double cv = end ? -1.0 : curvature_at(s_xyz, li, (uint32_t)np);
    // code 1 at j clears j-np..j+np, code 2 j+1..j+np, code 3 j-(np-1)..j, code 4 j (extract_math.h)
    const bool ok = !end && !code_window(s_bits[0], li - np, 2 * np + 1) && !code_window(s_bits[1], li - np, np) &&
                    !code_window(s_bits[2], li, np) && !code_window(s_bits[3], li, 1);
    curv_out[line * (size_t)W + c] = cv;
    mask_out[line * (size_t)W + c] = ok ? 1 : 0;
  }
}

// The same kernel for a compile-time neighbor_points, restructured around the LDS pipe (PMC on the kernel above:
// VALU 58 %, LDS 60 % busy, and float input — a third fewer HBM bytes — ran no faster): the tile is kept as three
// coordinate arrays, every thread owns TWO adjacent columns and reads the 2 NP + 2 values its two sums share as
// 16-byte pairs (NP = 3: 5 ds_read_b128 per coordinate for two points instead of 2 x 7 ds_read_b64), and writes
// both results with one 16-byte + one 2-byte store. Arithmetic and its order are those of curvature_at.
// Tiles one workgroup walks. Round 5: ONE. With two (rounds 3 - 4: the next tile's loads fly in registers during this tile's
// arithmetic) the kernel held 116 - 128 registers = 4 wavefronts per SIMD; without the prefetch it holds 64 = 8, and the other
// workgroups of the CU do the overlapping: 0.92 -> 0.73 - 0.76 ms (4.6 -> 5.7 TB/s).
#ifndef LOAMX_CURV_TILES
#define LOAMX_CURV_TILES 1
#endif
constexpr int kCurvTilesPerGroup = LOAMX_CURV_TILES;

__device__ __forceinline__ void wave_lds_sync() {
  // LDS operations of one wavefront complete in issue order; this only stops the compiler from
  // moving LDS accesses across the point where lanes exchange data through LDS.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Tile geometry of the two-columns-per-thread form, shared by curvature_valid2_kernel and extract_fused_kernel.
template <int NP, int THREADS = kCurvThreads>
struct Curv2 {
  static constexpr int kThreads = THREADS, kTileC = 2 * THREADS;  // two adjacent columns per thread
  static constexpr int np = NP, halo = NP + 1;
  static constexpr int hp = (halo + 1) & ~1;        // local index of the tile's first column: even, >= halo
  static constexpr int A = (NP + 1) & ~1;           // the pairs read start A columns before the thread's first column
  static constexpr int kPairs = (A + NP + 3) / 2;   // ... and cover li0 - A .. li0 + NP + 1
  static constexpr int kLoc = kTileC + 2 * hp;
  static constexpr int kWords = (kLoc + 63) / 64 + 1;
  static constexpr int kLoads = ((kTileC + 2 * halo) * 3 + THREADS - 1) / THREADS;  // elements per thread and tile
  static constexpr size_t kLdsBytes = sizeof(double) * 4 * kLoc + sizeof(unsigned long long) * 4 * kWords;
};

// Walks n_tiles consecutive tiles of one scan line (THREADS threads together: a workgroup with barriers, or — THREADS = 64 —
// one wavefront on its own with wavefront-level ordering only, `tid` then being the lane) and
// hands every thread its two results: sink(c0, cv[2], ok[2]) for the columns c0, c0 + 1 (c0 < W; c0 + 1 may be == W).
// s_p / s_r / s_bits: Curv2<NP>::kLoc doubles x 3, kLoc doubles, kWords words x 4 of LDS.
// AOS (round 5, curvature_valid2_kernel): the tile is kept in LDS as it lies in the scan — [point][xyz], the same 3 * kLoc doubles —
// instead of as three coordinate arrays: element k of the staged span goes to word k (no k / 3, k % 3 per element: 42 of the
// kernel's ~340 vector instructions per thread), and a thread reads the nine points its two sums need as one run of 216 bytes
// (14 ds_read_b128 at a lane stride of 48 bytes: the 16 lanes of a quarter-wavefront cover the 64 banks once) instead of 15.
template <int NP, int THREADS, typename T, typename Sink, bool AOS = false>
__device__ __forceinline__ void curvature_line_tiles(const T* __restrict__ g, const ExtractParams& P, int t_first, int n_tiles,
                                                     double (*s_p)[Curv2<NP, THREADS>::kLoc], double* s_r,
                                                     unsigned long long (*s_bits)[Curv2<NP, THREADS>::kWords], Sink sink) {
  using G = Curv2<NP, THREADS>;
  constexpr int np = NP, halo = G::halo, hp = G::hp, A = G::A, kPairs = G::kPairs, kWords = G::kWords, kLoads = G::kLoads;
  constexpr int kCurvThreads = THREADS, kTile = G::kTileC;  // (shadow the workgroup-wide constants)
  auto barrier = [] {
    if constexpr (THREADS == 64) wave_lds_sync();
    else __syncthreads();
  };
  const int tid = threadIdx.x % THREADS;
  const int W = (int)P.W;
  // the tile (+halo) as registers: element k of the row-major points, coalesced 8-byte loads (16-byte loads of
  // element pairs were measured slower: 1.03 vs 0.95 ms — the pair straddles two coordinate arrays)
  T regs[kLoads];
  auto fetch = [&](int t0) {
    const int lo = t0 - halo > 0 ? t0 - halo : 0;
    const int hi = t0 + kTile + halo < W ? t0 + kTile + halo : W;
    const int nd = (hi - lo) * 3;
#pragma unroll
    for (int q = 0; q < kLoads; q++) {
      const int k = tid + q * kCurvThreads;
      regs[q] = k < nd ? g[(size_t)lo * 3 + k] : (T)0;
    }
  };
  fetch(t_first);
#pragma unroll 1
  for (int tt = 0; tt < n_tiles; tt++) {
    const int t0 = t_first + tt * kTile;
    if (t0 >= W) break;  // uniform
    const int base = t0 - hp;  // column of local index 0
    const int lo = t0 - halo > 0 ? t0 - halo : 0;
    const int hi = t0 + kTile + halo < W ? t0 + kTile + halo : W;
    double* const aos = &s_p[0][0];  // (AOS) the same LDS as [point][xyz]
    {  // element k goes to coordinate k % 3 of point k / 3
      const int nd = (hi - lo) * 3;
#pragma unroll
      for (int q = 0; q < kLoads; q++) {
        const int k = tid + q * kCurvThreads;
        if (k < nd) {
          if constexpr (AOS) {
            aos[3 * (lo - base) + k] = (double)regs[q];
          } else {
            const uint32_t pt = __umulhi((uint32_t)k, 0x55555556u);  // k / 3
            s_p[(uint32_t)k - 3u * pt][lo - base + (int)pt] = (double)regs[q];
          }
        }
      }
    }
    barrier();
    if (tt + 1 < n_tiles && t0 + kTile < W) fetch(t0 + kTile);  // in flight until the next trip
    for (int c = lo + tid; c < hi; c += kCurvThreads) {
      const int li = c - base;
      if constexpr (AOS) s_r[li] = point_range(aos[3 * li], aos[3 * li + 1], aos[3 * li + 2]);
      else s_r[li] = point_range(s_p[0][li], s_p[1][li], s_p[2][li]);
    }
    barrier();
    {
      const int clo = t0 - np > 0 ? t0 - np : 0;
      const int chi = t0 + kTile + np < W ? t0 + kTile + np : W;
      for (int l0 = 0; l0 < kWords * 64; l0 += kCurvThreads) {  // whole wavefronts: the ballots below need every lane
        const int li = l0 + tid, c = base + li;
        uint8_t code = kCodeNone;
        if (c >= clo && c < chi && !is_line_end((uint32_t)c, P.W, (uint32_t)np)) code = point_code(s_r[li - 1], s_r[li], s_r[li + 1], P);
        const unsigned long long b1 = __ballot(code == kCodeRange), b2 = __ballot(code == kCodeOcc1);
        const unsigned long long b3 = __ballot(code == kCodeOcc2), b4 = __ballot(code == kCodeParallel);
        const int w = li >> 6;
        if ((tid & 63) == 0 && w < kWords) s_bits[0][w] = b1, s_bits[1][w] = b2, s_bits[2][w] = b3, s_bits[3][w] = b4;
      }
    }
    barrier();
    const int c0 = t0 + 2 * tid, li0 = hp + 2 * tid;
    if (c0 < W) {
      double d[2][3];
      if constexpr (AOS) {
        static_assert(!AOS || (A % 2 == 0 && A >= NP), "the run of points li0 - A .. li0 + 1 + NP starts 16-byte aligned (li0 is even) and holds every neighbour");
        constexpr int kRun = (3 * (A + NP + 2) + 1) / 2;  // 16-byte pieces covering points li0 - A .. li0 + 1 + NP
        double r[2 * kRun];
#pragma unroll
        for (int q = 0; q < kRun; q++) {
          const double2 v = *reinterpret_cast<const double2*>(&aos[3 * (li0 - A) + 2 * q]);
          r[2 * q] = v.x, r[2 * q + 1] = v.y;
        }
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int col = 0; col < 2; col++) {  // features-inl.h:73-82, the order of curvature_at
            double acc = -(2.0 * np) * r[3 * (A + col) + a];
#pragma unroll
            for (int n = 1; n <= NP; n++) acc = acc + r[3 * (A + col - n) + a] + r[3 * (A + col + n) + a];
            d[col][a] = acc;
          }
      } else {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        double w[2 * kPairs];
#pragma unroll
        for (int q = 0; q < kPairs; q++) {
          const double2 v = *reinterpret_cast<const double2*>(&s_p[a][li0 - A + 2 * q]);
          w[2 * q] = v.x, w[2 * q + 1] = v.y;
        }
#pragma unroll
        for (int col = 0; col < 2; col++) {  // features-inl.h:73-82, the order of curvature_at
          double acc = -(2.0 * np) * w[A + col];
#pragma unroll
          for (int n = 1; n <= NP; n++) acc = acc + w[A + col - n] + w[A + col + n];
          d[col][a] = acc;
        }
      }
      }
      double cv[2];
      bool ok[2];
#pragma unroll
      for (int col = 0; col < 2; col++) {
        const int c = c0 + col, li = li0 + col;
        const bool end = c >= W || is_line_end((uint32_t)c, P.W, (uint32_t)np);
        cv[col] = end ? -1.0 : d[col][0] * d[col][0] + d[col][1] * d[col][1] + d[col][2] * d[col][2];
        // code 1 at j clears j-np..j+np, code 2 j+1..j+np, code 3 j-(np-1)..j, code 4 j (extract_math.h)
        ok[col] = !end && !code_window(s_bits[0], li - np, 2 * np + 1) && !code_window(s_bits[1], li - np, np) &&
                  !code_window(s_bits[2], li, np) && !code_window(s_bits[3], li, 1);
      }
      sink(c0, cv, ok);
    }
    barrier();  // the arrays are rewritten by the next trip
  }
}

#ifndef LOAMX_CURV_WAVES
#define LOAMX_CURV_WAVES 4
#endif
// SPLIT (round 5; between this kernel and select_rows_kernel only): the curvature leaves as two 32-bit arrays — hi words
// (n points), then lo words (n points) in the same buffer — with the validity in the hi word's sign bit (a curvature is a sum
// of squares: its own sign bit is never set; a line end's -1.0 already carries it) and no validity bytes at all. The
// selection orders non-negative doubles by their hi words alone and fetches the lo words of a sector only when two hi words
// it compares are equal: it reads 4 instead of 9 bytes per point.
template <int NP, typename T, bool SPLIT = false>
__global__ __launch_bounds__(kCurvThreads, LOAMX_CURV_WAVES) void curvature_valid2_kernel(const T* __restrict__ xyz, ExtractParams P,
                                                               double* __restrict__ curv_out,
                                                               uint8_t* __restrict__ mask_out) {
  using G = Curv2<NP>;
  __shared__ __align__(16) double s_p[3][G::kLoc];
  __shared__ double s_r[G::kLoc];
  __shared__ unsigned long long s_bits[4][G::kWords];  // range / occlusion 1 / occlusion 2 / parallel
  const size_t line = blockIdx.x;  // scan * H + line
  const int W = (int)P.W;
  const T* __restrict__ g = xyz + line * (size_t)W * 3;
  auto sink = [&](int c0, const double cv[2], const bool ok[2]) {
                                const size_t o = line * (size_t)W + c0;
                                if constexpr (SPLIT) {  // (launched for even W only)
                                  uint32_t* __restrict__ hi_out = reinterpret_cast<uint32_t*>(curv_out);
                                  uint32_t* __restrict__ lo_out = hi_out + (size_t)gridDim.x * (size_t)W;
                                  const uint32_t h0 = (uint32_t)__double2hiint(cv[0]) | (ok[0] ? 0u : 0x80000000u);
                                  const uint32_t h1 = (uint32_t)__double2hiint(cv[1]) | (ok[1] ? 0u : 0x80000000u);
                                  *reinterpret_cast<uint2*>(hi_out + o) = make_uint2(h0, h1);
                                  *reinterpret_cast<uint2*>(lo_out + o) = make_uint2((uint32_t)__double2loint(cv[0]), (uint32_t)__double2loint(cv[1]));
                                  return;
                                }
                                if (c0 + 1 < W && (W & 1) == 0) {  // both columns, aligned
                                  *reinterpret_cast<double2*>(curv_out + o) = make_double2(cv[0], cv[1]);
                                  *reinterpret_cast<uint16_t*>(mask_out + o) = (uint16_t)((ok[0] ? 1u : 0u) | (ok[1] ? 0x100u : 0u));
                                } else {
                                  curv_out[o] = cv[0], mask_out[o] = ok[0] ? 1 : 0;
                                  if (c0 + 1 < W) curv_out[o + 1] = cv[1], mask_out[o + 1] = ok[1] ? 1 : 0;
                                }
                              };
#ifndef LOAMX_CURV_AOS
#define LOAMX_CURV_AOS 1
#endif
  curvature_line_tiles<NP, kCurvThreads, T, decltype(sink), LOAMX_CURV_AOS != 0>(g, P, (int)blockIdx.y * kCurvTilesPerGroup * kTile, kCurvTilesPerGroup, s_p, s_r,
                                                                                  s_bits, sink);
}


// LDS bytes of replay_kernel's wavefront: curvature, mask, and the 16-bit index array that is sorted
__host__ __device__ inline size_t replay_lds_bytes(int W) { return (size_t)W * 8 + (((size_t)W + 7) & ~(size_t)7) + (((size_t)W * 2 + 7) & ~(size_t)7); }

// tie: set when the best remaining candidate is not unique (two valid candidates of equal curvature compete for a
// pick: which one the reference takes is std::sort's business — the caller replays the line, ring_replay)
template <bool EDGE>
__device__ __forceinline__ uint32_t select_pass(const double* s_c, volatile uint8_t* s_v, int lane, int start, int end,
                                                double thr, uint32_t max_feats, uint32_t np, uint32_t line_base,
                                                uint32_t* __restrict__ stage, bool& tie) {
  uint32_t n = 0;
  for (;;) {
    double bc = 0.0;
    int32_t bi = -1;
    bool tied = false;  // the value bc is held by more than one candidate seen so far
    for (int i = start + lane; i < end; i += 64) {
      if (s_v[i]) {
        const double c = s_c[i];
        const bool cand = EDGE ? (c > thr) : (c < thr);
        if (cand) {
          if (bi >= 0 && c == bc) tied = true;
          else if (bi < 0 || (EDGE ? edge_before(c, i, bc, bi) : planar_before(c, i, bc, bi))) bc = c, bi = i, tied = false;
        }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double oc = __shfl_xor(bc, off);
      const int32_t oi = __shfl_xor(bi, off);
      const bool ot = __shfl_xor((int)tied, off) != 0;
      if (oi >= 0) {
        if (bi >= 0 && oc == bc) tied = true, bi = (EDGE ? oi > bi : oi < bi) ? oi : bi;
        else if (bi < 0 || (EDGE ? edge_before(oc, oi, bc, bi) : planar_before(oc, oi, bc, bi))) bc = oc, bi = oi, tied = ot;
      }
    }
    if (bi < 0) break;  // wave-uniform: every lane holds the same winner
    if (tied) tie = true;  // (uniform)
    if (lane == 0) stage[n] = line_base + (uint32_t)bi;
    if ((uint32_t)lane < np) {  // features-inl.h:148-151 / :170-173: idx +- n for n = 0 .. np-1
      s_v[bi + lane] = 0;
      s_v[bi - lane] = 0;
    }
    wave_lds_sync();
    n++;
    if (n > max_feats) break;  // features-inl.h:155 / :177 (max + 1 features can be taken)
  }
  return n;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void select_kernel(const double* __restrict__ curv,
                                                            const uint8_t* __restrict__ mask, size_t n_lines,
                                                            ExtractParams P, ExtractStage st, unsigned long long* line_tot,
                                                            uint32_t* flags) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t line = (size_t)blockIdx.x * WAVES + wave;
  if (line >= n_lines) return;  // whole wavefront leaves; no workgroup barrier is used below
  const int W = (int)P.W;
  const size_t per_wave = (size_t)W * 8 + (((size_t)W + 7) & ~(size_t)7);
  double* s_c = reinterpret_cast<double*>(smem + wave * per_wave);
  volatile uint8_t* s_v = reinterpret_cast<volatile uint8_t*>(s_c + W);
  for (int i = lane; i < W; i += 64) {
    s_c[i] = curv[line * (size_t)W + i];
    s_v[i] = mask[line * (size_t)W + i];
  }
  wave_lds_sync();
  const uint32_t line_base = (uint32_t)(line % P.H) * P.W;
  bool tie = false;  // wave-uniform
  for (uint32_t s = 0; s < P.S; s++) {
    const int start = (int)(s * P.pps);
    const int end = (s == P.S - 1) ? W : start + (int)P.pps;  // features-inl.h:31-35
    const size_t group = line * P.S + s;
    const uint32_t ne = select_pass<true>(s_c, s_v, lane, start, end, P.edge_thr, P.max_edge, P.np, line_base,
                                          st.edge_stage + group * P.cap_edge, tie);
    const uint32_t npl = select_pass<false>(s_c, s_v, lane, start, end, P.planar_thr, P.max_planar, P.np, line_base,
                                            st.planar_stage + group * P.cap_planar, tie);
    if (lane == 0) {
      st.edge_cnt[group] = ne;
      st.planar_cnt[group] = npl;
    }
  }
  if ((tie || (P.flags & kFlagForceReplay)) && line_tot && lane == 0) {  // replay_kernel redoes this line in the reference's own order
    __hip_atomic_store(line_tot + line, 1ull << 62, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    atomicOr(flags, 2u);
  }
}


/* ------------------------------------------------------------------------------------------------
 * select_mis_kernel — the same selection as select_kernel, restated as a lexicographically-first
 * maximal independent set on 64-bit lane masks (extract_math.h) instead of one wave-wide arg-max per
 * pick: per (sector, pass) a few rounds of bit operations + 4 neighbour shuffles, one 64-lane
 * bitonic sort of the picks (gives the reference's output order and the max+1 cap), one more
 * shuffle pair for the suppression. Used when np-1 = R in 1..4, R <= CH, CH + 2R <= 64 and a sector
 * can hold at most 64 picks; select_kernel remains the general fallback (identical results).
 * ---------------------------------------------------------------------------------------------- */
// LDS bytes per wavefront: padded curvature (W + 64 doubles), 64 or 128 pick slots (double + int), padded mask
__host__ __device__ inline size_t select_mis_lds_bytes(int W, int slots) {
  return ((size_t)W + 64) * 8 + (size_t)slots * 12;
}
// picks a sector can yield (they are at least R + 1 points apart): 64 slots, or 128 for the long sectors of
// 2048-column scans (then only with caps <= 64: the lanes hold the first 64 picks of the order)
__host__ __device__ inline int select_mis_slots(const ExtractParams& P, int R) {
  const uint32_t longest = P.W - (P.S - 1) * P.pps;
  return (longest + (uint32_t)R) / (uint32_t)(R + 1) <= 64u ? 64 : 128;
}
// value of the previous / next lane (0 at the wave's ends): one DPP move per dword
// (wave_shr:1 / wave_shl:1 with bound_ctrl zero fill) instead of an LDS-crossbar ds_bpermute
__device__ __forceinline__ uint64_t shfl_prev(uint64_t v, int lane) {
  (void)lane;
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x138, 0xF, 0xF, true);          // wave_shr:1
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x138, 0xF, 0xF, true);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ uint64_t shfl_next(uint64_t v, int lane) {
  (void)lane;
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x130, 0xF, 0xF, true);          // wave_shl:1
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x130, 0xF, 0xF, true);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}

// v of lane (l ^ J) without the LDS pipe: DPP quad / row operations inside a row of 16 lanes, gfx950's
// v_permlane16_swap / v_permlane32_swap across rows (patterns checked on the device by tools/probes/dpp_xor.hip).
template <int J>
__device__ __forceinline__ int xor_lane_b32(int v, int lane) {
  if constexpr (J == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
  else if constexpr (J == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  else if constexpr (J == 4) {
    const int t = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0x5, false);  // row_shl:4: quads 0, 2 take lane + 4
    return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);         // row_shr:4: quads 1, 3 take lane - 4
  } else if constexpr (J == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);  // row_ror:8
  else if constexpr (J == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // r[0] = rows [v0 v0 v2 v2], r[1] = [v1 v1 v3 v3]
    return (lane & 16) ? r[0] : r[1];
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // r[0] = halves [lo lo], r[1] = [hi hi]
    return (lane & 32) ? r[0] : r[1];
  }
}
// inclusive prefix sum over the 64 lanes in six DPP adds: row_shr 1, 2, 4, 8 inside a row, row_bcast15 / row_bcast31
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x) {
  int v = (int)x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);  // (lanes without a source add 0)
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);  // row_bcast15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);  // row_bcast31 -> rows 2, 3
  return (uint32_t)v;
}

template <int J>
__device__ __forceinline__ double xor_lane_f64(double v, int lane) {
  return __hiloint2double(xor_lane_b32<J>(__double2hiint(v), lane), xor_lane_b32<J>(__double2loint(v), lane));
}
__device__ __forceinline__ double xor_lane_f64(double v, int j, int lane) {  // j: a constant after unrolling
  switch (j) {
    case 1: return xor_lane_f64<1>(v, lane);
    case 2: return xor_lane_f64<2>(v, lane);
    case 4: return xor_lane_f64<4>(v, lane);
    case 8: return xor_lane_f64<8>(v, lane);
    case 16: return xor_lane_f64<16>(v, lane);
    default: return xor_lane_f64<32>(v, lane);
  }
}

/* ------------------------------------------------------------------------------------------------
 * The tie path (row a7). When two candidates of EQUAL curvature can decide a pick or the output order, the
 * reference's result depends on where libstdc++'s std::sort leaves them (features-inl.h:38). The selection kernels
 * only DETECT such scan lines (they poison the line's slot in line_tot and raise bit 1 of the flag word);
 * replay_kernel, launched right behind them, then replays those lines literally: every sector's index array is
 * sorted by stl_sort (extract_math.h: libstdc++'s introsort restated; one lane per sector), and one lane walks the
 * sectors in order exactly as extractSectorEdgeFeatures / extractSectorPlanarFeatures do (features-inl.h:137-180),
 * rewriting the line's stage entries and counts, from which the compaction kernel gathers. Slow and rare: noisy
 * scans have no exact ties, and then every workgroup of replay_kernel leaves after one load.
 * ---------------------------------------------------------------------------------------------- */
constexpr unsigned long long kLinePublished = 1ull << 63, kLineTied = 1ull << 62;
constexpr uint32_t kFlagGaveUp = 1u, kFlagTie = 2u;  // bits of the flag word behind line_tot

__device__ void ring_replay(const double* s_c, uint8_t* s_v, uint16_t* s_ord, int lane, int W, const ExtractParams& P,
                            uint32_t line_base, size_t line, const ExtractStage& st) {
  for (int i = lane; i < W; i += 64) s_ord[i] = (uint16_t)i;
  wave_lds_sync();
  for (uint32_t s = (uint32_t)lane; s < P.S; s += 64) {
    const int start = (int)(s * P.pps), end = (s == P.S - 1) ? W : start + (int)P.pps;  // features-inl.h:31-35
    stl_sort(s_ord, start, end, [&](uint16_t a, uint16_t b) { return s_c[a] < s_c[b]; });
  }
  wave_lds_sync();
  if (lane == 0) {
    const int np = (int)P.np;
    for (uint32_t s = 0; s < P.S; s++) {
      const int start = (int)(s * P.pps), end = (s == P.S - 1) ? W : start + (int)P.pps;
      const size_t group = line * P.S + s;
      uint32_t ne = 0, npl = 0;
      uint32_t* __restrict__ se = st.edge_stage + group * P.cap_edge;
      uint32_t* __restrict__ spn = st.planar_stage + group * P.cap_planar;
      for (int k = end; k > start; k--) {  // features-inl.h:143-156
        const int idx = (int)s_ord[k - 1];
        if (s_v[idx] && s_c[idx] > P.edge_thr) {
          if (ne < P.cap_edge) se[ne] = line_base + (uint32_t)idx;
          for (int n = 0; n < np; n++) {
            if (idx + n < W) s_v[idx + n] = 0;
            if (idx - n >= 0) s_v[idx - n] = 0;
          }
          ne++;
        }
        if (ne > P.max_edge) break;
      }
      for (int k = start; k < end; k++) {  // features-inl.h:166-178
        const int idx = (int)s_ord[k];
        if (s_v[idx] && s_c[idx] < P.planar_thr) {
          if (npl < P.cap_planar) spn[npl] = line_base + (uint32_t)idx;
          for (int n = 0; n < np; n++) {
            if (idx + n < W) s_v[idx + n] = 0;
            if (idx - n >= 0) s_v[idx - n] = 0;
          }
          npl++;
        }
        if (npl > P.max_planar) break;
      }
      st.edge_cnt[group] = ne < P.cap_edge ? ne : P.cap_edge;
      st.planar_cnt[group] = npl < P.cap_planar ? npl : P.cap_planar;
    }
  }
}

// one wavefront per workgroup, lines dealt round-robin; a line is replayed iff its line_tot slot carries kLineTied
__global__ __launch_bounds__(64) void replay_kernel(const double* __restrict__ curv, const uint8_t* __restrict__ mask, size_t n_lines,
                                                    ExtractParams P, ExtractStage st, const unsigned long long* __restrict__ line_tot,
                                                    const uint32_t* __restrict__ flags, unsigned long long* __restrict__ events) {
  if ((__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kFlagTie) == 0u) return;  // uniform: no tied line in this launch
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x, W = (int)P.W;
  double* s_c = reinterpret_cast<double*>(smem);
  uint8_t* s_v = reinterpret_cast<uint8_t*>(s_c + W);
  uint16_t* s_ord = reinterpret_cast<uint16_t*>(s_v + (((size_t)W + 7) & ~(size_t)7));
  for (size_t line = blockIdx.x; line < n_lines; line += gridDim.x) {
    if ((__hip_atomic_load(line_tot + line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kLineTied) == 0ull) continue;  // uniform
    wave_lds_sync();
    if (P.flags & kFlagSplitCurv) {  // hi | lo words, validity in the hi word's sign bit (curvature_valid2_kernel<.., SPLIT>)
      const uint32_t* __restrict__ hi = reinterpret_cast<const uint32_t*>(curv);
      const uint32_t* __restrict__ lo = hi + n_lines * (size_t)W;
      for (int i = lane; i < W; i += 64) {
        const uint32_t h = hi[line * (size_t)W + i];
        // (std::sort's permutation depends on every value of the sector, valid or not: a line end is the -1.0 the curvature
        // kernel wrote, every other point's own sign bit was clear)
        s_c[i] = is_line_end((uint32_t)i, P.W, P.np) ? -1.0 : __hiloint2double((int)(h & 0x7FFFFFFFu), (int)lo[line * (size_t)W + i]);
        s_v[i] = (h >> 31) ? 0 : 1;
      }
    } else {
      for (int i = lane; i < W; i += 64) {
        s_c[i] = curv[line * (size_t)W + i];
        s_v[i] = mask[line * (size_t)W + i];
      }
    }
    wave_lds_sync();
    ring_replay(s_c, s_v, s_ord, lane, W, P, (uint32_t)(line % P.H) * P.W, line, st);
    if (lane == 0 && events) atomicAdd(&events[0], 1ull);
  }
}

template <bool EDGE>
__device__ __forceinline__ bool before_or_invalid(double ca, int32_t ia, double cb, int32_t ib) {
  if (ia < 0) return false;  // padding sorts last
  if (ib < 0) return true;
  return EDGE ? edge_before(ca, ia, cb, ib) : planar_before(ca, ia, cb, ib);
}

// tie: set when a comparison between two candidates of EQUAL curvature could decide something in this pass — two
// candidates within R points of each other (the pick depends on which the sort put first), or two picks whose order
// the keys cannot tell (the output order / the cap depends on it). The caller then replays the line (ring_replay).
template <int R, bool EDGE, bool TWO>
__device__ __forceinline__ uint32_t mis_pass(int lane, int CH, int base, int start, int end, uint64_t& V, uint64_t T,
                                             const uint64_t gt[R], const uint64_t eqm[R], bool& tie, const double* s_c, double* m_c,
                                             int32_t* m_i, uint32_t cap, uint32_t line_base, uint32_t* __restrict__ stage,
                                             uint32_t idx_mask, uint32_t ch_magic, uint32_t slots) {
  const uint64_t cm = low_mask(CH);
  const int pbase = lane * (CH + 1);  // s_c is stored with one spare slot per lane chunk (see select_mis_kernel)
  int lo = start - base, hi = end - base;
  lo = lo < 0 ? 0 : lo;
  hi = hi > CH ? CH : hi;
  const uint64_t sm = hi > lo ? (low_mask(hi) & ~low_mask(lo)) : 0ull;
  uint64_t U = V & T & sm;
  if (__ballot(U != 0) == 0) return 0;
  uint64_t Pk = 0;
  {  // equal curvatures among candidates within R points of each other (window bit t <-> t + d, as gt)
    const uint64_t Uw = mis_window<R>(U, shfl_prev(U, lane), shfl_next(U, lane), CH);
    uint64_t adj = 0;
#pragma unroll
    for (int d = 1; d <= R; d++) adj |= Uw & (Uw >> d) & eqm[d - 1];
    if (__ballot(adj != 0) != 0) tie = true;
  }
  do {  // rounds of "local maxima win, their neighbours leave"
    const uint64_t Uw = mis_window<R>(U, shfl_prev(U, lane), shfl_next(U, lane), CH);
    const uint64_t win = (mis_winners<R, EDGE>(Uw, gt) >> R) & cm;
    const uint64_t Ww = mis_window<R>(win, shfl_prev(win, lane), shfl_next(win, lane), CH);
    Pk |= win;
    U &= ~((mis_spread<R>(Ww) >> R) & cm);
  } while (__ballot(U != 0) != 0);
  // compact the picks into LDS slots (at most `slots` = 64 or 128 by construction of the launch condition)
  const uint32_t cnt = (uint32_t)__popcll(Pk);
  const uint32_t incl = wave_incl_scan_u32(cnt);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  uint32_t slot = incl - cnt;
  for (uint64_t bits = Pk; bits; bits &= bits - 1) {
    const int j = __ffsll((unsigned long long)bits) - 1;
    if (slot < slots) {
      m_c[slot] = s_c[pbase + j];
      m_i[slot] = base + j;
    }
    slot++;
  }
  wave_lds_sync();
  // lane l holds pick l and, when a long sector yields more than 64 picks (2048-column scans), pick 64 + l
  const bool two = TWO && total > 64u;  // wave-uniform (TWO: the kernel variant for sectors that can yield more than 64 picks)
  double c = (uint32_t)lane < total ? m_c[lane] : 0.0, c1 = 0.0;
  int32_t i = (uint32_t)lane < total ? m_i[lane] : -1, i1 = -1;
  if (two && 64u + (uint32_t)lane < total) c1 = m_c[64 + lane], i1 = m_i[64 + lane];
  wave_lds_sync();
  // Sort the picks into the walk order of the reference (edge: descending, planar: ascending by
  // (curvature, index)) with a bitonic network over the next power of two >= total lanes.
  // Fast path: curvature and index folded into one double (low index bits replace low mantissa
  // bits; edge keys negated, padding = +inf), so a compare-exchange is v_min_f64 / v_max_f64 on one
  // exchanged value. Exact unless two picks share a truncated curvature (or a curvature is not
  // finite): then the network is re-run on the exact (curvature, index) pairs.
  // More than 64 picks: only the first `cap` <= 64 of the order are ever used, so the first 64 picks are sorted
  // ascending and the rest descending by the same exchanges; the lane-wise minimum of the two is then the set of
  // the 64 first picks as a bitonic sequence, which one more merge (6 layers) puts in order.
  if (total > 1) {
    const uint32_t n2 = two ? 64u : (total <= 2 ? 2u : (1u << (32 - __clz((int)total - 1))));
    auto fold = [&](double cc, int32_t ii) {
      double k = __hiloint2double(__double2hiint(cc), (int)(((uint32_t)__double2loint(cc) & ~idx_mask) | (uint32_t)ii));
      k = EDGE ? -k : k;
      return ii < 0 ? __builtin_huge_val() : k;
    };
    const bool finite = (i < 0 || c <= 1.7976931348623157e308) && (i1 < 0 || c1 <= 1.7976931348623157e308);
    double key = fold(c, i), key1 = two ? fold(c1, i1) : __builtin_huge_val();
    bool undecided = __ballot(!finite) != 0;
    if (!undecided) {
#pragma unroll
      for (int k = 2; k <= 64; k <<= 1) {
        if ((uint32_t)k > n2) break;  // wave-uniform
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
          const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
          const double other = xor_lane_f64(key, j, lane);
          double mn, mx;
          asm("v_min_f64 %0, %1, %2" : "=v"(mn) : "v"(key), "v"(other));
          asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(key), "v"(other));
          key = keep_min ? mn : mx;
          if (two) {  // the second half, descending
            const double o1 = xor_lane_f64(key1, j, lane);
            asm("v_min_f64 %0, %1, %2" : "=v"(mn) : "v"(key1), "v"(o1));
            asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(key1), "v"(o1));
            key1 = keep_min ? mx : mn;
          }
        }
      }
      if (two) {
        double lo, hi;
        asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(key), "v"(key1));
        asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(key), "v"(key1));
        key = lo;
#pragma unroll
        for (int j = 32; j > 0; j >>= 1) {  // bitonic -> ascending
          const double other = xor_lane_f64(key, j, lane);
          double mn, mx;
          asm("v_min_f64 %0, %1, %2" : "=v"(mn) : "v"(key), "v"(other));
          asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(key), "v"(other));
          key = (lane & j) == 0 ? mn : mx;
        }
        // the best of the set-aside picks must be decided against the last kept one too
#pragma unroll
        for (int j = 32; j > 0; j >>= 1) {
          const double other = xor_lane_f64(hi, j, lane);
          asm("v_min_f64 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(other));
        }
        const double last = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(key), 63), __builtin_amdgcn_readlane(__double2loint(key), 63));
        if (__double2hiint(last) == __double2hiint(hi) && (((uint32_t)__double2loint(last) ^ (uint32_t)__double2loint(hi)) & ~idx_mask) == 0u)
          undecided = true;  // (uniform: hi is the same on every lane)
      }
      // neighbours in the sorted order with the same truncated curvature: undecided by the keys
      const double nxt = __longlong_as_double((long long)shfl_next((uint64_t)__double_as_longlong(key), lane));  // lane + 1 (DPP)
      const bool both = (uint32_t)lane + 1 < (two ? 64u : total);
      const bool same = both && __double2hiint(key) == __double2hiint(nxt) &&
                        (((uint32_t)__double2loint(key) ^ (uint32_t)__double2loint(nxt)) & ~idx_mask) == 0u;
      if (__ballot(same) != 0) undecided = true;
    }
    if (undecided) {
      // picks whose truncated curvatures collide (equal curvatures, practically) or that are not finite: the order
      // among them is the sort's — the line is replayed in the reference's own order (the values below are then unused)
      tie = true;
    } else {
      const uint32_t have = two ? 64u : total;
      i = (uint32_t)lane < have ? (int32_t)((uint32_t)__double2loint(key) & idx_mask) : -1;
      c = (uint32_t)lane < have ? s_c[i < 0 ? 0 : i + (ch_magic ? (int32_t)__umulhi((uint32_t)i, ch_magic) : i)] : 0.0;
    }
  }
  const uint32_t kept = total < cap ? total : cap;  // features-inl.h:155/:177: at most max+1 picks
  if ((uint32_t)lane < kept) stage[lane] = line_base + (uint32_t)i;
  // which of my picks survive the cap: those not after the last kept one
  uint64_t K = Pk;
  if (kept < total) {
    // (kept is wave-uniform: scalar lane reads)
    const double tc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(c), (int)kept - 1),
                                       __builtin_amdgcn_readlane(__double2loint(c), (int)kept - 1));
    const int32_t ti = __builtin_amdgcn_readlane(i, (int)kept - 1);
    K = 0;
    for (uint64_t bits = Pk; bits; bits &= bits - 1) {
      const int j = __ffsll((unsigned long long)bits) - 1;
      const double cj = s_c[pbase + j];
      if (!(EDGE ? edge_before(tc, ti, cj, base + j) : planar_before(tc, ti, cj, base + j))) K |= 1ull << j;
    }
  }
  // suppression of +-(np-1) around every kept pick (features-inl.h:148-151 / :170-173)
  const uint64_t Kw = mis_window<R>(K, shfl_prev(K, lane), shfl_next(K, lane), CH);
  V &= ~((mis_spread<R>(Kw) >> R) & cm);
  return kept;
}

// Fused compaction (fz.line_tot != nullptr, number_sectors <= 64): the wavefront of a scan line publishes the
// line's edge / planar totals, sums the totals of the lines before it in the same scan (a chained scan over at
// most scan_lines - 1 published values, one per lane), and writes its picks — indices and point copies — straight
// to their final places in the reference's output order. The gather of the picked points (27 % of a scan, but
// nearly every 128-byte line of it: HBM-bound) then overlaps with the selection of other lines (VALU-bound)
// instead of running as a kernel of its own. Waiting is safe because workgroups start in index order and a line
// only waits for lines of lower index; the wait is bounded all the same (kLookbackSpins), and a wavefront that
// gives up raises fz.error instead of hanging.
constexpr uint32_t kLookbackSpins = 1u << 20;

template <typename T>
__device__ __forceinline__ void fused_copy(const T* __restrict__ scan_xyz, uint32_t idx, double* __restrict__ dst) {
  dst[0] = (double)scan_xyz[3 * (size_t)idx], dst[1] = (double)scan_xyz[3 * (size_t)idx + 1], dst[2] = (double)scan_xyz[3 * (size_t)idx + 2];
}

// Selection + fused compaction of ONE scan line by one wavefront, from the line's curvature / mask in the padded LDS
// arrays s_c / s_v (select_mis_kernel stages them from global memory, extract_fused_kernel computes them in place).
// Returns true iff the line was marked as tied (replay_kernel redoes it and needs its curvature / mask in global memory).
template <int R, bool TWO>
__device__ __forceinline__ bool select_line(int lane, size_t line, int W, int CH, uint32_t ch_magic, double* s_c, double* m_c, int32_t* m_i,
                                            uint64_t V, const ExtractParams& P, const ExtractStage& st, const ExtractFused& fz) {
  // V: validity of the lane's chunk, bit j = point lane * CH + j (bits of points past the line's end are 0)
  constexpr int slots = TWO ? 128 : 64;
  const int base = lane * CH;
  const int pbase = lane * (CH + 1);
  uint64_t ET = 0, PT = 0, gt[R], eqm[R];
  for (int j = 0; j < CH; j++) {
    const int i = base + j;
    if (i < W) {
      const double c = s_c[pbase + j];
      ET |= (uint64_t)(c > P.edge_thr) << j;
      PT |= (uint64_t)(c < P.planar_thr) << j;
    }
  }
  // window values base-R .. base+CH+R-1 (+R more for the comparisons), read once: the halo on either
  // side belongs to the neighbouring lanes' chunks (R <= CH)
#pragma unroll
  for (int d = 1; d <= R; d++) gt[d - 1] = 0, eqm[d - 1] = 0;
  {
    double win[2 * R + 1];  // sliding window of R + 1 consecutive values
    // value at line index i (any lane's chunk within one chunk of mine)
    auto at = [&](int i) -> double {
      const int owner = i < base ? lane - 1 : (i >= base + CH ? (i >= base + 2 * CH ? lane + 2 : lane + 1) : lane);
      return s_c[i + owner];
    };
#pragma unroll
    for (int u = 0; u < R; u++) {
      const int i = base - R + u;
      win[u] = (i >= 0 && i < W) ? at(i) : 0.0;
    }
    for (int t = 0; t < CH + 2 * R; t++) {
      const int i = base - R + t;
      const int in = i + R;  // newest value entering the window
      win[R] = (in >= 0 && in < W) ? at(in) : 0.0;
#pragma unroll
      for (int d = 1; d <= R; d++)
        if (i >= 0 && i + d < W) gt[d - 1] |= (uint64_t)(win[0] > win[d]) << t, eqm[d - 1] |= (uint64_t)(win[0] == win[d]) << t;
#pragma unroll
      for (int u = 0; u < R; u++) win[u] = win[u + 1];
    }
  }
  const uint32_t line_base = (uint32_t)(line % P.H) * P.W;
  const uint32_t idx_mask = W <= 2 ? 1u : (0xFFFFFFFFu >> __clz(W - 1));  // index-in-line bits of the sort keys
  uint32_t my_ne = 0, my_np = 0;  // lane s: picks of sector s
  bool tie = false;  // wave-uniform
  for (uint32_t s = 0; s < P.S; s++) {
    const int start = (int)(s * P.pps);
    const int end = (s == P.S - 1) ? W : start + (int)P.pps;  // features-inl.h:31-35
    const size_t group = line * P.S + s;
    const uint32_t ne = mis_pass<R, true, TWO>(lane, CH, base, start, end, V, ET, gt, eqm, tie, s_c, m_c, m_i, P.cap_edge, line_base,
                                          st.edge_stage + group * P.cap_edge, idx_mask, ch_magic, (uint32_t)slots);
    const uint32_t npl = mis_pass<R, false, TWO>(lane, CH, base, start, end, V, PT, gt, eqm, tie, s_c, m_c, m_i, P.cap_planar,
                                            line_base, st.planar_stage + group * P.cap_planar, idx_mask, ch_magic, (uint32_t)slots);
    if (lane == 0) {
      st.edge_cnt[group] = ne;
      st.planar_cnt[group] = npl;
    }
    if ((uint32_t)lane == s) my_ne = ne, my_np = npl;  // (fused path: number_sectors <= 64)
  }
  if (tie || (P.flags & kFlagForceReplay)) {  // uniform: a tie could decide something on this line
    // replay_kernel redoes the line in the reference's own order; the lines behind it in the scan see the poisoned slot,
    // stop waiting and leave their picks in the stage arrays, from which the fallback compaction gathers the scan
    if (lane == 0 && fz.line_tot) {
      __hip_atomic_store(fz.line_tot + line, kLinePublished | kLineTied, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicOr(fz.error, kFlagTie | kFlagGaveUp);
    }
    return true;
  }
  if (!fz.fuse) return false;  // uniform
  // ---- fused compaction -----------------------------------------------------------------------------
  const uint32_t li = (uint32_t)(line % P.H);
  const size_t scan = line / P.H;
  // inclusive scans over the sectors (lane s = sector s)
  const uint32_t e_incl = wave_incl_scan_u32(my_ne), p_incl = wave_incl_scan_u32(my_np);
  const uint32_t E_l = (uint32_t)__builtin_amdgcn_readlane((int)e_incl, 63), P_l = (uint32_t)__builtin_amdgcn_readlane((int)p_incl, 63);
  if (lane == 0)
    __hip_atomic_store(fz.line_tot + line, kLinePublished | ((unsigned long long)E_l << 32) | (unsigned long long)P_l,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t* __restrict__ oe = fz.edge_idx ? fz.edge_idx + scan * fz.edge_stride : nullptr;  // (optional, as the point copies)
  uint32_t* __restrict__ op = fz.planar_idx ? fz.planar_idx + scan * fz.planar_stride : nullptr;
  double* __restrict__ xe = fz.edge_xyz ? fz.edge_xyz + scan * fz.edge_stride * 3 : nullptr;
  double* __restrict__ xp = fz.planar_xyz ? fz.planar_xyz + scan * fz.planar_stride * 3 : nullptr;
  const size_t scan_off = scan * (size_t)P.H * P.W * 3;
  const bool want_xyz = xe != nullptr || xp != nullptr;  // (both or none in practice)
  uint32_t base_e = 0, base_p = 0;
  // Sectors in batches: all index loads of a batch, then all point loads, then the stores — the chain stage
  // index -> point -> store is three dependent memory round trips, so a batch keeps 2 * kFuseBatch of them in
  // flight per lane. The wait for the preceding lines sits between the loads and the stores of the first batch.
  constexpr uint32_t kFuseBatch = 3;
  for (uint32_t s0 = 0; s0 < P.S; s0 += kFuseBatch) {
    uint32_t idx[2 * kFuseBatch], off[2 * kFuseBatch];
    bool on[2 * kFuseBatch];
#pragma unroll
    for (uint32_t b = 0; b < kFuseBatch; b++) {
      const uint32_t s = s0 + b;
      const bool in = s < P.S;
      const int sl = in ? (int)s : 0;
      const size_t group = line * P.S + (size_t)sl;
      // (sl is wave-uniform: scalar lane reads)
      const uint32_t ce = (uint32_t)__builtin_amdgcn_readlane((int)my_ne, sl), cp = (uint32_t)__builtin_amdgcn_readlane((int)my_np, sl);
      off[2 * b] = (uint32_t)__builtin_amdgcn_readlane((int)e_incl, sl) - ce + (uint32_t)lane;
      off[2 * b + 1] = (uint32_t)__builtin_amdgcn_readlane((int)p_incl, sl) - cp + (uint32_t)lane;
      on[2 * b] = in && (uint32_t)lane < ce, on[2 * b + 1] = in && (uint32_t)lane < cp;
      // (a lane reads what it wrote itself in mis_pass)
      idx[2 * b] = on[2 * b] ? st.edge_stage[group * P.cap_edge + lane] : 0u;
      idx[2 * b + 1] = on[2 * b + 1] ? st.planar_stage[group * P.cap_planar + lane] : 0u;
    }
    double v[2 * kFuseBatch][3];
    if (want_xyz) {
#pragma unroll
      for (uint32_t q = 0; q < 2 * kFuseBatch; q++) {
        if (fz.f32) {
          const float* __restrict__ src = static_cast<const float*>(fz.xyz) + scan_off + 3 * (size_t)idx[q];
#pragma unroll
          for (int k = 0; k < 3; k++) v[q][k] = on[q] ? (double)src[k] : 0.0;
        } else {
          const double* __restrict__ src = static_cast<const double*>(fz.xyz) + scan_off + 3 * (size_t)idx[q];
#pragma unroll
          for (int k = 0; k < 3; k++) v[q][k] = on[q] ? src[k] : 0.0;
        }
      }
    }
    if (s0 == 0) {  // uniform: totals of the lines before this one in the scan
      bool gave_up = false;
      for (uint32_t c0 = 0; c0 < li; c0 += 64) {
        const uint32_t j = c0 + (uint32_t)lane;
        unsigned long long t = 1ull << 63;  // no predecessor on this lane: nothing to wait for, contributes 0
        if (j < li) {
          const unsigned long long* src = fz.line_tot + (line - li + j);
          t = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (uint32_t spins = 0; !(t >> 63) && spins < kLookbackSpins; spins++) {  // (a tied line publishes too: bit 62)
            __builtin_amdgcn_s_sleep(4);
            t = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        gave_up = gave_up || !(t >> 63) || (t & kLineTied) != 0ull;
        const uint32_t se = (uint32_t)(t >> 32) & 0x3FFFFFFFu, sp = (uint32_t)t;
        base_e += (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(se), 63);
        base_p += (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(sp), 63);
      }
      if (__ballot(gave_up) != 0 || ((P.flags & kFlagForceGiveUp) && li > 0)) {  // uniform
        // this line's features stay in the stage arrays (complete, as always): compact_fallback_kernel, launched
        // right behind this kernel, sees the flag and gathers the whole batch from them
        if (lane == 0) atomicOr(fz.error, kFlagGaveUp);
        return false;
      }
    }
#pragma unroll
    for (uint32_t q = 0; q < 2 * kFuseBatch; q++) {
      if (on[q]) {
        const bool edge = (q & 1u) == 0u;
        const uint32_t o = (edge ? base_e : base_p) + off[q];
        if (oe && op) (edge ? oe : op)[o] = idx[q];
        double* __restrict__ x = edge ? xe : xp;
        if (x) {
#pragma unroll
          for (int k = 0; k < 3; k++) x[3 * (size_t)o + k] = v[q][k];
        }
      }
    }
  }
  if (li == P.H - 1 && lane == 0) {
    fz.n_edge[scan] = base_e + E_l, fz.n_planar[scan] = base_p + P_l;
    if (fz.events) atomicAdd(&fz.events[2], (unsigned long long)(base_e + E_l + base_p + P_l));  // (roofline bytes of the fused kernel)
  }
  return false;
}

// WT: the line length as a compile-time constant (1024, 2048) or 0 = P.W. Everything below is inlined, so a constant W
// and CH turn the per-point loops of select_line / mis_pass into straight code: bit masks built with immediate
// operands instead of 64-bit variable shifts, chunk indices without the magic multiply.
template <int R, int WAVES, bool TWO, int WT = 0>
__global__ __launch_bounds__(WAVES * 64) void select_mis_kernel(const double* __restrict__ curv,
                                                                const uint8_t* __restrict__ mask, size_t n_lines,
                                                                ExtractParams P, ExtractStage st, ExtractFused fz) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t line = (size_t)blockIdx.x * WAVES + wave;
  if (line >= n_lines) return;  // whole wavefront leaves; no workgroup barrier below
  const int W = WT ? WT : (int)P.W, CH = (W + 63) / 64;
  // LDS layout: every lane's chunk of CH points is followed by one spare double (4 spare bytes in
  // the mask): with the chunk stride CH (16 for W = 1024) the lanes of a wavefront would all read the
  // same LDS bank when each walks its own chunk; CH + 1 spreads them over all banks.
  constexpr int slots = TWO ? 128 : 64;
  const size_t per_wave = select_mis_lds_bytes(W, slots);
  double* s_c = reinterpret_cast<double*>(smem + wave * per_wave);
  double* m_c = s_c + W + 64;
  int32_t* m_i = reinterpret_cast<int32_t*>(m_c + slots);
  // i / CH == umulhi(i, ch_magic) for i < 2^16 when CH >= 2; ch_magic == 0 stands for CH == 1 (i / CH == i)
  const uint32_t ch_magic = CH > 1 ? 0xFFFFFFFFu / (uint32_t)CH + 1u : 0u;
  for (int i = lane; i < W; i += 64) {
    const int owner = ch_magic ? (int)__umulhi((uint32_t)i, ch_magic) : i;
    s_c[i + owner] = curv[line * (size_t)W + i];
  }
  // the validity bytes of the lane's own chunk straight into its bit mask (the selection works on register bit masks:
  // the bytes need no LDS — 1.3 KB per wavefront less, which is one more wavefront per SIMD)
  uint64_t V = 0;
  {
    const uint8_t* __restrict__ mrow = mask + line * (size_t)W;
    const int base = lane * CH;
    if ((CH & 3) == 0 && (W & 3) == 0) {
      for (int j = 0; j < CH; j += 4) {
        if (base + j < W) {  // (W and CH are multiples of four: the word is inside the line)
          const uint32_t w4 = *reinterpret_cast<const uint32_t*>(mrow + base + j);
          V |= (uint64_t)(((w4 & 0xFFu) != 0u) | (((w4 & 0xFF00u) != 0u) << 1) | (((w4 & 0xFF0000u) != 0u) << 2) |
                          (((w4 & 0xFF000000u) != 0u) << 3)) << j;
        }
      }
    } else {
      for (int j = 0; j < CH; j++)
        if (base + j < W) V |= (uint64_t)(mrow[base + j] != 0) << j;
    }
  }
  wave_lds_sync();
  (void)select_line<R, TWO>(lane, line, W, CH, ch_magic, s_c, m_c, m_i, V, P, st, fz);
}

#include "select_rows.h"

// The fused form of select_rows_kernel never writes curvature or validity to HBM; replay_kernel (row a7: std::sort's order on
// equal curvatures) needs both for the scan lines that were marked as tied. This kernel, launched behind the fused one,
// recomputes exactly those lines into the workspace (curvature_line_tiles: the arithmetic of curvature_valid2_kernel); every
// workgroup leaves after one load when no line is tied.
template <int NP, typename T>
__global__ __launch_bounds__(64) void curvature_tied_kernel(const T* __restrict__ xyz, size_t n_lines, ExtractParams P,
                                                            const unsigned long long* __restrict__ line_tot, const uint32_t* __restrict__ flags,
                                                            double* __restrict__ curv_ws, uint8_t* __restrict__ mask_ws) {
  if ((__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kFlagTie) == 0u) return;  // uniform
  using G = Curv2<NP, 64>;
  __shared__ __align__(16) double s_p[3][G::kLoc];
  __shared__ double s_r[G::kLoc];
  __shared__ unsigned long long s_bits[4][G::kWords];
  const int W = (int)P.W;
  for (size_t line = blockIdx.x; line < n_lines; line += gridDim.x) {
    if ((__hip_atomic_load(line_tot + line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kLineTied) == 0ull) continue;  // uniform
    wave_lds_sync();
    curvature_line_tiles<NP, 64, T>(xyz + line * (size_t)W * 3, P, 0, (W + G::kTileC - 1) / G::kTileC, s_p, s_r, s_bits,
                                    [&](int c0, const double cv[2], const bool ok[2]) {
#pragma unroll
                                      for (int col = 0; col < 2; col++)
                                        if (c0 + col < W) curv_ws[line * (size_t)W + c0 + col] = cv[col], mask_ws[line * (size_t)W + c0 + col] = ok[col] ? 1 : 0;
                                    });
  }
}

/* ------------------------------------------------------------------------------------------------
 * extract_fused_kernel — rows a5-a10 in ONE pass over the scan (round 2): curvature + validity, selection and
 * compaction without the 9 B/point of curvature / mask ever reaching HBM.
 * One wavefront (a workgroup of its own: no barriers) owns a scan line. Phase 1: the tile loop of
 * curvature_valid2_kernel in its one-wavefront form (128-column tiles: coalesced loads -> coordinate planes in LDS ->
 * ranges -> invalidation flag words -> two columns per lane), with the results stored into the padded selection
 * arrays in LDS instead of global memory. Phase 2: select_line (bitmask MIS, keyed bitonic sort, chained scan over
 * the lines of the scan, indices + point copies to their final places). (First built with four lines per workgroup
 * and the 512-column tiles behind workgroup barriers: 60 KB of LDS, two workgroups per CU, 3.29 ms vs 2.80 ms for
 * the separate kernels.) A line on which a curvature tie
 * can decide something writes its curvature / mask to the workspace for replay_kernel.
 * HBM per scan: 24 B/point read once + the picked points again (mostly L2 / MALL hits: the workgroup read them
 * microseconds ago) + (4 + 24) B per feature written; before: 24 + 9 written + 9 read + the gather's re-read.
 * ---------------------------------------------------------------------------------------------- */
template <int NP, int R, typename T>
__global__ __launch_bounds__(64) void extract_fused_kernel(const T* __restrict__ xyz, size_t n_lines, ExtractParams P, ExtractStage st,
                                                           ExtractFused fz, double* __restrict__ curv_ws, uint8_t* __restrict__ mask_ws) {
  using G = Curv2<NP, 64>;  // tiles of 128 columns, two per lane
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const size_t line = blockIdx.x;  // one wavefront per workgroup: nothing is shared, nobody waits at a barrier
  const int W = (int)P.W, CH = (W + 63) / 64;
  const uint32_t ch_magic = CH > 1 ? 0xFFFFFFFFu / (uint32_t)CH + 1u : 0u;
  double(*s_p)[G::kLoc] = reinterpret_cast<double(*)[G::kLoc]>(smem);
  double* s_r = reinterpret_cast<double*>(smem) + 3 * G::kLoc;
  unsigned long long(*s_bits)[G::kWords] = reinterpret_cast<unsigned long long(*)[G::kWords]>(s_r + G::kLoc);
  double* s_c = reinterpret_cast<double*>(smem + ((G::kLdsBytes + 15) & ~(size_t)15));
  double* m_c = s_c + W + 64;
  int32_t* m_i = reinterpret_cast<int32_t*>(m_c + 64);
  uint8_t* s_v = reinterpret_cast<uint8_t*>(m_i + 64);
  // ---- phase 1: curvature + validity of the line into the selection arrays
  curvature_line_tiles<NP, 64, T>(xyz + line * (size_t)W * 3, P, 0, (W + G::kTileC - 1) / G::kTileC, s_p, s_r, s_bits,
                                  [&](int c0, const double cv[2], const bool ok[2]) {
#pragma unroll
                                    for (int col = 0; col < 2; col++) {
                                      const int i = c0 + col;
                                      if (i < W) {
                                        const int owner = ch_magic ? (int)__umulhi((uint32_t)i, ch_magic) : i;
                                        s_c[i + owner] = cv[col];
                                        s_v[i + 4 * owner] = ok[col] ? 1 : 0;
                                      }
                                    }
                                  });
  wave_lds_sync();
  // ---- phase 2: selection + compaction
  uint64_t V = 0;
  for (int j = 0; j < CH; j++)
    if (lane * CH + j < W) V |= (uint64_t)(s_v[lane * (CH + 4) + j] != 0) << j;
  if (select_line<R, false>(lane, line, W, CH, ch_magic, s_c, m_c, m_i, V, P, st, fz)) {
    // tied: replay_kernel reads the line's curvature and (unselected) mask from the workspace. The mask bytes in LDS
    // are still the input's: the selection works on register bit masks.
    for (int i = lane; i < W; i += 64) {
      const int owner = ch_magic ? (int)__umulhi((uint32_t)i, ch_magic) : i;
      curv_ws[line * (size_t)W + i] = s_c[i + owner];
      mask_ws[line * (size_t)W + i] = s_v[i + 4 * owner];
    }
  }
}

// block-wide exclusive scan of one value per thread (256 threads); returns the exclusive prefix and
// writes the block total to *total
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_scan, uint32_t* total) {
  const int tid = threadIdx.x;
  s_scan[tid] = v;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const uint32_t add = tid >= off ? s_scan[tid - off] : 0;
    __syncthreads();
    s_scan[tid] += add;
    __syncthreads();
  }
  const uint32_t incl = s_scan[tid];
  *total = s_scan[255];
  __syncthreads();
  return incl - v;
}

// kCompactSplit workgroups share a scan: every one of them runs the (cheap) prefix over the sector counts and
// copies an interleaved share of the items; a thread keeps kCompactBatch items in flight (stage index ->
// gathered point -> store is a chain of dependent HBM accesses, so the copy is bound by latency, not bytes).
constexpr uint32_t kCompactSplit = 4, kCompactBatch = 4;

template <typename T>
__device__ __forceinline__ void compact_one(const T* __restrict__ scan_xyz, uint32_t groups, uint32_t cap,
                                            const uint32_t* __restrict__ stage, const uint32_t* __restrict__ cnt,
                                            uint32_t* __restrict__ out_idx, double* __restrict__ out_xyz,
                                            uint32_t* __restrict__ out_n, uint32_t split, uint32_t* s_scan, uint32_t* s_off,
                                            uint32_t* s_cnt) {
  const int tid = threadIdx.x;
  // it / cap by multiplication: exact while it * cap < 2^32 (it < 256 * cap)
  const uint32_t magic = (cap > 1 && cap < 4096) ? 0xFFFFFFFFu / cap + 1u : 0u;
  uint32_t carry = 0;
  for (uint32_t g0 = 0; g0 < groups; g0 += 256) {
    const uint32_t g = g0 + tid;
    const uint32_t c = g < groups ? cnt[g] : 0;
    uint32_t total;
    const uint32_t excl = block_exclusive_scan(c, s_scan, &total);
    s_off[tid] = carry + excl;
    s_cnt[tid] = c;
    __syncthreads();
    const uint32_t chunk_groups = groups - g0 < 256 ? groups - g0 : 256;
    const uint32_t items = chunk_groups * cap;
    constexpr uint32_t step = 256 * kCompactSplit;
    for (uint32_t it0 = split * 256 + tid; it0 < items; it0 += step * kCompactBatch) {
      uint32_t idx[kCompactBatch], o[kCompactBatch];
      bool ok[kCompactBatch];
#pragma unroll
      for (uint32_t b = 0; b < kCompactBatch; b++) {
        const uint32_t it = it0 + b * step;
        ok[b] = it < items;
        const uint32_t gl = ok[b] ? (magic ? __umulhi(it, magic) : it / cap) : 0u, j = it - gl * cap;
        ok[b] = ok[b] && j < s_cnt[gl];
        o[b] = s_off[gl] + j;
        idx[b] = ok[b] ? stage[(size_t)(g0 + gl) * cap + j] : 0u;
      }
      if (out_xyz) {
        double v[kCompactBatch][3];
#pragma unroll
        for (uint32_t b = 0; b < kCompactBatch; b++) {
#pragma unroll
          for (int k = 0; k < 3; k++) v[b][k] = ok[b] ? (double)scan_xyz[3 * (size_t)idx[b] + k] : 0.0;
        }
#pragma unroll
        for (uint32_t b = 0; b < kCompactBatch; b++) {
          if (ok[b]) {
#pragma unroll
            for (int k = 0; k < 3; k++) out_xyz[3 * (size_t)o[b] + k] = v[b][k];
          }
        }
      }
#pragma unroll
      for (uint32_t b = 0; b < kCompactBatch; b++)
        if (ok[b] && out_idx) out_idx[o[b]] = idx[b];
    }
    carry += total;
    __syncthreads();
  }
  if (tid == 0 && split == 0) *out_n = carry;
}

template <typename T>
__global__ __launch_bounds__(256) void compact_kernel(const T* __restrict__ xyz, ExtractParams P, ExtractStage st,
                                                      uint32_t* __restrict__ edge_idx, uint32_t* __restrict__ n_edge,
                                                      double* __restrict__ edge_xyz, size_t edge_stride,
                                                      uint32_t* __restrict__ planar_idx,
                                                      uint32_t* __restrict__ n_planar, double* __restrict__ planar_xyz,
                                                      size_t planar_stride, const uint32_t* __restrict__ only_if,
                                                      unsigned long long* __restrict__ fallback_counter) {
  __shared__ uint32_t s_scan[256], s_off[256], s_cnt[256];
  // the fallback of the fused compaction: nothing to do unless a scan line of select_mis_kernel gave up (uniform)
  if (only_if && __hip_atomic_load(only_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
  if (only_if && fallback_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(fallback_counter, 1ull);
  const size_t scan = blockIdx.x / kCompactSplit;
  const uint32_t split = blockIdx.x % kCompactSplit;
  const uint32_t groups = P.H * P.S;
  const T* scan_xyz = xyz + scan * (size_t)P.H * P.W * 3;
  compact_one(scan_xyz, groups, P.cap_edge, st.edge_stage + scan * (size_t)groups * P.cap_edge,
              st.edge_cnt + scan * (size_t)groups, edge_idx ? edge_idx + scan * edge_stride : nullptr,
              edge_xyz ? edge_xyz + scan * edge_stride * 3 : nullptr, n_edge + scan, split, s_scan, s_off, s_cnt);
  compact_one(scan_xyz, groups, P.cap_planar, st.planar_stage + scan * (size_t)groups * P.cap_planar,
              st.planar_cnt + scan * (size_t)groups, planar_idx ? planar_idx + scan * planar_stride : nullptr,
              planar_xyz ? planar_xyz + scan * planar_stride * 3 : nullptr, n_planar + scan, split, s_scan, s_off, s_cnt);
}

}  // namespace

template <typename T>
static void launch_curvature_valid_t(const T* d_xyz, const dim3& grid, const ExtractParams& P, double* d_curv, uint8_t* d_mask,
                                     hipStream_t s) {
  if (P.np == 3 && !(P.flags & kFlagCurvV1)) {  // the reference's default neighbor_points (features.h:40)
    const dim3 grid2(grid.x, (P.W + kCurvTilesPerGroup * kTile - 1) / (kCurvTilesPerGroup * kTile));
    if (P.flags & kFlagSplitCurv) launch_kernel((curvature_valid2_kernel<3, T, true>), grid2, dim3(kCurvThreads), 0, s, d_xyz, P, d_curv, d_mask);
    else launch_kernel((curvature_valid2_kernel<3, T>), grid2, dim3(kCurvThreads), 0, s, d_xyz, P, d_curv, d_mask);
    return;
  }
  if (P.np == 3)
    launch_kernel((curvature_valid_kernel<3, T>), grid, dim3(kCurvThreads), 0, s, d_xyz, P, d_curv, d_mask);
  else
    launch_kernel((curvature_valid_kernel<0, T>), grid, dim3(kCurvThreads), 0, s, d_xyz, P, d_curv, d_mask);
}

void launch_curvature_valid(const void* d_xyz, bool f32, size_t n_scans, const ExtractParams& P, double* d_curv,
                            uint8_t* d_mask, hipStream_t s) {
  const size_t n_lines = n_scans * P.H;
  if (n_lines == 0 || P.W == 0) return;
  const dim3 grid((unsigned)n_lines, (P.W + kTile - 1) / kTile);
  if (f32) launch_curvature_valid_t(static_cast<const float*>(d_xyz), grid, P, d_curv, d_mask, s);
  else launch_curvature_valid_t(static_cast<const double*>(d_xyz), grid, P, d_curv, d_mask, s);
}


template <int R, bool TWO>
static void launch_select_mis2(const double* d_curv, const uint8_t* d_mask, size_t n_lines, const ExtractParams& P,
                               const ExtractStage& st, const ExtractFused& fz, hipStream_t s) {
  const size_t per_wave = select_mis_lds_bytes((int)P.W, TWO ? 128 : 64);
#if defined(LOAMX_SELECT_ONE_WAVE)
  if (false) {
#else
  if (per_wave * 4 <= 48 * 1024) {
#endif
    // (the reference's neighbor_points = 3 on the two usual line lengths: W compiled in)
    if (R == 2 && !TWO && P.W == 1024)
      launch_kernel((select_mis_kernel<R, 4, TWO, (R == 2 && !TWO) ? 1024 : 0>), dim3((unsigned)((n_lines + 3) / 4)), dim3(256), per_wave * 4, s,
                         d_curv, d_mask, n_lines, P, st, fz);
    else if (R == 2 && TWO && P.W == 2048)
      launch_kernel((select_mis_kernel<R, 4, TWO, (R == 2 && TWO) ? 2048 : 0>), dim3((unsigned)((n_lines + 3) / 4)), dim3(256), per_wave * 4, s,
                         d_curv, d_mask, n_lines, P, st, fz);
    else
      launch_kernel((select_mis_kernel<R, 4, TWO>), dim3((unsigned)((n_lines + 3) / 4)), dim3(256), per_wave * 4, s, d_curv,
                         d_mask, n_lines, P, st, fz);
  } else {
    launch_kernel((select_mis_kernel<R, 1, TWO>), dim3((unsigned)n_lines), dim3(64), per_wave, s, d_curv, d_mask, n_lines,
                       P, st, fz);
  }
}
template <int R>
static void launch_select_mis(const double* d_curv, const uint8_t* d_mask, size_t n_lines, const ExtractParams& P,
                              const ExtractStage& st, const ExtractFused& fz, hipStream_t s) {
  if (select_mis_slots(P, R) > 64) launch_select_mis2<R, true>(d_curv, d_mask, n_lines, P, st, fz, s);
  else launch_select_mis2<R, false>(d_curv, d_mask, n_lines, P, st, fz, s);
}

// Four scan lines per wavefront (select_rows.h) where the parameters allow; false: nothing was launched.
template <int R>
static void launch_select_rows_r(const double* d_curv, const uint8_t* d_mask, size_t n_lines, const ExtractParams& P, const ExtractStage& st,
                                 const ExtractFused& fz, const RowSelGeom& G, hipStream_t s) {
  const dim3 grid((unsigned)((n_lines + 15) / 16));
  if (fz.only_if) {  // the conditional second launch (launch_select_rows)
    if (R == 2 && G.ch == 11 && (P.flags & kFlagSplitCurv))
      launch_kernel((select_rows_stage_kernel<R, R == 2 ? 11 : 0, R == 2>), grid, dim3(256), (size_t)G.bytes * 4, s, d_curv, d_mask, n_lines, P, st, fz, G);
    else if (R == 2 && G.ch == 11)
      launch_kernel((select_rows_stage_kernel<R, R == 2 ? 11 : 0, false>), grid, dim3(256), (size_t)G.bytes * 4, s, d_curv, d_mask, n_lines, P, st, fz, G);
    else
      launch_kernel((select_rows_stage_kernel<R, 0, false>), grid, dim3(256), (size_t)G.bytes * 4, s, d_curv, d_mask, n_lines, P, st, fz, G);
    return;
  }
  if (R == 2 && G.ch == 11 && (P.flags & kFlagSplitCurv))
    launch_kernel((select_rows_kernel<R, R == 2 ? 11 : 0, false, R == 2>), grid, dim3(256), (size_t)G.bytes * 4, s, d_curv, d_mask, n_lines, P, st, fz, G);
  else if (R == 2 && G.ch == 11)
    launch_kernel((select_rows_kernel<R, R == 2 ? 11 : 0>), grid, dim3(256), (size_t)G.bytes * 4, s, d_curv, d_mask, n_lines, P, st, fz, G);
  else
    launch_kernel((select_rows_kernel<R, 0>), grid, dim3(256), (size_t)G.bytes * 4, s, d_curv, d_mask, n_lines, P, st, fz, G);
}
static bool launch_select_rows(const double* d_curv, const uint8_t* d_mask, size_t n_lines, const ExtractParams& P, const ExtractStage& st,
                               const ExtractFused& fz, hipStream_t s) {
  RowSelGeom G;
  if (!row_select_geom(P, G) || (reinterpret_cast<uintptr_t>(d_mask) & 15u) != 0 || n_lines % 4 != 0) return false;
  if (fz.fuse && P.W > 65535) return false;
  if ((P.flags & kFlagSplitCurv) && !(P.np == 3 && G.ch == 11)) return false;  // (launch_extract_split_ok said otherwise: not reached)
  // With the fused compaction: first without the stage arrays, then — a launch that leaves at once unless a line gave up or
  // was tied — the plain selection that writes them for replay_kernel / compact_kernel (ExtractFused::no_stage / only_if)
  const bool two = fz.fuse && fz.error != nullptr && fz.line_tot != nullptr && !(P.flags & kFlagStageAlways);
  for (int pass = 0; pass < (two ? 2 : 1); pass++) {
    ExtractFused f = fz;
    if (two && pass == 0) f.no_stage = 1u;
    if (two && pass == 1) f.fuse = 0u, f.only_if = fz.error, f.box_min = f.box_max = nullptr;
    switch ((int)P.np - 1) {
      case 1: launch_select_rows_r<1>(d_curv, d_mask, n_lines, P, st, f, G, s); break;
      case 2: launch_select_rows_r<2>(d_curv, d_mask, n_lines, P, st, f, G, s); break;
      case 3: launch_select_rows_r<3>(d_curv, d_mask, n_lines, P, st, f, G, s); break;
      default: launch_select_rows_r<4>(d_curv, d_mask, n_lines, P, st, f, G, s); break;
    }
  }
  return true;
}

// Opt-in (context option FUSED_ROWS): rows a5-a10 in one kernel, four scan lines per wavefront (select_rows_kernel<.., FUSED>): for the reference's default
// neighbor_points on sectors of 161-174 points (64 x 1024 scans in 6 sectors). false: not applicable, nothing was launched.
bool launch_extract_rows_fused(const void* d_xyz, bool f32, size_t n_scans, const ExtractParams& P, const ExtractStage& st,
                               const ExtractFused& fz_in, double* d_curv, uint8_t* d_mask, hipStream_t s) {
  const size_t n_lines = n_scans * P.H;
  if (n_lines == 0 || P.W == 0 || (P.flags & (kFlagNoMisSelect | kFlagNoRowSelect | kFlagFusedExtract)) || !(P.flags & kFlagFusedRows)) return false;
  RowSelGeom G;
  if (P.np != 3 || !fz_in.line_tot || !fz_in.error || !row_select_geom(P, G, true) || G.ch != 11 || n_lines % 4 != 0) return false;
  ExtractFused fz = fz_in;
  fz.fuse = (P.S <= 64 && !(P.flags & kFlagNoFusedCompact)) ? 1u : 0u;
  fz.xyz = d_xyz, fz.f32 = f32 ? 1u : 0u;
  launch_kernel((select_rows_kernel<2, 11, true>), dim3((unsigned)((n_lines + 7) / 8)), dim3(128), (size_t)G.bytes * 2, s,
                static_cast<const double*>(nullptr), static_cast<const uint8_t*>(nullptr), n_lines, P, st, fz, G);
  const unsigned grid = (unsigned)(n_lines < 4096 ? n_lines : 4096);
  if (f32)
    launch_kernel((curvature_tied_kernel<3, float>), dim3(grid), dim3(64), 0, s, static_cast<const float*>(d_xyz), n_lines, P, fz.line_tot, fz.error, d_curv, d_mask);
  else
    launch_kernel((curvature_tied_kernel<3, double>), dim3(grid), dim3(64), 0, s, static_cast<const double*>(d_xyz), n_lines, P, fz.line_tot, fz.error, d_curv, d_mask);
  return true;
}

// May the curvature go from curvature_valid2_kernel to select_rows_kernel in the split form (hi words | lo words, validity in
// the sign bit: see those kernels)? Exactly when launch_curvature_valid and launch_select will run those two kernels in the
// instantiations that know the form, and the thresholds are comparable through their hi words (finite, not negative).
bool launch_extract_split_ok(const ExtractParams& P, size_t n_scans) {
  if (P.flags & (kFlagNoSplitCurv | kFlagNoMisSelect | kFlagNoRowSelect | kFlagCurvV1 | kFlagFusedExtract | kFlagFusedRows)) return false;
  RowSelGeom G;
  if (P.np != 3 || (P.W & 1u) || !row_select_geom(P, G) || G.ch != 11 || (n_scans * P.H) % 4 != 0) return false;
  if (!(P.edge_thr >= 0.0 && P.edge_thr <= 1.7976931348623157e308 && P.planar_thr >= 0.0 && P.planar_thr <= 1.7976931348623157e308)) return false;
  // (-0.0 passes ">= 0.0", but its hi word is 0x80000000: compared as signed integers a +0.0 curvature would beat it)
  if (__builtin_signbit(P.edge_thr) || __builtin_signbit(P.planar_thr)) return false;
  return true;
}

bool launch_select_takes_boxes(const ExtractParams& P) {
  RowSelGeom G;
  return !(P.flags & (kFlagNoMisSelect | kFlagNoRowSelect | kFlagNoFusedCompact | kFlagFusedExtract)) && P.S <= 64 && row_select_geom(P, G);
}

bool launch_select(const double* d_curv, const uint8_t* d_mask, size_t n_scans, const ExtractParams& P,
                   const ExtractStage& st, const ExtractFused* fused, hipStream_t s, bool* rows_ran) {
  if (rows_ran) *rows_ran = false;
  const size_t n_lines = n_scans * P.H;
  if (n_lines == 0 || P.W == 0) return false;
  // the fused compaction keeps the per-sector counts of a line on the lanes of its wavefront
  const bool fuse = fused && fused->line_tot && P.S <= 64 && !(P.flags & kFlagNoFusedCompact);
  ExtractFused fz{};
  if (fused) fz = *fused;  // (line_tot / error carry the tie flags either way)
  fz.fuse = fuse ? 1u : 0u;
  // bitmask-MIS fast path: R = np-1 in 1..4, lane chunk wide enough for the halo, at most 64 picks
  // per sector (picks are >= R+1 points apart), the cap itself at most 64
  const int R = (int)P.np - 1, CH = ((int)P.W + 63) / 64;
  const uint32_t longest = P.W - (P.S - 1) * P.pps;
  const uint32_t picks = (longest + R) / (R + 1);  // most picks a sector can yield
  const bool mis_ok = R >= 1 && R <= 4 && CH >= R && CH + 2 * R <= 64 &&
                      (picks <= 64 || (picks <= 128 && P.cap_edge <= 64 && P.cap_planar <= 64));
  if (!(P.flags & (kFlagNoMisSelect | kFlagNoRowSelect)) && launch_select_rows(d_curv, d_mask, n_lines, P, st, fz, s)) {
    if (rows_ran) *rows_ran = true;
    return fuse;
  }
  if (mis_ok && !(P.flags & kFlagNoMisSelect)) {
    switch (R) {
      case 1: launch_select_mis<1>(d_curv, d_mask, n_lines, P, st, fz, s); return fuse;
      case 2: launch_select_mis<2>(d_curv, d_mask, n_lines, P, st, fz, s); return fuse;
      case 3: launch_select_mis<3>(d_curv, d_mask, n_lines, P, st, fz, s); return fuse;
      default: launch_select_mis<4>(d_curv, d_mask, n_lines, P, st, fz, s); return fuse;
    }
  }
  const size_t per_wave = (size_t)P.W * 8 + (((size_t)P.W + 7) & ~(size_t)7);
  if (P.W <= 1024) {
    constexpr int WAVES = 4;
    launch_kernel(select_kernel<WAVES>, dim3((unsigned)((n_lines + WAVES - 1) / WAVES)), dim3(WAVES * 64),
                       per_wave * WAVES, s, d_curv, d_mask, n_lines, P, st, fz.line_tot, fz.error);
  } else {
    launch_kernel(select_kernel<1>, dim3((unsigned)n_lines), dim3(64), per_wave, s, d_curv, d_mask, n_lines, P,
                       st, fz.line_tot, fz.error);
  }
  return false;
}

// The fused path can run when the selection's fast path does (launch_select: mis_ok, one pick per lane), the compaction
// can be fused (number_sectors <= 64) and neighbor_points is the compiled-in default.
bool launch_extract_fused(const void* d_xyz, bool f32, size_t n_scans, const ExtractParams& P, const ExtractStage& st,
                          const ExtractFused& fz_in, double* d_curv, uint8_t* d_mask, hipStream_t s) {
  const size_t n_lines = n_scans * P.H;
  // Opt-in (context option FUSED_EXTRACT): measured on 2 048 scans of 64 x 1024, the fused kernel moves 7.9 GB instead of the
  // 10.3 GB of the two kernels (PMC FETCH_SIZE x 2 + WRITE_SIZE; 4.2 GB are algorithmic) but takes 2.92 ms against their
  // 0.94 + 1.86 ms: both phases are instruction-bound, so fusing them saves traffic, not time, and the gather of the
  // picked points still re-reads most of the scan (their lines have left the L2 by the time the selection is done).
  if (n_lines == 0 || P.W == 0 || !(P.flags & kFlagFusedExtract) || (P.flags & (kFlagNoFusedCompact | kFlagNoMisSelect))) return false;
  const int R = (int)P.np - 1, CH = ((int)P.W + 63) / 64;
  const uint32_t longest = P.W - (P.S - 1) * P.pps;
  const uint32_t picks = (longest + R) / (R + 1);
  if (P.np != 3 || !(CH >= R && CH + 2 * R <= 64) || picks > 64 || P.S > 64 || !fz_in.line_tot || (P.W & 1)) return false;
  const size_t lds = ((Curv2<3, 64>::kLdsBytes + 15) & ~(size_t)15) + select_mis_lds_bytes((int)P.W, 64) +
                     (((size_t)P.W + 256 + 7) & ~(size_t)7);  // + the line's validity bytes (4 spare per lane chunk)
  if (lds > 64 * 1024) return false;  // (wider lines: the separate kernels)
  ExtractFused fz = fz_in;
  fz.fuse = 1u;
  const dim3 grid((unsigned)n_lines);
  if (f32)
    launch_kernel((extract_fused_kernel<3, 2, float>), grid, dim3(64), lds, s, static_cast<const float*>(d_xyz), n_lines, P, st, fz, d_curv, d_mask);
  else
    launch_kernel((extract_fused_kernel<3, 2, double>), grid, dim3(64), lds, s, static_cast<const double*>(d_xyz), n_lines, P, st, fz, d_curv, d_mask);
  return true;
}

// What the fused selection needs cleared before it runs — the per-line slots of its chained scan (+ the give-up word behind
// them) and the bounding boxes its copy phase takes by atomic min / max — in ONE launch: the three hipMemsetAsync calls this
// replaces were four fill kernels with ~35 us of stream turnaround between a step's last kernel and the next curvature pass.
__global__ __launch_bounds__(256) void extract_init_kernel(unsigned long long* __restrict__ line_tot, size_t n_tot,
                                                           unsigned long long* __restrict__ box_min, unsigned long long* __restrict__ box_max,
                                                           size_t n_box) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_tot) line_tot[i] = 0ull;
  if (i < n_box) box_min[i] = ~0ull, box_max[i] = 0ull;
}
void launch_extract_init(unsigned long long* line_tot, size_t n_tot, unsigned long long* box_min, unsigned long long* box_max, size_t n_box,
                         hipStream_t s) {
  const size_t n = n_tot > n_box ? n_tot : n_box;
  if (n == 0) return;
  launch_kernel(extract_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, line_tot, n_tot, box_min, box_max, box_min ? n_box : 0);
}

void launch_replay(const double* d_curv, const uint8_t* d_mask, size_t n_scans, const ExtractParams& P, const ExtractStage& st,
                   const ExtractFused& fz, hipStream_t s) {
  const size_t n_lines = n_scans * P.H;
  if (n_lines == 0 || P.W == 0 || !fz.line_tot || !fz.error) return;
  const unsigned grid = (unsigned)(n_lines < 2048 ? n_lines : 2048);
  launch_kernel(replay_kernel, dim3(grid), dim3(64), replay_lds_bytes((int)P.W), s, d_curv, d_mask, n_lines, P, st, fz.line_tot, fz.error,
                fz.events);
}

void launch_compact(const void* d_xyz, bool f32, size_t n_scans, const ExtractParams& P, const ExtractStage& st,
                    uint32_t* d_edge_idx, uint32_t* d_n_edge, double* d_edge_xyz, size_t edge_stride,
                    uint32_t* d_planar_idx, uint32_t* d_n_planar, double* d_planar_xyz, size_t planar_stride,
                    hipStream_t s, const uint32_t* only_if, unsigned long long* fallback_counter) {
  if (n_scans == 0) return;
  const dim3 grid((unsigned)(n_scans * kCompactSplit));
  if (f32)
    launch_kernel(compact_kernel<float>, grid, dim3(256), 0, s, static_cast<const float*>(d_xyz), P, st, d_edge_idx, d_n_edge,
                       d_edge_xyz, edge_stride, d_planar_idx, d_n_planar, d_planar_xyz, planar_stride, only_if, fallback_counter);
  else
    launch_kernel(compact_kernel<double>, grid, dim3(256), 0, s, static_cast<const double*>(d_xyz), P, st, d_edge_idx, d_n_edge,
                       d_edge_xyz, edge_stride, d_planar_idx, d_n_planar, d_planar_xyz, planar_stride, only_if, fallback_counter);
}

}  // namespace loamx
