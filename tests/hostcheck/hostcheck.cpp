// hostcheck.cpp — TEST-ONLY serial emulation of the HIP kernels' per-thread math on the CPU.
//
// It includes the product's __host__ __device__ headers (extract_math.h, reg_math.h, synth.h) and
// replaces each kernel's thread grid / wave reduction by plain loops, so that the arithmetic the
// GPU will execute can be compared with the oracle in this GPU-less container. It is built only by
// tests/ (tests/hostcheck/Makefile), is never linked into libloamx.so and is not a product path.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

// search statistics (candidates / rows visited) for tools/knn_stats.py
#define LOAMX_KNN_STATS 1
static uint64_t g_cand = 0, g_rows = 0, g_general = 0;
static uint32_t g_lean_trips = 0, g_lean_nrow = 0, g_lean_taken = 0, g_lean_reason = 0;
#define LOAMX_STL_SORT_STATS 1
static uint64_t g_heap_sorts = 0;  // times stl_sort fell back to heap sort (depth limit)

#include "../../include/loamx.h"
#include "../../loam_amd/csrc/extract_math.h"
#include "../../loam_amd/csrc/reg_math.h"
#include "../../loam_amd/csrc/synth.h"

using namespace loamx;

static ExtractParams make_params(uint64_t H, uint64_t W, double rmin, double rmax, const loamx_fe_params* fe) {
  ExtractParams P{};
  P.H = (uint32_t)H, P.W = (uint32_t)W, P.np = (uint32_t)fe->neighbor_points, P.S = (uint32_t)fe->number_sectors;
  P.pps = P.S ? P.W / P.S : 0;
  P.max_edge = (uint32_t)fe->max_edge_feats_per_sector, P.max_planar = (uint32_t)fe->max_planar_feats_per_sector;
  P.min_range = rmin, P.max_range = rmax;
  P.edge_thr = fe->edge_feat_threshold, P.planar_thr = fe->planar_feat_threshold;
  P.occ_thr = fe->occlusion_thresh, P.par_thr = fe->parallel_thresh;
  return P;
}

namespace {
void replay_line(const double* c, std::vector<uint8_t> valid, const ExtractParams& P, uint32_t line, std::vector<uint32_t>& edge,
                 std::vector<uint32_t>& planar);
uint64_t g_replayed_lines = 0;
}  // namespace

extern "C" {

// curvature + validity exactly as curvature_valid_kernel computes them
int hostcheck_curvature_valid(const double* xyz, uint64_t H, uint64_t W, double rmin, double rmax,
                              const loamx_fe_params* fe, double* curv, uint8_t* mask) {
  const ExtractParams P = make_params(H, W, rmin, rmax, fe);
  std::vector<double> r(W);
  std::vector<uint8_t> code(W);
  for (uint64_t line = 0; line < H; line++) {
    const double* p = xyz + line * W * 3;
    for (uint32_t c = 0; c < W; c++) r[c] = point_range(p[3 * c], p[3 * c + 1], p[3 * c + 2]);
    for (uint32_t c = 0; c < W; c++)
      code[c] = is_line_end(c, P.W, P.np) ? (uint8_t)kCodeNone : point_code(r[c - 1], r[c], r[c + 1], P);
    for (uint32_t c = 0; c < W; c++) {
      curv[line * W + c] = is_line_end(c, P.W, P.np) ? -1.0 : curvature_at(p, (int)c, P.np);
      mask[line * W + c] = valid_from_codes(code.data(), (int)c, c, P.W, P.np) ? 1 : 0;
    }
  }
  return 0;
}

// per-sector greedy selection exactly as select_kernel does it (repeated arg-max / arg-min under
// the total order of extract_math.h, suppression of +-(np-1))
int hostcheck_select(const double* curv, const uint8_t* mask_in, uint64_t H, uint64_t W, const loamx_fe_params* fe,
                     uint32_t* edge_idx, uint64_t* n_edge, uint32_t* planar_idx, uint64_t* n_planar) {
  const ExtractParams P = make_params(H, W, 0, 0, fe);
  uint64_t ne = 0, npl = 0;
  std::vector<uint8_t> valid(W);
  for (uint64_t line = 0; line < H; line++) {
    const double* c = curv + line * W;
    for (uint32_t i = 0; i < W; i++) valid[i] = mask_in[line * W + i];
    const std::vector<uint8_t> valid0 = valid;
    const uint64_t ne0 = ne, npl0 = npl;
    bool tie = false;
    for (uint32_t s = 0; s < P.S; s++) {
      const uint32_t start = s * P.pps, end = (s == P.S - 1) ? P.W : start + P.pps;
      for (int pass = 0; pass < 2; pass++) {
        const uint32_t maxf = pass == 0 ? P.max_edge : P.max_planar;
        uint32_t n = 0;
        for (;;) {
          int32_t best = -1;
          double bc = 0;
          for (uint32_t i = start; i < end; i++) {
            if (!valid[i]) continue;
            if (pass == 0 ? !(c[i] > P.edge_thr) : !(c[i] < P.planar_thr)) continue;
            if (best < 0 || (pass == 0 ? edge_before(c[i], (int32_t)i, bc, best) : planar_before(c[i], (int32_t)i, bc, best)))
              best = (int32_t)i, bc = c[i];
          }
          if (best < 0) break;
          for (uint32_t i = start; i < end; i++)  // the best is not unique (select_pass: tied)
            if ((int32_t)i != best && valid[i] && c[i] == bc) tie = true;
          if (pass == 0)
            edge_idx[ne++] = (uint32_t)(line * W + best);
          else
            planar_idx[npl++] = (uint32_t)(line * W + best);
          for (uint32_t k = 0; k < P.np; k++) valid[best + k] = 0, valid[best - k] = 0;
          n++;
          if (n > maxf) break;
        }
      }
    }
    if (tie || (P.flags & kFlagForceReplay)) {  // select_kernel: the line again, in the reference's own order
      std::vector<uint32_t> e, p;
      replay_line(c, valid0, P, (uint32_t)line, e, p);
      ne = ne0, npl = npl0;
      for (uint32_t v : e) edge_idx[ne++] = v;
      for (uint32_t v : p) planar_idx[npl++] = v;
      g_replayed_lines++;
    }
  }
  *n_edge = ne, *n_planar = npl;
  return 0;
}


/* ---- lane-level emulation of select_mis_kernel (bitmask MIS + cap by priority order) -------------- */
}  // extern "C"
namespace {
// the serial twin of ring_replay (extract_kernels.hip): stl_sort per sector + the reference's two walks
void replay_line(const double* c, std::vector<uint8_t> valid, const ExtractParams& P, uint32_t line, std::vector<uint32_t>& edge,
                 std::vector<uint32_t>& planar) {
  const int W = (int)P.W, np = (int)P.np;
  std::vector<uint16_t> ord(W);
  for (int i = 0; i < W; i++) ord[i] = (uint16_t)i;
  for (uint32_t s = 0; s < P.S; s++) {
    const int start = (int)(s * P.pps), end = (s == P.S - 1) ? W : start + (int)P.pps;
    stl_sort(ord.data(), start, end, [&](uint16_t a, uint16_t b) { return c[a] < c[b]; });
  }
  for (uint32_t s = 0; s < P.S; s++) {
    const int start = (int)(s * P.pps), end = (s == P.S - 1) ? W : start + (int)P.pps;
    uint32_t ne = 0, npl = 0;
    for (int k = end; k > start; k--) {
      const int idx = ord[k - 1];
      if (valid[idx] && c[idx] > P.edge_thr) {
        edge.push_back(line * P.W + (uint32_t)idx);
        for (int n = 0; n < np; n++) {
          if (idx + n < W) valid[idx + n] = 0;
          if (idx - n >= 0) valid[idx - n] = 0;
        }
        ne++;
      }
      if (ne > P.max_edge) break;
    }
    for (int k = start; k < end; k++) {
      const int idx = ord[k];
      if (valid[idx] && c[idx] < P.planar_thr) {
        planar.push_back(line * P.W + (uint32_t)idx);
        for (int n = 0; n < np; n++) {
          if (idx + n < W) valid[idx + n] = 0;
          if (idx - n >= 0) valid[idx - n] = 0;
        }
        npl++;
      }
      if (npl > P.max_planar) break;
    }
  }
}


template <int R>
void select_mis_line(const double* c, std::vector<uint8_t>& valid, const ExtractParams& P, uint32_t line,
                     std::vector<uint32_t>& edge, std::vector<uint32_t>& planar, bool* ok) {
  const int W = (int)P.W, CH = (W + 63) / 64;
  if (CH < R || CH + 2 * R > 64) {
    *ok = false;
    return;
  }
  const std::vector<uint8_t> valid0 = valid;
  const size_t edge0 = edge.size(), planar0 = planar.size();
  bool tie = false;
  uint64_t V[64], ET[64], PT[64], GT[64][R > 0 ? R : 1], EQ[64][R > 0 ? R : 1];
  for (int l = 0; l < 64; l++) {
    V[l] = ET[l] = PT[l] = 0;
    for (int j = 0; j < CH; j++) {
      const int i = l * CH + j;
      if (i >= W) continue;
      if (valid[i]) V[l] |= 1ull << j;
      if (c[i] > P.edge_thr) ET[l] |= 1ull << j;
      if (c[i] < P.planar_thr) PT[l] |= 1ull << j;
    }
    for (int d = 1; d <= R; d++) {
      uint64_t g = 0;
      for (int t = 0; t < CH + 2 * R; t++) {
        const int i = l * CH - R + t;
        if (i >= 0 && i + d < W && c[i] > c[i + d]) g |= 1ull << t;
      }
      GT[l][d - 1] = g;
      uint64_t e = 0;
      for (int t = 0; t < CH + 2 * R; t++) {
        const int i = l * CH - R + t;
        if (i >= 0 && i + d < W && c[i] == c[i + d]) e |= 1ull << t;
      }
      EQ[l][d - 1] = e;
    }
  }
  const uint64_t cm = low_mask(CH);
  for (uint32_t s = 0; s < P.S; s++) {
    const int start = (int)(s * P.pps), end = (s == P.S - 1) ? W : start + (int)P.pps;
    for (int pass = 0; pass < 2; pass++) {
      const bool EDGE = pass == 0;
      const uint32_t maxf = EDGE ? P.max_edge : P.max_planar;
      uint64_t U[64], Pk[64];
      bool any = false;
      for (int l = 0; l < 64; l++) {
        uint64_t sm = 0;
        for (int j = 0; j < CH; j++) {
          const int i = l * CH + j;
          if (i >= start && i < end) sm |= 1ull << j;
        }
        U[l] = V[l] & (EDGE ? ET[l] : PT[l]) & sm;
        Pk[l] = 0;
        any |= U[l] != 0;
      }
      for (int l = 0; l < 64; l++) {  // equal curvatures among candidates within R points of each other (mis_pass)
        const uint64_t Uw = mis_window<R>(U[l], l ? U[l - 1] : 0, l < 63 ? U[l + 1] : 0, CH);
        for (int d = 1; d <= R; d++)
          if (Uw & (Uw >> d) & EQ[l][d - 1]) tie = true;
      }
      while (any) {
        uint64_t win[64];
        for (int l = 0; l < 64; l++) {
          const uint64_t Uw = mis_window<R>(U[l], l ? U[l - 1] : 0, l < 63 ? U[l + 1] : 0, CH);
          const uint64_t w = EDGE ? mis_winners<R, true>(Uw, GT[l]) : mis_winners<R, false>(Uw, GT[l]);
          win[l] = (w >> R) & cm;
        }
        any = false;
        uint64_t nU[64];
        for (int l = 0; l < 64; l++) {
          const uint64_t Ww = mis_window<R>(win[l], l ? win[l - 1] : 0, l < 63 ? win[l + 1] : 0, CH);
          const uint64_t rem = (mis_spread<R>(Ww) >> R) & cm;
          Pk[l] |= win[l];
          nU[l] = U[l] & ~rem;
          any |= nU[l] != 0;
        }
        for (int l = 0; l < 64; l++) U[l] = nU[l];
      }
      // members in priority order, keep the first max+1
      std::vector<std::pair<double, int>> mem;
      for (int l = 0; l < 64; l++)
        for (int j = 0; j < CH; j++)
          if (Pk[l] >> j & 1) mem.push_back({c[l * CH + j], l * CH + j});
      std::sort(mem.begin(), mem.end(), [&](const std::pair<double, int>& a, const std::pair<double, int>& b) {
        return EDGE ? edge_before(a.first, a.second, b.first, b.second) : planar_before(a.first, a.second, b.first, b.second);
      });
      for (size_t k = 0; k + 1 < mem.size(); k++)
        if (mem[k].first == mem[k + 1].first) tie = true;  // equal curvatures among the picks (the kernel: colliding keys)
      const size_t kept = std::min<size_t>(mem.size(), (size_t)maxf + 1);
      uint64_t K[64] = {0};
      for (size_t k = 0; k < kept; k++) {
        (EDGE ? edge : planar).push_back(line * P.W + (uint32_t)mem[k].second);
        K[mem[k].second / CH] |= 1ull << (mem[k].second % CH);
      }
      for (int l = 0; l < 64; l++) {
        const uint64_t Kw = mis_window<R>(K[l], l ? K[l - 1] : 0, l < 63 ? K[l + 1] : 0, CH);
        V[l] &= ~((mis_spread<R>(Kw) >> R) & cm);
      }
    }
  }
  if (tie || (P.flags & kFlagForceReplay)) {  // the reference's own order (select_mis_kernel: ring_replay)
    edge.resize(edge0), planar.resize(planar0);
    replay_line(c, valid0, P, line, edge, planar);
    g_replayed_lines++;
  }
  *ok = true;
}
}  // namespace
extern "C" {

uint64_t hostcheck_replayed_lines() { return g_replayed_lines; }

// features-inl.h:27-48 + :137-180 restated literally on given curvature / mask arrays with the REAL std::sort on the
// reference's element type: what select_mis_kernel must reproduce, ties included
int hostcheck_select_stdsort(const double* curv, const uint8_t* mask_in, uint64_t H, uint64_t W, const loamx_fe_params* fe,
                             uint32_t* edge_idx, uint64_t* n_edge, uint32_t* planar_idx, uint64_t* n_planar) {
  struct PC {
    size_t index;
    double curvature;
  };
  std::vector<PC> cv(H * W);
  std::vector<bool> valid(H * W);
  for (size_t i = 0; i < H * W; i++) cv[i] = PC{i, curv[i]}, valid[i] = mask_in[i] != 0;
  const size_t S = fe->number_sectors, pps = S ? W / S : 0, np = fe->neighbor_points;
  size_t ne = 0, npl = 0;
  for (size_t line = 0; line < H; line++)
    for (size_t sector = 0; sector < S; sector++) {
      const size_t start = line * W + sector * pps, end = sector == S - 1 ? (line + 1) * W : start + pps;
      std::sort(cv.begin() + start, cv.begin() + end, [](const PC& l, const PC& r) { return l.curvature < r.curvature; });
      size_t cnt = 0;
      for (size_t k = end; k > start; k--) {
        const PC c = cv[k - 1];
        if (valid[c.index] && c.curvature > fe->edge_feat_threshold) {
          edge_idx[ne++] = (uint32_t)c.index;
          for (size_t n = 0; n < np; n++) valid[c.index + n] = false, valid[c.index - n] = false;
          cnt++;
        }
        if (cnt > fe->max_edge_feats_per_sector) break;
      }
      cnt = 0;
      for (size_t k = start; k < end; k++) {
        const PC c = cv[k];
        if (valid[c.index] && c.curvature < fe->planar_feat_threshold) {
          planar_idx[npl++] = (uint32_t)c.index;
          for (size_t n = 0; n < np; n++) valid[c.index + n] = false, valid[c.index - n] = false;
          cnt++;
        }
        if (cnt > fe->max_planar_feats_per_sector) break;
      }
    }
  *n_edge = ne, *n_planar = npl;
  return 0;
}

// returns 0 if the MIS formulation applies to these parameters (else 1: kernel falls back)
int hostcheck_select_mis(const double* curv, const uint8_t* mask_in, uint64_t H, uint64_t W, const loamx_fe_params* fe,
                         uint32_t* edge_idx, uint64_t* n_edge, uint32_t* planar_idx, uint64_t* n_planar) {
  const ExtractParams P = make_params(H, W, 0, 0, fe);
  std::vector<uint32_t> e, p;
  std::vector<uint8_t> valid(W);
  for (uint64_t line = 0; line < H; line++) {
    for (uint32_t i = 0; i < W; i++) valid[i] = mask_in[line * W + i];
    bool ok = false;
    switch (P.np) {
      case 2: select_mis_line<1>(curv + line * W, valid, P, (uint32_t)line, e, p, &ok); break;
      case 3: select_mis_line<2>(curv + line * W, valid, P, (uint32_t)line, e, p, &ok); break;
      case 4: select_mis_line<3>(curv + line * W, valid, P, (uint32_t)line, e, p, &ok); break;
      case 5: select_mis_line<4>(curv + line * W, valid, P, (uint32_t)line, e, p, &ok); break;
      default: break;
    }
    if (!ok) return 1;
  }
  std::copy(e.begin(), e.end(), edge_idx);
  std::copy(p.begin(), p.end(), planar_idx);
  *n_edge = e.size(), *n_planar = p.size();
  return 0;
}

/* ---- registration ------------------------------------------------------------------------------ */
struct HostGrid {
  GridDesc g;
  std::vector<uint32_t> cell_start;
  std::vector<GridPoint> sp;
  std::vector<float> rel;  // 3 planes of sp.size() floats: offsets from the grid origin (FP32 pre-selection)
};

static void build_grid(const double* pts, uint32_t n, double max_dist, HostGrid& G) {
  Vec3 lo = v3(0, 0, 0), hi = v3(0, 0, 0);
  for (uint32_t i = 0; i < n; i++) {
    const Vec3 p = v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    if (i == 0) lo = hi = p;
    lo = v3(std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z));
    hi = v3(std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z));
  }
  grid_choose(G.g, lo, hi, n, max_dist, kGridCellsCap);
  const uint32_t ncell = (uint32_t)(G.g.nx * G.g.ny * G.g.nz);
  G.cell_start.assign(ncell + 1 + 4, 0);  // (+4 spare entries, as the device tables)
  std::vector<uint32_t> cell(n);
  for (uint32_t i = 0; i < n; i++) {
    cell[i] = grid_cell_of_point(G.g, v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
    G.cell_start[cell[i] + 1]++;
  }
  for (uint32_t c = 0; c < ncell; c++) G.cell_start[c + 1] += G.cell_start[c];
  std::vector<uint32_t> cursor(G.cell_start.begin(), G.cell_start.end() - 1);
  G.sp.assign((size_t)n + kGridPad, GridPoint{0, 0, 0, 0xFFFFFFFFu, 0});
  // the GPU scatters with atomics and then orders every cell by original index; ascending fill
  // order gives the same layout
  for (uint32_t k = 0; k < n; k++) {
    const uint32_t pos = cursor[cell[k]]++;
    G.sp[pos] = GridPoint{pts[3 * k], pts[3 * k + 1], pts[3 * k + 2], k, 0};
  }
  const size_t plane = G.sp.size();
  G.rel.assign(3 * plane, kRelPad);  // (the pad entries behind the set keep this value, as on the device)
  for (uint32_t p = 0; p < n; p++) {
    G.rel[p] = (float)(G.sp[p].x - G.g.ox), G.rel[plane + p] = (float)(G.sp[p].y - G.g.oy);
    G.rel[2 * plane + p] = (float)(G.sp[p].z - G.g.oz);
  }
}

extern "C++" {
// Runs the keyed fast path (what the kernels run) and the exact collector on the same query and
// requires identical answers; returns the answer. g_knn_fallbacks counts undecided keyed queries.
static uint64_t g_plane_streams = 0;
static uint64_t g_knn_fallbacks = 0, g_knn_mismatch = 0, g_knn_queued = 0, g_knn_round2 = 0;
template <int KM>
static int knn_both(const HostGrid& G, Vec3 q, int k, double max_dist, uint32_t pos[KM]) {
  uint32_t rows[kLean2RowWords], fb = 0;
  // as the kernels do: round-1-only keyed search; what it cannot finish goes to the complete search
  const double pass_max = knn_radius_pass_max(max_dist);
  int kept = knn_search_f32_round1<KM>(G.g, G.cell_start.data(), G.sp.data(), G.rel.data(), (uint32_t)G.sp.size(), q, k, max_dist,
                                       pass_max, pos, rows, 1);
  if (kept < 0) {
    g_knn_queued++;
    // the queue kernel (associate_knn_rest_kernel): a query that only ran out of 8-bit running numbers retries the FP32
    // pre-selection with wide ones, any other one the lean FP32 search of the 5x5x5 block; then the complete FP64 search
    if (kept == -2)
      kept = knn_search_f32_round1<KM, true>(G.g, G.cell_start.data(), G.sp.data(), G.rel.data(), (uint32_t)G.sp.size(), q, k,
                                             max_dist, pass_max, pos, rows, 1);
    else if (G.g.n_points <= kLeanMaxPoints) {
      kept = knn_lean_round2<KM>(G.g, G.cell_start.data(), G.sp.data(), G.rel.data(), (uint32_t)G.sp.size(), q, k, max_dist, pass_max, pos,
                                 rows, 1);
      if (kept >= 0) g_knn_round2++;
    }
  }
  if (kept < 0) {
    kept = knn_search_positions<KM>(G.g, G.cell_start.data(), G.sp.data(), q, k, max_dist, pass_max, pos, rows, 1, &fb);
  }
  g_knn_fallbacks += fb;
  KnnResult<KM> r;
  const int kept_exact = knn_search(G.g, G.cell_start.data(), G.sp.data(), q, k, max_dist, r, rows, 1);
  bool same = kept == kept_exact;
  for (int j = 0; same && j < kept; j++) same = pos[(KM - k) + j] == r.pos[j];
  for (int j = 0; j < kept; j++) pos[j] = pos[(KM - k) + j];  // callers read the plain prefix
  if (!same) g_knn_mismatch++;
  return kept;
}
}
uint64_t hostcheck_plane_streams(void) { return g_plane_streams; }
uint64_t hostcheck_knn_fallbacks(void) { return g_knn_fallbacks; }
uint64_t hostcheck_knn_queued(void) { return g_knn_queued; }
uint64_t hostcheck_knn_round2(void) { return g_knn_round2; }  // queued queries the lean 5x5x5 search finished
uint64_t hostcheck_knn_mismatches(void) { return g_knn_mismatch; }

uint64_t hostcheck_knn(const double* pts, uint64_t n, const double q[3], uint64_t k, double max_dist, uint64_t* idx_out) {
  HostGrid G;
  build_grid(pts, (uint32_t)n, max_dist, G);
  uint32_t pos[16];
  // the kernels are instantiated for KM = 5 (default parameters), 8 and 16
  const int kept = k <= 5 ? knn_both<5>(G, v3(q[0], q[1], q[2]), (int)k, max_dist, pos)
                          : (k <= 8 ? knn_both<8>(G, v3(q[0], q[1], q[2]), (int)k, max_dist, pos)
                                    : knn_both<16>(G, v3(q[0], q[1], q[2]), (int)k, max_dist, pos));
  for (int j = 0; j < kept; j++) idx_out[j] = G.sp[pos[j]].orig;
  return (uint64_t)kept;
}

// per-query search statistics of the keyed path (analysis only): candidates and rows visited;
// grid_out = {nx, ny, nz, h}
void hostcheck_knn_stats(const double* pts, uint64_t n, const double* queries, uint64_t nq, uint64_t k, double max_dist,
                         uint32_t* cand_out, uint32_t* rows_out, double grid_out[4], uint32_t* block_out) {
  HostGrid G;
  build_grid(pts, (uint32_t)n, max_dist, G);
  grid_out[0] = G.g.nx, grid_out[1] = G.g.ny, grid_out[2] = G.g.nz, grid_out[3] = G.g.h;
  const double pass = knn_radius_pass_max(max_dist);
  for (uint64_t i = 0; i < nq; i++) {
    uint32_t rows[kLean4RowWords], pos[8];
    g_cand = g_rows = g_general = 0;
    const Vec3 q = v3(queries[3 * i], queries[3 * i + 1], queries[3 * i + 2]);
    if (k <= 5) knn_search_keyed<5>(G.g, G.cell_start.data(), G.sp.data(), q, (int)k, max_dist, pass, pos, rows, 1);
    else knn_search_keyed<8>(G.g, G.cell_start.data(), G.sp.data(), q, (int)k, max_dist, pass, pos, rows, 1);
    cand_out[i] = (uint32_t)g_cand, rows_out[i] = (uint32_t)g_rows | ((uint32_t)g_general << 16);
    // points in the 3x3x3 block of cells around the query (known before the candidate loop starts)
    const int32_t cx = grid_cell_coord(q.x, G.g.ox, G.g.inv_h), cy = grid_cell_coord(q.y, G.g.oy, G.g.inv_h);
    const int32_t cz = grid_cell_coord(q.z, G.g.oz, G.g.inv_h);
    uint32_t blockpts = 0;
    for (int dz = -1; dz <= 1; dz++)
      for (int dy = -1; dy <= 1; dy++) {
        const int32_t iy = cy + dy, iz = cz + dz;
        const int32_t xa = std::max(cx - 1, 0), xb = std::min(cx + 1, G.g.nx - 1);
        if (iy < 0 || iy >= G.g.ny || iz < 0 || iz >= G.g.nz || xa > xb) continue;
        const uint32_t row = (uint32_t)((iz * G.g.ny + iy) * G.g.nx);
        blockpts += (G.cell_start[row + xb + 1] - G.cell_start[row + xa]);
      }
    block_out[i] = blockpts;
  }
}

// the lean round-1 search per query: out[5 * i + ...] = trips, non-empty rows, rows taken, candidates offered, return value + 2
void hostcheck_lean_stats(const double* pts, uint64_t n, const double* queries, uint64_t nq, uint64_t k, double max_dist,
                          uint32_t* out) {
  HostGrid G;
  build_grid(pts, (uint32_t)n, max_dist, G);
  const double pass = knn_radius_pass_max(max_dist);
  for (uint64_t i = 0; i < nq; i++) {
    uint32_t rows[kLean4RowWords], pos[8];
    g_cand = 0, g_lean_trips = g_lean_nrow = g_lean_taken = g_lean_reason = 0;
    const Vec3 q = v3(queries[3 * i], queries[3 * i + 1], queries[3 * i + 2]);
    const int r = knn_search_f32_round1<5>(G.g, G.cell_start.data(), G.sp.data(), G.rel.data(), (uint32_t)G.sp.size(), q, (int)k,
                                           max_dist, pass, pos, rows, 1);
    out[5 * i] = g_lean_trips, out[5 * i + 1] = g_lean_nrow, out[5 * i + 2] = g_lean_taken, out[5 * i + 3] = (uint32_t)g_cand;
    out[5 * i + 4] = (uint32_t)(r + 2) | (g_lean_reason << 8);
  }
}

// the queue's FP32 pass per query (analysis): out[i] = return value + 2 of knn_lean_round2 for the queries round 1 hands on
// with -1 (0xFF: round 1 finished the query, or handed it on as "too many batches")
void hostcheck_lean2_stats(const double* pts, uint64_t n, const double* queries, uint64_t nq, uint64_t k, double max_dist, uint32_t* out) {
  HostGrid G;
  build_grid(pts, (uint32_t)n, max_dist, G);
  const double pass = knn_radius_pass_max(max_dist);
  for (uint64_t i = 0; i < nq; i++) {
    uint32_t rows[kLean4RowWords], pos[8];
    const Vec3 q = v3(queries[3 * i], queries[3 * i + 1], queries[3 * i + 2]);
    const int r1 = knn_search_f32_round1<5>(G.g, G.cell_start.data(), G.sp.data(), G.rel.data(), (uint32_t)G.sp.size(), q, (int)k, max_dist, pass, pos, rows, 1);
    out[2 * i] = 0xFFu, out[2 * i + 1] = 0;
    if (r1 != -1) continue;
    g_lean_reason = 0;
    const int r2 = knn_lean_round2<5>(G.g, G.cell_start.data(), G.sp.data(), G.rel.data(), (uint32_t)G.sp.size(), q, (int)k, max_dist, pass, pos, rows, 1);
    out[2 * i] = (uint32_t)(r2 + 2), out[2 * i + 1] = g_lean_reason;
    if (r2 == -1) {  // (analysis: what the 9x9x9 block then does with it: bits 8.. = its return value + 4, bits 16.. = candidates)
      g_lean_trips = 0;
      const int r4 = knn_lean_block<5, 4>(G.g, G.cell_start.data(), G.sp.data(), G.rel.data(), (uint32_t)G.sp.size(), q, (int)k, max_dist, pass, pos, rows, 1);
      out[2 * i] |= (uint32_t)(r4 + 4) << 8;
      out[2 * i + 1] |= (uint32_t)g_lean_trips << 8;
    }
  }
}

// analysis only: trips of one query under three walks of its 3x3x3 block (exact distances, exact pruning):
// out[6 * i + ...] = {batches A, rejections A, batches B (rows sorted by slab distance, stop at the first miss),
//                     batches C (rows split into a near two-cell and a far one-cell piece, sorted), candidates A, candidates C}
void hostcheck_walk_stats(const double* pts, uint64_t n, const double* queries, uint64_t nq, uint64_t k, double max_dist,
                          uint32_t* out) {
  HostGrid G;
  build_grid(pts, (uint32_t)n, max_dist, G);
  const GridDesc& g = G.g;
  const double r2 = max_dist > 0 ? max_dist * max_dist : 1e300;
  struct Piece { double s2; uint32_t b, e; };
  for (uint64_t qi = 0; qi < nq; qi++) {
    const Vec3 q = v3(queries[3 * qi], queries[3 * qi + 1], queries[3 * qi + 2]);
    const int32_t cx = grid_cell_coord(q.x, g.ox, g.inv_h), cy = grid_cell_coord(q.y, g.oy, g.inv_h), cz = grid_cell_coord(q.z, g.oz, g.inv_h);
    for (int j = 0; j < 6; j++) out[6 * qi + j] = 0;
    if (grid_outside_distance(g, cx, cy, cz) > 1) continue;
    auto gap = [&](double v, double o, int32_t c, int32_t cc) {  // distance from v to cell cc along one axis
      if (cc == c) return 0.0;
      const double lo = o + cc * g.h, hi = lo + g.h;
      return cc < c ? v - hi : lo - v;
    };
    std::vector<Piece> rows, pieces;
    constexpr int kOrder[9] = {4, 1, 3, 5, 7, 0, 2, 6, 8};
    const bool left = (q.x - (g.ox + cx * g.h)) < 0.5 * g.h;
    for (int o = 0; o < 9; o++) {
      const int j = kOrder[o];
      const int32_t iy = cy + (j % 3) - 1, iz = cz + (j / 3) - 1;
      if (iy < 0 || iy >= g.ny || iz < 0 || iz >= g.nz) continue;
      const double gy = gap(q.y, g.oy, cy, iy), gz = gap(q.z, g.oz, cz, iz);
      const double s2 = gy * gy + gz * gz;
      const int32_t xa = std::max(cx - 1, 0), xb = std::min(cx + 1, g.nx - 1);
      const uint32_t row = (uint32_t)((iz * g.ny + iy) * g.nx);
      const uint32_t b = G.cell_start[row + xa], e = G.cell_start[row + xb + 1];
      if (b < e) rows.push_back({s2, b, e});
      // pieces: near = {cx, cx -/+ 1 on the query's side}, far = the third cell
      const int32_t fx = left ? cx + 1 : cx - 1;
      const int32_t na = left ? std::max(cx - 1, 0) : cx, nb = left ? cx : std::min(cx + 1, g.nx - 1);
      const uint32_t pb = G.cell_start[row + na], pe = G.cell_start[row + nb + 1];
      if (pb < pe) pieces.push_back({s2, pb, pe});
      if (fx >= 0 && fx < g.nx) {
        const uint32_t fb = G.cell_start[row + fx], fe = G.cell_start[row + fx + 1];
        const double gx = gap(q.x, g.ox, cx, fx);
        if (fb < fe) pieces.push_back({s2 + gx * gx, fb, fe});
      }
    }
    auto walk = [&](std::vector<Piece> list, bool sorted, uint32_t& batches, uint32_t& rejects, uint32_t& cands) {
      if (sorted) std::stable_sort(list.begin(), list.end(), [](const Piece& a, const Piece& b) { return a.s2 < b.s2; });
      std::vector<double> best;  // the k smallest so far
      batches = rejects = cands = 0;
      for (const Piece& P : list) {
        const double kth = best.size() >= k ? best[k - 1] : 1e300;
        const double bound = std::min(kth, r2);
        if (P.s2 > bound) {
          if (sorted) break;
          rejects++;
          continue;
        }
        for (uint32_t p = P.b; p < P.e; p += 4) {
          batches++;
          for (uint32_t t = p; t < std::min(p + 4, P.e); t++) {
            const double dx = q.x - G.sp[t].x, dy = q.y - G.sp[t].y, dz = q.z - G.sp[t].z;
            best.push_back(dx * dx + dy * dy + dz * dz);
            cands++;
          }
          std::sort(best.begin(), best.end());
          if (best.size() > k) best.resize(k);
        }
      }
    };
    uint32_t b, r, c;
    walk(rows, false, b, r, c);
    out[6 * qi] = b, out[6 * qi + 1] = r, out[6 * qi + 4] = c;
    walk(rows, true, b, r, c);
    out[6 * qi + 2] = b;
    walk(pieces, true, b, r, c);
    out[6 * qi + 3] = b, out[6 * qi + 5] = c;
  }
}

double hostcheck_fit_plane(const double* pts, uint64_t k, double out[4]) {
  Vec3 P[kMaxK];
  for (uint64_t i = 0; i < k; i++) P[i] = v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
  Vec3 n;
  double d;
  const double avg = k <= 5 ? fit_plane<5>(P, (int)k, n, d) : fit_plane<8>(P, (int)k, n, d);
  out[0] = n.x, out[1] = n.y, out[2] = n.z, out[3] = d;
  return avg;
}
void hostcheck_fit_line(const double* pts, uint64_t k, double out[6]) {
  Vec3 P[kMaxK];
  for (uint64_t i = 0; i < k; i++) P[i] = v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
  Vec3 a, b;
  if (k <= 5) fit_line<5>(P, (int)k, a, b); else fit_line<8>(P, (int)k, a, b);
  out[0] = a.x, out[1] = a.y, out[2] = a.z, out[3] = b.x, out[4] = b.y, out[5] = b.z;
}

struct Slot {
  bool valid;
  Vec3 p;
  double prim[6];
  uint32_t nearest;
};

extern "C++" {
template <int KM>
static uint32_t associate_t(const double* src, uint32_t n_src, const double* tgt, const HostGrid& G, const double est[7],
                          bool is_plane, const loamx_reg_params* prm, std::vector<Slot>& slots) {
  const int k = (int)(is_plane ? prm->num_plane_neighbors : prm->num_edge_neighbors);
  const double maxd = is_plane ? prm->max_plane_neighbor_dist : prm->max_edge_neighbor_dist;
  const int minfit = (int)(is_plane ? prm->min_plane_fit_points : prm->min_line_fit_points);
  slots.assign(n_src, Slot{});
  uint32_t count = 0;
  (void)tgt;
  for (uint32_t i = 0; i < n_src; i++) {
    Slot& s = slots[i];
    s.valid = false;
    s.p = pose_act(est, v3(src[3 * i], src[3 * i + 1], src[3 * i + 2]));
    uint32_t pos[KM];
    const int kept = knn_both<KM>(G, s.p, k, maxd, pos);
    if (kept < minfit) continue;
    Vec3 nb[KM];
    for (int j = 0; j < KM; j++)
      if (j < kept) nb[j] = v3(G.sp[pos[j]].x, G.sp[pos[j]].y, G.sp[pos[j]].z);
    if (is_plane) {
      Vec3 n;
      double d;
      const double avg = fit_plane<KM>(nb, kept, n, d);
      if (avg > prm->max_avg_point_plane_dist) continue;
      s.prim[0] = n.x, s.prim[1] = n.y, s.prim[2] = n.z, s.prim[3] = d;
    } else {
      Vec3 a, b;
      fit_line<KM>(nb, kept, a, b);
      // min_line_condition_number guard is dead code in the reference (condition number == DBL_MAX)
      if (kDblMax < prm->min_line_condition_number) continue;
      s.prim[0] = a.x, s.prim[1] = a.y, s.prim[2] = a.z, s.prim[3] = b.x, s.prim[4] = b.y, s.prim[5] = b.z;
    }
    s.valid = true;
    s.nearest = G.sp[pos[0]].orig;
    count++;
  }
  return count;
}

static uint32_t associate(const double* src, uint32_t n_src, const double* tgt, const HostGrid& G, const double est[7],
                          bool is_plane, const loamx_reg_params* prm, std::vector<Slot>& slots) {
  const uint64_t k = is_plane ? prm->num_plane_neighbors : prm->num_edge_neighbors;
  return k <= 5 ? associate_t<5>(src, n_src, tgt, G, est, is_plane, prm, slots)
                : associate_t<8>(src, n_src, tgt, G, est, is_plane, prm, slots);
}
}  // extern "C++"

int hostcheck_associate(const double* src, uint64_t n_src, const double* tgt, uint64_t n_tgt, const double est[7],
                        int is_plane, const loamx_reg_params* prm, uint8_t* valid, uint64_t* nearest, double* moved,
                        double* prims) {
  HostGrid G;
  build_grid(tgt, (uint32_t)n_tgt, is_plane ? prm->max_plane_neighbor_dist : prm->max_edge_neighbor_dist, G);
  std::vector<Slot> slots;
  associate(src, (uint32_t)n_src, tgt, G, est, is_plane != 0, prm, slots);
  const int pw = is_plane ? 4 : 6;
  for (uint64_t i = 0; i < n_src; i++) {
    valid[i] = slots[i].valid;
    nearest[i] = slots[i].valid ? slots[i].nearest : 0;
    moved[3 * i] = slots[i].p.x, moved[3 * i + 1] = slots[i].p.y, moved[3 * i + 2] = slots[i].p.z;
    for (int j = 0; j < pw; j++) prims[i * pw + j] = slots[i].valid ? slots[i].prim[j] : 0.0;
  }
  return 0;
}

static void sweep(const std::vector<Slot>& edges, const std::vector<Slot>& planes, const double x[7], double acc[kAccSize]) {
  for (int j = 0; j < kAccSize; j++) acc[j] = 0;
  for (const Slot& s : edges)
    if (s.valid) residual_accumulate(false, s.p, s.prim, x, acc);
  for (const Slot& s : planes)
    if (s.valid) residual_accumulate(true, s.p, s.prim, x, acc);
}

// plane terms of the normal equations at x: per-record accumulation vs the moment matrix
void hostcheck_plane_moments(const double* v, const double* n, const double* d, uint64_t count, const double x[7],
                             double direct[kAccSize], double moments[kAccSize], double bound_inputs[2]) {
  for (int j = 0; j < kAccSize; j++) direct[j] = moments[j] = 0.0;
  double M[kMomSize];
  for (int j = 0; j < kMomSize; j++) M[j] = 0.0;
  double s0max = 0.0, v2max = 0.0;
  for (uint64_t i = 0; i < count; i++) {
    const Vec3 vv = v3(v[3 * i], v[3 * i + 1], v[3 * i + 2]), nn = v3(n[3 * i], n[3 * i + 1], n[3 * i + 2]);
    const double prim[6] = {nn.x, nn.y, nn.z, d[i], 0, 0};
    residual_accumulate(true, vv, prim, x, direct);
    double c[kMomDim];
    plane_coeffs(vv, nn, d[i], c);
    for (int a = 0; a < kMomDim; a++)
      for (int b = 0; b < kMomDim; b++) M[a * kMomStride + b] += c[a] * c[b];
    s0max = std::max(s0max, fabs(c[0]));
    v2max = std::max(v2max, vdot(vv, vv));
  }
  plane_eval_from_moments(M, x, moments);
  bound_inputs[0] = s0max, bound_inputs[1] = plane_moments_valid_at(s0max, v2max, x) ? 1.0 : 0.0;
}

// the relative validity bound of the first ICF iteration's moments (plane_moments_valid_rel) and the residuals it speaks about:
// out[0] = the bound's verdict (1 / 0), out[1] = max |s_i(r)| over the records with |s_i(r)| <= kMomInlier (what moment_kernel
// hands the bound), out[2] = max |s_i(x)| over the same records (what the bound must keep below the Huber threshold),
// out[3] = largest |c_i . phi(x) - s_i(x)| (the moment form of a residual against residual_accumulate's own point formula)
void hostcheck_moments_rel(const double* v, const double* n, const double* d, uint64_t count, const double x[7], const double r[7],
                           double out[4]) {
  double phi_x[kMomDim], phi_r[kMomDim];
  plane_phi(x, phi_x), plane_phi(r, phi_r);
  double sref_max = 0.0, v2max = 0.0, sx_max = 0.0, form_err = 0.0;
  auto residual = [](Vec3 vv, Vec3 nn, double dd, const double* xx) {
    const Vec3 u = v3(xx[0], xx[1], xx[2]);
    Vec3 uv = vcross(u, vv);
    uv = vadd(uv, uv);
    const Vec3 pp = vadd(vadd(vadd(vv, vscale(xx[3], uv)), vcross(u, uv)), v3(xx[4], xx[5], xx[6]));  // as residual_accumulate
    return vdot(nn, pp) - dd;
  };
  for (uint64_t i = 0; i < count; i++) {
    const Vec3 vv = v3(v[3 * i], v[3 * i + 1], v[3 * i + 2]), nn = v3(n[3 * i], n[3 * i + 1], n[3 * i + 2]);
    double c[kMomDim];
    plane_coeffs(vv, nn, d[i], c);
    double sr = 0.0, sx = 0.0;
    for (int j = 0; j < kMomDim; j++) sr += c[j] * phi_r[j], sx += c[j] * phi_x[j];
    form_err = std::max(form_err, fabs(sx - residual(vv, nn, d[i], x)));
    if (!(fabs(sr) <= kMomInlier)) continue;  // flagged: stays out of the moments
    sref_max = std::max(sref_max, fabs(sr));
    v2max = std::max(v2max, vdot(vv, vv));
    sx_max = std::max(sx_max, fabs(residual(vv, nn, d[i], x)));
  }
  out[0] = plane_moments_valid_rel(sref_max, v2max, x, r) ? 1.0 : 0.0;
  out[1] = sref_max, out[2] = sx_max, out[3] = form_err;
}

int hostcheck_register(const double* src_edge, uint64_t n_se, const double* src_planar, uint64_t n_sp,
                       const double* tgt_edge, uint64_t n_te, const double* tgt_planar, uint64_t n_tp,
                       const double init[7], const loamx_reg_params* prm, loamx_reg_result* out,
                       loamx_iter_info* info) {
  HostGrid GE, GP;
  build_grid(tgt_edge, (uint32_t)n_te, prm->max_edge_neighbor_dist, GE);
  build_grid(tgt_planar, (uint32_t)n_tp, prm->max_plane_neighbor_dist, GP);
  double est[7];
  memcpy(est, init, sizeof(est));
  uint32_t term = LOAMX_MAX_ITER, iters = 0;
  for (uint64_t it = 0; it < prm->max_iterations; it++) {
    std::vector<Slot> edges, planes;
    const uint32_t ne = associate(src_edge, (uint32_t)n_se, tgt_edge, GE, est, false, prm, edges);
    const uint32_t np = associate(src_planar, (uint32_t)n_sp, tgt_planar, GP, est, true, prm, planes);
    if ((uint64_t)ne + np < prm->min_associations) {
      term = LOAMX_INSUFFICIENT_ASSOCIATIONS;
      break;
    }
    LmState st;
    lm_init(st);
    // as the kernels: from the second ICF iteration on the plane records enter through their moment matrix
    // while that is provably exact (Huber inactive at the candidate), else they are streamed like the edges
    const bool use_moments = it >= 1 && !getenv("HOSTCHECK_NO_MOMENTS");
    std::vector<double> M(kMomSize + 2, 0.0);
    std::vector<Slot> flagged;  // plane records that start far from their plane: evaluated one by one
    if (use_moments) {
      for (const Slot& sl : planes) {
        if (!sl.valid) continue;
        double c[kMomDim];
        plane_coeffs(sl.p, v3(sl.prim[0], sl.prim[1], sl.prim[2]), sl.prim[3], c);
        if (!(fabs(c[0]) <= kMomInlier)) {
          flagged.push_back(sl);
          continue;
        }
        for (int a = 0; a < kMomDim; a++)
          for (int b = 0; b < kMomDim; b++) M[a * kMomStride + b] += c[a] * c[b];
        M[kMomSize] = std::max(M[kMomSize], fabs(c[0]));
        M[kMomSize + 1] = std::max(M[kMomSize + 1], vdot(sl.p, sl.p));
      }
    }
    double acc[kAccSize];
    bool first = true;
    while (st.active) {
      const bool stream = !use_moments || !plane_moments_valid_at(M[kMomSize], M[kMomSize + 1], st.xeval);
      sweep(edges, stream ? planes : flagged, st.xeval, acc);
      if (!stream) plane_eval_from_moments(M.data(), st.xeval, acc);
      else g_plane_streams++;
      lm_consume(st, acc, first);
      first = false;
    }
    if (info) {
      memcpy(info[it].target_T_source_init, est, sizeof(est));
      memcpy(info[it].estimate_update, st.x_user, sizeof(est));
      info[it].n_edge_associations = ne, info[it].n_plane_associations = np;
    }
    iters = (uint32_t)it + 1;
    if (outer_update(est, st.x_user, prm->rotation_convergence_thresh, prm->position_convergence_thresh)) {
      term = LOAMX_CONVERGED;
      break;
    }
  }
  memcpy(out->pose, est, sizeof(est));
  out->termination = term, out->iterations = iters;
  return 0;
}

void hostcheck_synth_scan(uint64_t seed, uint64_t pair, uint32_t which, uint32_t H, uint32_t W, double sigma, double* xyz) {
  const loamx_synth::Pose7 pose = loamx_synth::pair_pose(seed, pair);
  for (uint32_t l = 0; l < H; l++)
    for (uint32_t c = 0; c < W; c++) loamx_synth::scan_point(seed, pair, which, pose, l, c, H, W, sigma, xyz + 3 * ((size_t)l * W + c));
}
void hostcheck_synth_pose(uint64_t seed, uint64_t pair, double out[7]) {
  const loamx_synth::Pose7 p = loamx_synth::pair_pose(seed, pair);
  for (int i = 0; i < 4; i++) out[i] = p.q[i];
  for (int i = 0; i < 3; i++) out[4 + i] = p.t[i];
}

}  // extern "C"

// ---- row a7: the tie order. stl_sort (extract_math.h) replays libstdc++'s std::sort; here it runs next to the real one.
extern "C" {
// order[] <- the indices 0..n-1 as stl_sort leaves them for the comparator c[a] < c[b]
uint64_t hostcheck_heap_sorts() { return g_heap_sorts; }
void hostcheck_stl_sort(const double* c, uint64_t n, uint32_t* order) {
  std::vector<uint16_t> idx(n);
  for (uint64_t i = 0; i < n; i++) idx[i] = (uint16_t)i;
  stl_sort(idx.data(), 0, (int)n, [&](uint16_t a, uint16_t b) { return c[a] < c[b]; });
  for (uint64_t i = 0; i < n; i++) order[i] = idx[i];
}
// the same with the real std::sort on the reference's element type (features.h:79-91: {size_t index; double curvature},
// comparator lhs.curvature < rhs.curvature)
void hostcheck_std_sort(const double* c, uint64_t n, uint32_t* order) {
  struct PC {
    size_t index;
    double curvature;
  };
  std::vector<PC> v(n);
  for (uint64_t i = 0; i < n; i++) v[i] = PC{(size_t)i, c[i]};
  std::sort(v.begin(), v.end(), [](const PC& l, const PC& r) { return l.curvature < r.curvature; });
  for (uint64_t i = 0; i < n; i++) order[i] = (uint32_t)v[i].index;
}
// McIlroy's adversary ("A killer adversary for quicksort", 1999) played against the real std::sort: returns an input
// on which the introsort's depth limit runs out, so that the heap-sort branch is part of the comparison above
void hostcheck_killer_input(uint64_t n, double* out) {
  std::vector<int> val(n), ptr(n);
  const int gas = (int)n - 1;
  int nsolid = 0, candidate = 0;
  for (uint64_t i = 0; i < n; i++) ptr[i] = (int)i, val[i] = gas;
  std::sort(ptr.begin(), ptr.end(), [&](int x, int y) {
    if (val[x] == gas && val[y] == gas) {
      if (x == candidate) val[x] = nsolid++;
      else val[y] = nsolid++;
    }
    if (val[x] == gas) candidate = x;
    else if (val[y] == gas) candidate = y;
    return val[x] < val[y];
  });
  for (uint64_t i = 0; i < n; i++) out[i] = (double)val[i];
}
}  // extern "C"
