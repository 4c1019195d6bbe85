// loam_oracle.cpp — CPU restatement of the reference hot path. TEST INFRASTRUCTURE, NOT PRODUCT
// (see loam_oracle.h for the scope statement and the parity status).
//
// Every function cites the reference file:line it follows (paths relative to /root/reference).
// Third-party semantics that are not vendored with the reference are restated from their published
// algorithms and marked [RECALLED]:
//   Eigen 3   : quaternion algebra, SelfAdjointEigenSolver<3x3>, ColPivHouseholderQR, HouseholderQR
//   nanoflann : v1.5.5 KDTreeSingleIndexAdaptor (middle split, leaf 20), KNNResultSet
//   Ceres     : 2.2.0 TrustRegionMinimizer + LevenbergMarquardtStrategy + DENSE_QR + HuberLoss +
//               QuaternionManifold (applied to Eigen-ordered storage)
// Build: g++ -std=c++17 -O3 -DNDEBUG -ffp-contract=off (no -march=native), see oracle/Makefile.
#include "loam_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <utility>
#include <vector>

namespace {

struct V3 {
  double x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }

// Eigen::Quaterniond, coefficient storage (x, y, z, w)
struct Quat {
  double x, y, z, w;
};
struct Pose {
  Quat q;
  V3 t;
};

// [RECALLED] Eigen quat_product (generic path)
inline Quat qmul(const Quat& a, const Quat& b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
// [RECALLED] Eigen QuaternionBase::_transformVector: v + w*(2 u x v) + u x (2 u x v)
inline V3 qrot(const Quat& q, V3 v) {
  V3 u{q.x, q.y, q.z};
  V3 uv = cross(u, v);
  uv = uv + uv;
  return v + q.w * uv + cross(u, uv);
}
// [RECALLED] Eigen QuaternionBase::inverse: conjugate / squaredNorm (zero quaternion if norm 0)
inline Quat qinv(const Quat& q) {
  double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  if (n2 > 0.0) return {-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2};
  return {0, 0, 0, 0};
}
// [RECALLED] Eigen angularDistance: d = a * conj(b); 2*atan2(|d.vec|, |d.w|)
inline double qangdist(const Quat& a, const Quat& b) {
  Quat d = qmul(a, Quat{-b.x, -b.y, -b.z, b.w});
  return 2.0 * std::atan2(std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z), std::fabs(d.w));
}

inline Pose from7(const double* p) { return {{p[0], p[1], p[2], p[3]}, {p[4], p[5], p[6]}}; }
inline void to7(const Pose& P, double* p) {
  p[0] = P.q.x, p[1] = P.q.y, p[2] = P.q.z, p[3] = P.q.w, p[4] = P.t.x, p[5] = P.t.y, p[6] = P.t.z;
}
// loam/src/geometry.cpp:16-18
inline Pose compose(const Pose& a, const Pose& b) { return {qmul(a.q, b.q), a.t + qrot(a.q, b.t)}; }
// loam/src/geometry.cpp:10-13
inline Pose inverse(const Pose& a) {
  Quat qi = qinv(a.q);
  return {qi, qrot(qi, V3{-a.t.x, -a.t.y, -a.t.z})};
}
// loam/src/geometry.cpp:21
inline V3 act(const Pose& a, V3 p) { return qrot(a.q, p) + a.t; }

/* ------------------------------------------------------------------------------------------------
 * Feature extraction
 * ---------------------------------------------------------------------------------------------- */
struct PointCurvature {  // loam/include/loam/features.h:79-88
  size_t index;
  double curvature;
};
inline bool curvatureComparator(const PointCurvature& l, const PointCurvature& r) {  // features.h:91
  return l.curvature < r.curvature;
}
inline const double* P(const double* xyz, size_t i) { return xyz + 3 * i; }
// loam/include/loam/common.h:81-86
inline double pointRange(const double* p) { return std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]); }

// loam/include/loam/features-inl.h:53-87
std::vector<PointCurvature> computeCurvature(const double* xyz, size_t H, size_t W, const oracle_fe_params& prm) {
  std::vector<PointCurvature> curv;
  curv.reserve(H * W);
  const size_t np = prm.neighbor_points;
  for (size_t line = 0; line < H; line++) {
    for (size_t col = 0; col < W; col++) {
      const size_t idx = line * W + col;
      if (col < np || col >= W - np) {  // :66-68 (size_t arithmetic, as in the reference)
        curv.push_back({idx, -1});
      } else {
        double dx = -(2.0 * np) * P(xyz, idx)[0];  // :73-75
        double dy = -(2.0 * np) * P(xyz, idx)[1];
        double dz = -(2.0 * np) * P(xyz, idx)[2];
        for (size_t n = 1; n <= np; n++) {  // :77-81
          dx = dx + P(xyz, idx - n)[0] + P(xyz, idx + n)[0];
          dy = dy + P(xyz, idx - n)[1] + P(xyz, idx + n)[1];
          dz = dz + P(xyz, idx - n)[2] + P(xyz, idx + n)[2];
        }
        curv.push_back({idx, dx * dx + dy * dy + dz * dz});  // :82
      }
    }
  }
  return curv;
}

// loam/include/loam/features-inl.h:90-124 with the four mark* helpers of loam/src/features.cpp:20-68
std::vector<uint8_t> computeValidPoints(const double* xyz, size_t H, size_t W, double min_range, double max_range,
                                        const oracle_fe_params& prm) {
  std::vector<uint8_t> mask(H * W, 1);
  const size_t np = prm.neighbor_points;
  for (size_t line = 0; line < H; line++) {
    for (size_t col = 0; col < W; col++) {
      const size_t idx = line * W + col;
      // CHECK 1 (features.cpp:20-27)
      if (col < np || col >= W - np) {
        mask[idx] = 0;
        continue;
      }
      const double point_range = pointRange(P(xyz, idx));
      const double next_point_range = pointRange(P(xyz, idx + 1));
      const double prev_point_range = pointRange(P(xyz, idx - 1));
      // CHECK 2 (features.cpp:30-41)
      if (point_range < min_range || point_range > max_range) {
        mask[idx] = 0;
        for (size_t n = 1; n <= np; n++) {
          mask[idx + n] = 0;
          mask[idx - n] = 0;
        }
        continue;
      }
      // CHECK 3 (features.cpp:44-54)
      if (next_point_range - point_range > prm.occlusion_thresh) {
        for (size_t n = 1; n <= np; n++) mask[idx + n] = 0;
        continue;
      } else if (point_range - next_point_range > prm.occlusion_thresh) {
        for (size_t n = 0; n < np; n++) mask[idx - n] = 0;
        continue;
      }
      // CHECK 4 (features.cpp:57-68)
      const double diff_next = std::abs(prev_point_range - point_range);
      const double diff_prev = std::abs(next_point_range - point_range);
      if (diff_next > prm.parallel_thresh * point_range && diff_prev > prm.parallel_thresh * point_range) {
        mask[idx] = 0;
      }
    }
  }
  return mask;
}

// loam/include/loam/features-inl.h:11-50, :137-180. stable=false: std::sort exactly as the reference
// (libstdc++ introsort, tie order implementation-defined); stable=true: std::stable_sort.
void extractFeatures(const double* xyz, size_t H, size_t W, double min_range, double max_range,
                     const oracle_fe_params& prm, bool stable, std::vector<uint32_t>& edge,
                     std::vector<uint32_t>& planar, size_t* n_ties) {
  const size_t S = prm.number_sectors;
  const size_t np = prm.neighbor_points;
  const size_t points_per_sector = W / S;  // :18
  std::vector<PointCurvature> curvature = computeCurvature(xyz, H, W, prm);
  std::vector<uint8_t> valid_mask = computeValidPoints(xyz, H, W, min_range, max_range, prm);
  size_t ties = 0;
  for (size_t line = 0; line < H; line++) {
    for (size_t sector = 0; sector < S; sector++) {
      const size_t start = line * W + sector * points_per_sector;                          // :31
      const size_t end = (sector == S - 1) ? ((line + 1) * W) : start + points_per_sector;  // :33-35
      if (stable)
        std::stable_sort(curvature.begin() + start, curvature.begin() + end, curvatureComparator);
      else
        std::sort(curvature.begin() + start, curvature.begin() + end, curvatureComparator);  // :38
      if (n_ties) {
        for (size_t k = start + 1; k < end; k++) {
          const PointCurvature &a = curvature[k - 1], &b = curvature[k];
          if (a.curvature == b.curvature && valid_mask[a.index] && valid_mask[b.index] &&
              (a.curvature > prm.edge_feat_threshold || a.curvature < prm.planar_feat_threshold))
            ties++;
        }
      }
      // extractSectorEdgeFeatures :137-157
      size_t n_e = 0;
      for (size_t kp1 = end; kp1 > start; kp1--) {
        const PointCurvature curv = curvature[kp1 - 1];
        if (valid_mask[curv.index] && curv.curvature > prm.edge_feat_threshold) {
          edge.push_back((uint32_t)curv.index);
          for (size_t n = 0; n < np; n++) {  // :148-151 (strict <)
            valid_mask[curv.index + n] = 0;
            valid_mask[curv.index - n] = 0;
          }
          n_e++;
        }
        if (n_e > prm.max_edge_feats_per_sector) break;  // :155 (off by one kept)
      }
      // extractSectorPlanarFeatures :160-180
      size_t n_p = 0;
      for (size_t k = start; k < end; k++) {
        const PointCurvature curv = curvature[k];
        if (valid_mask[curv.index] && curv.curvature < prm.planar_feat_threshold) {
          planar.push_back((uint32_t)curv.index);
          for (size_t n = 0; n < np; n++) {
            valid_mask[curv.index + n] = 0;
            valid_mask[curv.index - n] = 0;
          }
          n_p++;
        }
        if (n_p > prm.max_planar_feats_per_sector) break;  // :177
      }
    }
  }
  if (n_ties) *n_ties = ties;
}

/* ------------------------------------------------------------------------------------------------
 * Small dense linear algebra ([RECALLED] Eigen algorithms)
 * ---------------------------------------------------------------------------------------------- */

// Symmetric 3x3 eigen-decomposition, eigenvalues ascending, eigenvectors in columns of V.
// Eigen's SelfAdjointEigenSolver<Matrix3d>::compute is tridiagonalisation + implicit QL; any
// accurate symmetric solver gives the same eigenvector to rounding. Cyclic Jacobi here.
void symeig3(const double A_in[3][3], double evals[3], double V[3][3]) {
  double A[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      A[i][j] = A_in[i][j];
      V[i][j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    if (off == 0.0) break;
    for (int p = 0; p < 2; p++) {
      for (int q = p + 1; q < 3; q++) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) {  // A <- A * J
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {  // A <- J^T * A
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
    }
  }
  // [RECALLED] Eigen sorts the (eigenvalue, eigenvector) pairs ascending with a selection sort that takes the FIRST
  // minimum of the remaining ones (SelfAdjointEigenSolver.h, computeFromTridiagonal_impl: diag.segment(i, n - i)
  // .minCoeff(&k); swap if k > 0). It decides which eigenvector "the largest" is when eigenvalues are EQUAL — a
  // covariance that is exactly diagonal with two equal entries, i.e. neighbours on a lattice: (0, s, s) -> z,
  // (s, s, 0) -> x, (s, 0, s) -> z, (s, s, s) -> z (for a diagonal input Eigen's tridiagonalisation and QL iteration
  // leave values and vectors in place, as the Jacobi sweeps above do).
  int order[3] = {0, 1, 2};
  for (int i = 0; i < 2; i++) {
    int k = 0;
    for (int j = 1; j < 3 - i; j++)
      if (A[order[i + j]][order[i + j]] < A[order[i + k]][order[i + k]]) k = j;
    if (k > 0) std::swap(order[i], order[i + k]);
  }
  double Vs[3][3];
  for (int k = 0; k < 3; k++) {
    evals[k] = A[order[k]][order[k]];
    for (int i = 0; i < 3; i++) Vs[i][k] = V[i][order[k]];
  }
  std::memcpy(V, Vs, sizeof(Vs));
}

// loam/src/geometry.cpp:42-59. Returns the (dead, always DBL_MAX) condition number.
double fitLine(const std::vector<V3>& pts, V3& a, V3& b) {
  const size_t K = pts.size();
  V3 center{0, 0, 0};
  for (const V3& p : pts) center = center + p;
  center = (1.0 / (double)K) * center;  // colwise().mean()
  double C[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (const V3& p : pts) {
    const double d[3] = {p.x - center.x, p.y - center.y, p.z - center.z};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) C[i][j] += d[i] * d[j];
  }
  double ev[3], V[3][3];
  symeig3(C, ev, V);
  const V3 dir{V[0][2], V[1][2], V[2][2]};  // eigenvector of the largest eigenvalue (:51)
  a = center + 0.1 * dir;                   // :53
  b = center - 0.1 * dir;
  return std::numeric_limits<double>::max();  // :55-56: the division result is discarded
}

// [RECALLED] Eigen ColPivHouseholderQR<MatrixXd>(K x 3).solve(ones) — loam/src/geometry.cpp:62-73
double fitPlane(const std::vector<V3>& pts, V3& normal, double& d_out) {
  const int rows = (int)pts.size(), cols = 3;
  const int size = std::min(rows, cols);
  std::vector<double> qr(rows * cols);  // column-major
  auto Q = [&](int r, int c) -> double& { return qr[c * rows + r]; };
  for (int r = 0; r < rows; r++) Q(r, 0) = pts[r].x, Q(r, 1) = pts[r].y, Q(r, 2) = pts[r].z;
  double hCoeffs[3] = {0, 0, 0};
  int transp[3] = {0, 1, 2};
  double normsUpdated[3], normsDirect[3];
  const double eps = std::numeric_limits<double>::epsilon();
  for (int k = 0; k < cols; k++) {
    double s = 0;
    for (int r = 0; r < rows; r++) s += Q(r, k) * Q(r, k);
    normsDirect[k] = normsUpdated[k] = std::sqrt(s);
  }
  const double maxn = *std::max_element(normsUpdated, normsUpdated + cols);
  const double threshold_helper = (maxn * eps) * (maxn * eps) / (double)rows;
  const double norm_downdate_threshold = std::sqrt(eps);
  int nonzero_pivots = size;
  for (int k = 0; k < size; ++k) {
    int biggest = k;
    for (int j = k + 1; j < cols; j++)
      if (normsUpdated[j] > normsUpdated[biggest]) biggest = j;
    const double biggest_sq = normsUpdated[biggest] * normsUpdated[biggest];
    if (nonzero_pivots == size && biggest_sq < threshold_helper * (double)(rows - k)) nonzero_pivots = k;
    transp[k] = biggest;
    if (k != biggest) {
      for (int r = 0; r < rows; r++) std::swap(Q(r, k), Q(r, biggest));
      std::swap(normsUpdated[k], normsUpdated[biggest]);
      std::swap(normsDirect[k], normsDirect[biggest]);
    }
    // makeHouseholderInPlace on col(k).tail(rows-k)
    double tailSq = 0;
    for (int r = k + 1; r < rows; r++) tailSq += Q(r, k) * Q(r, k);
    const double c0 = Q(k, k);
    double beta, tau;
    if (tailSq <= std::numeric_limits<double>::min()) {
      tau = 0;
      beta = c0;
      for (int r = k + 1; r < rows; r++) Q(r, k) = 0;
    } else {
      beta = std::sqrt(c0 * c0 + tailSq);
      if (c0 >= 0) beta = -beta;
      for (int r = k + 1; r < rows; r++) Q(r, k) = Q(r, k) / (c0 - beta);
      tau = (beta - c0) / beta;
    }
    hCoeffs[k] = tau;
    Q(k, k) = beta;
    // apply to the remaining columns
    if (tau != 0) {
      for (int j = k + 1; j < cols; j++) {
        double tmp = Q(k, j);
        for (int r = k + 1; r < rows; r++) tmp += Q(r, k) * Q(r, j);
        Q(k, j) -= tau * tmp;
        for (int r = k + 1; r < rows; r++) Q(r, j) -= tau * Q(r, k) * tmp;
      }
    }
    // norm down-dating
    for (int j = k + 1; j < cols; ++j) {
      if (normsUpdated[j] != 0) {
        double temp = std::fabs(Q(k, j)) / normsUpdated[j];
        temp = (1.0 + temp) * (1.0 - temp);
        temp = temp < 0 ? 0 : temp;
        const double ratio = normsUpdated[j] / normsDirect[j];
        const double temp2 = temp * ratio * ratio;
        if (temp2 <= norm_downdate_threshold) {
          double s = 0;
          for (int r = k + 1; r < rows; r++) s += Q(r, j) * Q(r, j);
          normsDirect[j] = std::sqrt(s);
          normsUpdated[j] = normsDirect[j];
        } else {
          normsUpdated[j] *= std::sqrt(temp);
        }
      }
    }
  }
  int perm[3] = {0, 1, 2};
  for (int k = 0; k < size; k++) std::swap(perm[k], perm[transp[k]]);
  // solve: c = Q^T * ones (first nonzero_pivots reflectors), back-substitute, un-permute
  std::vector<double> c(rows, 1.0);
  for (int k = 0; k < nonzero_pivots; k++) {
    const double tau = hCoeffs[k];
    if (rows - k == 1) {
      c[k] *= (1.0 - tau);
    } else if (tau != 0) {
      double tmp = c[k];
      for (int r = k + 1; r < rows; r++) tmp += Q(r, k) * c[r];
      c[k] -= tau * tmp;
      for (int r = k + 1; r < rows; r++) c[r] -= tau * Q(r, k) * tmp;
    }
  }
  for (int i = nonzero_pivots - 1; i >= 0; i--) {
    double s = c[i];
    for (int j = i + 1; j < nonzero_pivots; j++) s -= Q(i, j) * c[j];
    c[i] = s / Q(i, i);
  }
  double abc[3] = {0, 0, 0};
  for (int i = 0; i < nonzero_pivots; i++) abc[perm[i]] = c[i];
  const double n = std::sqrt(abc[0] * abc[0] + abc[1] * abc[1] + abc[2] * abc[2]);
  normal = {abc[0] / n, abc[1] / n, abc[2] / n};  // geometry.cpp:69
  d_out = 1.0 / n;
  double sum = 0;  // geometry.cpp:71 — signed mean
  for (const V3& p : pts) sum += (p.x * normal.x + p.y * normal.y + p.z * normal.z) - d_out;
  return sum / (double)rows;
}

// loam/include/loam/geometry-inl.h:21-27
inline double pointToLineDistance(V3 p, V3 a, V3 b) { return norm(cross(p - a, p - b)) / norm(a - b); }
// loam/include/loam/geometry-inl.h:30-33
inline double pointToPlaneDistance(V3 p, V3 n, double d) { return std::fabs(dot(n, p) - d); }

/* ------------------------------------------------------------------------------------------------
 * [RECALLED] nanoflann v1.5.5 KDTreeSingleIndexAdaptor<L2_Simple, ..., 3>, leaf_max_size = 20
 * (loam/include/loam/kdtree.h:24-41, loam/include/loam/registration-inl.h:20-23)
 * ---------------------------------------------------------------------------------------------- */
struct KDTree {
  struct Node {
    // leaf: child1 == child2 == -1, [left, right) into vAcc
    int child1 = -1, child2 = -1;
    size_t left = 0, right = 0;
    int divfeat = 0;
    double divlow = 0, divhigh = 0;
  };
  struct Interval {
    double low, high;
  };
  const double* pts = nullptr;
  size_t n = 0;
  std::vector<size_t> vAcc;
  std::vector<Node> nodes;
  Interval root_bbox[3];
  int root = -1;
  static constexpr size_t kLeaf = 20;

  double at(size_t i, int d) const { return pts[3 * i + d]; }

  void build(const double* p, size_t count) {
    pts = p;
    n = count;
    vAcc.resize(n);
    for (size_t i = 0; i < n; i++) vAcc[i] = i;
    nodes.clear();
    if (n == 0) return;
    for (int d = 0; d < 3; d++) {
      root_bbox[d].low = root_bbox[d].high = at(0, d);
      for (size_t i = 1; i < n; i++) {
        root_bbox[d].low = std::min(root_bbox[d].low, at(i, d));
        root_bbox[d].high = std::max(root_bbox[d].high, at(i, d));
      }
    }
    Interval bbox[3] = {root_bbox[0], root_bbox[1], root_bbox[2]};
    root = divideTree(0, n, bbox);
  }

  int divideTree(size_t left, size_t right, Interval bbox[3]) {
    const int id = (int)nodes.size();
    nodes.emplace_back();
    if ((right - left) <= kLeaf) {
      nodes[id].left = left;
      nodes[id].right = right;
      for (int d = 0; d < 3; d++) {
        bbox[d].low = bbox[d].high = at(vAcc[left], d);
        for (size_t k = left + 1; k < right; k++) {
          bbox[d].low = std::min(bbox[d].low, at(vAcc[k], d));
          bbox[d].high = std::max(bbox[d].high, at(vAcc[k], d));
        }
      }
    } else {
      size_t idx;
      int cutfeat;
      double cutval;
      middleSplit(left, right - left, idx, cutfeat, cutval, bbox);
      nodes[id].divfeat = cutfeat;
      Interval lb[3] = {bbox[0], bbox[1], bbox[2]};
      lb[cutfeat].high = cutval;
      const int c1 = divideTree(left, left + idx, lb);
      Interval rb[3] = {bbox[0], bbox[1], bbox[2]};
      rb[cutfeat].low = cutval;
      const int c2 = divideTree(left + idx, right, rb);
      nodes[id].child1 = c1;
      nodes[id].child2 = c2;
      nodes[id].divlow = lb[cutfeat].high;
      nodes[id].divhigh = rb[cutfeat].low;
      for (int d = 0; d < 3; d++) {
        bbox[d].low = std::min(lb[d].low, rb[d].low);
        bbox[d].high = std::max(lb[d].high, rb[d].high);
      }
    }
    return id;
  }

  void middleSplit(size_t ind, size_t count, size_t& index, int& cutfeat, double& cutval, const Interval bbox[3]) {
    const double EPS = 0.00001;
    double max_span = bbox[0].high - bbox[0].low;
    for (int i = 1; i < 3; i++) max_span = std::max(max_span, bbox[i].high - bbox[i].low);
    double max_spread = -1;
    cutfeat = 0;
    double min_elem = 0, max_elem = 0;
    for (int i = 0; i < 3; i++) {
      const double span = bbox[i].high - bbox[i].low;
      if (span > (1 - EPS) * max_span) {
        double mn = at(vAcc[ind], i), mx = mn;
        for (size_t k = 1; k < count; k++) {
          const double v = at(vAcc[ind + k], i);
          mn = std::min(mn, v);
          mx = std::max(mx, v);
        }
        const double spread = mx - mn;
        if (spread > max_spread) {
          cutfeat = i;
          max_spread = spread;
          min_elem = mn;
          max_elem = mx;
        }
      }
    }
    const double split_val = (bbox[cutfeat].low + bbox[cutfeat].high) / 2;
    if (split_val < min_elem)
      cutval = min_elem;
    else if (split_val > max_elem)
      cutval = max_elem;
    else
      cutval = split_val;
    size_t lim1, lim2;
    planeSplit(ind, count, cutfeat, cutval, lim1, lim2);
    if (lim1 > count / 2)
      index = lim1;
    else if (lim2 < count / 2)
      index = lim2;
    else
      index = count / 2;
  }

  void planeSplit(size_t ind, size_t count, int cutfeat, double cutval, size_t& lim1, size_t& lim2) {
    size_t left = 0, right = count - 1;
    for (;;) {
      while (left <= right && at(vAcc[ind + left], cutfeat) < cutval) ++left;
      while (right && left <= right && at(vAcc[ind + right], cutfeat) >= cutval) --right;
      if (left > right || !right) break;
      std::swap(vAcc[ind + left], vAcc[ind + right]);
      ++left;
      --right;
    }
    lim1 = left;
    right = count - 1;
    for (;;) {
      while (left <= right && at(vAcc[ind + left], cutfeat) <= cutval) ++left;
      while (right && left <= right && at(vAcc[ind + right], cutfeat) > cutval) --right;
      if (left > right || !right) break;
      std::swap(vAcc[ind + left], vAcc[ind + right]);
      ++left;
      --right;
    }
    lim2 = left;
  }

  // KNNResultSet<double>
  struct ResultSet {
    size_t* indices;
    double* dists;
    size_t capacity, count = 0;
    ResultSet(size_t* i, double* d, size_t cap) : indices(i), dists(d), capacity(cap) {
      if (capacity) dists[capacity - 1] = std::numeric_limits<double>::max();
    }
    double worstDist() const { return dists[capacity - 1]; }
    void addPoint(double dist, size_t index) {
      size_t i;
      for (i = count; i > 0; --i) {
        if (dists[i - 1] > dist) {  // equal distances: newcomer goes after the existing entry
          if (i < capacity) {
            dists[i] = dists[i - 1];
            indices[i] = indices[i - 1];
          }
        } else
          break;
      }
      if (i < capacity) {
        dists[i] = dist;
        indices[i] = index;
      }
      if (count < capacity) count++;
    }
  };

  void searchLevel(ResultSet& rs, const double* vec, int node, double mindistsq, double dists[3]) const {
    const Node& nd = nodes[node];
    if (nd.child1 == -1 && nd.child2 == -1) {
      double worst = rs.worstDist();
      for (size_t i = nd.left; i < nd.right; ++i) {
        const size_t index = vAcc[i];
        double dist = 0;  // L2_Simple_Adaptor::evalMetric
        for (int d = 0; d < 3; d++) {
          const double diff = vec[d] - at(index, d);
          dist += diff * diff;
        }
        if (dist < worst) {
          rs.addPoint(dist, index);
          worst = rs.worstDist();
        }
      }
      return;
    }
    const int idx = nd.divfeat;
    const double val = vec[idx];
    const double diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
    int best, other;
    double cut_dist;
    if ((diff1 + diff2) < 0) {
      best = nd.child1;
      other = nd.child2;
      cut_dist = (val - nd.divhigh) * (val - nd.divhigh);
    } else {
      best = nd.child2;
      other = nd.child1;
      cut_dist = (val - nd.divlow) * (val - nd.divlow);
    }
    searchLevel(rs, vec, best, mindistsq, dists);
    const double dst = dists[idx];
    mindistsq = mindistsq + cut_dist - dst;
    dists[idx] = cut_dist;
    if (mindistsq <= rs.worstDist()) searchLevel(rs, vec, other, mindistsq, dists);
    dists[idx] = dst;
  }

  // findNeighbors(result_set, query): false (nothing found) on an empty dataset
  size_t findNeighbors(const double* vec, size_t k, size_t* indices, double* dists_out) const {
    ResultSet rs(indices, dists_out, k);
    if (n == 0 || k == 0) return 0;
    double dists[3] = {0, 0, 0};
    double distsq = 0;
    for (int i = 0; i < 3; i++) {
      if (vec[i] < root_bbox[i].low) {
        dists[i] = (vec[i] - root_bbox[i].low) * (vec[i] - root_bbox[i].low);
        distsq += dists[i];
      }
      if (vec[i] > root_bbox[i].high) {
        dists[i] = (vec[i] - root_bbox[i].high) * (vec[i] - root_bbox[i].high);
        distsq += dists[i];
      }
    }
    searchLevel(rs, vec, root, distsq, dists);
    return rs.count;
  }
};

// loam/src/kdtree.cpp:10-28
std::vector<size_t> knnSearch(const KDTree& tree, V3 query, size_t k, double max_dist) {
  std::vector<size_t> knn_indices(k);
  std::vector<double> knn_distances_sq(k);
  const double q[3] = {query.x, query.y, query.z};
  const size_t found = tree.findNeighbors(q, k, knn_indices.data(), knn_distances_sq.data());
  std::vector<size_t> result;
  for (size_t i = 0; i < found; i++) {
    if (max_dist <= 0 || std::sqrt(knn_distances_sq[i]) < max_dist) result.push_back(knn_indices[i]);  // :25
  }
  return result;
}

/* ------------------------------------------------------------------------------------------------
 * Residual blocks + [RECALLED] Ceres 2.2.0 evaluation and trust-region LM
 * ---------------------------------------------------------------------------------------------- */
struct Residual {
  bool is_plane;
  V3 p;        // moved source point (registration.cpp:34, :75)
  V3 a, b;     // line (edge)
  V3 n;        // plane normal
  double d;    // plane distance
};

// Evaluates one residual block at ambient x = (qx,qy,qz,qw,tx,ty,tz).
// Returns false if the value or the Jacobian is not finite (Ceres: evaluation failure).
// jac6: d r / d(tangent) = [ambient_1x4 * PlusJacobian_4x3 , ambient_1x3]   (registration.h:171-173)
bool evalResidual(const Residual& R, const double x[7], double* r_out, double jac6[6]) {
  const V3 u{x[0], x[1], x[2]};
  const double w = x[3];
  const V3 t{x[4], x[5], x[6]};
  const V3 v = R.p;
  V3 uv = cross(u, v);
  uv = uv + uv;
  const V3 pp = (v + w * uv + cross(u, uv)) + t;  // registration-inl.h:94-98 / :108-112
  V3 g;                                           // d r / d p'
  double r;
  if (R.is_plane) {
    const double s = dot(R.n, pp) - R.d;  // geometry-inl.h:32
    r = std::fabs(s);
    const double sg = std::signbit(s) ? -1.0 : 1.0;  // Jet abs: copysign(1, s)
    g = sg * R.n;
  } else {
    const V3 c = cross(pp - R.a, pp - R.b);  // geometry-inl.h:24-26
    const double cn = norm(c);
    const V3 ab = R.a - R.b;
    const double den = norm(ab);
    r = cn / den;
    g = (1.0 / (cn * den)) * cross(ab, c);  // NaN/inf when the point is exactly on the line
  }
  *r_out = r;
  if (!std::isfinite(r)) return false;
  if (!jac6) return true;
  // ambient Jacobian of p' wrt (ux,uy,uz,w): SURVEY App. A
  const double udv = dot(u, v);
  // dp'/du = -2w[v]x + 2(u.v)I + 2 u v^T - 4 v u^T ; dp'/dw = 2(u x v)
  double M[3][3];
  const double vv[3] = {v.x, v.y, v.z}, uu[3] = {u.x, u.y, u.z};
  const double vx[3][3] = {{0, -v.z, v.y}, {v.z, 0, -v.x}, {-v.y, v.x, 0}};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      M[i][j] = -2.0 * w * vx[i][j] + (i == j ? 2.0 * udv : 0.0) + 2.0 * uu[i] * vv[j] - 4.0 * vv[i] * uu[j];
  const V3 dw = 2.0 * cross(u, v);
  const double gg[3] = {g.x, g.y, g.z};
  double amb[4];  // storage order (qx,qy,qz,qw)
  for (int j = 0; j < 3; j++) amb[j] = gg[0] * M[0][j] + gg[1] * M[1][j] + gg[2] * M[2][j];
  amb[3] = dot(g, dw);
  // Ceres QuaternionManifold::PlusJacobian reads the 4 doubles as (W,X,Y,Z) = (x[0],x[1],x[2],x[3])  (Q9)
  const double W = x[0], X = x[1], Y = x[2], Z = x[3];
  const double PJ[4][3] = {{-X, -Y, -Z}, {W, Z, -Y}, {-Z, W, X}, {Y, -X, W}};
  for (int j = 0; j < 3; j++) jac6[j] = amb[0] * PJ[0][j] + amb[1] * PJ[1][j] + amb[2] * PJ[2][j] + amb[3] * PJ[3][j];
  jac6[3] = g.x, jac6[4] = g.y, jac6[5] = g.z;  // EuclideanManifold<3>
  for (int j = 0; j < 6; j++)
    if (!std::isfinite(jac6[j])) return false;
  return true;
}

// Ceres HuberLoss(a=1): rho(s) and derivatives
inline void huber(double s, double rho[3]) {
  if (s > 1.0) {
    const double r = std::sqrt(s);
    rho[0] = 2.0 * r - 1.0;
    rho[1] = std::max(std::numeric_limits<double>::min(), 1.0 / r);
    rho[2] = -rho[1] / (2.0 * s);
  } else {
    rho[0] = s, rho[1] = 1.0, rho[2] = 0.0;
  }
}

// Program evaluation: cost, corrected residuals f, corrected tangent Jacobian J (row-major M x 6),
// gradient g = J^T f. J/f/g may be null (cost only).
bool evaluate(const std::vector<Residual>& res, const double x[7], double* cost, std::vector<double>* f,
              std::vector<double>* J, double g[6]) {
  const size_t M = res.size();
  double total = 0;
  if (f) f->assign(M, 0.0);
  if (J) J->assign(M * 6, 0.0);
  if (g)
    for (int j = 0; j < 6; j++) g[j] = 0;
  for (size_t i = 0; i < M; i++) {
    double r, jac[6];
    if (!evalResidual(res[i], x, &r, J ? jac : nullptr)) return false;
    double rho[3];
    huber(r * r, rho);
    total += 0.5 * rho[0];
    if (J) {
      // Corrector: rho[2] <= 0 always for Huber => residual and Jacobian scaled by sqrt(rho[1])
      const double sc = std::sqrt(rho[1]);
      for (int j = 0; j < 6; j++) (*J)[i * 6 + j] = jac[j] * sc;
      (*f)[i] = r * sc;
      for (int j = 0; j < 6; j++) g[j] += (*J)[i * 6 + j] * (*f)[i];
    }
  }
  *cost = total;
  return true;
}

// Ceres QuaternionManifold::Plus on 4 raw doubles read as (W,X,Y,Z), then EuclideanManifold<3>
void plus(const double x[7], const double delta[6], double out[7]) {
  const double nd = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
  if (nd == 0.0) {
    for (int i = 0; i < 4; i++) out[i] = x[i];
  } else {
    const double sbd = std::sin(nd) / nd;
    const double z[4] = {std::cos(nd), sbd * delta[0], sbd * delta[1], sbd * delta[2]};
    const double* w = x;
    out[0] = z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3];
    out[1] = z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2];
    out[2] = z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1];
    out[3] = z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0];
  }
  for (int i = 0; i < 3; i++) out[4 + i] = x[4 + i] + delta[3 + i];
}

// DENSE_QR: Eigen HouseholderQR of [J; diag(D)] ((M+6) x 6, column-major), solve for rhs [f; 0].
// Returns false on non-finite output.
bool denseQrSolve(const std::vector<double>& J, const std::vector<double>& f, const double D[6], size_t M,
                  double y[6]) {
  const size_t rows = M + 6;
  std::vector<double> A(rows * 6, 0.0), b(rows, 0.0);
  for (size_t i = 0; i < M; i++) {
    for (int j = 0; j < 6; j++) A[j * rows + i] = J[i * 6 + j];
    b[i] = f[i];
  }
  for (int j = 0; j < 6; j++) A[j * rows + M + j] = D[j];
  double tau[6];
  for (int k = 0; k < 6; k++) {
    double* col = &A[k * rows];
    double tailSq = 0;
    for (size_t r = k + 1; r < rows; r++) tailSq += col[r] * col[r];
    const double c0 = col[k];
    double beta;
    if (tailSq <= std::numeric_limits<double>::min()) {
      tau[k] = 0;
      beta = c0;
      for (size_t r = k + 1; r < rows; r++) col[r] = 0;
    } else {
      beta = std::sqrt(c0 * c0 + tailSq);
      if (c0 >= 0) beta = -beta;
      for (size_t r = k + 1; r < rows; r++) col[r] /= (c0 - beta);
      tau[k] = (beta - c0) / beta;
    }
    col[k] = beta;
    if (tau[k] != 0) {
      for (int j = k + 1; j < 6; j++) {
        double* cj = &A[j * rows];
        double tmp = cj[k];
        for (size_t r = k + 1; r < rows; r++) tmp += col[r] * cj[r];
        cj[k] -= tau[k] * tmp;
        for (size_t r = k + 1; r < rows; r++) cj[r] -= tau[k] * col[r] * tmp;
      }
      double tmp = b[k];
      for (size_t r = k + 1; r < rows; r++) tmp += col[r] * b[r];
      b[k] -= tau[k] * tmp;
      for (size_t r = k + 1; r < rows; r++) b[r] -= tau[k] * col[r] * tmp;
    }
  }
  for (int i = 5; i >= 0; i--) {
    double s = b[i];
    for (int j = i + 1; j < 6; j++) s -= A[j * rows + i] * y[j];
    y[i] = s / A[i * rows + i];
  }
  for (int i = 0; i < 6; i++)
    if (!std::isfinite(y[i])) return false;
  return true;
}

struct LMStats {
  size_t iterations = 0, successful = 0;
  double initial_cost = 0, final_cost = 0;
};

// [RECALLED] ceres::Solve with options {DENSE_QR, max_num_iterations = 4}, everything else default
// (registration-inl.h:51-56). x (7 ambient doubles) is updated in place like the user's parameter
// blocks; on failure at iteration 0 it is left untouched.
void ceresSolve(const std::vector<Residual>& res, double x_user[7], LMStats* stats) {
  const size_t M = res.size();
  const int max_num_iterations = 4;
  const double min_relative_decrease = 1e-3, function_tolerance = 1e-6, gradient_tolerance = 1e-10,
               parameter_tolerance = 1e-8, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32, max_radius = 1e16,
               min_radius = 1e-32;
  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  double diagonal[6];
  double x[7];
  std::memcpy(x, x_user, sizeof(x));
  double x_norm = 0;
  for (int i = 0; i < 7; i++) x_norm += x[i] * x[i];
  x_norm = std::sqrt(x_norm);

  std::vector<double> f, J;
  double g[6], x_cost, scaling[6];
  // IterationZero: EvaluateGradientAndJacobian
  if (!evaluate(res, x, &x_cost, &f, &J, g)) return;  // FAILURE: parameters untouched
  if (stats) stats->initial_cost = stats->final_cost = x_cost;
  for (int j = 0; j < 6; j++) {  // jacobi scaling, computed once at iteration 0
    double s = 0;
    for (size_t i = 0; i < M; i++) s += J[i * 6 + j] * J[i * 6 + j];
    scaling[j] = 1.0 / (1.0 + std::sqrt(s));
  }
  auto scaleColumns = [&]() {
    for (size_t i = 0; i < M; i++)
      for (int j = 0; j < 6; j++) J[i * 6 + j] *= scaling[j];
  };
  auto gradientMaxNorm = [&]() {
    double neg[6], xp[7];
    for (int j = 0; j < 6; j++) neg[j] = -g[j];
    plus(x, neg, xp);
    double m = 0;
    for (int i = 0; i < 7; i++) m = std::max(m, std::fabs(x[i] - xp[i]));
    return m;
  };
  scaleColumns();
  double gradient_max_norm = gradientMaxNorm();
  double minimum_cost = x_cost;
  int iteration = 0;
  bool step_is_successful = false;
  int num_consecutive_invalid_steps = 0;

  for (;;) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (step_is_successful && x_cost < minimum_cost) {
      minimum_cost = x_cost;
      std::memcpy(x_user, x, sizeof(x));
    }
    if (iteration >= max_num_iterations) break;                                   // NO_CONVERGENCE
    if (step_is_successful && gradient_max_norm <= gradient_tolerance) break;     // CONVERGENCE
    if (radius <= min_radius) break;                                              // CONVERGENCE
    iteration++;
    step_is_successful = false;
    if (stats) stats->iterations = iteration;

    // LevenbergMarquardtStrategy::ComputeStep (on the column-scaled Jacobian)
    if (!reuse_diagonal) {
      for (int j = 0; j < 6; j++) {
        double s = 0;
        for (size_t i = 0; i < M; i++) s += J[i * 6 + j] * J[i * 6 + j];
        diagonal[j] = std::min(std::max(s, min_lm_diagonal), max_lm_diagonal);
      }
    }
    double lm_diagonal[6], step[6];
    for (int j = 0; j < 6; j++) lm_diagonal[j] = std::sqrt(diagonal[j] / radius);
    const bool solved = denseQrSolve(J, f, lm_diagonal, M, step);
    reuse_diagonal = true;
    bool step_is_valid = false;
    double model_cost_change = 0;
    if (solved) {
      for (int j = 0; j < 6; j++) step[j] = -step[j];
      // model_cost_change = -(J step)^T (f + J step / 2)
      for (size_t i = 0; i < M; i++) {
        double m = 0;
        for (int j = 0; j < 6; j++) m += J[i * 6 + j] * step[j];
        model_cost_change += -m * (f[i] + m / 2.0);
      }
      step_is_valid = model_cost_change > 0.0;
    }
    if (!step_is_valid) {
      // HandleInvalidStep: LM StepIsInvalid => radius *= 0.5, diagonal recomputed
      if (++num_consecutive_invalid_steps >= 5) break;  // FAILURE
      radius *= 0.5;
      reuse_diagonal = false;
      continue;
    }
    num_consecutive_invalid_steps = 0;
    double delta[6], cand[7];
    for (int j = 0; j < 6; j++) delta[j] = step[j] * scaling[j];
    plus(x, delta, cand);
    double cand_cost;
    if (!evaluate(res, cand, &cand_cost, nullptr, nullptr, nullptr)) cand_cost = std::numeric_limits<double>::max();
    // ParameterToleranceReached (candidate discarded)
    double step_norm = 0;
    for (int i = 0; i < 7; i++) step_norm += (x[i] - cand[i]) * (x[i] - cand[i]);
    step_norm = std::sqrt(step_norm);
    if (step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) break;
    // FunctionToleranceReached (candidate discarded)
    if (std::fabs(x_cost - cand_cost) <= function_tolerance * x_cost) break;
    const double relative_decrease = (x_cost - cand_cost) / model_cost_change;
    if (relative_decrease > min_relative_decrease) {
      // HandleSuccessfulStep
      std::memcpy(x, cand, sizeof(x));
      x_norm = 0;
      for (int i = 0; i < 7; i++) x_norm += x[i] * x[i];
      x_norm = std::sqrt(x_norm);
      if (!evaluate(res, x, &x_cost, &f, &J, g)) break;  // FAILURE; x_user keeps the last finalized point
      scaleColumns();
      gradient_max_norm = gradientMaxNorm();
      step_is_successful = true;
      if (stats) stats->successful++, stats->final_cost = x_cost;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3));
      radius = std::min(max_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = false;
    } else {
      radius = radius / decrease_factor;  // StepRejected
      decrease_factor *= 2.0;
      reuse_diagonal = true;
    }
  }
}

std::vector<V3> toV3(const double* p, size_t n) {
  std::vector<V3> v(n);
  for (size_t i = 0; i < n; i++) v[i] = {p[3 * i], p[3 * i + 1], p[3 * i + 2]};
  return v;
}

// loam/src/registration.cpp:23-62 (edges) and :65-103 (planes)
struct Assoc {
  size_t src, nearest;
};
void associate(const oracle_reg_params& prm, const std::vector<V3>& src, const std::vector<V3>& tgt,
               const KDTree& tree, const Pose& est, bool is_plane, std::vector<Residual>& problem,
               std::vector<Assoc>& assoc) {
  const size_t k = is_plane ? prm.num_plane_neighbors : prm.num_edge_neighbors;
  const double maxd = is_plane ? prm.max_plane_neighbor_dist : prm.max_edge_neighbor_dist;
  const size_t minfit = is_plane ? prm.min_plane_fit_points : prm.min_line_fit_points;
  for (size_t si = 0; si < src.size(); si++) {
    const V3 point_tgt = act(est, src[si]);
    std::vector<size_t> nbr = knnSearch(tree, point_tgt, k, maxd);
    if (nbr.size() < minfit) continue;
    std::vector<V3> npts(nbr.size());
    for (size_t i = 0; i < nbr.size(); i++) npts[i] = tgt[nbr[i]];
    Residual R{};
    R.is_plane = is_plane;
    R.p = point_tgt;
    if (is_plane) {
      const double avg_dist = fitPlane(npts, R.n, R.d);
      if (avg_dist > prm.max_avg_point_plane_dist) continue;  // :90
    } else {
      const double cond = fitLine(npts, R.a, R.b);
      if (cond < prm.min_line_condition_number) continue;  // :49 (never)
    }
    problem.push_back(R);
    assoc.push_back({si, nbr.front()});
  }
}

}  // namespace

/* ================================================================================================
 * C ABI
 * ============================================================================================== */
extern "C" {

void oracle_default_fe_params(oracle_fe_params* p) { *p = {3, 6, 10, 50, 100.0, 1.0, 0.5, 1.0}; }
void oracle_default_reg_params(oracle_reg_params* p) { *p = {5, 1.0, 3, 10.0, 5, 2.0, 4, 0.1, 10, 1e-3, 1e-2, 100}; }

int oracle_compute_curvature(const double* xyz, size_t n_points, size_t H, size_t W, const oracle_fe_params* p,
                             double* out) {
  if (n_points != H * W) return 1;  // common.h:105-113
  auto c = computeCurvature(xyz, H, W, *p);
  for (size_t i = 0; i < c.size(); i++) out[i] = c[i].curvature;
  return 0;
}

int oracle_compute_valid_points(const double* xyz, size_t n_points, size_t H, size_t W, double min_range,
                                double max_range, const oracle_fe_params* p, uint8_t* mask_out) {
  if (n_points != H * W) return 1;
  auto m = computeValidPoints(xyz, H, W, min_range, max_range, *p);
  std::memcpy(mask_out, m.data(), m.size());
  return 0;
}

int oracle_extract_features(const double* xyz, size_t n_points, size_t H, size_t W, double min_range,
                            double max_range, const oracle_fe_params* p, uint32_t* edge_idx, size_t* n_edge,
                            uint32_t* planar_idx, size_t* n_planar) {
  if (n_points != H * W) return 1;
  std::vector<uint32_t> e, pl;
  if (n_points) extractFeatures(xyz, H, W, min_range, max_range, *p, false, e, pl, nullptr);
  std::copy(e.begin(), e.end(), edge_idx);
  std::copy(pl.begin(), pl.end(), planar_idx);
  *n_edge = e.size();
  *n_planar = pl.size();
  return 0;
}

int oracle_extract_features_stable(const double* xyz, size_t n_points, size_t H, size_t W, double min_range,
                                   double max_range, const oracle_fe_params* p, uint32_t* edge_idx, size_t* n_edge,
                                   uint32_t* planar_idx, size_t* n_planar, size_t* n_candidate_ties) {
  if (n_points != H * W) return 1;
  std::vector<uint32_t> e, pl;
  size_t ties = 0;
  if (n_points) extractFeatures(xyz, H, W, min_range, max_range, *p, true, e, pl, &ties);
  std::copy(e.begin(), e.end(), edge_idx);
  std::copy(pl.begin(), pl.end(), planar_idx);
  *n_edge = e.size();
  *n_planar = pl.size();
  if (n_candidate_ties) *n_candidate_ties = ties;
  return 0;
}

void oracle_pose_compose(const double a[7], const double b[7], double out[7]) { to7(compose(from7(a), from7(b)), out); }
void oracle_pose_inverse(const double a[7], double out[7]) { to7(inverse(from7(a)), out); }
void oracle_pose_act(const double a[7], const double p[3], double out[3]) {
  V3 r = act(from7(a), V3{p[0], p[1], p[2]});
  out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
// loam/src/geometry.cpp:24-29 with [RECALLED] Eigen toRotationMatrix
void oracle_pose_matrix(const double a[7], double m[16]) {
  const Pose Pp = from7(a);
  const Quat& q = Pp.q;
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  const double R[3][3] = {{1 - (tyy + tzz), txy - twz, txz + twy},
                          {txy + twz, 1 - (txx + tzz), tyz - twx},
                          {txz - twy, tyz + twx, 1 - (txx + tyy)}};
  const double t[3] = {Pp.t.x, Pp.t.y, Pp.t.z};
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) m[i * 4 + j] = R[i][j];
    m[i * 4 + 3] = t[i];
  }
  m[12] = m[13] = m[14] = 0;
  m[15] = 1;
}
double oracle_quat_angular_distance(const double qa[4], const double qb[4]) {
  return qangdist(Quat{qa[0], qa[1], qa[2], qa[3]}, Quat{qb[0], qb[1], qb[2], qb[3]});
}
double oracle_point_to_line_distance(const double p[3], const double a[3], const double b[3]) {
  return pointToLineDistance(V3{p[0], p[1], p[2]}, V3{a[0], a[1], a[2]}, V3{b[0], b[1], b[2]});
}
double oracle_point_to_plane_distance(const double p[3], const double n[3], double d) {
  return pointToPlaneDistance(V3{p[0], p[1], p[2]}, V3{n[0], n[1], n[2]}, d);
}
double oracle_fit_line(const double* pts, size_t k, double line_out[6]) {
  V3 a, b;
  const double c = fitLine(toV3(pts, k), a, b);
  line_out[0] = a.x, line_out[1] = a.y, line_out[2] = a.z, line_out[3] = b.x, line_out[4] = b.y, line_out[5] = b.z;
  return c;
}
double oracle_fit_plane(const double* pts, size_t k, double plane_out[4]) {
  V3 n;
  double d;
  const double avg = fitPlane(toV3(pts, k), n, d);
  plane_out[0] = n.x, plane_out[1] = n.y, plane_out[2] = n.z, plane_out[3] = d;
  return avg;
}

struct oracle_kdtree {
  std::vector<double> pts;
  KDTree tree;
};
oracle_kdtree* oracle_kdtree_build(const double* pts, size_t n) {
  auto* t = new oracle_kdtree;
  t->pts.assign(pts, pts + 3 * n);
  t->tree.build(t->pts.data(), n);
  return t;
}
void oracle_kdtree_free(oracle_kdtree* t) { delete t; }
size_t oracle_knn_search(const oracle_kdtree* t, const double q[3], size_t k, double max_dist, uint64_t* idx_out) {
  auto r = knnSearch(t->tree, V3{q[0], q[1], q[2]}, k, max_dist);
  for (size_t i = 0; i < r.size(); i++) idx_out[i] = r[i];
  return r.size();
}
size_t oracle_knn_bruteforce(const double* pts, size_t n, const double q[3], size_t k, double max_dist,
                             uint64_t* idx_out) {
  std::vector<std::pair<double, size_t>> d(n);
  for (size_t i = 0; i < n; i++) {
    double s = 0;
    for (int a = 0; a < 3; a++) {
      const double diff = q[a] - pts[3 * i + a];
      s += diff * diff;
    }
    d[i] = {s, i};
  }
  const size_t kk = std::min(k, n);
  std::partial_sort(d.begin(), d.begin() + kk, d.end());
  size_t m = 0;
  for (size_t i = 0; i < kk; i++)
    if (max_dist <= 0 || std::sqrt(d[i].first) < max_dist) idx_out[m++] = d[i].second;
  return m;
}

// loam/include/loam/registration-inl.h:11-78
int oracle_register_features(const double* src_edge, size_t n_se, const double* src_planar, size_t n_sp,
                             const double* tgt_edge, size_t n_te, const double* tgt_planar, size_t n_tp,
                             const double init_pose[7], const oracle_reg_params* prm, double out_pose[7],
                             int* termination_type, uint64_t* n_iterations, oracle_iter_info* info) {
  const std::vector<V3> se = toV3(src_edge, n_se), sp = toV3(src_planar, n_sp);
  const std::vector<V3> te = toV3(tgt_edge, n_te), tp = toV3(tgt_planar, n_tp);
  KDTree edge_tree, plane_tree;  // :20-23
  edge_tree.build(tgt_edge, n_te);
  plane_tree.build(tgt_planar, n_tp);
  Pose est = from7(init_pose);  // :26
  int term = ORACLE_MAX_ITER;   // :27
  uint64_t iters = 0;
  for (size_t it = 0; it < prm->max_iterations; it++) {  // :28
    std::vector<Residual> problem;
    std::vector<Assoc> ea, pa;
    associate(*prm, se, te, edge_tree, est, false, problem, ea);  // :40-41
    const size_t n_edge_res = problem.size();
    associate(*prm, sp, tp, plane_tree, est, true, problem, pa);  // :42-43
    if (ea.size() + pa.size() < prm->min_associations) {           // :45-48
      term = ORACLE_INSUFFICIENT_ASSOCIATIONS;
      break;
    }
    (void)n_edge_res;
    double upd[7] = {0, 0, 0, 1, 0, 0, 0};  // Pose3d estimate_update (identity), :35
    LMStats st;
    ceresSolve(problem, upd, &st);  // :51-56
    if (info) {
      to7(est, info[it].est_before);
      std::memcpy(info[it].update, upd, sizeof(upd));
      info[it].n_edge_assoc = ea.size();
      info[it].n_plane_assoc = pa.size();
      info[it].lm_iterations = st.iterations;
      info[it].lm_successful = st.successful;
      info[it].initial_cost = st.initial_cost;
      info[it].final_cost = st.final_cost;
    }
    iters = it + 1;
    const Pose update = from7(upd);
    est = compose(update, est);  // :65
    const double angle_change = qangdist(update.q, Quat{0, 0, 0, 1});  // :68
    const double position_change = norm(update.t);                     // :69
    if (angle_change < prm->rotation_convergence_thresh && position_change < prm->position_convergence_thresh) {
      term = ORACLE_CONVERGED;  // :70-73
      break;
    }
  }
  to7(est, out_pose);
  if (termination_type) *termination_type = term;
  if (n_iterations) *n_iterations = iters;
  return 0;
}

int oracle_associate(const double* src, size_t n_src, const double* tgt, size_t n_tgt, const double est7[7],
                     int is_plane, const oracle_reg_params* prm, uint8_t* valid, uint64_t* nearest, double* moved,
                     double* prims) {
  const std::vector<V3> s = toV3(src, n_src), t = toV3(tgt, n_tgt);
  KDTree tree;
  tree.build(tgt, n_tgt);
  std::vector<Residual> problem;
  std::vector<Assoc> as;
  const Pose est = from7(est7);
  associate(*prm, s, t, tree, est, is_plane != 0, problem, as);
  const int pw = is_plane ? 4 : 6;
  std::memset(valid, 0, n_src);
  for (size_t i = 0; i < n_src; i++) {
    const V3 m = act(est, s[i]);
    moved[3 * i] = m.x, moved[3 * i + 1] = m.y, moved[3 * i + 2] = m.z;
    nearest[i] = 0;
    for (int j = 0; j < pw; j++) prims[i * pw + j] = 0;
  }
  for (size_t k = 0; k < as.size(); k++) {
    const size_t i = as[k].src;
    valid[i] = 1;
    nearest[i] = as[k].nearest;
    const Residual& R = problem[k];
    if (is_plane) {
      prims[i * 4] = R.n.x, prims[i * 4 + 1] = R.n.y, prims[i * 4 + 2] = R.n.z, prims[i * 4 + 3] = R.d;
    } else {
      prims[i * 6] = R.a.x, prims[i * 6 + 1] = R.a.y, prims[i * 6 + 2] = R.a.z;
      prims[i * 6 + 3] = R.b.x, prims[i * 6 + 4] = R.b.y, prims[i * 6 + 5] = R.b.z;
    }
  }
  return 0;
}

}  // extern "C"
