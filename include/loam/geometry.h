/** @brief Geometry types of the loam API (drop-in for the reference's loam/include/loam/geometry.h).
 * Pose3d is a plain value type (quaternion + translation); its algebra is a handful of flops and
 * stays on the host exactly like the reference's (loam/src/geometry.cpp:10-29). The line / plane
 * fits of registration run inside the HIP association kernel; geometry_internal::fitLine / fitPlane
 * (reference geometry.h:102, :123) call the same device functions through loamx_fit_lines / loamx_fit_planes.
 */
#pragma once
#include <cmath>
#include <limits>
#include <utility>
#include <vector>

#include "common.h"

namespace loam {

/// A pose in 3d space (reference geometry.h:27-50)
struct Pose3d {
  Quaterniond rotation;
  Vector3d translation;

  Pose3d(Quaterniond rot, Vector3d trans) : rotation(rot), translation(trans) {}
  Pose3d() : rotation(Quaterniond::Identity()), translation(Vector3d::Zero()) {}
  static Pose3d Identity() { return Pose3d(Quaterniond::Identity(), Vector3d::Zero()); }

  /// P^{-1}
  Pose3d inverse() const {
    const Quaterniond inv = rotation.inverse();
    return Pose3d(inv, inv * (-translation));
  }
  /// P (+) other
  Pose3d compose(const Pose3d& other) const {
    return Pose3d(rotation * other.rotation, translation + (rotation * other.translation));
  }
  /// p_e = e_T_s * p_s
  Vector3d act(const Vector3d& p) const { return rotation * p + translation; }
  /// 4x4 homogeneous matrix
  Matrix4d matrix() const {
    Matrix4d mat = Matrix4d::Identity();
    const double x = rotation.x(), y = rotation.y(), z = rotation.z(), w = rotation.w();
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    mat(0, 0) = 1 - (tyy + tzz), mat(0, 1) = txy - twz, mat(0, 2) = txz + twy;
    mat(1, 0) = txy + twz, mat(1, 1) = 1 - (txx + tzz), mat(1, 2) = tyz - twx;
    mat(2, 0) = txz - twy, mat(2, 1) = tyz + twx, mat(2, 2) = 1 - (txx + tyy);
    for (int i = 0; i < 3; i++) mat(i, 3) = translation(i);
    return mat;
  }

  /// C-ABI layout {qx,qy,qz,qw,tx,ty,tz}
  void toArray(double out[7]) const {
    out[0] = rotation.x(), out[1] = rotation.y(), out[2] = rotation.z(), out[3] = rotation.w();
    for (int i = 0; i < 3; i++) out[4 + i] = translation(i);
  }
  static Pose3d fromArray(const double p[7]) {
    return Pose3d(Quaterniond(p[3], p[0], p[1], p[2]), Vector3d(p[4], p[5], p[6]));
  }
};

namespace geometry_internal {

/// A line through two points (reference geometry.h:69-78)
struct Line {
  const Vector3d a;
  const Vector3d b;
  Line(Vector3d a, Vector3d b) : a(a), b(b) {}
};
/// A plane n.p - d = 0 (reference geometry.h:84-93)
struct Plane {
  const Vector3d normal;
  const double d;
  Plane(Vector3d normal, double d) : normal(normal), d(d) {}
};

/// K x 3 point matrix as the fits take it: Eigen::MatrixXd when Eigen is installed (the reference's parameter type),
/// otherwise any container of Vector3d. Both are packed row-major for the C ABI.
inline std::vector<double> packRows(const std::vector<Vector3d>& points) {
  std::vector<double> xyz(points.size() * 3);
  for (size_t i = 0; i < points.size(); i++) xyz[3 * i] = points[i](0), xyz[3 * i + 1] = points[i](1), xyz[3 * i + 2] = points[i](2);
  return xyz;
}
#if LOAM_HAVE_EIGEN
inline std::vector<double> packRows(const Eigen::MatrixXd& points) {
  std::vector<double> xyz((size_t)points.rows() * 3);
  for (Eigen::Index i = 0; i < points.rows(); i++) xyz[3 * i] = points(i, 0), xyz[3 * i + 1] = points(i, 1), xyz[3 * i + 2] = points(i, 2);
  return xyz;
}
#endif

/** @brief Fits a line to K >= 2 points by PCA (reference geometry.h:102, geometry.cpp:42-59): the line through the
 * centroid along the eigenvector of the largest eigenvalue, as Line(centre + 0.1 dir, centre - 0.1 dir), and the
 * condition number the reference returns — std::numeric_limits<double>::max() always (geometry.cpp:55-56).
 * Runs on the device (loamx_fit_lines). LIMIT (the reference's Eigen::MatrixXd has none): K <= 32 rows — beyond it
 * loamx_fit_lines returns LOAMX_ERR_UNSUPPORTED and this throws std::runtime_error. */
template <typename Points>
std::pair<Line, double> fitLine(const Points& points) {
  const std::vector<double> xyz = packRows(points);
  double line[6], cond = std::numeric_limits<double>::max();
  gpu::check(gpu::defaultContext(), loamx_fit_lines(gpu::defaultContext(), xyz.data(), 1, xyz.size() / 3, line, &cond));
  return std::make_pair(Line(Vector3d(line[0], line[1], line[2]), Vector3d(line[3], line[4], line[5])), cond);
}

/** @brief Fits a plane to K >= 3 points: least squares of points * [a b c]^T = 1 by column-pivoted Householder QR,
 * normal = abc / |abc|, d = 1 / |abc|; second = the signed mean of points * normal - d (reference geometry.h:123,
 * geometry.cpp:62-73). Runs on the device (loamx_fit_planes). LIMIT: K <= 32 rows (std::runtime_error beyond, as fitLine). */
template <typename Points>
std::pair<Plane, double> fitPlane(const Points& points) {
  const std::vector<double> xyz = packRows(points);
  double plane[4], avg = 0.0;
  gpu::check(gpu::defaultContext(), loamx_fit_planes(gpu::defaultContext(), xyz.data(), 1, xyz.size() / 3, plane, &avg));
  return std::make_pair(Plane(Vector3d(plane[0], plane[1], plane[2]), plane[3]), avg);
}

/// Distance between a point and the line through a and b (reference geometry-inl.h:21-27)
template <typename Vec>
auto pointToLineDistance(const Vec& point, const Vec& line_a, const Vec& line_b) {
  return ((point - line_a).cross(point - line_b)).norm() / (line_a - line_b).norm();
}
/// Distance between a point and the plane (normal, distance) (reference geometry-inl.h:30-33)
template <typename Vec, typename T>
auto pointToPlaneDistance(const Vec& point, const Vec& normal, const T distance) {
  using std::abs;
  return abs(normal.dot(point) - distance);
}

}  // namespace geometry_internal
}  // namespace loam
