/** @brief Geometry types of the loam API (drop-in for the reference's loam/include/loam/geometry.h).
 * Pose3d is a plain value type (quaternion + translation); its algebra is a handful of flops and
 * stays on the host exactly like the reference's (loam/src/geometry.cpp:10-29). The line / plane
 * fits of registration run inside the HIP association kernel.
 */
#pragma once
#include <cmath>
#include <utility>

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
