/** @brief Minimal stand-ins for the three Eigen types that appear in loam's public API
 * (Eigen::Quaterniond, Eigen::Vector3d, Eigen::Matrix4d), used ONLY when <Eigen/Dense> is not
 * installed. With Eigen present, geometry.h aliases the real Eigen types and this file is unused.
 * Storage of the quaternion is (x, y, z, w) like Eigen's coeffs().
 */
#pragma once
#include <cmath>
#include <cstddef>

namespace loam {
namespace mini {

struct Vector3d {
  double v[3];
  Vector3d() : v{0, 0, 0} {}
  Vector3d(double x, double y, double z) : v{x, y, z} {}
  static Vector3d Zero() { return Vector3d(); }
  double& operator()(size_t i) { return v[i]; }
  const double& operator()(size_t i) const { return v[i]; }
  double& operator[](size_t i) { return v[i]; }
  const double& operator[](size_t i) const { return v[i]; }
  double& x() { return v[0]; }
  double& y() { return v[1]; }
  double& z() { return v[2]; }
  const double& x() const { return v[0]; }
  const double& y() const { return v[1]; }
  const double& z() const { return v[2]; }
  double* data() { return v; }
  const double* data() const { return v; }
  Vector3d operator+(const Vector3d& o) const { return {v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}; }
  Vector3d operator-(const Vector3d& o) const { return {v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}; }
  Vector3d operator-() const { return {-v[0], -v[1], -v[2]}; }
  Vector3d operator*(double s) const { return {v[0] * s, v[1] * s, v[2] * s}; }
  Vector3d operator/(double s) const { return {v[0] / s, v[1] / s, v[2] / s}; }
  double dot(const Vector3d& o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
  Vector3d cross(const Vector3d& o) const {
    return {v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]};
  }
  double squaredNorm() const { return dot(*this); }
  double norm() const { return std::sqrt(squaredNorm()); }
  bool isApprox(const Vector3d& o, double prec = 1e-12) const {
    const double a = squaredNorm(), b = o.squaredNorm();
    return (*this - o).squaredNorm() <= prec * prec * (a < b ? a : b);
  }
};
inline Vector3d operator*(double s, const Vector3d& a) { return a * s; }

struct Matrix4d {
  double m[4][4];
  static Matrix4d Identity() {
    Matrix4d r;
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) r.m[i][j] = i == j ? 1.0 : 0.0;
    return r;
  }
  double& operator()(size_t i, size_t j) { return m[i][j]; }
  const double& operator()(size_t i, size_t j) const { return m[i][j]; }
  bool isApprox(const Matrix4d& o, double prec = 1e-12) const {
    double d = 0, a = 0, b = 0;
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        d += (m[i][j] - o.m[i][j]) * (m[i][j] - o.m[i][j]);
        a += m[i][j] * m[i][j];
        b += o.m[i][j] * o.m[i][j];
      }
    return d <= prec * prec * (a < b ? a : b);
  }
};

struct Quaterniond {
  double c[4];  // x, y, z, w
  struct Coeffs {
    double* p;
    double* data() { return p; }
    double& operator()(size_t i) { return p[i]; }
  };
  Quaterniond() : c{0, 0, 0, 1} {}
  /// Same argument order as Eigen: (w, x, y, z)
  Quaterniond(double w, double x, double y, double z) : c{x, y, z, w} {}
  static Quaterniond Identity() { return Quaterniond(1, 0, 0, 0); }
  /// Unit quaternion of a rotation of `angle` radians about the unit vector `axis`
  static Quaterniond FromAngleAxis(double angle, const Vector3d& axis) {
    const double s = std::sin(0.5 * angle);
    return Quaterniond(std::cos(0.5 * angle), s * axis(0), s * axis(1), s * axis(2));
  }
  double& x() { return c[0]; }
  double& y() { return c[1]; }
  double& z() { return c[2]; }
  double& w() { return c[3]; }
  const double& x() const { return c[0]; }
  const double& y() const { return c[1]; }
  const double& z() const { return c[2]; }
  const double& w() const { return c[3]; }
  Coeffs coeffs() { return Coeffs{c}; }
  Vector3d vec() const { return {c[0], c[1], c[2]}; }
  double squaredNorm() const { return c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]; }
  Quaterniond conjugate() const { return Quaterniond(c[3], -c[0], -c[1], -c[2]); }
  Quaterniond inverse() const {
    const double n2 = squaredNorm();
    if (n2 > 0) return Quaterniond(c[3] / n2, -c[0] / n2, -c[1] / n2, -c[2] / n2);
    return Quaterniond(0, 0, 0, 0);
  }
  Quaterniond operator*(const Quaterniond& b) const {
    const Quaterniond& a = *this;
    return Quaterniond(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                       a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                       a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                       a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
  }
  Vector3d operator*(const Vector3d& v) const {
    Vector3d uv = vec().cross(v);
    uv = uv + uv;
    return v + uv * w() + vec().cross(uv);
  }
  double angularDistance(const Quaterniond& o) const {
    const Quaterniond d = (*this) * o.conjugate();
    return 2.0 * std::atan2(d.vec().norm(), std::fabs(d.w()));
  }
  bool isApprox(const Quaterniond& o, double prec = 1e-12) const {
    double d = 0;
    for (int i = 0; i < 4; i++) d += (c[i] - o.c[i]) * (c[i] - o.c[i]);
    const double a = squaredNorm(), b = o.squaredNorm();
    return d <= prec * prec * (a < b ? a : b);
  }
};

}  // namespace mini
}  // namespace loam
