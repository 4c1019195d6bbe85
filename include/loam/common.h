/** @brief Common types of the loam API: LidarParams, point accessors, scan validation.
 * Drop-in for the reference's loam/include/loam/common.h (same names, members and error
 * behaviour); the heavy lifting happens in libloamx.so (MI355X HIP kernels) behind include/loamx.h.
 */
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../loamx.h"

#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#define LOAM_HAVE_EIGEN 1
namespace loam {
using Quaterniond = Eigen::Quaterniond;
using Vector3d = Eigen::Vector3d;
using Matrix4d = Eigen::Matrix4d;
}  // namespace loam
#else
#include "mini_eigen.h"
#define LOAM_HAVE_EIGEN 0
namespace loam {
using Quaterniond = mini::Quaterniond;
using Vector3d = mini::Vector3d;
using Matrix4d = mini::Matrix4d;
}  // namespace loam
#endif

namespace loam {

/// Intrinsic LiDAR parameters (reference common.h:29-41): const members, 4-argument constructor.
struct LidarParams {
  const size_t scan_lines;
  const size_t points_per_line;
  const double min_range;
  const double max_range;
  LidarParams(size_t scan_lines, size_t points_per_line, double min_range, double max_range)
      : scan_lines(scan_lines), points_per_line(points_per_line), min_range(min_range), max_range(max_range) {}
};

/// Accessor for points with public fields x, y, z (e.g. PCL points) — reference common.h:55-60
template <typename PointType>
struct FieldAccessor {
  static double x(PointType pt) { return pt.x; }
  static double y(PointType pt) { return pt.y; }
  static double z(PointType pt) { return pt.z; }
};
/// Accessor for points indexed with parentheses (e.g. Eigen vectors) — reference common.h:64-69
template <typename PointType>
struct ParenAccessor {
  static double x(PointType pt) { return pt(0); }
  static double y(PointType pt) { return pt(1); }
  static double z(PointType pt) { return pt(2); }
};
/// Accessor for points indexed with .at() (e.g. std::vector) — reference common.h:73-78
template <typename PointType>
struct AtAccessor {
  static double x(PointType pt) { return pt.at(0); }
  static double y(PointType pt) { return pt.at(1); }
  static double z(PointType pt) { return pt.at(2); }
};

/// Range from the LiDAR to the point (reference common.h:81-86)
template <template <typename> class Accessor = FieldAccessor, typename PointType>
double pointRange(const PointType& pt) {
  const double x = Accessor<PointType>::x(pt), y = Accessor<PointType>::y(pt), z = Accessor<PointType>::z(pt);
  return std::sqrt(x * x + y * y + z * z);
}

/// Converts a point into a 3-vector (reference common.h:89-93)
template <template <typename> class Accessor = FieldAccessor, typename PointType>
Vector3d pointToEigen(const PointType& pt) {
  return Vector3d(Accessor<PointType>::x(pt), Accessor<PointType>::y(pt), Accessor<PointType>::z(pt));
}

/// Throws std::runtime_error if the scan size does not match the parameters (reference common.h:104-113)
template <typename PointType, template <typename> class Alloc>
void validateLidarScan(const std::vector<PointType, Alloc<PointType>>& input_scan, const LidarParams& lidar_params) {
  if (input_scan.size() != lidar_params.scan_lines * lidar_params.points_per_line) {
    std::stringstream msg;
    msg << "LOAM: provided lidar scan size ( " << input_scan.size() << ")  does not match provided lidar parameters ("
        << lidar_params.scan_lines << " x " << lidar_params.points_per_line << ")";
    throw std::runtime_error(msg.str());
  }
}

/// MI355X back end plumbing shared by features.h and registration.h
namespace gpu {

/// Process-wide default context (device 0 or $LOAMX_DEVICE). Calls are serialised per context by
/// the library. Throws if no MI355X is usable: there is no CPU fallback.
inline loamx_ctx* defaultContext() {
  static loamx_ctx* ctx = nullptr;
  static std::once_flag once;
  static int status = LOAMX_OK;
  std::call_once(once, [] {
    int device = 0;
    if (const char* env = std::getenv("LOAMX_DEVICE")) device = std::atoi(env);
    status = loamx_ctx_create(device, &ctx);
  });
  if (status != LOAMX_OK || !ctx)
    throw std::runtime_error(std::string("loam (MI355X back end): cannot create a device context: ") +
                             loamx_status_string(status));
  return ctx;
}

inline void check(loamx_ctx* ctx, int status) {
  if (status == LOAMX_OK) return;
  const char* detail = loamx_last_error(ctx);
  throw std::runtime_error(detail && detail[0] ? std::string(detail) : std::string(loamx_status_string(status)));
}

inline loamx_lidar_params toC(const LidarParams& p) {
  return loamx_lidar_params{p.scan_lines, p.points_per_line, p.min_range, p.max_range};
}

/// Packs any point container into row-major N x 3 doubles through the Accessor
template <template <typename> class Accessor, typename PointType, typename Alloc>
std::vector<double> pack(const std::vector<PointType, Alloc>& pts) {
  std::vector<double> xyz(pts.size() * 3);
  for (size_t i = 0; i < pts.size(); i++) {
    xyz[3 * i] = Accessor<PointType>::x(pts[i]);
    xyz[3 * i + 1] = Accessor<PointType>::y(pts[i]);
    xyz[3 * i + 2] = Accessor<PointType>::z(pts[i]);
  }
  return xyz;
}

/// Scans of points whose x, y, z are float fields read through FieldAccessor (PCL points) go to the device as
/// floats: FieldAccessor widens every coordinate to double before any arithmetic (reference common.h:55-60), and
/// the FP32-input entry points of the C ABI do the same on load, so the results are identical at half the bytes.
template <template <typename> class A, template <typename> class B>
struct same_accessor : std::false_type {};
template <template <typename> class A>
struct same_accessor<A, A> : std::true_type {};
template <typename P, typename = void>
struct has_float_fields : std::false_type {};
template <typename P>
struct has_float_fields<P, std::void_t<decltype(std::declval<P>().x), decltype(std::declval<P>().y), decltype(std::declval<P>().z)>>
    : std::integral_constant<bool, std::is_same<std::decay_t<decltype(std::declval<P>().x)>, float>::value &&
                                       std::is_same<std::decay_t<decltype(std::declval<P>().y)>, float>::value &&
                                       std::is_same<std::decay_t<decltype(std::declval<P>().z)>, float>::value> {};
template <template <typename> class Accessor, typename PointType>
constexpr bool float_scan_v = same_accessor<Accessor, FieldAccessor>::value && has_float_fields<PointType>::value;

/// Packs a container of float-field points into row-major N x 3 floats
template <typename PointType, typename Alloc>
std::vector<float> packFloat(const std::vector<PointType, Alloc>& pts) {
  std::vector<float> xyz(pts.size() * 3);
  for (size_t i = 0; i < pts.size(); i++) xyz[3 * i] = pts[i].x, xyz[3 * i + 1] = pts[i].y, xyz[3 * i + 2] = pts[i].z;
  return xyz;
}

}  // namespace gpu
}  // namespace loam
