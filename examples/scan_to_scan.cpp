// The reference's C++ API on the MI355X back end: the same calls a user of DanMcGann/loam writes
// (loam/include/loam/features.h:108-111, registration.h:128-131), against the drop-in headers in include/loam/.
//
//   python -m loam_amd.build
//   g++ -std=c++17 -O2 -I include examples/scan_to_scan.cpp -o scan_to_scan \
//       -L loam_amd/lib -lloamx -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,loam_amd/lib -Wl,-rpath,/opt/rocm/lib
#include <cmath>
#include <cstdio>
#include <vector>

#include "loam/loam.h"

struct PointXYZ {  // a PCL-style point: float fields, read through the default FieldAccessor
  float x, y, z;
};

// a 16 x 512 scan of a room with a pillar, seen from `shift` metres along x
static std::vector<PointXYZ> makeScan(double shift) {
  std::vector<PointXYZ> scan;
  for (int line = 0; line < 16; line++) {
    for (int col = 0; col < 512; col++) {
      const double el = -0.25 + 0.5 * line / 15.0, az = 2.0 * 3.14159265358979323846 * col / 512.0;
      const double dx = std::cos(el) * std::cos(az), dy = std::cos(el) * std::sin(az), dz = std::sin(el);
      double r = 1e9;  // nearest wall of the box [-8, 8] x [-6, 6] x [-2, 3] seen from (shift, 0, 0)
      const double o[3] = {shift, 0, 0}, d[3] = {dx, dy, dz}, lo[3] = {-8, -6, -2}, hi[3] = {8, 6, 3};
      for (int a = 0; a < 3; a++) {
        if (d[a] > 1e-12) r = std::fmin(r, (hi[a] - o[a]) / d[a]);
        if (d[a] < -1e-12) r = std::fmin(r, (lo[a] - o[a]) / d[a]);
      }
      // a pillar of radius 0.4 at (3, 2): first intersection, if any
      const double px = 3 - shift, py = 2, b = dx * px + dy * py, c2 = px * px + py * py - 0.16, aa = dx * dx + dy * dy;
      const double disc = b * b - aa * c2;
      if (disc > 0 && (b - std::sqrt(disc)) / aa > 0) r = std::fmin(r, (b - std::sqrt(disc)) / aa);
      r += 1e-3 * std::sin(37.0 * col + 11.0 * line + 5.0 * shift);  // a little texture instead of sensor noise
      scan.push_back(PointXYZ{(float)(r * dx), (float)(r * dy), (float)(r * dz)});
    }
  }
  return scan;
}

int main() {
  const loam::LidarParams lidar(16, 512, 0.5, 100.0);
  const std::vector<PointXYZ> scan_a = makeScan(0.0), scan_b = makeScan(0.2);
  const auto feat_a = loam::extractFeatures(scan_a, lidar);  // float fields: the FP32-input path
  const auto feat_b = loam::extractFeatures(scan_b, lidar);
  std::printf("features: %zu edge / %zu planar, %zu edge / %zu planar\n", feat_a.edge_points.size(), feat_a.planar_points.size(),
              feat_b.edge_points.size(), feat_b.planar_points.size());
  auto detail = std::make_shared<loam::RegistrationDetail>();
  const loam::Pose3d a_T_b = loam::registerFeatures(feat_b, feat_a, loam::Pose3d(), loam::RegistrationParams(), detail);
  std::printf("a_T_b translation = (%.4f %.4f %.4f), expected (0.2 0 0); %zu ICF iterations, termination %d\n", a_T_b.translation(0),
              a_T_b.translation(1), a_T_b.translation(2), detail->iteration_info.size(), (int)detail->termination_type);
  return std::fabs(a_T_b.translation(0) - 0.2) < 0.02 ? 0 : 1;
}
