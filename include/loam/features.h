/** @brief LOAM feature extraction — drop-in for the reference's loam/include/loam/features.h.
 * Same templates, parameter structs, defaults, output order and exceptions; the computation runs on
 * the MI355X through the C ABI (loamx_extract_features & co.).
 */
#pragma once
#include <memory>
#include <vector>

#include "common.h"

namespace loam {

/// Feature extraction parameters (reference features.h:37-66; same field order and defaults)
struct FeatureExtractionParams {
  size_t neighbor_points{3};
  size_t number_sectors{6};
  size_t max_edge_feats_per_sector{10};
  size_t max_planar_feats_per_sector{50};
  double edge_feat_threshold{100.0};
  double planar_feat_threshold{1.0};
  double occlusion_thresh{0.5};
  double parallel_thresh{1.0};
};

/// Edge and planar feature points of one scan (reference features.h:70-76)
template <typename PointType, template <typename> class Alloc = std::allocator>
struct LoamFeatures {
  std::vector<PointType, Alloc<PointType>> edge_points;
  std::vector<PointType, Alloc<PointType>> planar_points;
};

/// Curvature of one point (reference features.h:79-88)
struct PointCurvature {
  size_t index;
  double curvature;
  PointCurvature(size_t index, double curvature) : index(index), curvature(curvature) {}
  PointCurvature() = default;
};
inline bool curvatureComparator(const PointCurvature& lhs, const PointCurvature& rhs) {
  return lhs.curvature < rhs.curvature;
}

namespace gpu {
inline loamx_fe_params toC(const FeatureExtractionParams& p) {
  return loamx_fe_params{p.neighbor_points, p.number_sectors, p.max_edge_feats_per_sector,
                         p.max_planar_feats_per_sector, p.edge_feat_threshold, p.planar_feat_threshold,
                         p.occlusion_thresh, p.parallel_thresh};
}
}  // namespace gpu

/// Extracts LOAM features from a row-major LiDAR scan (reference features.h:108-111)
template <template <typename> class Accessor = FieldAccessor, typename PointType, template <typename> class Alloc>
LoamFeatures<PointType, Alloc> extractFeatures(const std::vector<PointType, Alloc<PointType>>& input_scan,
                                               const LidarParams& lidar_params,
                                               const FeatureExtractionParams& params = FeatureExtractionParams()) {
  validateLidarScan(input_scan, lidar_params);
  LoamFeatures<PointType, Alloc> out;
  if (input_scan.empty()) return out;
  loamx_ctx* ctx = gpu::defaultContext();
  const loamx_lidar_params lp = gpu::toC(lidar_params);
  const loamx_fe_params fp = gpu::toC(params);
  std::vector<uint32_t> edge(loamx_edge_capacity(&lp, &fp) + 1), planar(loamx_planar_capacity(&lp, &fp) + 1);
  size_t n_edge = 0, n_planar = 0;
  if constexpr (gpu::float_scan_v<Accessor, PointType>) {  // PCL-style float points: FP32-input path
    const std::vector<float> xyz = gpu::packFloat(input_scan);
    gpu::check(ctx, loamx_extract_features_f32(ctx, xyz.data(), input_scan.size(), &lp, &fp, edge.data(), edge.size(), &n_edge,
                                               planar.data(), planar.size(), &n_planar));
  } else {
    const std::vector<double> xyz = gpu::pack<Accessor>(input_scan);
    gpu::check(ctx, loamx_extract_features(ctx, xyz.data(), input_scan.size(), &lp, &fp, edge.data(), edge.size(), &n_edge,
                                           planar.data(), planar.size(), &n_planar));
  }
  out.edge_points.reserve(n_edge);
  out.planar_points.reserve(n_planar);
  for (size_t i = 0; i < n_edge; i++) out.edge_points.push_back(input_scan.at(edge[i]));  // copies, like the reference
  for (size_t i = 0; i < n_planar; i++) out.planar_points.push_back(input_scan.at(planar[i]));
  return out;
}

/// Un-normalised curvature of every point (reference features.h:119-122)
template <template <typename> class Accessor = FieldAccessor, typename PointType, template <typename> class Alloc>
std::vector<PointCurvature> computeCurvature(const std::vector<PointType, Alloc<PointType>>& input_scan,
                                             const LidarParams& lidar_params,
                                             const FeatureExtractionParams& params = FeatureExtractionParams()) {
  validateLidarScan(input_scan, lidar_params);
  std::vector<PointCurvature> out;
  if (input_scan.empty()) return out;
  loamx_ctx* ctx = gpu::defaultContext();
  const loamx_lidar_params lp = gpu::toC(lidar_params);
  const loamx_fe_params fp = gpu::toC(params);
  std::vector<double> curv(input_scan.size());
  if constexpr (gpu::float_scan_v<Accessor, PointType>) {
    const std::vector<float> xyz = gpu::packFloat(input_scan);
    gpu::check(ctx, loamx_compute_curvature_f32(ctx, xyz.data(), input_scan.size(), &lp, &fp, curv.data()));
  } else {
    const std::vector<double> xyz = gpu::pack<Accessor>(input_scan);
    gpu::check(ctx, loamx_compute_curvature(ctx, xyz.data(), input_scan.size(), &lp, &fp, curv.data()));
  }
  out.reserve(curv.size());
  for (size_t i = 0; i < curv.size(); i++) out.emplace_back(i, curv[i]);
  return out;
}

/// Validity mask of every point (reference features.h:166-169)
template <template <typename> class Accessor = FieldAccessor, typename PointType, template <typename> class Alloc>
std::vector<bool> computeValidPoints(const std::vector<PointType, Alloc<PointType>>& input_scan,
                                     const LidarParams& lidar_params,
                                     const FeatureExtractionParams& params = FeatureExtractionParams()) {
  validateLidarScan(input_scan, lidar_params);
  std::vector<bool> out;
  if (input_scan.empty()) return out;
  loamx_ctx* ctx = gpu::defaultContext();
  const loamx_lidar_params lp = gpu::toC(lidar_params);
  const loamx_fe_params fp = gpu::toC(params);
  std::vector<uint8_t> mask(input_scan.size());
  if constexpr (gpu::float_scan_v<Accessor, PointType>) {
    const std::vector<float> xyz = gpu::packFloat(input_scan);
    gpu::check(ctx, loamx_compute_valid_points_f32(ctx, xyz.data(), input_scan.size(), &lp, &fp, mask.data()));
  } else {
    const std::vector<double> xyz = gpu::pack<Accessor>(input_scan);
    gpu::check(ctx, loamx_compute_valid_points(ctx, xyz.data(), input_scan.size(), &lp, &fp, mask.data()));
  }
  out.assign(mask.begin(), mask.end());
  return out;
}

namespace features_internal {
/// Converts features of PointType to 3-vectors (reference features.h:188-198)
template <template <typename> class Accessor = FieldAccessor, typename PointType, template <typename> class Alloc>
LoamFeatures<Vector3d> featuresToEigen(const LoamFeatures<PointType, Alloc>& in_features) {
  LoamFeatures<Vector3d> result;
  for (const PointType& pt : in_features.edge_points) result.edge_points.push_back(pointToEigen<Accessor>(pt));
  for (const PointType& pt : in_features.planar_points) result.planar_points.push_back(pointToEigen<Accessor>(pt));
  return result;
}
}  // namespace features_internal

}  // namespace loam
