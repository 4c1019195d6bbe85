/** @brief LOAM feature registration — drop-in for the reference's loam/include/loam/registration.h.
 * Same template, parameter / detail structs and defaults; the iterative closest feature loop
 * (association, line / plane fits, robust Levenberg-Marquardt) runs on the MI355X through
 * loamx_register_features.
 */
#pragma once
#include <memory>
#include <utility>
#include <vector>

#include "common.h"
#include "features.h"
#include "geometry.h"
#include "kdtree.h"

namespace loam {

/// Registration parameters (reference registration.h:40-75; same field order and defaults)
struct RegistrationParams {
  size_t num_edge_neighbors{5};
  double max_edge_neighbor_dist{1.0};
  size_t min_line_fit_points{3};
  double min_line_condition_number{10};
  size_t num_plane_neighbors{5};
  double max_plane_neighbor_dist{2.0};
  size_t min_plane_fit_points{4};
  double max_avg_point_plane_dist{0.1};
  size_t max_iterations{10};
  double rotation_convergence_thresh{1e-3};
  double position_convergence_thresh{1e-2};
  size_t min_associations{100};
};

/// Detailed information about one registration (reference registration.h:79-109)
struct RegistrationDetail {
  enum TerminationType { CONVERGED, MAX_ITER, INSUFFICIENT_ASSOCIATIONS };
  struct IterationInfo {
    Pose3d target_T_source_init;
    std::vector<std::pair<size_t, size_t>> edge_associations;
    std::vector<std::pair<size_t, size_t>> plane_associations;
    Pose3d estimate_update;
    IterationInfo(const Pose3d target_T_source_init, const std::vector<std::pair<size_t, size_t>> edge_associations,
                  const std::vector<std::pair<size_t, size_t>> plane_associations, const Pose3d estimate_update)
        : target_T_source_init(target_T_source_init),
          edge_associations(edge_associations),
          plane_associations(plane_associations),
          estimate_update(estimate_update) {}
  };
  std::vector<IterationInfo> iteration_info;
  TerminationType termination_type;
};

namespace gpu {
inline loamx_reg_params toC(const RegistrationParams& p) {
  return loamx_reg_params{p.num_edge_neighbors,     p.max_edge_neighbor_dist,      p.min_line_fit_points,
                          p.min_line_condition_number, p.num_plane_neighbors,      p.max_plane_neighbor_dist,
                          p.min_plane_fit_points,   p.max_avg_point_plane_dist,    p.max_iterations,
                          p.rotation_convergence_thresh, p.position_convergence_thresh, p.min_associations};
}
}  // namespace gpu

/// Registers source to target, returning target_T_source (reference registration.h:128-131)
template <template <typename> class Accessor = FieldAccessor, typename PointType, template <typename> class Alloc>
Pose3d registerFeatures(const LoamFeatures<PointType, Alloc>& source, const LoamFeatures<PointType, Alloc>& target,
                        const Pose3d& target_T_source_init, const RegistrationParams& params = RegistrationParams(),
                        std::shared_ptr<RegistrationDetail> detail = nullptr) {
  loamx_ctx* ctx = gpu::defaultContext();
  const std::vector<double> se = gpu::pack<Accessor>(source.edge_points), sp = gpu::pack<Accessor>(source.planar_points);
  const std::vector<double> te = gpu::pack<Accessor>(target.edge_points), tp = gpu::pack<Accessor>(target.planar_points);
  const loamx_reg_params rp = gpu::toC(params);
  double init[7];
  target_T_source_init.toArray(init);
  loamx_reg_result result{};
  const size_t n_se = source.edge_points.size(), n_sp = source.planar_points.size(), mi = params.max_iterations;
  std::vector<loamx_iter_info> info;
  std::vector<uint32_t> edge_pairs, plane_pairs, n_edge_pairs, n_plane_pairs;
  loamx_reg_detail cdetail{};
  if (detail) {
    info.resize(mi ? mi : 1);
    edge_pairs.resize(2 * (n_se ? n_se : 1) * (mi ? mi : 1));
    plane_pairs.resize(2 * (n_sp ? n_sp : 1) * (mi ? mi : 1));
    n_edge_pairs.assign(mi ? mi : 1, 0);
    n_plane_pairs.assign(mi ? mi : 1, 0);
    cdetail.iter_info = info.data();
    cdetail.edge_pairs = edge_pairs.data(), cdetail.pairs_cap_edge = n_se ? n_se : 1, cdetail.n_edge_pairs = n_edge_pairs.data();
    cdetail.plane_pairs = plane_pairs.data(), cdetail.pairs_cap_plane = n_sp ? n_sp : 1, cdetail.n_plane_pairs = n_plane_pairs.data();
  }
  gpu::check(ctx, loamx_register_features(ctx, se.data(), n_se, sp.data(), n_sp, te.data(), target.edge_points.size(),
                                          tp.data(), target.planar_points.size(), init, &rp, &result,
                                          detail ? &cdetail : nullptr));
  if (detail) {
    for (uint32_t it = 0; it < cdetail.n_iter_info; it++) {
      std::vector<std::pair<size_t, size_t>> ea, pa;
      const uint32_t* e = edge_pairs.data() + (size_t)it * 2 * cdetail.pairs_cap_edge;
      const uint32_t* p = plane_pairs.data() + (size_t)it * 2 * cdetail.pairs_cap_plane;
      for (uint32_t k = 0; k < n_edge_pairs[it]; k++) ea.emplace_back(e[2 * k], e[2 * k + 1]);
      for (uint32_t k = 0; k < n_plane_pairs[it]; k++) pa.emplace_back(p[2 * k], p[2 * k + 1]);
      detail->iteration_info.emplace_back(Pose3d::fromArray(info[it].target_T_source_init), ea, pa,
                                          Pose3d::fromArray(info[it].estimate_update));
    }
    detail->termination_type = static_cast<RegistrationDetail::TerminationType>(result.termination);
  }
  return Pose3d::fromArray(result.pose);
}


/** @brief Extension (not in the reference): the spatial index of a target feature set built once and
 * kept on the device, for scan-to-map registration against a slowly changing local map. The
 * reference rebuilds both KD-trees on every call (registration-inl.h:20-23). */
class TargetIndex {
 public:
  template <template <typename> class Accessor = FieldAccessor, typename PointType, template <typename> class Alloc>
  static TargetIndex build(const LoamFeatures<PointType, Alloc>& target, const RegistrationParams& params = RegistrationParams()) {
    loamx_ctx* ctx = gpu::defaultContext();
    const std::vector<double> te = gpu::pack<Accessor>(target.edge_points), tp = gpu::pack<Accessor>(target.planar_points);
    const loamx_reg_params rp = gpu::toC(params);
    loamx_target_index* h = nullptr;
    gpu::check(ctx, loamx_target_index_create(ctx, te.data(), target.edge_points.size(), tp.data(),
                                              target.planar_points.size(), &rp, &h));
    return TargetIndex(h);
  }
  /// Appends features to the target (a map that grows scan by scan): afterwards the index is the one `build`
  /// gives for the concatenated feature sets.
  template <template <typename> class Accessor = FieldAccessor, typename PointType, template <typename> class Alloc>
  void insert(const LoamFeatures<PointType, Alloc>& more) {
    const std::vector<double> e = gpu::pack<Accessor>(more.edge_points), p = gpu::pack<Accessor>(more.planar_points);
    gpu::check(gpu::defaultContext(), loamx_target_index_insert(gpu::defaultContext(), handle_.get(), e.data(), more.edge_points.size(),
                                                                p.data(), more.planar_points.size()));
  }
  size_t numEdgePoints() const {
    size_t n = 0;
    loamx_target_index_size(handle_.get(), &n, nullptr);
    return n;
  }
  size_t numPlanarPoints() const {
    size_t n = 0;
    loamx_target_index_size(handle_.get(), nullptr, &n);
    return n;
  }
  const loamx_target_index* handle() const { return handle_.get(); }

 private:
  explicit TargetIndex(loamx_target_index* h)
      : handle_(h, [](loamx_target_index* p) { loamx_target_index_destroy(gpu::defaultContext(), p); }) {}
  std::shared_ptr<loamx_target_index> handle_;
};

/// registerFeatures against a prebuilt TargetIndex (params must carry the neighbour radii the index was built with)
template <template <typename> class Accessor = FieldAccessor, typename PointType, template <typename> class Alloc>
Pose3d registerFeatures(const LoamFeatures<PointType, Alloc>& source, const TargetIndex& target,
                        const Pose3d& target_T_source_init, const RegistrationParams& params = RegistrationParams()) {
  loamx_ctx* ctx = gpu::defaultContext();
  const std::vector<double> se = gpu::pack<Accessor>(source.edge_points), sp = gpu::pack<Accessor>(source.planar_points);
  const loamx_reg_params rp = gpu::toC(params);
  double init[7];
  target_T_source_init.toArray(init);
  loamx_reg_result result{};
  gpu::check(ctx, loamx_register_features_indexed(ctx, target.handle(), se.data(), source.edge_points.size(), sp.data(),
                                                  source.planar_points.size(), init, &rp, &result, nullptr));
  return Pose3d::fromArray(result.pose);
}

}  // namespace loam
