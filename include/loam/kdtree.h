/** @brief KD-tree helpers of the loam API (drop-in for the reference's loam/include/loam/kdtree.h).
 * The reference wraps nanoflann: kdtree_internal::KDTree over a KDTreeDataAdaptor, searched by knnSearch
 * (kdtree.h:24-49, kdtree.cpp:10-28). The MI355X back end replaces the tree by a uniform-grid index built and searched
 * on the GPU (loam_amd/csrc/reg_math.h: same contract — exact k nearest, ascending, strict radius filter); KDTree here
 * owns such an index (a loamx_target_index whose planar set is the data) and knnSearch queries it through
 * loamx_knn_search. Same names, constructor shape and return type as the reference, so code written against
 * kdtree_internal compiles unchanged. Order among EQUIDISTANT neighbours: nanoflann's is its traversal order
 * (SURVEY Appendix C), here it is ascending index.
 */
#pragma once
#include <memory>
#include <vector>

#include "common.h"

namespace loam {
namespace kdtree_internal {

/// Adaptor over the indexed points (reference kdtree.h:24-35); the interface nanoflann required is kept
struct KDTreeDataAdaptor {
  const std::vector<Vector3d>& data;
  size_t kdtree_get_point_count() const { return data.size(); }
  double kdtree_get_pt(const size_t idx, const size_t dim) const { return data.at(idx)(dim); }
  template <class BBOX>
  bool kdtree_get_bbox(BBOX&) const {
    return false;
  }
  KDTreeDataAdaptor(const std::vector<Vector3d>& data) : data(data) {}
};

/// nanoflann::KDTreeSingleIndexAdaptorParams (reference kdtree.h:41): the leaf size has no meaning for a grid
struct KDTreeParams {
  size_t leaf_max_size;
  explicit KDTreeParams(size_t leaf_max_size = 10) : leaf_max_size(leaf_max_size) {}
};

/// The index over a point set, built in the constructor like the reference's (registration-inl.h:20-23:
/// `KDTree tree(3, adaptor, KDTreeParams(20))`). Device resident; copies share one index.
class KDTree {
 public:
  KDTree(int dimensionality, const KDTreeDataAdaptor& adaptor, const KDTreeParams& = KDTreeParams()) : size_(adaptor.data.size()) {
    if (dimensionality != 3) throw std::runtime_error("loam::kdtree_internal::KDTree: only 3-d point sets");
    std::vector<double> xyz(adaptor.data.size() * 3);
    for (size_t i = 0; i < adaptor.data.size(); i++)
      xyz[3 * i] = adaptor.data[i](0), xyz[3 * i + 1] = adaptor.data[i](1), xyz[3 * i + 2] = adaptor.data[i](2);
    loamx_ctx* ctx = gpu::defaultContext();
    loamx_reg_params rp;
    loamx_default_reg_params(&rp);  // (the radii only choose the cell size; every search is exact for any max_dist)
    loamx_target_index* h = nullptr;
    gpu::check(ctx, loamx_target_index_create(ctx, nullptr, 0, xyz.data(), adaptor.data.size(), &rp, &h));
    handle_ = std::shared_ptr<loamx_target_index>(h, [](loamx_target_index* p) { loamx_target_index_destroy(gpu::defaultContext(), p); });
  }
  size_t size() const { return size_; }
  const loamx_target_index* handle() const { return handle_.get(); }

 private:
  std::shared_ptr<loamx_target_index> handle_;
  size_t size_;
};

/** @brief Radius limited k nearest neighbour search (reference kdtree.h:49, kdtree.cpp:10-28).
 * @param max_dist: if <= 0 no radius limit; otherwise neighbours with sqrt(d^2) < max_dist are kept (strict).
 * @returns indices into the adaptor's data, ascending distance; fewer than k when the set or the radius gives fewer
 * LIMITS (the reference, nanoflann's KNNResultSet(k), has none): k <= 16 — beyond it loamx_knn_search returns
 * LOAMX_ERR_UNSUPPORTED and this throws std::runtime_error; every call is one synchronous host -> device -> host round
 * trip on the shared default context (it takes that context's mutex): the batch entry points (loamx_knn_search with
 * many queries, loamx_associate) are the ones to use in a loop. */
inline std::vector<size_t> knnSearch(const KDTree& tree, const Vector3d& query, const size_t k, const double max_dist = -1) {
  std::vector<uint32_t> idx(k ? k : 1);
  uint32_t count = 0;
  const double q[3] = {query(0), query(1), query(2)};
  gpu::check(gpu::defaultContext(), loamx_knn_search(gpu::defaultContext(), tree.handle(), 1, q, 1, k, max_dist, idx.data(), &count));
  return std::vector<size_t>(idx.begin(), idx.begin() + count);
}

}  // namespace kdtree_internal
}  // namespace loam
