/** @brief Placeholder for the reference's loam/include/loam/kdtree.h.
 * The reference wraps nanoflann (kdtree_internal::KDTree, knnSearch) and uses it only inside
 * registerFeatures. The MI355X back end replaces it by a uniform-grid index built and searched on
 * the GPU (loam_amd/csrc/reg_math.h: knn_search — same contract: exact k nearest, ascending,
 * strict radius filter), so nothing is exported here; the header exists so that
 * `#include "loam/kdtree.h"` and the umbrella header keep compiling.
 */
#pragma once
#include "common.h"
