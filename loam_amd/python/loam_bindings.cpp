// loam_bindings.cpp — pybind11 module `loam_python` (re-exported as `loam`), same surface as the
// reference's python/loam_bindings.cpp:24-144: LidarParams, Pose3d, Quaterniond,
// FeatureExtractionParams, LoamFeatures, extractFeatures, computeCurvature, computeValidPoints,
// RegistrationParams, RegistrationIterationInfo, RegistrationTerminationType, RegistrationDetail,
// registerFeatures — with the same keyword arguments. Point clouds are contiguous (N,3) float64
// arrays handed to the C ABI without per-point objects (a list of 3-vectors is converted once).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstring>

#include "loam/loam.h"

namespace py = pybind11;
using Arr = py::array_t<double, py::array::c_style | py::array::forcecast>;
// float32 scans (PCL-style sensors data) are taken as they are (no cast: only a C-contiguous float32 array
// matches) and go through the FP32-input entry points of the C ABI
using ArrF = py::array_t<float, py::array::c_style>;

namespace {

struct PyFeatures {
  Arr edge_points;
  Arr planar_points;
  PyFeatures() : edge_points(std::vector<py::ssize_t>{0, 3}), planar_points(std::vector<py::ssize_t>{0, 3}) {}
};

template <typename A>
size_t check_points(const A& a, const char* what) {
  if (a.ndim() == 1 && a.shape(0) == 0) return 0;
  if (a.ndim() != 2 || a.shape(1) != 3) throw std::runtime_error(std::string(what) + ": expected an (N, 3) array of points");
  return (size_t)a.shape(0);
}

template <typename A>
Arr gather_points(const A& scan, const std::vector<uint32_t>& idx, size_t n) {
  Arr out(std::vector<py::ssize_t>{(py::ssize_t)n, 3});
  const auto* src = scan.data();
  double* dst = out.mutable_data();
  for (size_t i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) dst[3 * i + k] = (double)src[3 * (size_t)idx[i] + k];
  return out;
}

// one body for float64 and float32 scans: the C ABI entry points differ only in the scalar type
inline int c_extract(loamx_ctx* c, const double* x, size_t n, const loamx_lidar_params* l, const loamx_fe_params* f, uint32_t* e,
                     size_t ec, size_t* ne, uint32_t* p, size_t pc, size_t* np) {
  return loamx_extract_features(c, x, n, l, f, e, ec, ne, p, pc, np);
}
inline int c_extract(loamx_ctx* c, const float* x, size_t n, const loamx_lidar_params* l, const loamx_fe_params* f, uint32_t* e,
                     size_t ec, size_t* ne, uint32_t* p, size_t pc, size_t* np) {
  return loamx_extract_features_f32(c, x, n, l, f, e, ec, ne, p, pc, np);
}
inline int c_curvature(loamx_ctx* c, const double* x, size_t n, const loamx_lidar_params* l, const loamx_fe_params* f, double* o) {
  return loamx_compute_curvature(c, x, n, l, f, o);
}
inline int c_curvature(loamx_ctx* c, const float* x, size_t n, const loamx_lidar_params* l, const loamx_fe_params* f, double* o) {
  return loamx_compute_curvature_f32(c, x, n, l, f, o);
}
inline int c_valid(loamx_ctx* c, const double* x, size_t n, const loamx_lidar_params* l, const loamx_fe_params* f, uint8_t* o) {
  return loamx_compute_valid_points(c, x, n, l, f, o);
}
inline int c_valid(loamx_ctx* c, const float* x, size_t n, const loamx_lidar_params* l, const loamx_fe_params* f, uint8_t* o) {
  return loamx_compute_valid_points_f32(c, x, n, l, f, o);
}

loam::Vector3d vec_from(const Arr& a) {
  if (a.size() != 3) throw std::runtime_error("expected a 3-vector");
  return loam::Vector3d(a.data()[0], a.data()[1], a.data()[2]);
}
Arr vec_to(const loam::Vector3d& v) {
  Arr out(3);
  for (int i = 0; i < 3; i++) out.mutable_data()[i] = v(i);
  return out;
}

void scan_size_check(size_t n, const loam::LidarParams& lp) {
  if (n != lp.scan_lines * lp.points_per_line) {
    std::stringstream msg;
    msg << "LOAM: provided lidar scan size ( " << n << ")  does not match provided lidar parameters (" << lp.scan_lines
        << " x " << lp.points_per_line << ")";
    throw std::runtime_error(msg.str());
  }
}

template <typename A>
PyFeatures extract_features(const A& scan, const loam::LidarParams& lp, const loam::FeatureExtractionParams& params) {
  const size_t n = check_points(scan, "input_scan");
  scan_size_check(n, lp);
  PyFeatures out;
  if (n == 0) return out;
  loamx_ctx* ctx = loam::gpu::defaultContext();
  const loamx_lidar_params clp = loam::gpu::toC(lp);
  const loamx_fe_params cfp = loam::gpu::toC(params);
  std::vector<uint32_t> e(loamx_edge_capacity(&clp, &cfp) + 1), p(loamx_planar_capacity(&clp, &cfp) + 1);
  size_t ne = 0, np = 0;
  {
    py::gil_scoped_release release;
    loam::gpu::check(ctx, c_extract(ctx, scan.data(), n, &clp, &cfp, e.data(), e.size(), &ne, p.data(), p.size(), &np));
  }
  out.edge_points = gather_points(scan, e, ne);
  out.planar_points = gather_points(scan, p, np);
  return out;
}

template <typename A>
Arr compute_curvature(const A& scan, const loam::LidarParams& lp, const loam::FeatureExtractionParams& params) {
  const size_t n = check_points(scan, "input_scan");
  scan_size_check(n, lp);
  Arr out((py::ssize_t)n);
  if (n == 0) return out;
  loamx_ctx* ctx = loam::gpu::defaultContext();
  const loamx_lidar_params clp = loam::gpu::toC(lp);
  const loamx_fe_params cfp = loam::gpu::toC(params);
  loam::gpu::check(ctx, c_curvature(ctx, scan.data(), n, &clp, &cfp, out.mutable_data()));
  return out;
}

template <typename A>
py::array_t<bool> compute_valid_points(const A& scan, const loam::LidarParams& lp, const loam::FeatureExtractionParams& params) {
  const size_t n = check_points(scan, "input_scan");
  scan_size_check(n, lp);
  py::array_t<bool> out((py::ssize_t)n);
  if (n == 0) return out;
  loamx_ctx* ctx = loam::gpu::defaultContext();
  const loamx_lidar_params clp = loam::gpu::toC(lp);
  const loamx_fe_params cfp = loam::gpu::toC(params);
  static_assert(sizeof(bool) == 1, "bool must be one byte");
  loam::gpu::check(ctx, c_valid(ctx, scan.data(), n, &clp, &cfp, reinterpret_cast<uint8_t*>(out.mutable_data())));
  return out;
}

}  // namespace

PYBIND11_MODULE(loam_python, m) {
  m.doc() = "loam (MI355X back end): LOAM feature extraction and registration";

  py::class_<loam::LidarParams>(m, "LidarParams")
      .def(py::init<size_t, size_t, double, double>(), py::arg("scan_lines"), py::arg("points_per_line"),
           py::arg("min_range"), py::arg("max_range"))
      .def_readonly("scan_lines", &loam::LidarParams::scan_lines)
      .def_readonly("points_per_line", &loam::LidarParams::points_per_line)
      .def_readonly("min_range", &loam::LidarParams::min_range)
      .def_readonly("max_range", &loam::LidarParams::max_range);

  py::class_<loam::Quaterniond>(m, "Quaterniond")
      .def(py::init<double, double, double, double>(), py::arg("w"), py::arg("x"), py::arg("y"), py::arg("z"))
      .def("w", [](const loam::Quaterniond& q) { return q.w(); })
      .def("x", [](const loam::Quaterniond& q) { return q.x(); })
      .def("y", [](const loam::Quaterniond& q) { return q.y(); })
      .def("z", [](const loam::Quaterniond& q) { return q.z(); });

  py::class_<loam::Pose3d>(m, "Pose3d")
      .def(py::init([](const loam::Quaterniond& q, const Arr& t) { return loam::Pose3d(q, vec_from(t)); }),
           py::arg("rotation"), py::arg("translation"))
      .def_static("Identity", &loam::Pose3d::Identity)
      .def("inverse", &loam::Pose3d::inverse)
      .def("compose", &loam::Pose3d::compose, py::arg("other"))
      .def("act", [](const loam::Pose3d& p, const Arr& pt) { return vec_to(p.act(vec_from(pt))); }, py::arg("point"))
      .def("matrix",
           [](const loam::Pose3d& p) {
             const loam::Matrix4d mm = p.matrix();
             Arr out(std::vector<py::ssize_t>{4, 4});
             for (int i = 0; i < 4; i++)
               for (int j = 0; j < 4; j++) out.mutable_at(i, j) = mm(i, j);
             return out;
           })
      .def_readwrite("rotation", &loam::Pose3d::rotation)
      .def_property(
          "translation", [](const loam::Pose3d& p) { return vec_to(p.translation); },
          [](loam::Pose3d& p, const Arr& t) { p.translation = vec_from(t); });

  py::class_<loam::FeatureExtractionParams>(m, "FeatureExtractionParams")
      .def(py::init<>())
      .def_readwrite("neighbor_points", &loam::FeatureExtractionParams::neighbor_points)
      .def_readwrite("number_sectors", &loam::FeatureExtractionParams::number_sectors)
      .def_readwrite("max_edge_feats_per_sector", &loam::FeatureExtractionParams::max_edge_feats_per_sector)
      .def_readwrite("max_planar_feats_per_sector", &loam::FeatureExtractionParams::max_planar_feats_per_sector)
      .def_readwrite("edge_feat_threshold", &loam::FeatureExtractionParams::edge_feat_threshold)
      .def_readwrite("planar_feat_threshold", &loam::FeatureExtractionParams::planar_feat_threshold)
      .def_readwrite("occlusion_thresh", &loam::FeatureExtractionParams::occlusion_thresh)
      .def_readwrite("parallel_thresh", &loam::FeatureExtractionParams::parallel_thresh);

  py::class_<PyFeatures>(m, "LoamFeatures")
      .def(py::init<>())
      .def_readwrite("edge_points", &PyFeatures::edge_points)
      .def_readwrite("planar_points", &PyFeatures::planar_points);

  m.def("extractFeatures", &extract_features<ArrF>, py::arg("input_scan"), py::arg("lidar_params"),
        py::arg("params") = loam::FeatureExtractionParams());
  m.def("extractFeatures", &extract_features<Arr>, py::arg("input_scan"), py::arg("lidar_params"),
        py::arg("params") = loam::FeatureExtractionParams());
  m.def("computeCurvature", &compute_curvature<ArrF>, py::arg("input_scan"), py::arg("lidar_params"),
        py::arg("params") = loam::FeatureExtractionParams());
  m.def("computeCurvature", &compute_curvature<Arr>, py::arg("input_scan"), py::arg("lidar_params"),
        py::arg("params") = loam::FeatureExtractionParams());
  m.def("computeValidPoints", &compute_valid_points<ArrF>, py::arg("input_scan"), py::arg("lidar_params"),
        py::arg("params") = loam::FeatureExtractionParams());
  m.def("computeValidPoints", &compute_valid_points<Arr>, py::arg("input_scan"), py::arg("lidar_params"),
        py::arg("params") = loam::FeatureExtractionParams());

  py::class_<loam::RegistrationParams>(m, "RegistrationParams")
      .def(py::init<>())
      .def_readwrite("num_edge_neighbors", &loam::RegistrationParams::num_edge_neighbors)
      .def_readwrite("max_edge_neighbor_dist", &loam::RegistrationParams::max_edge_neighbor_dist)
      .def_readwrite("min_line_fit_points", &loam::RegistrationParams::min_line_fit_points)
      .def_readwrite("min_line_condition_number", &loam::RegistrationParams::min_line_condition_number)
      .def_readwrite("num_plane_neighbors", &loam::RegistrationParams::num_plane_neighbors)
      .def_readwrite("max_plane_neighbor_dist", &loam::RegistrationParams::max_plane_neighbor_dist)
      .def_readwrite("min_plane_fit_points", &loam::RegistrationParams::min_plane_fit_points)
      .def_readwrite("max_avg_point_plane_dist", &loam::RegistrationParams::max_avg_point_plane_dist)
      .def_readwrite("max_iterations", &loam::RegistrationParams::max_iterations)
      .def_readwrite("rotation_convergence_thresh", &loam::RegistrationParams::rotation_convergence_thresh)
      .def_readwrite("position_convergence_thresh", &loam::RegistrationParams::position_convergence_thresh)
      .def_readwrite("min_associations", &loam::RegistrationParams::min_associations);

  py::class_<loam::RegistrationDetail::IterationInfo>(m, "RegistrationIterationInfo")
      .def(py::init<const loam::Pose3d, const std::vector<std::pair<size_t, size_t>>,
                    const std::vector<std::pair<size_t, size_t>>, const loam::Pose3d>(),
           py::arg("target_T_source_init"), py::arg("edge_associations"), py::arg("plane_associations"),
           py::arg("estimate_update"))
      .def_readwrite("target_T_source_init", &loam::RegistrationDetail::IterationInfo::target_T_source_init)
      .def_readwrite("edge_associations", &loam::RegistrationDetail::IterationInfo::edge_associations)
      .def_readwrite("plane_associations", &loam::RegistrationDetail::IterationInfo::plane_associations)
      .def_readwrite("estimate_update", &loam::RegistrationDetail::IterationInfo::estimate_update);

  py::enum_<loam::RegistrationDetail::TerminationType>(m, "RegistrationTerminationType")
      .value("CONVERGED", loam::RegistrationDetail::TerminationType::CONVERGED)
      .value("MAX_ITER", loam::RegistrationDetail::TerminationType::MAX_ITER)
      .value("INSUFFICIENT_ASSOCIATIONS", loam::RegistrationDetail::TerminationType::INSUFFICIENT_ASSOCIATIONS)
      .export_values();

  py::class_<loam::RegistrationDetail, std::shared_ptr<loam::RegistrationDetail>>(m, "RegistrationDetail")
      .def(py::init<>())
      .def_readwrite("iteration_info", &loam::RegistrationDetail::iteration_info)
      .def_readwrite("termination_type", &loam::RegistrationDetail::termination_type);

  m.def(
      "registerFeatures",
      [](const PyFeatures& source, const PyFeatures& target, const loam::Pose3d& init,
         const loam::RegistrationParams& params, std::shared_ptr<loam::RegistrationDetail> detail) {
        // wrap the (N,3) arrays as feature sets of row views without copying the coordinates twice
        struct Row {
          const double* p;
          double operator()(int i) const { return p[i]; }
        };
        auto rows = [](const Arr& a, const char* what) {
          const size_t n = check_points(a, what);
          std::vector<Row> r(n);
          for (size_t i = 0; i < n; i++) r[i] = Row{a.data() + 3 * i};
          return r;
        };
        loam::LoamFeatures<Row> s, t;
        s.edge_points = rows(source.edge_points, "source.edge_points");
        s.planar_points = rows(source.planar_points, "source.planar_points");
        t.edge_points = rows(target.edge_points, "target.edge_points");
        t.planar_points = rows(target.planar_points, "target.planar_points");
        py::gil_scoped_release release;
        return loam::registerFeatures<loam::ParenAccessor>(s, t, init, params, detail);
      },
      py::arg("source"), py::arg("target"), py::arg("target_T_source_init"),
      py::arg("params") = loam::RegistrationParams(), py::arg("detail") = std::shared_ptr<loam::RegistrationDetail>());
}
