// test_shim.cpp — the reference's own unit tests (tests/test_feature_extraction.cpp,
// test_geometry.cpp, test_registration.cpp of DanMcGann/loam) re-expressed against the drop-in C++
// headers in include/loam/, i.e. through the C ABI on the MI355X. Same inputs, same expected values,
// same tolerances; a tiny CHECK macro stands in for gtest (not installed in this image).
// Built and run by tests/test_gpu_cpp_shim.py (needs a GPU).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>
#include <memory>
#include <vector>

#include "loam/loam.h"

using namespace loam;

static int g_failures = 0, g_checks = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    g_checks++;                                                            \
    if (!(cond)) {                                                         \
      g_failures++;                                                        \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);          \
    }                                                                      \
  } while (0)
#define CHECK_NEAR(a, b, tol) CHECK(std::fabs((a) - (b)) <= (tol))

struct Point {
  double x, y, z;
  Point(double x, double y, double z) : x(x), y(y), z(z) {}
};

static const FeatureExtractionParams kKatParams{5, 6, 5, 5, 100, 0.1, 0.25, 0.02};

static void test_curvature() {
  {  // TestCurvaturePlane
    std::vector<Point> pcd;
    for (int i = -5; i <= 5; i++) pcd.push_back(Point(i, 1, 0.0));
    LidarParams lp(1, 11, 0.1, 10);
    auto curv = computeCurvature(pcd, lp, kKatParams);
    CHECK(curv.size() == 11);
    for (size_t i = 0; i < 5; i++) {
      CHECK_NEAR(curv[i].curvature, -1, 1e-9);
      CHECK_NEAR(curv[10 - i].curvature, -1, 1e-9);
    }
    CHECK_NEAR(curv[5].curvature, 0.0, 1e-9);
    CHECK(curv[7].index == 7);
  }
  {  // TestCurvatureCorner
    std::vector<Point> pcd;
    for (int i = -5; i <= 5; i++) pcd.push_back(Point(i, std::abs(i) + 1, 0.0));
    LidarParams lp(1, 11, 0.1, 50);
    std::vector<PointCurvature> curv = computeCurvature(pcd, lp, kKatParams);
    CHECK_NEAR(curv[5].curvature, 900.0, 1e-9);
  }
}

static void test_valid_points() {
  {  // TestInvalidEdges
    std::vector<Point> pcd;
    for (int i = -5; i <= 5; i++) pcd.push_back(Point(i * 0.1, 1, 0.0));
    std::vector<bool> m = computeValidPoints(pcd, LidarParams(1, 11, 0.1, 50), kKatParams);
    CHECK(m.size() == 11);
    for (size_t i = 0; i < 5; i++) CHECK(!m[i] && !m[10 - i]);
    CHECK(m[5]);
  }
  {  // TestInvalidRanges
    std::vector<Point> pcd;
    for (int i = -5; i < 0; i++) pcd.push_back(Point(i, 1, 0.0));
    pcd.push_back(Point(-0.5, 20.0, 0.0));
    pcd.push_back(Point(0.0, 0.2, 0.0));
    for (int i = 1; i <= 5; i++) pcd.push_back(Point(i, 1, 0.0));
    std::vector<bool> m = computeValidPoints(pcd, LidarParams(1, 12, 0.5, 6.0), kKatParams);
    CHECK(m.size() == 12);
    CHECK(!m[5] && !m[6]);
  }
  {  // TestOcclusionCase1 / Case2
    std::vector<Point> a, b;
    for (int i = -15; i < 0; i++) a.push_back(Point(i * 0.1, 4.0, 0.0)), b.push_back(Point(i * 0.1, 6.0, 0.0));
    for (int i = 0; i < 15; i++) a.push_back(Point(i * 0.1, 6.0, 0.0)), b.push_back(Point(i * 0.1, 4.0, 0.0));
    LidarParams lp(1, 30, 0.1, 100);
    std::vector<bool> m = computeValidPoints(a, lp, kKatParams);
    for (size_t i = 5; i < 15; i++) CHECK(m[i]);
    for (size_t i = 15; i < 20; i++) CHECK(!m[i]);
    for (size_t i = 20; i < 25; i++) CHECK(m[i]);
    m = computeValidPoints(b, lp, kKatParams);
    for (size_t i = 5; i < 10; i++) CHECK(m[i]);
    for (size_t i = 10; i < 15; i++) CHECK(!m[i]);
    for (size_t i = 15; i < 25; i++) CHECK(m[i]);
  }
  for (int variant = 0; variant < 2; variant++) {  // TestParallelPlaneCase1 / Case2
    std::vector<Point> pcd;
    for (int i = -15; i < 0; i++) pcd.push_back(Point(i * 0.1, variant ? 2.1 : 2.0, 0.0));
    pcd.push_back(Point(0, 0, 2.05));
    for (int i = 1; i <= 15; i++) pcd.push_back(Point(i * 0.1, variant ? 2.0 : 2.1, 0.0));
    std::vector<bool> m = computeValidPoints(pcd, LidarParams(1, 31, 0.1, 100), kKatParams);
    for (size_t i = 5; i < 15; i++) CHECK(m[i]);
    for (size_t i = 16; i < 26; i++) CHECK(m[i]);
    CHECK(!m[15]);
  }
}

// PCL-style points (float fields, FieldAccessor) take the FP32-input path; the result must be the
// one the FP64 path gives on the widened coordinates.
struct PointF {
  float x, y, z;
  float intensity;
};
static_assert(gpu::float_scan_v<FieldAccessor, PointF>, "float-field points must select the FP32-input path");
static_assert(!gpu::float_scan_v<FieldAccessor, Point>, "double-field points must not");
static_assert(!gpu::float_scan_v<ParenAccessor, PointF>, "only FieldAccessor reads the fields directly");

static void test_float_points() {
  const size_t H = 8, W = 256;
  std::vector<PointF> pf;
  std::vector<Point> pd;
  for (size_t l = 0; l < H; l++) {
    for (size_t c = 0; c < W; c++) {
      const double az = 2.0 * 3.14159265358979323846 * (double)c / (double)W, el = -0.2 + 0.05 * (double)l;
      // a square room: range jumps at the corners give edges, walls give planar points
      const double ca = std::cos(az), sa = std::sin(az);
      const double r = 5.0 / std::fmax(std::fabs(ca), std::fabs(sa)) + ((c * 2654435761u + l * 40503u) % 1000) * 1e-5;
      PointF p{(float)(r * ca * std::cos(el)), (float)(r * sa * std::cos(el)), (float)(r * std::sin(el)), 1.0f};
      pf.push_back(p);
      pd.push_back(Point((double)p.x, (double)p.y, (double)p.z));
    }
  }
  const LidarParams lp(H, W, 0.5, 100.0);
  const FeatureExtractionParams params{3, 6, 10, 50, 0.01, 1.0, 0.5, 1.0};
  const auto ff = extractFeatures(pf, lp, params);
  const auto fd = extractFeatures(pd, lp, params);
  CHECK(ff.edge_points.size() == fd.edge_points.size());
  CHECK(ff.planar_points.size() == fd.planar_points.size());
  CHECK(!ff.planar_points.empty());
  bool same = ff.edge_points.size() == fd.edge_points.size() && ff.planar_points.size() == fd.planar_points.size();
  for (size_t i = 0; same && i < ff.edge_points.size(); i++) same = (double)ff.edge_points[i].x == fd.edge_points[i].x && (double)ff.edge_points[i].y == fd.edge_points[i].y;
  for (size_t i = 0; same && i < ff.planar_points.size(); i++) same = (double)ff.planar_points[i].x == fd.planar_points[i].x && (double)ff.planar_points[i].z == fd.planar_points[i].z;
  CHECK(same);
  const auto cf = computeCurvature(pf, lp, params);
  const auto cd = computeCurvature(pd, lp, params);
  bool curv_same = cf.size() == cd.size();
  for (size_t i = 0; curv_same && i < cf.size(); i++) curv_same = cf[i].curvature == cd[i].curvature;
  CHECK(curv_same);
  CHECK(computeValidPoints(pf, lp, params) == computeValidPoints(pd, lp, params));
}

static void test_errors_and_empty() {
  std::vector<Vector3d> empty;
  LoamFeatures<Vector3d> out = extractFeatures<ParenAccessor>(empty, LidarParams(0, 0, 0.1, 100), kKatParams);
  CHECK(out.edge_points.empty() && out.planar_points.empty());
  std::vector<Point> pcd(10, Point(1, 1, 1));
  bool threw = false;
  try {
    computeCurvature(pcd, LidarParams(1, 11, 0.1, 10));
  } catch (const std::runtime_error& e) {
    threw = std::string(e.what()).find("does not match provided lidar parameters (1 x 11)") != std::string::npos;
  }
  CHECK(threw);
}

static void test_pose() {
  {  // TestCopyConstructor
    Pose3d pa;
    Pose3d pb(pa);
    pa.rotation.x() = 1;
    pa.translation(0) = 1;
    CHECK_NEAR(pb.rotation.x(), 0.0, 1e-12);
    CHECK_NEAR(pb.translation(0), 0.0, 1e-12);
    CHECK_NEAR(pa.rotation.x(), 1.0, 1e-12);
  }
  {  // TestCompose / TestInverse (constants generated with GTSAM by the reference author)
    Quaterniond q1(0.7473257838894183, 0.38405116269438366, -0.17015746936361906, -0.5148352287741462);
    Quaterniond q2(0.8378767472656409, -0.040374739652255895, -0.40934599608063865, 0.3588429911288663);
    Pose3d p1(q1, Vector3d(-0.4, 3., -8.9)), p2(q2, Vector3d(4, -5, 1));
    Pose3d comp = p1.compose(p2);
    CHECK(comp.rotation.isApprox(Quaterniond(0.7567645973045605, 0.019808900212688513, -0.5655135339985058, -0.32727571648894294), 1e-8));
    CHECK(comp.translation.isApprox(Vector3d(-2.59584795, -1.87410099, -12.56352171), 1e-8));
    Pose3d inv = p1.inverse();
    CHECK(inv.rotation.isApprox(Quaterniond(0.7473257838894183, -0.38405116269438366, 0.17015746936361906, 0.5148352287741462), 1e-8));
    CHECK(inv.translation.isApprox(Vector3d(1.60941772, 6.39896027, 6.69575105), 1e-8));
  }
  {  // TestMatrix
    Pose3d p1(Quaterniond(0.9693342323515085, 0.018781217536151106, 0.15609411554196426, 0.18887307630401792), Vector3d(1., -5., 2.));
    Matrix4d mat = p1.matrix();
    const double expected[4][4] = {{0.87992318, -0.360299, 0.30970927, 1.}, {0.37202555, 0.92794845, 0.0225534, -5.},
                                   {-0.29552021, 0.09537451, 0.95056379, 2.}, {0., 0., 0., 1.}};
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) CHECK_NEAR(mat(i, j), expected[i][j], 1e-5);
  }
  for (double x = -5; x < 5; x += 0.5)  // TestPoint2Line / TestPoint2Plane
    for (double y = -5; y < 5; y += 0.5) {
      Vector3d p(x, y, x + y);
      CHECK_NEAR(geometry_internal::pointToLineDistance(p, Vector3d(0, 0, 0), Vector3d(0, 0, 1)), std::sqrt(x * x + y * y), 1e-8);
      CHECK_NEAR(geometry_internal::pointToPlaneDistance(p, Vector3d(1, 0, 0), 2.25), std::fabs(x - 2.25), 1e-8);
    }
}

static LoamFeatures<Vector3d> constructSimpleScene() {
  LoamFeatures<Vector3d> r;
  for (double y = 3; y < 6; y += 0.05)
    for (double z = -1; z < 2; z += 0.05) r.planar_points.push_back(Vector3d(-3, y, z));
  for (double x = -1; x < 2; x += 0.05)
    for (double z = -1; z < 2; z += 0.05) r.planar_points.push_back(Vector3d(x, 5, z));
  for (double x = 1; x < 3; x += 0.05)
    for (double y = 1; y < 3; y += 0.05) r.planar_points.push_back(Vector3d(x, y, -1));
  for (double z = -1; z < 3; z += 0.05) r.edge_points.push_back(Vector3d(-1, 4, z));
  for (double z = -1; z < 3; z += 0.05) r.edge_points.push_back(Vector3d(3, 2, z));
  return r;
}
static LoamFeatures<Vector3d> transformFeatures(const LoamFeatures<Vector3d>& in, const Pose3d& t) {
  LoamFeatures<Vector3d> r;
  for (const auto& p : in.planar_points) r.planar_points.push_back(t.rotation * p + t.translation);
  for (const auto& p : in.edge_points) r.edge_points.push_back(t.rotation * p + t.translation);
  return r;
}
static Quaterniond angleAxis(double angle, Vector3d axis) {
  const double s = std::sin(angle / 2);
  return Quaterniond(std::cos(angle / 2), s * axis(0), s * axis(1), s * axis(2));
}
static void checkRegistration(const Pose3d& source_T_target, const Pose3d& init, const RegistrationParams& params,
                              double rot_tol, double trans_tol, std::shared_ptr<RegistrationDetail> detail = nullptr) {
  LoamFeatures<Vector3d> target = constructSimpleScene();
  CHECK(target.planar_points.size() == 8941 && target.edge_points.size() == 162);
  LoamFeatures<Vector3d> source = transformFeatures(target, source_T_target);
  Pose3d target_T_source = registerFeatures<ParenAccessor>(source, target, init, params, detail);
  Quaterniond err_rot = source_T_target.rotation * target_T_source.rotation;
  Vector3d err_trans = source_T_target.rotation * target_T_source.translation + source_T_target.translation;
  CHECK_NEAR(err_rot.angularDistance(Quaterniond::Identity()), 0.0, rot_tol);
  for (int i = 0; i < 3; i++) CHECK_NEAR(err_trans(i), 0.0, trans_tol);
}

static void test_registration() {
  const Quaterniond q(0.9993921140970299, 0.014692022378442412, 0.030140550562090015, 0.009544316157523478);
  auto detail = std::make_shared<RegistrationDetail>();
  checkRegistration(Pose3d(q, Vector3d(0.01, 0.03, -0.01)), Pose3d(), RegistrationParams(), 1e-4, 1e-4, detail);  // TestSimpleCase
  CHECK(detail->termination_type == RegistrationDetail::CONVERGED);
  CHECK(!detail->iteration_info.empty());
  if (!detail->iteration_info.empty()) {
    CHECK(detail->iteration_info[0].edge_associations.size() == 162);
    CHECK(detail->iteration_info[0].plane_associations.size() == 8941);
  }
  checkRegistration(Pose3d(q, Vector3d(-0.1, 0.1, 0.0)), Pose3d(), RegistrationParams(), 1e-4, 1e-3);  // LargeTranslation
  checkRegistration(Pose3d(q, Vector3d(-0.3, 0.2, 0.1)), Pose3d(), RegistrationParams(), 1e-4, 1e-3);  // EvenLargerTranslation
  Vector3d axis(1, 3, 1);
  checkRegistration(Pose3d(angleAxis(0.2, axis / axis.norm()), Vector3d(-0.01, 0.02, 0.1)), Pose3d(), RegistrationParams(), 1e-4,
                    1e-3);  // LargeRotation
  RegistrationParams one;
  one.max_iterations = 1;
  checkRegistration(Pose3d(angleAxis(0.1, Vector3d(0, 0, 1)), Vector3d::Zero()),
                    Pose3d(angleAxis(-0.1, Vector3d(0, 0, 1)), Vector3d(0.1, 0, 0)), one, 1e-4, 1e-3);  // CompositionDirection
  {  // extension: persistent target index gives the same pose as the plain call
    LoamFeatures<Vector3d> target = constructSimpleScene();
    const Pose3d sTt(q, Vector3d(-0.1, 0.1, 0.0));
    LoamFeatures<Vector3d> source = transformFeatures(target, sTt);
    const Pose3d plain = registerFeatures<ParenAccessor>(source, target, Pose3d());
    TargetIndex index = TargetIndex::build<ParenAccessor>(target);
    const Pose3d viaIndex = registerFeatures<ParenAccessor>(source, index, Pose3d());
    for (int i = 0; i < 3; i++) CHECK(plain.translation(i) == viaIndex.translation(i));
    CHECK(plain.rotation.w() == viaIndex.rotation.w());
    // ... and so does an index grown in two steps (first half of every feature set, then the rest)
    LoamFeatures<Vector3d> first, rest;
    for (size_t i = 0; i < target.edge_points.size(); i++) (i < target.edge_points.size() / 2 ? first : rest).edge_points.push_back(target.edge_points[i]);
    for (size_t i = 0; i < target.planar_points.size(); i++) (i < target.planar_points.size() / 2 ? first : rest).planar_points.push_back(target.planar_points[i]);
    TargetIndex grown = TargetIndex::build<ParenAccessor>(first);
    grown.insert<ParenAccessor>(rest);
    CHECK(grown.numEdgePoints() == target.edge_points.size() && grown.numPlanarPoints() == target.planar_points.size());
    const Pose3d viaGrown = registerFeatures<ParenAccessor>(source, grown, Pose3d());
    for (int i = 0; i < 3; i++) CHECK(plain.translation(i) == viaGrown.translation(i));
    CHECK(plain.rotation.w() == viaGrown.rotation.w());
  }
  {  // NonStandardAllocator: planar only, self registration
    LoamFeatures<Vector3d> t;
    for (double y = 3; y < 6; y += 0.05)
      for (double z = -1; z < 2; z += 0.05) t.planar_points.push_back(Vector3d(-3, y, z));
    Pose3d r = registerFeatures<ParenAccessor>(t, t, Pose3d());
    CHECK_NEAR(r.rotation.angularDistance(Quaterniond::Identity()), 0.0, 1e-4);
    for (int i = 0; i < 3; i++) CHECK_NEAR(r.translation(i), 0.0, 1e-3);
  }
}

// geometry_internal::fitLine / fitPlane (reference geometry.h:102, :123) and kdtree_internal::KDTree / knnSearch
// (kdtree.h:24-49) through the shim: the reference's tests never call them directly, so the expectations here are
// analytic (points on a known line / plane) and a brute-force search.
static void test_internal_namespaces() {
  {
    std::vector<Vector3d> pts;
    const Vector3d o(1.0, -2.0, 0.5), dir(2.0 / 3.0, -1.0 / 3.0, 2.0 / 3.0);
    for (double t : {-0.4, -0.1, 0.0, 0.3, 0.7}) pts.push_back(o + dir * t);
    const auto [line, cond] = geometry_internal::fitLine(pts);
    CHECK(cond == std::numeric_limits<double>::max());  // geometry.cpp:55-56: always DBL_MAX
    const Vector3d d = line.a - line.b;
    CHECK_NEAR(d.norm(), 0.2, 1e-12);  // centre +- 0.1 dir (geometry.cpp:53)
    CHECK_NEAR(std::fabs(d.dot(dir)) / d.norm(), 1.0, 1e-12);
    for (const auto& p : pts) CHECK_NEAR(geometry_internal::pointToLineDistance(p, line.a, line.b), 0.0, 1e-12);
  }
  {
    std::vector<Vector3d> pts;
    const Vector3d n(0.6, 0.0, 0.8);  // plane n.p = 2.5
    for (double u : {-1.0, 0.2, 0.9})
      for (double v : {-0.5, 0.6}) pts.push_back(n * 2.5 + Vector3d(0.8, 0, -0.6) * u + Vector3d(0, 1, 0) * v);
    pts.resize(5);
    const auto [plane, avg] = geometry_internal::fitPlane(pts);
    CHECK_NEAR(plane.d, 2.5, 1e-12);
    for (int i = 0; i < 3; i++) CHECK_NEAR(plane.normal(i), n(i), 1e-12);
    CHECK_NEAR(avg, 0.0, 1e-12);
  }
  {
    std::vector<Vector3d> data;
    unsigned s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / 16777216.0 * 10.0 - 5.0; };
    for (int i = 0; i < 3000; i++) data.push_back(Vector3d(rnd(), rnd(), rnd()));
    kdtree_internal::KDTreeDataAdaptor adaptor(data);
    kdtree_internal::KDTree tree(3, adaptor, kdtree_internal::KDTreeParams(20));  // registration-inl.h:20-23
    for (int q = 0; q < 50; q++) {
      const Vector3d query(rnd(), rnd(), rnd());
      for (double radius : {-1.0, 0.6}) {
        const std::vector<size_t> got = kdtree_internal::knnSearch(tree, query, 5, radius);
        std::vector<std::pair<double, size_t>> all;
        for (size_t i = 0; i < data.size(); i++) all.emplace_back((data[i] - query).squaredNorm(), i);
        std::sort(all.begin(), all.end());
        size_t want = 0;
        while (want < 5 && (radius <= 0 || std::sqrt(all[want].first) < radius)) want++;
        CHECK(got.size() == want);
        for (size_t j = 0; j < got.size() && j < want; j++) CHECK(got[j] == all[j].second);
      }
    }
    std::vector<Vector3d> none;
    kdtree_internal::KDTreeDataAdaptor empty_adaptor(none);
    kdtree_internal::KDTree empty(3, empty_adaptor, kdtree_internal::KDTreeParams(20));
    CHECK(kdtree_internal::knnSearch(empty, Vector3d(0, 0, 0), 5, 1.0).empty());
  }
}

int main() {
  test_internal_namespaces();
  test_curvature();
  test_valid_points();
  test_float_points();
  test_errors_and_empty();
  test_pose();
  test_registration();
  std::printf("%d checks, %d failures\n", g_checks, g_failures);
  return g_failures ? 1 : 0;
}
