"""Known-answer data of the reference's own unit tests, restated as data (inputs + expected outputs).

Sources (all under /root/reference/tests): test_feature_extraction.cpp, test_geometry.cpp,
test_registration.cpp. Nothing here is read from /root/reference at run time.
"""
import math

import numpy as np

# FeatureExtractionParams used by every feature-extraction KAT (test_feature_extraction.cpp:40 etc.)
KAT_FE_PARAMS = (5, 6, 5, 5, 100.0, 0.1, 0.25, 0.02)


def _line(xs, y, z=0.0):
    return [(float(x), float(y), float(z)) for x in xs]


def fe_kats():
    """List of dicts: name, pts (N x 3), H, W, min_range, max_range, and expected curvature / mask facts."""
    k = []
    # TestCurvaturePlane (:27-53)
    k.append(dict(name="curvature_plane", pts=_line(range(-5, 6), 1.0), H=1, W=11, rmin=0.1, rmax=10.0,
                  curvature={**{i: -1.0 for i in list(range(5)) + list(range(6, 11))}, 5: 0.0}))
    # TestCurvatureCorner (:55-84)
    k.append(dict(name="curvature_corner", pts=[(float(i), float(abs(i) + 1), 0.0) for i in range(-5, 6)], H=1,
                  W=11, rmin=0.1, rmax=50.0,
                  curvature={**{i: -1.0 for i in list(range(5)) + list(range(6, 11))}, 5: 900.0}))
    # TestInvalidEdges (:96-122)
    k.append(dict(name="invalid_edges", pts=[(i * 0.1, 1.0, 0.0) for i in range(-5, 6)], H=1, W=11, rmin=0.1,
                  rmax=50.0, invalid=list(range(5)) + list(range(6, 11)), valid=[5]))
    # TestInvalidRanges (:124-155)
    pts = _line(range(-5, 0), 1.0) + [(-0.5, 20.0, 0.0), (0.0, 0.2, 0.0)] + _line(range(1, 6), 1.0)
    k.append(dict(name="invalid_ranges", pts=pts, H=1, W=12, rmin=0.5, rmax=6.0,
                  invalid=list(range(5)) + [10 - i for i in range(5)] + [5, 6], valid=[]))
    # TestOcclusionCase1 (:157-190)
    pts = [(i * 0.1, 4.0, 0.0) for i in range(-15, 0)] + [(i * 0.1, 6.0, 0.0) for i in range(0, 15)]
    k.append(dict(name="occlusion_case1", pts=pts, H=1, W=30, rmin=0.1, rmax=100.0,
                  invalid=list(range(5)) + list(range(25, 30)) + list(range(15, 20)),
                  valid=list(range(5, 15)) + list(range(20, 25))))
    # TestOcclusionCase2 (:192-225)
    pts = [(i * 0.1, 6.0, 0.0) for i in range(-15, 0)] + [(i * 0.1, 4.0, 0.0) for i in range(0, 15)]
    k.append(dict(name="occlusion_case2", pts=pts, H=1, W=30, rmin=0.1, rmax=100.0,
                  invalid=list(range(5)) + list(range(25, 30)) + list(range(10, 15)),
                  valid=list(range(5, 10)) + list(range(15, 25))))
    # TestParallelPlaneCase1 (:227-262)
    pts = [(i * 0.1, 2.0, 0.0) for i in range(-15, 0)] + [(0.0, 0.0, 2.05)] + [(i * 0.1, 2.1, 0.0) for i in range(1, 16)]
    k.append(dict(name="parallel_case1", pts=pts, H=1, W=31, rmin=0.1, rmax=100.0,
                  invalid=list(range(5)) + list(range(26, 31)) + [15],
                  valid=list(range(5, 15)) + list(range(16, 26))))
    # TestParallelPlaneCase2 (:264-299)
    pts = [(i * 0.1, 2.1, 0.0) for i in range(-15, 0)] + [(0.0, 0.0, 2.05)] + [(i * 0.1, 2.0, 0.0) for i in range(1, 16)]
    k.append(dict(name="parallel_case2", pts=pts, H=1, W=31, rmin=0.1, rmax=100.0,
                  invalid=list(range(5)) + list(range(26, 31)) + [15],
                  valid=list(range(5, 15)) + list(range(16, 26))))
    for d in k:
        d["pts"] = np.asarray(d["pts"], dtype=np.float64)
    return k


def _frange(start, stop, step):
    """Replays `for (double v = start; v < stop; v += step)` in IEEE doubles."""
    v = float(start)
    out = []
    while v < stop:
        out.append(v)
        v += step
    return out


def registration_scene():
    """The feature-level scene of test_registration.cpp:8-56: (edge N x 3, planar N x 3).
    Sizes: planar 61*60 + 60*60 + 41*41 = 8941, edge 81 + 81 = 162."""
    planar = []
    for y in _frange(3, 6, 0.05):
        for z in _frange(-1, 2, 0.05):
            planar.append((-3.0, y, z))
    for x in _frange(-1, 2, 0.05):
        for z in _frange(-1, 2, 0.05):
            planar.append((x, 5.0, z))
    for x in _frange(1, 3, 0.05):
        for y in _frange(1, 3, 0.05):
            planar.append((x, y, -1.0))
    edge = [(-1.0, 4.0, z) for z in _frange(-1, 3, 0.05)] + [(3.0, 2.0, z) for z in _frange(-1, 3, 0.05)]
    return np.asarray(edge, dtype=np.float64), np.asarray(planar, dtype=np.float64)


def plane_only_scene():
    """test_registration.cpp:177-199 NonStandardAllocator: plane x=-3 only, no edges."""
    planar = [(-3.0, y, z) for y in _frange(3, 6, 0.05) for z in _frange(-1, 2, 0.05)]
    return np.zeros((0, 3)), np.asarray(planar, dtype=np.float64)


def quat_wxyz(w, x, y, z):
    """Eigen::Quaterniond(w,x,y,z) -> storage order (x,y,z,w)."""
    return np.array([x, y, z, w], dtype=np.float64)


def quat_angle_axis(angle, axis):
    axis = np.asarray(axis, dtype=np.float64)
    s = math.sin(angle / 2.0)
    return np.array([s * axis[0], s * axis[1], s * axis[2], math.cos(angle / 2.0)], dtype=np.float64)


def pose7(q_xyzw, t):
    return np.concatenate([np.asarray(q_xyzw, float), np.asarray(t, float)])


_Q_SMALL = quat_wxyz(0.9993921140970299, 0.014692022378442412, 0.030140550562090015, 0.009544316157523478)
_AXIS131 = np.array([1.0, 3.0, 1.0]) / math.sqrt(11.0)

# name, source_T_target, init, max_iterations(None=default), rot tol, trans tol  (test_registration.cpp:69-175)
REGISTRATION_CASES = [
    dict(name="simple", source_T_target=pose7(_Q_SMALL, (0.01, 0.03, -0.01)), init=None, max_iter=None,
         rot_tol=1e-4, trans_tol=1e-4),
    dict(name="large_translation", source_T_target=pose7(_Q_SMALL, (-0.1, 0.1, 0.0)), init=None, max_iter=None,
         rot_tol=1e-4, trans_tol=1e-3),
    dict(name="even_larger_translation", source_T_target=pose7(_Q_SMALL, (-0.3, 0.2, 0.1)), init=None,
         max_iter=None, rot_tol=1e-4, trans_tol=1e-3),
    dict(name="large_rotation", source_T_target=pose7(quat_angle_axis(0.2, _AXIS131), (-0.01, 0.02, 0.1)),
         init=None, max_iter=None, rot_tol=1e-4, trans_tol=1e-3),
    dict(name="composition_direction", source_T_target=pose7(quat_angle_axis(0.1, (0, 0, 1)), (0, 0, 0)),
         init=pose7(quat_angle_axis(-0.1, (0, 0, 1)), (0.1, 0, 0)), max_iter=1, rot_tol=1e-4, trans_tol=1e-3),
]

# test_geometry.cpp:31-79 (constants generated with GTSAM by the reference author)
POSE_KATS = dict(
    compose=dict(p1=pose7(quat_wxyz(0.7473257838894183, 0.38405116269438366, -0.17015746936361906,
                                    -0.5148352287741462), (-0.4, 3.0, -8.9)),
                 p2=pose7(quat_wxyz(0.8378767472656409, -0.040374739652255895, -0.40934599608063865,
                                    0.3588429911288663), (4, -5, 1)),
                 expected_q=quat_wxyz(0.7567645973045605, 0.019808900212688513, -0.5655135339985058,
                                      -0.32727571648894294),
                 expected_t=np.array([-2.59584795, -1.87410099, -12.56352171])),
    inverse=dict(p1=pose7(quat_wxyz(0.7473257838894183, 0.38405116269438366, -0.17015746936361906,
                                    -0.5148352287741462), (-0.4, 3.0, -8.9)),
                 expected_q=quat_wxyz(0.7473257838894183, -0.38405116269438366, 0.17015746936361906,
                                      0.5148352287741462),
                 expected_t=np.array([1.60941772, 6.39896027, 6.69575105])),
    matrix=dict(p1=pose7(quat_wxyz(0.9693342323515085, 0.018781217536151106, 0.15609411554196426,
                                   0.18887307630401792), (1.0, -5.0, 2.0)),
                expected=np.array([[0.87992318, -0.360299, 0.30970927, 1.0], [0.37202555, 0.92794845, 0.0225534, -5.0],
                                   [-0.29552021, 0.09537451, 0.95056379, 2.0], [0.0, 0.0, 0.0, 1.0]])),
)


def is_approx(a, b, prec):
    """Eigen isApprox: ||a-b||^2 <= prec^2 * min(||a||^2, ||b||^2)."""
    a = np.asarray(a, float).ravel()
    b = np.asarray(b, float).ravel()
    return np.sum((a - b) ** 2) <= prec * prec * min(np.sum(a * a), np.sum(b * b))


def transform_points(pose, pts):
    """transformFeatures of test_registration.cpp:58-67: q * p + t per point (numpy, not bit-critical)."""
    q = np.asarray(pose[:4], float)
    t = np.asarray(pose[4:], float)
    u = q[:3]
    w = q[3]
    uv = 2.0 * np.cross(u, pts)
    return pts + w * uv + np.cross(u, uv) + t


def registration_error(source_T_target, target_T_source, compose, angdist):
    """err_rot/err_trans exactly as the reference tests compute them (test_registration.cpp:80-81)."""
    err = compose(source_T_target, target_T_source)
    return angdist(err[:4], np.array([0, 0, 0, 1.0])), err[4:]
