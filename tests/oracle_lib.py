"""ctypes binding of the CPU oracle (oracle/libloam_oracle.so). TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(ORACLE_DIR, "libloam_oracle.so")


class FeParams(C.Structure):
    """reference: loam/include/loam/features.h:37-66"""
    _fields_ = [("neighbor_points", C.c_uint64), ("number_sectors", C.c_uint64),
                ("max_edge_feats_per_sector", C.c_uint64), ("max_planar_feats_per_sector", C.c_uint64),
                ("edge_feat_threshold", C.c_double), ("planar_feat_threshold", C.c_double),
                ("occlusion_thresh", C.c_double), ("parallel_thresh", C.c_double)]

    def __init__(self, *args, **kw):
        if not args and not kw:
            args = (3, 6, 10, 50, 100.0, 1.0, 0.5, 1.0)
        super().__init__(*args, **kw)


class RegParams(C.Structure):
    """reference: loam/include/loam/registration.h:40-75"""
    _fields_ = [("num_edge_neighbors", C.c_uint64), ("max_edge_neighbor_dist", C.c_double),
                ("min_line_fit_points", C.c_uint64), ("min_line_condition_number", C.c_double),
                ("num_plane_neighbors", C.c_uint64), ("max_plane_neighbor_dist", C.c_double),
                ("min_plane_fit_points", C.c_uint64), ("max_avg_point_plane_dist", C.c_double),
                ("max_iterations", C.c_uint64), ("rotation_convergence_thresh", C.c_double),
                ("position_convergence_thresh", C.c_double), ("min_associations", C.c_uint64)]

    def __init__(self, *args, **kw):
        if not args and not kw:
            args = (5, 1.0, 3, 10.0, 5, 2.0, 4, 0.1, 10, 1e-3, 1e-2, 100)
        super().__init__(*args, **kw)


class IterInfo(C.Structure):
    _fields_ = [("est_before", C.c_double * 7), ("update", C.c_double * 7), ("n_edge_assoc", C.c_uint64),
                ("n_plane_assoc", C.c_uint64), ("lm_iterations", C.c_uint64), ("lm_successful", C.c_uint64),
                ("initial_cost", C.c_double), ("final_cost", C.c_double)]


CONVERGED, MAX_ITER, INSUFFICIENT_ASSOCIATIONS = 0, 1, 2

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        override = os.environ.get("ORACLE_LIB")  # (tests/test_sanitizers.py: the -fsanitize build)
        if not override and not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(override or _LIB_PATH)
        dp = C.POINTER(C.c_double)
        _lib.oracle_quat_angular_distance.restype = C.c_double
        _lib.oracle_point_to_line_distance.restype = C.c_double
        _lib.oracle_point_to_plane_distance.restype = C.c_double
        _lib.oracle_fit_line.restype = C.c_double
        _lib.oracle_fit_plane.restype = C.c_double
        _lib.oracle_kdtree_build.restype = C.c_void_p
        _lib.oracle_kdtree_build.argtypes = [dp, C.c_size_t]
        _lib.oracle_kdtree_free.argtypes = [C.c_void_p]
        _lib.oracle_knn_search.restype = C.c_size_t
        _lib.oracle_knn_search.argtypes = [C.c_void_p, dp, C.c_size_t, C.c_double, C.POINTER(C.c_uint64)]
        _lib.oracle_knn_bruteforce.restype = C.c_size_t
        _lib.oracle_knn_bruteforce.argtypes = [dp, C.c_size_t, dp, C.c_size_t, C.c_double, C.POINTER(C.c_uint64)]
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def compute_curvature(xyz, scan_lines, points_per_line, params=None):
    params = params or FeParams()
    xyz, p = _d(xyz)
    n = xyz.shape[0] if xyz.ndim == 2 else xyz.size // 3
    out = np.empty(n, dtype=np.float64)
    rc = lib().oracle_compute_curvature(p, C.c_size_t(n), C.c_size_t(scan_lines), C.c_size_t(points_per_line),
                                        C.byref(params), out.ctypes.data_as(C.POINTER(C.c_double)))
    if rc:
        raise RuntimeError("LOAM: provided lidar scan size does not match provided lidar parameters")
    return out


def compute_valid_points(xyz, scan_lines, points_per_line, min_range, max_range, params=None):
    params = params or FeParams()
    xyz, p = _d(xyz)
    n = xyz.shape[0] if xyz.ndim == 2 else xyz.size // 3
    out = np.empty(n, dtype=np.uint8)
    rc = lib().oracle_compute_valid_points(p, C.c_size_t(n), C.c_size_t(scan_lines), C.c_size_t(points_per_line),
                                           C.c_double(min_range), C.c_double(max_range), C.byref(params),
                                           out.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc:
        raise RuntimeError("LOAM: provided lidar scan size does not match provided lidar parameters")
    return out.astype(bool)


def extract_features(xyz, scan_lines, points_per_line, min_range, max_range, params=None, stable=False):
    """Returns (edge_idx, planar_idx[, n_candidate_ties if stable])."""
    params = params or FeParams()
    xyz, p = _d(xyz)
    n = xyz.shape[0] if xyz.ndim == 2 else xyz.size // 3
    cap_e = scan_lines * params.number_sectors * (params.max_edge_feats_per_sector + 1) + 1
    cap_p = scan_lines * params.number_sectors * (params.max_planar_feats_per_sector + 1) + 1
    e = np.empty(cap_e, dtype=np.uint32)
    pl = np.empty(cap_p, dtype=np.uint32)
    ne, npl, ties = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    u32 = C.POINTER(C.c_uint32)
    args = [p, C.c_size_t(n), C.c_size_t(scan_lines), C.c_size_t(points_per_line), C.c_double(min_range),
            C.c_double(max_range), C.byref(params), e.ctypes.data_as(u32), C.byref(ne), pl.ctypes.data_as(u32),
            C.byref(npl)]
    if stable:
        rc = lib().oracle_extract_features_stable(*args, C.byref(ties))
    else:
        rc = lib().oracle_extract_features(*args)
    if rc:
        raise RuntimeError("LOAM: provided lidar scan size does not match provided lidar parameters")
    if stable:
        return e[:ne.value].copy(), pl[:npl.value].copy(), ties.value
    return e[:ne.value].copy(), pl[:npl.value].copy()


def pose_compose(a, b):
    a, pa = _d(a)
    b, pb = _d(b)
    out = np.empty(7)
    lib().oracle_pose_compose(pa, pb, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def pose_inverse(a):
    a, pa = _d(a)
    out = np.empty(7)
    lib().oracle_pose_inverse(pa, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def pose_act(a, p):
    a, pa = _d(a)
    p, pp = _d(p)
    out = np.empty(3)
    lib().oracle_pose_act(pa, pp, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def pose_matrix(a):
    a, pa = _d(a)
    out = np.empty(16)
    lib().oracle_pose_matrix(pa, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out.reshape(4, 4)


def quat_angular_distance(qa, qb):
    qa, pa = _d(qa)
    qb, pb = _d(qb)
    return lib().oracle_quat_angular_distance(pa, pb)


def point_to_line_distance(p, a, b):
    p, pp = _d(p)
    a, pa = _d(a)
    b, pb = _d(b)
    return lib().oracle_point_to_line_distance(pp, pa, pb)


def point_to_plane_distance(p, n, d):
    p, pp = _d(p)
    n, pn = _d(n)
    return lib().oracle_point_to_plane_distance(pp, pn, C.c_double(d))


def fit_line(pts):
    pts, pp = _d(pts)
    out = np.empty(6)
    cond = lib().oracle_fit_line(pp, C.c_size_t(pts.shape[0]), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out[:3].copy(), out[3:].copy(), cond


def fit_plane(pts):
    pts, pp = _d(pts)
    out = np.empty(4)
    avg = lib().oracle_fit_plane(pp, C.c_size_t(pts.shape[0]), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out[:3].copy(), float(out[3]), avg


class KDTree:
    def __init__(self, pts):
        self.pts, pp = _d(pts)
        self.h = lib().oracle_kdtree_build(pp, C.c_size_t(self.pts.shape[0]))

    def knn(self, q, k, max_dist=-1.0):
        q, pq = _d(q)
        out = np.empty(max(k, 1), dtype=np.uint64)
        m = lib().oracle_knn_search(self.h, pq, C.c_size_t(k), C.c_double(max_dist),
                                    out.ctypes.data_as(C.POINTER(C.c_uint64)))
        return out[:m].copy()

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_kdtree_free(self.h)
            self.h = None


def knn_bruteforce(pts, q, k, max_dist=-1.0):
    pts, pp = _d(pts)
    q, pq = _d(q)
    out = np.empty(max(k, 1), dtype=np.uint64)
    m = lib().oracle_knn_bruteforce(pp, C.c_size_t(pts.shape[0]), pq, C.c_size_t(k), C.c_double(max_dist),
                                    out.ctypes.data_as(C.POINTER(C.c_uint64)))
    return out[:m].copy()


def register_features(src_edge, src_planar, tgt_edge, tgt_planar, init_pose=None, params=None, want_info=False):
    """Returns (pose7, termination_type, n_iterations[, iter_info list])."""
    params = params or RegParams()
    init = np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64) if init_pose is None else np.asarray(init_pose, float)
    arrs = []
    ptrs = []
    for a in (src_edge, src_planar, tgt_edge, tgt_planar):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, 3))
        arrs.append(a)
        ptrs.append(a.ctypes.data_as(C.POINTER(C.c_double)))
    init, pi = _d(init)
    out = np.empty(7)
    term = C.c_int(0)
    iters = C.c_uint64(0)
    info = (IterInfo * max(1, params.max_iterations))()
    lib().oracle_register_features(ptrs[0], C.c_size_t(len(arrs[0])), ptrs[1], C.c_size_t(len(arrs[1])), ptrs[2],
                                   C.c_size_t(len(arrs[2])), ptrs[3], C.c_size_t(len(arrs[3])), pi,
                                   C.byref(params), out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(term),
                                   C.byref(iters), info if want_info else None)
    if want_info:
        return out, term.value, iters.value, [info[i] for i in range(iters.value)]
    return out, term.value, iters.value


def associate(src, tgt, est, is_plane, params=None):
    params = params or RegParams()
    src, ps = _d(np.asarray(src, float).reshape(-1, 3))
    tgt, pt = _d(np.asarray(tgt, float).reshape(-1, 3))
    est, pe = _d(est)
    n = len(src)
    pw = 4 if is_plane else 6
    valid = np.zeros(n, dtype=np.uint8)
    nearest = np.zeros(n, dtype=np.uint64)
    moved = np.zeros((n, 3))
    prims = np.zeros((n, pw))
    lib().oracle_associate(ps, C.c_size_t(n), pt, C.c_size_t(len(tgt)), pe, C.c_int(1 if is_plane else 0),
                           C.byref(params), valid.ctypes.data_as(C.POINTER(C.c_uint8)),
                           nearest.ctypes.data_as(C.POINTER(C.c_uint64)),
                           moved.ctypes.data_as(C.POINTER(C.c_double)), prims.ctypes.data_as(C.POINTER(C.c_double)))
    return valid.astype(bool), nearest, moved, prims
