"""ctypes binding of tests/hostcheck/libhostcheck.so — TEST-ONLY serial CPU emulation of the HIP
kernels' per-thread math (same headers the kernels include). Never used by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "hostcheck")
_lib = None


class FeParams(C.Structure):
    _fields_ = [("neighbor_points", C.c_uint64), ("number_sectors", C.c_uint64),
                ("max_edge_feats_per_sector", C.c_uint64), ("max_planar_feats_per_sector", C.c_uint64),
                ("edge_feat_threshold", C.c_double), ("planar_feat_threshold", C.c_double),
                ("occlusion_thresh", C.c_double), ("parallel_thresh", C.c_double)]


class RegParams(C.Structure):
    _fields_ = [("num_edge_neighbors", C.c_uint64), ("max_edge_neighbor_dist", C.c_double),
                ("min_line_fit_points", C.c_uint64), ("min_line_condition_number", C.c_double),
                ("num_plane_neighbors", C.c_uint64), ("max_plane_neighbor_dist", C.c_double),
                ("min_plane_fit_points", C.c_uint64), ("max_avg_point_plane_dist", C.c_double),
                ("max_iterations", C.c_uint64), ("rotation_convergence_thresh", C.c_double),
                ("position_convergence_thresh", C.c_double), ("min_associations", C.c_uint64)]


class RegResult(C.Structure):
    _fields_ = [("pose", C.c_double * 7), ("termination", C.c_uint32), ("iterations", C.c_uint32)]


class IterInfo(C.Structure):
    _fields_ = [("target_T_source_init", C.c_double * 7), ("estimate_update", C.c_double * 7),
                ("n_edge_associations", C.c_uint32), ("n_plane_associations", C.c_uint32)]


def fe_params(*a):
    return FeParams(*(a or (3, 6, 10, 50, 100.0, 1.0, 0.5, 1.0)))


def reg_params(*a):
    return RegParams(*(a or (5, 1.0, 3, 10.0, 5, 2.0, 4, 0.1, 10, 1e-3, 1e-2, 100)))


def conv_fe(p):
    """any struct with the reference's field names -> FeParams"""
    return FeParams(*[getattr(p, f[0]) for f in FeParams._fields_])


def conv_reg(p):
    return RegParams(*[getattr(p, f[0]) for f in RegParams._fields_])


def lib():
    global _lib
    if _lib is None:
        override = os.environ.get("HOSTCHECK_LIB")  # (tests/test_sanitizers.py: the -fsanitize build)
        if not override:
            subprocess.check_call(["make", "-s", "-C", _DIR])
        _lib = C.CDLL(override or os.path.join(_DIR, "libhostcheck.so"))
        _lib.hostcheck_knn.restype = C.c_uint64
        _lib.hostcheck_knn_fallbacks.restype = C.c_uint64
        _lib.hostcheck_knn_mismatches.restype = C.c_uint64
        _lib.hostcheck_knn_round2.restype = C.c_uint64
        _lib.hostcheck_knn_queued.restype = C.c_uint64
        _lib.hostcheck_fit_plane.restype = C.c_double
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def curvature_valid(xyz, H, W, rmin, rmax, fe):
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    curv = np.empty(H * W)
    mask = np.empty(H * W, dtype=np.uint8)
    lib().hostcheck_curvature_valid(_dp(xyz), C.c_uint64(H), C.c_uint64(W), C.c_double(rmin), C.c_double(rmax),
                                    C.byref(fe), _dp(curv), mask.ctypes.data_as(C.POINTER(C.c_uint8)))
    return curv, mask.astype(bool)


def select(curv, mask, H, W, fe):
    curv = np.ascontiguousarray(curv, dtype=np.float64)
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    e = np.empty(H * W + 1, dtype=np.uint32)
    p = np.empty(H * W + 1, dtype=np.uint32)
    ne, npl = C.c_uint64(0), C.c_uint64(0)
    lib().hostcheck_select(_dp(curv), mask.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_uint64(H), C.c_uint64(W),
                           C.byref(fe), e.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(ne),
                           p.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(npl))
    return e[:ne.value].copy(), p[:npl.value].copy()


def select_mis(curv, mask, H, W, fe):
    """Lane-level emulation of the bitmask-MIS selection kernel; None if the parameters need the fallback."""
    curv = np.ascontiguousarray(curv, dtype=np.float64)
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    e = np.empty(H * W + 1, dtype=np.uint32)
    p = np.empty(H * W + 1, dtype=np.uint32)
    ne, npl = C.c_uint64(0), C.c_uint64(0)
    rc = lib().hostcheck_select_mis(_dp(curv), mask.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_uint64(H), C.c_uint64(W),
                                    C.byref(fe), e.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(ne),
                                    p.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(npl))
    if rc:
        return None
    return e[:ne.value].copy(), p[:npl.value].copy()


def select_stdsort(curv, mask, H, W, fe):
    """features-inl.h:27-48 + :137-180 literally, with the real std::sort, on given curvature / mask arrays"""
    curv = np.ascontiguousarray(curv, dtype=np.float64)
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    e = np.empty(H * W + 1, dtype=np.uint32)
    p = np.empty(H * W + 1, dtype=np.uint32)
    ne, npl = C.c_uint64(0), C.c_uint64(0)
    lib().hostcheck_select_stdsort(_dp(curv), mask.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_uint64(H), C.c_uint64(W),
                                   C.byref(fe), e.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(ne),
                                   p.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(npl))
    return e[:ne.value].copy(), p[:npl.value].copy()


def replayed_lines():
    lib().hostcheck_replayed_lines.restype = C.c_uint64
    return lib().hostcheck_replayed_lines()


def knn(pts, q, k, max_dist):
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    out = np.empty(max(k, 1), dtype=np.uint64)
    m = lib().hostcheck_knn(_dp(pts), C.c_uint64(len(pts)), _dp(q), C.c_uint64(k), C.c_double(max_dist),
                            out.ctypes.data_as(C.POINTER(C.c_uint64)))
    return out[:m].copy()


def fit_plane(pts):
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    out = np.empty(4)
    avg = lib().hostcheck_fit_plane(_dp(pts), C.c_uint64(len(pts)), _dp(out))
    return out[:3].copy(), float(out[3]), avg


def fit_line(pts):
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    out = np.empty(6)
    lib().hostcheck_fit_line(_dp(pts), C.c_uint64(len(pts)), _dp(out))
    return out[:3].copy(), out[3:].copy()


def associate(src, tgt, est, is_plane, prm):
    src = np.ascontiguousarray(np.asarray(src, float).reshape(-1, 3))
    tgt = np.ascontiguousarray(np.asarray(tgt, float).reshape(-1, 3))
    est = np.ascontiguousarray(est, dtype=np.float64)
    n = len(src)
    pw = 4 if is_plane else 6
    valid = np.zeros(n, dtype=np.uint8)
    nearest = np.zeros(n, dtype=np.uint64)
    moved = np.zeros((n, 3))
    prims = np.zeros((n, pw))
    lib().hostcheck_associate(_dp(src), C.c_uint64(n), _dp(tgt), C.c_uint64(len(tgt)), _dp(est),
                              C.c_int(1 if is_plane else 0), C.byref(prm),
                              valid.ctypes.data_as(C.POINTER(C.c_uint8)),
                              nearest.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(moved), _dp(prims))
    return valid.astype(bool), nearest, moved, prims


def register(src_edge, src_planar, tgt_edge, tgt_planar, init=None, prm=None, want_info=False):
    prm = prm or reg_params()
    arrs = [np.ascontiguousarray(np.asarray(a, float).reshape(-1, 3)) for a in
            (src_edge, src_planar, tgt_edge, tgt_planar)]
    init = np.ascontiguousarray([0, 0, 0, 1, 0, 0, 0] if init is None else init, dtype=np.float64)
    res = RegResult()
    info = (IterInfo * max(1, prm.max_iterations))()
    lib().hostcheck_register(_dp(arrs[0]), C.c_uint64(len(arrs[0])), _dp(arrs[1]), C.c_uint64(len(arrs[1])),
                             _dp(arrs[2]), C.c_uint64(len(arrs[2])), _dp(arrs[3]), C.c_uint64(len(arrs[3])),
                             _dp(init), C.byref(prm), C.byref(res), info)
    pose = np.array(list(res.pose))
    if want_info:
        return pose, res.termination, res.iterations, [info[i] for i in range(res.iterations)]
    return pose, res.termination, res.iterations


def synth_scan(seed, pair, which, H, W, sigma):
    xyz = np.empty((H * W, 3))
    lib().hostcheck_synth_scan(C.c_uint64(seed), C.c_uint64(pair), C.c_uint32(which), C.c_uint32(H), C.c_uint32(W),
                               C.c_double(sigma), _dp(xyz))
    return xyz


def synth_pose(seed, pair):
    out = np.empty(7)
    lib().hostcheck_synth_pose(C.c_uint64(seed), C.c_uint64(pair), _dp(out))
    return out


def knn_fallbacks():
    """keyed-collector queries that were undecided and re-ran through the exact collector (cumulative)"""
    return int(lib().hostcheck_knn_fallbacks())


def knn_round2():
    """queued queries the lean FP32 search of the 5x5x5 block finished / all queued queries so far"""
    return int(lib().hostcheck_knn_round2()), int(lib().hostcheck_knn_queued())


def knn_mismatches():
    """queries on which the keyed path and the exact collector disagreed (must stay 0)"""
    return int(lib().hostcheck_knn_mismatches())


def plane_moments(v, n, d, x):
    """(direct, via_moments, max|s0|, valid): plane terms of the normal equations at ambient point x"""
    v, n, d, x = (np.ascontiguousarray(a, dtype=np.float64) for a in (v, n, d, x))
    a, b, bi = np.zeros(30), np.zeros(30), np.zeros(2)
    lib().hostcheck_plane_moments(_dp(v), _dp(n), _dp(d), C.c_uint64(len(d)), _dp(x), _dp(a), _dp(b), _dp(bi))
    return a[:29], b[:29], bi[0], bool(bi[1])


def moments_rel(v, n, d, x, r):
    """(bound says valid, max |s_i(r)| over unflagged records, max |s_i(x)| over them, largest moment-form error)"""
    v, n, d, x, r = (np.ascontiguousarray(a, dtype=np.float64) for a in (v, n, d, x, r))
    out = np.zeros(4)
    lib().hostcheck_moments_rel(_dp(v), _dp(n), _dp(d), C.c_uint64(len(d)), _dp(x), _dp(r), _dp(out))
    return bool(out[0]), float(out[1]), float(out[2]), float(out[3])
