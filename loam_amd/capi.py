"""ctypes binding of libloamx.so — the C ABI declared in include/loamx.h.

This is host plumbing only: every compute call goes to the HIP kernels through the C ABI. If the
library is missing it raises; there is no Python or CPU fallback.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

OK, ERR_SCAN_SIZE, ERR_BAD_PARAM, ERR_HIP, ERR_CAPACITY, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_COMM = range(8)
CONVERGED, MAX_ITER, INSUFFICIENT_ASSOCIATIONS = 0, 1, 2
K_CURVATURE, K_SELECT, K_COMPACT, K_GRID, K_ASSOC, K_SWEEP, K_LM, K_MOMENT, K_KNN_PLANE, K_EXTRACT_FUSED, K_COUNT = range(11)


class LidarParams(C.Structure):
    """loam::LidarParams (reference: loam/include/loam/common.h:29-41)"""
    _fields_ = [("scan_lines", C.c_uint64), ("points_per_line", C.c_uint64), ("min_range", C.c_double),
                ("max_range", C.c_double)]


class FeatureExtractionParams(C.Structure):
    """loam::FeatureExtractionParams (reference: loam/include/loam/features.h:37-66)"""
    _fields_ = [("neighbor_points", C.c_uint64), ("number_sectors", C.c_uint64),
                ("max_edge_feats_per_sector", C.c_uint64), ("max_planar_feats_per_sector", C.c_uint64),
                ("edge_feat_threshold", C.c_double), ("planar_feat_threshold", C.c_double),
                ("occlusion_thresh", C.c_double), ("parallel_thresh", C.c_double)]

    def __init__(self, *a, **kw):
        if not a and not kw:
            a = (3, 6, 10, 50, 100.0, 1.0, 0.5, 1.0)
        super().__init__(*a, **kw)


class RegistrationParams(C.Structure):
    """loam::RegistrationParams (reference: loam/include/loam/registration.h:40-75)"""
    _fields_ = [("num_edge_neighbors", C.c_uint64), ("max_edge_neighbor_dist", C.c_double),
                ("min_line_fit_points", C.c_uint64), ("min_line_condition_number", C.c_double),
                ("num_plane_neighbors", C.c_uint64), ("max_plane_neighbor_dist", C.c_double),
                ("min_plane_fit_points", C.c_uint64), ("max_avg_point_plane_dist", C.c_double),
                ("max_iterations", C.c_uint64), ("rotation_convergence_thresh", C.c_double),
                ("position_convergence_thresh", C.c_double), ("min_associations", C.c_uint64)]

    def __init__(self, *a, **kw):
        if not a and not kw:
            a = (5, 1.0, 3, 10.0, 5, 2.0, 4, 0.1, 10, 1e-3, 1e-2, 100)
        super().__init__(*a, **kw)


class RegResult(C.Structure):
    _fields_ = [("pose", C.c_double * 7), ("termination", C.c_uint32), ("iterations", C.c_uint32)]


class IterInfo(C.Structure):
    _fields_ = [("target_T_source_init", C.c_double * 7), ("estimate_update", C.c_double * 7),
                ("n_edge_associations", C.c_uint32), ("n_plane_associations", C.c_uint32)]


class RegDetail(C.Structure):
    _fields_ = [("iter_info", C.POINTER(IterInfo)), ("n_iter_info", C.c_uint32),
                ("edge_pairs", C.POINTER(C.c_uint32)), ("pairs_cap_edge", C.c_size_t),
                ("n_edge_pairs", C.POINTER(C.c_uint32)), ("plane_pairs", C.POINTER(C.c_uint32)),
                ("pairs_cap_plane", C.c_size_t), ("n_plane_pairs", C.POINTER(C.c_uint32))]


class KernelStat(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("total_ms", C.c_double), ("algorithmic_bytes", C.c_double)]


RESULT_DTYPE = np.dtype([("pose", np.float64, 7), ("termination", np.uint32), ("iterations", np.uint32)])

class AssocDump(C.Structure):
    """loamx_assoc_dump (include/loamx.h)"""
    _fields_ = [(n, C.c_void_p) for n in ("edge_nn_count", "edge_nn_idx", "edge_valid", "edge_moved", "edge_lines",
                                          "plane_nn_count", "plane_nn_idx", "plane_valid", "plane_moved", "plane_planes", "queue_lengths")]


EXPORTS = [
    "loamx_default_fe_params", "loamx_default_reg_params", "loamx_status_string", "loamx_last_error",
    "loamx_ctx_create", "loamx_ctx_destroy", "loamx_ctx_set_stream", "loamx_ctx_synchronize",
    "loamx_compute_curvature", "loamx_compute_valid_points", "loamx_extract_features", "loamx_register_features",
    "loamx_target_index_create", "loamx_target_index_destroy", "loamx_register_features_indexed",
    "loamx_edge_capacity", "loamx_planar_capacity", "loamx_extract_features_batch_dev",
    "loamx_register_features_batch_dev", "loamx_register_scan_pairs_dev", "loamx_register_scan_pairs",
    "loamx_register_scan_pairs_f32", "loamx_ctx_enable_kernel_timing",
    "loamx_ctx_reset_kernel_stats", "loamx_ctx_get_kernel_stats", "loamx_kernel_name", "loamx_synth_pair_pose",
    "loamx_synth_scan_host", "loamx_synth_scan_pairs_dev", "loamx_dev_alloc", "loamx_dev_free",
    "loamx_copy_to_device", "loamx_copy_to_host",
    "loamx_compute_curvature_f32", "loamx_compute_valid_points_f32", "loamx_extract_features_f32",
    "loamx_extract_features_batch_dev_f32", "loamx_register_scan_pairs_dev_f32",
    "loamx_target_index_insert", "loamx_target_index_size",
    "loamx_shard_range", "loamx_comm_get_unique_id", "loamx_comm_create", "loamx_comm_wrap", "loamx_comm_destroy",
    "loamx_comm_info", "loamx_gather_results_dev", "loamx_comm_barrier", "loamx_comm_stats", "loamx_ctx_extract_counters",
    "loamx_ctx_set_option", "loamx_ctx_get_option",
    "loamx_fit_lines", "loamx_fit_planes", "loamx_knn_search", "loamx_associate", "loamx_target_index_stats",
]

_lib = None


def load(build_if_missing=True):
    """Loads libloamx.so (building it with hipcc if needed). Raises if it cannot be had."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if build_if_missing and _build.needs_build():
        _build.build()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -m loam_amd.build` (no CPU fallback exists)")
    lib = C.CDLL(path)
    dp, u32p, vp = C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.c_void_p
    lib.loamx_status_string.restype = C.c_char_p
    lib.loamx_last_error.restype = C.c_char_p
    lib.loamx_last_error.argtypes = [vp]
    lib.loamx_kernel_name.restype = C.c_char_p
    lib.loamx_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.loamx_ctx_destroy.argtypes = [vp]
    lib.loamx_ctx_destroy.restype = None
    lib.loamx_ctx_set_stream.argtypes = [vp, vp]
    lib.loamx_ctx_synchronize.argtypes = [vp]
    lib.loamx_compute_curvature.argtypes = [vp, dp, C.c_size_t, C.POINTER(LidarParams),
                                            C.POINTER(FeatureExtractionParams), dp]
    lib.loamx_compute_valid_points.argtypes = [vp, dp, C.c_size_t, C.POINTER(LidarParams),
                                               C.POINTER(FeatureExtractionParams), C.POINTER(C.c_uint8)]
    lib.loamx_extract_features.argtypes = [vp, dp, C.c_size_t, C.POINTER(LidarParams),
                                           C.POINTER(FeatureExtractionParams), u32p, C.c_size_t,
                                           C.POINTER(C.c_size_t), u32p, C.c_size_t, C.POINTER(C.c_size_t)]
    fp = C.POINTER(C.c_float)
    lib.loamx_compute_curvature_f32.argtypes = [vp, fp, C.c_size_t, C.POINTER(LidarParams),
                                                C.POINTER(FeatureExtractionParams), dp]
    lib.loamx_compute_valid_points_f32.argtypes = [vp, fp, C.c_size_t, C.POINTER(LidarParams),
                                                   C.POINTER(FeatureExtractionParams), C.POINTER(C.c_uint8)]
    lib.loamx_extract_features_f32.argtypes = [vp, fp, C.c_size_t, C.POINTER(LidarParams),
                                               C.POINTER(FeatureExtractionParams), u32p, C.c_size_t,
                                               C.POINTER(C.c_size_t), u32p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.loamx_register_features.argtypes = [vp, dp, C.c_size_t, dp, C.c_size_t, dp, C.c_size_t, dp, C.c_size_t, dp,
                                            C.POINTER(RegistrationParams), C.POINTER(RegResult),
                                            C.POINTER(RegDetail)]
    lib.loamx_target_index_create.argtypes = [vp, dp, C.c_size_t, dp, C.c_size_t, C.POINTER(RegistrationParams), C.POINTER(vp)]
    lib.loamx_target_index_insert.argtypes = [vp, vp, dp, C.c_size_t, dp, C.c_size_t]
    lib.loamx_target_index_size.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.loamx_target_index_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.loamx_target_index_destroy.argtypes = [vp, vp]
    lib.loamx_target_index_destroy.restype = None
    lib.loamx_register_features_indexed.argtypes = [vp, vp, dp, C.c_size_t, dp, C.c_size_t, dp, C.POINTER(RegistrationParams),
                                                    C.POINTER(RegResult), C.POINTER(RegDetail)]
    lib.loamx_edge_capacity.restype = C.c_size_t
    lib.loamx_edge_capacity.argtypes = [C.POINTER(LidarParams), C.POINTER(FeatureExtractionParams)]
    lib.loamx_planar_capacity.restype = C.c_size_t
    lib.loamx_planar_capacity.argtypes = [C.POINTER(LidarParams), C.POINTER(FeatureExtractionParams)]
    lib.loamx_extract_features_batch_dev.argtypes = [vp, vp, C.c_size_t, C.POINTER(LidarParams),
                                                     C.POINTER(FeatureExtractionParams), vp, vp, vp, vp, vp, vp]
    lib.loamx_extract_features_batch_dev_f32.argtypes = lib.loamx_extract_features_batch_dev.argtypes
    lib.loamx_register_features_batch_dev.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t,
                                                      C.c_size_t, vp, C.POINTER(RegistrationParams), vp]
    lib.loamx_register_scan_pairs_dev.argtypes = [vp, vp, C.c_size_t, C.POINTER(LidarParams),
                                                  C.POINTER(FeatureExtractionParams), C.POINTER(RegistrationParams),
                                                  vp]
    lib.loamx_register_scan_pairs_dev_f32.argtypes = lib.loamx_register_scan_pairs_dev.argtypes
    lib.loamx_register_scan_pairs.argtypes = lib.loamx_register_scan_pairs_dev.argtypes  # (host pointers)
    lib.loamx_register_scan_pairs_f32.argtypes = lib.loamx_register_scan_pairs_dev.argtypes
    lib.loamx_ctx_enable_kernel_timing.argtypes = [vp, C.c_int]
    lib.loamx_ctx_reset_kernel_stats.argtypes = [vp]
    lib.loamx_ctx_get_kernel_stats.argtypes = [vp, C.POINTER(KernelStat)]
    lib.loamx_synth_pair_pose.argtypes = [C.c_uint64, C.c_uint64, dp]
    lib.loamx_synth_pair_pose.restype = None
    lib.loamx_synth_scan_host.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, dp]
    lib.loamx_synth_scan_host.restype = None
    lib.loamx_synth_scan_pairs_dev.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_size_t, C.c_uint32, C.c_uint32,
                                               C.c_double, vp]
    lib.loamx_dev_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.loamx_dev_free.argtypes = [vp, vp]
    lib.loamx_copy_to_device.argtypes = [vp, vp, vp, C.c_size_t]
    lib.loamx_copy_to_host.argtypes = [vp, vp, vp, C.c_size_t]
    szp = C.POINTER(C.c_size_t)
    lib.loamx_shard_range.argtypes = [C.c_size_t, C.c_int, C.c_int, szp, szp]
    lib.loamx_shard_range.restype = None
    lib.loamx_comm_get_unique_id.argtypes = [C.c_char_p]
    lib.loamx_comm_create.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
    lib.loamx_comm_wrap.argtypes = [vp, vp, C.POINTER(vp)]
    lib.loamx_comm_destroy.argtypes = [vp]
    lib.loamx_comm_destroy.restype = None
    lib.loamx_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.loamx_gather_results_dev.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, vp]
    lib.loamx_comm_barrier.argtypes = [vp, vp, dp]
    lib.loamx_comm_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.loamx_ctx_extract_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.loamx_fit_lines.argtypes = [vp, dp, C.c_size_t, C.c_size_t, dp, dp]
    lib.loamx_fit_planes.argtypes = [vp, dp, C.c_size_t, C.c_size_t, dp, dp]
    lib.loamx_knn_search.argtypes = [vp, vp, C.c_int, dp, C.c_size_t, C.c_size_t, C.c_double, vp, vp]
    lib.loamx_associate.argtypes = [vp, dp, C.c_size_t, dp, C.c_size_t, dp, C.c_size_t, dp, C.c_size_t, dp,
                                    C.POINTER(RegistrationParams), C.POINTER(AssocDump)]
    lib.loamx_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    lib.loamx_ctx_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
    _lib = lib
    return lib


class LoamxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _pts(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, 3))
    return a


def _scan(a):
    """A scan as an (N,3) array: float32 input stays float32 (FP32-input entry points), anything else is float64."""
    a = np.asarray(a)
    return np.ascontiguousarray(a.reshape(-1, 3)) if a.dtype == np.float32 else _pts(a)


def _xyzp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float if a.dtype == np.float32 else C.c_double))


def synth_pair_pose(seed, pair_id):
    out = np.empty(7)
    load().loamx_synth_pair_pose(seed, pair_id, _dp(out))
    return out


def synth_scan_host(seed, pair_id, which, scan_lines, points_per_line, sigma=0.01):
    out = np.empty((scan_lines * points_per_line, 3))
    load().loamx_synth_scan_host(seed, pair_id, which, scan_lines, points_per_line, sigma, _dp(out))
    return out


COMM_ID_BYTES = 128


def shard_range(total_pairs, world_size, rank):
    """[first, first + count) of `rank`: the partition loamx_gather_results_dev expects (pure host arithmetic)."""
    f, n = C.c_size_t(0), C.c_size_t(0)
    load().loamx_shard_range(total_pairs, world_size, rank, C.byref(f), C.byref(n))
    return f.value, n.value


def comm_unique_id():
    """128-byte RCCL unique id; rank 0 creates it and hands it to every rank (loamx_comm_get_unique_id)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = load().loamx_comm_get_unique_id(buf)
    if rc != OK:
        raise LoamxError(rc, "loamx_comm_get_unique_id: " + load().loamx_status_string(rc).decode())
    return buf.raw


class Comm:
    """RCCL communicator of the multi-GPU batch mode (loamx_comm): one rank per process and GPU."""

    def __init__(self, ctx, unique_id, world_size, rank):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._check(ctx.lib.loamx_comm_create(ctx.h, bytes(unique_id), world_size, rank, C.byref(h)))
        self.h = h

    def info(self):
        w, r, d = C.c_int(0), C.c_int(0), C.c_int(0)
        self.ctx._check(self.ctx.lib.loamx_comm_info(self.h, C.byref(w), C.byref(r), C.byref(d)))
        return dict(world_size=w.value, rank=r.value, device=d.value)

    def gather_results_dev(self, d_local, n_local, total_pairs, d_all):
        """all ranks: d_all[total_pairs] <- every rank's records in pair-id order (asynchronous on the context stream)"""
        self.ctx._check(self.ctx.lib.loamx_gather_results_dev(self.ctx.h, self.h, d_local, n_local, total_pairs, d_all))

    def barrier(self, value=0.0):
        """waits for all ranks; returns the maximum of `value` over the ranks"""
        v = C.c_double(value)
        self.ctx._check(self.ctx.lib.loamx_comm_barrier(self.ctx.h, self.h, C.byref(v)))
        return v.value

    def stats(self):
        """what gather_results_dev / barrier have really enqueued so far (loamx_comm_stats): collectives by kind, and the
        one-rank device-copy shortcuts"""
        v = (C.c_uint64 * 4)()
        self.ctx._check(self.ctx.lib.loamx_comm_stats(self.h, v))
        return dict(ncclAllGather=int(v[0]), ncclBroadcast=int(v[1]), ncclAllReduce=int(v[2]), memcpy=int(v[3]))

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.loamx_comm_destroy(self.h)
            self.h = None


class DeviceBuffer:
    """Raw device allocation owned by a Context (for hosts that do not use torch tensors)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        p = C.c_void_p()
        ctx._check(ctx.lib.loamx_dev_alloc(ctx.h, nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.loamx_copy_to_device(self.ctx.h, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.loamx_copy_to_host(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.loamx_dev_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    """One device + stream + workspace (loamx_ctx)."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.loamx_ctx_create(device, C.byref(h))
        if rc != OK:
            raise LoamxError(rc, "loamx_ctx_create: " + self.lib.loamx_status_string(rc).decode() +
                             " (libloamx has no CPU fallback)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.loamx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            msg = self.lib.loamx_last_error(self.h).decode() or self.lib.loamx_status_string(rc).decode()
            raise LoamxError(rc, msg)

    def set_stream(self, hip_stream_handle):
        self._check(self.lib.loamx_ctx_set_stream(self.h, hip_stream_handle))

    def synchronize(self):
        self._check(self.lib.loamx_ctx_synchronize(self.h))

    def set_option(self, name, value=1):
        """debug / measurement switch of this context (include/loamx.h: loamx_ctx_set_option)"""
        self._check(self.lib.loamx_ctx_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int(0)
        self._check(self.lib.loamx_ctx_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def extract_counters(self):
        """(scan lines replayed in the reference's tie order, give-up fallbacks of the fused compaction), cumulative"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.loamx_ctx_extract_counters(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    # ---- host entry points -------------------------------------------------------------------------
    def compute_curvature(self, xyz, lidar, fe=None):
        fe = fe or FeatureExtractionParams()
        xyz = _scan(xyz)
        out = np.empty(len(xyz))
        fn = self.lib.loamx_compute_curvature_f32 if xyz.dtype == np.float32 else self.lib.loamx_compute_curvature
        self._check(fn(self.h, _xyzp(xyz), len(xyz), C.byref(lidar), C.byref(fe), _dp(out)))
        return out

    def compute_valid_points(self, xyz, lidar, fe=None):
        fe = fe or FeatureExtractionParams()
        xyz = _scan(xyz)
        out = np.empty(len(xyz), dtype=np.uint8)
        fn = self.lib.loamx_compute_valid_points_f32 if xyz.dtype == np.float32 else self.lib.loamx_compute_valid_points
        self._check(fn(self.h, _xyzp(xyz), len(xyz), C.byref(lidar), C.byref(fe), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out.astype(bool)

    def extract_features(self, xyz, lidar, fe=None):
        """Returns (edge_idx, planar_idx) in the reference's output order. A float32 array takes the FP32-input path."""
        fe = fe or FeatureExtractionParams()
        xyz = _scan(xyz)
        ecap = max(1, self.lib.loamx_edge_capacity(C.byref(lidar), C.byref(fe)))
        pcap = max(1, self.lib.loamx_planar_capacity(C.byref(lidar), C.byref(fe)))
        e = np.empty(ecap, dtype=np.uint32)
        p = np.empty(pcap, dtype=np.uint32)
        ne, npl = C.c_size_t(0), C.c_size_t(0)
        u32p = C.POINTER(C.c_uint32)
        fn = self.lib.loamx_extract_features_f32 if xyz.dtype == np.float32 else self.lib.loamx_extract_features
        self._check(fn(self.h, _xyzp(xyz), len(xyz), C.byref(lidar), C.byref(fe), e.ctypes.data_as(u32p), ecap, C.byref(ne),
                       p.ctypes.data_as(u32p), pcap, C.byref(npl)))
        return e[:ne.value].copy(), p[:npl.value].copy()

    def register_features(self, src_edge, src_planar, tgt_edge, tgt_planar, init_pose=None, reg=None,
                          want_detail=False):
        """Returns (pose7, termination, iterations[, detail dict])."""
        reg = reg or RegistrationParams()
        arrs = [_pts(a) for a in (src_edge, src_planar, tgt_edge, tgt_planar)]
        init = np.ascontiguousarray([0, 0, 0, 1, 0, 0, 0] if init_pose is None else init_pose, dtype=np.float64)
        res = RegResult()
        detail = None
        if want_detail:
            mi = max(1, reg.max_iterations)
            info = (IterInfo * mi)()
            ce, cp = max(1, len(arrs[0])), max(1, len(arrs[1]))
            ep = np.zeros((mi, ce, 2), dtype=np.uint32)
            pp = np.zeros((mi, cp, 2), dtype=np.uint32)
            nep = np.zeros(mi, dtype=np.uint32)
            npp = np.zeros(mi, dtype=np.uint32)
            u32p = C.POINTER(C.c_uint32)
            detail = RegDetail(info, 0, ep.ctypes.data_as(u32p), ce, nep.ctypes.data_as(u32p),
                               pp.ctypes.data_as(u32p), cp, npp.ctypes.data_as(u32p))
        self._check(self.lib.loamx_register_features(
            self.h, _dp(arrs[0]), len(arrs[0]), _dp(arrs[1]), len(arrs[1]), _dp(arrs[2]), len(arrs[2]),
            _dp(arrs[3]), len(arrs[3]), _dp(init), C.byref(reg), C.byref(res),
            C.byref(detail) if detail is not None else None))
        pose = np.array(list(res.pose))
        if want_detail:
            d = dict(iterations=[dict(target_T_source_init=np.array(list(info[i].target_T_source_init)),
                                      estimate_update=np.array(list(info[i].estimate_update)),
                                      n_edge=info[i].n_edge_associations, n_plane=info[i].n_plane_associations,
                                      edge_pairs=ep[i, :nep[i]].copy(), plane_pairs=pp[i, :npp[i]].copy())
                                 for i in range(detail.n_iter_info)])
            return pose, res.termination, res.iterations, d
        return pose, res.termination, res.iterations

    # ---- rows a16-a19 one by one (geometry_internal / kdtree_internal / registration_internal of the reference) ----
    def fit_lines(self, points):
        """geometry_internal::fitLine over (n_sets, k, 3) points -> (a (n,3), b (n,3), cond (n,))"""
        pts = np.ascontiguousarray(points, dtype=np.float64)
        n, k = pts.shape[0], pts.shape[1]
        out, cond = np.zeros((n, 6)), np.zeros(n)
        self._check(self.lib.loamx_fit_lines(self.h, _dp(pts), n, k, _dp(out), _dp(cond)))
        return out[:, :3].copy(), out[:, 3:].copy(), cond

    def fit_planes(self, points):
        """geometry_internal::fitPlane over (n_sets, k, 3) points -> (normal (n,3), d (n,), avg signed distance (n,))"""
        pts = np.ascontiguousarray(points, dtype=np.float64)
        n, k = pts.shape[0], pts.shape[1]
        out, avg = np.zeros((n, 4)), np.zeros(n)
        self._check(self.lib.loamx_fit_planes(self.h, _dp(pts), n, k, _dp(out), _dp(avg)))
        return out[:, :3].copy(), out[:, 3].copy(), avg

    def knn_search(self, index, which_set, queries, k, max_dist=-1.0):
        """kdtree_internal::knnSearch for every query against one set of a target index -> list of index arrays"""
        q = _pts(queries)
        idx, cnt = np.zeros((len(q), max(k, 1)), dtype=np.uint32), np.zeros(len(q), dtype=np.uint32)
        self._check(self.lib.loamx_knn_search(self.h, index, which_set, _dp(q), len(q), k, float(max_dist), idx.ctypes.data, cnt.ctypes.data))
        return [idx[i, :cnt[i]].copy() for i in range(len(q))]

    def associate(self, src_edge, src_planar, tgt_edge, tgt_planar, pose=None, reg=None):
        """One association pass of the registration kernels at `pose` (registration.cpp:23-103), per source feature:
        dict(edge=..., plane=...) of dict(nn (list of index arrays), valid, moved, prim)."""
        reg = reg or RegistrationParams()
        arrs = [_pts(a) for a in (src_edge, src_planar, tgt_edge, tgt_planar)]
        pose = np.ascontiguousarray([0, 0, 0, 1, 0, 0, 0] if pose is None else pose, dtype=np.float64)
        n, k, pw = [len(arrs[0]), len(arrs[1])], [int(reg.num_edge_neighbors), int(reg.num_plane_neighbors)], [6, 4]
        bufs = []
        for kind in range(2):
            bufs.append(dict(cnt=np.zeros(n[kind], dtype=np.uint32), idx=np.full((n[kind], max(k[kind], 1)), 0xFFFFFFFF, dtype=np.uint32),
                             valid=np.zeros(n[kind], dtype=np.uint8), moved=np.zeros((n[kind], 3)), prim=np.zeros((n[kind], pw[kind]))))
        queues = np.zeros(4, dtype=np.uint32)
        d = AssocDump(*([b[f].ctypes.data for b in bufs for f in ("cnt", "idx", "valid", "moved", "prim")] + [queues.ctypes.data]))
        self._check(self.lib.loamx_associate(self.h, _dp(arrs[0]), n[0], _dp(arrs[1]), n[1], _dp(arrs[2]), len(arrs[2]), _dp(arrs[3]),
                                             len(arrs[3]), _dp(pose), C.byref(reg), C.byref(d)))
        out = {}
        for kind, name in enumerate(("edge", "plane")):
            b = bufs[kind]
            out[name] = dict(nn=[b["idx"][i, :b["cnt"][i]].copy() for i in range(n[kind])], valid=b["valid"].astype(bool),
                             moved=b["moved"], prim=b["prim"], queued=(int(queues[kind]), int(queues[2 + kind])))
        return out

    # ---- persistent target index (scan-to-map) -----------------------------------------------------------
    def target_index(self, tgt_edge, tgt_planar, reg=None):
        reg = reg or RegistrationParams()
        te, tp = _pts(tgt_edge), _pts(tgt_planar)
        h = C.c_void_p()
        self._check(self.lib.loamx_target_index_create(self.h, _dp(te), len(te), _dp(tp), len(tp), C.byref(reg), C.byref(h)))
        return h

    def target_index_insert(self, index, edge, planar):
        """Appends points to a target index (same result as an index built over the concatenated sets)."""
        e, p = _pts(edge), _pts(planar)
        self._check(self.lib.loamx_target_index_insert(self.h, index, _dp(e), len(e), _dp(p), len(p)))

    def target_index_size(self, index):
        ne, npl = C.c_size_t(0), C.c_size_t(0)
        self._check(self.lib.loamx_target_index_size(index, C.byref(ne), C.byref(npl)))
        return ne.value, npl.value

    def target_index_stats(self, index):
        """(full builds of a feature kind's grid, merges into an existing grid) so far"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.loamx_target_index_stats(index, C.byref(a), C.byref(b)))
        return a.value, b.value

    def target_index_destroy(self, index):
        self.lib.loamx_target_index_destroy(self.h, index)

    def register_features_indexed(self, index, src_edge, src_planar, init_pose=None, reg=None):
        reg = reg or RegistrationParams()
        se, sp = _pts(src_edge), _pts(src_planar)
        init = np.ascontiguousarray([0, 0, 0, 1, 0, 0, 0] if init_pose is None else init_pose, dtype=np.float64)
        res = RegResult()
        self._check(self.lib.loamx_register_features_indexed(self.h, index, _dp(se), len(se), _dp(sp), len(sp), _dp(init),
                                                             C.byref(reg), C.byref(res), None))
        return np.array(list(res.pose)), res.termination, res.iterations

    # ---- device-resident batch entry points (raw device pointers as ints) -----------------------------
    def edge_capacity(self, lidar, fe):
        return self.lib.loamx_edge_capacity(C.byref(lidar), C.byref(fe))

    def planar_capacity(self, lidar, fe):
        return self.lib.loamx_planar_capacity(C.byref(lidar), C.byref(fe))

    def extract_features_batch_dev(self, d_xyz, n_scans, lidar, fe, d_edge_idx, d_n_edge, d_edge_xyz, d_planar_idx,
                                   d_n_planar, d_planar_xyz, f32=False):
        fn = self.lib.loamx_extract_features_batch_dev_f32 if f32 else self.lib.loamx_extract_features_batch_dev
        self._check(fn(self.h, d_xyz, n_scans, C.byref(lidar), C.byref(fe), d_edge_idx, d_n_edge, d_edge_xyz, d_planar_idx,
                       d_n_planar, d_planar_xyz))

    def register_features_batch_dev(self, n_pairs, d_src_edge, d_n_src_edge, d_src_planar, d_n_src_planar,
                                    d_tgt_edge, d_n_tgt_edge, d_tgt_planar, d_n_tgt_planar, edge_stride,
                                    planar_stride, d_init, reg, d_results):
        self._check(self.lib.loamx_register_features_batch_dev(
            self.h, n_pairs, d_src_edge, d_n_src_edge, d_src_planar, d_n_src_planar, d_tgt_edge, d_n_tgt_edge,
            d_tgt_planar, d_n_tgt_planar, edge_stride, planar_stride, d_init, C.byref(reg), d_results))

    def register_scan_pairs_dev(self, d_xyz, n_pairs, lidar, fe, reg, d_results, f32=False):
        fn = self.lib.loamx_register_scan_pairs_dev_f32 if f32 else self.lib.loamx_register_scan_pairs_dev
        self._check(fn(self.h, d_xyz, n_pairs, C.byref(lidar), C.byref(fe), C.byref(reg), d_results))

    def register_scan_pairs(self, xyz, n_pairs, lidar, fe=None, reg=None, out=None):
        """Host memory in, host memory out (loamx_register_scan_pairs): xyz = a C-contiguous float64 / float32 array (or an
        integer address + dtype via `xyz=(ptr, np.float32)`) of n_pairs x 2 scans, target scan first; returns the result
        records (RESULT_DTYPE). Pinned memory (e.g. a torch pin_memory tensor's numpy view) lets the uploads overlap."""
        fe, reg = fe or FeatureExtractionParams(), reg or RegistrationParams()
        if isinstance(xyz, tuple):
            ptr, dt = xyz
            f32 = np.dtype(dt) == np.float32
        else:
            assert xyz.flags["C_CONTIGUOUS"] and xyz.dtype in (np.float64, np.float32)
            ptr, f32 = xyz.ctypes.data, xyz.dtype == np.float32
        res = out if out is not None else np.zeros(n_pairs, dtype=RESULT_DTYPE)
        fn = self.lib.loamx_register_scan_pairs_f32 if f32 else self.lib.loamx_register_scan_pairs
        self._check(fn(self.h, ptr, n_pairs, C.byref(lidar), C.byref(fe), C.byref(reg), res.ctypes.data))
        return res

    def synth_scan_pairs_dev(self, seed, first_pair, n_pairs, scan_lines, points_per_line, sigma, d_xyz):
        self._check(self.lib.loamx_synth_scan_pairs_dev(self.h, seed, first_pair, n_pairs, scan_lines,
                                                        points_per_line, sigma, d_xyz))

    # ---- kernel timing --------------------------------------------------------------------------------------
    def enable_kernel_timing(self, on=True):
        self._check(self.lib.loamx_ctx_enable_kernel_timing(self.h, 1 if on else 0))

    def reset_kernel_stats(self):
        self._check(self.lib.loamx_ctx_reset_kernel_stats(self.h))

    def kernel_stats(self):
        st = (KernelStat * K_COUNT)()
        self._check(self.lib.loamx_ctx_get_kernel_stats(self.h, st))
        return {self.lib.loamx_kernel_name(i).decode(): dict(launches=int(st[i].launches), total_ms=st[i].total_ms,
                                                              algorithmic_bytes=st[i].algorithmic_bytes)
                for i in range(K_COUNT)}
