"""GPU: the pybind11 module `loam` (same surface as the reference's python/loam_bindings.cpp) on the
README's scan-to-scan loop, against the oracle."""
import os
import sys

import numpy as np
import pytest

from gpu_common import pose_diff
from loam_amd import build as B, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _loam():
    B.build_pybind()
    p = os.path.join(ROOT, "loam_amd", "python")
    if p not in sys.path:
        sys.path.insert(0, p)
    import loam
    return loam


def test_module_surface():
    loam = _loam()
    for name in ("LidarParams", "Pose3d", "Quaterniond", "FeatureExtractionParams", "LoamFeatures", "extractFeatures",
                 "computeCurvature", "computeValidPoints", "RegistrationParams", "RegistrationIterationInfo",
                 "RegistrationTerminationType", "RegistrationDetail", "registerFeatures", "CONVERGED", "MAX_ITER",
                 "INSUFFICIENT_ASSOCIATIONS"):
        assert hasattr(loam, name), name
    lp = loam.LidarParams(scan_lines=64, points_per_line=1024, min_range=1.0, max_range=120.0)
    with pytest.raises(AttributeError):
        lp.scan_lines = 3  # read-only like the reference
    q = loam.Quaterniond(w=1.0, x=0.0, y=0.0, z=0.0)
    p = loam.Pose3d(rotation=q, translation=np.array([1.0, 2.0, 3.0]))
    assert np.allclose(p.compose(other=p.inverse()).translation, 0)
    assert np.allclose(p.act(point=[0, 0, 1.0]), [1, 2, 4])


@pytest.mark.gpu
def test_readme_scan_to_scan_loop(oracle):
    loam = _loam()
    H, W = 32, 512
    lidar_params = loam.LidarParams(H, W, 1.0, 120.0)
    A = capi.synth_scan_host(9, 0, 0, H, W, 0.01)
    Bs = capi.synth_scan_host(9, 0, 1, H, W, 0.01)
    feat_i = loam.extractFeatures(A, lidar_params)
    feat_ip1 = loam.extractFeatures([list(r) for r in Bs], lidar_params)  # list of 3-vectors also accepted
    oe, op = oracle.extract_features(A, H, W, 1.0, 120.0)
    assert np.array_equal(feat_i.edge_points, A[oe]) and np.array_equal(feat_i.planar_points, A[op])
    assert np.array_equal(loam.computeCurvature(A, lidar_params).view(np.uint64),
                          oracle.compute_curvature(A, H, W).view(np.uint64))
    assert np.array_equal(loam.computeValidPoints(A, lidar_params), oracle.compute_valid_points(A, H, W, 1.0, 120.0))
    detail = loam.RegistrationDetail()
    i_T_ip1 = loam.registerFeatures(source=feat_ip1, target=feat_i, target_T_source_init=loam.Pose3d.Identity(),
                                    detail=detail)
    oe2, op2 = oracle.extract_features(Bs, H, W, 1.0, 120.0)
    po, to, io = oracle.register_features(Bs[oe2], Bs[op2], A[oe], A[op])
    r = i_T_ip1.rotation
    got = np.array([r.x(), r.y(), r.z(), r.w(), *i_T_ip1.translation])
    rot, trans = pose_diff(oracle, po, got)
    assert rot < 1e-5 and trans < 1e-5
    assert int(detail.termination_type) == to and len(detail.iteration_info) == io
    assert detail.termination_type == loam.CONVERGED
    assert len(detail.iteration_info[0].plane_associations) > 100
    with pytest.raises(RuntimeError):
        loam.extractFeatures(A[:100], lidar_params)


@pytest.mark.gpu
def test_float32_scans_take_the_fp32_input_path(oracle):
    """a float32 (N,3) array is passed through uncast; results equal the oracle on the widened scan."""
    loam = _loam()
    H, W = 16, 256
    lp = loam.LidarParams(H, W, 1.0, 120.0)
    A32 = capi.synth_scan_host(13, 0, 0, H, W, 0.01).astype(np.float32)
    wide = A32.astype(np.float64)
    f = loam.extractFeatures(A32, lp)
    oe, op = oracle.extract_features(wide, H, W, 1.0, 120.0)
    assert f.edge_points.dtype == np.float64
    assert np.array_equal(f.edge_points, wide[oe]) and np.array_equal(f.planar_points, wide[op])
    assert np.array_equal(loam.computeCurvature(A32, lp).view(np.uint64), oracle.compute_curvature(wide, H, W).view(np.uint64))
    assert np.array_equal(loam.computeValidPoints(A32, lp), oracle.compute_valid_points(wide, H, W, 1.0, 120.0))
