"""Pins the CPU oracle against every known-answer test the reference holds for the hot path
(SURVEY.md section 4 / 8c): 8 feature-extraction KATs, Pose3d KATs, distance KATs, and the six
registration scenes with the reference's own tolerances. CPU only."""
import math

import numpy as np
import pytest

import reference_kats as K


def _fe(oracle):
    return oracle.FeParams(*K.KAT_FE_PARAMS)


@pytest.mark.parametrize("kat", K.fe_kats(), ids=lambda k: k["name"])
def test_feature_extraction_kats(oracle, kat):
    prm = _fe(oracle)
    if "curvature" in kat:
        c = oracle.compute_curvature(kat["pts"], kat["H"], kat["W"], prm)
        assert c.shape == (kat["W"],)
        for i, v in kat["curvature"].items():
            assert abs(c[i] - v) < 1e-9, (i, c[i], v)
    else:
        m = oracle.compute_valid_points(kat["pts"], kat["H"], kat["W"], kat["rmin"], kat["rmax"], prm)
        assert m.shape == (kat["W"],)
        for i in kat["invalid"]:
            assert not m[i], i
        for i in kat["valid"]:
            assert m[i], i


def test_empty_scan_and_size_mismatch(oracle):
    # test_feature_extraction.cpp:311-319: empty scan with 0x0 params runs
    e, p = oracle.extract_features(np.zeros((0, 3)), 0, 0, 0.1, 100.0, _fe(oracle))
    assert len(e) == 0 and len(p) == 0
    # common.h:105-113: size mismatch is an error
    with pytest.raises(RuntimeError):
        oracle.compute_curvature(np.zeros((10, 3)), 1, 11, _fe(oracle))


def test_pose_kats(oracle):
    c = K.POSE_KATS["compose"]
    comp = oracle.pose_compose(c["p1"], c["p2"])
    assert K.is_approx(comp[:4], c["expected_q"], 1e-8)
    assert K.is_approx(comp[4:], c["expected_t"], 1e-8)
    i = K.POSE_KATS["inverse"]
    inv = oracle.pose_inverse(i["p1"])
    assert K.is_approx(inv[:4], i["expected_q"], 1e-8)
    assert K.is_approx(inv[4:], i["expected_t"], 1e-8)
    m = K.POSE_KATS["matrix"]
    assert K.is_approx(oracle.pose_matrix(m["p1"]), m["expected"], 1e-6)
    # act == matrix * p
    p = np.array([0.3, -1.2, 2.5])
    M = oracle.pose_matrix(m["p1"])
    assert np.allclose(oracle.pose_act(m["p1"], p), M[:3, :3] @ p + M[:3, 3], atol=1e-12)


def test_distance_kats(oracle):
    # test_geometry.cpp:91-113
    la, lb = np.zeros(3), np.array([0.0, 0.0, 1.0])
    n = np.array([1.0, 0.0, 0.0])
    for xi in range(20):
        for yi in range(20):
            x, y = -5 + 0.5 * xi, -5 + 0.5 * yi
            p = np.array([x, y, x + y])
            assert abs(oracle.point_to_line_distance(p, la, lb) - math.sqrt(x * x + y * y)) < 1e-8
            assert abs(oracle.point_to_plane_distance(p, n, 2.25) - abs(x - 2.25)) < 1e-8


def test_scene_sizes():
    e, p = K.registration_scene()
    assert e.shape == (162, 3) and p.shape == (8941, 3)


@pytest.mark.parametrize("case", K.REGISTRATION_CASES, ids=lambda c: c["name"])
def test_registration_kats(oracle, case):
    tgt_e, tgt_p = K.registration_scene()
    src_e = K.transform_points(case["source_T_target"], tgt_e)
    src_p = K.transform_points(case["source_T_target"], tgt_p)
    prm = oracle.RegParams()
    if case["max_iter"] is not None:
        prm.max_iterations = case["max_iter"]
    pose, term, iters = oracle.register_features(src_e, src_p, tgt_e, tgt_p, case["init"], prm)
    rot_err, trans_err = K.registration_error(case["source_T_target"], pose, oracle.pose_compose,
                                              oracle.quat_angular_distance)
    assert rot_err < case["rot_tol"], (rot_err, term, iters)
    assert np.all(np.abs(trans_err) < case["trans_tol"]), (trans_err, term, iters)


def test_registration_plane_only_identity(oracle):
    # test_registration.cpp:177-199: zero edge points, self registration -> identity
    e, p = K.plane_only_scene()
    pose, term, iters = oracle.register_features(e, p, e, p)
    assert oracle.quat_angular_distance(pose[:4], np.array([0, 0, 0, 1.0])) < 1e-4
    assert np.all(np.abs(pose[4:]) < 1e-3)


def test_insufficient_associations(oracle):
    e, p = K.plane_only_scene()
    far = p + np.array([100.0, 0, 0])
    pose, term, iters = oracle.register_features(e, far, e, p)
    assert term == oracle.INSUFFICIENT_ASSOCIATIONS
    assert np.allclose(pose, [0, 0, 0, 1, 0, 0, 0])
