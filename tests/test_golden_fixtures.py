"""CPU: the oracle against the committed golden fixtures (tests/golden/*.json, *.npz — the reference-held
known-answer data as files, written by tests/golden/make_golden.py), read from disk only."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _is_approx(a, b, prec):  # Eigen isApprox
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    return np.sum((a - b) ** 2) <= prec * prec * min(np.sum(a * a), np.sum(b * b))


def test_fixture_files_are_what_the_generator_writes(tmp_path):
    import reference_kats as K
    fe = json.load(open(os.path.join(GOLD, "fe_kats.json")))
    assert [k["name"] for k in fe] == [k["name"] for k in K.fe_kats()] and len(fe) == 8
    for f, k in zip(fe, K.fe_kats()):
        assert np.array_equal(np.array(f["points"]), k["pts"]) and f["params"] == list(K.KAT_FE_PARAMS)
    z = np.load(os.path.join(GOLD, "registration_kats.npz"))
    e, p = K.registration_scene()
    assert np.array_equal(z["scene_edge"], e) and np.array_equal(z["scene_planar"], p)
    assert z["scene_edge"].shape == (162, 3) and z["scene_planar"].shape == (8941, 3)
    assert list(z["case_names"]) == [c["name"] for c in K.REGISTRATION_CASES]


@pytest.mark.parametrize("i", range(8))
def test_oracle_extraction_against_golden(oracle, i):
    k = json.load(open(os.path.join(GOLD, "fe_kats.json")))[i]
    prm = oracle.FeParams(*k["params"])
    pts = np.array(k["points"])
    if "expected_curvature" in k:
        c = oracle.compute_curvature(pts, k["scan_lines"], k["points_per_line"], prm)
        for idx, v in k["expected_curvature"].items():
            assert abs(c[int(idx)] - v) < k["tolerance"]
    else:
        m = oracle.compute_valid_points(pts, k["scan_lines"], k["points_per_line"], k["min_range"], k["max_range"], prm)
        assert not m[k["expected_invalid"]].any() and m[k["expected_valid"]].all()


def test_oracle_geometry_against_golden(oracle):
    g = json.load(open(os.path.join(GOLD, "pose_kats.json")))
    c = g["compose"]
    out = oracle.pose_compose(c["p1"], c["p2"])
    assert _is_approx(out[:4], c["expected_q"], c["is_approx_prec"]) and _is_approx(out[4:], c["expected_t"], c["is_approx_prec"])
    c = g["inverse"]
    out = oracle.pose_inverse(c["p1"])
    assert _is_approx(out[:4], c["expected_q"], c["is_approx_prec"]) and _is_approx(out[4:], c["expected_t"], c["is_approx_prec"])
    c = g["matrix"]
    assert _is_approx(oracle.pose_matrix(c["p1"]), c["expected"], c["is_approx_prec"])
    d = g["distance_grid"]
    for p, el, ep in zip(d["points"], d["expected_line_distance"], d["expected_plane_distance"]):
        assert abs(oracle.point_to_line_distance(p, d["line_a"], d["line_b"]) - el) < d["tolerance"]
        assert abs(oracle.point_to_plane_distance(p, d["plane_n"], d["plane_d"]) - ep) < d["tolerance"]


@pytest.mark.parametrize("i", range(5))
def test_oracle_registration_against_golden(oracle, i):
    import reference_kats as K  # (only the point transform helper)
    z = np.load(os.path.join(GOLD, "registration_kats.npz"))
    T = z["source_T_target"][i]
    src_e, src_p = K.transform_points(T, z["scene_edge"]), K.transform_points(T, z["scene_planar"])
    prm = oracle.RegParams()
    prm.max_iterations = int(z["max_iterations"][i])
    pose, term, iters = oracle.register_features(src_e, src_p, z["scene_edge"], z["scene_planar"], z["init"][i], prm)
    err = oracle.pose_compose(T, pose)  # test_registration.cpp:80-81
    assert oracle.quat_angular_distance(err[:4], [0, 0, 0, 1.0]) < z["rot_tol"][i]
    assert np.all(np.abs(err[4:]) < z["trans_tol"][i])
