#!/usr/bin/env python3
"""Writes the golden fixtures of tests/golden/ — DATA only: the inputs and expected outputs of the known-answer
tests the reference holds for the hot path (/root/reference/tests/test_feature_extraction.cpp, test_geometry.cpp,
test_registration.cpp), as restated in tests/reference_kats.py. Nothing is read from /root/reference and no
reference code is executed: the reference is unbuildable in this image.

    python tests/golden/make_golden.py        # rewrites the three files below (same content on every run)

  fe_kats.json            8 feature-extraction KATs: points, scan shape, range gate, params, expected curvature
                          values / valid and invalid indices (test_feature_extraction.cpp:27-299)
  pose_kats.json          Pose3d compose / inverse / matrix constants (GTSAM-generated, test_geometry.cpp:31-79)
                          and the two distance grids (:91-113) with their closed forms evaluated
  registration_kats.npz   the feature-level scene (162 edge + 8 941 planar points, test_registration.cpp:8-56),
                          the plane-only scene (:177-199), and per case: source_T_target, init, max_iterations,
                          tolerances (:69-175)
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import reference_kats as K  # noqa: E402


def main():
    fe = []
    for k in K.fe_kats():
        d = dict(name=k["name"], scan_lines=k["H"], points_per_line=k["W"], min_range=k["rmin"], max_range=k["rmax"],
                 params=list(K.KAT_FE_PARAMS), points=k["pts"].tolist())
        if "curvature" in k:
            d["expected_curvature"] = {str(i): v for i, v in sorted(k["curvature"].items())}
            d["tolerance"] = 1e-9
        else:
            d["expected_invalid"], d["expected_valid"] = sorted(k["invalid"]), sorted(k["valid"])
        fe.append(d)
    json.dump(fe, open(os.path.join(HERE, "fe_kats.json"), "w"), indent=1)

    P = K.POSE_KATS
    pose = dict(
        layout="pose = [qx, qy, qz, qw, tx, ty, tz] (Eigen coefficient order)",
        compose=dict(p1=P["compose"]["p1"].tolist(), p2=P["compose"]["p2"].tolist(), expected_q=P["compose"]["expected_q"].tolist(),
                     expected_t=P["compose"]["expected_t"].tolist(), is_approx_prec=1e-8),
        inverse=dict(p1=P["inverse"]["p1"].tolist(), expected_q=P["inverse"]["expected_q"].tolist(),
                     expected_t=P["inverse"]["expected_t"].tolist(), is_approx_prec=1e-8),
        matrix=dict(p1=P["matrix"]["p1"].tolist(), expected=P["matrix"]["expected"].tolist(), is_approx_prec=1e-6),
        distance_grid=dict(line_a=[0, 0, 0], line_b=[0, 0, 1], plane_n=[1, 0, 0], plane_d=2.25, tolerance=1e-8,
                           points=[[-5 + 0.5 * xi, -5 + 0.5 * yi, (-5 + 0.5 * xi) + (-5 + 0.5 * yi)] for xi in range(20) for yi in range(20)]))
    g = pose["distance_grid"]
    g["expected_line_distance"] = [math.sqrt(p[0] * p[0] + p[1] * p[1]) for p in g["points"]]
    g["expected_plane_distance"] = [abs(p[0] - 2.25) for p in g["points"]]
    json.dump(pose, open(os.path.join(HERE, "pose_kats.json"), "w"), indent=1)

    e, p = K.registration_scene()
    pe, pp = K.plane_only_scene()
    arrays = dict(scene_edge=e, scene_planar=p, plane_only_edge=pe, plane_only_planar=pp,
                  case_names=np.array([c["name"] for c in K.REGISTRATION_CASES]),
                  source_T_target=np.array([c["source_T_target"] for c in K.REGISTRATION_CASES]),
                  init=np.array([c["init"] if c["init"] is not None else [0, 0, 0, 1, 0, 0, 0] for c in K.REGISTRATION_CASES], dtype=np.float64),
                  max_iterations=np.array([c["max_iter"] if c["max_iter"] is not None else 10 for c in K.REGISTRATION_CASES]),
                  rot_tol=np.array([c["rot_tol"] for c in K.REGISTRATION_CASES]), trans_tol=np.array([c["trans_tol"] for c in K.REGISTRATION_CASES]))
    np.savez_compressed(os.path.join(HERE, "registration_kats.npz"), **arrays)
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith((".json", ".npz"))))


if __name__ == "__main__":
    main()
