#!/usr/bin/env python3
"""GPU: who is queued by round 1 of the plane k-NN, and who is left for the cooperative kernel? (bench pairs, at the identity
and at the true pose)   python tools/queue_stats.py [pair ...]"""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from loam_amd import capi  # noqa: E402

c = capi.Context(0)
H, W = 64, 1024
lidar = capi.LidarParams(H, W, 1.0, 120.0)
for pair in [int(a) for a in sys.argv[1:]] or [0, 1, 2]:
    A, B = capi.synth_scan_host(20240311, pair, 0, H, W, 0.01), capi.synth_scan_host(20240311, pair, 1, H, W, 0.01)
    ea, pa = c.extract_features(A, lidar)
    eb, pb = c.extract_features(B, lidar)
    pose = c.register_features(B[eb], B[pb], A[ea], A[pa])[0]
    tree = cKDTree(A[pa])
    for name, P in (("identity", None), ("final pose", pose)):
        d = c.associate(B[eb], B[pb], A[ea], A[pa], pose=P)
        q, listed = d["plane"]["queued"]
        moved = d["plane"]["moved"]
        dist, _ = tree.query(moved, k=5)
        d5 = dist[:, 4]
        n_in_r = np.array([len(x) for x in tree.query_ball_point(moved, 2.0)])
        print(f"pair {pair} at the {name}: {len(moved)} plane queries, {q} queued ({100*q/len(moved):.1f} %), {listed} left for the cooperative kernel; "
              f"d5 > 0.5 m: {(d5 > 0.5).sum()}, > 1.0 m: {(d5 > 1.0).sum()}, > 1.25 m: {(d5 > 1.25).sum()}, fewer than 5 within 2 m: {(n_in_r < 5).sum()}, none within 2 m: {(n_in_r == 0).sum()}")
