#!/usr/bin/env python3
"""The reference README's scan-to-scan loop (README.md:44-60 of DanMcGann/loam), unchanged, on the MI355X back end.

    python -m loam_amd.build            # once: libloamx.so + the pybind11 module
    python examples/scan_to_scan.py

`loam` below is this repository's pybind11 module (loam_amd/python/loam): same classes, functions and keyword
arguments as the reference's python/loam_bindings.cpp; point clouds are (N, 3) arrays (float32 arrays take the
FP32-input path)."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "loam_amd", "python"))
sys.path.insert(0, ROOT)
import loam  # noqa: E402
from loam_amd import capi  # noqa: E402  (only for the synthetic scans)

lidar_params = loam.LidarParams(64, 1024, 1.0, 120.0)
scans = [capi.synth_scan_host(7, 0, which, 64, 1024, 0.01) for which in (0, 1)]  # stand-in for a sensor stream

world_T_lidar = loam.Pose3d.Identity()
feat_prev = loam.extractFeatures(scans[0], lidar_params)
for pcd in scans[1:]:
    feat = loam.extractFeatures(pcd, lidar_params)
    prev_T_cur = loam.registerFeatures(source=feat, target=feat_prev, target_T_source_init=loam.Pose3d.Identity())
    world_T_lidar = world_T_lidar.compose(prev_T_cur)
    feat_prev = feat
    q = world_T_lidar.rotation
    print("pose: q = (%.6f %.6f %.6f %.6f)  t = %s" % (q.x(), q.y(), q.z(), q.w(), np.round(world_T_lidar.translation, 4)))
print("expected (the generator's ground truth):", np.round(capi.synth_pair_pose(7, 0), 6))
