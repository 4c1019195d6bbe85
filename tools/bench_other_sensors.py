#!/usr/bin/env python3
"""Scan-pair throughput for other sensor shapes than the headline 64x1024 (sanity check of the fast-path
conditions: column counts, 128-beam feature counts, short rings). Runs on the GPU box:
    python tools/bench_other_sensors.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from loam_amd import capi  # noqa: E402

c = capi.Context(0)
for (H, W, n_pairs) in ((64, 1024, 256), (128, 1024, 128), (64, 2048, 128), (128, 2048, 64), (32, 1024, 512), (16, 1800, 512), (64, 512, 512)):
    N = H * W
    lidar, fe, reg = capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams(), capi.RegistrationParams()
    d_xyz = c.alloc(n_pairs * 2 * N * 24)
    d_res = c.alloc(n_pairs * 64)
    c.synth_scan_pairs_dev(20240311, 0, n_pairs, H, W, 0.01, d_xyz.ptr)
    best = 1e9
    for rep in range(3):
        c.synchronize()
        t0 = time.perf_counter()
        c.register_scan_pairs_dev(d_xyz.ptr, n_pairs, lidar, fe, reg, d_res.ptr)
        c.synchronize()
        best = min(best, time.perf_counter() - t0)
    res = d_res.download(capi.RESULT_DTYPE, n_pairs)
    print(f"{H:4d} x {W:4d}: {n_pairs:4d} pairs in {best*1e3:7.2f} ms = {n_pairs/best:9.0f} pairs/s = {n_pairs*2*N/best/1e9:6.2f} G points/s; "
          f"converged {int((res['termination'] == 0).sum())}, mean ICF iterations {res['iterations'].mean():.2f}")
    d_xyz.free()
    d_res.free()
