#!/usr/bin/env python3
"""config 2: one device-resident 64x1024 scan pair, registered K times (for a kernel trace of the single-pair latency path)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from loam_amd import capi
c = capi.Context(0)
H, W = 64, 1024
lidar, fe, reg = capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams(), capi.RegistrationParams()
d_xyz, d_res = c.alloc(2 * H * W * 24), c.alloc(64)
c.synth_scan_pairs_dev(20240311, 0, 1, H, W, 0.01, d_xyz.ptr)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(3):
    c.register_scan_pairs_dev(d_xyz.ptr, 1, lidar, fe, reg, d_res.ptr)
c.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    c.register_scan_pairs_dev(d_xyz.ptr, 1, lidar, fe, reg, d_res.ptr)
c.synchronize()
print("ms per pair %.3f" % ((time.perf_counter() - t0) / K * 1e3))
