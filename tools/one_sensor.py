#!/usr/bin/env python3
"""Debug aid: scan-pair registration of the given shapes in one context, e.g.
    LOAMX_DEBUG_SYNC=1 python tools/one_sensor.py 64x2048x128 128x2048x64"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from loam_amd import capi
c = capi.Context(0)
for spec in sys.argv[1:]:
    H, W, n_pairs = (int(v) for v in spec.split("x"))
    N = H * W
    lidar, fe, reg = capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams(), capi.RegistrationParams()
    d_xyz = c.alloc(n_pairs * 2 * N * 24); d_res = c.alloc(n_pairs * 64)
    c.synth_scan_pairs_dev(20240311, 0, n_pairs, H, W, 0.01, d_xyz.ptr)
    c.synchronize()
    print(spec, "synth ok", flush=True)
    print("=== " + spec, file=sys.stderr, flush=True)
    c.register_scan_pairs_dev(d_xyz.ptr, n_pairs, lidar, fe, reg, d_res.ptr)
    c.synchronize()
    res = d_res.download(capi.RESULT_DTYPE, n_pairs)
    print(spec, "converged", int((res['termination'] == 0).sum()), res['iterations'].mean(), flush=True)
    d_xyz.free(); d_res.free()
