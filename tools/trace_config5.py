#!/usr/bin/env python3
"""BASELINE config 5 alone (one 128x2048 scan against a ~1 M-point map through the persistent index), for a kernel trace:
    rocprofv3 --kernel-trace --stats -d <dir> --output-format csv -- python3 tools/trace_config5.py
The map is made of the GPU's own planar / edge features of nearby scans (no oracle: this is a timing tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from loam_amd import capi

ctx = capi.Context(0)
fe = capi.FeatureExtractionParams()
H5, W5 = 128, 2048
lidar5 = capi.LidarParams(H5, W5, 1.0, 120.0)
src = capi.synth_scan_host(99, 0, 1, H5, W5, 0.01)
e5, p5 = ctx.extract_features(src, lidar5, fe)
maps_p, maps_e, k = [], [], 0
while sum(len(m) for m in maps_p) < 1_000_000:
    s = capi.synth_scan_host(1000 + k, 0, 0, H5, W5, 0.01)
    e, p = ctx.extract_features(s, lidar5, fe)
    maps_p.append(s[p]); maps_e.append(s[e]); k += 1
map_p, map_e = np.concatenate(maps_p), np.concatenate(maps_e)
idx = ctx.target_index(map_e, map_p)
ctx.register_features_indexed(idx, src[e5], src[p5])
t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(n):
    pose, term, it = ctx.register_features_indexed(idx, src[e5], src[p5])
print("config 5, persistent index: %.3f ms per registration, %d ICF iterations, %d map points, %d source planar" %
      ((time.perf_counter() - t0) / n * 1e3, it, len(map_p), len(p5)))
