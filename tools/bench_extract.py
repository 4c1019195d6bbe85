"""Extraction only, on the bench workload's scans (2 048 scans of 64 x 1024): the kernels of rows a5-a10 by HIP events.
    python tools/bench_extract.py [--scans 2048] [--opt NAME ...]      # options as loamx_ctx_set_option names
Under rocprofv3 (--kernel-trace --stats) the per-kernel table names the selection kernel that ran."""
import argparse, sys, time
import numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from loam_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--scans", type=int, default=2048)
ap.add_argument("--H", type=int, default=64)
ap.add_argument("--W", type=int, default=1024)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
c = capi.Context(0)
for o in a.opt:
    c.set_option(o, 1)
H, W, ns = a.H, a.W, a.scans
N = H * W
lidar, fe = capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams()
d_xyz = c.alloc(ns * N * 24)
c.synth_scan_pairs_dev(5, 0, ns // 2, H, W, 0.01, d_xyz.ptr)
ecap, pcap = c.edge_capacity(lidar, fe), c.planar_capacity(lidar, fe)
d_ei, d_pi = c.alloc(ns * ecap * 4), c.alloc(ns * pcap * 4)
d_ne, d_np = c.alloc(ns * 4), c.alloc(ns * 4)
d_ex, d_px = c.alloc(ns * ecap * 24), c.alloc(ns * pcap * 24)
c.enable_kernel_timing(True)
for rep in range(a.reps + 1):
    if rep == 1:
        c.reset_kernel_stats()
    c.synchronize(); t0 = time.perf_counter()
    c.extract_features_batch_dev(d_xyz.ptr, ns, lidar, fe, d_ei.ptr, d_ne.ptr, d_ex.ptr, d_pi.ptr, d_np.ptr, d_px.ptr)
    c.synchronize(); dt = time.perf_counter() - t0
print(f"{H}x{W} x {ns} scans, options {a.opt}: last call {dt*1e3:.3f} ms; planar/scan {d_np.download(np.uint32, ns).mean():.0f} edge/scan {d_ne.download(np.uint32, ns).mean():.0f}")
print("  counters (replayed lines, give-ups, features):", c.extract_counters())
for k, v in c.kernel_stats().items():
    if v["launches"]:
        print("  ", k.ljust(26), v["launches"], "%.4f ms" % (v["total_ms"] / v["launches"]))
chk = (d_pi.download(np.uint32, ns * pcap).astype(np.uint64).sum(), d_ei.download(np.uint32, ns * ecap).astype(np.uint64).sum())
print("  index checksums", chk)
