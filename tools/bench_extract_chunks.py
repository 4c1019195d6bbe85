"""Experiment: the extraction of the bench batch in chunks of K scans (curvature + selection of a chunk back to back, so that
the selection's second read of the scans finds them in the 256 MB memory-side cache) against one call over all of them.
    python tools/bench_extract_chunks.py [chunk sizes ...]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from loam_amd import capi

c = capi.Context(0)
H, W, ns = 64, 1024, 2048
N = H * W
lidar, fe = capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams()
d_xyz = c.alloc(ns * N * 24)
c.synth_scan_pairs_dev(5, 0, ns // 2, H, W, 0.01, d_xyz.ptr)
ecap, pcap = c.edge_capacity(lidar, fe), c.planar_capacity(lidar, fe)
d_ne, d_np = c.alloc(ns * 4), c.alloc(ns * 4)
d_ex, d_px = c.alloc(ns * ecap * 24), c.alloc(ns * pcap * 24)
for K in [ns] + [int(x) for x in sys.argv[1:]] + [ns]:
    ts = []
    for rep in range(6):
        c.synchronize(); t0 = time.perf_counter()
        for s0 in range(0, ns, K):
            c.extract_features_batch_dev(d_xyz.ptr + s0 * N * 24, K, lidar, fe, None, d_ne.ptr + 4 * s0, d_ex.ptr + s0 * ecap * 24, None,
                                         d_np.ptr + 4 * s0, d_px.ptr + s0 * pcap * 24)
        c.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"chunks of {K:5d} scans: {min(ts[1:])*1e3:.3f} ms (median {sorted(ts[1:])[2]*1e3:.3f}); planar/scan {d_np.download(np.uint32, ns).mean():.0f}", flush=True)
