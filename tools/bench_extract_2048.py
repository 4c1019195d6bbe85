import sys, time, numpy as np
sys.path.insert(0, '.')
from loam_amd import capi
c = capi.Context(0)
for (H, W, ns) in ((128, 2048, 64), (64, 2048, 128)):
    N = H * W
    lidar, fe = capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams()
    d_xyz = c.alloc(ns * N * 24)
    c.synth_scan_pairs_dev(5, 0, ns // 2, H, W, 0.01, d_xyz.ptr)
    ecap, pcap = c.edge_capacity(lidar, fe), c.planar_capacity(lidar, fe)
    d_ei, d_pi = c.alloc(ns * ecap * 4), c.alloc(ns * pcap * 4)
    d_ne, d_np = c.alloc(ns * 4), c.alloc(ns * 4)
    d_ex, d_px = c.alloc(ns * ecap * 24), c.alloc(ns * pcap * 24)
    for rep in range(3):
        c.synchronize(); t0 = time.perf_counter()
        c.extract_features_batch_dev(d_xyz.ptr, ns, lidar, fe, d_ei.ptr, d_ne.ptr, d_ex.ptr, d_pi.ptr, d_np.ptr, d_px.ptr)
        c.synchronize(); dt = time.perf_counter() - t0
    print(f"{H}x{W}: {ns} scans extracted in {dt*1e3:.2f} ms ({dt*1e6/ns:.1f} us/scan), planar/scan {d_np.download(np.uint32, ns).mean():.0f}")
    for b in (d_xyz, d_ei, d_pi, d_ne, d_np, d_ex, d_px): b.free()
