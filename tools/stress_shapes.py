#!/usr/bin/env python3
"""Stress (GPU box): batches of random sensor shapes through loamx_register_scan_pairs_dev in one context, every batch
twice (bit-identical results asked) and one pair of it against the CPU oracle. Exercises the asynchronous multi-stream
paths with set sizes on both sides of every threshold (512 / 20 480 points, 64 / 128 picks per sector ...).
    python tools/stress_shapes.py [rounds] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from loam_amd import capi  # noqa: E402
import oracle_lib as O  # noqa: E402
from gpu_common import pose_diff  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
c = capi.Context(0)
bad = 0
for r in range(rounds):
    H = int(rng.choice([16, 32, 64, 128]))
    W = int(rng.choice([512, 1024, 1800, 2048]))
    P = int(rng.choice([1, 3, 8, 9, 24, 40, 64]))
    while P * 2 * H * W * 24 > 1.5e9:
        P //= 2
    seed, first = int(rng.integers(1, 1 << 30)), int(rng.integers(0, 5000))
    N = H * W
    lidar, fe, reg = capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams(), capi.RegistrationParams()
    ofe, oreg = O.FeParams(), O.RegParams()
    if rng.integers(0, 3) == 0:  # non-default parameters on both sides (same field names in both bindings)
        vals_fe = {"neighbor_points": int(rng.choice([2, 3, 4, 5])), "number_sectors": int(rng.choice([3, 6, 8])),
                   "max_edge_feats_per_sector": int(rng.choice([2, 10])), "max_planar_feats_per_sector": int(rng.choice([20, 50, 70])),
                   "planar_feat_threshold": float(rng.choice([0.5, 1.0, 2.0]))}
        vals_reg = {"num_edge_neighbors": int(rng.choice([3, 5, 8])), "num_plane_neighbors": int(rng.choice([4, 5, 8])),
                    "max_plane_neighbor_dist": float(rng.choice([1.0, 2.0, 3.0])), "max_edge_neighbor_dist": float(rng.choice([1.0, 2.0]))}
        for k, v in vals_fe.items():
            setattr(fe, k, v), setattr(ofe, k, v)
        for k, v in vals_reg.items():
            setattr(reg, k, v), setattr(oreg, k, v)
    d_xyz, d_res = c.alloc(P * 2 * N * 24), c.alloc(P * 64)
    c.synth_scan_pairs_dev(seed, first, P, H, W, 0.01, d_xyz.ptr)
    f32 = bool(rng.integers(0, 4) == 0) and P * 2 * N * 24 < 4e8  # float-field scans: the FP32-input entry point
    scans32 = None
    if f32:
        c.synchronize()
        scans32 = d_xyz.download(np.float64, P * 2 * N * 3).astype(np.float32)
        d_xyz.upload(scans32)
    out = []
    for _ in range(2):
        c.register_scan_pairs_dev(d_xyz.ptr, P, lidar, fe, reg, d_res.ptr, f32=f32)
        c.synchronize()
        out.append(d_res.download(np.uint8, P * 64).copy())
    same = np.array_equal(out[0], out[1])
    res = out[0].view(capi.RESULT_DTYPE)
    pr = int(rng.integers(0, P))
    A, B = capi.synth_scan_host(seed, first + pr, 0, H, W, 0.01), capi.synth_scan_host(seed, first + pr, 1, H, W, 0.01)
    if f32:  # the oracle on the widened float scan (FieldAccessor widens float fields to double, common.h:55-60)
        sc = scans32.reshape(P, 2, N, 3)
        A, B = sc[pr, 0].astype(np.float64), sc[pr, 1].astype(np.float64)
    ea, pa = O.extract_features(A, H, W, 1.0, 120.0, ofe)
    eb, pb = O.extract_features(B, H, W, 1.0, 120.0, ofe)
    po, to, io = O.register_features(B[eb], B[pb], A[ea], A[pa], None, oreg)
    rot, trans = pose_diff(O, po, res[pr]["pose"])
    ok = same and (res[pr]["termination"], res[pr]["iterations"]) == (to, io) and ((rot < 1e-5 and trans < 1e-5) or to != 0)
    bad += 0 if ok else 1
    print(f"{'ok ' if ok else 'BAD'} {H:3d} x {W:4d} x {P:3d} pairs{' f32' if f32 else ''} (np {fe.neighbor_points} S {fe.number_sectors} k {reg.num_edge_neighbors}/{reg.num_plane_neighbors}): planar {len(pa)} / edge {len(ea)} features, repeat identical {same}, "
          f"pair {pr}: term {res[pr]['termination']}/{to} iters {res[pr]['iterations']}/{io} diff {rot:.1e} {trans:.1e}", flush=True)
    d_xyz.free()
    d_res.free()
print("bad:", bad)
sys.exit(1 if bad else 0)
