#!/usr/bin/env python3
"""Measurements for the BASELINE.json configs that are not the bench headline:
   config 2 (single 64x1024 pair latency) and config 5 (128x2048 scan against a ~1M-point local map).
   Prints one JSON object; run on a GPU box."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # (initialised before the library's context: pinned host memory for the host-memory entry point)
torch.cuda.init()
from loam_amd import capi
import oracle_lib as O

out = {}
ctx = capi.Context(0)
fe, reg = capi.FeatureExtractionParams(), capi.RegistrationParams()

# ---- config 2: one 64x1024 pair, device resident and host-buffer paths
H, W = 64, 1024
N = H * W
lidar = capi.LidarParams(H, W, 1.0, 120.0)
d_xyz, d_res = ctx.alloc(2 * N * 24), ctx.alloc(64)
ctx.synth_scan_pairs_dev(20240311, 0, 1, H, W, 0.01, d_xyz.ptr)
ctx.synchronize()
for _ in range(3):
    ctx.register_scan_pairs_dev(d_xyz.ptr, 1, lidar, fe, reg, d_res.ptr)
ctx.synchronize()
t0 = time.perf_counter()
K = 20
for _ in range(K):
    ctx.register_scan_pairs_dev(d_xyz.ptr, 1, lidar, fe, reg, d_res.ptr)
ctx.synchronize()
lat_dev = (time.perf_counter() - t0) / K
scans = d_xyz.download(np.float64, 2 * N * 3).reshape(2, N, 3)
t0 = time.perf_counter()
for _ in range(K):
    ea, pa = ctx.extract_features(scans[0], lidar, fe)
    eb, pb = ctx.extract_features(scans[1], lidar, fe)
    pose, term, iters = ctx.register_features(scans[1][eb], scans[1][pb], scans[0][ea], scans[0][pa])
lat_host = (time.perf_counter() - t0) / K
# (round 6) the same pair through the host-memory entry point, from a pinned buffer (loamx_register_scan_pairs)
pinned = torch.empty(2 * N * 3, dtype=torch.float64, pin_memory=True)
pinned.copy_(torch.from_numpy(scans.reshape(-1)))
one = np.zeros(1, dtype=capi.RESULT_DTYPE)
for _ in range(5):
    ctx.register_scan_pairs(pinned.numpy(), 1, lidar, fe, reg, out=one)
t0 = time.perf_counter()
for _ in range(50):
    ctx.register_scan_pairs(pinned.numpy(), 1, lidar, fe, reg, out=one)
lat_stream = (time.perf_counter() - t0) / 50
t0 = time.perf_counter()
oea, opa = O.extract_features(scans[0], H, W, 1.0, 120.0)
oeb, opb = O.extract_features(scans[1], H, W, 1.0, 120.0)
opose, oterm, oiters = O.register_features(scans[1][oeb], scans[1][opb], scans[0][oea], scans[0][opa])
lat_cpu = time.perf_counter() - t0
d = O.pose_compose(O.pose_inverse(opose), pose)
out["config2_single_pair_64x1024"] = dict(
    gpu_device_resident_ms=round(lat_dev * 1e3, 3), gpu_host_buffers_ms=round(lat_stream * 1e3, 3),
    gpu_host_buffers_note="loamx_register_scan_pairs from a pinned buffer (scan pair in, pose out)",
    gpu_host_feature_entry_points_ms=round(lat_host * 1e3, 3),  # extract_features x2 + register_features with numpy gathers on the host (round 5's figure)
    cpu_oracle_one_core_ms=round(lat_cpu * 1e3, 2), icf_iterations=int(iters),
    se3_diff_vs_oracle=[O.quat_angular_distance(d[:4], [0, 0, 0, 1.0]), float(np.linalg.norm(d[4:]))])

# ---- config 5: 128x2048 scan registered against a local map of ~1M planar points
H5, W5 = 128, 2048
lidar5 = capi.LidarParams(H5, W5, 1.0, 120.0)
src = capi.synth_scan_host(99, 0, 1, H5, W5, 0.01)
t0 = time.perf_counter()
e5, p5 = ctx.extract_features(src, lidar5, fe)
t_ext = time.perf_counter() - t0
oe5, op5 = O.extract_features(src, H5, W5, 1.0, 120.0)
# map: planar/edge features of many scans taken from nearby poses (different noise seeds), all in frame A
maps_p, maps_e = [], []
k = 0
while sum(len(m) for m in maps_p) < 1_000_000:
    s = capi.synth_scan_host(1000 + k, 0, 0, H5, W5, 0.01)
    oe, op = O.extract_features(s, H5, W5, 1.0, 120.0)
    maps_p.append(s[op]); maps_e.append(s[oe]); k += 1
map_p, map_e = np.concatenate(maps_p), np.concatenate(maps_e)
t0 = time.perf_counter()
pose5, term5, it5 = ctx.register_features(src[e5], src[p5], map_e, map_p)
t_first = time.perf_counter() - t0
t0 = time.perf_counter()
pose5, term5, it5 = ctx.register_features(src[e5], src[p5], map_e, map_p)
t_reg = time.perf_counter() - t0
t0 = time.perf_counter()
idx = ctx.target_index(map_e, map_p)
t_index = time.perf_counter() - t0
ctx.register_features_indexed(idx, src[e5], src[p5])
t0 = time.perf_counter()
for _ in range(5):
    posei, termi, iti = ctx.register_features_indexed(idx, src[e5], src[p5])
t_indexed = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
opose5, oterm5, oit5 = O.register_features(src[oe5], src[op5], map_e, map_p)
t_cpu = time.perf_counter() - t0
d = O.pose_compose(O.pose_inverse(opose5), pose5)
out["config5_128x2048_vs_1M_map"] = dict(
    map_planar_points=int(len(map_p)), map_edge_points=int(len(map_e)), source_planar=int(len(p5)), source_edge=int(len(e5)),
    feature_index_sequences_equal=bool(np.array_equal(e5, oe5) and np.array_equal(p5, op5)),
    gpu_extract_ms=round(t_ext * 1e3, 2), gpu_register_ms=round(t_reg * 1e3, 2), gpu_index_build_ms=round(t_index * 1e3, 2),
    gpu_register_with_persistent_index_ms=round(t_indexed * 1e3, 2), indexed_result_bit_identical=bool(np.array_equal(posei, pose5)), gpu_register_first_call_ms=round(t_first * 1e3, 2),
    cpu_oracle_register_ms=round(t_cpu * 1e3, 1), termination=[int(term5), int(oterm5)], iterations=[int(it5), int(oit5)],
    se3_diff_vs_oracle=[O.quat_angular_distance(d[:4], [0, 0, 0, 1.0]), float(np.linalg.norm(d[4:]))])
print(json.dumps(out, indent=1))
