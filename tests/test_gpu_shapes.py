"""GPU: batches of other sensor shapes and parameter sets through the scan-pair entry point — the asynchronous,
multi-stream path with feature-set sizes on both sides of the kernels' thresholds (512 / 20 480 points per set, 64 / 128
picks per sector, sets that take the scratch-based index builds). Every batch runs twice in one context (bit-identical
results asked: a race between streams shows up as a difference or as a fault) and one pair of it is checked against the
CPU oracle. tools/stress_shapes.py is the randomised long form."""
import numpy as np
import pytest

from gpu_common import ctx, pose_diff
from loam_amd import capi

pytestmark = pytest.mark.gpu

CASES = [
    # H, W, pairs, extraction overrides, registration overrides
    (128, 2048, 9, {}, {}),
    (128, 512, 8, {}, {}),
    (16, 1800, 8, {"neighbor_points": 2}, {"num_plane_neighbors": 8}),
    (64, 512, 40, {"neighbor_points": 4, "number_sectors": 8}, {"num_edge_neighbors": 8}),
    (32, 1024, 24, {"number_sectors": 3, "max_planar_feats_per_sector": 70}, {"num_edge_neighbors": 3, "max_plane_neighbor_dist": 1.0}),
    (64, 2048, 3, {"neighbor_points": 5}, {}),
]


@pytest.mark.parametrize("H,W,P,fe_over,reg_over", CASES)
def test_batches_of_other_shapes_twice_and_against_the_oracle(oracle, H, W, P, fe_over, reg_over):
    c = ctx()
    N = H * W
    lidar, fe, reg = capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams(), capi.RegistrationParams()
    ofe, oreg = oracle.FeParams(), oracle.RegParams()
    for k, v in fe_over.items():
        setattr(fe, k, v), setattr(ofe, k, v)
    for k, v in reg_over.items():
        setattr(reg, k, v), setattr(oreg, k, v)
    seed, first = 424242, 17
    d_xyz, d_res = c.alloc(P * 2 * N * 24), c.alloc(P * 64)
    c.synth_scan_pairs_dev(seed, first, P, H, W, 0.01, d_xyz.ptr)
    runs = []
    for _ in range(2):
        c.register_scan_pairs_dev(d_xyz.ptr, P, lidar, fe, reg, d_res.ptr)
        c.synchronize()
        runs.append(d_res.download(np.uint8, P * 64).copy())
    d_xyz.free()
    d_res.free()
    assert np.array_equal(runs[0], runs[1])
    res = runs[0].view(capi.RESULT_DTYPE)
    pr = P // 2
    A = capi.synth_scan_host(seed, first + pr, 0, H, W, 0.01)
    B = capi.synth_scan_host(seed, first + pr, 1, H, W, 0.01)
    ea, pa = oracle.extract_features(A, H, W, 1.0, 120.0, ofe)
    eb, pb = oracle.extract_features(B, H, W, 1.0, 120.0, ofe)
    po, to, io = oracle.register_features(B[eb], B[pb], A[ea], A[pa], None, oreg)
    assert (res[pr]["termination"], res[pr]["iterations"]) == (to, io)
    if to == capi.CONVERGED:
        rot, trans = pose_diff(oracle, po, res[pr]["pose"])
        assert rot < 1e-5 and trans < 1e-5, (rot, trans)
