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
    rot, trans = pose_diff(oracle, po, res[pr]["pose"])  # (whatever the termination type)
    assert rot < 1e-5 and trans < 1e-5, (to, rot, trans)


def _scene(rng, n_e, n_p):
    """target sets on a few planes / lines of a room, source = a moved noisy subset"""
    def planes(n):
        parts = []
        for _ in range(3):
            o, u, v = rng.normal(size=3) * 3, rng.normal(size=3), rng.normal(size=3)
            parts.append(o + np.outer(rng.uniform(-4, 4, n), u / np.linalg.norm(u)) + np.outer(rng.uniform(-4, 4, n), v / np.linalg.norm(v)))
        return np.concatenate(parts)[rng.permutation(3 * n)[:n]]
    def lines(n):
        return np.concatenate([rng.normal(size=3) * 4 + np.outer(np.linspace(-3, 3, max(2, n // 3 + 1)), rng.normal(size=3)) for _ in range(3)])[:n]
    import reference_kats as K
    te, tp = (lines(n_e) if n_e else np.zeros((0, 3))), planes(n_p)
    ax = rng.normal(size=3)
    T = K.pose7(K.quat_angle_axis(rng.uniform(0, 0.03), ax / np.linalg.norm(ax)), rng.normal(size=3) * 0.03)
    se = K.transform_points(T, te[rng.random(len(te)) < 0.9]) if len(te) else te
    sp = K.transform_points(T, tp[rng.random(len(tp)) < 0.9])
    return (np.ascontiguousarray(se + rng.normal(size=se.shape) * 0.002), np.ascontiguousarray(sp + rng.normal(size=sp.shape) * 0.002),
            np.ascontiguousarray(te), np.ascontiguousarray(tp))


def test_ragged_batch_equals_single_pair_calls(oracle):
    """loamx_register_features_batch_dev with very different set sizes per pair (edge sets on both sides of the 512-point
    brute-force threshold, planar sets from 50 to 25 000 points — on both sides of the 20 480-point index-build threshold —
    in ONE batch): every pair's record equals what the single-pair entry point returns for it, bit for bit."""
    rng = np.random.default_rng(99)
    c = ctx()
    sizes = [(40, 50), (700, 3000), (0, 900), (300, 25000), (520, 21000), (511, 20480), (3, 6000), (900, 150),
             (100, 12000), (600, 22000), (60, 400), (513, 9000)]
    scenes = [_scene(rng, ne, np_) for ne, np_ in sizes]
    P = len(scenes)
    es = max(max(len(s[0]), len(s[2])) for s in scenes)
    ps = max(max(len(s[1]), len(s[3])) for s in scenes)
    reg = capi.RegistrationParams()
    reg.min_associations = 20
    bufs = []
    for k, stride in ((0, es), (1, ps), (2, es), (3, ps)):
        pts = np.zeros((P, stride, 3))
        cnt = np.zeros(P, np.uint32)
        for p, s in enumerate(scenes):
            pts[p, : len(s[k])] = s[k]
            cnt[p] = len(s[k])
        bufs.append((c.alloc(pts.nbytes).upload(pts.view(np.uint8).reshape(-1)), c.alloc(cnt.nbytes).upload(cnt.view(np.uint8))))
    d_res = c.alloc(P * 64)
    got = []
    for _ in range(2):
        c.register_features_batch_dev(P, bufs[0][0].ptr, bufs[0][1].ptr, bufs[1][0].ptr, bufs[1][1].ptr, bufs[2][0].ptr, bufs[2][1].ptr,
                                      bufs[3][0].ptr, bufs[3][1].ptr, es, ps, None, reg, d_res.ptr)
        c.synchronize()
        got.append(d_res.download(np.uint8, P * 64).copy())
    for b in bufs:
        b[0].free(), b[1].free()
    d_res.free()
    assert np.array_equal(got[0], got[1])
    res = got[0].view(capi.RESULT_DTYPE)
    for p, (se, sp, te, tp) in enumerate(scenes):
        pose, term, iters = c.register_features(se, sp, te, tp, reg=reg)
        assert (res[p]["termination"], res[p]["iterations"]) == (term, iters), p
        assert np.array_equal(res[p]["pose"], np.asarray(pose)), (p, sizes[p])
    for p in (1, 5, 8):
        oreg = oracle.RegParams()
        oreg.min_associations = 20
        po, to, io = oracle.register_features(*scenes[p], None, oreg)
        assert (res[p]["termination"], res[p]["iterations"]) == (to, io), p
        rot, trans = pose_diff(oracle, po, res[p]["pose"])  # (whatever the termination type)
        assert rot < 1e-5 and trans < 1e-5, (p, to, rot, trans)
