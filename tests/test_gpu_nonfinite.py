"""GPU: the non-finite input contract (include/loamx.h, "Non-finite input").

The reference is undefined on NaN / Inf coordinates (features-inl.h:38 sorts on curvatures computed from them, a NaN range
passes every comparison of features.cpp:30-68). Here every HOST entry point refuses them with LOAMX_ERR_BAD_PARAM before
anything is launched, a persistent index is left as it was, and the "_dev" entry points do the same once the context option
CHECK_FINITE is set. Every call returns; the context keeps working afterwards."""
import numpy as np
import pytest

from gpu_common import ctx, option
from loam_amd import capi

pytestmark = pytest.mark.gpu

H, W = 16, 1024
BAD = [np.nan, np.inf, -np.inf]


def _refused(fn):
    with pytest.raises(capi.LoamxError) as e:
        fn()
    assert e.value.status == capi.ERR_BAD_PARAM and "non-finite" in str(e.value)


@pytest.mark.parametrize("bad", BAD)
def test_a_non_finite_coordinate_in_a_scan_is_refused(bad):
    c = ctx()
    lidar = capi.LidarParams(H, W, 1.0, 120.0)
    xyz = capi.synth_scan_host(7, 0, 0, H, W, 0.01)
    ref = c.extract_features(xyz, lidar)
    for where in (0, 5 * W + 17, H * W - 1):
        for axis in range(3):
            dirty = xyz.copy()
            dirty[where, axis] = bad
            _refused(lambda: c.compute_curvature(dirty, lidar))
            _refused(lambda: c.compute_valid_points(dirty, lidar))
            _refused(lambda: c.extract_features(dirty, lidar))
    dirty32 = xyz.astype(np.float32)
    dirty32[3 * W + 3, 1] = bad
    _refused(lambda: c.extract_features(dirty32, lidar))
    again = c.extract_features(xyz, lidar)  # the context is as good as before
    assert np.array_equal(again[0], ref[0]) and np.array_equal(again[1], ref[1])


@pytest.mark.parametrize("bad", BAD)
def test_a_non_finite_coordinate_in_any_feature_set_is_refused(bad):
    c = ctx()
    lidar = capi.LidarParams(64, W, 1.0, 120.0)
    tgt, src = capi.synth_scan_host(11, 3, 0, 64, W, 0.01), capi.synth_scan_host(11, 3, 1, 64, W, 0.01)
    te, tp = c.extract_features(tgt, lidar)
    se, sp = c.extract_features(src, lidar)
    sets = [src[se], src[sp], tgt[te], tgt[tp]]
    good = c.register_features(*sets)
    for k in range(4):
        dirty = [a.copy() for a in sets]
        dirty[k][len(dirty[k]) // 2, k % 3] = bad
        _refused(lambda: c.register_features(*dirty))
        _refused(lambda: c.associate(*dirty))
    pose = np.array([0, 0, 0, 1, 0, 0, bad])
    _refused(lambda: c.register_features(*sets, init_pose=pose))
    # direct entry points of rows a16-a19
    pts = np.random.default_rng(1).normal(size=(4, 5, 3))
    pts[2, 3, 0] = bad
    _refused(lambda: c.fit_lines(pts))
    _refused(lambda: c.fit_planes(pts))
    index = c.target_index(sets[2], sets[3])
    size = c.target_index_size(index)
    q = sets[1][:8].copy()
    q[4, 2] = bad
    _refused(lambda: c.knn_search(index, 1, q, 5))
    # an insert that is refused leaves the index as it was
    add_e, add_p = sets[0][:20].copy(), sets[1][:200].copy()
    add_p[100, 1] = bad
    _refused(lambda: c.target_index_insert(index, add_e, add_p))
    add_e[3, 0], add_p[100, 1] = bad, 0.0
    _refused(lambda: c.target_index_insert(index, add_e, add_p))
    assert c.target_index_size(index) == size
    assert all(len(nb) == 5 for nb in c.knn_search(index, 1, sets[1][:8], 5))
    dirty_src = sets[1].copy()
    dirty_src[7, 0] = bad
    _refused(lambda: c.register_features_indexed(index, sets[0], dirty_src))
    through_index = c.register_features_indexed(index, sets[0], sets[1])
    c.target_index_destroy(index)
    again = c.register_features(*sets)
    assert np.array_equal(again[0], good[0]) and np.array_equal(through_index[0], good[0])


@pytest.mark.parametrize("bad", BAD)
def test_dev_entry_points_refuse_non_finite_input_when_asked_to_look(bad):
    c = ctx()
    Hs, P = 64, 3
    N = Hs * W
    lidar, fe, reg = capi.LidarParams(Hs, W, 1.0, 120.0), capi.FeatureExtractionParams(), capi.RegistrationParams()
    d_xyz, d_res = c.alloc(P * 2 * N * 24), c.alloc(P * 64)
    c.synth_scan_pairs_dev(5, 40, P, Hs, W, 0.01, d_xyz.ptr)
    c.register_scan_pairs_dev(d_xyz.ptr, P, lidar, fe, reg, d_res.ptr)
    c.synchronize()
    good = d_res.download(np.uint8, P * 64).copy()
    scans = d_xyz.download(np.float64, P * 2 * N * 3).reshape(P * 2, N, 3)
    scans[3, 12345, 2] = bad
    d_bad = c.alloc(scans.nbytes).upload(scans)
    ecap, pcap = c.edge_capacity(lidar, fe), c.planar_capacity(lidar, fe)
    ns = 2 * P
    d_ei, d_pi, d_ne, d_np = c.alloc(ns * ecap * 4), c.alloc(ns * pcap * 4), c.alloc(ns * 4), c.alloc(ns * 4)
    d_ex, d_px = c.alloc(ns * ecap * 24), c.alloc(ns * pcap * 24)
    with option("CHECK_FINITE"):
        _refused(lambda: c.register_scan_pairs_dev(d_bad.ptr, P, lidar, fe, reg, d_res.ptr))
        _refused(lambda: c.extract_features_batch_dev(d_bad.ptr, ns, lidar, fe, d_ei.ptr, d_ne.ptr, d_ex.ptr, d_pi.ptr, d_np.ptr, d_px.ptr))
        # clean input passes the check and gives the same bits as without it
        c.register_scan_pairs_dev(d_xyz.ptr, P, lidar, fe, reg, d_res.ptr)
        c.synchronize()
        assert np.array_equal(d_res.download(np.uint8, P * 64), good)
        # feature sets on the device (every scan registered against itself): a bad value in the unused capacity behind a set is
        # not looked at, one inside the set's count is found
        c.extract_features_batch_dev(d_xyz.ptr, ns, lidar, fe, d_ei.ptr, d_ne.ptr, d_ex.ptr, d_pi.ptr, d_np.ptr, d_px.ptr)
        c.synchronize()
        d_res2 = c.alloc(ns * 64)
        npl = d_np.download(np.uint32, ns)
        px = d_px.download(np.float64, ns * pcap * 3).reshape(ns, pcap, 3)

        def run_sets():
            c.register_features_batch_dev(ns, d_ex.ptr, d_ne.ptr, d_px.ptr, d_np.ptr, d_ex.ptr, d_ne.ptr, d_px.ptr, d_np.ptr, ecap, pcap, None, reg, d_res2.ptr)
            c.synchronize()

        assert npl[1] + 1 < pcap
        px[1, npl[1] + 1, 0] = bad  # behind the set
        d_px.upload(px)
        run_sets()
        px[1, npl[1] - 1, 0] = bad  # the set's last point
        d_px.upload(px)
        _refused(run_sets)
        d_res2.free()
    for b_ in (d_xyz, d_res, d_bad, d_ei, d_pi, d_ne, d_np, d_ex, d_px):
        b_.free()


def test_initial_poses_are_checked_as_exactly_seven_numbers_per_pair():
    """CHECK_FINITE once rounded n_pairs * 7 up to whole points and read up to two doubles past the caller's
    buffer. One pair, the pose followed directly by two NaNs that are not part of it: accepted; a NaN inside it: refused."""
    c = ctx()
    Hs = 64
    lidar, fe, reg = capi.LidarParams(Hs, W, 1.0, 120.0), capi.FeatureExtractionParams(), capi.RegistrationParams()
    N = Hs * W
    d_xyz = c.alloc(2 * N * 24)
    c.synth_scan_pairs_dev(5, 41, 1, Hs, W, 0.01, d_xyz.ptr)
    ecap, pcap = c.edge_capacity(lidar, fe), c.planar_capacity(lidar, fe)
    d_ei, d_pi, d_ne, d_np = c.alloc(2 * ecap * 4), c.alloc(2 * pcap * 4), c.alloc(2 * 4), c.alloc(2 * 4)
    d_ex, d_px = c.alloc(2 * ecap * 24), c.alloc(2 * pcap * 24)
    c.extract_features_batch_dev(d_xyz.ptr, 2, lidar, fe, d_ei.ptr, d_ne.ptr, d_ex.ptr, d_pi.ptr, d_np.ptr, d_px.ptr)
    c.synchronize()
    d_res = c.alloc(64)
    pose = np.array([0, 0, 0, 1, 0, 0, 0, np.nan, np.nan])
    d_init = c.alloc(pose.nbytes).upload(pose)

    def run():  # source = scan 1, target = scan 0 (sets one scan apart)
        c.register_features_batch_dev(1, d_ex.ptr + ecap * 24, d_ne.ptr + 4, d_px.ptr + pcap * 24, d_np.ptr + 4, d_ex.ptr, d_ne.ptr, d_px.ptr, d_np.ptr,
                                      ecap, pcap, d_init.ptr, reg, d_res.ptr)
        c.synchronize()
        return d_res.download(np.uint8, 64).copy()

    plain = run()
    with option("CHECK_FINITE"):
        assert np.array_equal(run(), plain)
        pose[6] = np.inf
        d_init.upload(pose)
        _refused(run)
    for b_ in (d_xyz, d_ei, d_pi, d_ne, d_np, d_ex, d_px, d_res, d_init):
        b_.free()
