"""GPU parity: loam::extractFeatures path (C ABI -> HIP kernels) against the CPU oracle.
Bar: curvature bit-exact, mask exact, feature index sequences identical."""
import numpy as np
import pytest

import reference_kats as K
from gpu_common import ctx, option, to_capi_fe
from loam_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kat", K.fe_kats(), ids=lambda k: k["name"])
def test_reference_kats_on_gpu(oracle, kat):
    lidar = capi.LidarParams(kat["H"], kat["W"], kat["rmin"], kat["rmax"])
    fe = capi.FeatureExtractionParams(*K.KAT_FE_PARAMS)
    ofe = oracle.FeParams(*K.KAT_FE_PARAMS)
    c = ctx().compute_curvature(kat["pts"], lidar, fe)
    m = ctx().compute_valid_points(kat["pts"], lidar, fe)
    assert np.array_equal(c.view(np.uint64), oracle.compute_curvature(kat["pts"], kat["H"], kat["W"], ofe).view(np.uint64))
    assert np.array_equal(m, oracle.compute_valid_points(kat["pts"], kat["H"], kat["W"], kat["rmin"], kat["rmax"], ofe))
    if "curvature" in kat:
        for i, v in kat["curvature"].items():
            assert abs(c[i] - v) < 1e-9
    else:
        for i in kat["invalid"]:
            assert not m[i]
        for i in kat["valid"]:
            assert m[i]


def test_empty_scan_and_size_mismatch():
    lidar0 = capi.LidarParams(0, 0, 0.1, 100.0)
    e, p = ctx().extract_features(np.zeros((0, 3)), lidar0)
    assert len(e) == 0 and len(p) == 0
    with pytest.raises(capi.LoamxError) as err:
        ctx().compute_curvature(np.zeros((10, 3)), capi.LidarParams(1, 11, 0.1, 10.0))
    assert err.value.status == capi.ERR_SCAN_SIZE
    assert "does not match provided lidar parameters (1 x 11)" in str(err.value)


PARAM_SETS = [(3, 6, 10, 50, 100.0, 1.0, 0.5, 1.0), (5, 4, 2, 7, 50.0, 0.5, 0.3, 0.5), (1, 3, 0, 3, 10.0, 2.0, 0.5, 1.0),
              (2, 1, 1000, 1000, 20.0, 5.0, 0.5, 1.0)]


@pytest.mark.parametrize("H,W,seed", [(64, 1024, 1), (16, 256, 3), (8, 100, 5), (4, 37, 9), (2, 2048, 11), (128, 2048, 2)])
def test_extraction_parity_synthetic(oracle, H, W, seed):
    xyz = capi.synth_scan_host(seed, 0, 0, H, W, 0.01)
    lidar = capi.LidarParams(H, W, 1.0, 120.0)
    for params in PARAM_SETS:
        ofe = oracle.FeParams(*params)
        fe = capi.FeatureExtractionParams(*params)
        c = ctx().compute_curvature(xyz, lidar, fe)
        m = ctx().compute_valid_points(xyz, lidar, fe)
        assert np.array_equal(c.view(np.uint64), oracle.compute_curvature(xyz, H, W, ofe).view(np.uint64))
        assert np.array_equal(m, oracle.compute_valid_points(xyz, H, W, 1.0, 120.0, ofe))
        e, p = ctx().extract_features(xyz, lidar, fe)
        oe, op = oracle.extract_features(xyz, H, W, 1.0, 120.0, ofe)  # the reference's std::sort order, ties included
        assert np.array_equal(e, oe) and np.array_equal(p, op)


def test_dropouts_and_out_of_range(oracle):
    xyz = capi.synth_scan_host(4, 0, 0, 16, 512, 0.01)
    rng = np.random.default_rng(0)
    xyz[rng.random(len(xyz)) < 0.05] = 0.0
    xyz[rng.random(len(xyz)) < 0.02] *= 30.0
    lidar = capi.LidarParams(16, 512, 1.0, 120.0)
    c = ctx().compute_curvature(xyz, lidar)
    m = ctx().compute_valid_points(xyz, lidar)
    assert np.array_equal(c.view(np.uint64), oracle.compute_curvature(xyz, 16, 512).view(np.uint64))
    assert np.array_equal(m, oracle.compute_valid_points(xyz, 16, 512, 1.0, 120.0))
    e, p = ctx().extract_features(xyz, lidar)
    oe, op = oracle.extract_features(xyz, 16, 512, 1.0, 120.0)  # (the zeroed points tie with each other: reference order)
    assert np.array_equal(e, oe) and np.array_equal(p, op)


@pytest.mark.parametrize("H,W", [(32, 512), (8, 2048), (64, 1024)])  # 2048 columns: more than 64 picks per sector, two per lane
def test_tie_policy_noise_free(oracle, H, W):
    """Row a7: noise-free scans hold exact curvature ties (356 on a 64 x 1024 scan); where a tie can decide
    a pick or the output order the kernels replay the scan line in the order libstdc++'s std::sort gives the
    reference (features-inl.h:38), so the index sequences are the oracle's std::sort ones — not the stable order."""
    xyz = capi.synth_scan_host(2, 0, 0, H, W, 0.0)
    oe, op = oracle.extract_features(xyz, H, W, 1.0, 120.0)  # std::sort, as the reference
    se, sp, ties = oracle.extract_features(xyz, H, W, 1.0, 120.0, stable=True)
    assert ties > 0
    r0 = ctx().extract_counters()[0]
    e, p = ctx().extract_features(xyz, capi.LidarParams(H, W, 1.0, 120.0))
    assert np.array_equal(e, oe) and np.array_equal(p, op)
    replayed = ctx().extract_counters()[0] - r0
    assert 0 < replayed <= H
    if not (np.array_equal(oe, se) and np.array_equal(op, sp)):
        assert not (np.array_equal(e, se) and np.array_equal(p, sp))
    # ... and the arg-max fallback kernel detects and replays them too
    with option("NO_MIS_SELECT"):
        e2, p2 = ctx().extract_features(xyz, capi.LidarParams(H, W, 1.0, 120.0))
    assert np.array_equal(e2, oe) and np.array_equal(p2, op)
    assert ctx().extract_counters()[0] - r0 > replayed


@pytest.mark.parametrize("params", PARAM_SETS)
def test_quantised_scan_ties_in_reference_order(oracle, params):
    """coordinates rounded to 1/64 m (a sensor that reports fixed-point ranges): ties in every sector"""
    H, W = 16, 1024
    xyz = np.round(capi.synth_scan_host(9, 1, 0, H, W, 0.01) * 64.0) / 64.0
    ofe, fe = oracle.FeParams(*params), capi.FeatureExtractionParams(*params)
    oe, op = oracle.extract_features(xyz, H, W, 1.0, 120.0, ofe)
    _, _, ties = oracle.extract_features(xyz, H, W, 1.0, 120.0, ofe, stable=True)
    assert ties > 0
    e, p = ctx().extract_features(xyz, capi.LidarParams(H, W, 1.0, 120.0), fe)
    assert np.array_equal(e, oe) and np.array_equal(p, op)


def test_forced_replay_and_forced_fallback_change_nothing(oracle):
    """The two rare paths on ordinary (tie-free) input: every line through the std::sort replay
    (LOAMX_FORCE_TIE_REPLAY), and every scan line after the first giving up its wait so that the fallback kernel
    gathers the features (LOAMX_FORCE_SCAN_GIVEUP): same sequences, no error, events counted."""
    H, W = 64, 1024
    xyz = capi.synth_scan_host(17, 0, 0, H, W, 0.01)
    lidar = capi.LidarParams(H, W, 1.0, 120.0)
    oe, op = oracle.extract_features(xyz, H, W, 1.0, 120.0)
    r0, f0 = ctx().extract_counters()
    e, p = ctx().extract_features(xyz, lidar)
    assert np.array_equal(e, oe) and np.array_equal(p, op)
    assert ctx().extract_counters() == (r0, f0)  # noisy scan: neither path ran
    with option("FORCE_TIE_REPLAY"):
        e, p = ctx().extract_features(xyz, lidar)
    assert np.array_equal(e, oe) and np.array_equal(p, op)
    assert ctx().extract_counters()[0] == r0 + H
    f0 = ctx().extract_counters()[1]  # (tied lines also hand their scan to the fallback compaction)
    with option("FORCE_SCAN_GIVEUP"):
        e, p = ctx().extract_features(xyz, lidar)
    assert np.array_equal(e, oe) and np.array_equal(p, op)
    assert ctx().extract_counters()[1] == f0 + 1
    # the batch entry point with point copies, asynchronously: still the oracle's, and the flag does not stick
    N, ns = H * W, 3
    c = ctx()
    fe = capi.FeatureExtractionParams()
    scans = np.stack([capi.synth_scan_host(200 + s, 0, 0, H, W, 0.01) for s in range(ns)])
    d_xyz = c.alloc(scans.nbytes).upload(scans)
    ecap, pcap = c.edge_capacity(lidar, fe), c.planar_capacity(lidar, fe)
    d_ei, d_pi, d_ne, d_np = c.alloc(ns * ecap * 4), c.alloc(ns * pcap * 4), c.alloc(ns * 4), c.alloc(ns * 4)
    d_ex, d_px = c.alloc(ns * ecap * 24), c.alloc(ns * pcap * 24)
    for forced in (True, False):
        f1 = c.extract_counters()[1]
        with option("FORCE_SCAN_GIVEUP", 1 if forced else 0):
            c.extract_features_batch_dev(d_xyz.ptr, ns, lidar, fe, d_ei.ptr, d_ne.ptr, d_ex.ptr, d_pi.ptr, d_np.ptr, d_px.ptr)
            c.synchronize()
        assert c.extract_counters()[1] == f1 + (1 if forced else 0)
        ne, npl = d_ne.download(np.uint32, ns), d_np.download(np.uint32, ns)
        pi = d_pi.download(np.uint32, ns * pcap).reshape(ns, pcap)
        px = d_px.download(np.float64, ns * pcap * 3).reshape(ns, pcap, 3)
        ei = d_ei.download(np.uint32, ns * ecap).reshape(ns, ecap)
        for s in range(ns):
            oe, op = oracle.extract_features(scans[s], H, W, 1.0, 120.0)
            assert np.array_equal(ei[s, :ne[s]], oe) and np.array_equal(pi[s, :npl[s]], op)
            assert np.array_equal(px[s, :npl[s]], scans[s][op])
    for b_ in (d_xyz, d_ei, d_pi, d_ne, d_np, d_ex, d_px):
        b_.free()


def test_device_generator_bit_identical_and_batch_extract(oracle):
    H, W, n_pairs, seed = 16, 256, 3, 21
    N = H * W
    c = ctx()
    lidar = capi.LidarParams(H, W, 1.0, 120.0)
    fe = capi.FeatureExtractionParams()
    d_xyz = c.alloc(n_pairs * 2 * N * 24)
    c.synth_scan_pairs_dev(seed, 5, n_pairs, H, W, 0.01, d_xyz.ptr)
    c.synchronize()
    dev = d_xyz.download(np.float64, n_pairs * 2 * N * 3).reshape(n_pairs * 2, N, 3)
    for pr in range(n_pairs):
        for which in (0, 1):
            host = capi.synth_scan_host(seed, 5 + pr, which, H, W, 0.01)
            assert np.array_equal(dev[2 * pr + which].view(np.uint64), host.view(np.uint64))
    ecap, pcap = c.edge_capacity(lidar, fe), c.planar_capacity(lidar, fe)
    ns = n_pairs * 2
    d_ei, d_pi = c.alloc(ns * ecap * 4), c.alloc(ns * pcap * 4)
    d_ne, d_np = c.alloc(ns * 4), c.alloc(ns * 4)
    d_ex, d_px = c.alloc(ns * ecap * 24), c.alloc(ns * pcap * 24)
    c.extract_features_batch_dev(d_xyz.ptr, ns, lidar, fe, d_ei.ptr, d_ne.ptr, d_ex.ptr, d_pi.ptr, d_np.ptr, d_px.ptr)
    c.synchronize()
    ne, npl = d_ne.download(np.uint32, ns), d_np.download(np.uint32, ns)
    ei = d_ei.download(np.uint32, ns * ecap).reshape(ns, ecap)
    pi = d_pi.download(np.uint32, ns * pcap).reshape(ns, pcap)
    ex = d_ex.download(np.float64, ns * ecap * 3).reshape(ns, ecap, 3)
    px = d_px.download(np.float64, ns * pcap * 3).reshape(ns, pcap, 3)
    for s in range(ns):
        oe, op = oracle.extract_features(dev[s], H, W, 1.0, 120.0)
        assert np.array_equal(ei[s, :ne[s]], oe) and np.array_equal(pi[s, :npl[s]], op)
        assert np.array_equal(ex[s, :ne[s]], dev[s][oe]) and np.array_equal(px[s, :npl[s]], dev[s][op])
    for b in (d_xyz, d_ei, d_pi, d_ne, d_np, d_ex, d_px):
        b.free()


@pytest.mark.parametrize("H,W,seed", [(64, 1024, 1), (8, 100, 5), (4, 37, 9)])
def test_fp32_input_path_equals_oracle_on_widened_scan(oracle, H, W, seed):
    """float scans (PCL points). The reference's FieldAccessor widens each coordinate to double
    (common.h:55-60) before any arithmetic, so the oracle on the widened scan is the expected result, bit for bit."""
    xyz32 = capi.synth_scan_host(seed, 0, 0, H, W, 0.01).astype(np.float32)
    wide = xyz32.astype(np.float64)
    lidar = capi.LidarParams(H, W, 1.0, 120.0)
    for params in PARAM_SETS[:2]:
        ofe = oracle.FeParams(*params)
        fe = capi.FeatureExtractionParams(*params)
        c = ctx().compute_curvature(xyz32, lidar, fe)
        m = ctx().compute_valid_points(xyz32, lidar, fe)
        assert np.array_equal(c.view(np.uint64), oracle.compute_curvature(wide, H, W, ofe).view(np.uint64))
        assert np.array_equal(m, oracle.compute_valid_points(wide, H, W, 1.0, 120.0, ofe))
        e, p = ctx().extract_features(xyz32, lidar, fe)
        se, sp = oracle.extract_features(wide, H, W, 1.0, 120.0, ofe)
        assert np.array_equal(e, se) and np.array_equal(p, sp)
        e64, p64 = ctx().extract_features(wide, lidar, fe)
        assert np.array_equal(e, e64) and np.array_equal(p, p64)


def test_fp32_scan_pairs_equal_fp64_pipeline_on_widened_scans():
    """Device-resident float scans through extract x2 + register: the same poses, bit for bit, as the FP64
    pipeline fed with the widened scans."""
    H, W, n_pairs, seed = 32, 512, 4, 77
    N = H * W
    c = ctx()
    lidar, fe, reg = capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams(), capi.RegistrationParams()
    scans = np.stack([capi.synth_scan_host(seed, pr, which, H, W, 0.01) for pr in range(n_pairs) for which in (0, 1)])
    s32 = np.ascontiguousarray(scans.astype(np.float32))
    s64 = np.ascontiguousarray(s32.astype(np.float64))
    d32, d64 = c.alloc(s32.nbytes), c.alloc(s64.nbytes)
    d32.upload(s32)
    d64.upload(s64)
    r32, r64 = c.alloc(n_pairs * 64), c.alloc(n_pairs * 64)
    c.register_scan_pairs_dev(d32.ptr, n_pairs, lidar, fe, reg, r32.ptr, f32=True)
    c.register_scan_pairs_dev(d64.ptr, n_pairs, lidar, fe, reg, r64.ptr)
    c.synchronize()
    a, b = r32.download(np.uint8, n_pairs * 64), r64.download(np.uint8, n_pairs * 64)
    assert np.array_equal(a, b)
    assert (a.view(capi.RESULT_DTYPE)["iterations"] > 0).all()
    for buf in (d32, d64, r32, r64):
        buf.free()


def test_fused_and_unfused_compaction_agree(oracle):
    """The selection kernel writes the final feature arrays itself (chained scan over the lines of a scan) when
    number_sectors <= 64; otherwise, and with LOAMX_NO_FUSED_COMPACT=1, compact_kernel does. All three routes
    must give the oracle's sequences."""
    H, W = 64, 1024
    xyz = capi.synth_scan_host(31, 0, 0, H, W, 0.01)
    lidar = capi.LidarParams(H, W, 1.0, 120.0)
    for params in [(3, 6, 10, 50, 100.0, 1.0, 0.5, 1.0), (3, 80, 2, 3, 100.0, 1.0, 0.5, 1.0), (2, 64, 1, 4, 50.0, 1.0, 0.5, 1.0)]:
        ofe = oracle.FeParams(*params)
        fe = capi.FeatureExtractionParams(*params)
        se, sp = oracle.extract_features(xyz, H, W, 1.0, 120.0, ofe)
        e1, p1 = ctx().extract_features(xyz, lidar, fe)
        with option("NO_FUSED_COMPACT"):
            e2, p2 = ctx().extract_features(xyz, lidar, fe)
        assert np.array_equal(e1, se) and np.array_equal(p1, sp), params
        assert np.array_equal(e2, se) and np.array_equal(p2, sp), params


def test_batch_extract_many_scans_fused_offsets(oracle):
    """Many scans in one launch: every scan's chained scan is independent of its neighbours' (counts, indices and
    point copies of all scans against the oracle)."""
    H, W, ns = 16, 512, 40
    N = H * W
    c = ctx()
    lidar, fe = capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams()
    scans = np.stack([capi.synth_scan_host(100 + s, 0, s & 1, H, W, 0.01) for s in range(ns)])
    d_xyz = c.alloc(scans.nbytes)
    d_xyz.upload(scans)
    ecap, pcap = c.edge_capacity(lidar, fe), c.planar_capacity(lidar, fe)
    d_ei, d_pi = c.alloc(ns * ecap * 4), c.alloc(ns * pcap * 4)
    d_ne, d_np = c.alloc(ns * 4), c.alloc(ns * 4)
    d_ex, d_px = c.alloc(ns * ecap * 24), c.alloc(ns * pcap * 24)
    c.extract_features_batch_dev(d_xyz.ptr, ns, lidar, fe, d_ei.ptr, d_ne.ptr, d_ex.ptr, d_pi.ptr, d_np.ptr, d_px.ptr)
    c.synchronize()
    ne, npl = d_ne.download(np.uint32, ns), d_np.download(np.uint32, ns)
    ei = d_ei.download(np.uint32, ns * ecap).reshape(ns, ecap)
    pi = d_pi.download(np.uint32, ns * pcap).reshape(ns, pcap)
    px = d_px.download(np.float64, ns * pcap * 3).reshape(ns, pcap, 3)
    for s in range(ns):
        oe, op = oracle.extract_features(scans[s], H, W, 1.0, 120.0)
        assert ne[s] == len(oe) and npl[s] == len(op)
        assert np.array_equal(ei[s, :ne[s]], oe) and np.array_equal(pi[s, :npl[s]], op)
        assert np.array_equal(px[s, :npl[s]], scans[s][op])
    for b in (d_xyz, d_ei, d_pi, d_ne, d_np, d_ex, d_px):
        b.free()


def test_fused_extraction_kernel_equals_the_separate_kernels(oracle):
    """extract_fused_kernel (LOAMX_FUSED_EXTRACT=1: curvature + validity + selection + compaction in one pass over the
    scan) against the oracle and against the default two-kernel path: indices, counts and point copies, double and
    float input, with ties (replayed from the workspace copy the fused kernel leaves for tied lines) and without."""
    H, W, ns = 64, 1024, 5
    N = H * W
    c = ctx()
    lidar, fe = capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams()
    scans = np.stack([capi.synth_scan_host(300 + s, 0, s & 1, H, W, 0.0 if s == 2 else 0.01) for s in range(ns)])  # scan 2: noise free (ties)
    ecap, pcap = c.edge_capacity(lidar, fe), c.planar_capacity(lidar, fe)
    for f32 in (False, True):
        data = scans.astype(np.float32) if f32 else scans
        wide = data.astype(np.float64)
        d_xyz = c.alloc(data.nbytes).upload(data)
        out = {}
        for fused in (True, False):
            d_ei, d_pi, d_ne, d_np = c.alloc(ns * ecap * 4), c.alloc(ns * pcap * 4), c.alloc(ns * 4), c.alloc(ns * 4)
            d_ex, d_px = c.alloc(ns * ecap * 24), c.alloc(ns * pcap * 24)
            r0 = c.extract_counters()[0]
            with option("FUSED_EXTRACT", 1 if fused else 0):
                c.extract_features_batch_dev(d_xyz.ptr, ns, lidar, fe, d_ei.ptr, d_ne.ptr, d_ex.ptr, d_pi.ptr, d_np.ptr, d_px.ptr, f32=f32)
                c.synchronize()
            assert c.extract_counters()[0] > r0  # the noise-free scan has tied lines
            ne, npl = d_ne.download(np.uint32, ns), d_np.download(np.uint32, ns)
            ei = d_ei.download(np.uint32, ns * ecap).reshape(ns, ecap)
            pi = d_pi.download(np.uint32, ns * pcap).reshape(ns, pcap)
            ex = d_ex.download(np.float64, ns * ecap * 3).reshape(ns, ecap, 3)
            px = d_px.download(np.float64, ns * pcap * 3).reshape(ns, pcap, 3)
            out[fused] = [(ei[s, :ne[s]].copy(), pi[s, :npl[s]].copy(), ex[s, :ne[s]].copy(), px[s, :npl[s]].copy()) for s in range(ns)]
            for b_ in (d_ei, d_pi, d_ne, d_np, d_ex, d_px):
                b_.free()
        d_xyz.free()
        for s in range(ns):
            oe, op = oracle.extract_features(wide[s], H, W, 1.0, 120.0)
            for fused in (True, False):
                e, p, xe, xp = out[fused][s]
                assert np.array_equal(e, oe) and np.array_equal(p, op), (f32, fused, s)
                assert np.array_equal(xe, wide[s][oe]) and np.array_equal(xp, wide[s][op])


def test_fused_extraction_with_the_rare_paths_forced(oracle):
    """the one-pass kernel's own tie dump (curvature / mask written out for replay_kernel) and its give-up
    handling only run when somebody opts in. FUSED_EXTRACT with every line replayed, and with every line after the first
    giving up its wait: the oracle's sequences and point copies, events counted."""
    H, W, ns = 32, 1024, 3
    c = ctx()
    lidar, fe = capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams()
    scans = np.stack([capi.synth_scan_host(400 + s, 0, s & 1, H, W, 0.01) for s in range(ns)])
    d_xyz = c.alloc(scans.nbytes).upload(scans)
    ecap, pcap = c.edge_capacity(lidar, fe), c.planar_capacity(lidar, fe)
    d_ei, d_pi, d_ne, d_np = c.alloc(ns * ecap * 4), c.alloc(ns * pcap * 4), c.alloc(ns * 4), c.alloc(ns * 4)
    d_ex, d_px = c.alloc(ns * ecap * 24), c.alloc(ns * pcap * 24)
    for forced in ("FORCE_TIE_REPLAY", "FORCE_SCAN_GIVEUP"):
        r0, f0 = c.extract_counters()
        with option("FUSED_EXTRACT"), option(forced):
            c.extract_features_batch_dev(d_xyz.ptr, ns, lidar, fe, d_ei.ptr, d_ne.ptr, d_ex.ptr, d_pi.ptr, d_np.ptr, d_px.ptr)
            c.synchronize()
        r1, f1 = c.extract_counters()
        assert (r1 - r0 == ns * H) if forced == "FORCE_TIE_REPLAY" else (f1 - f0 == 1)
        ne, npl = d_ne.download(np.uint32, ns), d_np.download(np.uint32, ns)
        ei = d_ei.download(np.uint32, ns * ecap).reshape(ns, ecap)
        pi = d_pi.download(np.uint32, ns * pcap).reshape(ns, pcap)
        ex = d_ex.download(np.float64, ns * ecap * 3).reshape(ns, ecap, 3)
        px = d_px.download(np.float64, ns * pcap * 3).reshape(ns, pcap, 3)
        for s in range(ns):
            oe, op = oracle.extract_features(scans[s], H, W, 1.0, 120.0)
            assert np.array_equal(ei[s, :ne[s]], oe) and np.array_equal(pi[s, :npl[s]], op), (forced, s)
            assert np.array_equal(ex[s, :ne[s]], scans[s][oe]) and np.array_equal(px[s, :npl[s]], scans[s][op])
    for b_ in (d_xyz, d_ei, d_pi, d_ne, d_np, d_ex, d_px):
        b_.free()


@pytest.mark.parametrize("form", ["rows", "NO_SPLIT_CURV", "NO_ROW_SELECT", "FUSED_ROWS"])
@pytest.mark.parametrize("forced", [None, "FORCE_TIE_REPLAY", "FORCE_SCAN_GIVEUP", "NO_FUSED_COMPACT"])
def test_selection_forms_against_the_oracle(oracle, form, forced):
    """Round 5: the selection with four scan lines per wavefront (select_rows_kernel, the default: curvature handed over as
    hi words | lo words with the validity in the sign bit; NO_SPLIT_CURV: as doubles + validity bytes), one line per wavefront
    (NO_ROW_SELECT: select_mis_kernel) and the opt-in fused form that computes curvature and validity itself (FUSED_ROWS; its
    tied lines are recomputed by curvature_tied_kernel for the std::sort replay): double and float input, one noise-free scan
    (ties in most lines), the rare paths forced — index sequences and point copies are the oracle's."""
    import contextlib
    H, W, ns = 64, 1024, 3
    c = ctx()
    lidar, fe = capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams()
    scans = np.stack([capi.synth_scan_host(300 + s, 0, s & 1, H, W, 0.0 if s == 2 else 0.01) for s in range(ns)])  # scan 2: noise free (ties)
    ecap, pcap = c.edge_capacity(lidar, fe), c.planar_capacity(lidar, fe)
    for f32 in (False, True):
        data = scans.astype(np.float32) if f32 else scans
        wide = data.astype(np.float64)
        d_xyz = c.alloc(data.nbytes).upload(data)
        d_ei, d_pi, d_ne, d_np = c.alloc(ns * ecap * 4), c.alloc(ns * pcap * 4), c.alloc(ns * 4), c.alloc(ns * 4)
        d_ex, d_px = c.alloc(ns * ecap * 24), c.alloc(ns * pcap * 24)
        r0 = c.extract_counters()[0]
        with (option(form) if form != "rows" else contextlib.nullcontext()), (option(forced) if forced else contextlib.nullcontext()):
            c.extract_features_batch_dev(d_xyz.ptr, ns, lidar, fe, d_ei.ptr, d_ne.ptr, d_ex.ptr, d_pi.ptr, d_np.ptr, d_px.ptr, f32=f32)
            c.synchronize()
        assert c.extract_counters()[0] > r0  # the noise-free scan has tied lines
        ne, npl = d_ne.download(np.uint32, ns), d_np.download(np.uint32, ns)
        ei = d_ei.download(np.uint32, ns * ecap).reshape(ns, ecap)
        pi = d_pi.download(np.uint32, ns * pcap).reshape(ns, pcap)
        ex = d_ex.download(np.float64, ns * ecap * 3).reshape(ns, ecap, 3)
        px = d_px.download(np.float64, ns * pcap * 3).reshape(ns, pcap, 3)
        for s in range(ns):
            oe, op = oracle.extract_features(wide[s], H, W, 1.0, 120.0)
            assert np.array_equal(ei[s, :ne[s]], oe) and np.array_equal(pi[s, :npl[s]], op), (form, forced, f32, s)
            assert np.array_equal(ex[s, :ne[s]], wide[s][oe]) and np.array_equal(px[s, :npl[s]], wide[s][op])
        for b_ in (d_xyz, d_ei, d_pi, d_ne, d_np, d_ex, d_px):
            b_.free()


def test_context_options_are_per_context_and_named():
    """loamx_ctx_set_option: unknown names are refused, a switch lives on the context it was set on, and the
    environment is not consulted after loamx_ctx_create."""
    import os
    c = ctx()
    with pytest.raises(capi.LoamxError):
        c.set_option("NO_SUCH_SWITCH", 1)
    assert c.get_option("NO_MOMENTS") == 0
    os.environ["LOAMX_NO_MOMENTS"] = "1"
    try:
        assert c.get_option("NO_MOMENTS") == 0  # (read once, at creation)
        other = capi.Context(0)
        assert other.get_option("NO_MOMENTS") == 1  # (the default of a context created now)
        other.set_option("NO_MOMENTS", 0)
        other.set_option("NO_PACKED_GRID", 1)
        assert c.get_option("NO_PACKED_GRID") == 0
        other.close()
    finally:
        del os.environ["LOAMX_NO_MOMENTS"]


@pytest.mark.parametrize("seed", range(80))
def test_extraction_fuzz_against_the_oracle(oracle, seed):
    """Random scan shapes and random FeatureExtractionParams (neighbor_points 1..6, 1..9 sectors, caps 0..60, thresholds
    on either side of the data, quantised and dropped-out ranges): curvature bits, mask and both index sequences against
    the oracle, whichever kernels the parameters select (bitmask MIS with one or two picks per lane, arg-max fallback,
    separate compaction, tie replay)."""
    rng = np.random.default_rng(5000 + seed)
    H, W = int(rng.integers(1, 10)), int(rng.choice([12, 37, 64, 100, 333, 512, 1024, 1500, 2048]))
    xyz = capi.synth_scan_host(int(rng.integers(1, 1000)), int(rng.integers(0, 5)), int(rng.integers(0, 2)), H, W, float(rng.choice([0.0, 0.01, 0.05])))
    if rng.random() < 0.4:
        xyz = np.round(xyz * 32.0) / 32.0  # fixed-point sensor: ties
    if rng.random() < 0.4:
        xyz[rng.random(len(xyz)) < 0.05] = 0.0  # drop-outs
    np_ = int(rng.integers(1, 7))
    params = (np_, int(rng.integers(1, 10)), int(rng.integers(0, 25)), int(rng.integers(0, 61)), float(rng.choice([5.0, 50.0, 100.0, 1e4])),
              float(rng.choice([0.05, 1.0, 20.0])), float(rng.choice([0.1, 0.5])), float(rng.choice([0.02, 1.0])))
    if W < 2 * np_ + 2:
        params = (1,) + params[1:]
    ofe, fe = oracle.FeParams(*params), capi.FeatureExtractionParams(*params)
    lidar = capi.LidarParams(H, W, 1.0, 120.0)
    c = ctx().compute_curvature(xyz, lidar, fe)
    m = ctx().compute_valid_points(xyz, lidar, fe)
    assert np.array_equal(c.view(np.uint64), oracle.compute_curvature(xyz, H, W, ofe).view(np.uint64)), (seed, params)
    assert np.array_equal(m, oracle.compute_valid_points(xyz, H, W, 1.0, 120.0, ofe)), (seed, params)
    e, p = ctx().extract_features(xyz, lidar, fe)
    oe, op = oracle.extract_features(xyz, H, W, 1.0, 120.0, ofe)
    assert np.array_equal(e, oe) and np.array_equal(p, op), (seed, H, W, params)
