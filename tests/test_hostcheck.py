"""CPU-only: the HIP kernels' per-thread math (the __host__ __device__ headers under
loam_amd/csrc, run serially by tests/hostcheck) against the oracle. This is what lets the kernels
be trusted before they reach a GPU; the GPU parity tests proper are in test_gpu_*.py."""
import numpy as np
import pytest

import hostcheck_lib as Hc
import reference_kats as K


def _pdiff(O, a, b):
    d = O.pose_compose(O.pose_inverse(a), b)
    return O.quat_angular_distance(d[:4], [0, 0, 0, 1.0]), float(np.linalg.norm(d[4:]))


@pytest.mark.parametrize("kat", K.fe_kats(), ids=lambda k: k["name"])
def test_kernel_math_on_reference_kats(oracle, kat):
    fe = Hc.fe_params(*K.KAT_FE_PARAMS)
    curv, mask = Hc.curvature_valid(kat["pts"], kat["H"], kat["W"], kat["rmin"], kat["rmax"], fe)
    ofe = oracle.FeParams(*K.KAT_FE_PARAMS)
    assert np.array_equal(curv.view(np.uint64), oracle.compute_curvature(kat["pts"], kat["H"], kat["W"], ofe).view(np.uint64))
    assert np.array_equal(mask, oracle.compute_valid_points(kat["pts"], kat["H"], kat["W"], kat["rmin"], kat["rmax"], ofe))


@pytest.mark.parametrize("H,W,seed", [(16, 256, 3), (64, 1024, 1), (8, 100, 5), (4, 37, 9)])
def test_extraction_bit_exact_on_synthetic(oracle, H, W, seed):
    xyz = Hc.synth_scan(seed, 0, 0, H, W, 0.01)
    for params in [(3, 6, 10, 50, 100.0, 1.0, 0.5, 1.0), (5, 4, 2, 7, 50.0, 0.5, 0.3, 0.5), (1, 3, 0, 3, 10.0, 2.0, 0.5, 1.0)]:
        ofe = oracle.FeParams(*params)
        fe = Hc.fe_params(*params)
        curv, mask = Hc.curvature_valid(xyz, H, W, 1.0, 120.0, fe)
        assert np.array_equal(curv.view(np.uint64), oracle.compute_curvature(xyz, H, W, ofe).view(np.uint64))
        assert np.array_equal(mask, oracle.compute_valid_points(xyz, H, W, 1.0, 120.0, ofe))
        e, p = Hc.select(curv, mask, H, W, fe)
        oe, op = oracle.extract_features(xyz, H, W, 1.0, 120.0, ofe)  # (std::sort order, ties included)
        assert np.array_equal(e, oe) and np.array_equal(p, op)


@pytest.mark.parametrize("H,W,seed,sigma", [(16, 1024, 1, 0.01), (8, 256, 3, 0.01), (8, 512, 2, 0.0), (4, 2048, 4, 0.01), (4, 700, 6, 0.01)])
def test_bitmask_mis_selection_matches_oracle(oracle, H, W, seed, sigma):
    """Lane-level emulation of select_mis_kernel (64-bit lane masks, local-maxima rounds, cap by
    priority order) against the oracle's sort + greedy walk, including tied curvatures."""
    xyz = Hc.synth_scan(seed, 0, 0, H, W, sigma)
    checked = 0
    for params in [(3, 6, 10, 50, 100.0, 1.0, 0.5, 1.0), (5, 4, 2, 7, 50.0, 0.5, 0.3, 0.5), (2, 3, 0, 3, 10.0, 2.0, 0.5, 1.0),
                   (4, 6, 10, 20, 50.0, 0.8, 0.5, 1.0), (3, 1, 1000, 1000, 20.0, 5.0, 0.5, 1.0), (3, 6, 10, 5, 100.0, 1.0, 0.5, 1.0)]:
        fe = Hc.fe_params(*params)
        curv, mask = Hc.curvature_valid(xyz, H, W, 1.0, 120.0, fe)
        r = Hc.select_mis(curv, mask, H, W, fe)
        if r is None:  # parameters outside the MIS kernel's domain: the general kernel is used
            continue
        oe, op = oracle.extract_features(xyz, H, W, 1.0, 120.0, oracle.FeParams(*params))  # the reference's order, ties included
        assert np.array_equal(r[0], oe) and np.array_equal(r[1], op)
        checked += 1
    assert checked >= 4


def test_extraction_tie_order_on_noise_free_scan(oracle):
    # noise-free synthetic scans contain exact curvature ties: the lines they can decide something on are
    # replayed in the order libstdc++'s std::sort gives the reference (row a7)
    xyz = Hc.synth_scan(2, 0, 0, 32, 512, 0.0)
    ofe = oracle.FeParams()
    curv, mask = Hc.curvature_valid(xyz, 32, 512, 1.0, 120.0, Hc.fe_params())
    oe, op = oracle.extract_features(xyz, 32, 512, 1.0, 120.0, ofe)
    se, sp, ties = oracle.extract_features(xyz, 32, 512, 1.0, 120.0, ofe, stable=True)
    assert ties > 0 and not (np.array_equal(oe, se) and np.array_equal(op, sp))  # the tie order is observable here
    before = Hc.replayed_lines()
    e, p = Hc.select(curv, mask, 32, 512, Hc.fe_params())
    assert np.array_equal(e, oe) and np.array_equal(p, op)
    e, p = Hc.select_mis(curv, mask, 32, 512, Hc.fe_params())
    assert np.array_equal(e, oe) and np.array_equal(p, op)
    assert 0 < Hc.replayed_lines() - before < 2 * 32  # some lines, not all of them


@pytest.mark.parametrize("W,params", [(1024, (3, 6, 10, 50, 100.0, 1.0, 0.5, 1.0)), (512, (3, 6, 3, 8, 60.0, 20.0, 0.5, 1.0)),
                                      (700, (4, 5, 10, 20, 50.0, 10.0, 0.5, 1.0)), (256, (2, 3, 0, 3, 10.0, 30.0, 0.5, 1.0)),
                                      (2048, (3, 6, 10, 50, 100.0, 1.0, 0.5, 1.0))])
def test_tie_detection_is_sufficient_on_quantised_curvature(W, params):
    """Curvature arrays with few distinct values (ties everywhere) and random masks: the selection emulations (tie
    detection + replay where a tie can decide something, the fast formulation elsewhere) must equal the literal
    restatement with the real std::sort — in particular on the lines the detection lets through."""
    rng = np.random.default_rng(W + params[0])
    fe = Hc.fe_params(*params)
    H = 6
    replayed = fast = 0
    for levels, pmask in [(4, 0.9), (30, 0.8), (300, 0.7), (5000, 0.95), (10 ** 6, 0.9)]:
        for trial in range(6):
            curv = rng.integers(0, levels, H * W).astype(np.float64) * (150.0 / levels)
            mask = rng.random(H * W) < pmask
            np_ = params[0]
            cols = np.arange(H * W) % W
            mask &= (cols >= np_) & (cols < W - np_)  # valid points keep neighbor_points from the line ends (features.cpp:22)
            curv[(cols < np_) | (cols >= W - np_)] = -1.0
            ref = Hc.select_stdsort(curv, mask, H, W, fe)
            before = Hc.replayed_lines()
            got = Hc.select(curv, mask, H, W, fe)
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), (levels, trial)
            r = Hc.select_mis(curv, mask, H, W, fe)
            if r is not None:
                assert np.array_equal(r[0], ref[0]) and np.array_equal(r[1], ref[1]), (levels, trial)
            n = Hc.replayed_lines() - before
            replayed += n
            fast += (2 if r is not None else 1) * H - n
    assert replayed > 0 and fast > 0


def test_out_of_range_and_dropout_points(oracle):
    xyz = Hc.synth_scan(4, 0, 0, 8, 256, 0.01)
    rng = np.random.default_rng(0)
    drop = rng.random(len(xyz)) < 0.05
    xyz[drop] = 0.0  # no-return points
    xyz[rng.random(len(xyz)) < 0.02] *= 30.0  # beyond max range
    ofe = oracle.FeParams()
    curv, mask = Hc.curvature_valid(xyz, 8, 256, 1.0, 120.0, Hc.fe_params())
    assert np.array_equal(curv.view(np.uint64), oracle.compute_curvature(xyz, 8, 256, ofe).view(np.uint64))
    assert np.array_equal(mask, oracle.compute_valid_points(xyz, 8, 256, 1.0, 120.0, ofe))
    e, p = Hc.select(curv, mask, 8, 256, Hc.fe_params())
    se, sp, _ = oracle.extract_features(xyz, 8, 256, 1.0, 120.0, ofe, stable=True)
    assert np.array_equal(e, se) and np.array_equal(p, sp)


def test_fits_match_oracle(oracle):
    rng = np.random.default_rng(0)
    for _ in range(500):
        k = int(rng.integers(4, 6))
        base = rng.normal(size=3) * 5
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        pts = base + rng.normal(size=(k, 3)) * 0.3
        pts -= np.outer((pts - base) @ n, n) * 0.98
        no, do, ao = oracle.fit_plane(pts)
        nh, dh, ah = Hc.fit_plane(pts)
        assert np.abs(no - nh).max() < 1e-12 and abs(do - dh) < 1e-12 * max(1, abs(do)) and abs(ao - ah) < 1e-12
        k = int(rng.integers(3, 6))
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        pts = base + np.outer(rng.normal(size=k), d) + rng.normal(size=(k, 3)) * 0.01
        a, b, cond = oracle.fit_line(pts)
        ah_, bh_ = Hc.fit_line(pts)
        err = min(max(np.abs(a - ah_).max(), np.abs(b - bh_).max()), max(np.abs(a - bh_).max(), np.abs(b - ah_).max()))
        assert err < 1e-10
        assert cond == np.finfo(np.float64).max


def test_grid_knn_is_exact(oracle):
    rng = np.random.default_rng(1)
    xyz = Hc.synth_scan(1, 0, 0, 32, 512, 0.01)
    e, p = oracle.extract_features(xyz, 32, 512, 1.0, 120.0)
    for pts, radii in ((xyz[p], (2.0, 0.3, -1.0)), (xyz[e], (1.0, -1.0))):
        tree = oracle.KDTree(pts)
        for _ in range(400):
            q = pts[rng.integers(len(pts))] + rng.normal(size=3) * rng.choice([0.05, 0.5, 3.0, 30.0])
            for R in radii:
                for k in (1, 5, 8):
                    a = tree.knn(q, k, R)
                    assert np.array_equal(a, Hc.knn(pts, q, k, R))
                    assert np.array_equal(a, oracle.knn_bruteforce(pts, q, k, R))
    # empty and tiny target sets
    assert len(Hc.knn(np.zeros((0, 3)), [0, 0, 0], 5, 1.0)) == 0
    assert np.array_equal(Hc.knn(np.array([[1.0, 0, 0], [0.5, 0, 0]]), [0, 0, 0], 5, -1.0), [1, 0])
    assert Hc.knn_mismatches() == 0


def test_keyed_knn_ties_and_radius_edges(oracle):
    """The kernels' keyed collector (distance and position folded into one double) must hand every
    query it cannot decide - exact distance ties, a distance within truncation of the radius - to the
    exact collector, and agree with it everywhere else."""
    # integer lattice: every query at a lattice point has 6 / 12 / 8 equidistant neighbours
    g = np.arange(-3, 4, dtype=np.float64)
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    before = Hc.knn_fallbacks()  # (exact ties: the grid orders by (d2, index) like the brute-force oracle; nanoflann's
    # tie order depends on its traversal - DESIGN.md §2.3)
    for q in ([0, 0, 0], [1, -1, 2], [0.5, 0.5, 0.5], [0.25, 0, 0], [3, 3, 3], [10, 0, 0]):
        for k in (1, 5, 8):
            for R in (-1.0, 1.0, 1.5, 2.0):  # R = 1.0: the 6 face neighbours sit exactly ON the (strict) radius
                a = oracle.knn_bruteforce(lattice, np.array(q, dtype=np.float64), k, R)
                assert np.array_equal(a, Hc.knn(lattice, np.array(q, dtype=np.float64), k, R)), (q, k, R)
    assert Hc.knn_fallbacks() > before  # the tie cases really went through the fallback
    # coincident points (d2 == 0 keys are denormal bit patterns) and duplicates
    dup = np.array([[0.0, 0, 0]] * 7 + [[1.0, 0, 0]] * 3)
    for k in (1, 5, 8):
        assert np.array_equal(oracle.knn_bruteforce(dup, np.zeros(3), k, 2.0), Hc.knn(dup, np.zeros(3), k, 2.0))
    # non-finite target coordinates never enter a result
    bad = np.array([[0.0, 0, 0], [np.nan, 0, 0], [0.5, 0, 0], [np.inf, 0, 0], [0.25, 0, 0]])
    assert np.array_equal(Hc.knn(bad, np.zeros(3), 5, -1.0)[:3], [0, 4, 2])
    assert Hc.knn_mismatches() == 0


def test_fp32_preselection_is_exact_in_adverse_cases(oracle):
    """Round 1 of the kernels' search runs on single-precision copies with 32-bit keys; whatever it cannot
    certify (error bound, running-number overflow, near ties) must fall through to the FP64 paths."""
    rng = np.random.default_rng(7)
    before = Hc.knn_mismatches()
    # (a) 3000 points inside one grid cell: more than 63 candidate batches for one query
    dense = rng.uniform(0.0, 0.2, size=(3000, 3))
    dense = np.vstack([dense, rng.uniform(-5, 5, size=(50, 3))])
    for q in (dense[5] + 1e-3, np.array([0.1, 0.1, 0.1])):
        for k in (1, 5, 8):
            assert np.array_equal(oracle.knn_bruteforce(dense, q, k, 2.0), Hc.knn(dense, q, k, 2.0))
    # (b) far from the origin and a huge extent: the FP32 offsets lose the digits that separate neighbours
    far = rng.normal(size=(400, 3)) * 0.05 + np.array([4.0e6, -7.0e6, 1.0e5])
    far = np.vstack([far, far[0] + np.array([3.0e4, 0, 0])])  # stretches the grid to 30 km
    for _ in range(30):
        q = far[rng.integers(400)] + rng.normal(size=3) * 0.01
        assert np.array_equal(oracle.knn_bruteforce(far, q, 5, -1.0), Hc.knn(far, q, 5, -1.0))
    # (c) 5th and 6th neighbour separated by one part in 1e9 / 1e12, and an exact tie
    base = np.array([[0.1, 0, 0], [0, 0.2, 0], [0, 0, 0.3], [0.4, 0, 0], [0, 0.5, 0]])
    for rel in (1e-9, 1e-12, 0.0):
        pts = np.vstack([base, [[0, 0, 0.5 * (1 + rel)]], rng.uniform(2, 3, size=(20, 3))])
        got = Hc.knn(pts, np.zeros(3), 5, -1.0)
        assert np.array_equal(oracle.knn_bruteforce(pts, np.zeros(3), 5, -1.0), got), rel
    assert Hc.knn_mismatches() == before


def test_plane_moments_equal_per_record_sums():
    """Sum over plane records of cost / J^T f / J^T J == the same from the 13x13 moment matrix, wherever the
    validity bound holds (Huber inactive); where the bound fails the kernels stream the records instead."""
    rng = np.random.default_rng(0)
    N = 5000
    v = rng.uniform(-10, 10, size=(N, 3))
    n = rng.normal(size=(N, 3))
    n /= np.linalg.norm(n, axis=1)[:, None]
    d = (v * n).sum(1) + rng.normal(size=N) * 0.02
    for x in ([0, 0, 0, 1, 0, 0, 0], [0.003, -0.002, 0.004, 0.99999, 0.02, -0.01, 0.03], [1e-4, 2e-4, -1e-4, 1.0, 1e-3, 2e-3, -1e-3]):
        direct, mom, s0max, valid = Hc.plane_moments(v, n, d, x)
        assert valid and s0max < 0.2
        scale = np.abs(direct).max()
        assert np.abs(direct - mom).max() <= 1e-11 * scale
    # a large step: some |s| exceed the Huber threshold, the sums differ, and the bound says so
    direct, mom, _, valid = Hc.plane_moments(v, n, d, [0.02, 0.01, -0.03, 0.9993, 0.1, 0.2, -0.1])
    assert not valid and not np.allclose(direct, mom, rtol=1e-6)
    # records far from their plane from the start
    d2 = d.copy()
    d2[::50] += 2.0
    assert not Hc.plane_moments(v, n, d2, [0, 0, 0, 1, 0, 0, 0])[3]


def test_relative_moment_bound_is_rigorous():
    """plane_moments_valid_rel (round 4: the first ICF iteration's moments are taken at its first candidate r): whenever it
    says yes for a candidate x, every record the moments hold (|s_i(r)| <= 0.5) has |s_i(x)| below the Huber threshold — on
    random records, reference candidates of the size of a first LOAM update (<= 0.1 rad, <= 0.5 m), refinements and wild
    steps around them, unit and slightly non-unit quaternions. And it is not vacuous: it says yes for the refinements a
    solve really makes. The moment form c_i . phi(x) of a residual agrees with the point formula to rounding."""
    rng = np.random.default_rng(5)
    said_yes = said_no = 0
    for trial in range(400):
        N = 400
        v = rng.normal(size=(N, 3)) * rng.uniform(2, 15)
        n = rng.normal(size=(N, 3))
        n /= np.linalg.norm(n, axis=1)[:, None]

        def quat(angle, scale=1.0):
            ax = rng.normal(size=3)
            ax /= np.linalg.norm(ax)
            return np.concatenate([np.sin(angle / 2) * ax, [np.cos(angle / 2)]]) * scale

        r = np.concatenate([quat(rng.uniform(0, 0.1), rng.choice([1.0, 1.0 + 1e-6, 0.9999])), rng.normal(size=3) * rng.uniform(0, 0.3)])
        # planes through the points as moved by r, give or take the noise of a scan: small residuals AT r, large ones at 0
        u, w = r[:3], r[3]
        t = 2.0 * np.cross(u, v)
        moved = v + w * t + np.cross(u, t) + r[4:]
        d = (moved * n).sum(1) + rng.normal(size=N) * rng.choice([0.01, 0.05, 0.3])
        kind = trial % 3
        if kind == 0:    # a refinement of r
            dq = quat(rng.uniform(0, 0.01))
            x = np.concatenate([r[:4] + dq[:4] * 0 + rng.normal(size=4) * rng.uniform(0, 3e-3), r[4:] + rng.normal(size=3) * rng.uniform(0, 0.03)])
        elif kind == 1:  # anything between the identity and twice r
            s_ = rng.uniform(0, 2)
            x = np.array([0, 0, 0, 1.0, 0, 0, 0]) * (1 - s_) + r * s_
        else:            # r itself
            x = r.copy()
        ok, sref, sx, form_err = Hc.moments_rel(v, n, d, x, r)
        assert form_err <= 1e-9 * (1.0 + np.abs(v).max() ** 2)
        assert sref <= 0.5
        if ok:
            said_yes += 1
            assert sx < 1.0, (trial, sref, sx)
        else:
            said_no += 1
        if kind == 2:
            assert ok  # at the reference itself the bound is max |s_i(r)| <= 0.5
    assert said_yes > 150 and said_no > 30, (said_yes, said_no)


@pytest.mark.parametrize("case", K.REGISTRATION_CASES, ids=lambda c: c["name"])
def test_registration_math_on_reference_scenes(oracle, case):
    tgt_e, tgt_p = K.registration_scene()
    src_e = K.transform_points(case["source_T_target"], tgt_e)
    src_p = K.transform_points(case["source_T_target"], tgt_p)
    prm = oracle.RegParams()
    if case["max_iter"] is not None:
        prm.max_iterations = case["max_iter"]
    po, to, io = oracle.register_features(src_e, src_p, tgt_e, tgt_p, case["init"], prm)
    ph, th, ih = Hc.register(src_e, src_p, tgt_e, tgt_p, case["init"], Hc.conv_reg(prm))
    assert Hc.knn_mismatches() == 0  # keyed fast path == exact collector on every query of the run
    assert (to, io) == (th, ih)
    rot, trans = _pdiff(oracle, po, ph)
    assert rot < 1e-5 and trans < 1e-5  # the north-star tolerance; observed ~1e-16
    rot_err, trans_err = K.registration_error(case["source_T_target"], ph, oracle.pose_compose, oracle.quat_angular_distance)
    assert rot_err < case["rot_tol"] and np.all(np.abs(trans_err) < case["trans_tol"])


def test_registration_math_on_synthetic_pair(oracle):
    H, W = 32, 512
    A = Hc.synth_scan(7, 1, 0, H, W, 0.01)
    B = Hc.synth_scan(7, 1, 1, H, W, 0.01)
    ea, pa = oracle.extract_features(A, H, W, 1.0, 120.0)
    eb, pb = oracle.extract_features(B, H, W, 1.0, 120.0)
    po, to, io = oracle.register_features(B[eb], B[pb], A[ea], A[pa])
    fb0 = Hc.knn_fallbacks()
    ph, th, ih = Hc.register(B[eb], B[pb], A[ea], A[pa])
    assert Hc.knn_mismatches() == 0
    print('keyed kNN fallbacks on a noisy pair:', Hc.knn_fallbacks() - fb0)
    assert (to, io) == (th, ih)
    rot, trans = _pdiff(oracle, po, ph)
    assert rot < 1e-5 and trans < 1e-5
    rot, trans = _pdiff(oracle, Hc.synth_pose(7, 1), ph)
    assert rot < 1e-2 and trans < 3e-2  # ICF convergence thresholds are 1e-3 rad / 1e-2 m per step


def test_plane_only_and_insufficient(oracle):
    e, p = K.plane_only_scene()
    ph, th, ih = Hc.register(e, p, e, p)
    assert oracle.quat_angular_distance(ph[:4], [0, 0, 0, 1.0]) < 1e-4 and np.all(np.abs(ph[4:]) < 1e-3)
    ph, th, ih = Hc.register(e, p + np.array([100.0, 0, 0]), e, p)
    assert th == 2 and ih == 0 and np.allclose(ph, [0, 0, 0, 1, 0, 0, 0])


# ---- row a7: libstdc++'s std::sort replayed (extract_math.h: stl_sort) vs the real one ---------------------------------
def _sort_pair(c):
    import ctypes as C
    lib = Hc.lib()
    c = np.ascontiguousarray(c, dtype=np.float64)
    a, b = np.zeros(len(c), np.uint32), np.zeros(len(c), np.uint32)
    dp, up = C.POINTER(C.c_double), C.POINTER(C.c_uint32)
    lib.hostcheck_stl_sort(c.ctypes.data_as(dp), C.c_uint64(len(c)), a.ctypes.data_as(up))
    lib.hostcheck_std_sort(c.ctypes.data_as(dp), C.c_uint64(len(c)), b.ctypes.data_as(up))
    return a, b


def test_stl_sort_replays_std_sort_on_ties():
    rng = np.random.default_rng(77)
    for n in (0, 1, 2, 3, 15, 16, 17, 18, 33, 170, 174, 341, 343, 1024, 4096):
        for levels in (1, 2, 3, 7, 40, 1000):  # few distinct values = many ties
            for _ in range(6):
                c = rng.integers(0, levels, n).astype(np.float64)
                if levels == 7:
                    c[rng.random(n) < 0.3] = -1.0  # the line-end curvature (features-inl.h:66-68)
                a, b = _sort_pair(c)
                assert np.array_equal(a, b), (n, levels)
                assert np.array_equal(np.sort(a), np.arange(n))
        c = rng.normal(size=n)  # tie free: any correct sort gives this
        a, b = _sort_pair(c)
        assert np.array_equal(a, b) and np.array_equal(a, np.argsort(c, kind="stable"))
        for c in (np.arange(n, dtype=float), np.arange(n, dtype=float)[::-1], np.zeros(n), np.r_[np.zeros(n // 2), np.ones(n - n // 2)]):
            a, b = _sort_pair(c)
            assert np.array_equal(a, b)


def test_stl_sort_replays_the_heap_sort_branch():
    import ctypes as C
    lib = Hc.lib()
    lib.hostcheck_heap_sorts.restype = C.c_uint64
    for n in (170, 174, 343, 1024, 4096):
        c = np.zeros(n)
        lib.hostcheck_killer_input(C.c_uint64(n), c.ctypes.data_as(C.POINTER(C.c_double)))
        before = lib.hostcheck_heap_sorts()
        a, b = _sort_pair(c)  # (the adversary's input drives std::sort to its depth limit: 2 lg n levels of bad splits)
        assert np.array_equal(a, b)
        assert lib.hostcheck_heap_sorts() > before, n  # the heap-sort branch really ran
        q = np.floor(c / 3.0)  # the same shape with ties
        a, b = _sort_pair(q)
        assert np.array_equal(a, b)


def test_fit_line_on_degenerate_lattice_neighbourhoods(oracle):
    """Neighbours on a lattice give an exactly diagonal covariance with EQUAL eigenvalues; "the largest" is then whatever
    Eigen's selection sort of the eigenvalues leaves last [RECALLED]: (0, s, s) -> z, (s, s, 0) -> x, (s, 0, s) -> z,
    (s, s, s) -> z. Oracle and kernel math must agree on these (found by the association fuzz test, round 2)."""
    c = np.array([1.25, -0.5, 2.0])
    h = 0.25
    ex, ey, ez = np.eye(3) * h
    cases = [([c, c + ey, c - ey, c + ez, c - ez], 2),            # (0, s, s) -> z
             ([c, c + ex, c - ex, c + ey, c - ey], 0),            # (s, s, 0) -> x
             ([c, c + ex, c - ex, c + ez, c - ez], 2),            # (s, 0, s) -> z
             ([c + ex, c - ex, c + ey, c - ey, c], 0)]
    six = [c + ex, c - ex, c + ey, c - ey, c + ez][:5]
    for pts, axis in cases:
        pts = np.array(pts)
        a, b, _ = oracle.fit_line(pts)
        d = (a - b) / np.linalg.norm(a - b)
        assert abs(abs(d[axis]) - 1.0) < 1e-12, (pts, d)
        ha, hb = Hc.fit_line(pts)
        assert np.allclose(ha, a, atol=1e-15) and np.allclose(hb, b, atol=1e-15)


@pytest.mark.parametrize("seed", range(8))
def test_knn_fuzz_every_path_is_exact(oracle, seed):
    """Random clouds (dense blobs, planes, shells with a far cluster, lattices) x radii (on / off / tiny): the search the
    kernels run — FP32 round 1 in its lean form, its wide retry, the keyed FP64 rounds, the exact collector — against
    brute force, and the fast paths against the exact collector query by query (mismatch counter). The dense blob with
    the radius filter off is the case that overran the lean walk's trip budget in round 2 (an admissible row was
    dropped silently instead of handing the query to the queue)."""
    rng = np.random.default_rng(900 + seed)
    kind = seed % 4
    n = int(rng.choice([300, 3000, 9000]))
    if kind == 0:
        pts = rng.normal(size=(n, 3)) * rng.uniform(0.5, 5.0)
    elif kind == 1:
        u, v = rng.normal(size=3), rng.normal(size=3)
        pts = np.outer(rng.uniform(-4, 4, n), u / np.linalg.norm(u)) + np.outer(rng.uniform(-4, 4, n), v / np.linalg.norm(v)) + rng.normal(size=(n, 3)) * 0.01
    elif kind == 2:
        d = rng.normal(size=(n, 3))
        pts = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(5, 30)
        pts[: n // 10] = rng.normal(size=(n // 10, 3)) * 0.2 + 100.0
    else:
        g = int(round(n ** (1 / 3))) + 1
        pts = np.stack(np.meshgrid(*[np.arange(g) * 0.25] * 3), -1).reshape(-1, 3)[:n] + rng.normal(size=3)
    pts = np.ascontiguousarray(pts)
    before = Hc.knn_mismatches()
    for R in (2.0, -1.0, 0.3):
        for q in pts[rng.integers(0, len(pts), 60)] + rng.normal(size=(60, 3)) * 0.05:
            for k in (5, 8, 13):
                assert np.array_equal(oracle.knn_bruteforce(pts, q, k, R), Hc.knn(pts, q, k, R)), (seed, R, k)
    assert Hc.knn_mismatches() == before


def test_lean_second_round_serves_the_sparse_queries(oracle):
    """Round 3: a query whose 3x3x3 block does not hold k points closer than its faces (sparse neighbourhood) gets the lean
    FP32 search of the 5x5x5 block before the FP64 search over all rounds. On a sparse shell nearly every query is such a
    query: most must be finished by the second lean round, and every answer must be brute force's."""
    rng = np.random.default_rng(77)
    d = rng.normal(size=(6000, 3))
    pts = np.ascontiguousarray(d / np.linalg.norm(d, axis=1, keepdims=True) * 12.0 + rng.normal(size=(6000, 3)) * 0.02)
    r0, q0 = Hc.knn_round2()
    before = Hc.knn_mismatches()
    for q in pts[rng.integers(0, len(pts), 400)] + rng.normal(size=(400, 3)) * 0.05:
        assert np.array_equal(oracle.knn_bruteforce(pts, q, 5, 2.0), Hc.knn(pts, q, 5, 2.0))
    r1, q1 = Hc.knn_round2()
    assert Hc.knn_mismatches() == before
    assert q1 - q0 > 100 and (r1 - r0) > 0.8 * (q1 - q0), (r1 - r0, q1 - q0)
