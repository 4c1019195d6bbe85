"""GPU: BASELINE configs[2]/[3] at full size on one card.
  * the RCCL gather behind the C ABI (loamx_comm_*, loamx_gather_results_dev), exercised with a 1-rank communicator;
  * the 8-shard x 1 024-pair plan of configs[3] run shard by shard on this GPU: the concatenation equals ONE
    8 192-pair call bit for bit (pairs are independent units: the sharding can not change a result);
  * the 1 024-pair batch of configs[2] against the CPU oracle on a 64-pair sample;
  * configs[5] at its real size: a 128 x 2048 scan against a 1.02 M-point map, plain call and persistent index;
  * the two routes of the LM loop kernel (moments / streaming) against each other and the oracle;
  * a batch of 128-beam scans (feature sets above 20 480 points: the index builds that need scratch memory)."""
import os

import numpy as np
import pytest

from gpu_common import ctx, option, pose_diff
from loam_amd import capi

pytestmark = pytest.mark.gpu

H, W, SEED = 64, 1024, 20240311  # the bench workload (bench.py)
N = H * W


def _run(c, d_xyz_ptr, n_pairs, d_res):
    lidar = capi.LidarParams(H, W, 1.0, 120.0)
    c.register_scan_pairs_dev(d_xyz_ptr, n_pairs, lidar, capi.FeatureExtractionParams(), capi.RegistrationParams(), d_res.ptr)
    c.synchronize()
    return d_res.download(np.uint8, n_pairs * 64).copy()


def test_unpacked_index_build_with_poisoned_scratch():
    """with the packed index build switched off, the ordered source builds scatter into their own
    scratch copy — which the workspace sized for the packed route only (one GridPoint). The workspace is sized with the
    launcher's predicate now; this runs the default 64 x 1024 batch through the unpacked builds with every scratch
    buffer poisoned and wants the packed route's bits back."""
    c = capi.Context(0)  # (its own workspace: sized by this route, not by whatever ran before)
    P = 24
    d_xyz, d_res = c.alloc(P * 2 * N * 24), c.alloc(P * 64)
    c.synth_scan_pairs_dev(SEED, 900, P, H, W, 0.01, d_xyz.ptr)
    with option("NO_PACKED_GRID", 1, c), option("DEBUG_POISON", 1, c):
        slow = _run(c, d_xyz.ptr, P, d_res)
    ref = _run(c, d_xyz.ptr, P, d_res)
    d_xyz.free()
    d_res.free()
    c.close()
    assert np.array_equal(slow, ref)
    assert (ref.view(capi.RESULT_DTYPE)["termination"] == capi.CONVERGED).all()


def test_rccl_gather_behind_the_c_abi_one_rank():
    c = ctx()
    comm = capi.Comm(c, capi.comm_unique_id(), 1, 0)
    info = comm.info()
    assert info["world_size"] == 1 and info["rank"] == 0
    rec = np.zeros(5, dtype=capi.RESULT_DTYPE)
    rec["pose"] = np.arange(35, dtype=np.float64).reshape(5, 7)
    rec["termination"], rec["iterations"] = np.arange(5), 7 - np.arange(5)
    d_local, d_all = c.alloc(5 * 64).upload(rec.view(np.uint8)), c.alloc(5 * 64)
    # default: a one-rank communicator takes the device-copy shortcut ...
    comm.gather_results_dev(d_local.ptr, 5, 5, d_all.ptr)
    assert comm.barrier(3.5) == 3.5
    assert np.array_equal(d_all.download(np.uint8, 5 * 64), rec.view(np.uint8))
    assert comm.stats() == dict(ncclAllGather=0, ncclBroadcast=0, ncclAllReduce=0, memcpy=2)
    # ... FORCE_RCCL: the same calls really enqueue ncclAllGather, the grouped ncclBroadcast (root 0) and
    # ncclAllReduce on the one-rank communicator, so the three collectives have executed before a multi-GPU node runs them
    d_all.upload(np.zeros(5 * 64, np.uint8))
    with option("FORCE_RCCL", 1, c):
        comm.gather_results_dev(d_local.ptr, 5, 5, d_all.ptr)
        assert comm.barrier(-2.25) == -2.25
    assert np.array_equal(d_all.download(np.uint8, 5 * 64), rec.view(np.uint8))
    assert comm.stats() == dict(ncclAllGather=1, ncclBroadcast=1, ncclAllReduce=1, memcpy=2)
    with pytest.raises(capi.LoamxError):  # not this rank's shard
        comm.gather_results_dev(d_local.ptr, 4, 5, d_all.ptr)
    comm.close()
    d_local.free()
    d_all.free()


def test_1024_pair_batch_against_the_oracle_sample(oracle):
    """configs[2]: the bench's own batch; 64 pairs spread over it are checked against the CPU oracle"""
    c = ctx()
    P = 1024
    d_xyz, d_res = c.alloc(P * 2 * N * 24), c.alloc(P * 64)
    c.synth_scan_pairs_dev(SEED, 0, P, H, W, 0.01, d_xyz.ptr)
    res = _run(c, d_xyz.ptr, P, d_res).view(capi.RESULT_DTYPE)
    d_xyz.free()
    d_res.free()
    assert (res["termination"] == capi.CONVERGED).all()
    worst = (0.0, 0.0)
    for pr in range(5, P, 16):  # 64 pairs
        A = capi.synth_scan_host(SEED, pr, 0, H, W, 0.01)
        B = capi.synth_scan_host(SEED, pr, 1, H, W, 0.01)
        ea, pa = oracle.extract_features(A, H, W, 1.0, 120.0)
        eb, pb = oracle.extract_features(B, H, W, 1.0, 120.0)
        po, to, io = oracle.register_features(B[eb], B[pb], A[ea], A[pa])
        assert (res[pr]["termination"], res[pr]["iterations"]) == (to, io), pr
        rot, trans = pose_diff(oracle, po, res[pr]["pose"])
        assert rot < 1e-5 and trans < 1e-5, (pr, rot, trans)
        worst = (max(worst[0], rot), max(worst[1], trans))
        rot, trans = pose_diff(oracle, capi.synth_pair_pose(SEED, pr), res[pr]["pose"])  # and the known SE(3)
        assert rot < 5e-3 and trans < 3e-2, (pr, rot, trans)
    print("1024-pair batch, 64-pair oracle sample: worst SE(3) difference", worst)


def test_eight_shard_plan_equals_one_8192_pair_call():
    """configs[3] on one GPU: rank r of 8 owns pairs [1024 r, 1024 (r + 1)) (loamx_shard_range); the eight shard
    results, concatenated in rank order as the gather does, are the bits of one 8 192-pair call."""
    c = ctx()
    world, P = 8, int(os.environ.get("LOAMX_TEST_SHARD_PAIRS", "1024"))
    total = world * P
    d_xyz, d_res = c.alloc(total * 2 * N * 24), c.alloc(total * 64)  # 25.8 GB of scans
    c.synth_scan_pairs_dev(SEED, 0, total, H, W, 0.01, d_xyz.ptr)
    whole = _run(c, d_xyz.ptr, total, d_res)
    parts = []
    for r in range(world):
        first, count = capi.shard_range(total, world, r)
        assert (first, count) == (r * P, P)
        parts.append(_run(c, d_xyz.ptr + first * 2 * N * 24, count, d_res))
    d_xyz.free()
    d_res.free()
    assert np.array_equal(np.concatenate(parts), whole)
    rec = whole.view(capi.RESULT_DTYPE)
    assert (rec["termination"] == capi.CONVERGED).all()
    # a rank that generates its own shard (first_pair = its offset), as bench.py does, gets the same scans
    d_xyz, d_res = c.alloc(P * 2 * N * 24), c.alloc(P * 64)
    c.synth_scan_pairs_dev(SEED, 5 * P, P, H, W, 0.01, d_xyz.ptr)
    assert np.array_equal(_run(c, d_xyz.ptr, P, d_res), parts[5])
    d_xyz.free()
    d_res.free()


def test_lm_loop_streaming_path_agrees_with_the_moment_path(oracle):
    """From the second ICF iteration on one wavefront per pair runs the whole LM solve (lm_pair_loop_kernel): off the
    plane moments, or — when a candidate leaves their validity bound — by streaming the pair's records itself. The bench
    workload never takes the second route; LOAMX_NO_MOMENTS=1 forces it for every evaluation. Both routes must tell the
    same story: terminations and iteration counts equal, poses within the summation-order noise of each other (1e-9),
    and the forced route within the 1e-5 bar of the oracle."""
    c = ctx()
    P = 96
    d_xyz, d_res = c.alloc(P * 2 * N * 24), c.alloc(P * 64)
    c.synth_scan_pairs_dev(SEED, 300, P, H, W, 0.01, d_xyz.ptr)
    ref = _run(c, d_xyz.ptr, P, d_res).view(capi.RESULT_DTYPE)
    with option("NO_MOMENTS"):
        forced = _run(c, d_xyz.ptr, P, d_res).view(capi.RESULT_DTYPE)
    # round 3's first ICF iteration — five sweeps of the records, moments only from the second iteration on — stays behind
    # NO_REF_MOMENTS (an option nobody compares is a liability): the same story again
    with option("NO_REF_MOMENTS"):
        five = _run(c, d_xyz.ptr, P, d_res).view(capi.RESULT_DTYPE)
    again = _run(c, d_xyz.ptr, P, d_res).view(capi.RESULT_DTYPE)
    d_xyz.free()
    d_res.free()
    assert np.array_equal(ref["termination"], five["termination"]) and np.array_equal(ref["iterations"], five["iterations"])
    assert not np.array_equal(ref["pose"], five["pose"])  # (the route really was taken)
    for pr in range(P):
        rot, trans = pose_diff(oracle, ref[pr]["pose"], five[pr]["pose"])
        assert rot < 1e-9 and trans < 1e-9, (pr, rot, trans)
    assert np.array_equal(ref.view(np.uint8), again.view(np.uint8))  # (the switch is read per call and leaves no state)
    assert np.array_equal(ref["termination"], forced["termination"]) and np.array_equal(ref["iterations"], forced["iterations"])
    assert not np.array_equal(ref["pose"], forced["pose"])  # a different summation order: the route really was taken
    for pr in range(P):
        rot, trans = pose_diff(oracle, ref[pr]["pose"], forced[pr]["pose"])
        assert rot < 1e-9 and trans < 1e-9, (pr, rot, trans)
    for pr in range(0, P, 12):
        A = capi.synth_scan_host(SEED, 300 + pr, 0, H, W, 0.01)
        B = capi.synth_scan_host(SEED, 300 + pr, 1, H, W, 0.01)
        ea, pa = oracle.extract_features(A, H, W, 1.0, 120.0)
        eb, pb = oracle.extract_features(B, H, W, 1.0, 120.0)
        po, to, io = oracle.register_features(B[eb], B[pb], A[ea], A[pa])
        assert (forced[pr]["termination"], forced[pr]["iterations"]) == (to, io), pr
        rot, trans = pose_diff(oracle, po, forced[pr]["pose"])
        assert rot < 1e-5 and trans < 1e-5, (pr, rot, trans)


def test_both_forms_of_the_queue_chain_give_the_same_bits():
    """The queued k-NN queries are searched in one kernel (round 3: small batches) or in two stages: the lean search of the
    5x5x5 block, then the listed leftovers — one WAVEFRONT per leftover since round 4 (associate_knn_coop_kernel), one lane
    per leftover with NO_COOP_LEFT. All are exact searches, so the whole registration must come out bit for bit the same
    whichever is forced — on a 96-pair batch, every ICF iteration of every pair."""
    c = ctx()
    P = 96
    d_xyz, d_res = c.alloc(P * 2 * N * 24), c.alloc(P * 64)
    c.synth_scan_pairs_dev(SEED, 700, P, H, W, 0.01, d_xyz.ptr)
    with option("QUEUE_ONE_STAGE"):
        one = _run(c, d_xyz.ptr, P, d_res)
    with option("QUEUE_TWO_STAGE"):
        two = _run(c, d_xyz.ptr, P, d_res)
    auto = _run(c, d_xyz.ptr, P, d_res)
    with option("NO_COOP_LEFT"), option("QUEUE_TWO_STAGE"):
        lanes = _run(c, d_xyz.ptr, P, d_res)
    assert np.array_equal(one, lanes)
    with option("NO_MIXED_ASSOC"):  # (edge and plane first kernels as separate launches on two streams instead of one launch each)
        separate = _run(c, d_xyz.ptr, P, d_res)
    d_xyz.free()
    d_res.free()
    assert np.array_equal(one, two) and np.array_equal(one, auto) and np.array_equal(one, separate)
    assert (one.view(capi.RESULT_DTYPE)["termination"] == capi.CONVERGED).all()


def test_index_builds_with_the_extraction_s_bounding_boxes_give_the_same_bits():
    """Round 5: in the scan-pair entry point the selection's copy phase hands the bounding boxes of every
    scan's feature sets to the index builds, which then skip their own read pass. Minima and maxima are exact, so the grids,
    and with them every result bit, are those of the builds that take the boxes themselves (NO_EXTRACT_BOXES) — also on a batch
    with a noise-free scan, whose tied lines send the extraction through the fallback compaction (boxes withdrawn)."""
    c = ctx()
    P = 24
    d_xyz, d_res = c.alloc(P * 2 * N * 24), c.alloc(P * 64)
    c.synth_scan_pairs_dev(SEED, 4100, P, H, W, 0.01, d_xyz.ptr)
    for tied in (False, True):
        if tied:  # scan 5 (the source of pair 2) without noise
            scans = d_xyz.download(np.float64, P * 2 * N * 3).reshape(P * 2, N, 3)
            scans[5] = capi.synth_scan_host(302, 0, 0, H, W, 0.0)
            d_xyz.upload(scans)
        r0 = c.extract_counters()[0]
        with option("NO_EXTRACT_BOXES"):
            own = _run(c, d_xyz.ptr, P, d_res)
        handed = _run(c, d_xyz.ptr, P, d_res)
        if tied:
            assert c.extract_counters()[0] > r0  # lines were replayed: the boxes were withdrawn on the device
        assert np.array_equal(own, handed), tied
    d_xyz.free()
    d_res.free()


def test_small_sets_build_agrees_with_the_big_build(oracle):
    """Round 5: pairs whose target edge set is brute-force sized (at most 512 points) get both edge sets from
    small_sets_build_kernel — the same grid, the points in their given order — instead of two 1 024-thread builds. The source
    edges then reach the sums in another order than the Morton order of grid_build_kernel (NO_SMALL_SETS): same terminations
    and iteration counts, poses within the summation-order noise."""
    c = ctx()
    P = 48
    d_xyz, d_res = c.alloc(P * 2 * N * 24), c.alloc(P * 64)
    c.synth_scan_pairs_dev(SEED, 5200, P, H, W, 0.01, d_xyz.ptr)
    small = _run(c, d_xyz.ptr, P, d_res).view(capi.RESULT_DTYPE)
    with option("NO_SMALL_SETS"):
        big = _run(c, d_xyz.ptr, P, d_res).view(capi.RESULT_DTYPE)
    d_xyz.free()
    d_res.free()
    assert np.array_equal(small["termination"], big["termination"]) and np.array_equal(small["iterations"], big["iterations"])
    assert not np.array_equal(small["pose"], big["pose"])  # (the other build really ran)
    for pr in range(P):
        rot, trans = pose_diff(oracle, small[pr]["pose"], big[pr]["pose"])
        assert rot < 1e-9 and trans < 1e-9, (pr, rot, trans)


def test_128_beam_batch_source_and_target_builds_do_not_share_scratch(oracle):
    """Feature sets above 20 480 points (128-beam scans: ~34 k planar features) take the index builds that need scratch
    memory — the multi-workgroup build of the target sets and the ordered single-workgroup build + rank of the source
    sets — and in a batch the two run side by side on two streams. They used to share one scratch buffer (a GPU memory
    fault on 64 pairs of 128 x 2048, found by tools/bench_other_sensors.py); now each has its own. Two runs must agree
    bit for bit, and with the oracle."""
    c = ctx()
    Hb, Wb, P = 128, 1024, 16
    Nb = Hb * Wb
    lidar = capi.LidarParams(Hb, Wb, 1.0, 120.0)
    d_xyz, d_res = c.alloc(P * 2 * Nb * 24), c.alloc(P * 64)
    c.synth_scan_pairs_dev(SEED, 0, P, Hb, Wb, 0.01, d_xyz.ptr)
    runs = []
    for _ in range(3):
        c.register_scan_pairs_dev(d_xyz.ptr, P, lidar, capi.FeatureExtractionParams(), capi.RegistrationParams(), d_res.ptr)
        c.synchronize()
        runs.append(d_res.download(np.uint8, P * 64).copy())
    d_xyz.free()
    d_res.free()
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])
    res = runs[0].view(capi.RESULT_DTYPE)
    for pr in (0, 9):
        A = capi.synth_scan_host(SEED, pr, 0, Hb, Wb, 0.01)
        B = capi.synth_scan_host(SEED, pr, 1, Hb, Wb, 0.01)
        ea, pa = oracle.extract_features(A, Hb, Wb, 1.0, 120.0)
        eb, pb = oracle.extract_features(B, Hb, Wb, 1.0, 120.0)
        assert len(pa) > 20480 and len(pb) > 20480
        po, to, io = oracle.register_features(B[eb], B[pb], A[ea], A[pa])
        assert (res[pr]["termination"], res[pr]["iterations"]) == (to, io), pr
        rot, trans = pose_diff(oracle, po, res[pr]["pose"])
        assert rot < 1e-5 and trans < 1e-5, (pr, rot, trans)


def test_config5_scan_against_a_million_point_map(oracle):
    """BASELINE configs[5] at its real size: a 128 x 2048 scan registered against a local map of > 1 M planar points
    (the planar / edge features of ~27 scans of the scene in one frame), through the plain call and through the persistent
    target index — bit-identical to each other, and within the 1e-5 bar of the CPU oracle (one oracle registration,
    ~0.3 s). The feature sets of the map are extracted on the GPU (their index sequences are pinned elsewhere)."""
    c = ctx()
    H5, W5 = 128, 2048
    lidar = capi.LidarParams(H5, W5, 1.0, 120.0)
    fe = capi.FeatureExtractionParams()
    src = capi.synth_scan_host(99, 0, 1, H5, W5, 0.01)
    e5, p5 = c.extract_features(src, lidar, fe)
    oe5, op5 = oracle.extract_features(src, H5, W5, 1.0, 120.0)
    assert np.array_equal(e5, oe5) and np.array_equal(p5, op5)
    maps_p, maps_e, k = [], [], 0
    while sum(len(m) for m in maps_p) < 1_000_000:
        s = capi.synth_scan_host(1000 + k, 0, 0, H5, W5, 0.01)
        e, p = c.extract_features(s, lidar, fe)
        maps_p.append(s[p]), maps_e.append(s[e])
        k += 1
    map_p, map_e = np.ascontiguousarray(np.concatenate(maps_p)), np.ascontiguousarray(np.concatenate(maps_e))
    assert len(map_p) > 1_000_000
    pose, term, iters = c.register_features(src[e5], src[p5], map_e, map_p)
    idx = c.target_index(map_e, map_p)
    pose_i, term_i, iters_i = c.register_features_indexed(idx, src[e5], src[p5])
    assert (term_i, iters_i) == (term, iters) and np.array_equal(np.asarray(pose_i), np.asarray(pose))
    # Round 6: more source scans against the same map. The two entry points size their slot arrays differently (the plain call by the
    # map, the index by the scan), so the pair's listed plane records are walked tile by tile in one and as a flat list in the
    # other; until round 6 the two walks dealt the records to the lanes differently and seeds 6 and 7 differed in the last bit.
    for seed in (5, 6, 7):
        s2 = capi.synth_scan_host(seed, 0, 1, H5, W5, 0.01)
        e2, p2 = c.extract_features(s2, lidar, fe)
        plain = c.register_features(s2[e2], s2[p2], map_e, map_p)
        through = c.register_features_indexed(idx, s2[e2], s2[p2])
        assert plain[1:] == through[1:] and np.array_equal(np.asarray(plain[0]), np.asarray(through[0])), seed
    c.target_index_destroy(idx)
    po, to, io = oracle.register_features(src[oe5], src[op5], map_e, map_p)
    assert (term, iters) == (to, io)
    rot, trans = pose_diff(oracle, po, np.asarray(pose))
    assert rot < 1e-5 and trans < 1e-5, (rot, trans)
    print("config 5: %d map points, %d ICF iterations, SE(3) difference %.1e rad %.1e m" % (len(map_p), iters, rot, trans))


def test_scan_pairs_from_host_memory_equal_the_device_resident_call():
    """Round 6: loamx_register_scan_pairs — host scans in, host results out, chunks uploaded under the
    registration of the chunk before. Same bits as loamx_register_scan_pairs_dev whatever the chunking (uneven last chunk, one
    chunk, a chunk per pair), for double and float scans; non-finite input is refused; an empty batch is a no-op."""
    c = ctx()
    P = 20
    lidar = capi.LidarParams(H, W, 1.0, 120.0)
    d_xyz, d_res = c.alloc(P * 2 * N * 24), c.alloc(P * 64)
    c.synth_scan_pairs_dev(SEED, 300, P, H, W, 0.01, d_xyz.ptr)
    want = _run(c, d_xyz.ptr, P, d_res).view(capi.RESULT_DTYPE)
    host = d_xyz.download(np.float64, P * 2 * N * 3).copy()
    for chunk in (7, 0, 1):
        with option("STREAM_CHUNK_PAIRS", chunk):
            got = c.register_scan_pairs(host, P, lidar)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), chunk
    # float scans: the same as the device-resident float call on the same rounded scans
    host32 = host.astype(np.float32)
    d_x32 = c.alloc(host32.nbytes).upload(host32)
    c.register_scan_pairs_dev(d_x32.ptr, P, lidar, capi.FeatureExtractionParams(), capi.RegistrationParams(), d_res.ptr, f32=True)
    c.synchronize()
    want32 = d_res.download(np.uint8, P * 64).copy()
    with option("STREAM_CHUNK_PAIRS", 6):
        got32 = c.register_scan_pairs(host32, P, lidar)
    assert np.array_equal(got32.view(np.uint8), want32)
    assert len(c.register_scan_pairs(host, 0, lidar)) == 0
    dirty = host.copy()
    dirty[(2 * 13 + 1) * N * 3 + 5] = np.nan  # the source scan of pair 13: third chunk of seven
    with option("STREAM_CHUNK_PAIRS", 5), pytest.raises(capi.LoamxError) as e:
        c.register_scan_pairs(dirty, P, lidar)
    assert e.value.status == capi.ERR_BAD_PARAM and "non-finite" in str(e.value)
    again = c.register_scan_pairs(host, P, lidar)  # the context is as good as before
    assert np.array_equal(again.view(np.uint8), want.view(np.uint8))
    for b_ in (d_xyz, d_res, d_x32):
        b_.free()
