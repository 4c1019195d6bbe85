"""GPU: truly incremental loamx_target_index_insert. A map-sized feature kind takes new
points by a merge into its existing grid (cost ~ the new points + one streaming copy), and every search / registration
afterwards returns what an index built over the concatenated sets returns, bit for bit. Rebuilds happen only when a new
point leaves the grid, when the kind has doubled since its grid was chosen, or for scan-sized kinds."""
import time

import numpy as np
import pytest

from gpu_common import ctx
from loam_amd import capi

pytestmark = pytest.mark.gpu


def test_merge_inserts_equal_a_fresh_index_and_the_oracle_tree(oracle):
    rng = np.random.default_rng(17)
    c = ctx()
    box_lo, box_hi = np.array([-10.0, -8.0, -2.0]), np.array([10.0, 8.0, 4.0])

    def surface_points(n):  # walls and floor of a room, jittered: cells of very different populations
        u = rng.uniform(0, 1, (n, 3)) * (box_hi - box_lo) + box_lo
        face = rng.integers(0, 3, n)
        u[np.arange(n), face] = np.where(rng.random(n) < 0.5, box_lo[face], box_hi[face]) + rng.normal(size=n) * 0.01
        return np.clip(u, box_lo - 0.05, box_hi + 0.05)

    base_p, base_e = surface_points(260_000), surface_points(900)
    base_p[:2] = [box_lo - 0.05, box_hi + 0.05]  # the grid's corners: later points stay inside
    idx = c.target_index(base_e, base_p)
    assert c.target_index_stats(idx) == (2, 0)  # (full builds are counted per feature kind)
    all_p, all_e = [base_p], [base_e]
    for step in range(6):
        add_p, add_e = surface_points(30_000), surface_points(100)
        c.target_index_insert(idx, add_e, add_p)
        all_p.append(add_p), all_e.append(add_e)
    builds, merges = c.target_index_stats(idx)
    # planar: the first insert outgrows the exactly-sized buffers (capacity doubles: a rebuild), then five merges
    # (290 k -> 440 k stays below twice the size at that build); edge (scan-sized): rebuilt every time
    assert merges == 5 and builds == 2 + 6 + 1, (builds, merges)
    cat_p, cat_e = np.concatenate(all_p), np.concatenate(all_e)
    assert c.target_index_size(idx) == (len(cat_e), len(cat_p))
    fresh = c.target_index(cat_e, cat_p)
    tree = oracle.KDTree(cat_p)
    q = np.concatenate([cat_p[rng.integers(0, len(cat_p), 300)] + rng.normal(size=(300, 3)) * 0.02, surface_points(100)])
    for k, radius in ((5, 2.0), (8, -1.0), (1, 0.05)):
        got, ref = c.knn_search(idx, 1, q, k, radius), c.knn_search(fresh, 1, q, k, radius)
        for i in range(len(q)):
            want = tree.knn(q[i], k, radius).astype(np.uint32)
            assert np.array_equal(got[i], want) and np.array_equal(ref[i], want), (k, radius, i)
    # registration of a sub-sampled, moved copy: grown index == fresh index, bit for bit
    ang = 0.01
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    src_p = cat_p[rng.choice(len(cat_p), 20_000, replace=False)] @ R.T + np.array([0.05, -0.03, 0.02])
    src_e = cat_e[rng.choice(len(cat_e), 500, replace=False)] @ R.T + np.array([0.05, -0.03, 0.02])
    a = c.register_features_indexed(idx, src_e, src_p)
    b = c.register_features_indexed(fresh, src_e, src_p)
    assert a[1:] == b[1:] and np.array_equal(a[0].view(np.uint64), b[0].view(np.uint64))
    # a point outside the grid: that kind is rebuilt around the larger set, results stay right
    out_pt = np.array([[box_hi[0] + 3.0, 0.0, 0.0]])
    c.target_index_insert(idx, np.zeros((0, 3)), out_pt)
    b2, m2 = c.target_index_stats(idx)
    assert (b2, m2) == (builds + 1, merges)
    cat_p2 = np.concatenate([cat_p, out_pt])
    tree2 = oracle.KDTree(cat_p2)
    got = c.knn_search(idx, 1, np.concatenate([out_pt + 0.01, q[:50]]), 5, -1.0)
    for i, qq in enumerate(np.concatenate([out_pt + 0.01, q[:50]])):
        assert np.array_equal(got[i], tree2.knn(qq, 5, -1.0).astype(np.uint32))
    # ... and merging goes on afterwards
    last_p = surface_points(10_000)
    c.target_index_insert(idx, np.zeros((0, 3)), last_p)
    assert c.target_index_stats(idx) == (b2, m2 + 1)
    # an edge-only insert rebuilds the (scan-sized) edge kind and leaves the merged planar map alone (the
    # planar kind's size at its last FULL build differs from its size now, which used to force a full re-sort)
    more_e = surface_points(50)
    c.target_index_insert(idx, more_e, np.zeros((0, 3)))
    assert c.target_index_stats(idx) == (b2 + 1, m2 + 1)
    got = c.knn_search(idx, 1, q[:50], 5, 2.0)  # (the planar kind still answers: the last merge's points included)
    tree3 = oracle.KDTree(np.concatenate([cat_p2, last_p]))
    for i in range(50):
        assert np.array_equal(got[i], tree3.knn(q[i], 5, 2.0).astype(np.uint32))
    c.target_index_insert(idx, np.zeros((0, 3)), np.zeros((0, 3)))  # nothing: nothing happens
    assert c.target_index_stats(idx) == (b2 + 1, m2 + 1)
    c.target_index_destroy(idx)
    c.target_index_destroy(fresh)


def test_insert_cost_is_the_new_points_not_the_map():
    """one 128 x 2048 scan's worth of features (39 k + 8 k) into a 1 M-point index: well under a rebuild (VERDICT: <= 0.3 ms
    of device work; the wall time here includes the 1.1 MB upload and two synchronisations)"""
    rng = np.random.default_rng(5)
    c = ctx()
    lo, hi = np.array([-10.0, -8.0, -2.0]), np.array([10.0, 8.0, 4.0])

    def pts(n):
        u = rng.uniform(0, 1, (n, 3)) * (hi - lo) + lo
        u[:, 2] = np.where(rng.random(n) < 0.7, -2.0, u[:, 2]) + rng.normal(size=n) * 0.005
        return np.clip(u, lo, hi)

    base = pts(1_000_000)
    base[:2] = [lo, hi]
    t0 = time.perf_counter()
    idx = c.target_index(pts(300), base)
    t_create = time.perf_counter() - t0
    add_p, add_e = pts(39_000), np.zeros((0, 3))
    c.target_index_insert(idx, add_e, pts(39_000))  # (outgrows the exactly-sized buffers: capacity doubles, one rebuild)
    c.target_index_insert(idx, add_e, pts(39_000))  # (first merge: allocates the twin buffers)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        c.target_index_insert(idx, add_e, add_p)
        best = min(best, time.perf_counter() - t0)
    builds, merges = c.target_index_stats(idx)
    assert merges == 6 and builds == 3  # (create: both kinds; the first insert: the planar kind)
    print(f"index create {t_create * 1e3:.2f} ms; insert of 39 k points into a 1.08-1.27 M-point kind: {best * 1e3:.3f} ms wall")
    assert best < 0.5 * t_create and best < 1.5e-3
    c.target_index_destroy(idx)
