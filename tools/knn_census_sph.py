#!/usr/bin/env python3
"""Census (CPU, test libraries only) of a sensor-centred index for the plane k-NN.

    python tools/knn_census_sph.py [pair ...]

Index: the target's planar features binned by two trig-free keys of the point itself — t = z / sqrt(x^2 + y^2) (NE bins over
the set's own range) and the diamond angle of (x, y) (NA bins over [0, 4)) — sorted row (t bin) by row, azimuth bin inside a row.
Walk, with the kernel's rules: rows in the order 0, -1, +1, -2, ... up to +-WR, each the azimuth bins ba - WA .. ba + WA as ONE
range; a row is taken while its rigorous distance (to the cone that bounds it: rho_xy |dt| / sqrt(1 + t_edge^2)) is not above the
running k-th distance or the radius; candidates in batches of four; the search is over when the k-th distance is below the
distance to the window's faces (the two vertical half-planes of the outer azimuth edges, the two outer cones), otherwise the
query is queued. A wavefront = 64 consecutive queries in the SOURCE set's own (row, azimuth bin) order, and pays its slowest lane.
Iterations >= 2 ("prev"): the a-priori bound w0 = d5_prev + |T_new q - T_old q| picks the window per query.
"""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hostcheck_lib as Hc  # noqa: E402
import oracle_lib  # noqa: E402
import knn_census as cart  # noqa: E402

K = 5
R = 2.0


def diamond(x, y):
    """monotone in the azimuth, [0, 4)"""
    s = np.abs(x) + np.abs(y)
    s = np.where(s == 0, 1.0, s)
    d = y / s
    return np.where(x >= 0, np.where(y >= 0, d, 4.0 + d), 2.0 - d)


def diamond_dir(a):
    """unit direction of diamond angle a"""
    a = np.mod(a, 4.0)
    q = np.floor(a)
    f = a - q
    x = np.select([q == 0, q == 1, q == 2, q == 3], [1 - f, -f, f - 1, f])
    y = np.select([q == 0, q == 1, q == 2, q == 3], [f, 1 - f, -f, f - 1])
    n = np.hypot(x, y)
    return x / n, y / n


class SphIndex:
    def __init__(self, pts, NE, NA, t_range=None):
        self.NE, self.NA = NE, NA
        rxy = np.hypot(pts[:, 0], pts[:, 1])
        t = pts[:, 2] / rxy
        lo, hi = (t.min(), t.max()) if t_range is None else t_range
        self.t0, self.dt = lo, (hi - lo) * (1 + 1e-9) / NE
        self.eb = np.clip(np.floor((t - self.t0) / self.dt).astype(int), 0, NE - 1)
        self.ab = np.clip(np.floor(diamond(pts[:, 0], pts[:, 1]) * (NA / 4.0)).astype(int), 0, NA - 1)
        cell = self.eb * NA + self.ab
        self.order = np.argsort(cell, kind="stable")
        self.pts = pts[self.order]
        self.start = np.searchsorted(cell[self.order], np.arange(NE * NA + 1))

    def row_range(self, e, a0, a1):
        """points of row e, azimuth bins a0..a1 inclusive (cyclic): list of (begin, end)"""
        NA = self.NA
        if e < 0 or e >= self.NE:
            return []
        if a1 - a0 + 1 >= NA:
            return [(self.start[e * NA], self.start[(e + 1) * NA])]
        a0m, a1m = a0 % NA, a1 % NA
        if a0m <= a1m:
            return [(self.start[e * NA + a0m], self.start[e * NA + a1m + 1])]
        return [(self.start[e * NA + a0m], self.start[(e + 1) * NA]), (self.start[e * NA], self.start[e * NA + a1m + 1])]


def census_query(ix, q, WR, WA, w0=None, adaptive=False, max_rows=None):
    """(candidates, batches, row steps, done)"""
    rxy = np.hypot(q[0], q[1])
    tq = q[2] / rxy
    aq = float(diamond(np.array(q[0]), np.array(q[1])))
    be = int(np.floor((tq - ix.t0) / ix.dt))
    ba = int(np.floor(aq * ix.NA / 4.0))
    binw = 4.0 / ix.NA

    def cone_dist(t_edge):
        if 1 + t_edge * tq <= 0:
            return 0.0
        return rxy * abs(t_edge - tq) / np.sqrt(1 + t_edge * t_edge)

    def az_dist(a_edge):
        cx, cy = diamond_dir(np.array(a_edge))
        d = abs(q[0] * cy - q[1] * cx)
        # gap above 90 degrees: the distance is rho_xy (never with these windows)
        return float(d)

    if adaptive and w0 is not None:
        # smallest window whose faces are farther than w0 (the bound holds the five previous neighbours: one pass suffices)
        WA = 0
        while WA < ix.NA // 4 and min(az_dist((ba - WA) * binw), az_dist((ba + WA + 1) * binw)) <= w0:
            WA += 1
        WR = 0
        while WR < ix.NE:
            lo_ok = be - WR <= 0 or cone_dist(ix.t0 + (be - WR) * ix.dt) > w0
            hi_ok = be + WR >= ix.NE - 1 or cone_dist(ix.t0 + (be + WR + 1) * ix.dt) > w0
            if lo_ok and hi_ok:
                break
            WR += 1
        if max_rows is not None and 2 * WR + 1 > max_rows:
            return 0, 0, 0, False
    rows = [0]
    for j in range(1, WR + 1):
        rows += [-j, j]
    best = np.empty(0)
    cand = batches = steps = 0
    bound0 = R * R if w0 is None else min(R * R, w0 * w0)
    for j in rows:
        e = be + j
        if e < 0 or e >= ix.NE:
            continue
        s = 0.0 if j == 0 else (cone_dist(ix.t0 + e * ix.dt) if j > 0 else cone_dist(ix.t0 + (e + 1) * ix.dt))
        kth = best[K - 1] if len(best) >= K else np.inf
        rr = ix.row_range(e, ba - WA, ba + WA)
        n = sum(e1 - b1 for b1, e1 in rr)
        if n == 0:
            continue  # (empty rows never reach the list)
        steps += 1
        if s * s > min(kth, bound0):
            continue
        for b1, e1 in rr:
            if e1 > b1:
                d = ((ix.pts[b1:e1] - q) ** 2).sum(1)
                best = np.sort(np.concatenate([best, d]))[:K]
                cand += e1 - b1
                batches += (e1 - b1 + 3) // 4
    kth = best[K - 1] if len(best) >= K else np.inf
    guard = min(az_dist((ba - WA) * binw), az_dist((ba + WA + 1) * binw))
    if be - WR > 0:
        guard = min(guard, cone_dist(ix.t0 + (be - WR) * ix.dt))
    if be + WR < ix.NE - 1:
        guard = min(guard, cone_dist(ix.t0 + (be + WR + 1) * ix.dt))
    done = kth < guard * guard or guard >= R
    return cand, batches, steps, done


def wave_stats(name, order, out):
    cand, batches, steps, done = (np.asarray(v)[order] for v in zip(*out))
    trips = np.maximum(batches, 1) + np.maximum(steps - batches, 0)
    nw = len(trips) // 64
    wt = trips[: nw * 64].reshape(nw, 64)
    wc = cand[: nw * 64].reshape(nw, 64)
    print("  %-28s cand %5.1f  batches %5.2f  trips/lane %5.2f  wave-max trips %5.2f (slots %5.1f)  wave-max cand %5.1f  util %.2f  queued %.3f" % (
        name, cand.mean(), batches.mean(), trips.mean(), wt.max(1).mean(), 4 * wt.max(1).mean(), wc.max(1).mean(),
        wt.mean() / wt.max(1).mean(), 1 - done.mean()))


def main():
    pairs = [int(a) for a in sys.argv[1:]] or [0]
    H, W = 64, 1024
    for pair in pairs:
        A = Hc.synth_scan(20240311, pair, 0, H, W, 0.01)
        B = Hc.synth_scan(20240311, pair, 1, H, W, 0.01)
        ea, pa = oracle_lib.extract_features(A, H, W, 1.0, 120.0)
        eb, pb = oracle_lib.extract_features(B, H, W, 1.0, 120.0)
        pose, term, iters, info = Hc.register(B[eb], B[pb], A[ea], A[pa], want_info=True)
        tgt, src = np.ascontiguousarray(A[pa]), np.ascontiguousarray(B[pb])
        tree = cKDTree(tgt)
        rows_c = cart.Rows(tgt, R / 4)
        morton = cart.morton_order(src)
        print(f"pair {pair}: {len(tgt)} targets, {len(src)} queries, {iters} ICF iterations")
        grids = {}
        for NE, NA in ((64, 128), (64, 256), (96, 128), (128, 256)):
            grids[(NE, NA)] = (SphIndex(tgt, NE, NA), SphIndex(src, NE, NA).order)
        prev_q = prev_d5 = None
        for it in range(iters):
            est = np.array(list(info[it].target_T_source_init))
            q = np.array([cart.quat_rot(est, s) for s in src])
            d5 = tree.query(q, k=K)[0][:, K - 1]
            print(" iteration %d: true d5 mean %.3f median %.3f p90 %.3f p99 %.3f" % (
                it + 1, d5.mean(), np.median(d5), np.percentile(d5, 90), np.percentile(d5, 99)))
            out = [cart.census_query(rows_c, q[i], scheme="cur")[:4] for i in range(len(q))]
            wave_stats("cartesian (today)", morton, out)
            for (NE, NA), (ix, order) in grids.items():
                for WR, WA in ((2, 1), (3, 1), (4, 1), (4, 2), (6, 2)):
                    if NA == 256:
                        WA *= 2
                    if NE >= 96:
                        WR = WR * NE // 64
                    out = [census_query(ix, q[i], WR, WA) for i in range(len(q))]
                    wave_stats(f"sph {NE}x{NA} +-{WR} x +-{WA}", order, out)
                if prev_q is not None:
                    w0 = (prev_d5 + np.linalg.norm(q - prev_q, axis=1)) * (1 + 1e-9)
                    for mr in (9, 13):
                        out = [census_query(ix, q[i], 0, 0, w0=w0[i], adaptive=True, max_rows=mr) for i in range(len(q))]
                        wave_stats(f"sph {NE}x{NA} prev (<= {mr} rows)", order, out)
                out = [census_query(ix, q[i], 0, 0, w0=d5[i] * (1 + 1e-9), adaptive=True, max_rows=13) for i in range(len(q))]
                wave_stats(f"sph {NE}x{NA} ideal window", order, out)
            prev_q, prev_d5 = q, d5


if __name__ == "__main__":
    main()
