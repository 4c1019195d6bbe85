#!/usr/bin/env python3
"""Census (CPU, test libraries only): candidates / trips per plane k-NN query under several x-window schemes.

    python tools/knn_census.py [pair ...]

For every ICF iteration of a bench pair (poses from the hostcheck registration) it counts, per query, the candidates a
search would examine and the batches of four a lane would issue, then what a 64-lane wavefront pays (max over lanes in
the source set's Morton order). Schemes:

  cur      today's kernel: nine rows of the 3x3x3 block of R/4 cells, three cells each, rows pruned by the running bound
  two(a,f) two-phase: centre row first over fine cells [fx-a, fx+a] (fine cell = h/f), its k-th distance w bounds every
           other piece: row windows qx +- sqrt(w^2 - slab^2) rounded out to fine cells; no bound -> today's ranges
  prev     (iterations >= 2) a-priori bound w = d5_prev + |T_new q - T_old q|, every row windowed (centre too)
"""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import hostcheck_lib as Hc  # noqa: E402
import oracle_lib  # noqa: E402

ORDER = (4, 1, 3, 5, 7, 0, 2, 6, 8)
# (name, cell edge = R / div, census_query arguments, a-priori bound: None | "prev" | "ideal")
VARIANTS = [
    ("prev f=1", 4, dict(scheme="cur", f=1), "prev"),
    ("prev f=2", 4, dict(scheme="cur", f=2), "prev"),
    ("prev f=4", 4, dict(scheme="cur", f=4), "prev"),
    ("ideal f=1", 4, dict(scheme="cur", f=1), "ideal"),
    ("ideal f=2", 4, dict(scheme="cur", f=2), "ideal"),
    ("dyn(1,2)", 4, dict(scheme="dyn", a=1, f=2), None),
]
K = 5
R = 2.0


def quat_rot(P, v):
    x, y, z, w = P[0], P[1], P[2], P[3]
    q = np.array([x, y, z])
    t = 2.0 * np.cross(q, v)
    return v + w * t + np.cross(q, t) + P[4:7]


def morton_order(src):
    lo, hi = src.min(0), src.max(0)
    hcell = (hi - lo).max() / 32 * (1 + 1e-9)
    c = np.clip(((src - lo) / hcell).astype(np.int64), 0, 31)

    def spread(v):
        v = (v | (v << 8)) & 0x100F
        v = (v | (v << 4)) & 0x10C3
        v = (v | (v << 2)) & 0x1249
        return v
    code = spread(c[:, 0]) | (spread(c[:, 1]) << 1) | (spread(c[:, 2]) << 2)
    return np.argsort(code, kind="stable")


class Rows:
    """target points row by row (y, z cells of edge h), x-sorted inside a row"""

    def __init__(self, tgt, h):
        self.h = h
        self.o = tgt.min(0)
        self.n = np.floor((tgt.max(0) - self.o) / h).astype(int) + 1
        c = np.clip(np.floor((tgt - self.o) / h).astype(int), 0, self.n - 1)
        row = c[:, 2] * self.n[1] + c[:, 1]
        idx = np.lexsort((tgt[:, 0], row))
        self.pts = tgt[idx]
        self.row_start = np.searchsorted(row[idx], np.arange(self.n[1] * self.n[2] + 1))

    def piece(self, iy, iz, x0, x1):
        """points of row (iy, iz) with x0 <= x < x1 as a slice of self.pts"""
        if iy < 0 or iz < 0 or iy >= self.n[1] or iz >= self.n[2] or x1 <= x0:
            return 0, 0
        r = iz * self.n[1] + iy
        b, e = self.row_start[r], self.row_start[r + 1]
        xs = self.pts[b:e, 0]
        return b + np.searchsorted(xs, x0, "left"), b + np.searchsorted(xs, x1, "left")


def kth(best):
    return best[K - 1] if len(best) >= K else np.inf


def merge(best, d):
    return np.sort(np.concatenate([best, d]))[:K]


def census_query(rows, q, scheme, a=2, f=4, w0=None, m=1):
    """returns (candidates, batches, row steps, done) of one query"""
    h, o = rows.h, rows.o
    c = np.floor((q - o) / h).astype(int)
    hx = h / f
    D = 2 * m + 1
    xlo_blk, xhi_blk = o[0] + (c[0] - m) * h, o[0] + (c[0] + m + 1) * h
    fx = int(np.floor((q[0] - o[0]) / hx))
    slab = {}
    for j in range(D * D):
        dy, dz = j % D - m, j // D - m
        sy = 0.0 if dy == 0 else (q[1] - (o[1] + (c[1] + dy + 1) * h) if dy < 0 else o[1] + (c[1] + dy) * h - q[1])
        sz = 0.0 if dz == 0 else (q[2] - (o[2] + (c[2] + dz + 1) * h) if dz < 0 else o[2] + (c[2] + dz) * h - q[2])
        slab[j] = sy * sy + sz * sz
    ctr = m * D + m
    order = sorted(range(D * D), key=lambda j: ((j % D - m) ** 2 + (j // D - m) ** 2, j))
    best = np.empty(0)
    cand = batches = steps = 0
    pieces = []  # (j, x0, x1)
    w2 = np.inf
    if scheme == "two":
        x0, x1 = max(o[0] + (fx - a) * hx, xlo_blk), min(o[0] + (fx + a + 1) * hx, xhi_blk)
        b, e = rows.piece(c[1], c[2], x0, x1)
        if e > b:
            d = ((rows.pts[b:e] - q) ** 2).sum(1)
            best = merge(best, d)
            cand += e - b
            batches += (e - b + 3) // 4
        steps += 1
        w2 = kth(best)
        done_x = (x0, x1)
    else:
        done_x = None
    if w0 is not None:
        w2 = min(w2, w0 * w0)
    w2 = min(w2, R * R)
    for j in order:
        if slab[j] > w2:
            continue
        half = np.sqrt(w2 - slab[j]) if np.isfinite(w2) else np.inf
        if np.isfinite(half):
            x0 = max(o[0] + np.floor((q[0] - half - o[0]) / hx) * hx, xlo_blk)
            x1 = min(o[0] + (np.floor((q[0] + half - o[0]) / hx) + 1) * hx, xhi_blk)
        else:
            x0, x1 = xlo_blk, xhi_blk
        if j == ctr and done_x is not None:
            pieces.append((j, x0, min(x1, done_x[0])))
            pieces.append((j, max(x0, done_x[1]), x1))
        else:
            pieces.append((j, x0, x1))
    if scheme == "dyn":
        # centre window first, then every row (the centre's remainders as two pieces) with the window the running bound allows
        x0c, x1c = max(o[0] + (fx - a) * hx, xlo_blk), min(o[0] + (fx + a + 1) * hx, xhi_blk)
        pieces = [(ctr, x0c, x1c)]
        for j in order:
            if j == ctr:
                pieces.append((j, xlo_blk, x0c))
                pieces.append((j, x1c, xhi_blk))
            else:
                pieces.append((j, xlo_blk, xhi_blk))
    for j, x0, x1 in pieces:
        if x1 <= x0:
            continue
        if slab[j] > min(kth(best), w2):  # running bound, as the kernel's row threshold
            steps += 1
            continue
        if scheme == "dyn":
            b2 = min(kth(best), w2)
            if np.isfinite(b2):
                half = np.sqrt(b2 - slab[j])
                x0 = max(x0, o[0] + np.floor((q[0] - half - o[0]) / hx) * hx)
                x1 = min(x1, o[0] + (np.floor((q[0] + half - o[0]) / hx) + 1) * hx)
                if x1 <= x0:
                    steps += 1
                    continue
        b, e = rows.piece(c[1] + j % D - m, c[2] + j // D - m, x0, x1)
        if e > b:
            d = ((rows.pts[b:e] - q) ** 2).sum(1)
            best = merge(best, d)
            cand += e - b
            batches += (e - b + 3) // 4
            steps += 1
    # is the search over? k-th distance below the distance to the block's faces (ignoring the grid's outer faces here)
    guard = min(q[0] - xlo_blk, xhi_blk - q[0], q[1] - (o[1] + (c[1] - m) * h), o[1] + (c[1] + m + 1) * h - q[1],
                q[2] - (o[2] + (c[2] - m) * h), o[2] + (c[2] + m + 1) * h - q[2])
    done = kth(best) < guard * guard
    return cand, batches, steps, done, np.sqrt(kth(best))


def wave_stats(name, order, cand, batches, steps, done):
    cand, batches, steps, done = (np.asarray(v)[order] for v in (cand, batches, steps, done))
    trips = np.maximum(batches, 1) + np.maximum(steps - batches, 0)  # a row step rides on a batch trip unless it rejects / is empty
    nw = len(trips) // 64
    wt = trips[: nw * 64].reshape(nw, 64)
    print("  %-24s cand %5.1f  batches %5.2f  trips/lane %5.2f  wave max %5.2f  util %.2f  queued %.3f" % (
        name, cand.mean(), batches.mean(), trips.mean(), wt.max(1).mean(), wt.mean() / wt.max(1).mean(), 1 - done.mean()))


def main():
    pairs = [int(a) for a in sys.argv[1:]] or [0]
    orc = oracle_lib
    H, W = 64, 1024
    for pair in pairs:
        A = Hc.synth_scan(20240311, pair, 0, H, W, 0.01)
        B = Hc.synth_scan(20240311, pair, 1, H, W, 0.01)
        ea, pa = orc.extract_features(A, H, W, 1.0, 120.0)
        eb, pb = orc.extract_features(B, H, W, 1.0, 120.0)
        pose, term, iters, info = Hc.register(B[eb], B[pb], A[ea], A[pa], want_info=True)
        tgt, src = np.ascontiguousarray(A[pa]), np.ascontiguousarray(B[pb])
        rows = Rows(tgt, R / 4)
        rows_by_div = {d: Rows(tgt, R / d) for d in (3, 3.5, 4, 6, 8, 12)}
        tree = cKDTree(tgt)
        order = morton_order(src)
        print(f"pair {pair}: {len(tgt)} targets, {len(src)} queries, {iters} ICF iterations, grid {rows.n} h={rows.h}")
        prev_q = prev_d5 = None
        for it in range(iters):
            est = np.array(list(info[it].target_T_source_init))
            q = np.array([quat_rot(est, s) for s in src])
            d5 = tree.query(q, k=K)[0][:, K - 1]
            print(" iteration %d: true d5 mean %.3f median %.3f p90 %.3f p99 %.3f" % (
                it + 1, d5.mean(), np.median(d5), np.percentile(d5, 90), np.percentile(d5, 99)))
            variants = [("cur", rows, dict(scheme="cur"), None)]
            for spec in VARIANTS:
                name, div, kw, apriori = spec
                if apriori == "prev" and prev_q is None:
                    continue
                variants.append((name, rows_by_div[div], kw, apriori))
            for name, rr, kw, apriori in variants:
                out = []
                for i in range(len(q)):
                    w0 = None
                    if apriori == "prev":
                        w0 = (prev_d5[i] + np.linalg.norm(q[i] - prev_q[i])) * (1 + 1e-9)
                    elif apriori == "ideal":
                        w0 = d5[i] * (1 + 1e-9)
                    out.append(census_query(rr, q[i], w0=w0, **kw)[:4])
                cand, batches, steps, done = zip(*out)
                wave_stats(name, order, cand, batches, steps, done)
            prev_q, prev_d5 = q, d5


if __name__ == "__main__":
    main()
