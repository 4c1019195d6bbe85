#!/usr/bin/env python3
"""Analysis (CPU, test libraries only): distribution of k-NN work per query on the bench workload.

    python tools/knn_stats.py [pair]

Uses tests/hostcheck (the kernels' search code compiled for the host with statistics enabled) and the
oracle's extraction on one synthetic 64x1024 scan pair; prints grid dimensions, candidates and rows
per query, and what a 64-lane wavefront pays (max over lanes) in the source set's Morton order."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import hostcheck_lib as Hc  # noqa: E402
import oracle_lib  # noqa: E402


def main():
    pair = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    orc = oracle_lib
    H, W = 64, 1024
    A = Hc.synth_scan(20240311, pair, 0, H, W, 0.01)
    B = Hc.synth_scan(20240311, pair, 1, H, W, 0.01)
    ea, pa = orc.extract_features(A, H, W, 1.0, 120.0)
    eb, pb = orc.extract_features(B, H, W, 1.0, 120.0)
    lib = Hc.lib()
    for name, tgt, src, k, R in (("plane", A[pa], B[pb], 5, 2.0), ("edge", A[ea], B[eb], 5, 1.0)):
        tgt = np.ascontiguousarray(tgt)
        src = np.ascontiguousarray(src)
        cand = np.zeros(len(src), np.uint32)
        rows = np.zeros(len(src), np.uint32)
        grid = np.zeros(4)
        blockpts = np.zeros(len(src), np.uint32)
        lib.hostcheck_knn_stats(tgt.ctypes.data_as(C.POINTER(C.c_double)), C.c_uint64(len(tgt)),
                                src.ctypes.data_as(C.POINTER(C.c_double)), C.c_uint64(len(src)), C.c_uint64(k),
                                C.c_double(R), cand.ctypes.data_as(C.POINTER(C.c_uint32)),
                                rows.ctypes.data_as(C.POINTER(C.c_uint32)), grid.ctypes.data_as(C.POINTER(C.c_double)),
                                blockpts.ctypes.data_as(C.POINTER(C.c_uint32)))
        general = rows >> 16  # (hostcheck packs the number of general rounds into the upper half)
        rows = rows & 0xFFFF
        print(f"{name}: {len(tgt)} targets, {len(src)} queries, grid {grid[:3].astype(int)} h={grid[3]:.3f} R={R}; "
              f"{(general > 0).mean() * 100:.1f} % of the queries need rounds beyond the 3x3x3 block")
        print("  candidates/query: mean %.1f median %d p90 %d p99 %d max %d" % (
            cand.mean(), np.median(cand), np.percentile(cand, 90), np.percentile(cand, 99), cand.max()))
        print("  rows/query: mean %.2f" % rows.mean())
        # Morton order of the queries (32^3 over the bbox), as the source grid build orders them
        lo, hi = src.min(0), src.max(0)
        hcell = (hi - lo).max() / 32 * (1 + 1e-9)
        c = np.clip(((src - lo) / hcell).astype(np.int64), 0, 31)
        def spread(v):
            v = (v | (v << 8)) & 0x100F
            v = (v | (v << 4)) & 0x10C3
            v = (v | (v << 2)) & 0x1249
            return v
        code = spread(c[:, 0]) | (spread(c[:, 1]) << 1) | (spread(c[:, 2]) << 2)
        order = np.argsort(code, kind="stable")
        b = (cand[order].astype(np.int64) + 3) // 4 + rows[order] // 2  # batches per lane (a partially filled batch per ~2 rows)
        nw = len(b) // 64
        wb = b[: nw * 64].reshape(nw, 64)
        print("  batches/lane mean %.1f; per wave max %.1f -> lane utilisation %.2f" % (
            b.mean(), wb.max(1).mean(), wb.mean() / wb.max(1).mean()))
        for blk in (256, 512, 1024):  # sort by (true) work inside blocks of consecutive queries only
            nb = len(b) // blk
            bb = np.sort(b[: nb * blk].reshape(nb, blk), axis=1).reshape(-1, 64)
            print("  sorted inside blocks of %d: utilisation %.2f (wave max mean %.1f)" % (blk, bb.mean() / bb.max(1).mean(), bb.max(1).mean()))
        for blk in (256, 1024):  # same, ordered by a proxy known before the loop: points in the 3x3x3 block
            nb = len(b) // blk
            pr = blockpts[order][: nb * blk].reshape(nb, blk)
            idx = np.argsort(pr, axis=1, kind="stable")
            bb = np.take_along_axis(b[: nb * blk].reshape(nb, blk), idx, axis=1).reshape(-1, 64)
            print("  ordered by block population inside blocks of %d: utilisation %.2f (wave max mean %.1f)" % (
                blk, bb.mean() / bb.max(1).mean(), bb.max(1).mean()))
        bs = np.sort(b)[: nw * 64].reshape(nw, 64)
        print("  (if queries were sorted by work: utilisation %.2f)" % (bs.mean() / bs.max(1).mean()))


if __name__ == "__main__":
    main()
