#!/usr/bin/env python3
"""Headless parameter sweep for loam::extractFeatures.

The reference ships an Open3D GUI (scripts/tune_feature_extraction.py) in which the eight
FeatureExtractionParams are edited by hand and the resulting edge / planar features are inspected.
This tool answers the same question without a display: it runs extractFeatures on the GPU for a grid
of parameter values and prints, per combination, how many features of each kind come out, how they
are spread over the rings and sectors, and how many points the validity mask removes — as a table or
as JSON lines for plotting.

    python tools/tune_feature_extraction.py scan.npy --rows 64 --cols 1024 \
        --sweep edge_feat_threshold=25,50,100,200 --sweep planar_feat_threshold=0.5,1,2
    python tools/tune_feature_extraction.py --synthetic 3 --sweep neighbor_points=3,5 --json

`scan.npy`: (rows*cols, 3) float array, row-major (ring, column). Parameters that are not swept keep the
reference's defaults (loam/include/loam/features.h:37-66) unless set with --set name=value.
"""
import argparse
import itertools
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

FE_FIELDS = ["neighbor_points", "number_sectors", "max_edge_feats_per_sector", "max_planar_feats_per_sector",
             "edge_feat_threshold", "planar_feat_threshold", "occlusion_thresh", "parallel_thresh"]
INT_FIELDS = set(FE_FIELDS[:4])


def parse_assign(items, multi):
    out = {}
    for it in items or []:
        name, _, val = it.partition("=")
        if name not in FE_FIELDS:
            raise SystemExit(f"unknown parameter '{name}' (one of {', '.join(FE_FIELDS)})")
        conv = int if name in INT_FIELDS else float
        vals = [conv(v) for v in val.split(",") if v]
        if not vals or (not multi and len(vals) != 1):
            raise SystemExit(f"bad value list for '{name}'")
        out[name] = vals if multi else vals[0]
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("scan", nargs="?", help="(rows*cols, 3) .npy file")
    ap.add_argument("--synthetic", type=int, default=None, metavar="SEED", help="use the built-in synthetic room scan")
    ap.add_argument("--rows", type=int, default=64)
    ap.add_argument("--cols", type=int, default=1024)
    ap.add_argument("--min-range", type=float, default=1.0)
    ap.add_argument("--max-range", type=float, default=100.0)
    ap.add_argument("--set", action="append", metavar="NAME=VALUE", help="fix a parameter")
    ap.add_argument("--sweep", action="append", metavar="NAME=V1,V2,...", help="sweep a parameter (cartesian product)")
    ap.add_argument("--json", action="store_true", help="one JSON object per combination")
    args = ap.parse_args()

    import torch  # noqa: F401  (loads the HIP runtime libloamx.so binds to)
    from loam_amd import capi

    H, W = args.rows, args.cols
    if args.synthetic is not None:
        xyz = capi.synth_scan_host(args.synthetic, 0, 0, H, W, 0.01)
    elif args.scan:
        xyz = np.ascontiguousarray(np.load(args.scan), dtype=np.float64).reshape(-1, 3)
    else:
        raise SystemExit("give a scan file or --synthetic SEED")
    if len(xyz) != H * W:
        raise SystemExit(f"scan has {len(xyz)} points, expected rows*cols = {H * W}")
    fixed = parse_assign(args.set, multi=False)
    sweeps = parse_assign(args.sweep, multi=True)
    names = list(sweeps)
    ctx = capi.Context(0)
    lidar = capi.LidarParams(H, W, args.min_range, args.max_range)
    if not args.json:
        print(" ".join(f"{n:>28s}" for n in names) + f" {'edge':>7s} {'planar':>7s} {'valid %':>8s} {'empty sectors e/p':>18s} {'rings w/o planar':>17s}")
    for combo in itertools.product(*[sweeps[n] for n in names]):
        fe = capi.FeatureExtractionParams()
        for n, v in {**fixed, **dict(zip(names, combo))}.items():
            setattr(fe, n, v)
        try:
            edge, planar = ctx.extract_features(xyz, lidar, fe)
            valid = ctx.compute_valid_points(xyz, lidar, fe)
        except Exception as e:  # unsupported combination (kernel limits) or bad parameters
            row = {"params": dict(zip(names, combo)), "error": str(e)}
            print(json.dumps(row) if args.json else " ".join(f"{v:>28}" for v in combo) + f"  {e}")
            continue
        S = int(fe.number_sectors)
        pps = W // S if S else W

        def sector_hist(idx):
            ring, col = idx // W, idx % W
            sec = np.minimum(col // max(pps, 1), S - 1)
            h = np.zeros((H, S), dtype=np.int64)
            np.add.at(h, (ring, sec), 1)
            return h

        he, hp = sector_hist(edge.astype(np.int64)), sector_hist(planar.astype(np.int64))
        row = {"params": {**fixed, **dict(zip(names, combo))}, "n_edge": int(len(edge)), "n_planar": int(len(planar)),
               "valid_fraction": float(valid.mean()), "empty_edge_sectors": int((he == 0).sum()),
               "empty_planar_sectors": int((hp == 0).sum()), "rings_without_planar": int((hp.sum(1) == 0).sum()),
               "edge_per_ring": he.sum(1).tolist(), "planar_per_ring": hp.sum(1).tolist()}
        if args.json:
            print(json.dumps(row))
        else:
            print(" ".join(f"{v:>28}" for v in combo) + f" {row['n_edge']:7d} {row['n_planar']:7d} {100 * row['valid_fraction']:8.2f} "
                  f"{row['empty_edge_sectors']:8d}/{row['empty_planar_sectors']:<9d} {row['rings_without_planar']:17d}")
    ctx.close()


if __name__ == "__main__":
    main()
