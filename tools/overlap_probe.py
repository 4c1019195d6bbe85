#!/usr/bin/env python3
"""Probe (round 6): how well does the HBM-bound extraction of one batch run beside the issue-bound registration of another?
Two contexts, two host threads: A = loamx_register_features_batch_dev on pre-extracted features of P pairs, B =
loamx_extract_features_batch_dev of 2 P scans. Each alone, then together (each repeated until both have done `reps` calls).
    python tools/overlap_probe.py [pairs] [reps]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from loam_amd import capi  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
H, W = 64, 1024
N = H * W
lidar, fe, reg = capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams(), capi.RegistrationParams()
ca, cb = capi.Context(0), capi.Context(0)
ns = 2 * P
d_xyz = ca.alloc(ns * N * 24)
ca.synth_scan_pairs_dev(20240311, 0, P, H, W, 0.01, d_xyz.ptr)
ecap, pcap = ca.edge_capacity(lidar, fe), ca.planar_capacity(lidar, fe)


def feature_buffers(c):
    return dict(ei=c.alloc(ns * ecap * 4), pi=c.alloc(ns * pcap * 4), ne=c.alloc(ns * 4), np=c.alloc(ns * 4), ex=c.alloc(ns * ecap * 24), px=c.alloc(ns * pcap * 24))


fa, fb = feature_buffers(ca), feature_buffers(cb)
d_res = ca.alloc(P * 64)


def extract(c, f):
    c.extract_features_batch_dev(d_xyz.ptr, ns, lidar, fe, f["ei"].ptr, f["ne"].ptr, f["ex"].ptr, f["pi"].ptr, f["np"].ptr, f["px"].ptr)


def register(c, f):  # pairs = (scan 2p target, scan 2p + 1 source): sets one scan apart, pitch 2 scans -> use in_pitch via strides
    # (register_features_batch_dev takes separate arrays with pitch 1: pass the even / odd scans as strided views is not
    # possible, so register every scan against its neighbour: source = scans 1.., target = scans 0.. for P pairs of adjacent scans)
    c.register_features_batch_dev(P, f["ex"].ptr + ecap * 24, f["ne"].ptr + 4, f["px"].ptr + pcap * 24, f["np"].ptr + 4,
                                  f["ex"].ptr, f["ne"].ptr, f["px"].ptr, f["np"].ptr, ecap, pcap, None, reg, d_res.ptr)


extract(ca, fa), ca.synchronize()
register(ca, fa), ca.synchronize()
extract(cb, fb), cb.synchronize()


def timed(fn, n):
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n


def run_a():
    register(ca, fa)
    ca.synchronize()


def run_b():
    extract(cb, fb)
    cb.synchronize()


ta, tb = timed(run_a, reps), timed(run_b, reps)
print(f"alone: registration of {P} pairs {ta*1e3:.3f} ms, extraction of {ns} scans {tb*1e3:.3f} ms, sum {1e3*(ta+tb):.3f} ms")
out = {}
bar = threading.Barrier(3)


def worker(name, fn):
    bar.wait()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    out[name] = (time.perf_counter() - t0) / reps


ths = [threading.Thread(target=worker, args=("a", run_a)), threading.Thread(target=worker, args=("b", run_b))]
for t in ths:
    t.start()
bar.wait()
t0 = time.perf_counter()
for t in ths:
    t.join()
wall = time.perf_counter() - t0
print(f"together ({reps} calls each): registration {out['a']*1e3:.3f} ms per call, extraction {out['b']*1e3:.3f} ms per call, wall {wall*1e3/reps:.3f} ms per (registration + extraction)")
