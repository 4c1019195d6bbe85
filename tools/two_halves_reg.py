#!/usr/bin/env python3
"""Experiment: the REGISTRATION of the headline batch (features extracted once) as one call on one context vs as n part-batches
on n contexts driven by n host threads (the LM tail of one part beside the association of another).
python tools/two_halves_reg.py [pairs_total] [steps]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from loam_amd import capi  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
H, W = 64, 1024
N = H * W
lidar, fe, reg = capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams(), capi.RegistrationParams()
c0 = capi.Context(0)
ecap, pcap = c0.edge_capacity(lidar, fe), c0.planar_capacity(lidar, fe)
ns = 2 * total
d_xyz = c0.alloc(ns * N * 24)
c0.synth_scan_pairs_dev(20240311, 0, total, H, W, 0.01, d_xyz.ptr)
d_ei, d_ne, d_ex = c0.alloc(ns * ecap * 4), c0.alloc(ns * 4), c0.alloc(ns * ecap * 24)
d_pi, d_np, d_px = c0.alloc(ns * pcap * 4), c0.alloc(ns * 4), c0.alloc(ns * pcap * 24)
c0.extract_features_batch_dev(d_xyz.ptr, ns, lidar, fe, d_ei.ptr, d_ne.ptr, d_ex.ptr, d_pi.ptr, d_np.ptr, d_px.ptr)
c0.synchronize()
ne, npl = d_ne.download(np.uint32, ns), d_np.download(np.uint32, ns)
cnt = {}
for name, arr in (("te", ne[0::2]), ("se", ne[1::2]), ("tp", npl[0::2]), ("sp", npl[1::2])):
    b = c0.alloc(total * 4)
    b.upload(np.ascontiguousarray(arr))
    cnt[name] = b
d_res = c0.alloc(total * 64)


def reg_part(c, p0, n):  # pairs [p0, p0 + n): target scan 2p, source scan 2p + 1 of the interleaved feature arrays
    c.register_features_batch_dev(n, d_ex.ptr + (2 * p0 + 1) * ecap * 24, cnt["se"].ptr + 4 * p0, d_px.ptr + (2 * p0 + 1) * pcap * 24, cnt["sp"].ptr + 4 * p0,
                                  d_ex.ptr + 2 * p0 * ecap * 24, cnt["te"].ptr + 4 * p0, d_px.ptr + 2 * p0 * pcap * 24, cnt["tp"].ptr + 4 * p0,
                                  2 * ecap, 2 * pcap, None, reg, d_res.ptr + 64 * p0)


def timed(n_parts):
    ctxs = [c0] + [capi.Context(0) for _ in range(n_parts - 1)]
    P = total // n_parts
    for k, c in enumerate(ctxs):
        reg_part(c, k * P, P)
        c.synchronize()
    bar = threading.Barrier(n_parts + 1)

    def worker(k):
        bar.wait()
        if k:
            time.sleep(0.0015 * k)
        for _ in range(steps):
            reg_part(ctxs[k], k * P, P)
        ctxs[k].synchronize()

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(n_parts)]
    for t in ths:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    res = d_res.download(capi.RESULT_DTYPE, total)
    print(f"{n_parts} part(s) x {P} pairs: {dt * 1e3 / steps:.3f} ms per {total} pairs; converged {int((res['termination'] == 0).sum())}", flush=True)
    for c in ctxs[1:]:
        c.close()


for n in (1, 2, 1, 2, 3, 4):
    timed(n)
