#!/usr/bin/env python3
"""Experiment: the headline batch (1024 scan pairs) as two half-batches on two contexts driven by two host threads,
so that kernels of different phases of the pipeline (instruction-bound association next to HBM-bound sweeps / index
builds) can share the chip.   python tools/two_contexts.py [contexts] [pairs_total] [steps]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from loam_amd import capi  # noqa: E402

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
total = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
H, W = 64, 1024
N = H * W
P = total // n_ctx
lidar, fe, reg = capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams(), capi.RegistrationParams()
ctxs = [capi.Context(0) for _ in range(n_ctx)]
bufs = []
for k, c in enumerate(ctxs):
    d_xyz = c.alloc(P * 2 * N * 24)
    d_res = c.alloc(P * 64)
    c.synth_scan_pairs_dev(20240311, k * P, P, H, W, 0.01, d_xyz.ptr)
    c.synchronize()
    bufs.append((d_xyz, d_res))


def run(k, n):
    c = ctxs[k]
    for _ in range(n):
        c.register_scan_pairs_dev(bufs[k][0].ptr, P, lidar, fe, reg, bufs[k][1].ptr)
    c.synchronize()


for k in range(n_ctx):
    run(k, 1)  # warm-up
barrier = threading.Barrier(n_ctx + 1)


def worker(k):
    barrier.wait()
    if k == 1:
        time.sleep(0.003)  # start the second half out of phase
    run(k, steps)


ths = [threading.Thread(target=worker, args=(k,)) for k in range(n_ctx)]
for t in ths:
    t.start()
barrier.wait()
t0 = time.perf_counter()
for t in ths:
    t.join()
dt = time.perf_counter() - t0
res = np.concatenate([b[1].download(capi.RESULT_DTYPE, P) for b in bufs])
print(f"{n_ctx} context(s) x {P} pairs, {steps} steps: {dt*1e3/steps:.3f} ms per {total} pairs = {total*steps/dt:.0f} pairs/s; converged {int((res['termination']==0).sum())}")
