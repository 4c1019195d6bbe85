import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import hostcheck_lib as Hc, oracle_lib as orc
H, W = 64, 1024
lib = Hc.lib()
for pair in (0, 5, 100):
    A = Hc.synth_scan(20240311, pair, 0, H, W, 0.01); B = Hc.synth_scan(20240311, pair, 1, H, W, 0.01)
    ea, pa = orc.extract_features(A, H, W, 1.0, 120.0); eb, pb = orc.extract_features(B, H, W, 1.0, 120.0)
    tgt, src = np.ascontiguousarray(A[pa]), np.ascontiguousarray(B[pb])
    out = np.zeros(2 * len(src), np.uint32)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    lib.hostcheck_lean2_stats(dp(tgt), C.c_uint64(len(tgt)), dp(src), C.c_uint64(len(src)), C.c_uint64(5), C.c_double(2.0), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    r = out[0::2]; x = out[1::2]
    r2 = (r & 0xFF).astype(int)
    queued = r2 != 0xFF
    ret2 = r2[queued] - 2
    print(f"pair {pair}: {len(src)} queries, queued {queued.sum()}; W=2 returns:", {v: int((ret2 == v).sum()) for v in (-3, -2, -1)}, "done", int((ret2 >= 0).sum()))
    iso = queued & (r2 - 2 == -1)
    r4 = ((r[iso] >> 8) & 0xFF).astype(int) - 4
    trips = (x[iso] >> 8)
    print("   W=4 on the isolated:", {v: int((r4 == v).sum()) for v in (-3, -2, -1)}, "done", int((r4 >= 0).sum()), "trips mean %.1f max %d" % (trips.mean() if len(trips) else 0, trips.max() if len(trips) else 0))
