import sys, os, shutil, subprocess, numpy as np
sys.path.insert(0, '.')
# results of a 64-pair batch with each library build, in child processes
code = '''
import sys, numpy as np
sys.path.insert(0, ".")
from loam_amd import capi
c = capi.Context(0)
H, W, P = 64, 1024, 64
d_xyz, d_res = c.alloc(P * 2 * H * W * 24), c.alloc(P * 64)
c.synth_scan_pairs_dev(20240311, 900, P, H, W, 0.01, d_xyz.ptr)
c.register_scan_pairs_dev(d_xyz.ptr, P, capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams(), capi.RegistrationParams(), d_res.ptr)
c.synchronize()
np.save(sys.argv[1], d_res.download(np.uint8, P * 64))
'''
out = {}
for v in "AB":
    shutil.copy(f"loam_amd/lib/libloamx_{v}.so", "loam_amd/lib/libloamx.so")
    subprocess.check_call([sys.executable, "-c", code, f"/tmp/res_{v}.npy"])
    out[v] = np.load(f"/tmp/res_{v}.npy")
print("results bit-identical between builds A and B:", np.array_equal(out["A"], out["B"]))
