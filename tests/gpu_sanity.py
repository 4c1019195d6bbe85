"""First-contact script for a GPU box: HIP runtime coexistence with torch, one tiny call per path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
t0 = time.time()
import torch
print("torch", torch.__version__, "cuda available", torch.cuda.is_available(), "%.1fs" % (time.time() - t0), flush=True)
x = torch.ones(4, device="cuda") * 2
print("torch tensor on gpu:", x.sum().item(), flush=True)
import numpy as np
from loam_amd import capi
ctx = capi.Context(0)
print("ctx ok", flush=True)
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "libamdhip64" in l or "libloamx" in l or "libhsa-runtime" in l})
print("loaded:", libs, flush=True)
H, W = 16, 256
lidar = capi.LidarParams(H, W, 1.0, 120.0)
A = capi.synth_scan_host(1, 0, 0, H, W, 0.01)
c = ctx.compute_curvature(A, lidar)
print("curvature[0:8]", c[:8], flush=True)
e, p = ctx.extract_features(A, lidar)
print("features", len(e), len(p), flush=True)
# torch tensor memory passed to the library on torch's current stream
N = H * W
t = torch.empty(2 * N * 3, dtype=torch.float64, device="cuda")
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.synth_scan_pairs_dev(1, 0, 1, H, W, 0.01, t.data_ptr())
torch.cuda.synchronize()
print("device gen == host gen:", np.array_equal(t[:N * 3].cpu().numpy().reshape(N, 3), A), flush=True)
import __graft_entry__ as g
g.smoke()
