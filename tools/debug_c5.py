import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from loam_amd import capi
from gpu_common import option
c = capi.Context(0)
H5, W5 = 128, 2048
lidar = capi.LidarParams(H5, W5, 1.0, 120.0)
fe = capi.FeatureExtractionParams()
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 99
src = capi.synth_scan_host(SEED, 0, 1, H5, W5, 0.01)
e5, p5 = c.extract_features(src, lidar, fe)
maps_p, maps_e, k = [], [], 0
while sum(len(m) for m in maps_p) < 1_000_000:
    s = capi.synth_scan_host(1000 + k, 0, 0, H5, W5, 0.01)
    e, p = c.extract_features(s, lidar, fe)
    maps_p.append(s[p]), maps_e.append(s[e])
    k += 1
map_p, map_e = np.ascontiguousarray(np.concatenate(maps_p)), np.ascontiguousarray(np.concatenate(maps_e))
idx = c.target_index(map_e, map_p)
def show(tag, r):
    print(tag, r[1], r[2], np.asarray(r[0]).view(np.uint64) % 100000)
for rep in range(1):
    show("plain  ", c.register_features(src[e5], src[p5], map_e, map_p))
    show("indexed", c.register_features_indexed(idx, src[e5], src[p5]))
for opt in ():
    with option(opt, 1, c):
        show("plain   " + opt, c.register_features(src[e5], src[p5], map_e, map_p))
        show("indexed " + opt, c.register_features_indexed(idx, src[e5], src[p5]))
