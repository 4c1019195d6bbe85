"""CPU-only: libloamx.so builds for gfx950, loads, exports every symbol include/loamx.h declares,
and refuses to compute without a device (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from loam_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    lib = capi.load()
    header = open(os.path.join(ROOT, "include", "loamx.h")).read()
    declared = sorted(set(re.findall(r"\b(loamx_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in loamx.h but not exported"
    assert sorted(capi.EXPORTS) == declared


def test_struct_layouts_match_reference_field_order():
    fe = capi.FeatureExtractionParams()
    assert [f[0] for f in fe._fields_] == ["neighbor_points", "number_sectors", "max_edge_feats_per_sector",
                                           "max_planar_feats_per_sector", "edge_feat_threshold",
                                           "planar_feat_threshold", "occlusion_thresh", "parallel_thresh"]
    assert (fe.neighbor_points, fe.number_sectors, fe.max_edge_feats_per_sector, fe.max_planar_feats_per_sector,
            fe.edge_feat_threshold, fe.planar_feat_threshold, fe.occlusion_thresh, fe.parallel_thresh) == \
        (3, 6, 10, 50, 100.0, 1.0, 0.5, 1.0)
    rp = capi.RegistrationParams()
    assert (rp.num_edge_neighbors, rp.max_edge_neighbor_dist, rp.min_line_fit_points, rp.min_line_condition_number,
            rp.num_plane_neighbors, rp.max_plane_neighbor_dist, rp.min_plane_fit_points,
            rp.max_avg_point_plane_dist, rp.max_iterations, rp.rotation_convergence_thresh,
            rp.position_convergence_thresh, rp.min_associations) == (5, 1.0, 3, 10.0, 5, 2.0, 4, 0.1, 10, 1e-3, 1e-2, 100)
    assert capi.RESULT_DTYPE.itemsize == 64  # the record gathered across ranks


def test_capacities():
    lib = capi.load()
    lidar = capi.LidarParams(64, 1024, 1.0, 120.0)
    fe = capi.FeatureExtractionParams()
    assert lib.loamx_edge_capacity(lidar, fe) == 64 * 6 * 11
    assert lib.loamx_planar_capacity(lidar, fe) == 64 * 6 * 51


def test_host_generator_matches_hostcheck():
    import hostcheck_lib as Hc
    a = capi.synth_scan_host(5, 3, 1, 8, 64, 0.01)
    b = Hc.synth_scan(5, 3, 1, 8, 64, 0.01)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    assert np.array_equal(capi.synth_pair_pose(5, 3), Hc.synth_pose(5, 3))


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.LoamxError) as e:
        capi.Context(0)
    assert e.value.status == capi.ERR_NO_DEVICE


def test_missing_rccl_library_is_an_error_code_not_a_crash():
    """the lazy RCCL loader built its message from two dlerror() calls — the second returns NULL and
    the process died in std::string. A loader pointed at a library that does not exist must come back with LOAMX_ERR_COMM."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from loam_amd import capi\n"
            "import ctypes as C\n"
            "buf = C.create_string_buffer(capi.COMM_ID_BYTES)\n"
            "print('rc', capi.load().loamx_comm_get_unique_id(buf))\n" % ROOT)
    env = dict(os.environ, LOAMX_RCCL_LIB="/nonexistent/librccl-not-here.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "rc 7" in out.stdout  # LOAMX_ERR_COMM
