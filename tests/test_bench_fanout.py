"""CPU: `python bench.py --gpus N` starts N ranks itself (torch.distributed.run as a child process) before it
imports torch or touches HIP, and the C ABI's shard arithmetic (loamx_shard_range) is the launcher's."""
import json
import os
import subprocess
import sys

import torch  # noqa: F401  (paged in before the timed subprocesses below)

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_capi_shard_range_is_the_python_launchers():
    from loam_amd import capi
    from loam_amd import distributed as D
    for total in (0, 1, 5, 8, 1000, 8192):
        for world in (1, 2, 3, 8):
            for r in range(world):
                lo, hi = D.shard_range(total, world, r)
                assert capi.shard_range(total, world, r) == (lo, hi - lo)


def test_gpus_flag_fans_out_before_torch_is_imported():
    probe = r"""
import json, subprocess, sys
sys.argv = ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "0"]
seen = {}
def fake_run(cmd, env=None, **kw):
    seen["cmd"], seen["torch_loaded"], seen["ipc"] = cmd, "torch" in sys.modules, (env or {}).get("HSA_ENABLE_IPC_MODE_LEGACY")
    class R: returncode = 7
    return R()
subprocess.run = fake_run
import bench
try:
    bench.main()
except SystemExit as e:
    seen["exit"] = e.code
print(json.dumps(seen))
"""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, "-c", probe], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    seen = json.loads(out.stdout.strip().splitlines()[-1])
    cmd = seen["cmd"]
    assert seen["torch_loaded"] is False  # the parent has not imported torch (nor initialised HIP) when it spawns
    assert seen["exit"] == 7              # the child's exit code is propagated
    assert seen["ipc"] == "0"
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "0"] and cmd[-7].endswith("bench.py")


def test_world_size_must_match_gpus_flag():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr


def test_two_ranks_really_start_and_fail_loudly_without_a_gpu():
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert out.stderr.count("no CPU fallback") >= 1  # the ranks refuse to measure anything without the HIP path


def test_multi_gpu_selfcheck_plan_and_loud_failure():
    """tools/multi_gpu_selfcheck.py: the shard plan needs no GPU (uneven totals take the grouped-broadcast path), and the
    tool refuses to report anything without one."""
    tool = os.path.join(ROOT, "tools", "multi_gpu_selfcheck.py")
    out = subprocess.run([sys.executable, tool, "--plan", "--gpus", "8", "--total-pairs", "8191"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    plan = json.loads(out.stdout.strip().splitlines()[-1])
    assert plan["covers_everything"] and plan["collective"].startswith("grouped ncclBroadcast")
    assert [s["count"] for s in plan["shards"]] == [1024] * 7 + [1023]
    out = subprocess.run([sys.executable, tool, "--plan", "--gpus", "8", "--total-pairs", "8192"], capture_output=True, text=True, timeout=120)
    assert json.loads(out.stdout.strip().splitlines()[-1])["collective"] == "ncclAllGather"
    if not torch.cuda.is_available():
        out = subprocess.run([sys.executable, tool, "--gpus", "2"], capture_output=True, text=True, timeout=300)
        assert out.returncode != 0


def test_total_pairs_flag_shards_a_fixed_total():
    """bench.py --total-pairs T (strong scaling): the help text names it and the launcher passes it through to the ranks"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert "--total-pairs" in out.stdout
