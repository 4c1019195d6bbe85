"""GPU: the reference's own unit tests re-expressed against the drop-in C++ headers
(include/loam/*.h -> C ABI -> HIP kernels), built with g++ and run as a child process."""
import os
import subprocess

import pytest

from loam_amd import build as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_shim_test():
    B.build()
    exe = os.path.join(ROOT, "tests", "cpp", "test_shim")
    src = os.path.join(ROOT, "tests", "cpp", "test_shim.cpp")
    cmd = ["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", B.LIB_DIR, "-lloamx",
           "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{B.LIB_DIR}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_shim_compiles_without_gpu():
    build_shim_test()


@pytest.mark.gpu
def test_reference_tests_through_cpp_shim():
    exe = build_shim_test()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0, out.stdout[-2000:]
    assert "0 failures" in out.stdout


def build_example():
    B.build()
    exe = os.path.join(ROOT, "tests", "cpp", "example_scan_to_scan")
    src = os.path.join(ROOT, "examples", "scan_to_scan.cpp")
    cmd = ["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", B.LIB_DIR, "-lloamx",
           "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{B.LIB_DIR}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_example_compiles_without_gpu():
    build_example()


@pytest.mark.gpu
def test_cpp_example_runs():
    out = subprocess.run([build_example()], capture_output=True, text=True, timeout=120)
    print(out.stdout[-1000:], out.stderr[-1000:])
    assert out.returncode == 0, out.stdout[-1000:]
